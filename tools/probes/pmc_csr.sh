#!/bin/bash
# HBM traffic and wait counters of the CSR kernels at 10^6 (tools/run_million.py under rocprofv3 --pmc) -> gpurun_out/pmc_csr.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/pmc_csr
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
    tag=$(echo "$c" | cut -d' ' -f1)
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/$tag" -o pmc -- python3 "$R/tools/run_million.py" > /dev/null 2> "$O/$tag.log"
done
cd "$R"
python3 - "$O" <<'PY' > gpurun_out/pmc_csr.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0]
        if "k_lower" in n or "precheck" in n or "k_neighbors_rows" in n:
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in acc.items():
    print(n)
    for c, v in sorted(cs.items()):
        print("   %-28s mean per dispatch %.4g  (%d dispatches)" % (c, sum(v) / len(v), len(v)))
PY
cat gpurun_out/pmc_csr.txt
rm -rf "$O"
