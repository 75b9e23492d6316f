#!/bin/bash
# PMC passes of the plain neighbour pass over 1e5 uniform L-mers (run on the GPU box): LDS / VALU activity and wait counters.
#   gpurun -- 'bash tools/pmc_lx.sh L X thr [tag]'   -> gpurun_out/pmc_lx/<tag>.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$1; X=$2; THR=$3; TAG=${4:-L${1}_X${2}_thr${3}}
O=$R/gpurun_out/pmc_lx/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for c in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH"; do
    tag=$(echo "$c" | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/pmc_$tag" -o pmc -- \
        python3 "$R/tools/run_lx.py" $L $X $THR 3 > /dev/null 2> "$O/pmc_$tag.log" || { tail -3 "$O/pmc_$tag.log"; }
done
cd "$R"
python3 - "$O" $L $X <<'PY'
import csv, glob, json, os, sys
out, L, X = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
acc = {}
for path in sorted(glob.glob(os.path.join(out, "pmc_*/**/*counter_collection.csv"), recursive=True)):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if "k_neighbors" in row["Kernel_Name"]:
                acc.setdefault((row["Kernel_Name"][:60], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
res = {}
for (k, c), v in acc.items():
    res.setdefault(k, {})[c] = sum(v) / len(v)
pairs = 100000 * 99999 // 2
for k, d in res.items():
    if "GRBM_GUI_ACTIVE" in d:
        cyc = d["GRBM_GUI_ACTIVE"] / 8
        d["_cycles_per_xcd"] = cyc
        if "SQ_LDS_IDX_ACTIVE" in d: d["_lds_busy"] = d["SQ_LDS_IDX_ACTIVE"] / (cyc * 256)
        if "SQ_ACTIVE_INST_VALU" in d: d["_valu_busy"] = d["SQ_ACTIVE_INST_VALU"] / (cyc * 256)
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD"):
        if c in d: d["_" + c.lower() + "_per_512_pairs"] = d[c] / (pairs / 512)
print(json.dumps(res, indent=1))
json.dump(res, open(out + ".json", "w"), indent=1)
PY
rm -rf "$O"
