#!/bin/bash
# per-round durations of the second loop's kernels at 10^6 (rocprofv3 kernel trace of tools/run_million.py) -> gpurun_out/loop_rounds.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/loop_trace
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O" -o m -- python3 "$R/tools/run_million.py" $FLAGS > "$O/run.json" 2> "$O/run.log"
cd "$R"
python3 - "$O" <<'PY' > gpurun_out/loop_rounds.txt
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "k_loop_" in n or "precheck" in n:
        per[n.split("(")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for n, v in per.items():
    d = [x[1] / 1e3 for x in v]
    print(n, "calls", len(d), "total ms %.2f" % (sum(d) / 1e3), "first 12 (us):", [round(x) for x in d[:12]], "last 6:", [round(x) for x in d[-6:]])
# the last call's loop: gaps between consecutive loop kernels
loop = sorted([(s, s + d, n) for n, v in per.items() if "k_loop_" in n for s, d in v])
half = loop[len(loop) // 2:]
busy = sum(e - s for s, e, _ in half); span = half[-1][1] - half[0][0]
print("second call: loop kernels busy %.2f ms of a %.2f ms span (%d launches)" % (busy / 1e6, span / 1e6, len(half)))
PY
cat gpurun_out/loop_rounds.txt
rm -rf "$O"
