#!/bin/bash
# kernel timeline of the LAST of three clustering calls at 1e5 (default order): start offsets and durations, microseconds
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/trace_greedy
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
env $1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$O/t" -o g -- python3 "$R/tools/greedy_phases.py" --sorted ${2:-100000} > "$O/out.txt" 2> "$O/err.txt"
cd "$R"
python3 - "$O" <<'PY'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "t/**/*kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# split into calls at gaps > 0.5 ms
calls, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - cur[-1][1] > 500_000 and "k_neighbors" in r[2]:
        calls.append(cur); cur = []
    cur.append(r)
calls.append(cur)
last = calls[-1]
t0 = last[0][0]
loop = [r for r in last if "k_loop" in r[2]]
for s, e, name in last:
    if "k_loop" in name: continue
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {name[:90]}")
if loop:
    print(f"{(loop[0][0] - t0) / 1e3:9.1f} us .. {(loop[-1][1] - t0) / 1e3:9.1f} us: {len(loop)} k_loop_* kernels, {sum(e - s for s, e, _ in loop) / 1e3:.1f} us busy")
PY
