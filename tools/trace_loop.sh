#!/bin/bash
# kernel timeline of the second loop of the LAST of three clustering calls: tools/trace_loop.sh antibodies | <n> [env]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/trace_loop
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
if [ "$1" = "antibodies" ]; then CMD="$R/tools/greedy_phases_fasta.py $R/tests/golden/antibodies.fa.gz"; else CMD="$R/tools/greedy_phases.py --sorted $1"; fi
env ${2:-X=1} timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$O/t" -o g -- python3 $CMD > "$O/out.txt" 2> "$O/err.txt"
cd "$R"
python3 - "$O" <<'PY'
import csv, glob, sys, os, collections
f = glob.glob(os.path.join(sys.argv[1], "t/**/*kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
idx = [i for i, r in enumerate(rows) if "k_loop_init" in r[2]]
last = rows[idx[-1]:]
loop = [r for r in last if "k_loop_eval_first" in r[2] or "k_loop_accept" in r[2] or "k_loop_apply" in r[2] or "k_loop_first" in r[2]]
t0, t1 = loop[0][0], loop[-1][1]
busy = sum(e - s for s, e, _ in loop)
print(f"{len(loop)} round kernels over {(t1 - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us ({busy / (t1 - t0):.2f})")
d = collections.defaultdict(list)
for s, e, n in loop: d[n.split("(")[0].split("<")[0][-20:]].append((e - s) / 1e3)
for k, v in d.items(): print(k, len(v), "avg", round(sum(v) / len(v), 2), "us, max", round(max(v), 1))
gaps = [(loop[i + 1][0] - loop[i][1]) / 1e3 for i in range(len(loop) - 1)]
print("gaps between consecutive round kernels: avg", round(sum(gaps) / len(gaps), 2), "us, max", round(max(gaps), 1), "; over 10 us:", sum(1 for g in gaps if g > 10))
PY
