#!/usr/bin/env python3
"""Timeline of one mixed-length pass (BASELINE config 4a) from a rocprofv3 kernel trace:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/t4a -o t -- python3 $REPO/tools/run_config4a.py && python3 $REPO/tools/trace_config4a.py /tmp/t4a
Takes the LAST pass in the trace (a run of k_neighbors_* kernels between two fills of the counters): span, the sum of the kernels' durations, how many kernels are
resident over time, and every kernel with its start relative to the pass."""
import csv, glob, os, sys
path = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
# passes: split at fill kernels
passes, cur = [], []
for s, e, name, q in rows:
    if "k_neighbors" in name:
        cur.append((s, e, name, q))
    elif cur and "fill" in name.lower():
        passes.append(cur); cur = []
if cur: passes.append(cur)
p = passes[-2] if len(passes) > 1 else passes[-1]
t0 = min(s for s, _, _, _ in p); t1 = max(e for _, e, _, _ in p)
span = (t1 - t0) / 1e6
total = sum(e - s for s, e, _, _ in p) / 1e6
ev = sorted([(s, 1) for s, _, _, _ in p] + [(e, -1) for _, e, _, _ in p])
res_time = {}
k, last = 0, t0
for t, d in ev:
    res_time[k] = res_time.get(k, 0) + (t - last); last = t; k += d
print(f"kernels {len(p)}  span {span:.3f} ms  sum of durations {total:.3f} ms  resident-kernel histogram (ms): " + ", ".join(f"{k}: {v / 1e6:.3f}" for k, v in sorted(res_time.items())))
for s, e, name, q in p:
    short = name[name.find("k_neighbors"):][:46]
    print(f"  +{(s - t0) / 1e6:6.3f} .. +{(e - t0) / 1e6:6.3f}  {(e - s) / 1e6:6.3f} ms  queue {q}  {short}")
