#!/usr/bin/env python3
"""Phase 1 of the merge at 10^6 for several window sizes / thread counts (HMK_PHASE1_WINDOW, HMK_PHASE1_THREADS):
    python tools/phase1_probe.py W=16 W=32 T=4 T=16,W=64 [--sorted]
(--sorted: the reference's default order) prints phase1_ms (incl. the wait for the band rows), wait_rows_ms and the call's total per setting."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SORTED = ["--sorted"] if "--sorted" in sys.argv else []
for spec in [a for a in sys.argv[1:] if a != "--sorted"] or ["T=8"]:
    env = dict(os.environ)
    for kv in spec.split(","):
        k, v = kv.split("=")
        env["HMK_PHASE1_WINDOW" if k == "W" else "HMK_PHASE1_THREADS"] = v
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "greedy_phases.py"), "1000000"] + SORTED, env=env, capture_output=True, text=True).stdout
    d = json.loads(out.strip().split("\n")[-1])
    print(spec, {k: round(d[k], 1) for k in ("phase1_ms", "wait_rows_ms", "score_ms", "total_ms")}, flush=True)
