#!/usr/bin/env python3
"""Repeats hmk_greedy_cluster on one resident context and compares every result with the first (ids, list order, member order):
    python tools/soak_greedy.py 100000 300 [--sorted] [--devices=0,0]
The band's lists come out of the device in a different order every call (atomics); the clustering must not."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hammock_amd
from hammock_amd.synth import synth_peptides
M = np.asarray(json.load(open(os.path.join(ROOT, "tests", "golden", "matrices.json")))["matrices"]["blosum62"], dtype=np.int32)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n, reps = int(args[0]), int(args[1])
devices = 0
for a in sys.argv[1:]:
    if a.startswith("--devices="): devices = [int(v) for v in a.split("=", 1)[1].split(",")]
res, off = synth_peptides(1, n, 12, 12)
if "--sorted" in sys.argv:
    letters = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)[res].reshape(n, 12)
    res = np.ascontiguousarray(res.reshape(n, 12)[np.lexsort(letters.T[::-1])[::-1]]).reshape(-1)
ctx = hammock_amd.Context(M, device=devices)
ctx.set_sequences(residues=res, offsets=off)
maxc = int(np.floor(n * 0.025 + 0.5))
cid0, order0, _ = ctx.greedy_cluster(3, 0, 20, maxc)
rank0 = ctx.member_rank[:n].copy()
bad = 0
t = time.perf_counter()
for k in range(reps):
    cid, order, _ = ctx.greedy_cluster(3, 0, 20, maxc)
    if not (np.array_equal(cid, cid0) and np.array_equal(order, order0) and np.array_equal(ctx.member_rank[:n], rank0)):
        bad += 1
print(json.dumps({"n": n, "calls": reps, "different_from_the_first": bad, "order": "size" if "--sorted" in sys.argv else "input", "devices": devices,
                  "seconds": time.perf_counter() - t}))
sys.exit(1 if bad else 0)
