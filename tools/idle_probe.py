#!/usr/bin/env python3
"""Does the plain 1e5 pass run slower after the GPU sat idle for a few milliseconds (as it does between two clustering calls)?
Times the pass back to back, with host sleeps between passes, and on a non-default stream."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hammock_amd
from hammock_amd import _native
from hammock_amd.synth import synth_peptides
from bench import load_blosum62
n = 100000
res, off = synth_peptides(1, n, 12)
dev = torch.device("cuda", 0)
S = _native.HMK_EDGE_SHARDS
cap = (1 << 26) // S * S
d_edges = torch.empty(cap, dtype=torch.int64, device=dev)
d_counts = torch.zeros(S, dtype=torch.int64, device=dev)
ctx = hammock_amd.Context(load_blosum62(), device=0)
ctx.set_sequences(residues=res, offsets=off)
def run(stream, gap_ms, reps=12):
    ms = []
    for _ in range(reps):
        if gap_ms: time.sleep(gap_ms * 1e-3)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        ctx.neighbors_shifted_dev(3, 0, 20, 0, 1, d_edges.data_ptr(), cap, d_counts.data_ptr(), stream.cuda_stream)
        b.record(stream)
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    return round(float(np.median(ms[3:])), 3)
out = {}
cur = torch.cuda.current_stream(dev)
side = torch.cuda.Stream(dev)
for _ in range(20): ctx.neighbors_shifted_dev(3, 0, 20, 0, 1, d_edges.data_ptr(), cap, d_counts.data_ptr(), cur.cuda_stream)
torch.cuda.synchronize()
out["default stream, back to back"] = run(cur, 0)
for g in (1, 3, 10, 30):
    out[f"default stream, {g} ms idle before each pass"] = run(cur, g)
out["side stream, back to back"] = run(side, 0)
out["side stream, 3 ms idle"] = run(side, 3)
print(json.dumps(out, indent=1))
