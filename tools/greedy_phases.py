#!/usr/bin/env python3
"""Per-phase timing of hmk_greedy_cluster (hmk_greedy_last_phases) on synthetic 12-mers.

    python tools/greedy_phases.py 100000 1000000 > gpurun_out/greedy_phases.jsonl

For every n: a fresh context, three calls (first = buffers being sized, then steady state); prints one JSON line
per call.  Used for profiles/round2_greedy_phases.jsonl."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hammock_amd  # noqa: E402
from hammock_amd.synth import synth_peptides  # noqa: E402

with open(os.path.join(ROOT, "tests", "golden", "matrices.json")) as fh:
    M = np.asarray(json.load(fh)["matrices"]["blosum62"], dtype=np.int32)

# --sorted: the sequences in Hammock's default order ("size": count descending, then the sequence string DESCENDING,
# UniqueSequence.java:238-248 -- with all counts 1 that is reverse alphabetical, what `hammock-hip greedy` clusters by
# default); without it the order in which the generator produced them (-R input)
SORTED = "--sorted" in sys.argv
DEVICES = 0          # --devices=0,0: a multi-device context (hmk_create_multi), e.g. the one GPU twice
for a in sys.argv[1:]:
    if a.startswith("--devices="):
        DEVICES = [int(v) for v in a.split("=", 1)[1].split(",")]
LETTERS = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
for n in [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [100000]:
    lo, hi = (12, 12)
    res, off = synth_peptides(1, n, lo, hi)
    if SORTED:
        rows = LETTERS[res].reshape(n, 12)
        order_idx = np.lexsort(rows.T[::-1])[::-1]     # descending by the sequence's letters
        res = np.ascontiguousarray(res.reshape(n, 12)[order_idx]).reshape(-1)
    maxc = int(np.floor(n * 0.025 + 0.5))
    ctx = hammock_amd.Context(M, device=DEVICES)
    ctx.set_sequences(residues=res, offsets=off)
    ref = None
    for call in range(3):
        t = time.perf_counter()
        cid, order, st = ctx.greedy_cluster(3, 0, 20, maxc)
        wall = time.perf_counter() - t
        if ref is None:
            ref = (cid.copy(), order.copy())
        assert np.array_equal(cid, ref[0]) and np.array_equal(order, ref[1])
        line = {"n": n, "order": "size" if SORTED else "input", "call": call, "wall_ms": wall * 1e3, "clusters": int(st.n_multi), "result_list": int(st.n_result_clusters),
                "edges": int(st.n_edges), "phase1_stop_index": int(st.phase1_stop_index)}
        line.update(ctx.greedy_phases())
        print(json.dumps(line), flush=True)
    if "--tight" in sys.argv:   # 20 calls with nothing between them: what a resident caller that issues calls back to back sees
        walls, scores = [], []
        for call in range(20):
            t = time.perf_counter()
            ctx.greedy_cluster(3, 0, 20, maxc)
            walls.append((time.perf_counter() - t) * 1e3)
            scores.append(ctx.greedy_phases()["score_ms"])
        print(json.dumps({"n": n, "order": "size" if SORTED else "input", "tight_calls": 20, "wall_ms_median": float(np.median(walls)), "wall_ms_min": min(walls),
                          "wall_ms_all": [round(w, 3) for w in walls], "score_ms_median": float(np.median(scores))}), flush=True)
    ctx.close()
