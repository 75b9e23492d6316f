#!/usr/bin/env python3
"""Is the scoring kernel slower inside a clustering call than in the bench loop because the GPU starts it from idle clocks?
Alternates calls after 20 ms of idling with calls right after 30 matmuls (measured: score_ms 4.57 against 4.05 at 10^5)."""
import sys, os, json, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import hammock_amd
from hammock_amd.synth import synth_peptides
from bench import load_blosum62
res, off = synth_peptides(1, 100000, 12)
ctx = hammock_amd.Context(load_blosum62(), device=0)
ctx.set_sequences(residues=res, offsets=off)
x = torch.randn(4096, 4096, device="cuda:0")
for mode in ("cold", "warm", "cold", "warm"):
    for rep in range(3):
        time.sleep(0.02)
        if mode == "warm":
            for _ in range(30): y = x @ x
            torch.cuda.synchronize()
        t = time.perf_counter()
        ctx.greedy_cluster(3, 0, 20, 2500)
        w = (time.perf_counter() - t) * 1e3
        ph = ctx.greedy_phases()
        print(mode, round(w, 2), round(ph["score_ms"], 2), round(ph["csr_ms"], 2), round(ph["device_loop_ms"], 2))
