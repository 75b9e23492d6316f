import sys, os, json, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import hammock_amd
from hammock_amd.synth import synth_peptides
from bench import load_blosum62
res, off = synth_peptides(1, 100000, 12)
ctx = hammock_amd.Context(load_blosum62(), device=0)
ctx.set_sequences(residues=res, offsets=off)
for thr in (17, 14):
    for call in range(3):
        t = time.perf_counter()
        cid, order, st = ctx.greedy_cluster(3, 0, thr, 2500)
        w = (time.perf_counter() - t) * 1e3
        ph = ctx.greedy_phases()
        print(thr, call, round(w, 1), "edges", int(st.n_edges), "score", round(ph["score_ms"], 1), "csr", round(ph["csr_ms"], 1), "p1", round(ph["phase1_ms"], 1), "pre", round(ph["precheck_ms"], 1), "loop", round(ph["device_loop_ms"], 1), ph["loop_rounds"], "cand", ph["cand_entries"])
