import sys, os, time, json, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hammock_amd
from hammock_amd.synth import synth_peptides
from bench import load_blosum62
n=int(sys.argv[1])
res, off = synth_peptides(1, n, 12)
ctx = hammock_amd.Context(load_blosum62(), device=0)
ctx.set_sequences(residues=res, offsets=off)
for _ in range(3):
    t=time.perf_counter()
    cid, order, st = ctx.greedy_cluster(3, 0, 20, int(n*0.025+0.5))
    print("wall", time.perf_counter()-t, st.neighbors_ms, st.greedy_ms, file=sys.stderr)
