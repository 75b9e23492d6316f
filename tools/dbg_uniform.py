#!/usr/bin/env python3
"""Debug: uniform L-mers at a dense threshold, GPU edges vs oracle; prints what differs.  python tools/dbg_uniform.py L X thr [n]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hammock_amd
from hammock_amd.synth import synth_peptides
from oracle import c_oracle
from bench import load_blosum62
L, X, thr = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 1400
M = load_blosum62()
res, off = synth_peptides(L, n, L)
ctx = hammock_amd.Context(M, device=0)
ctx.set_sequences(residues=res, offsets=off)
edges, stats = ctx.neighbors_shifted(X, 0, thr)
ii, jj = np.triu_indices(n, 1)
st, sc = c_oracle.score_pairs(M, res, off, jj.astype(np.uint32), ii.astype(np.uint32), 0, X, 0)
keep = sc >= thr
want = np.sort(hammock_amd.pack_edges(ii[keep], jj[keep], sc[keep]))
got = np.sort(np.asarray(edges, dtype=np.uint64))
print("n_tiles", stats.n_tiles, "rows", stats.classes_rows, "got", len(got), "want", len(want), "equal", bool(np.array_equal(got, want)))
only_got = np.setdiff1d(got, want); only_want = np.setdiff1d(want, got)
print("only_got", len(only_got), "only_want", len(only_want))
for name, arr in (("got", only_got[:12]), ("want", only_want[:12])):
    x, m, s = hammock_amd.edge_fields(arr)
    print(name, list(zip(x.tolist(), m.tolist(), s.tolist())))
if len(only_got):
    gx, gm, gs = hammock_amd.edge_fields(only_got); wx, wm, ws = hammock_amd.edge_fields(only_want)
    os.makedirs("gpurun_out/dbg", exist_ok=True)
    np.savez("gpurun_out/dbg/mismatch_%d_%d_%d_%d.npz" % (L, X, thr, n), gx=gx, gm=gm, gs=gs, wx=wx, wm=wm, ws=ws, got=got, want=want)
