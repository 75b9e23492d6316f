import json, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import hammock_amd
from hammock_amd.synth import synth_peptides
from oracle import c_oracle
mats = {k: np.asarray(v, dtype=np.int32) for k, v in json.load(open(os.path.join(ROOT, "tests", "golden", "matrices.json")))["matrices"].items()}
names = sorted(mats)
rng = np.random.default_rng(20261006)
for trial in range(295):
    M = mats["blosum62"] if trial % 3 else mats[names[int(rng.integers(len(names)))]]
    lo = int(rng.integers(6, 14)); hi = int(min(32, lo + rng.integers(0, 9))); n = int(rng.integers(2, 4000))
    res, off = synth_peptides(int(rng.integers(1, 10 ** 6)), n, lo, hi)
    if trial % 2 == 0:
        peps = [res[off[k]:off[k + 1]].copy() for k in range(n)]
        for k in range(n // 4, n):
            src = peps[int(rng.integers(0, max(1, n // 4)))].copy()
            for _ in range(int(rng.integers(1, 4))):
                src[int(rng.integers(len(src)))] = rng.integers(0, 20)
            peps[k] = src
        peps = list({bytes(q): q for q in peps}.values()); n = len(peps)
        res, off = hammock_amd.pack_sequences(peps)
    sizes = None
    if trial % 3 != 1:
        sizes = np.ones(n, dtype=np.int32); pick = rng.random(n) < 0.3
        sizes[pick] = 1 + rng.integers(0, 5, int(pick.sum()))
    L = np.diff(off.astype(np.int64))
    thr = int(round(L.mean() * 1.7)) + int(rng.integers(-9, 6))
    X = int(min(max(0, round(L.mean() / 4)), L.min() - 1)); p = int(rng.choice([0, 0, -1, -2]))
print("trial", trial, "n", n, "len", lo, hi, "X", X, "p", p, "thr", thr, "matrix", "blosum62" if trial % 3 else "other", "sizes", sizes is not None)
np.savez("/tmp/clink_case.npz", M=M, res=res, off=off, sizes=sizes if sizes is not None else np.zeros(0, np.int32), X=X, p=p, thr=thr)
st, ocid, oorder, orank, ostats = c_oracle.clinkage_cluster(M, res, off, sizes, X, p, thr, 16)
ctx = hammock_amd.Context(M, device=0)
ctx.set_sequences(residues=res, offsets=off, sizes=sizes)
cid, order, stats = ctx.clinkage_cluster(X, p, thr)
print("oracle merges", ostats.merges, "gpu merges", stats.merges, "cid equal", np.array_equal(cid, ocid), "order equal", np.array_equal(order, oorder), "rank equal", np.array_equal(ctx.member_rank[:n], orank))
print("n diff cid", int((cid != ocid).sum()), "first diffs", np.flatnonzero(cid != ocid)[:10])
# neighbour graph: GPU vs oracle (all pairs)
edges, _ = ctx.neighbors_shifted(X, p, thr)
ge = np.sort(np.asarray(edges, dtype=np.uint64))
rows = np.arange(n, dtype=np.uint32)
out = []
for r0 in range(0, n, 512):
    rr = rows[r0:r0 + 512]
    st2, sc = c_oracle.score_block(M, res, off, rows, rr, 0, X, p)
    mm, xx = np.meshgrid(rows, rr, indexing="ij")
    keep = (sc >= thr) & (mm > xx)
    out.append(hammock_amd.pack_edges(xx[keep], mm[keep], sc[keep]))
oe = np.sort(np.concatenate(out).astype(np.uint64))
print("edges gpu", len(ge), "oracle", len(oe), "equal", np.array_equal(ge, oe))
# clinkage from the ORACLE's edges through the host chain
cid2, order2, st2 = ctx.clinkage_from_edges(oe, thr)
print("from oracle edges: cid equal oracle", np.array_equal(cid2, ocid), "equal gpu", np.array_equal(cid2, cid))
