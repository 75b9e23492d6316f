#!/usr/bin/env python3
"""How often does the JVM's HashSet iteration order change what `clinkage` returns?  ClinkageSequenceClusterer takes the start
of every nearest-neighbour chain from activeClusters.iterator().next() and returns readyClusters in iteration order
(ClinkageSequenceClusterer.java:70,118-123); Java 8 changed java.util.HashMap (hash spreading, tail insertion, order-preserving
resize), the reference is a Java 1.7 project.  The C oracle emulates all three orders (8 = Java 8+, 7 = JDK 7u6+, 6 = JDK 6 /
early 7); this tool runs random tied inputs (small alphabets, counts) through all of them and counts the differences:
the PARTITION (which sequences share a cluster), the cluster ids, the returned list's order, the crash-parity status.
CPU only.  Usage: python tests/tools/fuzz_hashset_orders.py [cases] [seed0]  -> one JSON line"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import random_peptides  # noqa: E402
from oracle import c_oracle as co  # noqa: E402

with open(os.path.join(ROOT, "tests", "golden", "matrices.json")) as fh:
    M = np.asarray(json.load(fh)["matrices"]["blosum62"], dtype=np.int32)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def partition(cid):
    """canonical form of the clustering: every sequence -> the smallest index in its cluster"""
    first = {}
    return tuple(first.setdefault(int(c), k) for k, c in enumerate(cid))


tally = {"cases": 0, "with_merges": 0, "status_differs_7": 0, "status_differs_6": 0, "partition_differs_7": 0, "partition_differs_6": 0,
         "ids_differ_7": 0, "ids_differ_6": 0, "list_order_differs_7": 0, "list_order_differs_6": 0, "7_differs_from_6": 0}
for it in range(cases):
    rng = np.random.default_rng(seed0 + it)
    n = int(rng.integers(2, 260))
    peps = random_peptides(rng, n, 6, 12, alphabet=int(rng.integers(3, 8)))
    sizes = rng.integers(1, 5, size=n).astype(np.int32) if it % 2 else None
    res, off = co.pack(peps)
    X, p, thr = int(rng.integers(0, 4)), -int(rng.integers(0, 3)), int(rng.integers(8, 30))
    X = min(X, min(len(q) for q in peps) - 1)
    out = {}
    for v in (8, 7, 6):
        co.set_java_hashset(v)
        st, cid, order, rank, stats = co.clinkage_cluster(M, res, off, sizes, X, p, thr, 1)
        out[v] = (st, None if st else partition(cid), None if st else tuple(cid.tolist()), None if st else tuple(order.tolist()), stats.merges)
    co.set_java_hashset(8)
    tally["cases"] += 1
    tally["with_merges"] += int(out[8][0] == 0 and out[8][4] > 0)
    for v in (7, 6):
        tally[f"status_differs_{v}"] += int(out[v][0] != out[8][0])
        if out[v][0] == 0 and out[8][0] == 0:
            tally[f"partition_differs_{v}"] += int(out[v][1] != out[8][1])
            tally[f"ids_differ_{v}"] += int(out[v][2] != out[8][2])
            tally[f"list_order_differs_{v}"] += int(out[v][3] != out[8][3])
    tally["7_differs_from_6"] += int(out[7][:4] != out[6][:4])
tally["note"] = ("random peptides of length 6-12 over 3-7 letters, 2-260 sequences, counts 1-4 on every other case, BLOSUM62, max shift 0-3, "
                 "shift penalty 0..-2, threshold 8-29; differences are against the Java 8+ order")
print(json.dumps(tally))
