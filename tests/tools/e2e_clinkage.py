#!/usr/bin/env python3
"""clinkage mode end to end (Hammock.java:449-464, the span of "Ready. Clustering time"): hmk_clinkage_cluster on the GPU
against the oracle's ClinkageSequenceClusterer restatement (C form, all usable threads), identical clusters asserted.
MUSI (the reference's example, 2,457 peptides, all defaults) and 10^4 synthetic 12-mers (the largest input Hammock's
`full` mode still sends to clinkage).  One JSON line per input."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hammock_amd  # noqa: E402
from bench import load_blosum62, usable_cores  # noqa: E402
from hammock_amd.synth import synth_peptides  # noqa: E402
from oracle import c_oracle  # noqa: E402
from oracle import hammock_oracle as po  # noqa: E402

M = load_blosum62()
threads = min(usable_cores()[0], 64)
seqs = po.load_unique_sequences_from_fasta(os.path.join(ROOT, "tests", "golden", "musi.fa"))
inputs = [("examples/MUSI (2,457 12-mers), defaults thr 20, X 3",) + hammock_amd.pack_sequences([s.get_sequence_string() for s in seqs])]
inputs.append(("10^4 synthetic 12-mers, thr 20, X 3",) + synth_peptides(3, 10000, 12))
for name, res, off in inputs:
    ctx = hammock_amd.Context(M, device=0)
    ctx.set_sequences(residues=res, offsets=off)
    ctx.clinkage_cluster(3, 0, 20)                       # first call sizes the buffers
    t = time.perf_counter()
    cid, order, stats = ctx.clinkage_cluster(3, 0, 20)
    t_gpu = time.perf_counter() - t
    t = time.perf_counter()
    st, ocid, oorder, orank, ostats = c_oracle.clinkage_cluster(M, res, off, None, 3, 0, 20, threads)
    t_cpu = time.perf_counter() - t
    same = bool(st == 0 and np.array_equal(cid, ocid) and np.array_equal(order, oorder) and np.array_equal(ctx.member_rank[:len(cid)], orank))
    print(json.dumps({"input": name, "n": len(off) - 1, "gpu_clinkage_s": t_gpu, "gpu_scoring_ms": stats.neighbors_ms,
                      "host_chain_ms": stats.chain_ms, "edges": int(stats.n_edges), "merges": int(stats.merges),
                      "clusters": int(stats.n_result_clusters), "cpu_port_s": t_cpu, "cpu_threads": threads,
                      "cpu_sequenceScore_calls": int(ostats.score_calls), "identical": same}), flush=True)
    assert same
