"""The N > 1 path.  CPU: world_size-2 gloo run of the exchange + merge + broadcast with the
ranks' edge shards supplied by the oracle scorer (the product has no CPU scorer).  GPU: two
ranks sharing the one MI355X of the box over RCCL, end to end through the kernel."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, random_peptides


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cpu_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import json
    import hammock_amd
    from hammock_amd import dist as hd
    from oracle import c_oracle
    with open(os.path.join(ROOT, "tests", "golden", "matrices.json")) as fh:
        M = np.asarray(json.load(fh)["matrices"]["blosum62"], dtype=np.int32)
    rng = np.random.default_rng(42)
    peps = random_peptides(rng, 600, 12, 12, alphabet=5)
    res, off = c_oracle.pack(peps)
    n, X, p, thr, maxc = len(peps), 3, 0, 20, 15
    # this rank's row blocks (16 rows each, cyclic), scored by the ORACLE
    rows = np.array([r for r in range(n) if (r // 16) % world == rank], dtype=np.uint32)
    xs, ms, ss = [], [], []
    for x in rows:
        m = np.arange(x + 1, n, dtype=np.uint32)
        st, sc = c_oracle.score_pairs(M, res, off, m, np.full(len(m), x, np.uint32), 0, X, p)
        keep = sc >= thr
        xs.append(np.full(int(keep.sum()), x)); ms.append(m[keep]); ss.append(sc[keep])
    local = hammock_amd.pack_edges(np.concatenate(xs), np.concatenate(ms), np.concatenate(ss))
    local_t = torch.from_numpy(local.view(np.int64).copy())
    allv = hd.all_gather_edges(local_t)
    ctx = hammock_amd.Context(M, device=-1)
    ctx.set_sequences(residues=res, offsets=off)
    cid, order, info = hd.merge_and_broadcast(ctx, allv, True, thr, maxc)
    st, ocid, oorder, _ = c_oracle.greedy_cluster(M, res, off, None, 0, X, p, thr, maxc, 1)
    ok = st == 0 and np.array_equal(cid, ocid) and np.array_equal(order, oorder)
    q.put((rank, bool(ok), int(allv.numel()), int(local_t.numel())))
    dist.destroy_process_group()


def test_gloo_world2_exchange_and_merge():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort()
    assert all(o[1] for o in out), out
    assert out[0][2] == out[1][2] == out[0][3] + out[1][3]  # every rank holds the union


def _cpu_worker_rank0_fails(rank, world, port, q):
    """rank 0's merge fails with something other than the reference-crash parity case (here: an edge that names a
    sequence outside [0, n)): every rank must come back with an error instead of waiting in a broadcast."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import json
    import hammock_amd
    from hammock_amd import dist as hd
    from oracle import c_oracle
    with open(os.path.join(ROOT, "tests", "golden", "matrices.json")) as fh:
        M = np.asarray(json.load(fh)["matrices"]["blosum62"], dtype=np.int32)
    peps = random_peptides(np.random.default_rng(1), 50, 12, 12, alphabet=5)
    res, off = c_oracle.pack(peps)
    ctx = hammock_amd.Context(M, device=-1)
    ctx.set_sequences(residues=res, offsets=off)
    bad = hammock_amd.pack_edges(np.array([3]), np.array([50]), np.array([25]))   # m == n
    allv = torch.from_numpy(bad.view(np.int64).copy())
    try:
        hd.merge_and_broadcast(ctx, allv, True, 20, 5)
        q.put((rank, "no error", ""))
    except hd.RemoteMergeError as e:
        q.put((rank, "remote", str(e)))
    except ValueError as e:
        q.put((rank, "local", str(e)))
    dist.destroy_process_group()


def test_gloo_world2_rank0_failure_reaches_every_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cpu_worker_rank0_fails, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0][:2] == (0, "local") and "outside" in out[0][2]
    assert out[1][:2] == (1, "remote") and "outside" in out[1][2]


def _gpu_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # both ranks share ONE GPU: gloo carries the exchange
    import json
    import hammock_amd
    from hammock_amd import dist as hd
    from hammock_amd.synth import synth_peptides
    with open(os.path.join(ROOT, "tests", "golden", "matrices.json")) as fh:
        M = np.asarray(json.load(fh)["matrices"]["blosum62"], dtype=np.int32)
    res, off = synth_peptides(1, 20000, 12)
    ctx = hammock_amd.Context(M, device=0)
    ctx.set_sequences(residues=res, offsets=off)
    local = hd.neighbors_local(ctx, 3, 0, 20, rank, world, dev).cpu()
    allv = hd.all_gather_edges(local)
    cid, order, info = hd.merge_and_broadcast(ctx, allv, True, 20, 500)
    single_cid, single_order, _ = ctx.greedy_cluster(3, 0, 20, 500)
    ok = np.array_equal(cid, single_cid) and np.array_equal(order, single_order)
    q.put((rank, bool(ok), int(allv.numel()), int(local.numel())))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_one_gpu_match_single_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort()
    assert all(o[1] for o in out), out
    assert out[0][2] == out[1][2] == out[0][3] + out[1][3]


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["rows", "edges"])
def test_pipelined_exchange_single_rank_matches_neighbors(fmt):
    """PipelinedExchange (compute stream + pack + ship on a second stream, double buffered) returns the
    same edge set as hmk_neighbors_shifted on every pass, in both block formats."""
    import json
    import hammock_amd
    from hammock_amd import dist as hd
    from hammock_amd.synth import synth_peptides
    with open(os.path.join(ROOT, "tests", "golden", "matrices.json")) as fh:
        M = np.asarray(json.load(fh)["matrices"]["blosum62"], dtype=np.int32)
    res, off = synth_peptides(3, 30000, 12)
    ctx = hammock_amd.Context(M, device=0)
    ctx.set_sequences(residues=res, offsets=off)
    for thr in (20, 8):          # thr 8: far denser than the buffers are first sized for -> they are regrown
        want, _ = ctx.neighbors_shifted(3, 0, thr)
        px = hd.PipelinedExchange(ctx, 3, 0, thr, 0, 1, torch.device("cuda", 0), fmt=fmt)
        for k in range(5):
            px.step()
            if k in (0, 3, 4):
                got = px.last_result().cpu().numpy().view(np.uint64)
                assert np.array_equal(np.sort(got), np.sort(want)), (thr, k)
        del px


@pytest.mark.gpu
def test_row_blocks_round_trip_directed_edges_and_misfit():
    """hmk_pack_rows_dev / hmk_unpack_rows_dev: the 4-byte exchange format reproduces the packed edges
    (asymmetric matrix: directed edges, x > m occurs; mixed lengths), and reports scores that do not
    fit score - threshold in 8 bits instead of truncating them."""
    import json
    import hammock_amd
    from hammock_amd import _native as N
    with open(os.path.join(ROOT, "tests", "golden", "matrices.json")) as fh:
        M = np.asarray(json.load(fh)["matrices"]["blosum62"], dtype=np.int32).copy()
    rng = np.random.default_rng(5)
    M[:20, :20] += rng.integers(-1, 2, size=(20, 20))          # asymmetric
    peps = random_peptides(rng, 3000, 8, 14, alphabet=6)
    res, off = hammock_amd.pack_sequences(peps)
    n = len(peps)
    ctx = hammock_amd.Context(M, device=0)
    ctx.set_sequences(residues=res, offsets=off)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    for thr, expect_misfit in ((18, False), (-300, True)):
        want, _ = ctx.neighbors_shifted(2, -1, thr)
        cap = (len(want) * 2 // N.HMK_EDGE_SHARDS + 4096) * N.HMK_EDGE_SHARDS
        d_edges = torch.empty(cap, dtype=torch.int64, device=dev)
        d_counts = torch.zeros(N.HMK_EDGE_SHARDS, dtype=torch.int64, device=dev)
        ctx.neighbors_shifted_dev(2, -1, thr, 0, 1, d_edges.data_ptr(), cap, d_counts.data_ptr(), stream.cuda_stream)
        head = torch.zeros(n + 2, dtype=torch.int32, device=dev)
        adj = torch.zeros(len(want) + 8, dtype=torch.int32, device=dev)
        ctx.pack_rows_dev(d_edges.data_ptr(), cap, d_counts.data_ptr(), thr, head.data_ptr(), adj.data_ptr(), adj.numel(),
                          stream.cuda_stream)
        h = head.cpu().numpy()
        assert h[n] == len(want)
        assert np.all(np.diff(h[:n + 1].astype(np.int64)) >= 0)
        if expect_misfit:
            rel = hammock_amd.edge_fields(want)[2].astype(np.int64) - thr
            assert h[n + 1] == int(((rel < 0) | (rel > 255)).sum()) > 0
            continue
        assert h[n + 1] == 0
        out = torch.zeros(len(want), dtype=torch.int64, device=dev)
        ctx.unpack_rows_dev(head.data_ptr(), adj.data_ptr(), thr, out.data_ptr(), out.numel(), stream.cuda_stream)
        got = out.cpu().numpy().view(np.uint64)
        assert np.array_equal(np.sort(got), np.sort(want))
        x = hammock_amd.edge_fields(got)[0]
        assert np.all(np.diff(x.astype(np.int64)) >= 0)          # grouped by x
        assert (hammock_amd.edge_fields(got)[0] > hammock_amd.edge_fields(got)[1]).any()  # directed edges present


def _px_worker(rank, world, port, q):
    try:
        _px_worker_body(rank, world, port, q)
    except BaseException as e:   # (a rank that dies silently costs the parent its whole timeout and hides the reason)
        import traceback
        q.put((rank, ["FAILED: " + repr(e) + "\n" + traceback.format_exc()]))
        raise


def _px_worker_body(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # both ranks share ONE GPU: gloo carries the exchange
    import json
    import hammock_amd
    from hammock_amd import dist as hd
    from hammock_amd.synth import synth_peptides
    with open(os.path.join(ROOT, "tests", "golden", "matrices.json")) as fh:
        M = np.asarray(json.load(fh)["matrices"]["blosum62"], dtype=np.int32)
    res, off = synth_peptides(2, 20000, 12)
    ctx = hammock_amd.Context(M, device=0)
    ctx.set_sequences(residues=res, offsets=off)
    want, _ = ctx.neighbors_shifted(3, 0, 20)
    oks = []
    for fmt in ("rows", "edges"):
        px = hd.PipelinedExchange(ctx, 3, 0, 20, rank, world, dev, fmt=fmt)
        for _ in range(3):
            px.step()
        got = px.last_result().cpu().numpy().view(np.uint64)
        oks.append(bool(np.array_equal(np.sort(got), np.sort(want))))
        oks.append(px.bytes_per_step)
    q.put((rank, oks))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_pipelined_exchange_two_ranks_one_gpu():
    """Both block formats over a real 2-rank exchange: every rank ends with the whole neighbour graph;
    the row blocks ship about half the bytes."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_px_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=400) for _ in procs]
    assert not any(str(oks[0]).startswith("FAILED") for _, oks in out), out
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, oks in out:
        assert oks[0] is True and oks[2] is True, out
        assert oks[1] < 0.6 * oks[3], out


def _nccl_single_rank_worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)   # "nccl" is RCCL on ROCm
    import json
    import hammock_amd
    from hammock_amd import dist as hd
    from hammock_amd.synth import synth_peptides
    with open(os.path.join(ROOT, "tests", "golden", "matrices.json")) as fh:
        M = np.asarray(json.load(fh)["matrices"]["blosum62"], dtype=np.int32)
    res, off = synth_peptides(5, 20000, 12)
    ctx = hammock_amd.Context(M, device=0)
    ctx.set_sequences(residues=res, offsets=off)
    want, _ = ctx.neighbors_shifted(3, 0, 20)
    oks = []
    # the row-block all-gather of the distributed entry point, on the one-rank RCCL group
    d_edges, d_counts, cap = hd.neighbors_local(ctx, 3, 0, 20, 0, 1, dev, segments=True)
    got = hd.all_gather_rows(ctx, d_edges, d_counts, cap, 20).cpu().numpy().view(np.uint64)
    oks.append(bool(np.array_equal(np.sort(got), np.sort(want))))
    oks.append(hd.all_gather_rows(ctx, d_edges, d_counts, cap, -400) is None)   # scores - threshold > 255: refused
    for fmt in ("rows", "edges"):
        px = hd.PipelinedExchange(ctx, 3, 0, 20, 0, 1, dev, fmt=fmt, use_collectives=True)
        for _ in range(4):
            px.step()
        got = px.last_result().cpu().numpy().view(np.uint64)
        oks.append(bool(np.array_equal(np.sort(got), np.sort(want))))
    q.put(oks)
    dist.destroy_process_group()


@pytest.mark.gpu
def test_pipelined_exchange_over_rccl_single_rank():
    """The exchange's RCCL calls (all_reduce, all_gather_into_tensor on int32 / int64 blocks, issued on the
    communication stream) on a one-rank "nccl" group: the only RCCL run a one-GPU box allows."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_single_rank_worker, args=(_free_port(), q))
    p.start()
    oks = q.get(timeout=400)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert oks == [True, True, True, True], oks


def _nccl_rank_per_gpu_worker(rank, world, port, q):
    """One rank per GPU over RCCL (what `bench.py --gpus N` launches): the pipelined exchange in both block formats, then the
    distributed clustering entry point; rank 0 reports against the single-GPU results it computes itself."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm
    import json
    import hammock_amd
    from hammock_amd import dist as hd
    from hammock_amd.synth import synth_peptides
    with open(os.path.join(ROOT, "tests", "golden", "matrices.json")) as fh:
        M = np.asarray(json.load(fh)["matrices"]["blosum62"], dtype=np.int32)
    res, off = synth_peptides(6, 30000, 12)
    ctx = hammock_amd.Context(M, device=rank)
    ctx.set_sequences(residues=res, offsets=off)
    want, _ = ctx.neighbors_shifted(3, 0, 20)          # the whole graph, on this rank's own GPU
    oks = []
    for fmt in ("rows", "edges"):
        px = hd.PipelinedExchange(ctx, 3, 0, 20, rank, world, dev, fmt=fmt, use_collectives=True)
        for _ in range(4):
            px.step()
        got = px.last_result().cpu().numpy().view(np.uint64)
        oks.append(bool(np.array_equal(np.sort(got), np.sort(want))))
    cid, order, info = hd.greedy_cluster_distributed(ctx, 3, 0, 20, 750, dev)
    cid1, order1, _ = ctx.greedy_cluster(3, 0, 20, 750)
    oks.append(bool(np.array_equal(cid, cid1) and np.array_equal(order, order1)))
    q.put((rank, oks))
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_rank_per_gpu_over_rccl(world):
    """The N > 1 path on REAL GPUs: one process per GPU, RCCL over xGMI (all_gather_into_tensor of the edge blocks on the
    communication stream, broadcast of the ids).  Skipped where the box has fewer GPUs than ranks -- a one-GPU box runs the
    same code on gloo / one-rank RCCL in the tests around this one; on a multi-GPU box this test needs no editing."""
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, this box has {torch.cuda.device_count()}")
    # the pool's test boxes allow 6 processes on the card at once (the "process guard" kills the run beyond that): more ranks
    # run only where the box's owner says it takes them
    allowed = int(os.environ.get("HMK_TEST_MAX_GPU_PROCS", "") or 6)
    if world > allowed:
        pytest.skip(f"{world} GPU processes: set HMK_TEST_MAX_GPU_PROCS>={world} on a box that allows them (default 6)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_rank_per_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, oks in out:
        assert oks == [True, True, True], (rank, oks)


@pytest.mark.gpu
def test_distributed_entry_single_process_device_merge():
    """greedy_cluster_distributed without a process group (one rank): the gathered graph stays on the GPU and
    rank 0 merges straight from it (hmk_greedy_from_edges_dev) -- same result as hmk_greedy_cluster."""
    import json
    import hammock_amd
    from hammock_amd import dist as hd
    from hammock_amd.synth import synth_peptides
    with open(os.path.join(ROOT, "tests", "golden", "matrices.json")) as fh:
        M = np.asarray(json.load(fh)["matrices"]["blosum62"], dtype=np.int32)
    res, off = synth_peptides(8, 15000, 12)
    ctx = hammock_amd.Context(M, device=0)
    ctx.set_sequences(residues=res, offsets=off)
    cid, order, info = hd.greedy_cluster_distributed(ctx, 3, 0, 20, 375, torch.device("cuda", 0))
    cid1, order1, _ = ctx.greedy_cluster(3, 0, 20, 375)
    assert np.array_equal(cid, cid1) and np.array_equal(order, order1)
