#!/usr/bin/env python3
"""bench.py -- pairwise ShiftedScorer (BLOSUM62) scores per second on MI355X.

Workload (BASELINE.json configs[2], the one `metric` is quoted on): 10^5 synthetic
length-12 peptides (SplitMix64 seed 1), BLOSUM62, max shift 3, shift penalty 0,
threshold 20 -- the defaults Hammock's greedy mode derives for this input
(Hammock.java:394-401,1409-1434).

A "step" is one pass of the hot path over the whole pair space: every unordered
pair {i, j} is scored on the GPU with the reference's ShiftedScorer semantics and
the pairs with score >= threshold are written to HBM as the neighbour list the
host greedy merge consumes.  Inputs are resident in HBM before the timed region.
With N > 1 ranks the pair space is sharded row-block-wise (no data-path
collective inside the scoring); each step ends with the RCCL all-gather of the
ranks' edge blocks, so every rank holds the whole neighbour graph.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_SEQ = 100_000
SEQ_LEN = 12
MAX_SHIFT, SHIFT_PENALTY, THRESHOLD = 3, 0, 20
E2E_DEADLINE_S = 120          # N > 1: the end-to-end extra after the timed steps may take this long at most
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# MI355X_MICROARCH.md, LDS table: ds_read_b64 = 2 LDS cycles per wave-instruction = 256 B/clk/CU; 256 CUs at 2.4 GHz
LDS_PEAK_GBS = 256 * 256 * 2.4          # 157,286 GB/s ("~150 TB/s aggregate for ds_read_b64/b128")
# Algorithmic LDS bytes per pair = the cells ShiftedScorer.java:67-77 adds: m (d + 1) + 2 X m - X (X + 1) = 72 at length 12,
# max shift 3, one byte each: the row-packed kernel (k_neighbors_rows.hip) reads 72 ds_read_b64 per 8 pairs and nothing else.
# (Rounds 1-2 read 96 B per pair: 12 entries of 7 shift lanes + 1 pad lane, zero cells of the partial overlaps included.)
SETTLE_STEPS = 12   # untimed passes before the warm-up steps: the GPU's clock ramp from idle (see main())
CELLS_PER_PAIR = SEQ_LEN * (2 * MAX_SHIFT + 1) - MAX_SHIFT * (MAX_SHIFT + 1)
LDS_BYTES_PER_PAIR = CELLS_PER_PAIR


def load_blosum62():
    with open(os.path.join(ROOT, "tests", "golden", "matrices.json")) as fh:
        return np.asarray(json.load(fh)["matrices"]["blosum62"], dtype=np.int32)


def pmc_traffic(n, world):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/round5_pmc_summary.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this very
    command, gfx950 x2 read correction applied).  Counters cannot be read from inside the timed
    run, so the value is only reported for the workload it was collected on; otherwise null."""
    try:
        d = None
        for name in ("round5_pmc_summary.json", "round4_pmc_summary.json"):
            path = os.path.join(ROOT, "profiles", name)
            if os.path.exists(path):
                with open(path) as fh:
                    d = json.load(fh)
                break
        if n == N_SEQ and world == 1:
            return d["hbm_traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError, TypeError):
        pass
    return None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """Host cores this process may actually use: the affinity mask, cut by the cgroup CPU quota if there is one
    (a GPU box hands out a share of a large host; OpenMP teams larger than the share only spin against each other)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    why = "affinity mask"
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max" and int(period) > 0:
            q = max(1, int(int(quota) / int(period)))
            if q < n:
                n, why = q, f"cgroup cpu.max {quota}/{period}"
    except (OSError, ValueError):
        pass
    return n, why


def cpu_baseline(M, res, off, n_sample, threads):
    """The oracle's literal greedy (the reference's algorithm, OpenMP over the
    reference's own 4*T partitions) on the first n_sample peptides of the workload:
    counted sequenceScore calls / wall time."""
    from oracle import c_oracle
    sub_off = off[:n_sample + 1].copy()
    sub_res = res[:sub_off[-1]].copy()
    perm = c_oracle.sort_order(sub_res, sub_off, None, "size")
    peps = [sub_res[sub_off[k]:sub_off[k + 1]] for k in perm]
    sres, soff = c_oracle.pack(peps)
    t0 = time.perf_counter()
    st, cid, order, stats = c_oracle.greedy_cluster(M, sres, soff, None, 0, MAX_SHIFT, SHIFT_PENALTY, THRESHOLD,
                                                    int(np.floor(n_sample * 0.025 + 0.5)), threads)
    dt = time.perf_counter() - t0
    calls = int(stats.score_calls_phase1 + stats.score_calls_phase2)
    return {"value": calls / dt, "unit": "pair scores/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
            "nproc": os.cpu_count(),
            "sample": f"oracle greedy (sort + cluster, Hammock.java:406-411) on the first {n_sample} peptides of the "
                      f"workload: {calls} sequenceScore calls in {dt:.2f} s, status {st}",
            "seconds": dt, "score_calls": calls,
            "pair_space_fraction": calls / (n_sample * (n_sample - 1) / 2)}


def cells_per_pair(la, lb, X):
    """Cells ShiftedScorer.java:67-77 adds for one pair of lengths la >= lb: m (d + 1) + 2 X m - X (X + 1)."""
    return lb * (2 * X + (la - lb) + 1) - X * (X + 1)


def lds_ideal_ms(lengths, X):
    """Time the LDS byte rate alone allows for one all-vs-all pass over sequences of these lengths: one byte per cell."""
    cnt = np.bincount(lengths)
    total = 0.0
    for la in range(1, len(cnt)):
        for lb in range(1, la + 1):
            pairs = cnt[la] * cnt[lb] if la != lb else cnt[la] * (cnt[la] - 1) // 2
            total += float(pairs) * cells_per_pair(la, lb, X)
    return total / (LDS_PEAK_GBS * 1e9) * 1e3


def load_fasta_unique(path):
    """FileIOManager.loadUniqueSequencesFromFasta (FileIOManager.java:159-202) for `>id|count|label` files: duplicates merged,
    counts summed, then the reference's default order (-R size: count descending, then the string descending)."""
    import gzip
    opener = gzip.open if path.endswith(".gz") else open
    counts, seen = {}, []
    with opener(path, "rt") as fh:
        cnt = 1
        for line in fh:
            line = line.strip()
            if line.startswith(">"):
                f = line[1:].split("|")
                cnt = int(f[1]) if len(f) > 1 and f[1] else 1
            elif line:
                q = line.upper()
                if q not in counts:
                    counts[q] = 0
                    seen.append(q)
                counts[q] += cnt
    seqs = sorted(seen, key=lambda q: (-counts[q], [-ord(c) for c in q]))
    return seqs, np.array([counts[q] for q in seqs], dtype=np.int32)


# LocalAlignmentScorer.java:43-81, one DP cell: 2 selects (:43-48 up gap open/extend, :50-55 left), 3 adds (:57-59), 2 max (:61),
# 1 clamp at zero (:63-67), 3 direction tests (:73-81), 1 running maximum (:68-72)
LOCAL_OPS_PER_CELL = 12


def other_configs(M, dev, stream):
    """The other BASELINE configs that fit one GPU, run AFTER the timed region (never inside it): config 2 (1e4 12-mers),
    4a (1e5 peptides of length 7..20, ShiftedScorer p = -1, thr 23), 4b (the same set, LocalAlignmentScorer -5 / -1, all
    ordered pairs, thr 28), one of the 8 shards of config 5 (1e6 12-mers), and two inputs off the tuned shape: 1e5 7-mers at
    the reference's defaults (Ph.D.-7 libraries: max shift 2, threshold 12, 2.1 % of the pairs are hits) and the reference's
    own antibodies example.  Kernel time by HIP events on the launch stream, median of a few passes after a warm-up; each with
    its own roofline fraction.  The edge buffer is sized from a first pass (a segment that overflows drops edges, and a pass
    that drops edges has not done its work): `overflowed` says whether the timed passes stored every edge."""
    import torch
    import hammock_amd
    from hammock_amd import _native
    from hammock_amd.synth import synth_peptides
    out = []
    S = _native.HMK_EDGE_SHARDS
    d_counts = torch.zeros(S, dtype=torch.int64, device=dev)

    def shifted(name, res, off, sizes, X, p, thr, part, n_parts, reps, warm, seqs=None):
        ctx = hammock_amd.Context(M, device=dev.index)
        if seqs is not None:
            ctx.set_sequences(seqs, sizes=sizes)
            lengths = np.array([len(q) for q in seqs], dtype=np.int64)
        else:
            ctx.set_sequences(residues=res, offsets=off)
            lengths = np.diff(off.astype(np.int64))
        n = len(lengths)
        cap = (1 << 22) // S * S
        d_edges = torch.empty(cap, dtype=torch.int64, device=dev)
        # a first pass sizes the edge buffer (its counts are exact even when a segment overflowed); then, as for the BASELINE line,
        # `warm` untimed passes and `reps` timed ones BACK TO BACK, each between two events on the launch stream, one synchronise at
        # the end (a synchronise after every pass lets the GPU idle between passes: 4-8 % slower figures, 30 % at 1e4)
        est_ms = 1.0
        for _ in range(2):
            t_est = time.perf_counter()
            ctx.neighbors_shifted_dev(X, p, thr, part, n_parts, d_edges.data_ptr(), cap, d_counts.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize(dev)
            est_ms = (time.perf_counter() - t_est) * 1e3
            mx = int(d_counts.max().item())
            if mx <= cap // S:
                break
            cap = (mx + mx // 8 + 1024) * S
            del d_edges
            d_edges = torch.empty(cap, dtype=torch.int64, device=dev)
        t_est = time.perf_counter()   # (the passes above built the plan and sized the buffer: this one says how long a pass takes)
        ctx.neighbors_shifted_dev(X, p, thr, part, n_parts, d_edges.data_ptr(), cap, d_counts.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize(dev)
        est_ms = (time.perf_counter() - t_est) * 1e3
        warm = int(min(400, max(warm, np.ceil(40.0 / max(est_ms, 0.05)))))   # ~40 ms of load: the clocks of an idle MI355X settle in ~25 ms
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for k in range(warm + reps):
            if k >= warm:
                evs[k - warm][0].record(stream)
            ctx.neighbors_shifted_dev(X, p, thr, part, n_parts, d_edges.data_ptr(), cap, d_counts.data_ptr(), stream.cuda_stream)
            if k >= warm:
                evs[k - warm][1].record(stream)
        torch.cuda.synchronize(dev)
        ms = [a.elapsed_time(b) for a, b in evs]
        plan = ctx.last_plan()
        med = float(np.median(ms))
        ideal = lds_ideal_ms(lengths, X) * (int(plan.pairs_scored) / (n * (n - 1) / 2))
        out.append({"config": name, "kernel_ms": med, "pairs": int(plan.pairs_scored), "pairs_per_s": int(plan.pairs_scored) / (med * 1e-3),
                    "edges": int(d_counts.sum().item()), "overflowed": bool(int(d_counts.max().item()) > cap // S),
                    "row_packed_classes": int(plan.classes_rows),
                    "classes": int(plan.classes_u8 + plan.classes_u16 + plan.classes_direct),
                    "roofline": {"bound": "lds", "ideal_ms": ideal, "frac": ideal / med,
                                 "definition": "one LDS byte per cell the reference adds (ShiftedScorer.java:67-77) at 256 B/clk/CU x 256 CU "
                                               "x 2.4 GHz, over the measured kernel time"}})
        ctx.close()
        del d_edges

    def synth(name, n, lo, hi, X, p, thr, part, n_parts, reps, warm):
        res, off = synth_peptides(1, n, lo, hi)
        shifted(name, res, off, None, X, p, thr, part, n_parts, reps, warm)

    synth("2: 1e4 x 12, BLOSUM62, X 3, p 0, thr 20", 10000, 12, 12, 3, 0, 20, 0, 1, 10, 5)
    synth("4a: 1e5 x 7..20, ShiftedScorer X 3, p -1, thr 23", 100000, 7, 20, 3, -1, 23, 0, 1, 8, 6)
    synth("5, one of 8 shards: 1e6 x 12, BLOSUM62, X 3, p 0, thr 20", 1000000, 12, 12, 3, 0, 20, 0, 8, 3, 2)
    synth("1e5 x 7 (a Ph.D.-7 sized library), BLOSUM62, X 2, p 0, thr 12: the reference's defaults for 7-mers", 100000, 7, 7, 2, 0, 12, 0, 1, 8, 6)
    fa = os.path.join(ROOT, "tests", "golden", "antibodies.fa.gz")
    if os.path.exists(fa):
        seqs, sizes = load_fasta_unique(fa)
        L = np.array([len(q) for q in seqs])
        jr = lambda v: int(np.floor(v + 0.5))   # Math.round
        thr, X = jr(L.mean() * 1.7), min(jr(L.mean() / 4), int(L.min()) - 1)   # Hammock.java:1409-1434
        shifted(f"antibodies.fa (the reference's example: {len(seqs)} unique sequences, lengths {L.min()}..{L.max()}), BLOSUM62, "
                f"X {X}, p 0, thr {thr}: the reference's defaults", None, None, sizes, X, 0, thr, 0, 1, 8, 6, seqs=seqs)
    # 4b: LocalAlignmentScorer, all ordered pairs of the 4a set: VALU-issue roofline from the ALGORITHM's operation count.
    # One cell of LocalAlignmentScorer.java:43-81 is LOCAL_OPS_PER_CELL = 12 integer operations (counted above the function);
    # the packed tagged-max kernel (k_local.hip) carries TWO column sequences in the 16-bit halves of a 32-bit lane, so one
    # lane-operation does two cells' worth: peak = 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz lane-operations/s x 2 cells each.
    # What the kernel really issues (SQ_INSTS_VALU of the PMC pass) is reported beside it: issued / algorithmic = its overhead.
    res, off = synth_peptides(1, 100000, 7, 20)
    ctx = hammock_amd.Context(M, device=dev.index)
    ctx.set_sequences(residues=res, offsets=off)
    ms = []
    for _ in range(3):
        edges, st = ctx.neighbors_local(-5, -1, 28, capacity=1 << 26)
        ms.append(float(st.kernel_ms))
    lens = np.diff(off.astype(np.int64)).astype(np.float64)
    cells = float(lens.sum()) ** 2 - float((lens * lens).sum())      # sum over ordered pairs i != j of len_i * len_j
    issued_per_cell = None
    for name in ("round5_neighbors_local_pmc.json", "round4_neighbors_local_pmc.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                issued_per_cell = float(json.load(fh)["valu_wave_instructions_per_64_cells"])
            break
        except (OSError, KeyError, ValueError):
            pass
    peak_lane_ops = 256 * 4 * 16 * 2.4e9
    med = float(np.median(ms[1:]))
    alg_lane_ops_per_cell = LOCAL_OPS_PER_CELL / 2.0
    frac_convention = cells * alg_lane_ops_per_cell / (med * 1e-3) / peak_lane_ops
    # Two figures under two names (they answer different questions, and neither is "percent of a floor"):
    #   frac (= frac_at_6_lane_instructions_per_cell): the kernel's time against the time 6 lane-instructions per cell would take at
    #     full VALU issue -- a CONVENTION (12 operations of the Java per cell, two cells per lane-operation), not a lower bound: a DP
    #     step that needs fewer than 6 would read above 1.  The kernel's DP step proper is DP_STEP_LANE_INSTR_PER_CELL from the ISA.
    #   frac_issued (= VALU-issue busy): what the hardware really issued (SQ_INSTS_VALU of the PMC pass, all in: DP step, table
    #     loads' address arithmetic, threshold test, loop) x cells over the kernel time against the issue peak.  Bounded by 1.
    DP_STEP_LANE_INSTR_PER_CELL = 10.75 / 2.0   # k_neighbors_local_pk's inner step: 10.75 VALU instructions per PAIR of cells
    out.append({"config": "4b: 1e5 x 7..20, LocalAlignmentScorer open -5, extend -1, all ordered pairs, thr 28", "kernel_ms": med,
                "pairs": int(st.pairs_scored), "pairs_per_s": int(st.pairs_scored) / (med * 1e-3), "edges": int(len(edges)),
                "dp_cells_per_s": cells / (med * 1e-3),
                "roofline": {"bound": "valu-issue", "definition_version": 3,
                             "frac": frac_convention,
                             "frac_at_6_lane_instructions_per_cell": frac_convention,
                             "frac_issued": (cells * issued_per_cell / (med * 1e-3) / peak_lane_ops) if issued_per_cell else None,
                             "algorithmic_ops_per_cell": LOCAL_OPS_PER_CELL, "cells_per_lane_operation": 2,
                             "dp_step_lane_instructions_per_cell": DP_STEP_LANE_INSTR_PER_CELL,
                             "issued_valu_lane_instructions_per_cell": issued_per_cell,
                             "definition": "VALU-issue busy at the algorithm's instruction CONVENTION, not a fraction of a floor: DP cells x 6 "
                                           "lane-instructions per cell (LocalAlignmentScorer.java:43-81 counted as 12 operations -- 2 selects, 3 "
                                           "adds, 2 max, 1 clamp, 3 direction tests, 1 running max -- / 2 cells per lane-operation: two column "
                                           "sequences in the 16-bit halves of a lane) over the kernel time, against 256 CU x 4 SIMD x 16 lanes/clk "
                                           "x 2.4 GHz.  The kernel's DP step is dp_step_lane_instructions_per_cell (ISA), all in it issues "
                                           "issued_valu_lane_instructions_per_cell (rocprofv3 SQ_INSTS_VALU); frac_issued is the fraction from that "
                                           "measured count (<= 1: the VALU pipe has no slack left), frac the one at the fixed 6.  What would make "
                                           "config 4b faster is fewer instructions per cell, not better scheduling.  definition_version 3 (rounds "
                                           "1-3: issued count; round 4: the convention only; now both)"}})
    ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=N_SEQ, help="number of synthetic peptides (default: the BASELINE workload)")
    ap.add_argument("--cpu-sample", type=int, default=100000,
                    help="peptides in the CPU baseline sample (default: the whole workload, about 4 s on 16 threads; "
                         "bounded: the default run stays within a few minutes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-greedy", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the block with the other BASELINE configs (2, 4a, 4b, a shard of 5)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    if not os.path.exists(os.path.join(ROOT, "hammock_amd", "lib", "libhammock_hip.so")):
        # a tree without the built library (the .so is not in git): build it once, rank 0 of the node first
        import subprocess
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "hammock_amd", "csrc"), "-j4"], stdout=subprocess.DEVNULL)
        else:
            while not os.path.exists(os.path.join(ROOT, "hammock_amd", "bin", "hammock-hip")):
                time.sleep(1.0)
    import hammock_amd
    from hammock_amd import _native
    from hammock_amd import dist as hd
    from hammock_amd.synth import synth_peptides

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # Rehearsal switches (one-GPU box): HMK_BENCH_SHARE_GPU=1 puts every rank on device 0 and
    # HMK_BENCH_BACKEND=gloo carries the collectives; the driver's real runs use neither.
    if os.environ.get("HMK_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("HMK_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    M = load_blosum62()
    n = args.n
    res, off = synth_peptides(1, n, SEQ_LEN)
    ctx = hammock_amd.Context(M, device=local_rank)
    ctx.set_sequences(residues=res, offsets=off)

    # device buffers (torch = plumbing: memory, stream, collectives)
    pairs_total = n * (n - 1) // 2
    expect_edges = int(pairs_total * 3.2e-3 / world) + 65536
    cap = (max(expect_edges * 2, 1 << 20) // _native.HMK_EDGE_SHARDS) * _native.HMK_EDGE_SHARDS
    d_edges = torch.empty(cap, dtype=torch.int64, device=dev)
    d_counts = torch.zeros(_native.HMK_EDGE_SHARDS, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev)
    seg = cap // _native.HMK_EDGE_SHARDS

    def score_pass():
        ctx.neighbors_shifted_dev(MAX_SHIFT, SHIFT_PENALTY, THRESHOLD, rank, world, d_edges.data_ptr(), cap,
                                  d_counts.data_ptr(), stream.cuda_stream)

    # N > 1: score on a compute stream, pack + RCCL all-gather of the edge blocks on a communication
    # stream, double buffered (hammock_amd/dist.py PipelinedExchange): afterwards every rank holds
    # the whole neighbour graph in HBM, ready for the host-side greedy merge.
    px = hd.PipelinedExchange(ctx, MAX_SHIFT, SHIFT_PENALTY, THRESHOLD, rank, world, dev) if world > 1 else None

    def step(t0=None, t1=None):
        if px is None:
            if t0 is not None:
                t0.record(stream)   # HIP events on the stream the kernel is launched on
            score_pass()
            if t1 is not None:
                t1.record(stream)
        else:
            px.step(t0, t1)

    # An idle MI355X needs about 25 ms of load to reach its clocks (profiles/round3_bench_warmup.log: 20 timed steps after 3 / 5 / 10 / 20
    # untimed ones take 2.60 / 2.58 / 2.53 / 2.53 ms each): SETTLE_STEPS untimed passes first, then the W warm-up steps the
    # contract asks for, then exactly K timed steps.  Reported in the line as "settle_steps".
    # First the contract's sequence taken literally -- W warm-up steps from an idle GPU, then K timed steps -- reported beside
    # the settled figure as "ms_per_step_no_settle" (it measures the clock ramp as much as the kernel).
    for _ in range(args.warmup):
        step()
    if px is not None:
        px.finish()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step()
    if px is not None:
        px.finish()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    no_settle = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(no_settle, op=dist.ReduceOp.MAX)
    ms_per_step_no_settle = float(no_settle.item()) / args.steps * 1e3
    for _ in range(SETTLE_STEPS):   # (this rank's scoring pass alone: the clocks are what settles, the exchange of N > 1 has no part in it)
        score_pass()
    torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    if px is not None:
        px.finish()
    torch.cuda.synchronize(dev)
    plan = ctx.last_plan()
    if px is None:
        counts = d_counts.cpu().numpy().astype(np.int64)
        if counts.max() > seg:
            sys.exit(f"edge segment overflow: {counts.max()} > {seg}")
        n_edges_rank = int(counts.sum())
    else:
        n_edges_rank = px.local_total

    # ---- timed region: exactly K steps -----------------------------------------------
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(ev[k][0], ev[k][1])
    if px is not None:
        px.finish()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    kern_ms = torch.tensor([float(np.mean([a.elapsed_time(b) for a, b in ev]))], dtype=torch.float64, device=dev)
    tot_edges = torch.tensor([n_edges_rank], dtype=torch.int64, device=dev)
    tot_pairs = torch.tensor([int(plan.pairs_scored)], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(kern_ms, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot_edges)
        dist.all_reduce(tot_pairs)
        gathered = px.last_result()  # every rank must hold the union of all shards
        assert gathered.numel() == int(tot_edges.item()), (gathered.numel(), int(tot_edges.item()))
    elapsed = float(elapsed.item())
    kern_ms = float(kern_ms.item())
    assert int(tot_pairs.item()) == pairs_total, (int(tot_pairs.item()), pairs_total)

    e2e = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = pairs_total / (elapsed / args.steps)
        # dominant kernel: k_neighbors_rows<3, 0, 12, true, 1, 0> (the shift-packed k_neighbors_swar with HMK_NO_ROWS_KERNEL=1).
        # Algorithmic HBM bytes per launch (DESIGN.md "Roofline"): 8 B per emitted edge + 16 B per peptide read once.
        pairs_rank = int(plan.pairs_scored)
        alg_bytes = 8 * n_edges_rank + 16 * n
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        rows_kernel = int(plan.classes_rows) > 0   # the row-packed kernel; HMK_NO_ROWS_KERNEL=1 runs round 2's shift-packed one
        lds_per_pair = LDS_BYTES_PER_PAIR if rows_kernel else SEQ_LEN * 8
        lds_gbs = pairs_rank * lds_per_pair / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "pairwise BLOSUM62 ShiftedScorer scores/sec (all-vs-all, thresholded neighbour list)",
            "value": value, "unit": "pair scores/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "settle_steps": SETTLE_STEPS, "ms_per_step_no_settle": ms_per_step_no_settle,
            "passes_before_the_timed_region": args.warmup + args.steps + SETTLE_STEPS + args.warmup,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8 (8-bit SWAR lanes, host-proven ranges with 16-bit / literal fallbacks; scores are int32-exact)",
            "data": "synthetic",
            "config": {"workload": f"{n} synthetic length-{SEQ_LEN} peptides (SplitMix64 seed 1), BLOSUM62, max_shift "
                                   f"{MAX_SHIFT}, shift_penalty {SHIFT_PENALTY}, threshold {THRESHOLD}; "
                                   f"{pairs_total} unordered pairs per step",
                       "parallelism": f"row-block sharding over {world} GPU(s)" + (
                           ", per-step RCCL all-gather of the edge blocks (4-byte row-block format) on a second stream "
                           "(overlaps the next pass)" if world > 1 else "")},
            # the BINDING roofline first (LDS gather, not HBM): bytes the kernel must read from LDS
            "roofline": {"bound": "lds", "achieved": lds_gbs, "peak": LDS_PEAK_GBS, "unit": "GB/s",
                         "frac": lds_gbs / LDS_PEAK_GBS, "traffic": pmc_traffic(n, world),
                         "kernel": ("k_neighbors_rows<3, 0, 12, true, 1, 0> (max shift 3, equal lengths, length 12 at compile time, 1 group of 8 "
                                    "rows per tile, plain edge list)" if rows_kernel else
                                    "k_neighbors_swar<2, 6, 2, 12, true, 0> (NW=2 dwords/entry, 6 rows/tile, 2 columns/lane, length 12 exact)"),
                         "lds_bytes_per_pair": lds_per_pair,
                         "frac_at_round2_definition_96_bytes_per_pair": pairs_rank * 96 / (kern_ms * 1e-3) / 1e9 / LDS_PEAK_GBS,
                         "kernel_ms": kern_ms,
                         "definition": (f"{lds_per_pair} LDS bytes per pair (= the {CELLS_PER_PAIR} cells the reference adds per pair, one byte each: "
                                        f"{CELLS_PER_PAIR} ds_read_b64 per 8 pairs) " if rows_kernel else
                                        f"{lds_per_pair} LDS bytes per pair ({SEQ_LEN} ds_read_b64 table lookups) ") +
                                       "x pairs per launch / kernel time, against 256 B/clk/CU x 256 CU x 2.4 GHz "
                                       "(MI355X_MICROARCH.md, LDS table)",
                         "traffic_note": "HBM bytes per launch from rocprofv3 PMC passes (profiles/round5_pmc_summary.json, tools/collect_round5.sh); "
                                         "null when the workload differs from the one the counters were collected on",
                         "hbm": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg_bytes,
                                 "note": "8 B per emitted edge + 16 B per peptide read once; HBM does not bind this path"}},
            "edges_per_step": int(tot_edges.item()),
        }
        if px is not None:
            # (both are asserted above: a line that exists has passed them)
            line["checks"] = {"pairs_of_all_ranks_equal_the_pair_space": True, "gathered_edges_equal_the_sum_of_the_shards": True,
                              "backend": os.environ.get("HMK_BENCH_BACKEND", "nccl"), "ranks_share_one_gpu": os.environ.get("HMK_BENCH_SHARE_GPU") == "1"}
            line["exchange"] = {"format": px.fmt, "gathered_bytes_per_rank_per_step": px.bytes_per_step,
                                "collectives_per_step": 1}
        maxc = int(np.floor(n * 0.025 + 0.5))
        if world == 1:
            if not args.no_greedy:
                # the sequences in the reference's default order (-R size: count descending, then the sequence string
                # descending, UniqueSequence.java:238-248; all counts are 1 here), in a context of their own
                letters = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)[res].reshape(n, SEQ_LEN)
                t = time.perf_counter()
                by_size = np.lexsort(letters.T[::-1])[::-1]
                sort_s = time.perf_counter() - t
                res_sorted = np.ascontiguousarray(res.reshape(n, SEQ_LEN)[by_size]).reshape(-1)
                gctx = hammock_amd.Context(M, device=local_rank)
                gctx.set_sequences(residues=res_sorted, offsets=off)
                gctx.greedy_cluster(MAX_SHIFT, SHIFT_PENALTY, THRESHOLD, maxc)   # first call sizes the context's buffers
                first = gctx.greedy_phases()
                # As for the headline: an idle MI355X needs a few tens of ms of load to reach its clocks, and a clustering call is 4 ms -- the
                # first calls of a context run on a ramping GPU (4.4, 4.3, 4.3, 4.2, ... settling by the tenth).  So: the first three calls as
                # they come (wall_s_first_calls), then untimed calls until ~40 ms of them have run, then `wall_s` = the median of three and
                # `wall_s_back_to_back` = the median of twenty more, all with nothing but the call between two clock readings.
                def one_call():
                    t = time.perf_counter()
                    out = gctx.greedy_cluster(MAX_SHIFT, SHIFT_PENALTY, THRESHOLD, maxc)
                    return time.perf_counter() - t, out
                first_calls = [one_call()[0] for _ in range(3)]
                settle_calls = 7
                for _ in range(settle_calls):
                    one_call()
                walls, phases = [], []
                for _ in range(3):   # three resident calls: the median is reported (one sample swings by 0.2 ms with the host)
                    w, (cid, order, gstats) = one_call()
                    walls.append(w)
                    phases.append(gctx.greedy_phases())
                mid = int(np.argsort(walls)[1])
                wall = walls[mid]
                tight, tight_score = [], []
                for _ in range(20):
                    tight.append(one_call()[0])
                    tight_score.append(gctx.greedy_phases()["score_ms"])
                line["greedy_end_to_end"] = {
                    "wall_s": wall, "wall_s_all": walls, "wall_s_first_calls": first_calls, "settle_calls": settle_calls,
                    "wall_s_back_to_back": float(np.median(tight)), "wall_s_back_to_back_min": float(min(tight)),
                    "score_ms_back_to_back": float(np.median(tight_score)),
                    "first_call_s": first["total_ms"] * 1e-3, "host_sort_s": sort_s, "clusters": int(gstats.n_multi),
                    "result_list": int(gstats.n_result_clusters), "phases_ms": phases[mid],
                    "note": "wall_s: median of three resident calls after `settle_calls` untimed ones (a GPU at its clocks, as for the headline); "
                            "wall_s_first_calls: the first three as they come; wall_s_back_to_back: median of twenty more.  "
                            "hmk_greedy_cluster = the span of Hammock.java:409 on the sequences in the reference's default order "
                            "(-R size; host_sort_s = numpy's sort, the span of :407): scoring, CSR, phase 1 on the host over the "
                            "band rows while the rest is scored, second loop on the device; phases overlap (see "
                            "include/hammock_hip.h hmk_greedy_phases)"}
            if not args.no_configs and n == N_SEQ:
                # (the clustering context and the bench context stay OPEN while the configs run: the mixed-length pass of config 4a used
                # to lose 8 % when its streams shared hardware queues with other live streams of the process -- 4.53 ms inside this run
                # against 4.17 ms on its own -- and rounds 3-4 closed these contexts first; its side streams are probed now, hmk_pass.cpp)
                t = time.perf_counter()
                try:
                    line["configs"] = other_configs(M, dev, stream)
                except Exception as exc:   # reported in the line, never instead of it
                    line["configs"] = {"error": f"{type(exc).__name__}: {exc}"}
                line["configs_wall_s"] = time.perf_counter() - t
                if not args.no_greedy:
                    gctx.close()
                if world == 1:
                    ctx.close()   # (N > 1 still needs it for the end-to-end extra below)
            if not args.no_cpu_baseline:
                cores, why = usable_cores()
                cores = min(cores, int(os.environ.get("HMK_BENCH_CPU_THREADS", "64")))   # the oracle's teams stop scaling well before that
                line["cpu_baseline"] = cpu_baseline(M, res, off, min(args.cpu_sample, n), cores)
                line["cpu_baseline"]["cores_chosen_by"] = why
                # BASELINE.md 3, "C-restate-1": the same restatement on ONE thread, on a sample it finishes in a few seconds
                line["cpu_baseline_1_thread"] = cpu_baseline(M, res, off, min(20000, n), 1)
    else:
        line = None
    if world > 1 and not args.no_greedy:
        # end to end on N GPUs, AFTER the measurement is complete: one more scored + exchanged pass, then the merge on rank 0
        # from the gathered graph (hmk_greedy_from_edges_dev) and the broadcast of the cluster ids.  The throughput line must
        # not depend on this extra: a watchdog prints it without the end-to-end figures if the section does not finish.
        import threading
        emit_lock = threading.Lock()
        emitted = [False]

        def emit(extra):
            """Exactly one JSON line per run, whoever gets here first (the main thread or the watchdog)."""
            with emit_lock:
                if emitted[0]:
                    return False
                emitted[0] = True
                if rank == 0:
                    out = dict(line)
                    out["greedy_end_to_end"] = extra
                    print(json.dumps(out), flush=True)
                return True

        def give_up():
            emit({"error": f"not finished within {E2E_DEADLINE_S} s; the timed steps above are complete"})
            os._exit(1)   # the line is out; a hung collective must not read as success

        watchdog = threading.Timer(E2E_DEADLINE_S, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            dist.barrier()
            t = time.perf_counter()
            px.step()
            gathered = px.last_result()
            cid, order, info = hd.merge_and_broadcast(ctx, gathered, True, THRESHOLD, int(np.floor(n * 0.025 + 0.5)))
            torch.cuda.synchronize(dev)
            e2e = {"wall_s": time.perf_counter() - t, "result_list": int(len(order)),
                   "clusters": int(np.sum(np.bincount(np.unique(cid, return_inverse=True)[1]) > 1)),
                   "phases_ms_rank0": ctx.greedy_phases() if rank == 0 else None,
                   "note": "score (sharded) + RCCL all-gather + unpack + merge on rank 0 + broadcast of the ids"}
        except Exception as exc:   # reported in the line, never instead of it
            e2e = {"error": f"{type(exc).__name__}: {exc}"}
        watchdog.cancel()
        emit(e2e)
        if e2e is not None and "error" in e2e:
            os._exit(1)   # a rank that failed may have left the others inside a collective: do not wait for them, and say so
        dist.destroy_process_group()
        return
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
