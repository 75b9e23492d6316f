"""Multi-GPU form of the hot path: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI; "gloo" on CPU for the tests).

The pair space is sharded row-block-wise: rank r scores the row blocks with
(block index mod world) == r against all their columns -- no collective in the
scoring itself.  The only exchange is the all-gather of the ranks' edge blocks
(the thresholded neighbour lists) before the host-side greedy merge; after it
every rank holds the whole neighbour graph, rank 0 merges and broadcasts the
cluster ids.  torch is plumbing here: device memory, the stream and the
collectives.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import _native as N
from .api import Context


def compact_shards(d_edges: torch.Tensor, d_counts: torch.Tensor) -> torch.Tensor:
    """The kernel writes HMK_EDGE_SHARDS segments; returns their valid prefixes concatenated."""
    seg = d_edges.numel() // N.HMK_EDGE_SHARDS
    counts = d_counts.tolist()
    if max(counts) > seg:
        raise BufferError(f"edge segment overflow: {max(counts)} > {seg}")
    return torch.cat([d_edges[s * seg:s * seg + int(c)] for s, c in enumerate(counts)])


def neighbors_local(ctx: Context, max_shift, shift_penalty, threshold, rank, world, device, capacity=None,
                    segments=False):
    """This rank's shard of the neighbour graph as a device tensor of packed edges (int64 view of uint64);
    segments=True returns the raw (d_edges, d_counts, capacity) of the kernel instead."""
    n = ctx.n
    if capacity is None:
        capacity = int(n * (n - 1) // 2 * 6e-3 / world) + (1 << 20)
    capacity = (capacity // N.HMK_EDGE_SHARDS + 1) * N.HMK_EDGE_SHARDS
    while True:
        d_edges = torch.empty(capacity, dtype=torch.int64, device=device)
        d_counts = torch.zeros(N.HMK_EDGE_SHARDS, dtype=torch.int64, device=device)
        stream = torch.cuda.current_stream(device)
        ctx.neighbors_shifted_dev(max_shift, shift_penalty, threshold, rank, world, d_edges.data_ptr(), capacity,
                                  d_counts.data_ptr(), stream.cuda_stream)
        mx = int(d_counts.max().item())
        if mx <= capacity // N.HMK_EDGE_SHARDS:
            return (d_edges, d_counts, capacity) if segments else compact_shards(d_edges, d_counts)
        capacity = (mx + mx // 8 + 1024) * N.HMK_EDGE_SHARDS  # a segment overflowed: rescore with room


def all_gather_edges(local: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather of variable-length edge blocks: one small all-gather of the lengths,
    then one all-gather of blocks padded to the longest.  Returns every rank's edges
    concatenated in rank order (same on all ranks)."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    sizes = torch.zeros(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(sizes, torch.tensor([local.numel()], dtype=torch.int64, device=local.device),
                                group=group)
    sizes_h = sizes.tolist()
    mx = max(max(sizes_h), 1)
    padded = torch.zeros(mx, dtype=torch.int64, device=local.device)
    padded[:local.numel()] = local
    gathered = torch.empty(world * mx, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    return torch.cat([gathered[r * mx:r * mx + sizes_h[r]] for r in range(world)])


def all_gather_rows(ctx: Context, d_edges, d_counts, capacity, threshold, group=None):
    """The exchange in the 4-byte row-block format (hmk_pack_rows_dev): every rank's edge segments are regrouped
    by x on the device, ONE padded all-gather ships [row_start | adj] of all ranks, and the received blocks are
    unpacked to packed 8-byte edges.  Returns None if some score - threshold does not fit 8 bits (the caller
    falls back to all_gather_edges)."""
    world = dist.get_world_size(group)
    dev = d_edges.device
    n = ctx.n
    total = int(d_counts.sum().item())
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, torch.tensor([total], dtype=torch.int64, device=dev), group=group)
    sizes_h = sizes.tolist()
    pad = max(max(sizes_h), 1)
    head_len = n + 2
    msg = torch.zeros(head_len + pad, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    ctx.pack_rows_dev(d_edges.data_ptr(), capacity, d_counts.data_ptr(), threshold, msg.data_ptr(),
                      msg[head_len:].data_ptr(), pad, stream.cuda_stream)
    gathered = torch.empty(world * (head_len + pad), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(gathered, msg, group=group)
    msgs = gathered.view(world, head_len + pad)
    tail = msgs[:, n:head_len].tolist()           # per rank: [edges, misfits]
    if any(t[1] for t in tail):
        return None
    out = torch.empty(sum(t[0] for t in tail), dtype=torch.int64, device=dev)
    o = 0
    for r in range(world):
        if tail[r][0]:
            ctx.unpack_rows_dev(msgs[r].data_ptr(), msgs[r, head_len:].data_ptr(), threshold, out[o:].data_ptr(), tail[r][0],
                                stream.cuda_stream)
        o += tail[r][0]
    stream.synchronize()
    return out


class RemoteMergeError(RuntimeError):
    """The merge on rank 0 failed; raised on every OTHER rank with rank 0's message (rank 0 re-raises its own error)."""


def merge_and_broadcast(ctx: Context, edges_all: torch.Tensor, symmetric: bool, threshold, max_clusters, group=None):
    """Host greedy merge on rank 0 (hmk_greedy_from_edges[_dev]), result broadcast to every rank.
    -> (cluster_id int32[n], result_order int32[n_result], status dict).  A reference crash
    (NullPointerException parity) is raised on every rank; so is ANY other failure of rank 0 (device error, out of
    memory, bad edges): rank 0 always reaches the broadcasts and ships its status first, so no rank is left waiting in
    a collective for the timeout."""
    from .api import ReferenceWouldCrash
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    multi = dist.is_initialized() and dist.get_world_size(group) > 1
    n = ctx.n
    dev = edges_all.device
    MSG = 512
    # status (0 ok, HMK_ERR_REFERENCE_WOULD_CRASH, -1 other failure), n_result, crash_case, crash_index, message length
    header = torch.zeros(5, dtype=torch.int64, device=dev)
    message = torch.zeros(MSG, dtype=torch.uint8, device=dev)
    cid = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    order = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    err = None
    stats = None
    if rank == 0:
        try:
            if edges_all.is_cuda and ctx.device >= 0:
                # the gathered graph is already on this GPU: adjacency and the second loop stay there
                torch.cuda.current_stream(edges_all.device).synchronize()
                c, o, stats = ctx.greedy_from_edges_dev(edges_all.data_ptr(), edges_all.numel(), symmetric, max_clusters)
            else:
                edges = edges_all.cpu().numpy().view(np.uint64)
                c, o, stats = ctx.greedy_from_edges(edges, symmetric, threshold, max_clusters)
            header[1] = len(o)
            cid[:n] = torch.from_numpy(c).to(dev)
            order[:len(o)] = torch.from_numpy(o).to(dev)
        except ReferenceWouldCrash as e:
            err = e
            header[0], header[2], header[3] = N.HMK_ERR_REFERENCE_WOULD_CRASH, e.case, e.index
        except Exception as e:  # noqa: BLE001 -- every failure must reach the other ranks before anyone raises
            err = e
            text = f"{type(e).__name__}: {e}".encode("utf-8", "replace")[:MSG]
            header[0], header[4] = -1, len(text)
            message[:len(text)] = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
    if multi:
        dist.broadcast(header, 0, group=group)
        dist.broadcast(message, 0, group=group)
        dist.broadcast(cid, 0, group=group)
        dist.broadcast(order, 0, group=group)
    h = header.tolist()
    if h[0] == N.HMK_ERR_REFERENCE_WOULD_CRASH:
        raise err or ReferenceWouldCrash("the reference throws NullPointerException here", h[2], h[3])
    if h[0] != 0:
        if err is not None:
            raise err
        raise RemoteMergeError("greedy merge failed on rank 0: " + bytes(message[:h[4]].tolist()).decode("utf-8", "replace"))
    return cid[:n].cpu().numpy(), order[:h[1]].cpu().numpy(), {"n_result_clusters": h[1], "stats": stats}


def greedy_cluster_distributed(ctx: Context, max_shift, shift_penalty, threshold, max_clusters, device, group=None):
    """LimitedGreedySequenceClusterer.cluster over all ranks of `group`: identical result on every rank,
    identical to the single-GPU hmk_greedy_cluster."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    use_rows = world > 1 and device.type == "cuda" and dist.get_backend(group) == "nccl"
    if use_rows:   # 4 bytes per edge over xGMI; the 8-byte form if a score does not fit
        d_edges, d_counts, capacity = neighbors_local(ctx, max_shift, shift_penalty, threshold, rank, world, device,
                                                      segments=True)
        edges_all = all_gather_rows(ctx, d_edges, d_counts, capacity, threshold, group)
        if edges_all is None:
            edges_all = all_gather_edges(compact_shards(d_edges, d_counts), group)
    else:
        local = neighbors_local(ctx, max_shift, shift_penalty, threshold, rank, world, device)
        edges_all = all_gather_edges(local, group) if world > 1 else local
    symmetric = bool((ctx.matrix == ctx.matrix.T).all())
    return merge_and_broadcast(ctx, edges_all, symmetric, threshold, max_clusters, group)


class PipelinedExchange:
    """Steady-state form of score -> exchange for repeated passes (bench.py, N > 1 ranks).

    No host round trip inside a step: the neighbour kernel runs on a compute stream; on a
    communication stream small device kernels turn its 16 output segments into one exchange block and
    ONE fixed-size all_gather_into_tensor ships it, double buffered so the exchange of pass k overlaps
    the scoring of pass k + 1.  Two block formats:

      "rows"   (default) 4 bytes per edge: hmk_pack_rows_dev groups the edges by x into
               row_start[n + 2] + adj[] = m << 8 | score - threshold; half the xGMI bytes
      "edges"  8 bytes per edge: hmk_compact_edges_dev, the packed edges as they are

    The block size `pad` is the largest per-rank edge count (identical every pass for the same input),
    found by one warm-up pass."""

    def __init__(self, ctx: Context, max_shift, shift_penalty, threshold, rank, world, device, group=None, fmt="rows",
                 shard=None, use_collectives=None):
        if fmt not in ("rows", "edges"):
            raise ValueError(fmt)
        self.ctx, self.args = ctx, (int(max_shift), int(shift_penalty), int(threshold))
        self.rank, self.world, self.device, self.group, self.fmt = rank, world, device, group, fmt
        # shard = (part, n_parts) of the pair space this rank scores; default: one shard per rank.
        # (tools/px_step_time.py overrides it to time a 1/8 shard's step on a single GPU.)
        self.part, self.n_parts = shard if shard is not None else (rank, world)
        # the collectives run whenever there is more than one rank; a single-rank process group can ask for
        # them too (tests: exercises the RCCL calls, dtypes and stream order on a one-GPU box)
        self.collectives = (world > 1) if use_collectives is None else bool(use_collectives)
        n = ctx.n
        self.capacity = ((int(n * (n - 1) // 2 * 6e-3 / self.n_parts) + (1 << 20)) // N.HMK_EDGE_SHARDS + 1) * N.HMK_EDGE_SHARDS
        self.comp = torch.cuda.Stream(device)
        self.comm = torch.cuda.Stream(device)
        # warm-up pass sizes the score buffers (regrown if a segment overflows: denser data than uniform
        # random peptides) and the exchange block
        # (torch.zeros / torch.empty run on the CURRENT stream; comp and comm are non-blocking streams that do not wait for it by
        # themselves.  Without these waits the zero-fill of a freshly allocated counter block could land AFTER the warm-up pass had
        # written it: the block then sized itself for 0 edges and the first real pass "overflowed" it -- once in ~15 runs of the
        # two-rank test, and a possible failure of `bench.py --gpus N`.)
        cur = torch.cuda.current_stream(device)
        while True:
            e, c = self._alloc_score()
            self.comp.wait_stream(cur)
            ctx.neighbors_shifted_dev(*self.args, self.part, self.n_parts, e.data_ptr(), self.capacity, c.data_ptr(),
                                      self.comp.cuda_stream)
            self.comp.synchronize()
            cnt = c.tolist()
            if max(cnt) <= self.capacity // N.HMK_EDGE_SHARDS:
                break
            del e, c
            self.capacity = (max(cnt) + max(cnt) // 8 + 1024) * N.HMK_EDGE_SHARDS
        self.buf = [(e, c), self._alloc_score()]
        self.local_total = int(sum(cnt))
        mx = torch.tensor([self.local_total], dtype=torch.int64, device=device)
        if self.collectives:
            dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
        self.pad = int(mx.item()) + 64
        # One message per rank and pass: [head | block].  rows: head = row_start[n + 2], block = one uint32 per
        # edge.  edges: head = the valid count, block = the packed edges.  ONE all-gather ships it.
        dt = torch.int32 if fmt == "rows" else torch.int64
        self.head_len = n + 2 if fmt == "rows" else 1
        self.msg_len = self.head_len + self.pad
        self.msg = [torch.zeros(self.msg_len, dtype=dt, device=device) for _ in range(2)]
        self.head = [m[:self.head_len] for m in self.msg]
        self.block = [m[self.head_len:] for m in self.msg]
        self.gathered = [torch.empty(world * self.msg_len, dtype=dt, device=device) for _ in range(2)]
        self.bytes_per_step = self.gathered[0].numel() * self.gathered[0].element_size()
        self.scored = [torch.cuda.Event() for _ in range(2)]
        self.packed = [torch.cuda.Event() for _ in range(2)]
        self.comp.wait_stream(cur)   # the fills of buf[1] and msg[] above, before anything on the two streams touches them
        self.comm.wait_stream(cur)
        self.work = [None, None]   # the all-gather in flight for each message buffer (async: the pack of the next
                                   # pass does not queue behind it on the communication stream)
        self.k = 0

    def _alloc_score(self):
        return (torch.empty(self.capacity, dtype=torch.int64, device=self.device),
                torch.zeros(N.HMK_EDGE_SHARDS, dtype=torch.int64, device=self.device))

    def step(self, t0=None, t1=None):
        """One pass: score this rank's shard, ship it.  t0/t1: optional timing events recorded on the
        compute stream around the scoring kernel."""
        b = self.k & 1
        e, c = self.buf[b]
        if self.k >= 2:
            self.comp.wait_event(self.packed[b])  # score buffer b was packed into its block two passes ago
        if t0 is not None:
            t0.record(self.comp)
        self.ctx.neighbors_shifted_dev(*self.args, self.part, self.n_parts, e.data_ptr(), self.capacity, c.data_ptr(),
                                       self.comp.cuda_stream)
        if t1 is not None:
            t1.record(self.comp)
        self.scored[b].record(self.comp)
        self.comm.wait_event(self.scored[b])
        with torch.cuda.stream(self.comm):
            # msg[b] / gathered[b] are free again once the all-gather issued two passes ago has completed
            if self.work[b] is not None:
                self.work[b].wait()
                self.work[b] = None
            if self.fmt == "rows":
                self.ctx.pack_rows_dev(e.data_ptr(), self.capacity, c.data_ptr(), self.args[2], self.head[b].data_ptr(),
                                       self.block[b].data_ptr(), self.pad, self.comm.cuda_stream)
            else:
                self.ctx.compact_edges_dev(e.data_ptr(), self.capacity, c.data_ptr(), self.block[b].data_ptr(), self.pad,
                                           self.head[b].data_ptr(), self.comm.cuda_stream)
            self.packed[b].record(self.comm)
            if self.collectives:
                self.work[b] = dist.all_gather_into_tensor(self.gathered[b], self.msg[b], group=self.group, async_op=True)
            else:
                self.gathered[b][:self.msg_len].copy_(self.msg[b])
        self.k += 1

    def finish(self):
        with torch.cuda.stream(self.comm):
            for b in range(2):
                if self.work[b] is not None:
                    self.work[b].wait()
                    self.work[b] = None
        self.comp.synchronize()
        self.comm.synchronize()

    def last_result(self) -> torch.Tensor:
        """every rank's edges of the most recent pass as packed 8-byte edges, concatenated in rank order"""
        self.finish()
        b = (self.k - 1) & 1
        msgs = self.gathered[b].view(self.world, self.msg_len)
        if self.fmt == "edges":
            tot = msgs[:, 0].tolist()
            if max(tot) > self.pad:
                raise BufferError(f"exchange block overflow: {tot} edges per rank, blocks of {self.pad}")
            return torch.cat([msgs[r, 1:1 + int(tot[r])] for r in range(self.world)])
        n = self.ctx.n
        heads = msgs[:, :self.head_len]
        tail = heads[:, n:].tolist()   # per rank: [edges, misfits]
        if max(t[0] for t in tail) > self.pad:
            raise BufferError(f"exchange block overflow: [edges, misfits] per rank {tail}, blocks of {self.pad}")
        if any(t[1] for t in tail):
            raise OverflowError("a score - threshold does not fit 8 bits: use fmt='edges' for these parameters")
        out = torch.empty(sum(t[0] for t in tail), dtype=torch.int64, device=self.device)
        o = 0
        stream = torch.cuda.current_stream(self.device)
        for r in range(self.world):
            if tail[r][0]:
                self.ctx.unpack_rows_dev(msgs[r].data_ptr(), msgs[r, self.head_len:].data_ptr(), self.args[2],
                                         out[o:].data_ptr(), tail[r][0], stream.cuda_stream)
            o += tail[r][0]
        stream.synchronize()
        return out
