"""Edge-batch data parallelism: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference has no distributed code; the temporal embedding shards naturally: every rank
holds the graph, the feature tables and the weights, embeds its own slice of the edge batch, and the only exchange is one
sum all-reduce of the ~1 M fp32 gradients (4 MB, latency-bound on 7 x 153 GB/s links) in a single flat bucket, weighted so
that a mean-reduced loss equals its single-GPU value.  In the fused step the bucket is cut in two: the root layer's block is
reduced on RCCL's stream as soon as its backward is queued (GradAllReducer.segment_ready), under the lower layer's backward; the
rest follows at the end (finish).  TGN needs no data exchange at all: every rank advances the replicated memory / message state
with the whole batch's (identical) update (MemoryModel.compute_shard_embeddings_and_advance)."""
import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """Rendezvous from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run sets them).  Returns
    (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("FLID_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(n: int, rank: int, world: int):
    """contiguous slice [lo, hi) of an n-edge batch owned by `rank` (sizes differ by at most one)"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradAllReducer:
    """Flat-bucket gradient all-reduce.  `weight` = local_edges / global_edges makes the reduced gradient that of the
    mean loss over the global batch (BCELoss / .mean() CE of the reference trainers)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(self.numel, dtype=ref.dtype, device=ref.device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    # ---- bucketed form for the fused step: segments of the (single, flat) gradient as they become final
    def segment_ready(self, seg: torch.Tensor, weight: float = None):
        """start the all-reduce of a finished segment of the flat gradient (a view of it) without blocking the stream that
        produced it: the collective runs on the backend's own stream, ordered behind everything queued so far"""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1:
            return
        w = (1.0 / world) if weight is None else float(weight)
        if w != 1.0:
            seg.mul_(w)
        self._pending = getattr(self, "_pending", [])
        self._pending.append((dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True), seg.data_ptr(), seg.numel()))

    def finish(self, weight: float = None):
        """reduce whatever segment_ready has not covered (flat-parameter mode: one gradient tensor) and make the current stream
        wait for every collective in flight"""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1:
            return
        pend = getattr(self, "_pending", [])
        self._pending = []
        if not pend:
            return self.reduce(weight)
        assert len(self.params) == 1 and self.params[0].grad is not None, "segment_ready is for the flat-parameter mode"
        g = self.params[0].grad
        w = (1.0 / world) if weight is None else float(weight)
        esz = g.element_size()
        done = sorted(((p - g.data_ptr()) // esz, n) for _, p, n in pend)
        pos, rest = 0, []
        for lo, n in done:                      # the complement of the segments already in flight
            if lo > pos:
                rest.append(g[pos:lo])
            pos = max(pos, lo + n)
        if pos < g.numel():
            rest.append(g[pos:])
        works = [wk for wk, _, _ in pend]
        for r in rest:
            if w != 1.0:
                r.mul_(w)
            works.append(dist.all_reduce(r, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for wk in works:
            wk.wait()                           # (device backends: the current stream waits; the host does not block)

    def reduce(self, weight: float = None):
        """Gather the gradients into the flat bucket (one multi-tensor copy), scale, ONE all-reduce, and hand the bucket's views
        back as the .grad tensors (no copy back).  Call optimizer.zero_grad() between steps as usual."""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1:
            return
        w = (1.0 / world) if weight is None else float(weight)
        if len(self.params) == 1 and self.params[0].grad is not None and self.params[0].grad.is_contiguous():
            g = self.params[0].grad                 # flat-parameter mode: the gradient block is the bucket, reduced in place
            if w != 1.0:
                g.mul_(w)
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
            return
        dst, src = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)
        if w != 1.0:
            self.flat.mul_(w)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        for p, v in zip(self.params, self.views):
            p.grad = v


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None):
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
