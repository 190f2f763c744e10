"""CPU, world_size 2 over gloo: the data-parallel plumbing of flid_amd.dist (what bench.py --gpus N runs over RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from flid_amd import dist as fdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, single=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = fdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    model = _net(single)
    for p_ in model.parameters():                      # ranks start from different weights; rank 0's must win
        p_.data.add_(rank * 0.5)
    fdist.broadcast_parameters(model)
    x = torch.from_numpy(np.random.RandomState(1).standard_normal((11, 6)).astype(np.float32))
    y = torch.from_numpy(np.random.RandomState(2).standard_normal((11, 3)).astype(np.float32))
    lo, hi = fdist.shard_bounds(len(x), rank, world)   # uneven shards: 6 + 5
    loss = ((model(x[lo:hi]) - y[lo:hi]) ** 2).mean()
    loss.backward()
    red = fdist.GradAllReducer(model.parameters())
    red.reduce(weight=(hi - lo) / len(x))              # mean loss over the GLOBAL batch
    # numpy, not tensors: a tensor travels through the queue as a shared-memory handle that dies with this process
    q.put((rank, [p_.grad.numpy().copy() for p_ in model.parameters()], [p_.data.numpy().copy() for p_ in model.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def _net(single):
    # single = one parameter tensor: the in-place path of GradAllReducer (flat-parameter mode)
    return torch.nn.Sequential(torch.nn.Linear(6, 3, bias=False)) if single else \
        torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))


@pytest.mark.parametrize("single", [False, True])
def test_grad_allreduce_equals_single_process_full_batch(single):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, single)) for r in range(world)]
    for p_ in procs:
        p_.start()
    out = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    # single-process reference on the full batch with rank 0's weights
    torch.manual_seed(0)
    model = _net(single)
    x = torch.from_numpy(np.random.RandomState(1).standard_normal((11, 6)).astype(np.float32))
    y = torch.from_numpy(np.random.RandomState(2).standard_normal((11, 3)).astype(np.float32))
    ((model(x) - y) ** 2).mean().backward()
    for rank, grads, weights in out:
        for g_, w_, p_ in zip(grads, weights, model.parameters()):
            assert np.allclose(w_, p_.data.numpy()), "broadcast_parameters did not install rank 0's weights"
            assert np.allclose(g_, p_.grad.numpy(), atol=1e-6), "weighted all-reduce != full-batch gradient"


def test_shard_bounds_partition():
    for n in (0, 1, 7, 600, 601):
        for world in (1, 2, 3, 8):
            spans = [fdist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _seg_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    fdist.init_from_env(backend="gloo")
    p = torch.nn.Parameter(torch.zeros(37))
    p.grad = torch.arange(37, dtype=torch.float32) * (rank + 1)        # rank r holds (r + 1) * [0..36]
    red = fdist.GradAllReducer([p])
    w = 0.25 if rank == 0 else 0.75
    red.segment_ready(p.grad[20:30], w)                                 # an upper layer's block, early
    red.segment_ready(p.grad[30:37], w)
    red.finish(w)                                                       # the rest: [0, 20)
    q.put((rank, p.grad.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_segmented_allreduce_covers_the_flat_gradient_once():
    """GradAllReducer.segment_ready + finish (the bucketed, overlapped form of the fused step): every element of the flat gradient
    is scaled and reduced exactly once, whatever the segments handed in early"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_seg_worker, args=(r, world, port, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    want = np.arange(37, dtype=np.float32) * (0.25 * 1 + 0.75 * 2)
    for _, g in out:
        assert np.allclose(g, want, atol=1e-6)
