"""cProfile of the steady-state TGN training step's HOST side (python tools/host_profile_tgn.py [steps]): the TGN line of bench.py
is bound by this thread, not by the GPU."""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import time

import torch
from flid_amd import _lib

CALLS = collections.defaultdict(lambda: [0, 0.0])
if "--calls" in sys.argv:                     # time every C entry point (ctypes call duration = launch issue cost) instead of cProfile
    sys.argv.remove("--calls")
    _L = _lib.lib()

    class _Wrap:
        def __init__(self, name, fn):
            self.name, self.fn = name, fn

        def __call__(self, *a):
            t = time.perf_counter()
            r = self.fn(*a)
            e = CALLS[self.name]
            e[0] += 1
            e[1] += time.perf_counter() - t
            return r

    class _Proxy:
        def __getattr__(self, n):
            w = _Wrap(n, getattr(_L, n))
            setattr(self, n, w)
            return w

    _PROXY = _Proxy()
    _lib.lib = lambda: _PROXY
    MODE = "calls"
else:
    MODE = "cprofile"
from flid_amd import ops
from flid_amd.synth import reddit_like
from flid_amd.utils.utils import get_neighbor_sampler
from flid_amd.models.MemoryModel import MemoryModel
from flid_amd.optim import FlatAdam

dev = torch.device("cuda:0")
data = reddit_like(num_edges=200000, seed=0)
sampler = get_neighbor_sampler(data, "recent", seed=0)
model = MemoryModel(data.node_raw_features, data.edge_raw_features, sampler, 100, "TGN", 1, 2, 0.1, device="cuda:0").train()
model.memory_bank.__init_memory_bank__()
flat = model.flatten_parameters()
opt = FlatAdam([flat], lr=1e-4)
B, n0 = 600, 100
rw = torch.randn(2 * B, 172, device=dev)
rg, lo = rw / (B * 172), torch.zeros(1, device=dev)


def loss(e):
    return ops.weighted_sum(e, rw, 1.0 / (B * 172), out=lo), rg


jobs, prep = {}, {}


def beg(s):
    sl = slice((n0 + s) * B, (n0 + s + 1) * B)
    return model.prepare_batch_begin(data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl], 20, edge_ids=data.edge_ids[sl])


def step(s):
    if s not in prep:
        prep[s] = model.prepare_batch_finish(jobs.pop(s) if s in jobs else beg(s))
    if s + 1 not in prep:
        prep[s + 1] = model.prepare_batch_finish(jobs.pop(s + 1) if s + 1 in jobs else beg(s + 1))
    if s + 2 not in jobs:
        jobs[s + 2] = beg(s + 2)
    sl = slice((n0 + s) * B, (n0 + s + 1) * B)
    flat.grad = None
    model.train_step(prep.pop(s), data.edge_ids[sl], loss, 20)
    opt.step()


for s in range(20):
    step(s)
torch.cuda.synchronize()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
if MODE == "calls":
    import flid_amd.engine as engine
    import flid_amd.graph as graph
    import flid_amd.models.MemoryModel as MM
    for mod in (ops, engine, graph, MM):
        if hasattr(mod, "lib"):
            mod.lib = _lib.lib
    CALLS.clear()
    t0 = time.perf_counter()
    for s in range(20, 20 + N):
        step(s)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    tot_c = sum(v[1] for v in CALLS.values())
    print(f"host {host / N * 1e3:.3f} ms/step; inside C calls {tot_c / N * 1e3:.3f} ms/step over {sum(v[0] for v in CALLS.values()) / N:.1f} calls/step")
    for k_, v in sorted(CALLS.items(), key=lambda kv: -kv[1][1]):
        print(f"{k_:28s} {v[0] / N:6.1f} calls/step {v[1] / N * 1e6:8.1f} us/step {v[1] / max(1, v[0]) * 1e6:7.1f} us/call")
else:
    pr = cProfile.Profile()
    pr.enable()
    for s in range(20, 20 + N):
        step(s)
    pr.disable()
    torch.cuda.synchronize()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(45)
    print(out.getvalue())
