import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from flid_amd import engine, ops
from flid_amd.synth import reddit_like
from flid_amd.utils.utils import get_neighbor_sampler
from flid_amd.models.MemoryModel import MemoryModel
from flid_amd.optim import FlatAdam
import flid_amd.models.MemoryModel as MM
dev = torch.device("cuda:0")
data = reddit_like(num_edges=200000, seed=0)
sampler = get_neighbor_sampler(data, "recent", seed=0)
model = MemoryModel(data.node_raw_features, data.edge_raw_features, sampler, 100, "TGN", 1, 2, 0.1, device="cuda:0").train()
model.memory_bank.__init_memory_bank__()
flat = model.flatten_parameters()
opt = FlatAdam([flat], lr=1e-4)
B = 600
rw = torch.randn(2 * B, 172, device=dev); rg = rw / (B * 172); lo = torch.zeros(1, device=dev)
def loss(e): return ops.weighted_sum(e, rw, 1.0 / (B * 172), out=lo), rg
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
# wrap selected callables
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); tick(label or name, t0); return r
    setattr(obj, name, g)
for nm in ("gather_rows", "gru_cell_fwd", "wgrad_group", "h2d", "build_messages", "weighted_sum", "time_encode"):
    wrap(ops, nm)
wrap(engine, "_native_forward"); wrap(engine, "_native_backward")
wrap(model, "_advance_state"); wrap(model, "prepare_batch_begin"); wrap(model, "prepare_batch_finish")
wrap(torch, "where", "torch.where")
wrap(opt, "step", "opt.step")
n0 = 100
jobs = {}
def beg(s):
    sl = slice((n0 + s) * B, (n0 + s + 1) * B)
    return model.prepare_batch_begin(data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl], 20)
prep = {}
def step(s):
    if s not in prep: prep[s] = model.prepare_batch_finish(jobs.pop(s) if s in jobs else beg(s))
    if s + 1 not in prep: prep[s + 1] = model.prepare_batch_finish(jobs.pop(s + 1) if s + 1 in jobs else beg(s + 1))
    if s + 2 not in jobs: jobs[s + 2] = beg(s + 2)
    sl = slice((n0 + s) * B, (n0 + s + 1) * B)
    flat.grad = None
    t0 = time.perf_counter()
    model.train_step(prep.pop(s), data.edge_ids[sl], loss, 20)
    tick("train_step(total)", t0)
    opt.step()
for s in range(20): step(s)
torch.cuda.synchronize(); T.clear()
t0 = time.perf_counter()
N = 100
for s in range(20, 20 + N): step(s)
host = time.perf_counter() - t0
torch.cuda.synchronize()
print("host ms/step", host / N * 1e3, "wall", (time.perf_counter() - t0) / N * 1e3)
for k, v in sorted(T.items(), key=lambda kv: -kv[1]): print(f"{k:28s} {v / N * 1e3:8.3f} ms/step")
