"""host time per C-ABI entry point inside the TGAT bench step (ctypes call duration = launch issue cost)"""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flid_amd import _lib
L = _lib.lib()
acc = collections.defaultdict(lambda: [0, 0.0])
class Wrap:
    def __init__(self, name, fn): self.name, self.fn = name, fn
    def __call__(self, *a):
        t = time.perf_counter(); r = self.fn(*a); d = time.perf_counter() - t
        e = acc[self.name]; e[0] += 1; e[1] += d
        return r
class Proxy:
    def __getattr__(self, n):
        return Wrap(n, getattr(L, n))
_lib.lib = lambda: PROXY
PROXY = Proxy()
import flid_amd.ops as ops, flid_amd.engine as engine, flid_amd.graph as graph
for m in (ops, engine, graph):
    if hasattr(m, "lib"): m.lib = _lib.lib
from flid_amd.models.TGAT import TGAT
from flid_amd.synth import wikipedia_like
from flid_amd.utils.utils import get_neighbor_sampler
dev = torch.device("cuda:0")
data = wikipedia_like(seed=0)
n_train = int(0.7 * data.num_interactions)
sampler = get_neighbor_sampler(data.slice(0, n_train), "recent", seed=0)
torch.manual_seed(0)
model = TGAT(data.node_raw_features, data.edge_raw_features, sampler, 100, 2, 2, 0.1, "cuda:0").to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
B = 600
batches = []
for s in range(60):
    sl = slice((90 + s) * B, (91 + s) * B)
    batches.append((torch.from_numpy(data.src_node_ids[sl].astype(np.int32)).to(dev), torch.from_numpy(data.dst_node_ids[sl].astype(np.int32)).to(dev),
                    torch.from_numpy(data.node_interact_times[sl]).to(dev)))
rw = torch.randn(2, B, 172, device=dev)
prepared = {}
phase = collections.defaultdict(float)
def step(s):
    t0 = time.perf_counter()
    if s not in prepared: prepared[s] = model.prepare_batch(*batches[s], 20)
    if s + 1 < len(batches): prepared[s + 1] = model.prepare_batch(*batches[s + 1], 20)
    t1 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    se, de_ = model.compute_src_dst_node_temporal_embeddings(prepared.pop(s), None, None, 20)
    loss = torch.addcmul(se * rw[0], de_, rw[1]).mean()
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    phase["prepare"] += t1 - t0; phase["forward"] += t2 - t1; phase["backward"] += t3 - t2; phase["adam"] += t4 - t3
for s in range(10): step(s)
torch.cuda.synchronize()
acc.clear(); phase.clear()
t0 = time.perf_counter()
N = 40
for s in range(10, 10 + N): step(s)
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"host {host/N*1e3:.3f} ms/step, wall {wall/N*1e3:.3f} ms/step")
print("phases ms/step:", {k: round(v / N * 1e3, 3) for k, v in phase.items()})
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:28s} {c/N:5.1f} calls/step  {t/N*1e6:8.1f} us/step  {t/c*1e6:7.1f} us/call")
