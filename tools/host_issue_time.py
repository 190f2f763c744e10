"""how long does the HOST need to issue one TGAT step (no GPU wait)?  vs the GPU-side time of the same step"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flid_amd.models.TGAT import TGAT
from flid_amd.synth import wikipedia_like
from flid_amd.utils.utils import get_neighbor_sampler
from flid_amd import engine
dev = torch.device("cuda:0")
data = wikipedia_like(seed=0)
n_train = int(0.7 * data.num_interactions)
sampler = get_neighbor_sampler(data.slice(0, n_train), "recent", seed=0)
torch.manual_seed(0)
model = TGAT(data.node_raw_features, data.edge_raw_features, sampler, 100, 2, 2, 0.1, "cuda:0").to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
B = 600
batches = []
for s in range(40):
    sl = slice((90 + s) * B, (91 + s) * B)
    batches.append((torch.from_numpy(data.src_node_ids[sl].astype(np.int32)).to(dev), torch.from_numpy(data.dst_node_ids[sl].astype(np.int32)).to(dev),
                    torch.from_numpy(data.node_interact_times[sl]).to(dev)))
rw = torch.randn(2, B, 172, device=dev)
def step(s, parts=None):
    t0 = time.perf_counter()
    src, dst, t = batches[s]
    opt.zero_grad(set_to_none=True)
    se, de_ = model.compute_src_dst_node_temporal_embeddings(src, dst, t, 20)
    t1 = time.perf_counter()
    loss = (se * rw[0]).mean() + (de_ * rw[1]).mean()
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    if parts is not None: parts.append((t1 - t0, t2 - t1, t3 - t2))
for s in range(5): step(s)
torch.cuda.synchronize()
parts = []
t0 = time.perf_counter()
for s in range(5, 25): step(s, parts)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
p = np.array(parts) * 1e3
print(f"host issue time per step {t_issue/20*1e3:.3f} ms (fwd {p[:,0].mean():.3f}, bwd {p[:,1].mean():.3f}, opt {p[:,2].mean():.3f}); wall incl. GPU drain {t_all/20*1e3:.3f} ms")
