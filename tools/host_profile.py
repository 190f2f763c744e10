import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
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
for s in range(40):
    sl = slice((90 + s) * B, (91 + s) * B)
    batches.append((torch.from_numpy(data.src_node_ids[sl].astype(np.int32)).to(dev), torch.from_numpy(data.dst_node_ids[sl].astype(np.int32)).to(dev),
                    torch.from_numpy(data.node_interact_times[sl]).to(dev)))
rw = torch.randn(2, B, 172, device=dev)
prepared = {}
def step(s):
    if s not in prepared:
        prepared[s] = model.prepare_batch(*batches[s], 20)
    if s + 1 < len(batches):
        prepared[s + 1] = model.prepare_batch(*batches[s + 1], 20)
    opt.zero_grad(set_to_none=True)
    se, de_ = model.compute_src_dst_node_temporal_embeddings(prepared.pop(s), None, None, 20)
    loss = torch.addcmul(se * rw[0], de_, rw[1]).mean()
    loss.backward()
    opt.step()
for s in range(5): step(s)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for s in range(5, 25): step(s)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumulative").print_stats(30)
