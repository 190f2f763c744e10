"""cProfile of the host side of one TGN / DyGFormer training step (tools/bench_models.py loop)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flid_amd.synth import reddit_like
from flid_amd.utils.utils import get_neighbor_sampler
which = sys.argv[1] if len(sys.argv) > 1 else "tgn"
data = reddit_like(seed=0)
n_train = int(0.7 * data.num_interactions)
sampler = get_neighbor_sampler(data.slice(0, n_train), "recent", seed=0)
torch.manual_seed(0)
if which == "tgn":
    from flid_amd.models.MemoryModel import MemoryModel
    model = MemoryModel(data.node_raw_features, data.edge_raw_features, sampler, 100, "TGN", 1, 2, 0.1, device="cuda:0")
    model.memory_bank.__init_memory_bank__()
else:
    from flid_amd.models.DyGFormer import DyGFormer
    model = DyGFormer(data.node_raw_features, data.edge_raw_features, sampler, 100, 50, 1, 2, 2, 0.1, 32, "cuda:0")
model = model.to("cuda:0").train()
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
B = 600; first = (n_train // B) // 2
rw = torch.randn(2, B, 172, device="cuda:0")
def step(s):
    sl = slice((first + s) * B, (first + s + 1) * B)
    a = (data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl])
    opt.zero_grad(set_to_none=True)
    if which == "tgn":
        se, de = model.compute_src_dst_node_temporal_embeddings(*a, data.edge_ids[sl], True, 20)
    else:
        se, de = model.compute_src_dst_node_temporal_embeddings(*a)
    loss = torch.addcmul(se * rw[0], de, rw[1]).mean()
    loss.backward(); opt.step()
    if which == "tgn":
        model.memory_bank.detach_memory_bank()
for s in range(5): step(s)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for s in range(5, 25): step(s)
torch.cuda.synchronize()
pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(32); print(st.getvalue()[:7000])
