"""cProfile of the steady-state training step's HOST side for the autograd-driven backbones (python tools/host_profile_model.py
dygformer|tcl|graphmixer [steps]): their bench lines are bound by this thread."""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flid_amd.synth import reddit_like
from flid_amd.utils.utils import get_neighbor_sampler

name = sys.argv[1] if len(sys.argv) > 1 else "dygformer"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
data = reddit_like(num_edges=200000, seed=0)
sampler = get_neighbor_sampler(data, "recent", seed=0)
if name == "dygformer":
    from flid_amd.models.DyGFormer import DyGFormer
    model = DyGFormer(data.node_raw_features, data.edge_raw_features, sampler, 100, 50, 1, 2, 2, 0.1, 32, "cuda:0")
    call = lambda sl: model.compute_src_dst_node_temporal_embeddings(data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl])
elif name == "tcl":
    from flid_amd.models.TCL import TCL
    model = TCL(data.node_raw_features, data.edge_raw_features, sampler, 100, 2, 2, 21, 0.1, "cuda:0")
    call = lambda sl: model.compute_src_dst_node_temporal_embeddings(data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl], 20)
else:
    from flid_amd.models.GraphMixer import GraphMixer
    model = GraphMixer(data.node_raw_features, data.edge_raw_features, sampler, 100, 20, 2, 0.5, 4.0, 0.1, "cuda:0")
    call = lambda sl: model.compute_src_dst_node_temporal_embeddings(data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl], 20)
model = model.to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
B = 600
rw = torch.randn(2, B, 172, device=dev)


def step(s):
    sl = slice((100 + s) * B, (101 + s) * B)
    opt.zero_grad(set_to_none=True)
    a, b = call(sl)
    torch.addcmul(a * rw[0], b, rw[1]).mean().backward()
    opt.step()


for s in range(5):
    step(s)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for s in range(5, 5 + N):
    step(s)
pr.disable()
torch.cuda.synchronize()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(32)
print(out.getvalue())
