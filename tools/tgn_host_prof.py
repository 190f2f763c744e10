"""Host issue cost of the native TGN step with the GPU idle at the start of each call (how long each host call takes to queue its
launches), per phase.    python3 tools/tgn_host_prof.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flid_amd import ops                                             # noqa: E402
from flid_amd.models.MemoryModel import MemoryModel                  # noqa: E402
from flid_amd.optim import FlatAdam                                  # noqa: E402
from flid_amd.synth import reddit_like                               # noqa: E402
from flid_amd.utils.utils import get_neighbor_sampler                # noqa: E402

dev = torch.device("cuda:0")
data = reddit_like(seed=0)
n_train = int(0.7 * data.num_interactions)
sampler = get_neighbor_sampler(data.slice(0, n_train), "recent", seed=0)
torch.manual_seed(0)
m = MemoryModel(data.node_raw_features, data.edge_raw_features, sampler, 100, "TGN", 1, 2, 0.1, device=str(dev)).to(dev).train()
m.memory_bank.__init_memory_bank__()
flat = m.flatten_parameters()
opt = FlatAdam([flat], lr=1e-4)
m.enable_native_step(600, 20)
B, K = 600, 20
rw = torch.randn(2 * B, 172, device=dev)
g = rw / (2 * B * 172)
out = torch.zeros(1, device=dev)


def loss_fn(emb):
    return ops.weighted_sum(emb, rw, 1.0 / (2 * B * 172), out=out), g


T = {k: [] for k in ("begin", "finish", "step")}
for it in range(40):
    sl = slice(200000 + it * B, 200000 + (it + 1) * B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    job = m.prepare_batch_begin(data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl], K, edge_ids=data.edge_ids[sl])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    pf = m.prepare_batch_finish(job)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    opt.zero_grad(set_to_none=True)
    t4 = time.perf_counter()
    m.train_step(pf, data.edge_ids[sl], loss_fn, K, optimizer=opt)
    t5 = time.perf_counter()
    torch.cuda.synchronize()
    if it >= 8:
        T["begin"].append(t1 - t0); T["finish"].append(t3 - t2); T["step"].append(t5 - t4)
for k, v in T.items():
    print("%-7s host issue %.0f us" % (k, 1e6 * np.median(v)))
