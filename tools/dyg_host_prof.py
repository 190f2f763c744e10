"""Host issue cost of the native DyGFormer step with the GPU idle at the start of each call (how long the two C calls take to queue
their launches), and the step time.    python3 tools/dyg_host_prof.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flid_amd import ops                                             # noqa: E402
from flid_amd.models.DyGFormer import DyGFormer                      # noqa: E402
from flid_amd.optim import FlatAdam                                  # noqa: E402
from flid_amd.synth import reddit_like                               # noqa: E402
from flid_amd.utils.utils import get_neighbor_sampler                # noqa: E402

dev = torch.device("cuda:0")
data = reddit_like(seed=0)
n_train = int(0.7 * data.num_interactions)
sampler = get_neighbor_sampler(data.slice(0, n_train), "recent", seed=0)
torch.manual_seed(0)
m = DyGFormer(data.node_raw_features, data.edge_raw_features, sampler, 100, 50, 1, 2, 2, 0.1, 32, str(dev)).to(dev).train()
flat = m.flatten_parameters()
opt = FlatAdam([flat], lr=1e-4)
st = m.enable_native_step(600)
B = 600
rw = torch.randn(2 * B, 172, device=dev)
g = rw / (2 * B * 172)
tf, tb = [], []
for it in range(30):
    sl = slice(300000 + it * B, 300000 + (it + 1) * B)
    a = (data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    emb = st.forward(*a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    st.backward(g, optimizer=opt)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    if it >= 5:
        tf.append((t1 - t0, t2 - t0)); tb.append((t3 - t2, t4 - t2))
print("forward:  host issue %.0f us, until done %.0f us" % (1e6 * np.median([x[0] for x in tf]), 1e6 * np.median([x[1] for x in tf])))
print("backward: host issue %.0f us, until done %.0f us" % (1e6 * np.median([x[0] for x in tb]), 1e6 * np.median([x[1] for x in tb])))
