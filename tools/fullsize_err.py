"""per-tensor gradient error of the full-size TGAT fixtures on the GPU path (diagnostic): max |err| / max|g|, entries beyond 1e-4 max|g|"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import fullsize
from conftest import grads_compact_np, load_golden
from flid_amd import engine
from flid_amd.models.TGAT import TGAT
from flid_amd.utils.utils import get_neighbor_sampler
from flid_amd._lib import lib

for name in sys.argv[1:] or ["tgat_B600_full", "tgat_B600_kinkfree"]:
    for mode in (1, 0):
        lib().tg_set_gemm_mode(mode)
        g = load_golden(name)
        data, p, (bs, bd, bt), r = fullsize.tgat_case(g)
        m = TGAT(data.node_raw_features, data.edge_raw_features, get_neighbor_sampler(data, "recent", seed=0), 100, 2, 2, 0.0, "cuda:0")
        m.load_state_dict(p)
        m = m.to("cuda:0").train()
        s, d = m.compute_src_dst_node_temporal_embeddings(bs, bd, bt, 20)
        print(name, "gemm mode", mode, "emb err", float(np.abs(s.detach().cpu().numpy() - g["s_emb"]).max()), float(np.abs(d.detach().cpu().numpy() - g["d_emb"]).max()))
        rr = torch.from_numpy(r).cuda()
        ((s * rr[0]).sum() + (d * rr[1]).sum()).backward()
        mine = grads_compact_np({k: v.grad.cpu().numpy() for k, v in m.named_parameters()})
        for k in sorted(x for x in g if x.startswith("g:")):
            big = max(1.0, float(np.abs(g[k]).max()))
            err = np.abs(mine[k].astype(np.float64) - g[k])
            gs = "gs:" + k[2:]
            rl2 = float(np.sqrt((err ** 2).sum() / max(1e-30, (g[k].astype(np.float64) ** 2).sum())))
            print(f"  {k[2:]:55s} max|g| {big:10.3f}  max err/max|g| {err.max() / big:9.2e}  > 1e-4: {int((err > 1e-4 * big).sum()):5d} / {err.size}"
                  f"  median {np.median(err) / big:8.1e}  p99 {np.quantile(err, 0.99) / big:8.1e}  rel-L2(sample) {rl2:8.1e}"
                  f"  sum err {abs(mine[gs][0] - g[gs][0]) / max(1.0, np.sqrt(g[gs][1])):8.1e}  sumsq rel {abs(mine[gs][1] - g[gs][1]) / max(1.0, g[gs][1]):8.1e}")
lib().tg_set_gemm_mode(1)
