#!/usr/bin/env python3
"""Stand-alone timing of tg_attn_fwd / tg_attn_bwd at the BASELINE shape (layer-1 launch of a 600-edge TGAT batch on the
Wikipedia-shape graph: ~12 k distinct instances, k = 20, 172/172/100) and at the root-layer shape (1 200 instances, gathering
rows of the layer-1 output).  HIP events on the launch stream, median of --iters launches.  FLID_TG_LIB selects the build."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--fast", type=int, default=None, help="tg_set_attn_fast mask")
    args = ap.parse_args()
    from flid_amd import engine, ops
    from flid_amd._lib import lib
    from flid_amd.synth import wikipedia_like
    from flid_amd.utils.utils import get_neighbor_sampler
    if args.fast is not None:
        lib().tg_set_attn_fast(args.fast)
    dev = torch.device("cuda:0")
    data = wikipedia_like(seed=0)
    n_train = int(0.7 * data.num_interactions)
    sampler = get_neighbor_sampler(data.slice(0, n_train), "recent", seed=0)
    node = torch.from_numpy(data.node_raw_features).to(dev)
    edge = torch.from_numpy(data.edge_raw_features).to(dev)
    b = (n_train // 600) * 3 // 4
    sl = slice(b * 600, (b + 1) * 600)
    ids = torch.from_numpy(np.concatenate([data.src_node_ids[sl], data.dst_node_ids[sl]]).astype(np.int32)).to(dev)
    t = torch.from_numpy(np.concatenate([data.node_interact_times[sl]] * 2)).to(dev)
    fr = engine.sample_frontier(sampler.graph, ids, t, 20, 2)
    S_nbr, S_eid, S_t, S_dt = fr.S
    R1, R2 = fr.rows(1), fr.rows(0)
    te_w = torch.from_numpy((1 / 10 ** np.linspace(0, 9, 100)).astype(np.float32)).to(dev)
    te_b = torch.zeros(100, device=dev)
    H, dk = 2, 444
    torch.manual_seed(0)
    H1 = torch.randn(R1, 172, device=dev)
    cases = {
        "layer1": (R1, node, S_nbr[:R1].reshape(-1), None),
        "root": (R2, H1, fr.child[:R2 * 20], torch.zeros_like(H1)),
    }
    for name, (R, feat, fidx, dfeat) in cases.items():
        a = ops.AttnArgs(feat, fidx.contiguous(), edge, S_eid[:R].reshape(-1).contiguous(), S_nbr[:R].reshape(-1).contiguous(),
                         S_dt[:R].reshape(-1).contiguous(), te_w, te_b, 20, H, 136 ** -0.5, 0.1, 1234)
        u = torch.randn(R, H, dk, device=dev) * 0.1
        dagg = torch.randn(R, H, dk, device=dev) * 0.1
        res = {}
        for which in ("fwd", "bwd"):
            ts = []
            for it in range(args.iters + 3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if which == "fwd":
                    e0.record()
                    agg, prob = ops.attn_fwd(a, u)
                    e1.record()
                else:
                    if dfeat is not None:
                        dfeat.zero_()
                    du = torch.empty_like(u)
                    part = torch.empty((lib().tg_attn_bwd_parts(a.m), 200), device=dev)
                    import ctypes as C
                    from flid_amd._lib import check
                    from flid_amd.ops import _p, _stream
                    e0.record()
                    check(lib().tg_attn_bwd(C.byref(a.desc), _p(u), _p(agg), _p(prob), _p(dagg), _p(du), _p(dfeat),
                                            0 if dfeat is None else dfeat.stride(0), int(fr.pad_rows[0]) if dfeat is not None else -1,
                                            _p(None), 0, _p(part), _stream()), "bwd")
                    e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            res[which] = float(np.median(ts[3:]))
        fb = R * (20 * 1376 + 20 * 16 + 2 * H * dk * 4 + H * 20 * 4)
        bb = R * (20 * 1376 + 20 * 16 + 4 * H * dk * 4 + H * 20 * 4)
        print(f"{name:7s} R={R:6d}  fwd {res['fwd']:7.1f} us ({fb / res['fwd'] / 1e6:5.2f} TB/s)   bwd {res['bwd']:7.1f} us ({bb / res['bwd'] / 1e6:5.2f} TB/s)")


if __name__ == "__main__":
    main()
