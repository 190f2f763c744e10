"""Secondary measurement lines: TGN and DyGFormer through the same class API the reference's
trainers use (host numpy ids in, device embeddings out), fwd + bwd + Adam per 600-edge batch on the Reddit-shape synthetic graph.
The headline metric stays bench.py (TGAT).      python tools/bench_models.py [--model tgn|dygformer] [--steps K --warmup W]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="tgn", choices=["tgn", "dygformer"])
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=600)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--gemm-mode", type=int, default=None)
    args = ap.parse_args()
    if args.gemm_mode is not None:
        from flid_amd._lib import lib
        lib().tg_set_gemm_mode(args.gemm_mode)
    from flid_amd.synth import reddit_like
    from flid_amd.utils.utils import get_neighbor_sampler
    dev = torch.device("cuda:0")
    data = reddit_like(seed=0)
    n_train = int(0.7 * data.num_interactions)
    sampler = get_neighbor_sampler(data.slice(0, n_train), "recent", seed=0)
    torch.manual_seed(0)
    if args.model == "tgn":
        from flid_amd.models.MemoryModel import MemoryModel
        model = MemoryModel(data.node_raw_features, data.edge_raw_features, sampler, 100, "TGN", 1, 2, args.dropout, device="cuda:0")
        model.memory_bank.__init_memory_bank__()
    else:
        from flid_amd.models.DyGFormer import DyGFormer
        model = DyGFormer(data.node_raw_features, data.edge_raw_features, sampler, 100, 50, 1, 2, 2, args.dropout, 32, "cuda:0")
    model = model.to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
    B = args.batch
    first = (n_train // B) // 2
    rw = torch.randn(2, B, 172, device=dev)

    def step(s):
        sl = slice((first + s) * B, (first + s + 1) * B)
        a = (data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl])
        opt.zero_grad(set_to_none=True)
        if args.model == "tgn":
            se, de = model.compute_src_dst_node_temporal_embeddings(*a, data.edge_ids[sl], True, 20)      # M_step.py:224-257 order
        else:
            se, de = model.compute_src_dst_node_temporal_embeddings(*a)
        loss = torch.addcmul(se * rw[0], de, rw[1]).mean()
        loss.backward()
        opt.step()
        if args.model == "tgn":
            model.memory_bank.detach_memory_bank()                                                        # M_step.py:325

    for s in range(args.warmup):
        step(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.warmup, args.warmup + args.steps):
        step(s)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"metric": f"edges/sec (temporal-embedding fwd+bwd), {args.model} Reddit-shape", "value": round(args.steps * B / el, 1),
                      "unit": "edges/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 3),
                      "dtype": "f32", "data": "synthetic",
                      "config": {"workload": f"Reddit-shape synthetic (10984 nodes, 672447 edges) + {args.model}, batch {B}, dropout {args.dropout}, "
                                             "host numpy ids per call, fwd+bwd+Adam"}}), flush=True)


if __name__ == "__main__":
    main()
