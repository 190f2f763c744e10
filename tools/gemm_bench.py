#!/usr/bin/env python3
"""Micro-benchmark of the product kernels behind tg_gemm_f32 / tg_wgrad_group on MI355X (one parametrised tool; it replaces the
round-1 scratch scripts gemm_bench2..8 / gemm_one* / gemm_tn_probe).

    python tools/gemm_bench.py                         # the shapes of a TGAT step (main chain + weight gradients), every gemm mode
    python tools/gemm_bench.py --shape 0,1,12235,272,444 --modes 1       # one shape: ta,tb,M,N,K
    python tools/gemm_bench.py --wgrad 13622           # the grouped weight-gradient launches of a layer with that many rows
    FLID_GEMM_TUNE=1 FLID_GEMM_TM=2 FLID_GEMM_TN=3 python tools/gemm_bench.py ...     # tile overrides (read only in tuning mode)
FLID_TG_LIB=/path/to/another/libflid_tg.so selects another build for A/B timing."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flid_amd import ops                    # noqa: E402
from flid_amd._lib import lib               # noqa: E402

STEP_SHAPES = [  # (ta, tb, M, N, K): forward / input-gradient products of layer 1 (R ~ 12-13.6 k rows) and of the root layer (1 200 rows)
    (0, 1, 12235, 888, 172), (0, 1, 12235, 272, 888), (0, 1, 12235, 172, 272), (0, 1, 12235, 172, 172), (0, 1, 12235, 888, 272),
    (0, 1, 1200, 272, 172), (0, 1, 1200, 444, 136), (0, 1, 1200, 272, 272), (0, 1, 1200, 172, 272),
    (1, 0, 272, 888, 12235), (1, 0, 888, 172, 12235), (1, 0, 172, 272, 12235), (1, 0, 172, 172, 12235), (1, 0, 172, 172, 1200)]


def time_us(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def wgrad_groups(R, dev):
    """the three grouped weight-gradient launches of a merged-projection layer's backward (tg_layer.hip): name -> (jobs, M x N summed);
    A = the activation gradient, B = the layer input"""
    f = lambda *s: torch.randn(*s, device=dev)
    dout, f1, df1, yr, dres, ctx, dctx, agg, du, own = f(R, 172), f(R, 172), f(R, 172), f(R, 444), f(R, 272), f(R, 272), f(R, 272), f(R, 888), f(R, 888), f(R, 172)
    z = lambda *s_: torch.zeros(*s_, device=dev)
    W2, W1, Wr, Wv, dP, b = z(172, 172), z(172, 444), z(272, 272), z(272, 444), z(888, 172), z(888)
    g = _three_groups(dout, f1, df1, yr, dres, ctx, dctx, agg, du, own, W2, W1, Wr, Wv, dP, b)
    g["all six of a layer (one launch)"] = (sum((v[0] for v in g.values()), []), sum(v[1] for v in g.values()))
    return g


def _three_groups(dout, f1, df1, yr, dres, ctx, dctx, agg, du, own, W2, W1, Wr, Wv, dP, b):
    return {"merge: dW2, dW1 (+ biases)": ([(dout, f1, W2, b[:172]), (df1, yr, W1, b[:172])], 172 * 172 + 172 * 444),
            "dWr (+ d br), dWv_h x 2": ([(dres, ctx, Wr, b[:272]), (dctx[:, :136], agg[:, :444], Wv[:136], None),
                                         (dctx[:, 136:], agg[:, 444:], Wv[136:], None)], 272 * 272 + 272 * 444),
            "dP = du^T own (+ dub)": ([(du, own, dP, b)], 888 * 172)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", action="append", help="ta,tb,M,N,K (repeatable)")
    ap.add_argument("--modes", default="0,1,2", help="tg_set_gemm_mode values to run")
    ap.add_argument("--wgrad", type=int, default=None, help="rows: time tg_wgrad_group on a layer's weight-gradient groups")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    f = lambda *s: torch.randn(*s, device=dev)
    if args.wgrad is not None:
        R = args.wgrad
        groups = wgrad_groups(R, dev)
        for name, (jobs, mn) in groups.items():
            us = time_us(lambda: ops.wgrad_group(jobs))
            print(f"{name:36s} {us:8.1f} us  {2.0 * mn * R / us / 1e6:7.1f} TFLOP/s")
        return
    shapes = [tuple(int(v) for v in s.split(",")) for s in args.shape] if args.shape else STEP_SHAPES
    for mode in (int(v) for v in args.modes.split(",")):
        lib().tg_set_gemm_mode(mode)
        for ta, tb, M, N, K in shapes:
            a, b, c = f(*((K, M) if ta else (M, K))), f(*((N, K) if tb else (K, N))), torch.empty((M, N), device=dev)
            us = time_us(lambda: ops.gemm(a, b, c, ta=bool(ta), tb=bool(tb)))
            print(f"mode={mode} ta={ta} tb={tb} M={M:6d} N={N:4d} K={K:6d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
    lib().tg_set_gemm_mode(1)


if __name__ == "__main__":
    main()
