#!/usr/bin/env python3
"""Check + time the packed-weight row-block product (tg_gemm_rows_nt) against the tile kernel behind tg_gemm_f32 on the shapes of
a TGAT step.   python tools/rows_bench.py [--rows 13622,1200]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flid_amd import ops                    # noqa: E402

SHAPES = [(444, 172), (136, 444), (272, 272), (172, 444), (172, 172), (272, 172), (444, 136), (172, 888 // 2), (172, 272)]   # (N, K)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="13622,1200,600,77")
    ap.add_argument("--quick", action="store_true", help="no checks, no tile kernel: 30 launches per shape (for rocprofv3 runs)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for R in (int(v) for v in args.rows.split(",")):
        for N, K in SHAPES:
            a = torch.randn(R, K, device=dev)
            w = torch.randn(N, K, device=dev) / K ** 0.5
            bias = torch.randn(N, device=dev)
            if args.quick:
                (pk,) = ops.pack_weights([(w, False)])
                out = torch.empty((R, N), device=dev)
                for _ in range(30):
                    ops.gemm_rows(a, pk, out, bias=bias)
                torch.cuda.synchronize()
                continue
            ref = (a.double() @ w.double().t() + bias.double())
            (pk, pkt) = ops.pack_weights([(w, False), (w.t().contiguous(), True)])
            out = torch.full((R, N), float("nan"), device=dev)
            ops.gemm_rows(a, pk, out, bias=bias)
            out_t = torch.full((R, N), float("nan"), device=dev)
            ops.gemm_rows(a, pkt, out_t, bias=bias)
            old = torch.empty((R, N), device=dev)
            ops.gemm(a, w, old, tb=True, bias=bias)
            scale = ref.abs().max().item()
            e_new = (out.double() - ref).abs().max().item() / scale
            e_t = (out_t.double() - ref).abs().max().item() / scale
            e_old = (old.double() - ref).abs().max().item() / scale
            # epilogue variants: relu + accumulate + mask
            base = torch.randn(R, N, device=dev)
            m = torch.randn(R, N, device=dev)
            o2 = base.clone()
            ops.gemm_rows(a, pk, o2, bias=bias, relu=True, accumulate=True, mask=m)
            ref2 = torch.where(m > 0, torch.relu(ref + base.double()), torch.zeros_like(ref))
            e2 = (o2.double() - ref2).abs().max().item() / scale
            t_new = time_us(lambda: ops.gemm_rows(a, pk, out, bias=bias))
            t_old = time_us(lambda: ops.gemm(a, w, old, tb=True, bias=bias))
            flag = "" if max(e_new, e_t, e2) < 3e-5 else "   <-- ERROR"
            print(f"R={R:6d} N={N:4d} K={K:4d}: rows {t_new:7.1f} us  tile {t_old:7.1f} us   err rows {e_new:.1e} (trans {e_t:.1e}, epi {e2:.1e}) tile {e_old:.1e}{flag}", flush=True)
    if args.quick:
        return
    t_pack = time_us(lambda: ops.pack_weights([(torch.empty(n, k, device=dev), False) for n, k in SHAPES]))
    print(f"pack of {len(SHAPES)} weights (incl. allocation): {t_pack:.1f} us")


if __name__ == "__main__":
    main()
