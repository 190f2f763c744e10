#!/usr/bin/env python3
"""Sweep of the grouped weight-gradient launch's two knobs (column-tile width, K slices) on a layer's three launches; run with
FLID_GEMM_TUNE=1 (the overrides FLID_WG_TNW / FLID_WG_SLICES are read only in tuning mode)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flid_amd import ops            # noqa: E402
from gemm_bench import time_us      # noqa: E402

dev = torch.device("cuda:0")
f = lambda *s: torch.randn(*s, device=dev)
for R in (int(v) for v in (sys.argv[1:] or ["13622", "1200"])):
    dout, f1, df1, y, raw, dres, agg, du, own = f(R, 172), f(R, 172), f(R, 172), f(R, 272), f(R, 172), f(R, 272), f(R, 888), f(R, 888), f(R, 172)
    W2, W1, dV, dP, b = torch.zeros(172, 172, device=dev), torch.zeros(172, 444, device=dev), torch.zeros(272, 888, device=dev), \
        torch.zeros(888, 172, device=dev), torch.zeros(888, device=dev)
    groups = {"merge": [(dout, f1, W2, b[:172]), (df1, y, W1[:, :272], b[:172]), (df1, raw, W1[:, 272:], None)],
              "dV": [(dres, agg, dV, b[:272])], "dP": [(du, own, dP, b)]}
    for name, jobs in groups.items():
        os.environ.pop("FLID_WG_TNW", None); os.environ.pop("FLID_WG_SLICES", None)
        base = time_us(lambda: ops.wgrad_group(jobs))
        out = [f"{name:6s} R={R:6d} model {base:6.1f} |"]
        for tnw in (2, 3):
            os.environ["FLID_WG_TNW"] = str(tnw)
            for sl in (8, 16, 24, 32, 40, 48, 64):
                if sl * 128 > R:
                    continue
                os.environ["FLID_WG_SLICES"] = str(sl)
                out.append(f"t{tnw}s{sl}:{time_us(lambda: ops.wgrad_group(jobs), reps=10):5.1f}")
        print(" ".join(out), flush=True)
