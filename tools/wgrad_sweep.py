#!/usr/bin/env python3
"""Sweep of the grouped weight-gradient launch's two knobs (column-tile width, K slices) on a layer's three launches; run with
FLID_GEMM_TUNE=1 (the overrides FLID_WG_TNW / FLID_WG_SLICES are read only in tuning mode)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flid_amd import ops            # noqa: E402
from gemm_bench import time_us      # noqa: E402

from gemm_bench import wgrad_groups   # noqa: E402

dev = torch.device("cuda:0")
for R in (int(v) for v in (sys.argv[1:] or ["13622", "1200"])):
    for name, (jobs, _) in wgrad_groups(R, dev).items():
        os.environ.pop("FLID_WG_TNW", None); os.environ.pop("FLID_WG_SLICES", None)
        base = time_us(lambda: ops.wgrad_group(jobs))
        out = [f"{name[:14]:14s} R={R:6d} model {base:6.1f} |"]
        for tnw in (2, 3):
            os.environ["FLID_WG_TNW"] = str(tnw)
            for sl in (2, 4, 8, 16, 24, 32):
                if sl * 128 > R:
                    continue
                os.environ["FLID_WG_SLICES"] = str(sl)
                out.append(f"t{tnw}s{sl}:{time_us(lambda: ops.wgrad_group(jobs), reps=10):5.1f}")
        print(" ".join(out), flush=True)
