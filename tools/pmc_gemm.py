"""Workload for `rocprofv3 --pmc ... -- python3 tools/pmc_gemm.py`: five launches each of one NT product and of the three grouped
weight-gradient launches of a 13.6 k-row layer (counters per kernel are read from the rocpd database / CSV of that run)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flid_amd import ops                      # noqa: E402
from gemm_bench import wgrad_groups           # noqa: E402

dev = torch.device("cuda:0")
f = lambda *s: torch.randn(*s, device=dev)
R = 13622
a, b, c = f(R, 888), f(272, 888), torch.empty(R, 272, device=dev)
for _ in range(5):
    ops.gemm(a, b, c, tb=True)
for name, (jobs, _) in wgrad_groups(R, dev).items():
    for _ in range(5):
        ops.wgrad_group(jobs)
torch.cuda.synchronize()
