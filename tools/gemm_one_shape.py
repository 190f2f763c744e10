"""one NT product shape, a few launches (for rocprofv3 --pmc):  python tools/gemm_one_shape.py M N K"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flid_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4])
dev = torch.device("cuda:0")
a = torch.randn((M, K), device=dev); b = torch.randn((N, K), device=dev); c = torch.empty((M, N), device=dev)
for _ in range(8):
    ops.gemm(a, b, c, tb=True)
torch.cuda.synchronize()
