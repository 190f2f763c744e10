import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flid_amd import ops
M, N, K, ta, tb = (int(x) for x in sys.argv[1:6])
dev = torch.device("cuda:0")
a = torch.randn((K, M) if ta else (M, K), device=dev); b = torch.randn((N, K) if tb else (K, N), device=dev); c = torch.empty((M, N), device=dev)
for _ in range(10):
    ops.gemm(a, b, c, ta=bool(ta), tb=bool(tb))
torch.cuda.synchronize()
