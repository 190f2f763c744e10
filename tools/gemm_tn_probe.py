"""Weight-gradient (A^T B, split contraction) shapes of the TGAT step, a few launches each: run under
   rocprofv3 --kernel-trace --pmc FETCH_SIZE  to read the memory-side bytes per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flid_amd import ops
dev = torch.device("cuda:0")
K = 12235
for (M, N) in ((172, 172), (272, 272), (272, 444), (444, 272)):
    a = torch.randn((K, M), device=dev); b = torch.randn((K, N), device=dev); c = torch.zeros((M, N), device=dev)
    for _ in range(5):
        ops.gemm(a, b, c, ta=True)
    torch.cuda.synchronize()
    ref = a.double().T @ b.double()
    print(M, N, float((c.double() - ref).abs().max()))
