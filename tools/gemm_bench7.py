"""per-stage cost of the split-bf16 NT kernel: fixed 12235 x 272 output, growing K"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flid_amd import ops
dev = torch.device("cuda:0")
def run(M, N, K, reps=30):
    a = torch.randn((M, K), device=dev); b = torch.randn((N, K), device=dev); c = torch.empty((M, N), device=dev)
    for _ in range(3): ops.gemm(a, b, c, tb=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm(a, b, c, tb=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N) in ((12235, 272), (12235, 444), (38400, 800), (131072, 256)):
    t = [(K, run(M, N, K)) for K in (32, 64, 128, 272, 544, 1088, 4352)]
    print(M, N, "  ".join(f"K={K}: {us:6.1f}us ({2.0*M*N*K/us/1e6:5.1f}TF)" for K, us in t),
          f" per-stage {(t[-1][1]-t[-2][1])/((t[-1][0]-t[-2][0])/32):.2f}us")
