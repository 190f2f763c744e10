"""micro-benchmark of tg_gemm_f32 shapes (MI355X).  python tools/gemm_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flid_amd import ops

dev = torch.device("cuda:0")
shapes = [(0, 1, 12235, 272, 272), (0, 1, 131072, 272, 272), (0, 1, 12235, 272, 172), (0, 0, 12235, 444, 136), (0, 1, 12235, 136, 444),
          (0, 1, 12235, 172, 444), (1, 0, 272, 444, 12235), (0, 0, 12235, 272, 272), (0, 1, 1200, 272, 272), (1, 0, 272, 272, 1200),
          (0, 1, 25200, 272, 272), (0, 1, 50000, 288, 288), (0, 1, 50000, 256, 256)]
for ta, tb, M, N, K in shapes:
    a = torch.randn((K, M) if ta else (M, K), device=dev)
    b = torch.randn((N, K) if tb else (K, N), device=dev)
    c = torch.empty((M, N), device=dev)
    for _ in range(3):
        ops.gemm(a, b, c, ta=bool(ta), tb=bool(tb))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        ops.gemm(a, b, c, ta=bool(ta), tb=bool(tb))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"ta={ta} tb={tb} M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s")
