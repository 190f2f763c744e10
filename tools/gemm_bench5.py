import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flid_amd import ops
from flid_amd._lib import lib
dev = torch.device("cuda:0")
def run(M,N,K,ta=0,tb=1,reps=20):
    a = torch.randn((K, M) if ta else (M, K), device=dev); b = torch.randn((N, K) if tb else (K, N), device=dev); c = torch.empty((M, N), device=dev)
    for _ in range(3): ops.gemm(a, b, c, ta=bool(ta), tb=bool(tb))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm(a, b, c, ta=bool(ta), tb=bool(tb))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms*1e3, 2.0*M*N*K/ms/1e9
shapes=[(172,172,12235,1,0),(172,272,12235,1,0),(272,272,12235,1,0),(272,444,12235,1,0),(272,172,12235,1,0),(172,172,1200,1,0),(272,444,1200,1,0),(12235,272,272,0,1),(1200,272,272,0,1)]
for mode in (0,1,2):
    lib().tg_set_gemm_mode(mode)
    r=[run(*s) for s in shapes]
    print(f"mode={mode}: " + "  ".join(f"{us:6.1f}us {tf:5.1f}TF" for us,tf in r))
