"""weight-gradient shapes of the merged projections: dP (888 x 172) and dV (272 x 888), K = 12235 rows, per gemm mode"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flid_amd import ops
from flid_amd._lib import lib
dev = torch.device("cuda:0")
def run(M, N, K, reps=30):
    a = torch.randn((K, M), device=dev); b = torch.randn((K, N), device=dev); c = torch.empty((M, N), device=dev)
    for _ in range(3): ops.gemm(a, b, c, ta=True)
    ref = a.double().T @ b.double()
    err = float((c.double() - ref).abs().max() / ref.abs().max())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm(a, b, c, ta=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, err
for mode in (1, 2):
    lib().tg_set_gemm_mode(mode)
    print(f"mode={mode}: " + "  ".join(f"{s}: {us:5.1f}us e={e:.1e}" for s, (us, e) in ((s, run(*s)) for s in ((888, 172, 12235), (272, 888, 12235), (888, 172, 6000)))))
