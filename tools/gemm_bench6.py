"""Small-M (root layer) NT products: direct kernel vs the tiled kernels.  Run twice:
   python tools/gemm_bench6.py ; FLID_GEMM_TUNE=1 FLID_GEMM_NODIRECT=1 python tools/gemm_bench6.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flid_amd import ops
dev = torch.device("cuda:0")
def run(M, N, K, reps=50):
    a = torch.randn((M, K), device=dev); b = torch.randn((N, K), device=dev); c = torch.empty((M, N), device=dev)
    for _ in range(3): ops.gemm(a, b, c, tb=True)
    ref = a.double() @ b.double().T
    err = float((c.double() - ref).abs().max())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm(a, b, c, tb=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms * 1e3, 2.0 * M * N * K / ms / 1e9, err
shapes = [(1200, 272, 272), (1200, 272, 444), (1200, 444, 136), (1200, 172, 272), (1200, 172, 444), (1200, 136, 444), (600, 272, 444), (2400, 272, 444), (4000, 272, 444)]
print("direct" if not os.environ.get("FLID_GEMM_NODIRECT") else "tiled", "  ".join(f"{s}: {us:5.1f}us {tf:5.1f}TF e={e:.1e}" for s, (us, tf, e) in ((s, run(*s)) for s in shapes)))
