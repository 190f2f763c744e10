"""Row-panel product against pre-split weights (tg_gemm_pk_nt) vs the tile kernel (tg_gemm_f32, NT) on DyGFormer's shapes: time per
launch and the error against float64.    python3 tools/pk_bench.py [rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                         # noqa: E402

from flid_amd import ops                                             # noqa: E402
from flid_amd._lib import check, lib                                 # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 38400
dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(f, n=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


def pack32(w, trans=False):
    from flid_amd._lib import Pack32Job
    N, K = (w.shape[1], w.shape[0]) if trans else w.shape
    out = torch.empty(int(lib().tg_packed32_floats(N, K)), device=w.device)
    jobs = (Pack32Job * 1)(Pack32Job(w.data_ptr(), w.stride(0), N, K, int(trans), out.data_ptr()))
    check(lib().tg_pack32_weights(1, jobs, ops._stream()), "pack32")
    return out


SHAPES = [(600, 200), (200, 200), (800, 200), (200, 800), (200, 496), (200, 600)]
if os.environ.get("PK_SHAPES"):                                      # e.g. PK_SHAPES=888x172,516x172
    SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ["PK_SHAPES"].split(",")]
for N, K in SHAPES:
    a = torch.randn(R, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    pk = pack32(w)
    c0, c1 = torch.empty(R, N, device=dev), torch.zeros(R, N, device=dev)
    f0 = lambda: ops.gemm(a, w, c0, tb=True, bias=b)
    f1 = lambda: check(lib().tg_gemm_pk_nt(R, N, K, a.data_ptr(), K, pk.data_ptr(), c1.data_ptr(), N, b.data_ptr(), ops._stream()), "pk")
    t0, t1 = timeit(f0), timeit(f1)
    ref = a[:2048].double() @ w.double().t() + b.double()
    e0 = float((c0[:2048].double() - ref).abs().max()); e1 = float((c1[:2048].double() - ref).abs().max())
    e1t = float((c1[-300:].double() - (a[-300:].double() @ w.double().t() + b.double())).abs().max())
    print(f"R={R} N={N} K={K}: tile {t0:7.1f} us ({2e-6 * R * N * K / t0:6.1f} TF/s)  pk {t1:7.1f} us ({2e-6 * R * N * K / t1:6.1f} TF/s)  err {e0:.2e} / {e1:.2e} (tail rows {e1t:.2e})")
