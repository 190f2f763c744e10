import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from flid_amd import ops
from flid_amd._lib import lib, check
dev = torch.device("cuda:0")
def run(M,N,K,relu,reps=20):
    a = torch.randn((M, K), device=dev); b = torch.randn((N, K), device=dev); c = torch.empty((M, N), device=dev)
    call=lambda: check(lib().tg_gemm_f32(0,1,M,N,K,1.0,ops._p(a),K,ops._p(b),K,ops._p(c),N,ops._p(None),relu,0,ops._stream()))
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms*1e3, 2.0*M*N*K/ms/1e9
for shape in [(12235,272,272),(12235,256,256),(131072,256,256),(1200,272,272)]:
    print(shape, "normal %.1f us %.1f TF"%run(*shape,0), " | no steady-state loads %.1f us %.1f TF"%run(*shape,77))
