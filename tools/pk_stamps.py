"""Cycle counts per loop phase of the deep pre-split-weight product (wave 0 of workgroup 0), from a build with -DFLID_PK_STAMPS=1:
    bash tools/build_variant.sh stamps tg_gemm_pk.hip -DFLID_PK_STAMPS=1
    FLID_TG_LIB=flid_amd/csrc/variants/libflid_tg_stamps.so python3 tools/pk_stamps.py [rows]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                         # noqa: E402

from flid_amd import ops                                             # noqa: E402
from flid_amd._lib import Pack32Job, check, lib                      # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 38400
N, K = 200, 800
dev = torch.device("cuda:0")
a = torch.randn(R, K, device=dev)
w = torch.randn(N, K, device=dev) * 0.05
pk = torch.empty(int(lib().tg_packed32_floats(N, K)), device=dev)
jobs = (Pack32Job * 1)(Pack32Job(w.data_ptr(), w.stride(0), N, K, 0, pk.data_ptr()))
check(lib().tg_pack32_weights(1, jobs, ops._stream()), "pack")
c = torch.empty(R, N, device=dev)
h = C.CDLL(os.environ["FLID_TG_LIB"])
for _ in range(3):
    check(lib().tg_gemm_pk_nt(R, N, K, a.data_ptr(), K, pk.data_ptr(), c.data_ptr(), N, None, ops._stream()), "pk")
torch.cuda.synchronize()
out = (C.c_uint64 * 8)()
assert h.tg_pk_stamps_read(out) == 0
names = ["loop top", "wait for the stage's copies", "barrier", "A fragment reads + split", "issue stage i + 2", "B reads + MFMAs"]
nst = (K + 31) // 32
print(f"rows {R}: cycles per stage (sum over {nst} stages / {nst}), wave 0 of workgroup 0")
for n_, v in zip(names, list(out)[:6]):
    print(f"  {n_:32s} {v / nst:8.0f}")
print(f"  total {sum(list(out)[:6]) / nst:.0f} cycles per stage")
