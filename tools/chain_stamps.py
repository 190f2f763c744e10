#!/usr/bin/env python3
"""Where a workgroup of the chain kernels spends its time: run the headline step on the stamped build
(FLID_TG_LIB=flid_amd/csrc/variants/libflid_tg_stamps.so, built with -DFLID_CHAIN_STAMPS=1) and print the mean interval between
consecutive s_memtime stamps (shader cycles at 100 MHz... see the kernel for the stamp sites)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                  # noqa: E402
from flid_amd._lib import lib                 # noqa: E402


def main():
    h = C.CDLL(os.environ["FLID_TG_LIB"])
    buf = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda:0")
    sys.argv = [sys.argv[0], "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-breakdown"]
    lib()
    which = os.environ.get("CHAIN", "fwd")
    fn = h.tg_chain_debug_buffer if which == "fwd" else h.tg_chain_debug_buffer_bwd
    fn.argtypes = [C.c_void_p]
    fn(buf.data_ptr())
    bench.main()
    torch.cuda.synchronize()
    st = buf.cpu().numpy().reshape(-1, 16)
    st = st[st[:, 0] != 0]
    d = np.diff(st.astype(np.int64), axis=1)
    n = int((st[0] != 0).sum())
    print("workgroups stamped (last launch):", len(st), "stamps:", n)
    print("mean interval between stamps (memtime ticks):", np.round(d[:, :n - 1].mean(0), 0))
    print("total:", float((st[:, n - 1] - st[:, 0]).mean()))


if __name__ == "__main__":
    main()
