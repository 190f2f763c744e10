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
    st = buf.cpu().numpy().reshape(-1, 16).astype(np.int64)
    st = st[st[:, 0] != 0]
    n = int((st[-1] != 0).sum())
    # the root layer's launch (16-row workgroups: ceil(1200 / 16) = 75) ran last and overwrote the first rows: the rows behind them
    # still hold the 13.6 k-row launch of the same step
    small = int(os.environ.get("SMALL_BLOCKS", "75"))
    for name, blk in (("root launch", st[:small]), ("tall launch", st[small:])):
        if len(blk) == 0:
            continue
        nn = int((blk[0] != 0).sum())
        d = np.diff(blk, axis=1)[:, :nn - 1]
        tot = blk[:, nn - 1] - blk[:, 0]
        print(f"{name}: {len(blk)} workgroups, {nn} stamps")
        print("  mean interval between stamps (cycles):", np.round(d.mean(0), 0))
        print("  max  interval between stamps (cycles):", d.max(0))
        print(f"  per-workgroup total: mean {tot.mean():.0f}  min {tot.min()}  p50 {np.percentile(tot, 50):.0f}  p90 {np.percentile(tot, 90):.0f}  max {tot.max()}")
        print(f"  first start -> last end: {blk[:, nn - 1].max() - blk[:, 0].min()}   start skew: {blk[:, 0].max() - blk[:, 0].min()}")
        per_xcd = [tot[i::8].mean() for i in range(8)]
        print("  mean total by workgroup number % 8:", np.round(per_xcd, 0))


if __name__ == "__main__":
    main()
