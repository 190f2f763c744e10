#!/usr/bin/env python3
"""Average duration per (kernel, grid size) of a rocprofv3 --kernel-trace database -- for microbenchmarks whose launches differ
only in shape (tools/gemm_bench.py under rocprofv3: the Python loop is host-bound at ~11 us per call, the trace is not).

    python tools/rocpd_by_grid.py x_results.db [name-substring]"""
import re
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
by = {}
order = []
for name, dur, gx, gy, wx in cur.execute("select name, end - start, grid_x, grid_y, workgroup_x from kernels order by start"):
    if pat not in name:
        continue
    key = (re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "")), gx, gy, wx)
    if key not in by:
        by[key] = []
        order.append(key)
    by[key].append(dur)
for key in order:
    v = sorted(by[key])
    print(f"{key[0][:60]:60s} grid {key[1] // key[3]:6d} x {key[2]:3d}  wg {key[3]:4d}  n {len(v):4d}  median {v[len(v) // 2] / 1e3:7.1f} us  min {v[0] / 1e3:7.1f}")
