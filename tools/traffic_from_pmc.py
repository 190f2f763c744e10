"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) of
`python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline` into profiles/traffic_r01.json (HBM bytes per launch per kernel family).

    python tools/traffic_from_pmc.py gpurun_out/pmcB_FETCH_SIZE gpurun_out/pmcB_WRITE_SIZE profiles/traffic_r01.json

gfx950 correction: FETCH_SIZE is reported in KB assuming 64-B requests while the requests are 128 B -> doubled.  WRITE_SIZE is KB.
Attention kernels: only the layer-1 launches (the largest grid of each kernel) are averaged, the ones bench.py's roofline is
dominated by; GEMM: mean over every GEMM launch."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    rows = defaultdict(list)          # kernel name -> [(grid, value)]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            rows[r["Kernel_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    return rows


def family(name):
    if "attn_fwd_kernel" in name:
        return "attn_fwd"
    if "attn_bwd_kernel" in name:
        return "attn_bwd"
    if "gemm" in name:
        return "gemm"
    return None


def reduce(rows, big_only):
    acc = defaultdict(list)
    for name, lst in rows.items():
        fam = family(name)
        if fam:
            acc[fam] += lst
    out = {}
    for fam, lst in acc.items():
        if big_only(fam):
            g = max(x[0] for x in lst)
            lst = [x for x in lst if x[0] == g]
        out[fam] = sum(x[1] for x in lst) / len(lst)
    return out


def main():
    fdir, wdir, dst = sys.argv[1:4]
    big = lambda fam: fam.startswith("attn")
    fetch = reduce(load(fdir, "FETCH_SIZE"), big)
    write = reduce(load(wdir, "WRITE_SIZE"), big)
    res = {}
    for fam in sorted(fetch):
        res[fam] = {"fetch_size_kb_raw": round(fetch[fam], 1), "write_size_kb": round(write.get(fam, 0.0), 1),
                    "hbm_bytes_per_launch": int((2 * fetch[fam] + write.get(fam, 0.0)) * 1024),
                    "note": ("FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); " +
                             ("layer-1 launches only" if big(fam) else "mean over all GEMM launches of a step"))}
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
