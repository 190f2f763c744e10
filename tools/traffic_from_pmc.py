"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) of
`python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-breakdown` into profiles/traffic_rNN.json (HBM bytes per launch per
kernel family).  Both passes run with `--output-format csv`.

    python tools/traffic_from_pmc.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE profiles/traffic_r02.json "<source>"

`<source>` (commit + the profiled command) is stored in every entry so a reader can tell which build the bytes belong to.

gfx950 correction: FETCH_SIZE is reported in KB assuming 64-B requests while the requests are 128 B -> doubled.  WRITE_SIZE is KB.
`hbm_bytes_per_launch` is the mean over EVERY launch of the family (what bench.py's `achieved` averages over: the big layer-1
launch and the small root launch of the attention kernels alike); `hbm_bytes_layer1_launch` is the largest-grid launches only."""
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
    if "attn_fwd" in name:
        return "attn_fwd"
    if "attn_bwd" in name:
        return "attn_bwd"
    if "chain_fwd" in name:
        return "chain_fwd"
    if "chain_bwd" in name:
        return "chain_bwd"
    if "wgrad2_kernel" in name:
        return "wgrad2"
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
        if big_only:
            # one family = several kernels (fast / generic); "layer 1" = launches whose grid is within 2x of the family's largest
            g = max(x[0] for x in lst)
            lst = [x for x in lst if 2 * x[0] >= g]
        out[fam] = sum(x[1] for x in lst) / len(lst)
    return out


def main():
    fdir, wdir, dst = sys.argv[1:4]
    source = sys.argv[4] if len(sys.argv) > 4 else None
    frows, wrows = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    fetch, write = reduce(frows, False), reduce(wrows, False)
    fetch1, write1 = reduce(frows, True), reduce(wrows, True)
    res = {}
    for fam in sorted(fetch):
        res[fam] = {"fetch_size_kb_raw": round(fetch[fam], 1), "write_size_kb": round(write.get(fam, 0.0), 1),
                    "hbm_bytes_per_launch": int((2 * fetch[fam] + write.get(fam, 0.0)) * 1024),
                    "hbm_bytes_layer1_launch": int((2 * fetch1[fam] + write1.get(fam, 0.0)) * 1024),
                    "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); mean over every launch "
                            "of the family in the profiled steps",
                    "source": source}
    # identity of the attention kernels' source at the time of the passes: bench.py quotes these figures only for the same source
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        import bench
        res["_kernel_src_sha"] = bench.kernel_src_sha()
    except Exception as e:                      # noqa: BLE001
        res["_kernel_src_sha"] = None
        print("kernel_src_sha unavailable:", e, file=sys.stderr)
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
