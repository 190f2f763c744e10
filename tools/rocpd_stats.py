#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (`rocprofv3 --kernel-trace -d DIR -o NAME -- python3 bench.py ...` writes NAME_results.db):
per-kernel statistics as CSV (the layout of `--stats`' kernel table) and, with --timeline, every dispatch of the last full step
(between the last two adam_kernel launches) with its start offset, duration, queue and stream.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db --csv profiles/r02_x_kernel_stats.csv --timeline
"""
import argparse
import re
import sqlite3
import statistics


def short(name, n=110):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:n]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--csv")
    ap.add_argument("--timeline", action="store_true")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--marker", default="adam_kernel", help="kernel whose launches delimit steps")
    args = ap.parse_args()
    cur = sqlite3.connect(args.db).cursor()
    rows = list(cur.execute("select name, start, end, queue_id, stream_id, grid_x, workgroup_x, vgpr_count, lds_size from kernels order by start"))
    by = {}
    for r in rows:
        by.setdefault(r[0], []).append(r[2] - r[1])
    total = sum(sum(v) for v in by.values())
    table = sorted(((k, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / total, min(v), max(v), statistics.pstdev(v)) for k, v in by.items()),
                   key=lambda t: -t[2])
    if args.csv:
        with open(args.csv, "w") as f:
            f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n')
            for t in table:
                f.write('"%s",%d,%d,%.6f,%.2f,%d,%d,%.6f\n' % t)
    for t in table[:args.top]:
        print(f"{t[1]:6d} x {t[3] / 1e3:9.1f} us = {t[2] / 1e6:9.3f} ms {t[4]:6.2f}%  {short(t[0])}")
    if args.timeline:
        idx = [i for i, r in enumerate(rows) if args.marker in r[0]]
        if len(idx) >= 4:
            # the period of every step (end of one marker launch to the end of the next): what the bench's wall clock averages over
            per = [(rows[idx[i + 1]][2] - rows[idx[i]][2]) / 1e3 for i in range(len(idx) - 1)]
            tail = per[len(per) // 2:]
            print(f"--- step periods (second half of the run, {len(tail)} steps): mean {sum(tail) / len(tail):.1f} us  min {min(tail):.1f}  max {max(tail):.1f}"
                  f"   [{' '.join('%.0f' % x for x in tail[-12:])}]")
        if len(idx) >= 3:
            a, b = idx[-3], idx[-2]
            t0 = rows[a][2]
            print(f"--- one step: {(rows[b][2] - t0) / 1e3:.1f} us, {b - a} dispatches")
            busy, last_end = {}, {}
            for r in rows[a + 1:b + 1]:
                gap = (r[1] - last_end[r[4]]) / 1e3 if r[4] in last_end else 0.0
                last_end[r[4]] = r[2]
                busy[r[4]] = busy.get(r[4], 0) + (r[2] - r[1])
                print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:7.1f} gap{gap:7.1f} q{r[3]} s{r[4]} g{r[5]:>7} v{r[7]:>3} {short(r[0], 70)}")
            print("busy us per stream:", {k: round(v / 1e3, 1) for k, v in busy.items()})


if __name__ == "__main__":
    main()
