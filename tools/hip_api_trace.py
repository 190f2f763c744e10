#!/usr/bin/env python3
"""Host-side HIP API calls of the last full step from a rocprofv3 --hip-runtime-trace --kernel-trace rocpd database: prints every
API call longer than --min us with the two calls before it (where the host waits inside the runtime)."""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--min", type=float, default=12.0)
    args = ap.parse_args()
    cur = sqlite3.connect(args.db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    print("tables/views:", [t for t in tabs if "region" in t.lower() or "api" in t.lower() or "kernel" in t.lower()][:20])
    view = "regions" if "regions" in tabs else None
    if view is None:
        return
    cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
    print("columns:", cols)
    rows = list(cur.execute(f"select name, start, end, tid from {view} order by start"))
    adam = [r for r in cur.execute("select start from kernels where name like '%adam_kernel%' order by start")]
    if len(adam) < 3:
        return
    lo, hi = adam[-3][0], adam[-2][0]
    # host timestamps of the window: API calls whose start lies between the two Adam dispatches' GPU starts (same clock domain)
    win = [r for r in rows if lo <= r[1] <= hi]
    main_tid = max(set(r[3] for r in win), key=lambda t: sum(1 for r in win if r[3] == t))
    win = [r for r in win if r[3] == main_tid]
    print(f"{len(win)} API calls on the issuing thread in the window of {(hi - lo) / 1e3:.0f} us; total inside the runtime {sum(r[2] - r[1] for r in win) / 1e3:.0f} us")
    for i, r in enumerate(win):
        if (r[2] - r[1]) / 1e3 >= args.min:
            for q in win[max(0, i - 2):i + 1]:
                print(f"   {(q[1] - lo) / 1e3:9.1f} us  {(q[2] - q[1]) / 1e3:8.1f} us  {q[0]}")
            print("   --")


if __name__ == "__main__":
    main()
