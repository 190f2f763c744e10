#!/usr/bin/env python3
"""Per-line host time of selected functions (sys.settrace line events; only those functions are traced):
   FUNCS=engine._native_backward,engine._native_forward python tools/line_prof.py [bench.py arguments]"""
import collections
import linecache
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import bench  # noqa: E402
import flid_amd  # noqa: E402
from flid_amd import engine  # noqa: E402
from flid_amd.models import MemoryModel, TGAT  # noqa: E402

acc = collections.defaultdict(lambda: [0, 0.0])
targets = {}


def resolve(name):
    obj = {"engine": engine, "MemoryModel": MemoryModel, "TGAT": TGAT}[name.split(".")[0]]
    for part in name.split(".")[1:]:
        obj = getattr(obj, part)
    return getattr(obj, "__func__", obj).__code__


state = {}


def local(frame, event, arg):
    now = time.perf_counter()
    st = state.get(frame)
    if st is not None:
        rec = acc[(frame.f_code, st[0])]
        rec[0] += 1
        rec[1] += now - st[1]
    if event == "line":
        state[frame] = (frame.f_lineno, time.perf_counter())
    elif event == "return":
        state.pop(frame, None)
    return local


def tracer(frame, event, arg):
    if event == "call" and frame.f_code in targets:
        state[frame] = (frame.f_lineno, time.perf_counter())
        return local
    return None


def main():
    for n in os.environ.get("FUNCS", "engine._native_backward").split(","):
        targets[resolve(n)] = n
    sys.argv = ["bench.py"] + sys.argv[1:]
    sys.settrace(tracer)
    bench.main()
    sys.settrace(None)
    for code, name in targets.items():
        rows = [(ln, n, t) for (c, ln), (n, t) in acc.items() if c is code]
        tot = sum(t for _, _, t in rows)
        calls = max((n for _, n, _ in rows), default=1)
        print(f"[line_prof] {name}: {tot * 1e6 / calls:.0f} us per call over {calls} calls", file=sys.stderr)
        for ln, n, t in sorted(rows, key=lambda r: -r[2])[:14]:
            print(f"[line_prof]   line {ln:4d} {n:6d} x {t * 1e6 / n:8.1f} us  {linecache.getline(code.co_filename, ln).strip()[:110]}", file=sys.stderr)


if __name__ == "__main__":
    main()
