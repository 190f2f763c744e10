#!/usr/bin/env python3
"""Where the HOST spends a step: every C entry point of libflid_tg.so is wrapped with a timer (calls, total us), the rest of the
wall time between two synchronisation points is Python / torch.  Usage: python tools/host_prof.py [bench.py arguments]"""
import collections
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flid_amd import _lib      # noqa: E402

acc = collections.defaultdict(lambda: [0, 0.0])


class Proxy:
    def __init__(self, h):
        object.__setattr__(self, "_h", h)
        object.__setattr__(self, "_w", {})

    def __getattr__(self, name):
        w = self._w.get(name)
        if w is None:
            fn = getattr(self._h, name)
            rec = acc[name]

            def w(*a):
                t = time.perf_counter()
                r = fn(*a)
                rec[1] += time.perf_counter() - t
                rec[0] += 1
                return r
            self._w[name] = w
        return w


def main():
    h = _lib.lib()
    _lib._lib = Proxy(h)
    import torch
    import bench
    # host blocking points of the prefetch pipeline: Event.synchronize (the distinct-row count of a prepared batch), Stream.synchronize
    for cls, name in ((torch.cuda.Event, "synchronize"), (torch.cuda.Stream, "synchronize"), (torch.cuda.Stream, "wait_event"),
                      (torch.cuda.Stream, "wait_stream"), (torch.cuda.Event, "record")):
        orig = getattr(cls, name)
        rec = acc[f"torch {cls.__name__}.{name}"]

        def make(orig=orig, rec=rec):
            def w(*a, **k):
                t = time.perf_counter()
                r = orig(*a, **k)
                rec[1] += time.perf_counter() - t
                rec[0] += 1
                return r
            return w
        setattr(cls, name, make())
    sys.argv = ["bench.py"] + (sys.argv[1:] or ["--no-cpu-baseline", "--no-breakdown"])
    t0 = time.perf_counter()
    bench.main()
    wall = time.perf_counter() - t0
    tot = sum(v[1] for v in acc.values())
    print(f"[host_prof] wall {wall:.2f} s (whole program), inside C entry points {tot:.3f} s", file=sys.stderr)
    for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:30]:
        print(f"[host_prof] {k:34s} {n:7d} calls {t * 1e6 / max(n, 1):9.1f} us/call {t * 1e3:9.1f} ms", file=sys.stderr)


if __name__ == "__main__":
    main()
