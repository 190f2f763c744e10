import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def grads_compact_np(named):
    """Same compaction as tests/golden/make_golden.py::grads_compact (full tensor if small, strided sample + sums)."""
    out = {}
    for k, g in named.items():
        g = np.asarray(g, dtype=np.float64).reshape(-1)
        if g.size <= 4096:
            out["g:" + k] = g.astype(np.float32)
        else:
            out["g:" + k] = g[:: g.size // 2048].astype(np.float32)
        out["gs:" + k] = np.array([g.sum(), (g * g).sum()], dtype=np.float64)
    return out


def assert_grads_match(gold, named, atol=1e-4, rtol=1e-4, kink_frac=0.005, strict=False):
    """Gradients are long cancelling sums: the absolute tolerance scales with the tensor's largest entry.  A ReLU unit whose
    pre-activation is within rounding of zero may switch between two correct fp32 evaluations and move a handful of entries
    by a visible amount; at most `kink_frac` of a tensor's entries may do so, and only by < 2 % of the largest entry."""
    if strict:          # kink-free fixtures (oracle.kink_free_): every sampled entry within atol * max|g|, no allowance
        rtol, kink_frac = 0.0, 0.0
    mine = grads_compact_np(named)
    keys = [k for k in gold if k.startswith("g:")]
    assert keys, "fixture holds no gradients"
    for k in keys:
        assert k in mine, f"missing gradient for {k[2:]}"
        big = max(1.0, float(np.abs(gold[k]).max()))
        err = np.abs(mine[k].astype(np.float64) - gold[k])
        bad = err > atol * big + rtol * np.abs(gold[k])
        # (a tensor of <= 200 entries -- the time encoder's 100 -- may still have ONE such entry: its gradient entries are scaled by
        # time intervals of up to 2.7e6, so a single flipped unit is visible in one of them)
        assert bad.sum() <= (0 if strict else max(1.0, kink_frac * bad.size)), (k, int(bad.sum()), bad.size, float(err.max()), big)
        assert float(err.max()) <= 0.02 * big, (k, float(err.max()), big)
        s = "gs:" + k[2:]
        scale = max(1.0, np.sqrt(gold[s][1]))
        loose = not (strict or kink_frac < 0.1)             # entries may have moved by up to 2 % of the largest (flipped ReLU units)
        # (the plain sum of a tensor's entries: a flipped unit adds a rank-one term whose entries do not cancel in it -- checked only
        # where every entry was held to atol)
        assert loose or abs(mine[s][0] - gold[s][0]) <= 50 * atol * scale, (s, mine[s], gold[s])
        # (sum of squares: 1e-3 when every entry was held to atol; where flipped ReLU units may move entries by up to 2 % of the largest,
        # the squared norm follows by up to ~1 %)
        assert abs(mine[s][1] - gold[s][1]) <= (1e-2 if loose else 1e-3) * max(1.0, gold[s][1]), (s, mine[s], gold[s])
