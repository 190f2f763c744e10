"""Shared set-up of the full-size (B = 600) golden cases (tests/golden/make_golden.py::FULL): the graph is rebuilt from
flid_amd.synth with the arguments recorded in the fixture (a checksum of the arrays guards against generator drift), weights
and upstream gradients from the recorded seeds.  Used by the CPU oracle tests and by the GPU tests."""
import zlib

import numpy as np

from oracle import flid_oracle as O

B = 600


def crc(*arrs):
    c = 0
    for a in arrs:
        c = zlib.crc32(np.ascontiguousarray(a).view(np.uint8), c)
    return c


def _check(g, data):
    got = crc(data.src_node_ids, data.dst_node_ids, data.node_interact_times, data.edge_raw_features[:64])
    assert got == int(g["crc"]), "flid_amd.synth no longer reproduces the graph this fixture was generated on"


def tgat_case(g):
    """-> data, params (state_dict-keyed fp32 tensors), (bs, bd, bt), r (2, 600, 172)"""
    from flid_amd.synth import wikipedia_like
    data = wikipedia_like(num_edges=int(g["num_edges"]), seed=0, zero_node_feat=bool(g["zero_node_feat"]))
    _check(g, data)
    p = O.seeded_like(O.tgat_shapes(172, 172, 100, 2), int(g["seed"]), float(g["scale"]))
    if not bool(g["bias_te"]):
        p["time_encoder.w.bias"].zero_()
    if bool(g["kink_free"]):
        O.kink_free_(p)
    lo = int(g["lo"])
    sl = slice(lo, lo + B)
    r = np.random.RandomState(int(g["r_seed"])).standard_normal((2, B, 172)).astype(np.float32)
    return data, p, (data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl]), r


def tgn_case(g):
    """-> data, params, generator of (j or None, batch args); recorded batch j also gets its negatives and r"""
    from flid_amd.synth import reddit_like
    data = reddit_like(num_edges=int(g["num_edges"]), seed=3)
    _check(g, data)
    p = O.seeded_like(O.tgn_shapes(172, 172, 100, 1), int(g["seed"]), float(g["scale"]))
    p["time_encoder.w.bias"].zero_()
    O.kink_free_(p)
    return data, p


def tgn_batches(g, data):
    warm, rec = int(g["warm"]), int(g["rec"])
    for b in range(warm + rec):
        sl = slice(b * B, (b + 1) * B)
        args = (data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl], data.edge_ids[sl])
        if b < warm:
            yield None, args, None, None
        else:
            j = b - warm
            r = np.random.RandomState(int(g["seed"]) + 100 + j).standard_normal((4, B, 172)).astype(np.float32)
            yield j, args, g[f"neg{j}"], r


def tgn_grads_view(g, j):
    """the gradient entries of recorded batch j under the plain 'g:' / 'gs:' names assert_grads_match expects"""
    pre = f"b{j}:"
    return {k[len(pre):]: v for k, v in g.items() if k.startswith(pre)}


def dyg_case(g):
    from flid_amd.synth import reddit_like
    data = reddit_like(num_edges=int(g["num_edges"]), seed=4)
    _check(g, data)
    p = O.seeded_like(O.dyg_shapes(172, 172, 100, 50, 1, 2), int(g["seed"]), float(g["scale"]))
    p["time_encoder.w.bias"].zero_()
    lo = int(g["lo"])
    sl = slice(lo, lo + B)
    r = np.random.RandomState(int(g["r_seed"])).standard_normal((2, B, 172)).astype(np.float32)
    return data, p, (data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl]), r


def backbone_case(g, shapes):
    """TCL / GraphMixer at the BASELINE dims (make_golden.py::gold_tcl_full / gold_mixer_full) -> data, params, (bs, bd, bt), r"""
    from flid_amd.synth import wikipedia_like
    data = wikipedia_like(num_edges=int(g["num_edges"]), seed=0, zero_node_feat=False)
    _check(g, data)
    p = O.seeded_like(shapes, int(g["seed"]), float(g["scale"]))
    lo, nb = int(g["lo"]), int(g["batch"])
    sl = slice(lo, lo + nb)
    r = np.random.RandomState(int(g["r_seed"])).standard_normal((2, nb, 172)).astype(np.float32)
    return data, p, (data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl]), r
