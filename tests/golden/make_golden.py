#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF (CPU PyTorch).

Run in the build container only (needs /root/reference, read-only):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

What is committed is data: inputs (tiny synthetic graphs, query batches, weight SEEDS) and the outputs
the reference classes produced for them.  No reference source travels.  The one stub needed is an in-memory
`utils.DataLoader` module (the shipped file has a SyntaxError at :239 and the hot path only uses its `Data`
class as a type hint) -- SURVEY.md section 8(c).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = "/root/reference"
sys.path.insert(0, REF)

import utils  # noqa: E402  (reference package)

_stub = types.ModuleType("utils.DataLoader")


class Data:  # minimal stand-in used only as a record
    def __init__(self, src, dst, t, eid):
        self.src_node_ids, self.dst_node_ids = src, dst
        self.node_interact_times, self.edge_ids = t, eid


_stub.Data = Data
sys.modules["utils.DataLoader"] = _stub

from utils.utils import get_neighbor_sampler, NegativeEdgeSampler  # noqa: E402
from models.modules import TimeEncoder, MultiHeadAttention  # noqa: E402
from models.TGAT import TGAT  # noqa: E402
from models.MemoryModel import MemoryModel, compute_src_dst_node_time_shifts  # noqa: E402
from models.DyGFormer import DyGFormer  # noqa: E402
from models.TCL import TCL  # noqa: E402
from models.GraphMixer import GraphMixer  # noqa: E402

from oracle import flid_oracle as O  # noqa: E402  (seed formula + shape tables only)

torch.set_num_threads(2)
torch.manual_seed(0)


def toy_graph(seed, n_users=24, n_items=9, n_edges=260, t_max=5.0e4, frac=True, dup=True, isolated=3):
    """Bipartite toy interaction stream.  ids 1..n_users users, then `isolated` ids that never interact, then items.
    Timestamps: sorted, optionally fractional (3 decimals, so float32 rounding of hop times matters) and with
    runs of exact duplicates."""
    rs = np.random.RandomState(seed)
    src = rs.randint(1, n_users + 1, size=n_edges).astype(np.int64)
    lo = n_users + isolated + 1                      # ids n_users+1 .. n_users+isolated never interact
    dst = rs.randint(lo, lo + n_items, size=n_edges).astype(np.int64)
    dst[n_edges // 2] = lo + n_items - 1             # the reference sizes its adjacency by the max id seen
    t = np.sort(rs.uniform(0.0, t_max, size=n_edges))
    t = np.round(t, 3) if frac else np.round(t)
    if dup:
        for s in range(10, n_edges - 6, 37):
            t[s:s + 4] = t[s]
    # a few large float32-inexact stamps late in the stream
    t[-20:] = np.sort(100000.0 + rs.uniform(0, 2.5e6, size=20).round(3))
    eid = np.arange(1, n_edges + 1, dtype=np.int64)
    num_rows = n_users + n_items + isolated + 1
    return src, dst, eid, t.astype(np.float64), num_rows


def grads_compact(named):
    out = {}
    for k, g in named.items():
        g = g.detach().double().reshape(-1)
        if g.numel() <= 4096:
            out["g:" + k] = g.float().numpy()
        else:
            stride = g.numel() // 2048
            out["g:" + k] = g[::stride].float().numpy()
        out["gs:" + k] = np.array([g.sum().item(), (g * g).sum().item()], dtype=np.float64)
    return out


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


# ------------------------------------------------------------------------------------------------- sampler
def gold_sampler():
    src, dst, eid, t, num_rows = toy_graph(1)
    ns = get_neighbor_sampler(Data(src, dst, t, eid), "recent", seed=0)
    rs = np.random.RandomState(5)
    ids = np.concatenate([rs.randint(0, num_rows, size=64), [0, num_rows - 1, int(src[0]), int(dst[-1])]]).astype(np.int64)
    qt = np.concatenate([rs.choice(t, size=32), rs.uniform(0, t.max() * 1.1, size=32), [0.0, 1e9, t[0], t[-1]]])
    out = {"src": src, "dst": dst, "eid": eid, "t": t, "num_rows": np.int64(num_rows), "ids": ids, "qt64": qt}
    for k in (1, 3, 20):
        a, b, c = ns.get_historical_neighbors(ids, qt, k)
        out[f"k{k}_n"], out[f"k{k}_e"], out[f"k{k}_t"] = a, b, c
        # hop 2: the reference feeds float32 neighbor times straight back in (TGAT.py:110-111)
        a2, b2, c2 = ns.get_historical_neighbors(a.flatten(), c.flatten(), k)
        out[f"k{k}_n2"], out[f"k{k}_e2"], out[f"k{k}_t2"] = a2, b2, c2
    la, lb, lc = ns.get_all_first_hop_neighbors(ids, qt)
    out["fh_len"] = np.array([len(x) for x in la], dtype=np.int64)
    out["fh_n"] = np.concatenate(la) if len(la) else np.zeros(0, np.int64)
    out["fh_e"] = np.concatenate(lb)
    out["fh_t"] = np.concatenate(lc)
    # uniform / time_interval_aware, seed 1: two consecutive calls, then reset
    for strat, tsf in (("uniform", 0.0), ("time_interval_aware", 1e-4)):
        nu = get_neighbor_sampler(Data(src, dst, t, eid), strat, time_scaling_factor=tsf, seed=1)
        for call in (0, 1):
            a, b, c = nu.get_historical_neighbors(ids, qt, 5)
            out[f"{strat}{call}_n"], out[f"{strat}{call}_e"], out[f"{strat}{call}_t"] = a, b, c
        nu.reset_random_state()
        a, b, c = nu.get_historical_neighbors(ids, qt, 5)
        out[f"{strat}R_n"] = a
    save("sampler", **out)


def gold_neg_sampler():
    """NegativeEdgeSampler (utils/utils.py:305-495): the draws of the three strategies for fixed seeds"""
    src, dst, eid, t, num_rows = toy_graph(21, n_users=12, n_items=7, n_edges=120, isolated=0)
    out = {"src": src, "dst": dst, "t": t}
    half = len(src) // 2
    for strat in ("random", "historical", "inductive"):
        ns = NegativeEdgeSampler(src, dst, interact_times=t, last_observed_time=float(t[half]), negative_sample_strategy=strat, seed=3)
        for call, (lo, hi) in enumerate(((60, 80), (80, 100), (100, 120))):
            a, b = ns.sample(size=hi - lo, batch_src_node_ids=src[lo:hi], batch_dst_node_ids=dst[lo:hi],
                             current_batch_start_time=float(t[lo]), current_batch_end_time=float(t[hi - 1]))
            out[f"{strat}{call}_s"], out[f"{strat}{call}_d"] = np.asarray(a, dtype=np.int64), np.asarray(b, dtype=np.int64)
        ns.reset_random_state()
        a, b = ns.sample(size=20, batch_src_node_ids=src[60:80], batch_dst_node_ids=dst[60:80],
                         current_batch_start_time=float(t[60]), current_batch_end_time=float(t[79]))
        out[f"{strat}R_s"], out[f"{strat}R_d"] = np.asarray(a, dtype=np.int64), np.asarray(b, dtype=np.int64)
    save("neg_sampler", **out)


# ------------------------------------------------------------------------------------------------- time encoder
def gold_time_encoder():
    out = {}
    rs = np.random.RandomState(2)
    dts = np.array([0.0, 1.0, 3600.0, 2.678e6, 86400.5, 123456.789, 17.25, 999999.0], dtype=np.float32)
    grid = np.concatenate([dts, rs.uniform(0, 2.7e6, size=56).astype(np.float32)]).reshape(8, 8)
    for tag, bias in (("b0", False), ("b1", True)):
        enc = TimeEncoder(time_dim=100)
        if bias:
            with torch.no_grad():
                enc.w.bias.copy_(torch.from_numpy(rs.uniform(-1, 1, size=100).astype(np.float32)))
                enc.w.weight.mul_(torch.from_numpy((1 + 0.05 * rs.standard_normal((100, 1))).astype(np.float32)))
        out[tag + "_w"] = enc.w.weight.detach().numpy().copy()
        out[tag + "_b"] = enc.w.bias.detach().numpy().copy()
        with torch.no_grad():
            out[tag + "_bk"] = enc(torch.from_numpy(grid)).numpy()                          # (B,K) call shape
            out[tag + "_b1"] = enc(torch.from_numpy(grid.reshape(-1, 1))).numpy()           # (B,1) call shape
    out["grid"] = grid
    save("time_encoder", **out)


# ------------------------------------------------------------------------------------------------- attention
def gold_attention():
    dn, de, dt, heads, n, k = 8, 6, 4, 2, 7, 5
    rs = np.random.RandomState(3)
    mha = MultiHeadAttention(dn, de, dt, num_heads=heads, dropout=0.0)
    shapes = {k_: tuple(v.shape) for k_, v in mha.state_dict().items()}
    params = O.seeded_like(shapes, seed=30, scale=0.3)
    mha.load_state_dict(params)
    f = lambda *s: torch.from_numpy(rs.standard_normal(s).astype(np.float32)).requires_grad_(True)
    node, ntime, nbr, nbrt, nbre = f(n, dn), f(n, 1, dt), f(n, k, dn), f(n, k, dt), f(n, k, de)
    ids = rs.randint(1, 9, size=(n, k)).astype(np.int64)
    ids[1, :3] = 0          # partly padded (front)
    ids[2, :] = 0           # all padded
    ids[5, :4] = 0
    out, sc = mha(node, ntime, nbr, nbrt, nbre, ids)
    r = torch.from_numpy(rs.standard_normal(tuple(out.shape)).astype(np.float32))
    (out * r).sum().backward()
    arrs = {"node": node, "ntime": ntime, "nbr": nbr, "nbrt": nbrt, "nbre": nbre}
    save("attention", ids=ids, r=r.numpy(), out=out.detach().numpy(), scores=sc.detach().numpy(),
         seed=np.int64(30), scale=np.float64(0.3), dims=np.array([dn, de, dt, heads]),
         **{k_: v.detach().numpy() for k_, v in arrs.items()},
         **{"gi:" + k_: v.grad.numpy() for k_, v in arrs.items()},
         **grads_compact({k_: p.grad for k_, p in mha.named_parameters()}))


# ------------------------------------------------------------------------------------------------- TGAT
def run_tgat(tag, dn, de, dt, layers, k, batch, seed, graph_seed, bias_te=True, scale=0.15):
    src, dst, eid, t, num_rows = toy_graph(graph_seed)
    rs = np.random.RandomState(seed)
    node_feat = rs.standard_normal((num_rows, dn)).astype(np.float32)
    edge_feat = rs.standard_normal((len(eid) + 1, de)).astype(np.float32)
    node_feat[0] = 0
    edge_feat[0] = 0
    ns = get_neighbor_sampler(Data(src, dst, t, eid), "recent", seed=0)
    model = TGAT(node_feat, edge_feat, ns, time_feat_dim=dt, num_layers=layers, num_heads=2, dropout=0.0)
    shapes = {k_: tuple(v.shape) for k_, v in model.state_dict().items()}
    assert shapes == O.tgat_shapes(dn, de, dt, layers), "state_dict contract drifted"
    params = O.seeded_like(shapes, seed=seed, scale=scale)
    if not bias_te:
        params["time_encoder.w.bias"].zero_()
    model.load_state_dict(params)
    model.train()
    pick = np.sort(rs.choice(len(eid), size=batch, replace=False))
    pick[-1] = len(eid) - 1
    bs, bd, bt = src[pick], dst[pick], t[pick]
    if batch >= 4:      # roots with no history and the padding id itself
        bs[0], bt[0] = src[0], t[0]
        bd[1] = num_rows - 1
    s_emb, d_emb = model.compute_src_dst_node_temporal_embeddings(bs, bd, bt, num_neighbors=k)
    r = rs.standard_normal((2, batch, dn)).astype(np.float32)
    (s_emb * torch.from_numpy(r[0])).sum().add((d_emb * torch.from_numpy(r[1])).sum()).backward()
    save(tag, src=src, dst=dst, eid=eid, t=t, num_rows=np.int64(num_rows), node_feat=node_feat, edge_feat=edge_feat,
         dims=np.array([dn, de, dt, layers, k]), seed=np.int64(seed), scale=np.float64(scale), bias_te=np.bool_(bias_te),
         bs=bs, bd=bd, bt=bt, r=r, s_emb=s_emb.detach().numpy(), d_emb=d_emb.detach().numpy(),
         keys=np.array(sorted(shapes)), **grads_compact({k_: p.grad for k_, p in model.named_parameters()}))


# ------------------------------------------------------------------------------------------------- TGN
def gold_tgn():
    dn = de = 8
    dt, layers, k = 4, 1, 4
    src, dst, eid, t, num_rows = toy_graph(11, n_users=10, n_items=5, n_edges=90, isolated=2)
    # make one node act in both roles inside a batch: rewrite a few dst to user ids
    dst = dst.copy()
    dst[[13, 14, 40, 41]] = src[[14, 13, 41, 40]]
    rs = np.random.RandomState(12)
    node_feat = rs.standard_normal((num_rows, dn)).astype(np.float32)
    edge_feat = rs.standard_normal((len(eid) + 1, de)).astype(np.float32)
    node_feat[0] = 0
    edge_feat[0] = 0
    ns = get_neighbor_sampler(Data(src, dst, t, eid), "recent", seed=0)
    model = MemoryModel(node_feat, edge_feat, ns, time_feat_dim=dt, model_name="TGN", num_layers=layers,
                        num_heads=2, dropout=0.0)
    full = {k_: tuple(v.shape) for k_, v in model.state_dict().items()}
    shapes = O.tgn_shapes(dn, de, dt, layers)
    params = O.seeded_like(shapes, seed=13, scale=0.2)
    sd = dict(params)
    sd["embedding_module.time_encoder.w.weight"] = params["time_encoder.w.weight"]
    sd["embedding_module.time_encoder.w.bias"] = params["time_encoder.w.bias"]
    missing = model.load_state_dict(sd, strict=False)
    assert all("memory_bank" in m for m in missing.missing_keys), missing
    model.train()
    out = dict(src=src, dst=dst, eid=eid, t=t, num_rows=np.int64(num_rows), node_feat=node_feat, edge_feat=edge_feat,
               dims=np.array([dn, de, dt, layers, k]), seed=np.int64(13), scale=np.float64(0.2),
               keys=np.array(sorted(full)))
    bsz, nb = 12, 7
    model.memory_bank.__init_memory_bank__()
    backup = None
    for b in range(nb):
        sl = slice(b * bsz, (b + 1) * bsz)
        bs, bd, bt, be = src[sl], dst[sl], t[sl], eid[sl]
        neg = rs.randint(13, 18, size=bsz).astype(np.int64)
        out[f"neg{b}"] = neg
        # link-prediction call order of the reference trainer: negatives first, no state change
        ns_emb, nd_emb = model.compute_src_dst_node_temporal_embeddings(bs, neg, bt, None, False, k)
        ps_emb, pd_emb = model.compute_src_dst_node_temporal_embeddings(bs, bd, bt, be, True, k)
        if b == 3:   # gradients through GRU / attention on one batch
            r = rs.standard_normal((4, bsz, dn)).astype(np.float32)
            out["r3"] = r
            model.zero_grad()
            loss = sum((e * torch.from_numpy(r[i])).sum() for i, e in enumerate((ns_emb, nd_emb, ps_emb, pd_emb)))
            loss.backward()
            out.update(grads_compact({k_: p.grad for k_, p in model.named_parameters() if p.grad is not None}))
        model.memory_bank.detach_memory_bank()
        out[f"ns{b}"], out[f"nd{b}"] = ns_emb.detach().numpy(), nd_emb.detach().numpy()
        out[f"ps{b}"], out[f"pd{b}"] = ps_emb.detach().numpy(), pd_emb.detach().numpy()
        out[f"mem{b}"] = model.memory_bank.node_memories.detach().numpy().copy()
        out[f"lu{b}"] = model.memory_bank.node_last_updated_times.detach().numpy().copy()
        has = np.zeros(num_rows, dtype=bool)
        pm = np.zeros((num_rows, 2 * dn + dt + de), dtype=np.float32)
        pt = np.zeros(num_rows, dtype=np.float64)
        for nid, lst in model.memory_bank.node_raw_messages.items():
            if len(lst):
                has[nid], pm[nid], pt[nid] = True, lst[-1][0].detach().numpy(), lst[-1][1]
        out[f"has{b}"], out[f"pm{b}"], out[f"pt{b}"] = has, pm, pt
        if b == 4:
            backup = model.memory_bank.backup_memory_bank()
    # backup -> advance -> reload equality is asserted in the test from mem4/lu4/pm4; record a post-reload batch
    model.memory_bank.reload_memory_bank(backup)
    sl = slice(5 * bsz, 6 * bsz)
    with torch.no_grad():
        a, b_ = model.compute_src_dst_node_temporal_embeddings(src[sl], dst[sl], t[sl], eid[sl], True, k)
    out["reload_ps5"], out["reload_pd5"] = a.numpy(), b_.numpy()
    out["reload_mem5"] = model.memory_bank.node_memories.detach().numpy().copy()
    # the past-time assertion: replay an old batch after state has advanced
    try:
        with torch.no_grad():
            model.compute_src_dst_node_temporal_embeddings(src[:bsz], dst[:bsz], t[:bsz] * 0.0 - 5.0, eid[:bsz], True, k)
            model.compute_src_dst_node_temporal_embeddings(src[:bsz], dst[:bsz], t[:bsz] * 0.0 - 9.0, eid[:bsz], True, k)
        out["past_assert"] = np.bool_(False)
    except AssertionError as e:
        out["past_assert"] = np.bool_(True)
        out["past_msg"] = np.array(str(e))
    save("tgn_small", **out)


# ------------------------------------------------------------------------------------------------- DyGFormer
def run_dyg(tag, patch, max_len, seed, graph_seed):
    dn, de, dt, c, layers, heads = 8, 6, 4, 6, 2, 2
    src, dst, eid, t, num_rows = toy_graph(graph_seed)
    rs = np.random.RandomState(seed)
    node_feat = rs.standard_normal((num_rows, dn)).astype(np.float32)
    edge_feat = rs.standard_normal((len(eid) + 1, de)).astype(np.float32)
    node_feat[0] = 0
    edge_feat[0] = 0
    ns = get_neighbor_sampler(Data(src, dst, t, eid), "recent", seed=0)
    model = DyGFormer(node_feat, edge_feat, ns, time_feat_dim=dt, channel_embedding_dim=c, patch_size=patch,
                      num_layers=layers, num_heads=heads, dropout=0.0, max_input_sequence_length=max_len)
    shapes = {k_: tuple(v.shape) for k_, v in model.state_dict().items()}
    assert shapes == O.dyg_shapes(dn, de, dt, c, patch, layers), "state_dict contract drifted"
    params = O.seeded_like(shapes, seed=seed, scale=0.2)
    model.load_state_dict(params)
    model.train()
    batch = 10
    pick = np.sort(rs.choice(len(eid), size=batch, replace=False))
    bs, bd, bt = src[pick].copy(), dst[pick].copy(), t[pick].copy()
    bs[0], bt[0] = src[0], t[0]                  # no history on either side
    bd[0] = dst[0]
    bd[1] = bs[1]                                # identical sequences -> co-occurrence with itself
    la, lb, lc = ns.get_all_first_hop_neighbors(bs, bt)
    pn, pe, pt = model.pad_sequences(bs, bt, la, lb, lc, patch, max_len)
    la2, lb2, lc2 = ns.get_all_first_hop_neighbors(bd, bt)
    qn, qe, qt = model.pad_sequences(bd, bt, la2, lb2, lc2, patch, max_len)
    sc, dc = model.neighbor_co_occurrence_encoder.count_nodes_appearances(pn, qn)
    with torch.no_grad():
        _, ef, tf = model.get_features(bt, pn, pe, pt, model.time_encoder)
    s_emb, d_emb = model.compute_src_dst_node_temporal_embeddings(bs, bd, bt)
    r = rs.standard_normal((2, batch, dn)).astype(np.float32)
    (s_emb * torch.from_numpy(r[0])).sum().add((d_emb * torch.from_numpy(r[1])).sum()).backward()
    save(tag, src=src, dst=dst, eid=eid, t=t, num_rows=np.int64(num_rows), node_feat=node_feat, edge_feat=edge_feat,
         dims=np.array([dn, de, dt, c, patch, layers, heads, max_len]), seed=np.int64(seed), scale=np.float64(0.2),
         bs=bs, bd=bd, bt=bt, r=r, pn=pn, pe=pe, pt=pt, qn=qn, qe=qe, qt=qt, sc=sc.numpy(), dc=dc.numpy(),
         ef=ef.numpy(), tf=tf.numpy(), s_emb=s_emb.detach().numpy(), d_emb=d_emb.detach().numpy(),
         keys=np.array(sorted(shapes)), **grads_compact({k_: p.grad for k_, p in model.named_parameters()}))


def gold_state_dict_keys():
    """Key/shape contract at the BASELINE dims."""
    src, dst, eid, t, num_rows = toy_graph(1)
    nf = np.zeros((num_rows, 172), np.float32)
    ef = np.zeros((len(eid) + 1, 172), np.float32)
    ns = get_neighbor_sampler(Data(src, dst, t, eid), "recent", seed=0)
    out = {}
    for tag, m in (("tgat", TGAT(nf, ef, ns, 100, 2, 2, 0.1)),
                   ("tgn", MemoryModel(nf, ef, ns, 100, "TGN", 1, 2, 0.1)),
                   ("dyg", DyGFormer(nf, ef, ns, 100, 50, 1, 2, 2, 0.1, 32))):
        sd = m.state_dict()
        out[tag + "_keys"] = np.array(list(sd.keys()))
        out[tag + "_shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
        out[tag + "_nparams"] = np.int64(sum(p.numel() for p in m.parameters() if p.requires_grad))
    out["num_rows"] = np.int64(num_rows)
    save("state_dict_keys", **out)


# ------------------------------------------------------------------------------------------------- full size (B = 600)
# BASELINE batch size on Wikipedia / Reddit-shape graphs of reduced edge count: the row counts at which the product takes its
# production dispatch (split-bf16 products, merged projections, row sharing).  The graph is rebuilt in the tests from
# flid_amd.synth with the recorded arguments (a checksum of the arrays is stored); weights and upstream gradients are seeds.
def _crc(*arrs):
    import zlib
    c = 0
    for a in arrs:
        c = zlib.crc32(np.ascontiguousarray(a).view(np.uint8), c)
    return np.int64(c)


def _rows(x, step):
    return x.detach().numpy()[::step].copy()


def run_tgat_b600(tag, seed, lo, zero_node_feat, bias_te, num_edges=30000, scale=0.05, kink_free=False):
    from flid_amd.synth import wikipedia_like
    data = wikipedia_like(num_edges=num_edges, seed=0, zero_node_feat=zero_node_feat)
    ns = get_neighbor_sampler(Data(data.src_node_ids, data.dst_node_ids, data.node_interact_times, data.edge_ids), "recent", seed=0)
    model = TGAT(data.node_raw_features, data.edge_raw_features, ns, time_feat_dim=100, num_layers=2, num_heads=2, dropout=0.0)
    shapes = {k_: tuple(v.shape) for k_, v in model.state_dict().items()}
    params = O.seeded_like(shapes, seed=seed, scale=scale)
    if not bias_te:
        params["time_encoder.w.bias"].zero_()
    if kink_free:
        O.kink_free_(params)
    model.load_state_dict(params)
    model.train()
    sl = slice(lo, lo + 600)
    bs, bd, bt = data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl]
    s_emb, d_emb = model.compute_src_dst_node_temporal_embeddings(bs, bd, bt, num_neighbors=20)
    r = np.random.RandomState(seed + 1000).standard_normal((2, 600, 172)).astype(np.float32)
    (s_emb * torch.from_numpy(r[0])).sum().add((d_emb * torch.from_numpy(r[1])).sum()).backward()
    save(tag, num_edges=np.int64(num_edges), zero_node_feat=np.bool_(zero_node_feat), lo=np.int64(lo), seed=np.int64(seed),
         scale=np.float64(scale), bias_te=np.bool_(bias_te), r_seed=np.int64(seed + 1000), kink_free=np.bool_(kink_free),
         crc=_crc(data.src_node_ids, data.dst_node_ids, data.node_interact_times, data.edge_raw_features[:64]),
         s_emb=s_emb.detach().numpy(), d_emb=d_emb.detach().numpy(),
         **grads_compact({k_: p.grad for k_, p in model.named_parameters()}))


def gold_tgn_b600(num_edges=24000, warm=30, rec=3, seed=61, scale=0.05, step=4):
    from flid_amd.synth import reddit_like
    data = reddit_like(num_edges=num_edges, seed=3)
    src, dst, t, eid = data.src_node_ids, data.dst_node_ids, data.node_interact_times, data.edge_ids
    ns = get_neighbor_sampler(Data(src, dst, t, eid), "recent", seed=0)
    model = MemoryModel(data.node_raw_features, data.edge_raw_features, ns, time_feat_dim=100, model_name="TGN", num_layers=1,
                        num_heads=2, dropout=0.0)
    params = O.seeded_like(O.tgn_shapes(172, 172, 100, 1), seed=seed, scale=scale)
    params["time_encoder.w.bias"].zero_()
    O.kink_free_(params)
    sd = dict(params)
    sd["embedding_module.time_encoder.w.weight"] = params["time_encoder.w.weight"]
    sd["embedding_module.time_encoder.w.bias"] = params["time_encoder.w.bias"]
    missing = model.load_state_dict(sd, strict=False)
    assert all("memory_bank" in m for m in missing.missing_keys), missing
    model.train()
    B = 600
    out = dict(num_edges=np.int64(num_edges), warm=np.int64(warm), rec=np.int64(rec), seed=np.int64(seed), scale=np.float64(scale),
               step=np.int64(step), crc=_crc(src, dst, t, data.edge_raw_features[:64]))
    model.memory_bank.__init_memory_bank__()
    rs = np.random.RandomState(seed + 1)
    first_item, n_items = int(dst.min()), int(dst.max() - dst.min() + 1)
    for b in range(warm):                       # state warm-up: positives only, no autograd (negatives never change the state)
        sl = slice(b * B, (b + 1) * B)
        with torch.no_grad():
            model.compute_src_dst_node_temporal_embeddings(src[sl], dst[sl], t[sl], eid[sl], True, 20)
    for j in range(rec):
        b = warm + j
        sl = slice(b * B, (b + 1) * B)
        bs, bd, bt, be = src[sl], dst[sl], t[sl], eid[sl]
        neg = rs.randint(first_item, first_item + n_items, size=B).astype(np.int64)
        out[f"neg{j}"] = neg
        model.zero_grad()
        ns_emb, nd_emb = model.compute_src_dst_node_temporal_embeddings(bs, neg, bt, None, False, 20)
        ps_emb, pd_emb = model.compute_src_dst_node_temporal_embeddings(bs, bd, bt, be, True, 20)
        r = np.random.RandomState(seed + 100 + j).standard_normal((4, B, 172)).astype(np.float32)
        sum((e * torch.from_numpy(r[i])).sum() for i, e in enumerate((ns_emb, nd_emb, ps_emb, pd_emb))).backward()
        g = grads_compact({k_: p.grad for k_, p in model.named_parameters() if p.grad is not None})
        out.update({f"b{j}:" + k_: v for k_, v in g.items()})
        model.memory_bank.detach_memory_bank()
        for key, e in (("ns", ns_emb), ("nd", nd_emb), ("ps", ps_emb), ("pd", pd_emb)):
            out[f"{key}{j}"] = _rows(e, step)
        mem = model.memory_bank.node_memories.detach()
        touched = np.unique(np.concatenate([bs, bd]))[::3]
        out[f"touched{j}"] = touched
        out[f"mem{j}"] = mem.numpy()[touched].copy()
        out[f"memsum{j}"] = np.array([mem.double().sum().item(), (mem.double() ** 2).sum().item()])
        out[f"lu{j}"] = model.memory_bank.node_last_updated_times.detach().numpy().copy()
        has = np.zeros(mem.shape[0], dtype=bool)
        msum = np.zeros(mem.shape[0], dtype=np.float64)
        pt = np.zeros(mem.shape[0], dtype=np.float64)
        for nid, lst in model.memory_bank.node_raw_messages.items():
            if len(lst):
                has[nid], msum[nid], pt[nid] = True, lst[-1][0].detach().double().sum().item(), lst[-1][1]
        out[f"has{j}"], out[f"pmsum{j}"], out[f"pt{j}"] = has, msum, pt
    save("tgn_B600x3", **out)


def gold_dyg_b600(num_edges=24000, lo=20000, seed=71, scale=0.04):
    from flid_amd.synth import reddit_like
    data = reddit_like(num_edges=num_edges, seed=4)
    src, dst, t, eid = data.src_node_ids, data.dst_node_ids, data.node_interact_times, data.edge_ids
    ns = get_neighbor_sampler(Data(src, dst, t, eid), "recent", seed=0)
    model = DyGFormer(data.node_raw_features, data.edge_raw_features, ns, time_feat_dim=100, channel_embedding_dim=50, patch_size=1,
                      num_layers=2, num_heads=2, dropout=0.0, max_input_sequence_length=32)
    shapes = {k_: tuple(v.shape) for k_, v in model.state_dict().items()}
    assert shapes == O.dyg_shapes(172, 172, 100, 50, 1, 2)
    params = O.seeded_like(shapes, seed=seed, scale=scale)
    params["time_encoder.w.bias"].zero_()
    model.load_state_dict(params)
    model.train()
    sl = slice(lo, lo + 600)
    s_emb, d_emb = model.compute_src_dst_node_temporal_embeddings(src[sl], dst[sl], t[sl])
    r = np.random.RandomState(seed + 1000).standard_normal((2, 600, 172)).astype(np.float32)
    (s_emb * torch.from_numpy(r[0])).sum().add((d_emb * torch.from_numpy(r[1])).sum()).backward()
    save("dyg_B600", num_edges=np.int64(num_edges), lo=np.int64(lo), seed=np.int64(seed), scale=np.float64(scale),
         r_seed=np.int64(seed + 1000), crc=_crc(src, dst, t, data.edge_raw_features[:64]),
         s_emb=s_emb.detach().numpy(), d_emb=d_emb.detach().numpy(),
         **grads_compact({k_: p.grad for k_, p in model.named_parameters()}))


def run_tcl(tag, dn, de, dt, layers, heads, k, batch, seed, graph_seed, scale=0.2, strategy="recent"):
    """TCL (models/TCL.py) on the toy stream: embeddings + parameter gradients; batch rows with no history / a short history, and one
    edge whose two endpoints are the same node (identical sequences on both sides of the cross-attention)"""
    src, dst, eid, t, num_rows = toy_graph(graph_seed)
    rs = np.random.RandomState(seed)
    node_feat = rs.standard_normal((num_rows, dn)).astype(np.float32)
    edge_feat = rs.standard_normal((len(eid) + 1, de)).astype(np.float32)
    node_feat[0] = 0
    edge_feat[0] = 0
    ns = get_neighbor_sampler(Data(src, dst, t, eid), strategy, seed=3)
    model = TCL(node_feat, edge_feat, ns, time_feat_dim=dt, num_layers=layers, num_heads=heads, num_depths=k + 1, dropout=0.0)
    shapes = {k_: tuple(v.shape) for k_, v in model.state_dict().items()}
    assert shapes == O.tcl_shapes(dn, de, dt, layers, k + 1), "state_dict contract drifted"
    params = O.seeded_like(shapes, seed=seed, scale=scale)
    model.load_state_dict(params)
    model.train()
    pick = np.sort(rs.choice(len(eid), size=batch, replace=False))
    bs, bd, bt = src[pick].copy(), dst[pick].copy(), t[pick].copy()
    bs[0], bd[0], bt[0] = src[0], dst[0], t[0]              # no history on either side: every key but the node itself is masked
    bd[1] = bs[1]
    s_emb, d_emb = model.compute_src_dst_node_temporal_embeddings(bs, bd, bt, num_neighbors=k)
    r = rs.standard_normal((2, batch, dn)).astype(np.float32)
    (s_emb * torch.from_numpy(r[0])).sum().add((d_emb * torch.from_numpy(r[1])).sum()).backward()
    save(tag, src=src, dst=dst, eid=eid, t=t, num_rows=np.int64(num_rows), node_feat=node_feat, edge_feat=edge_feat,
         dims=np.array([dn, de, dt, layers, heads, k]), seed=np.int64(seed), scale=np.float64(scale), strategy=np.array(strategy),
         bs=bs, bd=bd, bt=bt, r=r, s_emb=s_emb.detach().numpy(), d_emb=d_emb.detach().numpy(), keys=np.array(sorted(shapes)),
         **grads_compact({k_: p.grad for k_, p in model.named_parameters()}))


def run_mixer(tag, dn, dt, layers, k, gap, batch, seed, graph_seed, scale=0.2, strategy="recent"):
    """GraphMixer (models/GraphMixer.py) on the toy stream; `gap` smaller than some histories and larger than others"""
    src, dst, eid, t, num_rows = toy_graph(graph_seed)
    rs = np.random.RandomState(seed)
    node_feat = rs.standard_normal((num_rows, dn)).astype(np.float32)
    node_feat[0] = rs.standard_normal(dn).astype(np.float32) * 0.5       # a NON-zero padding row: it shows in roots without neighbors
    edge_feat = np.zeros((len(eid) + 1, 4), dtype=np.float32)
    ns = get_neighbor_sampler(Data(src, dst, t, eid), strategy, seed=3)
    model = GraphMixer(node_feat, edge_feat, ns, time_feat_dim=dt, num_tokens=k, num_layers=layers, dropout=0.0)
    shapes = {k_: tuple(v.shape) for k_, v in model.state_dict().items()}
    assert shapes == O.mixer_shapes(dn, dt, k, layers), "state_dict contract drifted"
    params = O.seeded_like(shapes, seed=seed, scale=scale)
    model.load_state_dict(params)
    model.train()
    pick = np.sort(rs.choice(len(eid), size=batch, replace=False))
    bs, bd, bt = src[pick].copy(), dst[pick].copy(), t[pick].copy()
    bs[0], bd[0], bt[0] = src[0], dst[0], t[0]
    s_emb, d_emb = model.compute_src_dst_node_temporal_embeddings(bs, bd, bt, num_neighbors=k, time_gap=gap)
    r = rs.standard_normal((2, batch, dn)).astype(np.float32)
    (s_emb * torch.from_numpy(r[0])).sum().add((d_emb * torch.from_numpy(r[1])).sum()).backward()
    grads = {k_: p.grad for k_, p in model.named_parameters() if p.grad is not None}
    save(tag, src=src, dst=dst, eid=eid, t=t, num_rows=np.int64(num_rows), node_feat=node_feat,
         dims=np.array([dn, dt, layers, k, gap]), seed=np.int64(seed), scale=np.float64(scale), strategy=np.array(strategy),
         bs=bs, bd=bd, bt=bt, r=r, s_emb=s_emb.detach().numpy(), d_emb=d_emb.detach().numpy(), keys=np.array(sorted(shapes)),
         **grads_compact(grads))


def gold_tcl_full(num_edges=24000, lo=20000, batch=200, seed=81, scale=0.05):
    """TCL at the BASELINE dims (172 / 172 / 100, K = 20, 2 layers, 2 heads), 200 edges of the Wikipedia-shape stream"""
    from flid_amd.synth import wikipedia_like
    data = wikipedia_like(num_edges=num_edges, seed=0, zero_node_feat=False)
    src, dst, t, eid = data.src_node_ids, data.dst_node_ids, data.node_interact_times, data.edge_ids
    ns = get_neighbor_sampler(Data(src, dst, t, eid), "recent", seed=0)
    model = TCL(data.node_raw_features, data.edge_raw_features, ns, time_feat_dim=100, num_layers=2, num_heads=2, num_depths=21, dropout=0.0)
    shapes = {k_: tuple(v.shape) for k_, v in model.state_dict().items()}
    assert shapes == O.tcl_shapes(172, 172, 100, 2, 21)
    params = O.seeded_like(shapes, seed=seed, scale=scale)
    model.load_state_dict(params)
    model.train()
    sl = slice(lo, lo + batch)
    s_emb, d_emb = model.compute_src_dst_node_temporal_embeddings(src[sl], dst[sl], t[sl], num_neighbors=20)
    r = np.random.RandomState(seed + 1000).standard_normal((2, batch, 172)).astype(np.float32)
    (s_emb * torch.from_numpy(r[0])).sum().add((d_emb * torch.from_numpy(r[1])).sum()).backward()
    save("tcl_full", num_edges=np.int64(num_edges), lo=np.int64(lo), batch=np.int64(batch), seed=np.int64(seed), scale=np.float64(scale),
         r_seed=np.int64(seed + 1000), crc=_crc(src, dst, t, data.edge_raw_features[:64]),
         s_emb=s_emb.detach().numpy(), d_emb=d_emb.detach().numpy(),
         **grads_compact({k_: p.grad for k_, p in model.named_parameters()}))


def gold_mixer_full(num_edges=60000, lo=56000, batch=200, seed=91, scale=0.05):
    """GraphMixer at the BASELINE dims (172 / 100, K = 20 tokens, time_gap = 2000: popular items have longer histories than that)"""
    from flid_amd.synth import wikipedia_like
    data = wikipedia_like(num_edges=num_edges, seed=0, zero_node_feat=False)
    src, dst, t, eid = data.src_node_ids, data.dst_node_ids, data.node_interact_times, data.edge_ids
    ns = get_neighbor_sampler(Data(src, dst, t, eid), "recent", seed=0)
    model = GraphMixer(data.node_raw_features, data.edge_raw_features, ns, time_feat_dim=100, num_tokens=20, num_layers=2, dropout=0.0)
    shapes = {k_: tuple(v.shape) for k_, v in model.state_dict().items()}
    assert shapes == O.mixer_shapes(172, 100, 20, 2)
    params = O.seeded_like(shapes, seed=seed, scale=scale)
    model.load_state_dict(params)
    model.train()
    sl = slice(lo, lo + batch)
    longest = max(int(((src[:lo] == v) | (dst[:lo] == v)).sum()) for v in np.unique(dst[sl]))
    assert longest > 2000, longest
    s_emb, d_emb = model.compute_src_dst_node_temporal_embeddings(src[sl], dst[sl], t[sl], num_neighbors=20, time_gap=2000)
    r = np.random.RandomState(seed + 1000).standard_normal((2, batch, 172)).astype(np.float32)
    (s_emb * torch.from_numpy(r[0])).sum().add((d_emb * torch.from_numpy(r[1])).sum()).backward()
    save("mixer_full", num_edges=np.int64(num_edges), lo=np.int64(lo), batch=np.int64(batch), seed=np.int64(seed), scale=np.float64(scale),
         r_seed=np.int64(seed + 1000), crc=_crc(src, dst, t, data.edge_raw_features[:64]),
         s_emb=s_emb.detach().numpy(), d_emb=d_emb.detach().numpy(),
         **grads_compact({k_: p.grad for k_, p in model.named_parameters() if p.grad is not None}))


def gold_time_shifts():
    """compute_src_dst_node_time_shifts (models/MemoryModel.py:718-751) on two toy streams (integer and fractional stamps, ties)"""
    out = {}
    for tag, gs in (("a", 4), ("b", 7)):
        src, dst, eid, t, _ = toy_graph(gs)
        out[f"src_{tag}"], out[f"dst_{tag}"], out[f"t_{tag}"] = src, dst, t
        out[f"shifts_{tag}"] = np.array(compute_src_dst_node_time_shifts(src, dst, t), dtype=np.float64)
    save("time_shifts", **out)


def gold_backbones_small():
    run_tcl("tcl_K5", 8, 6, 4, 2, 2, 5, 9, seed=83, graph_seed=8)
    run_tcl("tcl_K3_uniform", 8, 6, 4, 1, 2, 3, 7, seed=84, graph_seed=8, strategy="uniform")
    run_mixer("mixer_K6", 8, 4, 2, 6, 7, 9, seed=93, graph_seed=9)
    run_mixer("mixer_K4_uniform", 8, 4, 1, 4, 5, 7, seed=94, graph_seed=9, strategy="uniform")


def gold_tgat_lp3(num_edges=30000, lo=20000, batch=200, steps=3, seed=48, scale=0.05, lr=1e-4, neg_seed=7):
    """The trainer's call sequence on TGAT (PTCL/EM_warmup.py:126-231), run on the reference classes: `steps` consecutive batches of
    the link-prediction warm-up -- seeded NegativeEdgeSampler draw, positive and negative embeddings, MergeLayer link predictor,
    sigmoid, BCELoss, optimizer.zero_grad / backward / Adam step over nn.Sequential(backbone, head).  Recorded: the negative draws,
    every step's loss, and the parameters after the last step (compacted like the gradients of the other fixtures)."""
    from flid_amd.synth import wikipedia_like
    from models.modules import MergeLayer
    from utils.utils import create_optimizer
    import torch.nn as nn
    data = wikipedia_like(num_edges=num_edges, seed=0, zero_node_feat=False)
    ns = get_neighbor_sampler(Data(data.src_node_ids, data.dst_node_ids, data.node_interact_times, data.edge_ids), "recent", seed=0)
    backbone = TGAT(data.node_raw_features, data.edge_raw_features, ns, time_feat_dim=100, num_layers=2, num_heads=2, dropout=0.0)
    head = MergeLayer(input_dim1=172, input_dim2=172, hidden_dim=172, output_dim=1)
    shapes = {k_: tuple(v.shape) for k_, v in backbone.state_dict().items()}
    params = O.seeded_like(shapes, seed=seed, scale=scale)
    O.kink_free_(params)
    backbone.load_state_dict(params)
    hshapes = {k_: tuple(v.shape) for k_, v in head.state_dict().items()}
    hparams = O.seeded_like(hshapes, seed=seed + 1, scale=scale)
    head.load_state_dict(hparams)
    model = nn.Sequential(backbone, head)
    model.train()
    optimizer = create_optimizer(model=model, optimizer_name="Adam", learning_rate=lr, weight_decay=0.0)      # EM_warmup.py:99-100
    neg = NegativeEdgeSampler(src_node_ids=data.src_node_ids[:lo + batch * steps], dst_node_ids=data.dst_node_ids[:lo + batch * steps], seed=neg_seed)
    loss_func = nn.BCELoss()
    losses, negs = [], []
    model[0].set_neighbor_sampler(ns)
    for b in range(steps):
        sl = slice(lo + b * batch, lo + (b + 1) * batch)
        bs, bd, bt = data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl]
        _, bn = neg.sample(size=len(bs))                                                             # :132
        se, de = model[0].compute_src_dst_node_temporal_embeddings(src_node_ids=bs, dst_node_ids=bd, node_interact_times=bt, num_neighbors=20)
        nse, nde = model[0].compute_src_dst_node_temporal_embeddings(src_node_ids=bs, dst_node_ids=bn, node_interact_times=bt, num_neighbors=20)
        pos = model[1](input_1=se, input_2=de).squeeze(dim=-1).sigmoid()                              # :212-215
        ngp = model[1](input_1=nse, input_2=nde).squeeze(dim=-1).sigmoid()
        predicts = torch.cat([pos, ngp], dim=0)
        labels = torch.cat([torch.ones_like(pos), torch.zeros_like(ngp)], dim=0)
        loss = loss_func(input=predicts, target=labels)                                              # :222
        losses.append(float(loss.item()))
        negs.append(np.asarray(bn, dtype=np.int64))
        optimizer.zero_grad()
        loss.backward()
        if b == 0:      # the first step's gradients: which entries carry a gradient above rounding level (Adam moves ALL by ~lr)
            first = {"g1:" + k_[2:]: v for k_, v in grads_compact({k_: p.grad for k_, p in model[0].named_parameters()}).items() if k_.startswith("g:")}
        optimizer.step()                                                                             # :229-231
    final = {"p:" + k_[2:]: v for k_, v in grads_compact({k_: p for k_, p in model[0].named_parameters()}).items() if k_.startswith("g:")}
    final.update({"ps:" + k_[3:]: v for k_, v in grads_compact({k_: p for k_, p in model[0].named_parameters()}).items() if k_.startswith("gs:")})
    hfinal = {"h:" + k_: p.detach().numpy().copy() for k_, p in model[1].named_parameters()}
    save("tgat_lp3", num_edges=np.int64(num_edges), lo=np.int64(lo), batch=np.int64(batch), steps=np.int64(steps), seed=np.int64(seed),
         scale=np.float64(scale), lr=np.float64(lr), neg_seed=np.int64(neg_seed), losses=np.asarray(losses, dtype=np.float64),
         neg=np.stack(negs), crc=_crc(data.src_node_ids, data.dst_node_ids, data.node_interact_times, data.edge_raw_features[:64]),
         **final, **hfinal, **first)


FULL = {
    "tgat_lp3": gold_tgat_lp3,
    "time_shifts": gold_time_shifts,
    "backbones_small": gold_backbones_small,
    "tcl_full": gold_tcl_full,
    "mixer_full": gold_mixer_full,
    "neg_sampler": gold_neg_sampler,
    # the full-size cases sit where bench.py's batches sit: on the WHOLE Wikipedia-shape stream (157 474 edges) at edge 100 000, and 200 000
    # edges into the Reddit-shape stream -- histories are full there (20 real neighbors almost everywhere, few padded slots, many repeated
    # (node, time) rows), unlike the 20-30 k-edge prefixes these fixtures were first generated on
    "tgat_B600_full": lambda: run_tgat_b600("tgat_B600_full", seed=46, lo=100000, zero_node_feat=True, bias_te=False, num_edges=157474),
    # non-zero node features, trained-like time-encoder bias, ReLU units away from their kink: gradients comparable at 1e-4 max|g|
    "tgat_B600_kinkfree": lambda: run_tgat_b600("tgat_B600_kinkfree", seed=47, lo=105000, zero_node_feat=False, bias_te=True,
                                                kink_free=True, num_edges=157474),
    "tgn_B600x3": lambda: gold_tgn_b600(num_edges=200000, warm=330),
    "dyg_B600": lambda: gold_dyg_b600(num_edges=200000, lo=190000),
}


if __name__ == "__main__":
    only = sys.argv[1:]
    if only:                                   # regenerate selected full-size fixtures only
        for name in only:
            FULL[name]()
        sys.exit(0)
    gold_sampler()
    gold_time_encoder()
    gold_attention()
    run_tgat("tgat_L1_K2", 8, 8, 4, 1, 2, 6, seed=41, graph_seed=4)
    run_tgat("tgat_L2_K2", 8, 8, 4, 2, 2, 6, seed=42, graph_seed=4)
    run_tgat("tgat_L2_K20", 8, 6, 4, 2, 20, 9, seed=43, graph_seed=5)
    # full BASELINE dims: weight scale ~ PyTorch default init (1/sqrt(fan_in)) so that |emb| = O(1), where "within 1e-4" is meant
    run_tgat("tgat_L2_K20_full", 172, 172, 100, 2, 20, 8, seed=44, graph_seed=6, bias_te=False, scale=0.05)
    run_tgat("tgat_L1_K20_full_bias", 172, 172, 100, 1, 20, 8, seed=45, graph_seed=6, bias_te=True, scale=0.05)
    gold_tgn()
    run_dyg("dyg_p1", 1, 8, seed=51, graph_seed=7)
    run_dyg("dyg_p2", 2, 9, seed=52, graph_seed=7)
    gold_state_dict_keys()
    for fn in FULL.values():
        fn()
