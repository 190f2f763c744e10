"""Device engine for the recursive temporal-attention embedding (TGAT, and TGN's GraphAttentionEmbedding).

replaces: models/TGAT.py:68-144 compute_node_temporal_embeddings and models/MemoryModel.py:632-715, including the
sampler calls (utils/utils.py:149-214) and every gather / cat / Linear / softmax / LayerNorm under them.

Frontier layout.  With n roots and k neighbors, all nodes touched by an L-layer embedding are kept in ONE row space:
rows [0, n) are the roots, and the j-th sampled neighbor of row r is row  n + r*k + j.  So
  * the sampler output arrays S_*[r, j] (r < R_1) are at the same time the ids / query times of the deeper rows,
  * layer l computes H^l for the first R_l = n (1 + k + ... + k^(L-l)) rows from H^(l-1) of the first R_(l-1) rows,
  * the neighbor features of layer l >= 2 are rows of the previous layer's output (feat_idx = n + r*k + j) and, for
    layer 1, rows of the node table itself (feat_idx = neighbor id) -- the deepest frontier is never materialised.
Every FLOP and gathered byte runs in libflid_tg.so (ops.py); torch supplies memory, the stream and autograd plumbing.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import ops
from .graph import TemporalGraph

LAYER_PARAMS = ("query_projection.weight", "key_projection.weight", "value_projection.weight", "layer_norm.weight",
                "layer_norm.bias", "residual_fc.weight", "residual_fc.bias")
MERGE_PARAMS = ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")


@dataclass
class _LayerCtx:
    R: int
    own: torch.Tensor = None
    raw: torch.Tensor = None
    q: torch.Tensor = None
    u: torch.Tensor = None
    attn: ops.AttnArgs = None
    agg: torch.Tensor = None
    prob: torch.Tensor = None
    ctxv: torch.Tensor = None
    res: torch.Tensor = None
    drop: Optional[torch.Tensor] = None
    x: torch.Tensor = None
    mean: torch.Tensor = None
    rstd: torch.Tensor = None
    y: torch.Tensor = None
    f1: torch.Tensor = None


DEDUPE = True     # share repeated (node, time) rows inside a call; False reproduces the reference's row-for-row recursion


@dataclass
class _Ctx:
    n: int
    k: int
    fr: "Frontier" = None
    ids_all: torch.Tensor = None
    cosb: torch.Tensor = None
    layers: List[_LayerCtx] = field(default_factory=list)
    table: torch.Tensor = None
    table_grad: bool = False


def frontier_rows(n: int, k: int, depth: int) -> int:
    """n (1 + k + ... + k^depth)"""
    return n * sum(k ** d for d in range(depth + 1))


@dataclass
class Frontier:
    """Rows touched by one L-layer embedding call, level by level, with repeated (node, time) pairs stored ONCE.

    level 0 = the n roots; level d+1 = the distinct (neighbor id, float32 neighbor time) pairs sampled from level d.
    The embedding of a node at a time is a pure function of that pair, so rows that recur inside a batch (the same user a
    few edges later shares 19 of its 20 most recent neighbors; every padded slot is the pair (0, 0.0)) are computed once
    and shared: `child[r*k + j]` is the row that holds slot j of row r.  Exact in eval mode and with dropout 0.  In train mode
    with dropout > 0 a shared row also shares ONE dropout mask, where the reference draws an independent mask for every
    occurrence: same expectation, slightly more correlated noise (engine.DEDUPE = False restores the reference's behaviour)."""
    counts: List[int]                 # rows per level, len L+... (levels 0..L-1 are sampled; level L is never materialised)
    ids_all: torch.Tensor             # int32 (sum(counts),)   node id of every row
    S: tuple                          # (nbr i32, eid i32, t f32, dt f32) each (R_1, k): slot lists of levels 0..L-1
    child: Optional[torch.Tensor]     # int32 (R_2 * k,) row index of each slot's node, for levels 0..L-2 (None if L == 1)
    pad_rows: List[int] = field(default_factory=list)   # per level d+1: the shared row of the padding pair (0, 0.0), or -1
    # layer -> its own slot lists (random sampling strategies: the reference draws a FRESH sample for the same root at every layer
    # of its recursion, models/TGAT.py:94,104, so the layers cannot share the level-0 lists); None = S serves every layer
    S_layers: Optional[dict] = None
    # compact base table (TGN's lazily updated memory rows): row of the table that holds each deepest-level slot's node / the
    # padding node, when the table is NOT indexed by node id
    feat_idx0: Optional[torch.Tensor] = None
    pad_row0: int = 0

    def rows(self, upto_level: int) -> int:
        return sum(self.counts[:upto_level + 1])

    def S_at(self, layer: int) -> tuple:
        return self.S if self.S_layers is None else self.S_layers[layer]


def sample_frontier(graph: TemporalGraph, ids_dev: torch.Tensor, times_dev: torch.Tensor, k: int, num_layers: int,
                    dedupe: bool = True) -> Frontier:
    """All sampler lookups of one embedding call: level 0 with float64 query times, deeper levels with the float32
    neighbor times fed straight back (models/TGAT.py:110-111)."""
    dev = ids_dev.device
    counts, ids_parts, S_parts, child_parts, pad_rows = [ids_dev.numel()], [ids_dev], [], [], []
    q_ids, q_t, off = ids_dev, times_dev, 0
    for d in range(num_layers):
        S = graph.sample_recent(q_ids, q_t, k)
        S_parts.append(S)
        if d == num_layers - 1:
            break
        nbr_flat, t_flat = S[0].reshape(-1), S[2].reshape(-1)
        nxt = off + counts[d]
        if dedupe:
            q_ids, q_t, rows, pad = graph.dedupe_pairs(nbr_flat, t_flat, nxt)            # hash set on the device (tg_dedupe_pairs)
            child_parts.append(rows)
            pad_rows.append(pad)
        else:
            pad_rows.append(-1)
            q_ids, q_t = nbr_flat, t_flat
            child_parts.append(torch.arange(nxt, nxt + nbr_flat.numel(), dtype=torch.int32, device=dev))
        ids_parts.append(q_ids)
        counts.append(q_ids.numel())
        off = nxt
    S_all = tuple(torch.cat([p[i] for p in S_parts]) if len(S_parts) > 1 else S_parts[0][i] for i in range(4))
    return Frontier(counts=counts, ids_all=torch.cat(ids_parts) if len(ids_parts) > 1 else ids_parts[0], S=S_all,
                    child=torch.cat(child_parts) if child_parts else None, pad_rows=pad_rows)


def frontier_from_host_sampler(sampler, ids: np.ndarray, times: np.ndarray, k: int, num_layers: int, dev, groups=None) -> Frontier:
    """Frontier of the `uniform` / `time_interval_aware` strategies: the neighbor lists come from the sampler mirror's numpy path,
    which consumes numpy's RandomState stream exactly as the reference does (utils/utils.py:176-199), in the reference's CALL ORDER:
    compute_node_temporal_embeddings(ids, L) first recurses for the node's own lower-layer embedding (which samples at the lower
    layers), THEN samples at its own layer, then recurses for the sampled neighbors (models/TGAT.py:94,104,110) -- and
    compute_src_dst... does all of that for the sources before the destinations (`groups` = row ranges processed one after the
    other, default one group).  Rows are not shared (an embedding is no longer a function of (node, time) alone)."""
    if num_layers > 2:
        raise NotImplementedError("random sampling strategies: the device engine lays out at most 2 layers of independent samples")
    ids = np.asarray(ids, dtype=np.int64)
    times = np.asarray(times)
    n = len(ids)
    groups = groups or [(0, n)]

    def draw(node_ids, t):
        nb, ne, nt = sampler.get_historical_neighbors(node_ids, t, k)
        dt = (np.asarray(t)[:, None] - nt).astype(np.float32)               # TGAT.py:120 (float64 - float32, or float32 - float32)
        return nb.astype(np.int32), ne.astype(np.int32), nt.astype(np.float32), dt

    def stack(parts):
        return tuple(np.concatenate([p[i] for p in parts]) for i in range(4))

    if num_layers == 1:
        host = {1: stack([draw(ids[a:b], times[a:b]) for a, b in groups])}
        counts, ids_all, child = [n], ids.astype(np.int32), None
    else:
        own, top, below = [], [], []
        for a, b in groups:
            own.append(draw(ids[a:b], times[a:b]))                              # layer-1 sample of the roots (the `own` recursion)
            t2 = draw(ids[a:b], times[a:b])                                     # layer-2 sample of the same roots: a fresh draw
            top.append(t2)
            below.append(draw(t2[0].reshape(-1).astype(np.int64), t2[2].reshape(-1)))   # layer-1 sample of the layer-2 neighbors
        top_s, own_s, below_s = stack(top), stack(own), stack(below)
        host = {2: top_s, 1: tuple(np.concatenate([own_s[i], below_s[i]]) for i in range(4))}
        counts = [n, n * k]
        ids_all = np.concatenate([ids.astype(np.int32), top_s[0].reshape(-1)])
        child = np.arange(n, n + n * k, dtype=np.int32)
    flat = [ids_all] + ([child] if child is not None else []) + [a for l in sorted(host) for a in host[l]]
    devs = ops.h2d(flat, dev)
    ids_d = devs[0]
    child_d = devs[1] if child is not None else None
    rest = devs[2:] if child is not None else devs[1:]
    S_layers = {l: tuple(rest[4 * i:4 * i + 4]) for i, l in enumerate(sorted(host))}
    return Frontier(counts=counts, ids_all=ids_d, S=S_layers[num_layers], child=child_d, pad_rows=[-1] * (num_layers - 1), S_layers=S_layers)


def frontier_from_device_random(sampler, ids_dev: torch.Tensor, times_dev: torch.Tensor, k: int, num_layers: int) -> Frontier:
    """as frontier_from_host_sampler for a sampler in device_random mode: the same layout (independent samples per layer, rows not
    shared), every draw on the device (tg_sample_random) -- no per-node host loop, not numpy's stream"""
    if num_layers > 2:
        raise NotImplementedError("random sampling strategies: the device engine lays out at most 2 layers of independent samples")
    n = ids_dev.numel()
    draw = lambda i, t: sampler.sample_on_device(i.contiguous(), t.contiguous(), k)
    if num_layers == 1:
        S1 = draw(ids_dev, times_dev)
        return Frontier(counts=[n], ids_all=ids_dev, S=S1, child=None, pad_rows=[], S_layers={1: S1})
    own = draw(ids_dev, times_dev)                                   # layer-1 sample of the roots (their own lower-layer embedding)
    top = draw(ids_dev, times_dev)                                   # layer-2 sample of the same roots: a fresh draw
    below = draw(top[0].reshape(-1), top[2].reshape(-1))            # layer-1 sample of the layer-2 neighbors (float32 times)
    S1 = tuple(torch.cat([own[i], below[i]]) for i in range(4))
    ids_all = torch.cat([ids_dev, top[0].reshape(-1)])
    child = torch.arange(n, n + n * k, dtype=torch.int32, device=ids_dev.device)
    return Frontier(counts=[n, n * k], ids_all=ids_all, S=top, child=child, pad_rows=[-1], S_layers={2: top, 1: S1})


class _EmbedFn(torch.autograd.Function):
    """One autograd node for the whole L-layer embedding.  Inputs after the fixed ones: te_w, te_b, then per layer the
    7 attention + 4 merge parameters, then (optionally) the layer-0 base table when it carries gradient (TGN)."""

    @staticmethod
    def forward(ctx, cfg, fr, table, te_w, te_b, *layer_params):
        n, k, L, H = cfg["n"], cfg["k"], cfg["num_layers"], cfg["num_heads"]
        edge = cfg["edge_table"]
        p_drop, training = cfg["dropout"], cfg["training"]
        dev = table.device
        Dn, T = table.shape[1], te_w.numel()
        Dq = Dn + T
        hd = Dq // H
        st = _Ctx(n=n, k=k)
        st.table = table
        st.table_grad = cfg["table_grad"]
        st.fr = fr
        ids_all = fr.ids_all
        st.ids_all = ids_all
        te_w_flat = te_w.reshape(-1)
        # time encoding of a zero interval: cos(b) (models/TGAT.py:84-85); one row, shared by every query
        cosb = ops.time_encode(torch.zeros(1, device=dev), te_w_flat, te_b).reshape(-1)
        st.cosb = cosb
        H_prev = None
        for l in range(1, L + 1):
            Wq, Wk, Wv, ln_g, ln_b, Wr, br, W1, b1, W2, b2 = layer_params[(l - 1) * 11:(l - 1) * 11 + 11]
            R = fr.rows(L - l)
            S_nbr, S_eid, S_t, S_dt = fr.S_at(l)
            lc = _LayerCtx(R=R)
            lc.raw = ops.gather_rows(table, ids_all[:R])
            lc.own = lc.raw if l == 1 else H_prev[:R]
            # q = [own | cos(b)] Wq^T : the constant half folds into a bias row
            qbias = torch.empty((1, Dq), device=dev)
            ops.gemm(cosb.view(1, T), Wq[:, Dn:], qbias, tb=True)
            lc.q = torch.empty((R, Dq), device=dev)
            ops.gemm(lc.own, Wq[:, :Dn], lc.q, tb=True, bias=qbias.view(-1))
            # u_h = Wk_h^T q_h : the query carried into key space (score = u . z)
            Dk = Wk.shape[1]
            lc.u = torch.empty((R, H, Dk), device=dev)
            ops.gemm_batched(lc.q[:, :hd], Wk[:hd], lc.u[:, 0, :], H, hd, hd * Dk, Dk)
            if l == 1:
                feat, feat_idx = table, (S_nbr[:R].reshape(-1) if fr.feat_idx0 is None else fr.feat_idx0[:R * k])
            else:
                feat, feat_idx = H_prev, fr.child[:R * k]
            seed = _next_seeds(1)[0] if (training and p_drop > 0) else 0
            lc.attn = ops.AttnArgs(feat, feat_idx, edge, S_eid[:R].reshape(-1), S_nbr[:R].reshape(-1), S_dt[:R].reshape(-1),
                                   te_w_flat, te_b, k, H, hd ** -0.5, p_drop if training else 0.0, seed)
            lc.agg, lc.prob = ops.attn_fwd(lc.attn, lc.u)
            lc.ctxv = torch.empty((R, Dq), device=dev)
            ops.gemm_batched(lc.agg[:, 0, :], Wv[:hd], lc.ctxv[:, :hd], H, Dk, hd * Dk, hd, tb=True)
            lc.res = torch.empty((R, Dq), device=dev)
            ops.gemm(lc.ctxv, Wr, lc.res, tb=True, bias=br)
            if training and p_drop > 0:                                     # modules.py:235
                lc.drop = (torch.rand_like(lc.res) >= p_drop).to(torch.float32) / (1.0 - p_drop)
                lc.res = lc.res * lc.drop
            lc.x = torch.cat([lc.own, cosb.view(1, T).expand(R, T)], dim=1)
            lc.y, lc.mean, lc.rstd = ops.add_layernorm_fwd(lc.res, lc.x, ln_g, ln_b)
            lc.f1 = torch.empty((R, W1.shape[0]), device=dev)
            ops.gemm(lc.y, W1[:, :Dq], lc.f1, tb=True, bias=b1)
            ops.gemm(lc.raw, W1[:, Dq:], lc.f1, tb=True, accumulate=True, relu=True)
            H_cur = torch.empty((R, W2.shape[0]), device=dev)
            ops.gemm(lc.f1, W2, H_cur, tb=True, bias=b2)
            st.layers.append(lc)
            H_prev = H_cur
        ctx.st = st
        ctx.cfg = cfg
        ctx.save_for_backward(te_w, te_b, *layer_params)
        return H_prev

    @staticmethod
    def backward(ctx, dH):
        st, cfg = ctx.st, ctx.cfg
        te_w, te_b, *layer_params = ctx.saved_tensors
        n, k, L, H = st.n, st.k, cfg["num_layers"], cfg["num_heads"]
        dev = dH.device
        Dn, T = st.table.shape[1], te_w.numel()
        Dq = Dn + T
        hd = Dq // H
        grads = [None] * len(layer_params)
        d_tew = torch.zeros(T, device=dev)
        d_teb = torch.zeros(T, device=dev)
        d_cosb = torch.zeros(T, device=dev)
        d_table = torch.zeros_like(st.table) if st.table_grad else None
        dH = dH.contiguous()
        for l in range(L, 0, -1):
            lc = st.layers[l - 1]
            R = lc.R
            Wq, Wk, Wv, ln_g, ln_b, Wr, br, W1, b1, W2, b2 = layer_params[(l - 1) * 11:(l - 1) * 11 + 11]
            Dk = Wk.shape[1]
            need_own = l >= 2 or st.table_grad
            dHl = dH[:R]
            # merge layer
            dW2 = torch.empty_like(W2)
            ops.gemm(dHl, lc.f1, dW2, ta=True)
            db2 = ops.colsum(dHl)
            df1 = torch.empty_like(lc.f1)
            ops.gemm(dHl, W2, df1)
            ops.relu_bwd_(df1, lc.f1)
            dW1 = torch.empty_like(W1)
            ops.gemm(df1, lc.y, dW1[:, :Dq], ta=True)
            ops.gemm(df1, lc.raw, dW1[:, Dq:], ta=True)
            db1 = ops.colsum(df1)
            dy = torch.empty((R, Dq), device=dev)
            ops.gemm(df1, W1[:, :Dq], dy)
            d_rawrows = None
            if st.table_grad:
                d_rawrows = torch.empty((R, Dn), device=dev)
                ops.gemm(df1, W1[:, Dq:], d_rawrows)
            # residual + layer norm
            dsum, dg, dbeta = ops.add_layernorm_bwd(lc.res, lc.x, dy, ln_g, lc.mean, lc.rstd)
            d_cosb += ops.colsum(dsum[:, Dn:])
            dres = dsum * lc.drop if lc.drop is not None else dsum
            dWr = torch.empty_like(Wr)
            ops.gemm(dres, lc.ctxv, dWr, ta=True)
            dbr = ops.colsum(dres)
            dctx = torch.empty((R, Dq), device=dev)
            ops.gemm(dres, Wr, dctx)
            # value path
            dagg = torch.empty((R, H, Dk), device=dev)
            dWv = torch.empty_like(Wv)
            ops.gemm_batched(dctx[:, :hd], Wv[:hd], dagg[:, 0, :], H, hd, hd * Dk, Dk)
            ops.gemm_batched(dctx[:, :hd], lc.agg[:, 0, :], dWv[:hd], H, hd, Dk, hd * Dk, ta=True)
            # fused attention backward: re-streams the neighbor rows once
            if l >= 2:
                R_prev = st.fr.rows(L - l + 1)
                dH_prev = torch.zeros((R_prev, Dn), device=dev)
                dfeat = dH_prev
                # padded slots of the rows of this layer all point at the first row of their child level (if shared)
                pad_row = st.fr.pad_rows[0] if (l == L and st.fr.pad_rows) else -1     # one child level only
            else:
                dH_prev = None
                dfeat = d_table
                pad_row = st.fr.pad_row0         # node table: the padding node is row 0
            du, dw_part, db_part = ops.attn_bwd(lc.attn, lc.u, lc.agg, lc.prob, dagg, dfeat, pad_row)
            d_tew += dw_part
            d_teb += db_part
            # key / query path
            dq = torch.empty((R, Dq), device=dev)
            dWk = torch.empty_like(Wk)
            ops.gemm_batched(du[:, 0, :], Wk[:hd], dq[:, :hd], H, Dk, hd * Dk, hd, tb=True)
            ops.gemm_batched(lc.q[:, :hd], du[:, 0, :], dWk[:hd], H, hd, Dk, hd * Dk, ta=True)
            dWq = torch.empty_like(Wq)
            ops.gemm(dq, lc.own, dWq[:, :Dn], ta=True)
            dq_sum = ops.colsum(dq)
            dWq[:, Dn:] = torch.outer(dq_sum, st.cosb)
            tmp = torch.empty((1, T), device=dev)
            ops.gemm(dq_sum.view(1, Dq), Wq[:, Dn:], tmp)
            d_cosb += tmp.view(-1)
            if need_own:
                d_own = torch.empty((R, Dn), device=dev)
                ops.gemm(dq, Wq[:, :Dn], d_own)
                d_own += dsum[:, :Dn]
                if l >= 2:
                    dH_prev[:R] += d_own
                    if st.table_grad:
                        ops.scatter_add_rows(d_rawrows, st.ids_all[:R], d_table)
                else:
                    d_own += d_rawrows
                    ops.scatter_add_rows(d_own, st.ids_all[:R], d_table)
            grads[(l - 1) * 11:(l - 1) * 11 + 11] = [dWq, dWk, dWv, dg, dbeta, dWr, dbr, dW1, db1, dW2, db2]
            dH = dH_prev
        # d cos(b) -> d b  (t = 0, so no weight gradient from the zero-interval encoding)
        sinb = torch.sin(te_b)
        d_teb -= sinb * d_cosb
        ctx.st = None
        return (None, None, d_table, d_tew.view_as(te_w), d_teb, *grads)


_SCRATCH = {}
NATIVE = True     # one native call per layer (tg_tgat_layer_fwd/bwd); False = the op-by-op Python composition below (same kernels)


def _r4(n):
    return (n + 3) // 4 * 4


_FWD_FIELDS = ("qbias", "q", "u", "agg", "prob", "ctx", "res", "y", "mean", "rstd", "f1", "wT")
_BWD_FIELDS = ("df1", "dy", "dsum", "dres", "dctx", "dagg", "du", "dq", "part")


class _NativeLayer:
    """buffers + C descriptors of one tg_tgat_layer_fwd/bwd call pair.  Host cost matters here (the step is ~150 launches and
    the Python around them was as long as the GPU work): the saved activations of a layer are ONE allocation addressed by
    offset, the backward scratch is one cached allocation, and only tensors that leave this class are torch views."""

    def __init__(self, attn: ops.AttnArgs, params, own, table, ids, cosb, p_res, seed_res, compute_cosb=False):
        """own = None: the layer's own rows ARE its raw rows (layer 1).  The raw rows are gathered (table[ids]) straight into the
        right part of the saved [y | raw] buffer: the merge layer's torch.cat (modules.py:66) never materialises."""
        from ._lib import LayerDesc, LayerParams, lib
        dev = table.device
        R, H, Dn, T, Dk = attn.m, attn.heads, attn.dn, attn.dt_dim, attn.dk
        Dq = Dn + T
        self.attn, self.R, self.dims = attn, R, (H, Dn, T, Dq, Dk)
        # sized for the row count rounded up to a multiple of 1024: the number of distinct rows changes every step, and a fresh
        # 160 MB request that no cached block fits costs the caching allocator a hipMalloc (4-6 ms stalls, seen in step traces)
        Rc = (R + 1023) // 1024 * 1024
        sizes = (Dq, Rc * Dq, Rc * H * Dk, Rc * H * Dk, Rc * H * attn.k, Rc * Dq, Rc * Dq, Rc * (Dq + Dn), Rc, Rc, Rc * Dn,
                 int(lib().tg_tgat_layer_wt_floats(Dn, Dq, Dk)))
        total = 0
        offs = []
        for n in sizes:
            offs.append(total)
            total += _r4(n)
        self.act = torch.empty(total, dtype=torch.float32, device=dev)          # saved for backward as a whole
        self.out = torch.empty((R, Dn), dtype=torch.float32, device=dev)
        yo = offs[_FWD_FIELDS.index("y")]
        yr = self.act[yo:yo + R * (Dq + Dn)].view(R, Dq + Dn)
        assert ids.dtype == torch.int32 and ids.is_contiguous() and table.dtype == torch.float32
        raw = yr[:, Dq:]                       # filled by the layer's prelude launch (gather_table / gather_idx below)
        if own is None:
            own = raw
        self.keep = (params, own, cosb, table, ids)
        base = self.act.data_ptr()
        d = LayerDesc()
        d.attn = attn.desc
        d.params = LayerParams(*[t.data_ptr() for t in params])
        d.own, d.own_ld, d.raw, d.raw_ld, d.cosb = own.data_ptr(), ops._rowmajor_ld(own, "own"), raw.data_ptr(), ops._rowmajor_ld(raw, "raw"), cosb.data_ptr()
        d.res_dropout_p, d.res_seed = float(p_res), int(seed_res)
        for name, o in zip(_FWD_FIELDS, offs):
            setattr(d, name, base + 4 * o)
        d.out = self.out.data_ptr()
        d.y_ld = Dq + Dn
        d.compute_cosb = int(compute_cosb)
        d.gather_table, d.gather_ld, d.gather_idx = table.data_ptr(), ops._rowmajor_ld(table, "table"), ids.data_ptr()
        self.desc = d

    def forward(self):
        import ctypes as C
        from ._lib import check, lib
        with ops._timed("layer_fwd", self.R):
            check(lib().tg_tgat_layer_fwd(C.byref(self.desc), ops._stream()), "tg_tgat_layer_fwd")
        return self.out

    @staticmethod
    def grad_layout(params):
        """[(offset, numel, shape)] of every parameter gradient inside a layer's block (16-byte aligned) and the block size"""
        lay, off = [], 0
        for p in params:
            lay.append((off, p.numel(), p.shape))
            off += _r4(p.numel())
        return lay, off

    def backward(self, dout, params, gblock, vec, d_cosb, d_tew, d_teb, dfeat, pad_row, d_own, d_own_accumulate, want_d_raw,
                 slot=0, defer_join=False, finish_time_bias=False):
        """gblock: this layer's zero-filled gradient block (layout = grad_layout(params)); vec: dq zero floats of scratch"""
        import ctypes as C
        from ._lib import LayerBwdDesc, LayerParams, check, lib
        H, Dn, T, Dq, Dk = self.dims
        R, dev = self.R, dout.device
        # scratch that lives only inside this call is kept across steps (grown on demand); saved activations are NOT cached:
        # a caller may run several forwards before one backward (positive + negative edges of the reference's trainers)
        sizes = (R * Dn, R * Dq, R * Dq, R * Dq if self.desc.res_dropout_p > 0 else 0, R * Dq, R * H * Dk, R * H * Dk, R * Dq,
                 int(lib().tg_tgat_layer_part_floats(R, Dn, Dq, T)) + 32)
        total, offs = 0, []
        for n in sizes:
            offs.append(total)
            total += _r4(n)
        # one scratch block per layer slot: with a deferred join the side streams still read layer l's block while layer l-1 runs
        buf = _SCRATCH.get((dev, slot))
        if buf is None or buf.numel() < total:
            buf = torch.empty(int(total * 1.25) + 16, dtype=torch.float32, device=dev)
            _SCRATCH[(dev, slot)] = buf
        base = buf.data_ptr()
        d_raw = torch.empty((R, Dn), dtype=torch.float32, device=dev) if want_d_raw else None
        b = LayerBwdDesc()
        gb = gblock.data_ptr()
        b.grads = LayerParams(*[gb + 4 * o for o, _, _ in self.grad_layout(params)[0]])
        b.dout = dout.data_ptr()
        for name, o, n in zip(_BWD_FIELDS, offs, sizes):
            setattr(b, name, base + 4 * o if n else None)
        b.vec = vec.data_ptr()
        b.d_cosb, b.d_tew, b.d_teb = d_cosb.data_ptr(), d_tew.data_ptr(), d_teb.data_ptr()
        b.dfeat, b.dfeat_ld, b.pad_row = ops._p(dfeat), (0 if dfeat is None else ops._rowmajor_ld(dfeat, "dfeat")), int(pad_row)
        b.d_own, b.d_own_ld, b.d_own_accumulate = ops._p(d_own), (0 if d_own is None else ops._rowmajor_ld(d_own, "d_own")), int(d_own_accumulate)
        b.d_raw = ops._p(d_raw)
        b.defer_join = int(defer_join)
        b.finish_time_bias = int(finish_time_bias)
        with ops._timed("layer_bwd", self.R):
            check(lib().tg_tgat_layer_bwd(C.byref(self.desc), C.byref(b), ops._stream()), "tg_tgat_layer_bwd")
        return d_raw


def block_layout(tensors):
    """offsets (in floats, 16-byte aligned) of `tensors` laid end to end, and the total length: the layout of the gradient block
    one backward pass accumulates into -- and of TGAT.flatten_parameters()' flat parameter, so that the block IS its gradient"""
    offs, total = [], 0
    for t in tensors:
        offs.append(total)
        total += _r4(t.numel())
    return offs, total


def _native_forward(cfg, fr, table, te_w, te_b, layer_params):
    """every layer = one native forward call (tg_tgat_layer_fwd); returns H^L and what the backward needs"""
    n, k, L, H = cfg["n"], cfg["k"], cfg["num_layers"], cfg["num_heads"]
    edge, p_drop, training = cfg["edge_table"], cfg["dropout"], cfg["training"]
    dev = table.device
    Dn, T = table.shape[1], te_w.numel()
    hd = (Dn + T) // H
    te_w_flat = te_w.reshape(-1)
    cosb = torch.empty(T, dtype=torch.float32, device=dev)         # cos(b): written by the first layer's prelude launch
    p_eff = p_drop if training else 0.0
    layers, H_prev = [], None
    for l in range(1, L + 1):
        params = layer_params[(l - 1) * 11:(l - 1) * 11 + 11]
        R = fr.rows(L - l)
        S_nbr, S_eid, S_t, S_dt = fr.S_at(l)
        own = None if l == 1 else H_prev[:R]
        feat, feat_idx = (table, S_nbr[:R].reshape(-1) if fr.feat_idx0 is None else fr.feat_idx0[:R * k]) if l == 1 else (H_prev, fr.child[:R * k])
        seeds = _next_seeds(2) if p_eff > 0 else [0, 0]
        attn = ops.AttnArgs(feat, feat_idx, edge, S_eid[:R].reshape(-1), S_nbr[:R].reshape(-1), S_dt[:R].reshape(-1),
                            te_w_flat, te_b, k, H, hd ** -0.5, p_eff, seeds[0])
        lay = _NativeLayer(attn, params, own, table, fr.ids_all[:R], cosb, p_eff, seeds[1], compute_cosb=(l == 1))
        H_prev = lay.forward()
        layers.append(lay)
    return H_prev, (layers, cosb)


def _native_backward(cfg, fr, table, te_w, te_b, layer_params, saved, dH, extra_floats: int = 0, grad_ready=None):
    """every layer = one native backward call; returns (d_table or None, gradient block, offsets of [te_w, te_b, *layer_params]
    inside it, number of gradient floats).  The block is laid out like TGAT.flatten_parameters()' flat parameter."""
    layers, cosb = saved
    n, k, L = cfg["n"], cfg["k"], cfg["num_layers"]
    table_grad = cfg["table_grad"]
    dev = dH.device
    Dn, T = table.shape[1], te_w.numel()
    Dq = Dn + T
    # ONE zero fill: [gradient of te_w | te_b | every layer parameter (block_layout)] + scratch [d cos(b) | dq floats per layer]
    every = [te_w, te_b, *layer_params]
    offs, npar = block_layout(every)
    H, Dk = cfg["num_heads"], Dn + cfg["edge_table"].shape[1] + T
    from ._lib import check, lib
    vlen = _r4(int(lib().tg_tgat_layer_vec_floats(Dn, Dq, Dk, H)))      # per layer: zero scratch of tg_tgat_layer_bwd (`vec`)
    # (extra_floats: room right behind the parameter gradients for the caller's own ones -- TGN's GRU -- so that a flat parameter
    # spanning both gets its gradient as one tensor)
    xt = _r4(extra_floats)
    nz = npar + xt + _r4(T) + L * vlen
    # ... and, in the same fill, the lower layers' gradient rows (the attention backward scatters into them with atomics)
    dh_off, tot = {}, nz
    for l in range(L, 1, -1):
        dh_off[l] = tot
        tot += _r4(fr.rows(L - l + 1) * Dn)
    whole = torch.zeros(tot, device=dev)
    zeroed = whole[:nz]
    d_tew, d_teb, d_cosb = zeroed[:T], zeroed[offs[1]:offs[1] + T], zeroed[npar + xt:npar + xt + T]
    d_table = torch.zeros_like(table) if table_grad else None
    dH = dH.contiguous()
    # The side streams (weight gradients) are joined ONCE, after the last layer: until then everything they read stays alive
    # (`alive`) and untouched (per-layer scratch), and nothing that lives on them is consumed.  Whatever happens in between
    # (an allocation failure, a TgError from a layer), the join runs before `alive` is released: queued side-stream products
    # must not outlive their operands.
    alive = [dH]
    try:
        for l in range(L, 0, -1):
            lay = layers[l - 1]
            R = lay.R
            params = layer_params[(l - 1) * 11:(l - 1) * 11 + 11]
            if l >= 2:
                nr = fr.rows(L - l + 1)
                dH_prev = whole[dh_off[l]:dh_off[l] + nr * Dn].view(nr, Dn)
                dfeat, pad_row = dH_prev, (fr.pad_rows[0] if (l == L and fr.pad_rows) else -1)
                d_own, acc = dH_prev[:R], True               # rows [0, R) of the lower layer's gradient: its "own" inputs
            else:
                dH_prev, dfeat, pad_row = None, d_table, fr.pad_row0
                d_own, acc = (torch.empty((R, Dn), device=dev), False) if table_grad else (None, False)
            v0 = npar + xt + _r4(T) + (l - 1) * vlen
            d_raw = lay.backward(dH[:R], params, zeroed[offs[2 + (l - 1) * 11]:], zeroed[v0:v0 + vlen], d_cosb, d_tew, d_teb,
                                 dfeat, pad_row, d_own, acc, table_grad, slot=l, defer_join=l > 1 and grad_ready is None,
                                 finish_time_bias=l == 1)
            alive += [dH_prev, d_own, d_raw]
            if table_grad:
                if l >= 2:
                    ops.scatter_add_rows(d_raw, fr.ids_all[:R], d_table)
                else:
                    # (two scatters, not `d_own += d_raw` and one: that torch add cost the host 300 us per step, measured with a per-line host profile)
                    ops.scatter_add_rows(d_own, fr.ids_all[:R], d_table)
                    ops.scatter_add_rows(d_raw, fr.ids_all[:R], d_table)
            if grad_ready is not None and l >= 2:
                # layer l's own parameter gradients are final once its backward is queued (the time encoder's block keeps
                # accumulating until layer 1): a data-parallel caller can start reducing this segment under the lower layers' backward
                # (with grad_ready the layer joined its side streams before returning -- defer_join off above: the segment is complete
                # in stream order, also under tg_set_overlap(1))
                lo_ = offs[2 + (l - 1) * 11]
                hi_ = offs[2 + l * 11] if l < L else npar
                grad_ready(zeroed[lo_:hi_])
            dH = dH_prev
    finally:
        check(lib().tg_side_join(ops._stream()), "tg_side_join")
    del alive
    # (d cos(b) -> d b, d_teb -= sin(b) * d_cosb, rode at the end of layer 1's call: finish_time_bias)
    return d_table, zeroed, offs, npar


_SEED_GEN = None


def seed_dropout(seed: int, rank: int = 0):
    """(re)seed the dropout stream of the native layers; `rank` decorrelates data-parallel replicas that share `seed`"""
    global _SEED_GEN
    _SEED_GEN = np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFFFFFFFFFF, int(rank)]))


def _next_seeds(n):
    """n 62-bit seeds.  Default stream: torch's CPU generator mixed with the distributed rank (every data-parallel rank usually
    seeds torch identically, and identical mask streams on different data would correlate the replicas' noise)."""
    if _SEED_GEN is not None:
        return [int(v) for v in _SEED_GEN.integers(0, 2 ** 62, size=n)]
    vals = torch.randint(0, 2 ** 62, (n,)).tolist()
    rank = _rank()
    return [(v ^ (rank * 0x9E3779B97F4A7C15)) & (2 ** 62 - 1) for v in vals] if rank else vals


def _rank():
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class _EmbedFnNative(torch.autograd.Function):
    """Same contract as _EmbedFn; every layer is one native forward call and one native backward call.
    Differentiable inputs: (te_w, te_b, *layer_params), or -- cfg["flat_views"] set -- the single flat parameter those are views of."""

    @staticmethod
    def forward(ctx, cfg, fr, table, *tensors):
        views = cfg.get("flat_views")
        te_w, te_b, *layer_params = views if views is not None else tensors
        out, saved = _native_forward(cfg, fr, table, te_w, te_b, layer_params)
        ctx.saved, ctx.cfg, ctx.fr, ctx.table = saved, cfg, fr, table
        ctx.params = (te_w, te_b, layer_params)        # parameters (or views of the flat parameter): not outputs, no cycle
        return out

    @staticmethod
    def backward(ctx, dH):
        cfg, fr, table = ctx.cfg, ctx.fr, ctx.table
        te_w, te_b, layer_params = ctx.params
        d_table, zeroed, offs, npar = _native_backward(cfg, fr, table, te_w, te_b, layer_params, ctx.saved, dH)
        ctx.saved = ctx.params = None
        if cfg.get("flat_views") is not None:
            return (None, None, d_table, zeroed[:npar])       # the block is laid out like the flat parameter: it IS its gradient
        every = [te_w, te_b, *layer_params]
        grads = [zeroed[o:o + t.numel()].view(t.shape) for o, t in zip(offs, every)]
        return (None, None, d_table, *grads)


def forward_backward(cfg, fr, table, flat, loss_fn, grad_ready=None):
    """Fused-trainer form of one step: forward, the caller's loss on the embeddings, backward -- with no autograd
    graph in between.  flat = (flat parameter, views) of TGAT.flatten_parameters(); loss_fn(emb) -> (loss, d loss / d emb) sees the
    (n, Dn) embeddings detached.  Returns (embeddings, loss); the gradient is ADDED to flat[0].grad (set if None), exactly what
    loss.backward() through embed() would have left there.
    grad_ready(segment): called with each upper layer's block of the NEW gradient as soon as that layer's backward is queued (the
    segments are views of the gradient block; the rest -- time encoder + layer 1 -- is final when this returns).  With it the
    gradient is SET, not added (flat[0].grad must be None: the segments handed out are the .grad storage itself)."""
    views = flat[1]
    te_w, te_b, layer_params = views[0], views[1], views[2:]
    with torch.no_grad():
        out, saved = _native_forward(cfg, fr, table, te_w, te_b, layer_params)
    loss, d_out = loss_fn(out)
    with torch.no_grad():
        if grad_ready is not None and flat[0].grad is not None:
            raise RuntimeError("forward_backward(grad_ready=...): zero_grad(set_to_none=True) first")
        _, zeroed, _, npar = _native_backward(cfg, fr, table, te_w, te_b, layer_params, saved, d_out, grad_ready=grad_ready)
        g = zeroed[:npar]
        if flat[0].grad is None:
            flat[0].grad = g
        else:
            flat[0].grad.add_(g)
    return out, loss


_ZERO1 = {}


def _zero1(dev):
    z = _ZERO1.get(dev)
    if z is None:
        z = _ZERO1[dev] = torch.zeros(1, device=dev)
    return z


class _SplitRows(torch.autograd.Function):
    """(x[:n], x[n:]) whose backward is ONE concatenation (the stock slice backward is two zero fills, two copies and an add)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n, ctx.shape = n, x.shape
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, g0, g1):
        n, shape = ctx.n, ctx.shape
        if g0 is None and g1 is None:
            return None, None
        ref = g0 if g0 is not None else g1
        if g0 is None:
            g0 = ref.new_zeros((n,) + tuple(shape[1:]))
        if g1 is None:
            g1 = ref.new_zeros((shape[0] - n,) + tuple(shape[1:]))
        return torch.cat([g0, g1]), None


def split_rows(x: torch.Tensor, n: int):
    """the src / dst halves of a jointly computed embedding block"""
    if not x.requires_grad:
        return x[:n], x[n:]
    return _SplitRows.apply(x, n)


@dataclass
class PreparedFrontier:
    """Neighbor lookups + row sharing of one batch, done ahead of time on a side stream (they depend on the graph only, not
    on the weights): the data-loader style prefetch that keeps the one small readback of the step (the number of distinct
    rows) off the critical path."""
    frontier: Frontier
    ready: "torch.cuda.Event"
    graph: TemporalGraph
    n: int
    k: int
    num_layers: int


_prefetch_stream = None


_event_ring = [None, 0]


def _ring_event():
    """an event from a ring of 32, to be re-recorded: creating one costs ~9 us (hipEventCreate), a prepared batch needs two.  A
    holder that outlives 32 later records simply waits for later work of the same in-order stream."""
    if _event_ring[0] is None:
        _event_ring[0] = [torch.cuda.Event() for _ in range(32)]
    ev = _event_ring[0][_event_ring[1] % 32]
    _event_ring[1] += 1
    return ev


def _side_stream():
    global _prefetch_stream
    if _prefetch_stream is None:
        _prefetch_stream = torch.cuda.Stream()
    return _prefetch_stream


@dataclass
class FrontierJob:
    """first half of a prepared frontier: level-0 lookups + row sharing are in flight on the side stream, the count of distinct
    rows is on its way to a pinned host word; prepare_finish() reads it (no wait if a step has passed) and issues the rest"""
    graph: TemporalGraph
    n: int
    k: int
    num_layers: int
    main: "torch.cuda.Stream"
    done: Optional[Frontier] = None            # already complete (depth 1, or the synchronous fall-back)
    S: tuple = None
    ids_all: torch.Tensor = None
    uniq_t: torch.Tensor = None
    child: torch.Tensor = None
    count_host: torch.Tensor = None
    count_ready: "torch.cuda.Event" = None


def prepare_begin(graph: TemporalGraph, ids_dev, times_dev, k: int, num_layers: int, inputs_ready: bool = True) -> FrontierJob:
    """Stage 1 of the prefetch (all on the side stream, nothing blocks the host): concatenate the ids, level-0 lookups, row
    sharing, and an asynchronous copy of the distinct-row count into pinned memory.  ids int32 / times float64 on the device
    (or lists of such tensors).  inputs_ready=False makes the side stream first wait for everything queued on the current stream
    (only needed when the ids were produced by kernels still in flight there)."""
    side, main = _side_stream(), torch.cuda.current_stream()
    if not inputs_ready:
        side.wait_stream(main)
    with torch.cuda.stream(side):
        # lists of tensors are concatenated HERE, on the side stream: a cat issued on the busy main stream would not have
        # run yet when the side stream reads its result
        if isinstance(ids_dev, (list, tuple)):
            ids_dev, times_dev = torch.cat(list(ids_dev)), torch.cat(list(times_dev))
        ids_dev, times_dev = ids_dev.to(torch.int32).contiguous(), times_dev.contiguous()
        n, dev = ids_dev.numel(), ids_dev.device
        job = FrontierJob(graph, n, k, num_layers, main)
        if num_layers != 2 or not DEDUPE or n == 0:
            job.done = sample_frontier(graph, ids_dev, times_dev, k, num_layers, dedupe=DEDUPE)      # (reads counts back at once)
            return job
        cap = n + n * k                       # rows of levels 0 and 1 if no pair repeated
        S = (torch.empty((cap, k), dtype=torch.int32, device=dev), torch.empty((cap, k), dtype=torch.int32, device=dev),
             torch.empty((cap, k), dtype=torch.float32, device=dev), torch.empty((cap, k), dtype=torch.float32, device=dev))
        graph.sample_recent(ids_dev, times_dev, k, out=tuple(x[:n] for x in S))
        ids_all = torch.empty(cap, dtype=torch.int32, device=dev)
        ids_all[:n].copy_(ids_dev)
        uniq_t = torch.empty(n * k, dtype=torch.float32, device=dev)
        child = torch.empty(n * k, dtype=torch.int32, device=dev)
        cp = graph.dedupe_pairs_async(S[0][:n].reshape(-1), S[2][:n].reshape(-1), n, ids_all[n:], uniq_t, child)
        job.count_host = torch.empty(2, dtype=torch.int32, pin_memory=True)
        job.count_host.copy_(cp, non_blocking=True)
        job.count_ready = _ring_event()
        job.count_ready.record()
        job.S, job.ids_all, job.uniq_t, job.child = S, ids_all, uniq_t, child
    return job


def prepare_finish(job: FrontierJob) -> PreparedFrontier:
    """Stage 2: read the distinct-row count (pinned word, copied by stage 1) and issue the level-1 lookups on the side stream."""
    side = _side_stream()
    if job.done is not None:
        fr = job.done
    else:
        job.count_ready.synchronize()
        count, pad = job.count_host.tolist()
        n, k = job.n, job.k
        with torch.cuda.stream(side):
            job.graph.sample_recent(job.ids_all[n:n + count], job.uniq_t[:count], k, out=tuple(x[n:n + count] for x in job.S))
        fr = Frontier(counts=[n, count], ids_all=job.ids_all[:n + count], S=tuple(x[:n + count] for x in job.S), child=job.child,
                      pad_rows=[pad + n if pad >= 0 else -1])
    with torch.cuda.stream(side):
        ev = _ring_event()
        ev.record()
    for t in (fr.ids_all, fr.child) + tuple(fr.S):            # allocated on the side stream, consumed on the main stream
        if t is not None:
            t.record_stream(job.main)
    return PreparedFrontier(fr, ev, job.graph, job.n, job.k, job.num_layers)


def prepare_frontier(graph: TemporalGraph, ids_dev: torch.Tensor, times_dev: torch.Tensor, k: int, num_layers: int,
                     inputs_ready: bool = True) -> PreparedFrontier:
    """both stages at once (the host waits for the distinct-row count of THIS batch; a trainer that knows its batches two
    steps ahead calls prepare_begin / prepare_finish one step apart and never waits)"""
    return prepare_finish(prepare_begin(graph, ids_dev, times_dev, k, num_layers, inputs_ready))


def embed(graph: TemporalGraph, table: torch.Tensor, edge_table: torch.Tensor, te_w, te_b, layer_params, ids: np.ndarray,
          times: np.ndarray, k: int, num_layers: int, num_heads: int, dropout: float, training: bool,
          table_requires_grad: bool = False, flat=None, host_sampler=None, groups=None):
    """H^L for `ids` at `times` (host numpy in, device tensor out, autograd-connected to the parameters).
    `ids` may also be a PreparedFrontier (prepare_frontier): the neighbor lookups of that batch were already done."""
    dev = table.device
    assert k > 0, 'Number of sampled neighbors for each node should be greater than 0!'
    if isinstance(ids, PreparedFrontier):
        pf = ids
        assert pf.k == k and pf.num_layers == num_layers and pf.graph is graph, "prepared for a different sampler / k / depth"
        torch.cuda.current_stream().wait_event(pf.ready)
        cfg = dict(n=pf.n, k=k, num_layers=num_layers, num_heads=num_heads, dropout=float(dropout), training=bool(training),
                   edge_table=edge_table, table_grad=bool(table_requires_grad))
        return _apply(cfg, pf.frontier, table, te_w, te_b, layer_params, flat)
    if torch.is_tensor(ids):
        # already resident in HBM (int32 ids, float64/float32 times): the caller vouches for the id range
        ids_dev, times_dev = ids.to(device=dev, dtype=torch.int32).contiguous(), times.to(dev).contiguous()
        n = ids_dev.numel()
    else:
        ids = np.asarray(ids)
        if len(ids) and (int(ids.max()) >= graph.num_rows or int(ids.min()) < 0):
            raise IndexError("list index out of range")                      # what utils/utils.py:141 raises
        n = len(ids)
        tt = np.asarray(times)
        ids_dev, times_dev = ops.h2d([np.ascontiguousarray(ids, dtype=np.int32),
                                      np.ascontiguousarray(tt, dtype=np.float32 if tt.dtype == np.float32 else np.float64)], dev)
    if n == 0:
        return torch.zeros((0, table.shape[1]), device=dev)
    if num_layers == 0:
        return ops.gather_rows(table, ids_dev)
    if host_sampler is not None and getattr(host_sampler, "device_random", False):
        fr = frontier_from_device_random(host_sampler, ids_dev, times_dev, k, num_layers)      # opt-in: counter-based draws on the device
    elif host_sampler is not None:        # uniform / time_interval_aware: numpy RandomState stream on the host, kernels unchanged
        if torch.is_tensor(ids):
            ids, times = ids.cpu().numpy(), times.cpu().numpy()
        fr = frontier_from_host_sampler(host_sampler, ids, times, k, num_layers, dev, groups)
    else:
        fr = sample_frontier(graph, ids_dev, times_dev, k, num_layers, dedupe=DEDUPE)
    cfg = dict(n=n, k=k, num_layers=num_layers, num_heads=num_heads, dropout=float(dropout), training=bool(training),
               edge_table=edge_table, table_grad=bool(table_requires_grad))
    return _apply(cfg, fr, table, te_w, te_b, layer_params, flat)


def _apply(cfg, fr, table, te_w, te_b, layer_params, flat):
    """flat = (flat parameter, [views of it: te_w, te_b, *layer_params]) from TGAT.flatten_parameters(), or None"""
    native = NATIVE and cfg["num_heads"] <= 2          # the one-call-per-layer path covers 1 or 2 heads (the reference default: 2);
                                                       # more heads take the op-by-op composition of the same HIP kernels
    if flat is not None and native:
        cfg["flat_views"] = flat[1]
        return _EmbedFnNative.apply(cfg, fr, table, flat[0])
    if flat is not None:                                # flat parameter + op-by-op path: differentiate through its views
        views = [flat[0][o:o + v.numel()].view(v.shape) for o, v in zip(block_layout(flat[1])[0], flat[1])]
        te_w, te_b, layer_params = views[0], views[1], views[2:]
    fn = _EmbedFnNative if native else _EmbedFn
    return fn.apply(cfg, fr, table, te_w, te_b, *layer_params)
