"""Torch-tensor front ends of the C ABI.  torch supplies device memory and the stream; every FLOP and every gathered byte
goes through libflid_tg.so."""
import ctypes as C
from typing import Optional

import numpy as np
import torch

from ._lib import AttnDesc, check, lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """the calling thread's current HIP stream as a void*.  Two C calls (~0.3 us): torch.cuda.current_stream() builds a Stream object
    through several Python layers (~4 us), and a training step asks 30-70 times."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())                 # a plain int: ctypes converts it for a c_void_p parameter
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """HIP-event timing of selected kernel families ON THE STREAM THEY ARE LAUNCHED ON (bench.py roofline leg).
    Usage: KernelTimer.enable({"attn_fwd", ...}); run; KernelTimer.collect() -> {name: [(ms, tag), ...]}."""
    names = frozenset()
    pending = []

    @classmethod
    def enable(cls, names):
        cls.names, cls.pending = frozenset(names), []

    @classmethod
    def disable(cls):
        cls.names = frozenset()

    @classmethod
    def collect(cls):
        torch.cuda.synchronize()
        out = {}
        for name, tag, a, b in cls.pending:
            out.setdefault(name, []).append((a.elapsed_time(b), tag))
        cls.pending = []
        return out


class _timed:
    def __init__(self, name, tag=None):
        self.on = name in KernelTimer.names
        self.name, self.tag = name, tag

    def __enter__(self):
        if self.on:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.b.record()
            KernelTimer.pending.append((self.name, self.tag, self.a, self.b))
        return False


def _p(t: Optional[torch.Tensor]):
    # a plain int (None = NULL): ctypes converts either for a c_void_p parameter or structure field, and building a c_void_p
    # object per argument was ~0.4 us x ~350 arguments per training step
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, dtype, name):
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a device tensor (flid_amd has no CPU path)")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


def _rowmajor_ld(t: torch.Tensor, name):
    """(rows, cols) view with unit column stride -> leading dimension"""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f"{name}: need a 2-D tensor with contiguous columns, got shape {tuple(t.shape)} strides {t.stride()}")
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def gemm(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, ta=False, tb=False, bias=None, relu=False, accumulate=False,
         alpha=1.0):
    """out[M,N] = alpha * op(a) @ op(b) (+bias) (+out) (relu).  a: (M,K) or (K,M) if ta; b: (K,N) or (N,K) if tb."""
    _chk(a, torch.float32, "a"); _chk(b, torch.float32, "b"); _chk(out, torch.float32, "out")
    M, K = (a.shape[1], a.shape[0]) if ta else (a.shape[0], a.shape[1])
    Kb, N = (b.shape[1], b.shape[0]) if tb else (b.shape[0], b.shape[1])
    if K != Kb or out.shape[0] != M or out.shape[1] != N:
        raise ValueError(f"gemm: shape mismatch op(a)=({M},{K}) op(b)=({Kb},{N}) out={tuple(out.shape)}")
    with _timed("gemm", (M, N, K)):
        check(lib().tg_gemm_f32(int(ta), int(tb), M, N, K, float(alpha), _p(a), _rowmajor_ld(a, "a"), _p(b), _rowmajor_ld(b, "b"),
                                _p(out), _rowmajor_ld(out, "out"), _p(bias), int(relu), int(accumulate), _stream()), "tg_gemm_f32")
    return out


def gemm_batched(a0: torch.Tensor, b0: torch.Tensor, out0: torch.Tensor, batch: int, stride_a: int, stride_b: int, stride_c: int,
                 ta=False, tb=False, accumulate=False):
    """`batch` products in one launch; a0/b0/out0 are the views of problem 0, problem i starts stride_* floats further."""
    _chk(a0, torch.float32, "a"); _chk(b0, torch.float32, "b"); _chk(out0, torch.float32, "out")
    M, K = (a0.shape[1], a0.shape[0]) if ta else (a0.shape[0], a0.shape[1])
    Kb, N = (b0.shape[1], b0.shape[0]) if tb else (b0.shape[0], b0.shape[1])
    if K != Kb or out0.shape[0] != M or out0.shape[1] != N:
        raise ValueError(f"gemm_batched: shape mismatch op(a)=({M},{K}) op(b)=({Kb},{N}) out={tuple(out0.shape)}")
    with _timed("gemm", (M, N, K * batch)):
        check(lib().tg_gemm_f32_batched(int(ta), int(tb), M, N, K, 1.0, _p(a0), _rowmajor_ld(a0, "a"), stride_a, _p(b0),
                                        _rowmajor_ld(b0, "b"), stride_b, _p(out0), _rowmajor_ld(out0, "out"), stride_c, batch,
                                        _p(None), 0, int(accumulate), _stream()), "tg_gemm_f32_batched")


def pack_weights(jobs):
    """jobs: [(weight (N, K) fp32 row-major, or (K, N) with trans=True), trans]; returns one packed operand per job (tg_pack_weights:
    bf16 hi / lo in MFMA fragment order, what the chain kernels multiply with).  One launch for all of them."""
    import ctypes as C
    from ._lib import PackJob
    arr = (PackJob * len(jobs))()
    outs = []
    for i, (w, trans) in enumerate(jobs):
        _chk(w, torch.float32, "weight")
        N, K = (w.shape[1], w.shape[0]) if trans else (w.shape[0], w.shape[1])
        dst = torch.empty(int(lib().tg_packed_floats(N, K)), dtype=torch.float32, device=w.device)
        arr[i] = PackJob(_p(w), _rowmajor_ld(w, "weight"), N, K, int(bool(trans)), _p(dst), 0, 0, 0, 0, 0, 0)
        outs.append((dst, N, K))
    check(lib().tg_pack_weights(len(jobs), arr, _stream()), "tg_pack_weights")
    return outs


def gather_rows(table: torch.Tensor, idx: torch.Tensor, out: Optional[torch.Tensor] = None):
    _chk(table, torch.float32, "table"); _chk(idx, torch.int32, "idx")
    n, cols = idx.numel(), table.shape[1]
    if out is None:
        out = torch.empty((n, cols), dtype=torch.float32, device=table.device)
    check(lib().tg_gather_rows(_p(table), _rowmajor_ld(table, "table"), _p(idx), n, cols, _p(out), _rowmajor_ld(out, "out"),
                               _stream()), "tg_gather_rows")
    return out


def scatter_add_rows(src: torch.Tensor, idx: torch.Tensor, table: torch.Tensor):
    _chk(src, torch.float32, "src"); _chk(idx, torch.int32, "idx"); _chk(table, torch.float32, "table")
    check(lib().tg_scatter_add_rows(_p(src), _rowmajor_ld(src, "src"), _p(idx), idx.numel(), src.shape[1], _p(table),
                                    _rowmajor_ld(table, "table"), _stream()), "tg_scatter_add_rows")
    return table


def add_layernorm_fwd(a, b, gamma, beta):
    n, cols = a.shape
    y = torch.empty_like(a)
    mean = torch.empty(n, dtype=torch.float32, device=a.device)
    rstd = torch.empty(n, dtype=torch.float32, device=a.device)
    assert a.is_contiguous() and (b is None or b.is_contiguous())
    check(lib().tg_add_layernorm_fwd(_p(a), _p(b), n, cols, _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), _stream()),
          "tg_add_layernorm_fwd")
    return y, mean, rstd


def add_layernorm_bwd(a, b, dy, gamma, mean, rstd):
    """returns dx (n, cols) and (dgamma, dbeta)"""
    n, cols = a.shape
    assert a.is_contiguous() and (b is None or b.is_contiguous()) and dy.is_contiguous()
    parts = lib().tg_rowop_parts(n)
    dx = torch.empty_like(a)
    part = torch.empty((parts, 2 * cols), dtype=torch.float32, device=a.device)
    check(lib().tg_add_layernorm_bwd(_p(a), _p(b), _p(dy), n, cols, _p(gamma), _p(mean), _p(rstd), _p(dx), _p(part), _stream()),
          "tg_add_layernorm_bwd")
    dgb = colsum(part)
    return dx, dgb[:cols], dgb[cols:]


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate=False):
    _chk(x, torch.float32, "x")
    n, cols = x.shape
    if out is None:
        out = torch.empty(cols, dtype=torch.float32, device=x.device)
        accumulate = False
    check(lib().tg_colsum(_p(x), _rowmajor_ld(x, "x"), n, cols, _p(out), int(accumulate), _stream()), "tg_colsum")
    return out


def relu_bwd_(dy: torch.Tensor, y: torch.Tensor):
    assert dy.is_contiguous() and y.is_contiguous() and dy.shape == y.shape
    check(lib().tg_relu_bwd_inplace(_p(dy), _p(y), dy.numel(), _stream()), "tg_relu_bwd_inplace")
    return dy


def time_encode(t: torch.Tensor, w: torch.Tensor, b: torch.Tensor, fused_fma=True):
    _chk(t, torch.float32, "t")
    t = t.contiguous()
    dim = w.numel()
    out = torch.empty(tuple(t.shape) + (dim,), dtype=torch.float32, device=t.device)
    check(lib().tg_time_encode(_p(t), t.numel(), _p(w), _p(b), dim, int(fused_fma), _p(out), _stream()), "tg_time_encode")
    return out


class AttnArgs:
    """Keeps the tensors behind a tg_attn_desc alive and builds the C struct."""

    def __init__(self, feat, feat_idx, edge, edge_idx, nbr, dt, te_w, te_b, k, heads, scale, dropout_p=0.0, seed=0):
        _chk(feat, torch.float32, "feat"); _chk(edge, torch.float32, "edge")
        for t, nm in ((feat_idx, "feat_idx"), (edge_idx, "edge_idx"), (nbr, "nbr")):
            _chk(t, torch.int32, nm)
            assert t.is_contiguous()
        _chk(dt, torch.float32, "dt")
        self.keep = (feat, feat_idx, edge, edge_idx, nbr, dt, te_w, te_b)
        m = nbr.numel() // k
        self.m, self.k, self.heads = m, k, heads
        self.p, self.seed = float(dropout_p), int(seed)
        self.dn, self.de, self.dt_dim = feat.shape[1], edge.shape[1], (0 if te_w is None else te_w.numel())
        self.dk = self.dn + self.de + self.dt_dim
        self.desc = AttnDesc(_p(feat), _rowmajor_ld(feat, "feat"), _p(feat_idx), _p(edge), _rowmajor_ld(edge, "edge"), _p(edge_idx),
                             _p(nbr), _p(dt), _p(te_w), _p(te_b), m, k, heads, self.dn, self.de, self.dt_dim,
                             float(scale), float(dropout_p), int(seed))


def attn_fwd(args: AttnArgs, u: torch.Tensor):
    assert u.is_contiguous() and u.shape == (args.m, args.heads, args.dk)
    agg = torch.empty_like(u)
    prob = torch.empty((args.m, args.heads, args.k), dtype=torch.float32, device=u.device)
    with _timed("attn_fwd", args.m):
        check(lib().tg_attn_fwd(C.byref(args.desc), _p(u), _p(agg), _p(prob), _stream()), "tg_attn_fwd")
    return agg, prob


def attn_dropped_scores(args: AttnArgs, prob: torch.Tensor):
    """the attention scores AFTER dropout (what the reference's MultiHeadAttention.forward returns, modules.py:224,242): prob times
    the keep / rescale factor of the kernels' counter-based mask stream"""
    out = torch.empty_like(prob)
    check(lib().tg_attn_dropped_scores(_p(prob), args.m, args.heads, args.k, args.p, args.seed, _p(out), _stream()), "tg_attn_dropped_scores")
    return out


def attn_bwd(args: AttnArgs, u, agg, prob, dagg, dfeat: Optional[torch.Tensor] = None, pad_row: int = -1,
             dedge: Optional[torch.Tensor] = None):
    """returns du, (dw, db) of the time encoder; adds the neighbor-feature gradient into dfeat rows (and the edge-row gradient into
    dedge rows) if given"""
    assert dagg.is_contiguous() and dagg.shape == u.shape
    du = torch.empty_like(u)
    parts = lib().tg_attn_bwd_parts(args.m)
    part = torch.empty((parts, max(1, 2 * args.dt_dim)), dtype=torch.float32, device=u.device)
    with _timed("attn_bwd", args.m):
        check(lib().tg_attn_bwd(C.byref(args.desc), _p(u), _p(agg), _p(prob), _p(dagg), _p(du), _p(dfeat),
                                0 if dfeat is None else _rowmajor_ld(dfeat, "dfeat"), int(pad_row), _p(dedge),
                                0 if dedge is None else _rowmajor_ld(dedge, "dedge"), _p(part), _stream()), "tg_attn_bwd")
    if args.dt_dim == 0:                     # no time segment in z (stand-alone MultiHeadAttention.forward): nothing to sum
        return du, None, None
    dwb = colsum(part)
    return du, dwb[:args.dt_dim], dwb[args.dt_dim:]


def gru_cell_fwd(x, h, w_ih, w_hh, b_ih, b_hh):
    """nn.GRUCell forward on the MFMA GEMM + fused gate kernel.  Returns (h_new, gi, gh) -- gi/gh are kept for backward."""
    n, d = h.shape
    gi = torch.empty((n, 3 * d), device=h.device)
    gh = torch.empty((n, 3 * d), device=h.device)
    gemm(x, w_ih, gi, tb=True, bias=b_ih)
    gemm(h, w_hh, gh, tb=True, bias=b_hh)
    out = torch.empty_like(h)
    check(lib().tg_gru_gates_fwd(_p(gi), _p(gh), _p(h), n, d, _p(out), _stream()), "tg_gru_gates_fwd")
    return out, gi, gh


def gru_cell_bwd(x, h, gi, gh, dout, w_ih, w_hh):
    """parameter gradients of the GRU cell (inputs x, h are detached state in TGN): dW_ih, dW_hh, db_ih, db_hh"""
    n, d = h.shape
    dgi, dgh = torch.empty_like(gi), torch.empty_like(gh)
    check(lib().tg_gru_gates_bwd(_p(gi), _p(gh), _p(h), _p(dout.contiguous()), n, d, _p(dgi), _p(dgh), _p(None), _stream()),
          "tg_gru_gates_bwd")
    dw_ih, dw_hh = torch.empty_like(w_ih), torch.empty_like(w_hh)
    gemm(dgi, x, dw_ih, ta=True)
    gemm(dgh, h, dw_hh, ta=True)
    return dw_ih, dw_hh, colsum(dgi), colsum(dgh)


def build_messages(mem, last_update, a_ids, b_ids, t32, edge, eids, te_w, te_b):
    n, d, de, T = a_ids.numel(), mem.shape[1], edge.shape[1], te_w.numel()
    out = torch.empty((n, 2 * d + T + de), device=mem.device)
    check(lib().tg_build_messages(_p(mem), _rowmajor_ld(mem, "mem"), _p(last_update), _p(a_ids), _p(b_ids), _p(t32), _p(edge),
                                  _rowmajor_ld(edge, "edge"), _p(eids), _p(te_w), _p(te_b), n, d, de, T, _p(out), _stream()),
          "tg_build_messages")
    return out


def gemm_batched2(a00, b00, c00, outer, inner, sa, sb, sc, ta=False, tb=False, alpha=1.0, accumulate=False):
    """two-level strided batch: a00/b00/c00 are the views of problem (0, 0); sa/sb/sc = (outer stride, inner stride) in floats"""
    M, K = (a00.shape[1], a00.shape[0]) if ta else (a00.shape[0], a00.shape[1])
    Kb, N = (b00.shape[1], b00.shape[0]) if tb else (b00.shape[0], b00.shape[1])
    if K != Kb or c00.shape[0] != M or c00.shape[1] != N:
        raise ValueError(f"gemm_batched2: shape mismatch op(a)=({M},{K}) op(b)=({Kb},{N}) out={tuple(c00.shape)}")
    with _timed("gemm", (M, N, K * outer * inner)):
        check(lib().tg_gemm_f32_batched2(int(ta), int(tb), M, N, K, float(alpha), _p(a00), _rowmajor_ld(a00, "a"), sa[0], sa[1],
                                         _p(b00), _rowmajor_ld(b00, "b"), sb[0], sb[1], _p(c00), _rowmajor_ld(c00, "c"), sc[0], sc[1],
                                         int(outer), int(inner), int(accumulate), _stream()), "tg_gemm_f32_batched2")


def time_encode_masked(t, mask_ids, w, b):
    t = t.contiguous()
    out = torch.empty(tuple(t.shape) + (w.numel(),), dtype=torch.float32, device=t.device)
    check(lib().tg_time_encode_masked(_p(t), _p(mask_ids.contiguous()), t.numel(), _p(w), _p(b), w.numel(), _p(out), _stream()),
          "tg_time_encode_masked")
    return out


def cooccurrence(src_ids, dst_ids):
    """(n, ws), (n, wd) int32 -> (n, ws, 2), (n, wd, 2) float32"""
    n, ws = src_ids.shape
    wd = dst_ids.shape[1]
    o1 = torch.empty((n, ws, 2), device=src_ids.device)
    o2 = torch.empty((n, wd, 2), device=src_ids.device)
    check(lib().tg_cooccurrence(_p(src_ids), src_ids.stride(0), ws, _p(dst_ids), dst_ids.stride(0), wd, n, _p(o1), _p(o2), _stream()),
          "tg_cooccurrence")
    return o1, o2


def gelu_fwd(x):
    y = torch.empty_like(x)
    check(lib().tg_gelu_fwd(_p(x), x.numel(), _p(y), _stream()), "tg_gelu_fwd")
    return y


def gelu_bwd(x, dy):
    dx = torch.empty_like(x)
    check(lib().tg_gelu_bwd(_p(x), _p(dy), x.numel(), _p(dx), _stream()), "tg_gelu_bwd")
    return dx


def softmax_fwd(x):
    y = torch.empty_like(x)
    check(lib().tg_softmax_fwd(_p(x), x.numel() // x.shape[-1], x.shape[-1], _p(y), _stream()), "tg_softmax_fwd")
    return y


def softmax_keymask_fwd(x, key_ids, rows_per_batch):
    """row softmax with the columns whose key id is 0 masked out; x: (..., cols), key_ids: (batches, cols) int32"""
    _chk(key_ids, torch.int32, "key_ids")
    y = torch.empty_like(x)
    check(lib().tg_softmax_keymask_fwd(_p(x), x.numel() // x.shape[-1], x.shape[-1], _p(key_ids.contiguous()), int(rows_per_batch), _p(y),
                                       _stream()), "tg_softmax_keymask_fwd")
    return y


def softmax_bwd(y, dy):
    dx = torch.empty_like(y)
    check(lib().tg_softmax_bwd(_p(y), _p(dy), y.numel() // y.shape[-1], y.shape[-1], _p(dx), _stream()), "tg_softmax_bwd")
    return dx


def dropout(x, p, seed):
    y = torch.empty_like(x)
    check(lib().tg_dropout(_p(x), x.numel(), float(p), int(seed), _p(y), _stream()), "tg_dropout")
    return y


def gelu_dropout_fwd(x, p, seed):
    y = torch.empty_like(x)
    check(lib().tg_gelu_dropout_fwd(_p(x), x.numel(), float(p), int(seed), _p(y), _stream()), "tg_gelu_dropout_fwd")
    return y


def gelu_dropout_bwd(x, dy, p, seed):
    dx = torch.empty_like(x)
    check(lib().tg_gelu_dropout_bwd(_p(x), _p(dy), x.numel(), float(p), int(seed), _p(dx), _stream()), "tg_gelu_dropout_bwd")
    return dx


def dropout_add(x, res, p, seed):
    """res + dropout(x) in one pass"""
    y = torch.empty_like(x)
    check(lib().tg_dropout_add(_p(x), _p(res), x.numel(), float(p), int(seed), _p(y), _stream()), "tg_dropout_add")
    return y


def segment_mean_fwd(x, lo, hi):
    n, s, d = x.shape
    out = torch.empty((n, d), device=x.device)
    check(lib().tg_segment_mean_fwd(_p(x), n, s, d, lo, hi, _p(out), _stream()), "tg_segment_mean_fwd")
    return out


def segment_mean_bwd(dout, shape, lo, hi):
    n, s, d = shape
    dx = torch.zeros(shape, device=dout.device)
    check(lib().tg_segment_mean_bwd(_p(dout.contiguous()), n, s, d, lo, hi, _p(dx), _stream()), "tg_segment_mean_bwd")
    return dx


def time_encode_bwd(t, mask_ids, w, b, g):
    """(dw, db) of cos(fma(t, w, b)) [masked]; t (n,), g (n, dim)"""
    t = t.contiguous().reshape(-1)
    g = g.contiguous().reshape(t.numel(), -1)
    dim = w.numel()
    part = torch.empty((lib().tg_rowop_parts(t.numel()), 2 * dim), device=t.device)
    check(lib().tg_time_encode_bwd(_p(t), _p(mask_ids), t.numel(), _p(w), _p(b), dim, _p(g), _p(part), _stream()), "tg_time_encode_bwd")
    s = colsum(part)
    return s[:dim], s[dim:]


def wgrad_group(jobs):
    """jobs: list of (A (rows, M), B (rows, N), C (M, N) accumulated into, colsum_A (M,) accumulated into or None); one launch"""
    from ._lib import WgradJob
    arr = (WgradJob * len(jobs))()
    rows = jobs[0][0].shape[0]
    for q, (a, b, c, cs) in zip(arr, jobs):
        _chk(a, torch.float32, "A"); _chk(b, torch.float32, "B"); _chk(c, torch.float32, "C")
        assert a.shape[0] == rows and b.shape[0] == rows and c.shape == (a.shape[1], b.shape[1])
        q.A, q.lda, q.M, q.B, q.ldb, q.N = a.data_ptr(), _rowmajor_ld(a, "A"), a.shape[1], b.data_ptr(), _rowmajor_ld(b, "B"), b.shape[1]
        q.C, q.ldc, q.colsum_A = c.data_ptr(), _rowmajor_ld(c, "C"), (None if cs is None else cs.data_ptr())
    with _timed("gemm", rows):
        check(lib().tg_wgrad_group(len(jobs), arr, rows, _stream()), "tg_wgrad_group")


def weighted_sum(a: torch.Tensor, w: torch.Tensor, scale: float = 1.0, out: Optional[torch.Tensor] = None):
    """scale * sum(a * w) as a 1-element device tensor (one launch)"""
    _chk(a, torch.float32, "a"); _chk(w, torch.float32, "w")
    assert a.is_contiguous() and w.is_contiguous() and a.numel() == w.numel()
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=a.device)
    check(lib().tg_weighted_sum(_p(a), _p(w), a.numel(), float(scale), _p(out), _stream()), "tg_weighted_sum")
    return out


PROFILE_TAGS = {"attn_fwd": 1, "attn_bwd": 2, "gemm": 4, "tgn_advance": 8}


def profile_enable(on):
    """on: False / True (all families) / a tag name / an iterable of tag names"""
    if isinstance(on, str):
        on = (on,)
    mask = (7 if on else 0) if isinstance(on, (bool, int)) else sum(PROFILE_TAGS[t] for t in on)
    lib().tg_profile_enable(int(mask))


def profile_collect(tag: str, reset=True):
    """(total ms, total units, launches) of a kernel family since the last reset (HIP events on the launch stream)"""
    ms, units, cnt = C.c_double(), C.c_double(), C.c_int64()
    check(lib().tg_profile_collect(tag.encode(), C.byref(ms), C.byref(units), C.byref(cnt), int(reset)), "tg_profile_collect")
    return ms.value, units.value, cnt.value


def hash_features(nrows: int, cols: int, seed: int, device, chunk_rows: int = 1 << 22) -> torch.Tensor:
    """(nrows, cols) fp32 table generated straight into HBM: element = f(row, col, seed), row 0 zero (synth.hash_features_host
    recomputes any row on the host)"""
    out = torch.empty((nrows, cols), dtype=torch.float32, device=device)
    for r0 in range(0, nrows, chunk_rows):
        n = min(chunk_rows, nrows - r0)
        check(lib().tg_hash_features(_p(out[r0:]), cols, r0, n, cols, int(seed), _stream()), "tg_hash_features")
    return out


_NP2T = {np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64, np.dtype(np.float32): torch.float32,
         np.dtype(np.float64): torch.float64, np.dtype(np.uint8): torch.uint8, np.dtype(np.bool_): torch.bool}


def h2d(arrays, device):
    """Several host numpy arrays -> device tensors with ONE asynchronous copy from a pinned staging block.
    `torch.from_numpy(x).to(device)` on pageable memory blocks the host until the stream has drained up to the copy; a step that
    hands over a dozen small id / time arrays that way serialises host and GPU a dozen times (TGN: 2.0 ms wall for 1.0 ms of
    kernels).  The pinned block comes from torch's caching host allocator, which keeps it alive until the copy has run."""
    arrays = [np.ascontiguousarray(a) for a in arrays]
    offs, total = [], 0
    for a in arrays:
        offs.append(total)
        total += (a.nbytes + 15) // 16 * 16
    if total == 0:
        return [torch.empty(a.shape, dtype=_NP2T[a.dtype], device=device) for a in arrays]
    host = torch.empty(total, dtype=torch.uint8, pin_memory=True)
    hv = host.numpy()
    for a, o in zip(arrays, offs):
        if a.nbytes:
            hv[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
    dev = host.to(device, non_blocking=True)
    return [dev[o:o + a.nbytes].view(_NP2T[a.dtype]).reshape(a.shape) for a, o in zip(arrays, offs)]
