"""Mirror of the reference's building blocks (models/modules.py) -- same class names, constructor arguments, parameter
names and shapes (the state_dict contract of SURVEY.md 8b).  On the fused path the backbones read these modules'
parameters and run libflid_tg kernels; the modules' own forward() methods are HIP-backed too (ops.py), for callers that
use a block stand-alone."""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .._lib import TgShapeNotCovered


class _exact_products:
    """`with _exact_products(on):` -- the products issued inside run on the f32-input MFMA kernels (tg_set_gemm_mode(0), exact fp32
    multiply-add chains) instead of the split-bf16 ones (2^-17 per product).  For the deep sequence backbones: through TCL's eight
    post-LN block applications the split products' error reaches 1e-3 of the first layers' gradients (tests/test_gpu_backbones.py),
    the exact ones stay at 1e-4."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        if self.on:
            from .._lib import lib
            self.prev = lib().tg_get_gemm_mode_thread()      # (scopes nest: the enclosing override comes back on exit)
            lib().tg_set_gemm_mode_thread(0)        # this thread's calls only: other issuing threads keep the process-wide mode

    def __exit__(self, *exc):
        if self.on:
            from .._lib import lib
            lib().tg_set_gemm_mode_thread(self.prev)
        return False


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b on the MFMA GEMM (tg_gemm_f32), with its two transposed products in backward."""

    @staticmethod
    def forward(ctx, x, w, b, relu, exact=False):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        y = torch.empty((x2.shape[0], w.shape[0]), device=x.device)
        with _exact_products(exact):
            ops.gemm(x2, w, y, tb=True, bias=b, relu=relu)
        ctx.save_for_backward(x2, w, y if relu else None)
        ctx.has_bias, ctx.shape, ctx.exact = b is not None, x.shape, exact
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        if y is not None:
            dy2 = ops.relu_bwd_(dy2.clone(), y)
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx = dw = db = None
        if need_x:                                   # (raw-feature projections have no input gradient: skip the product)
            dx = torch.empty_like(x2)
            with _exact_products(ctx.exact):
                ops.gemm(dy2, w, dx)
            dx = dx.reshape(ctx.shape)
        if need_w and ctx.exact:
            dw = torch.empty_like(w)
            with _exact_products(True):
                ops.gemm(dy2, x2, dw, ta=True)
        elif need_w:
            # weight and bias gradient in ONE launch (tg_wgrad_group: split-bf16 MFMA, bias sum through the ones column, atomic fold)
            # when the shapes allow it; otherwise one exact product + one column sum
            n_out, n_in = w.shape
            if n_out % 4 == 0 and n_in % 4 == 0 and dy2.shape[0] >= 256 and dy2.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0:
                zb = torch.zeros(n_out * n_in + (n_out if need_b else 0), device=w.device)       # one fill for both gradients
                dw = zb[:n_out * n_in].view(n_out, n_in)
                db = zb[n_out * n_in:] if need_b else None
                try:
                    ops.wgrad_group([(dy2, x2, dw, db)])
                    need_b = False
                except TgShapeNotCovered:                # (e.g. operands beyond the kernel's 32-bit offsets): the general product + a column sum
                    dw = torch.empty_like(w)
                    ops.gemm(dy2, x2, dw, ta=True)
            else:
                dw = torch.empty_like(w)
                ops.gemm(dy2, x2, dw, ta=True)
        if need_b:
            db = ops.colsum(dy2)
        return dx, dw, db, None, None


def linear(x, w, b=None, relu=False, exact=False):
    return _LinearFn.apply(x, w, b, relu, exact)


class _TimeEncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, w, b, fused):
        out = ops.time_encode(t, w.reshape(-1), b, fused_fma=fused)
        ctx.save_for_backward(t, w, b)
        return out

    @staticmethod
    def backward(ctx, g):
        t, w, b = ctx.saved_tensors
        dw, db = ops.time_encode_bwd(t, None, w.reshape(-1), b, g)      # timestamps are data: no gradient w.r.t. t
        return None, dw.reshape(w.shape), db, None


class TimeEncoder(nn.Module):
    """cos(w t + b), w_j = 10^(-9 j / (T-1))   (reference models/modules.py:7-40)"""

    def __init__(self, time_dim: int, parameter_requires_grad: bool = True):
        super().__init__()
        self.time_dim = time_dim
        self.w = nn.Linear(1, time_dim)
        self.w.weight = nn.Parameter((torch.from_numpy(1 / 10 ** np.linspace(0, 9, time_dim, dtype=np.float32))).reshape(time_dim, -1))
        self.w.bias = nn.Parameter(torch.zeros(time_dim))
        if not parameter_requires_grad:
            self.w.weight.requires_grad = False
            self.w.bias.requires_grad = False

    def forward(self, timestamps: torch.Tensor):
        # (batch, seq) -> (batch, seq, time_dim); t*w+b is rounded once (fma), as the reference's CPU Linear(1,T) does
        return _TimeEncodeFn.apply(timestamps.contiguous().float(), self.w.weight, self.w.bias, True)


class MergeLayer(nn.Module):
    """fc2(relu(fc1([a | b])))   (reference models/modules.py:43-69)"""

    def __init__(self, input_dim1: int, input_dim2: int, hidden_dim: int, output_dim: int):
        super().__init__()
        self.fc1 = nn.Linear(input_dim1 + input_dim2, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, output_dim)
        self.act = nn.ReLU()

    def forward(self, input_1: torch.Tensor, input_2: torch.Tensor):
        x = torch.cat([input_1, input_2], dim=1)
        return linear(linear(x, self.fc1.weight, self.fc1.bias, relu=True), self.fc2.weight, self.fc2.bias)


class MLPClassifier(nn.Module):
    """decoder head (reference models/modules.py:72-97) -- caller-side, HIP GEMMs + torch dropout"""

    def __init__(self, input_dim: int, dropout: float = 0.1, num_classes: int = 2):
        super().__init__()
        self.fc1 = nn.Linear(input_dim, 80)
        self.fc2 = nn.Linear(80, 10)
        self.fc3 = nn.Linear(10, num_classes)
        self.act = nn.ReLU()
        self.dropout = nn.Dropout(dropout)

    def forward(self, x: torch.Tensor):
        x = self.dropout(linear(x, self.fc1.weight, self.fc1.bias, relu=True))
        x = self.dropout(linear(x, self.fc2.weight, self.fc2.bias, relu=True))
        return linear(x, self.fc3.weight, self.fc3.bias)


class MLPClassifier_BN(nn.Module):
    """decoder head with batch norm (reference models/modules.py:99-123; PTCL/EM_init.py:55-60 builds it for --emodel mlp_bn).
    Caller-side: HIP GEMMs, torch BatchNorm1d / Dropout."""

    def __init__(self, input_dim: int, dropout: float = 0.1, num_classes: int = 1):
        super().__init__()
        self.fc1 = nn.Linear(input_dim, 80)
        self.bn1 = nn.BatchNorm1d(80)
        self.fc2 = nn.Linear(80, 10)
        self.bn2 = nn.BatchNorm1d(10)
        self.fc3 = nn.Linear(10, num_classes)
        self.act = nn.ReLU()
        self.dropout = nn.Dropout(dropout)

    def forward(self, x: torch.Tensor):
        x = self.dropout(self.act(self.bn1(linear(x, self.fc1.weight, self.fc1.bias))))
        x = self.dropout(self.act(self.bn2(linear(x, self.fc2.weight, self.fc2.bias))))
        return linear(x, self.fc3.weight, self.fc3.bias)


class TransformerEncoder(nn.Module):
    """post-LN encoder block of the reference's TCL backbone (models/modules.py:248-312): nn.MultiheadAttention with a key padding
    mask built from the neighbor ids, residual + LayerNorm, Linear -> ReLU -> Linear, residual + LayerNorm.  Parameter names / shapes
    are the reference's (nn.MultiheadAttention stays as the parameter holder); every product, the masked softmax, LayerNorm and
    dropout run in libflid_tg (seqops.py)."""

    def __init__(self, attention_dim: int, num_heads: int, dropout: float = 0.1):
        super().__init__()
        self.multi_head_attention = nn.MultiheadAttention(embed_dim=attention_dim, num_heads=num_heads, dropout=dropout)
        self.num_heads, self.p = num_heads, dropout
        self.exact_products = True          # f32-input MFMA products (see _exact_products); False = the faster split-bf16 ones
        self.dropout = nn.Dropout(dropout)
        self.linear_layers = nn.ModuleList([nn.Linear(attention_dim, 4 * attention_dim), nn.Linear(4 * attention_dim, attention_dim)])
        self.norm_layers = nn.ModuleList([nn.LayerNorm(attention_dim), nn.LayerNorm(attention_dim)])

    def forward(self, inputs_query: torch.Tensor, inputs_key: torch.Tensor = None, inputs_value: torch.Tensor = None,
                neighbor_masks=None):
        """neighbor_masks: (batch, source_seq_length) neighbor ids, numpy (as the reference passes them) or an int32 device tensor"""
        from .. import seqops
        if inputs_key is None or inputs_value is None:
            assert inputs_key is None and inputs_value is None
            inputs_key = inputs_value = inputs_query
        mha, tr, p = self.multi_head_attention, self.training, self.p
        d = inputs_query.shape[-1]
        W, b = mha.in_proj_weight, mha.in_proj_bias
        key_ids = None
        if neighbor_masks is not None:
            key_ids = neighbor_masks if torch.is_tensor(neighbor_masks) else torch.from_numpy(np.ascontiguousarray(neighbor_masks))
            key_ids = key_ids.to(device=inputs_query.device, dtype=torch.int32)
        ex = self.exact_products
        q = linear(inputs_query, W[:d], b[:d], exact=ex)
        if inputs_key is inputs_value:
            kv = linear(inputs_key, W[d:], b[d:], exact=ex)                         # keys and values in one product
        else:
            kv = torch.cat([linear(inputs_key, W[d:2 * d], b[d:2 * d], exact=ex), linear(inputs_value, W[2 * d:], b[2 * d:], exact=ex)], dim=-1)
        att = seqops.attention(q, kv, key_ids, self.num_heads, p, tr)
        att = linear(att, mha.out_proj.weight, mha.out_proj.bias, exact=ex)
        out = seqops.layer_norm(inputs_query + seqops.dropout(att, p, tr), self.norm_layers[0].weight, self.norm_layers[0].bias)
        ff = linear(seqops.dropout(linear(out, self.linear_layers[0].weight, self.linear_layers[0].bias, relu=True, exact=ex), p, tr),
                    self.linear_layers[1].weight, self.linear_layers[1].bias, exact=ex)
        return seqops.layer_norm(out + seqops.dropout(ff, p, tr), self.norm_layers[1].weight, self.norm_layers[1].bias)


class _StandaloneAttnFn(torch.autograd.Function):
    """MultiHeadAttention.forward on materialised inputs, on the same gather-fused kernels the backbones use: the neighbor rows
    are handed over as tables with identity indices (node rows | [edge | time-feature] rows, dt_dim = 0), the query side runs on the
    HIP GEMMs.  Differentiable w.r.t. all five inputs and the seven parameters (the golden `attention.npz` pins all of them)."""

    @staticmethod
    def forward(ctx, node, ntime, nbr, nbrt, nbre, Wq, Wk, Wv, ln_g, ln_b, Wr, br, masks, heads, p_drop, training):
        n, k, dn = nbr.shape
        de, T = nbre.shape[2], nbrt.shape[2]
        dq, dk, dev = dn + T, dn + de + T, node.device
        hd = dq // heads
        x = torch.cat([node, ntime.reshape(n, T)], dim=1).contiguous()                    # query = residual (modules.py:183)
        q = torch.empty((n, dq), device=dev)
        ops.gemm(x, Wq, q, tb=True)
        u = torch.empty((n, heads, dk), device=dev)                                       # u_h = Wk_h^T q_h
        ops.gemm_batched(q[:, :hd], Wk[:hd], u[:, 0, :], heads, hd, hd * dk, dk)
        feat = nbr.reshape(n * k, dn).contiguous()
        et = torch.cat([nbre, nbrt], dim=2).reshape(n * k, de + T).contiguous()           # [edge | time features] as the "edge" rows
        ident = torch.arange(n * k, dtype=torch.int32, device=dev)
        ids = torch.from_numpy(np.ascontiguousarray(masks).astype(np.int32)).to(dev).reshape(-1)
        seeds = _seeds(2) if (training and p_drop > 0) else [0, 0]
        zero = torch.zeros(n * k, device=dev)
        a = ops.AttnArgs(feat, ident, et, ident, ids, zero, None, None, k, heads, hd ** -0.5, p_drop if training else 0.0, seeds[0])
        agg, prob = ops.attn_fwd(a, u)
        ctxv = torch.empty((n, dq), device=dev)
        ops.gemm_batched(agg[:, 0, :], Wv[:hd], ctxv[:, :hd], heads, dk, hd * dk, hd, tb=True)
        res = torch.empty((n, dq), device=dev)
        ops.gemm(ctxv, Wr, res, tb=True, bias=br)
        drop = None
        if training and p_drop > 0:
            drop = (torch.rand_like(res) >= p_drop).to(torch.float32) / (1.0 - p_drop)
            res = res * drop
        y, mean, rstd = ops.add_layernorm_fwd(res, x, ln_g, ln_b)
        scores = prob if not (training and p_drop > 0) else ops.attn_dropped_scores(a, prob)
        ctx.args, ctx.dims = a, (n, k, dn, de, T, heads)
        ctx.save_for_backward(x, q, u, agg, prob, ctxv, res, drop, mean, rstd, Wq, Wk, Wv, ln_g, Wr)
        ctx.mark_non_differentiable(scores)
        return y, scores

    @staticmethod
    def backward(ctx, dy, _):
        x, q, u, agg, prob, ctxv, res, drop, mean, rstd, Wq, Wk, Wv, ln_g, Wr = ctx.saved_tensors
        n, k, dn, de, T, heads = ctx.dims
        dq, dk, dev = dn + T, dn + de + T, dy.device
        hd = dq // heads
        dsum, dg, dbeta = ops.add_layernorm_bwd(res, x, dy.contiguous(), ln_g, mean, rstd)
        dres = dsum * drop if drop is not None else dsum
        dWr = torch.empty_like(Wr)
        ops.gemm(dres, ctxv, dWr, ta=True)
        dbr = ops.colsum(dres)
        dctx = torch.empty((n, dq), device=dev)
        ops.gemm(dres, Wr, dctx)
        dagg = torch.empty((n, heads, dk), device=dev)
        dWv = torch.empty_like(Wv)
        ops.gemm_batched(dctx[:, :hd], Wv[:hd], dagg[:, 0, :], heads, hd, hd * dk, dk)
        ops.gemm_batched(dctx[:, :hd], agg[:, 0, :], dWv[:hd], heads, hd, dk, hd * dk, ta=True)
        dfeat = torch.zeros((n * k, dn), device=dev)
        det = torch.zeros((n * k, de + T), device=dev)
        du, _, _ = ops.attn_bwd(ctx.args, u, agg, prob, dagg, dfeat, -1, dedge=det)
        dqq = torch.empty((n, dq), device=dev)
        dWk = torch.empty_like(Wk)
        ops.gemm_batched(du[:, 0, :], Wk[:hd], dqq[:, :hd], heads, dk, hd * dk, hd, tb=True)
        ops.gemm_batched(q[:, :hd], du[:, 0, :], dWk[:hd], heads, hd, dk, hd * dk, ta=True)
        dWq = torch.empty_like(Wq)
        ops.gemm(dqq, x, dWq, ta=True)
        dx = torch.empty((n, dq), device=dev)
        ops.gemm(dqq, Wq, dx)
        dx += dsum
        det = det.reshape(n, k, de + T)
        return (dx[:, :dn].contiguous(), dx[:, dn:].reshape(n, 1, T), dfeat.reshape(n, k, dn), det[:, :, de:].contiguous(),
                det[:, :, :de].contiguous(), dWq, dWk, dWv, dg, dbeta, dWr, dbr, None, None, None, None)


def _seeds(n):
    from ..engine import _next_seeds
    return _next_seeds(n)


class MultiHeadAttention(nn.Module):
    """Parameter holder with the reference's names/shapes (models/modules.py:126-165).  The backbones run it through the
    gather-fused kernels (flid_amd/engine.py); forward() on materialised inputs uses the same kernels with identity
    indices."""

    def __init__(self, node_feat_dim: int, edge_feat_dim: int, time_feat_dim: int, num_heads: int = 2, dropout: float = 0.1):
        super().__init__()
        self.node_feat_dim, self.edge_feat_dim, self.time_feat_dim, self.num_heads = node_feat_dim, edge_feat_dim, time_feat_dim, num_heads
        self.query_dim = node_feat_dim + time_feat_dim
        self.key_dim = node_feat_dim + edge_feat_dim + time_feat_dim
        assert self.query_dim % num_heads == 0, "The sum of node_feat_dim and time_feat_dim should be divided by num_heads!"
        self.head_dim = self.query_dim // num_heads
        self.query_projection = nn.Linear(self.query_dim, num_heads * self.head_dim, bias=False)
        self.key_projection = nn.Linear(self.key_dim, num_heads * self.head_dim, bias=False)
        self.value_projection = nn.Linear(self.key_dim, num_heads * self.head_dim, bias=False)
        self.scaling_factor = self.head_dim ** -0.5
        self.layer_norm = nn.LayerNorm(self.query_dim)
        self.residual_fc = nn.Linear(num_heads * self.head_dim, self.query_dim)
        self.dropout = nn.Dropout(dropout)

    def fused_params(self):
        return [self.query_projection.weight, self.key_projection.weight, self.value_projection.weight,
                self.layer_norm.weight, self.layer_norm.bias, self.residual_fc.weight, self.residual_fc.bias]

    def forward(self, node_features: torch.Tensor, node_time_features: torch.Tensor, neighbor_node_features: torch.Tensor,
                neighbor_node_time_features: torch.Tensor, neighbor_node_edge_features: torch.Tensor, neighbor_masks: np.ndarray):
        """stand-alone call on materialised inputs (reference modules.py:167-245): node (n, dn), node time (n, 1, T), neighbor node /
        time / edge features (n, k, .), masks np (n, k) with 0 = padded.  Returns (output (n, dn + T), attention scores (n, H, k))."""
        if not node_features.is_cuda:
            raise RuntimeError("flid_amd.MultiHeadAttention runs on a ROCm device only; there is no CPU path")
        f32 = lambda t: t.contiguous().float()
        return _StandaloneAttnFn.apply(f32(node_features), f32(node_time_features), f32(neighbor_node_features),
                                       f32(neighbor_node_time_features), f32(neighbor_node_edge_features), *self.fused_params(),
                                       neighbor_masks, self.num_heads, float(self.dropout.p), bool(self.training))
