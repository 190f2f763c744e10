"""Autograd wrappers (HIP forward AND backward) for the sequence-side operators of the DyGFormer path."""
import torch

from . import ops


def _seed(training, p):
    return int(torch.randint(0, 2 ** 62, (1,)).item()) if (training and p > 0) else 0


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        y, mean, rstd = ops.add_layernorm_fwd(x2, None, gamma, beta)
        ctx.save_for_backward(x2, gamma, mean, rstd)
        ctx.shape = x.shape
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, gamma, mean, rstd = ctx.saved_tensors
        dx, dg, db = ops.add_layernorm_bwd(x2, None, dy.reshape(x2.shape).contiguous(), gamma, mean, rstd)
        return dx.reshape(ctx.shape), dg, db


def layer_norm(x, gamma, beta):
    return _LayerNormFn.apply(x, gamma, beta)


class _GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.gelu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.gelu_bwd(x, dy.contiguous())


def gelu(x):
    return _GeluFn.apply(x)


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        return ops.dropout(x.contiguous(), p, seed)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout(dy.contiguous(), ctx.p, ctx.seed), None, None


def dropout(x, p, training):
    if not training or p <= 0:
        return x
    return _DropoutFn.apply(x, p, _seed(training, p))


class _SegmentMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lo, hi):
        ctx.shape, ctx.lo, ctx.hi = x.shape, lo, hi
        return ops.segment_mean_fwd(x.contiguous(), lo, hi)

    @staticmethod
    def backward(ctx, dout):
        return ops.segment_mean_bwd(dout, ctx.shape, ctx.lo, ctx.hi), None, None


def segment_mean(x, lo, hi):
    return _SegmentMeanFn.apply(x, lo, hi)


FUSED_ATTENTION = True     # the attention core of a (sequence, head) as one launch per direction (tg_seq_attn_fwd / _bwd) where it fits


def _fused_attn_ok(S, d, heads):
    hd = d // heads
    return FUSED_ATTENTION and S <= 64 and d % heads == 0 and hd % 4 == 0 and hd <= 100 and d % 4 == 0


def _self_attn_fwd(qkv, heads, p_drop, seed):
    """softmax(Q K^T / sqrt(hd)) V per (sequence, head) on the packed (B, S, 3d) in-projection; returns (out, prob, dropped prob);
    the fused launch keeps no dropped probabilities (third item = prob: its backward regenerates the mask)"""
    from ._lib import check, lib
    B, S, d3 = qkv.shape
    d = d3 // 3
    hd = d // heads
    if _fused_attn_ok(S, d, heads):
        out = torch.empty((B, S, d), device=qkv.device)
        prob = torch.empty((B, heads, S, S), device=qkv.device)
        check(lib().tg_seq_attn_fwd(ops._p(qkv), B, S, d, heads, float(p_drop), int(seed), ops._p(out), ops._p(prob), ops._stream()),
              "tg_seq_attn_fwd")
        return out, prob, prob
    q00, k00, v00 = qkv[0, :, 0:hd], qkv[0, :, d:d + hd], qkv[0, :, 2 * d:2 * d + hd]
    sq = (S * d3, hd)
    scores = torch.empty((B, heads, S, S), device=qkv.device)
    ops.gemm_batched2(q00, k00, scores[0, 0], B, heads, sq, sq, (heads * S * S, S * S), tb=True, alpha=hd ** -0.5)
    prob = ops.softmax_fwd(scores)
    pd = ops.dropout(prob, p_drop, seed) if p_drop > 0 else prob
    out = torch.empty((B, S, d), device=qkv.device)
    ops.gemm_batched2(pd[0, 0], v00, out[0, :, 0:hd], B, heads, (heads * S * S, S * S), sq, (S * d, hd))
    return out, prob, pd


def _self_attn_bwd(qkv, prob, pd, dout, heads, p_drop, seed):
    from ._lib import check, lib
    B, S, d3 = qkv.shape
    d = d3 // 3
    hd = d // heads
    if _fused_attn_ok(S, d, heads) and pd.data_ptr() == prob.data_ptr():
        dqkv = torch.empty_like(qkv)
        check(lib().tg_seq_attn_bwd(ops._p(qkv), ops._p(prob), ops._p(dout), B, S, d, heads, float(p_drop), int(seed), ops._p(dqkv),
                                    ops._stream()), "tg_seq_attn_bwd")
        return dqkv
    sq, sp, so = (S * d3, hd), (heads * S * S, S * S), (S * d, hd)
    q00, k00, v00 = qkv[0, :, 0:hd], qkv[0, :, d:d + hd], qkv[0, :, 2 * d:2 * d + hd]
    dqkv = torch.empty_like(qkv)
    dpd = torch.empty_like(prob)
    ops.gemm_batched2(dout[0, :, 0:hd], v00, dpd[0, 0], B, heads, so, sq, sp, tb=True)                  # dP = dO V^T
    ops.gemm_batched2(pd[0, 0], dout[0, :, 0:hd], dqkv[0, :, 2 * d:2 * d + hd], B, heads, sp, so, sq, ta=True)   # dV = P^T dO
    dp = ops.dropout(dpd, p_drop, seed) if p_drop > 0 else dpd
    ds = ops.softmax_bwd(prob, dp)
    scale = hd ** -0.5
    ops.gemm_batched2(ds[0, 0], k00, dqkv[0, :, 0:hd], B, heads, sp, sq, sq, alpha=scale)               # dQ = dS K
    ops.gemm_batched2(ds[0, 0], q00, dqkv[0, :, d:d + hd], B, heads, sp, sq, sq, ta=True, alpha=scale)  # dK = dS^T Q
    return dqkv


class _SelfAttentionFn(torch.autograd.Function):
    """softmax(Q K^T / sqrt(hd)) V per (sequence, head) on the packed (B, S, 3d) in-projection, as nn.MultiheadAttention does
    (models/DyGFormer.py:454; no masks).  All five products run on the two-level batched MFMA GEMM."""

    @staticmethod
    def forward(ctx, qkv, heads, p_drop, seed):
        qkv = qkv.contiguous()
        out, prob, pd = _self_attn_fwd(qkv, heads, p_drop, seed)
        ctx.save_for_backward(qkv, prob, pd)
        ctx.cfg = (heads, p_drop, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, prob, pd = ctx.saved_tensors
        heads, p_drop, seed = ctx.cfg
        return _self_attn_bwd(qkv, prob, pd, dout.contiguous(), heads, p_drop, seed), None, None, None


def self_attention(qkv, heads, p_drop, training):
    p = p_drop if training else 0.0
    return _SelfAttentionFn.apply(qkv, heads, p, _seed(training, p))


class _AttentionFn(torch.autograd.Function):
    """nn.MultiheadAttention's core with a key padding mask and separate query / key-value sequences (the reference's TCL blocks,
    models/modules.py:297-307): softmax(Q K^T / sqrt(hd) with masked keys) V per (sequence, head).  q: (B, Sq, d); kv: (B, Sk, 2 d)
    = [keys | values] as the packed in-projection leaves them; key_ids: (B, Sk) int32, id 0 = padding (None: no mask)."""

    @staticmethod
    def forward(ctx, q, kv, key_ids, heads, p_drop, seed):
        B, Sq, d = q.shape
        Sk = kv.shape[1]
        hd = d // heads
        q, kv = q.contiguous(), kv.contiguous()
        sq, sk, sp, so = (Sq * d, hd), (Sk * 2 * d, hd), (heads * Sq * Sk, Sq * Sk), (Sq * d, hd)
        scores = torch.empty((B, heads, Sq, Sk), device=q.device)
        ops.gemm_batched2(q[0, :, 0:hd], kv[0, :, 0:hd], scores[0, 0], B, heads, sq, sk, sp, tb=True, alpha=hd ** -0.5)
        prob = ops.softmax_fwd(scores) if key_ids is None else ops.softmax_keymask_fwd(scores, key_ids, heads * Sq)
        pd = ops.dropout(prob, p_drop, seed) if p_drop > 0 else prob
        out = torch.empty((B, Sq, d), device=q.device)
        ops.gemm_batched2(pd[0, 0], kv[0, :, d:d + hd], out[0, :, 0:hd], B, heads, sp, sk, so)
        ctx.save_for_backward(q, kv, prob, pd)
        ctx.cfg = (heads, p_drop, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv, prob, pd = ctx.saved_tensors
        heads, p_drop, seed = ctx.cfg
        B, Sq, d = q.shape
        Sk = kv.shape[1]
        hd = d // heads
        dout = dout.contiguous()
        sq, sk, sp, so = (Sq * d, hd), (Sk * 2 * d, hd), (heads * Sq * Sk, Sq * Sk), (Sq * d, hd)
        dq, dkv, dpd = torch.empty_like(q), torch.empty_like(kv), torch.empty_like(prob)
        ops.gemm_batched2(dout[0, :, 0:hd], kv[0, :, d:d + hd], dpd[0, 0], B, heads, so, sk, sp, tb=True)          # dP = dO V^T
        ops.gemm_batched2(pd[0, 0], dout[0, :, 0:hd], dkv[0, :, d:d + hd], B, heads, sp, so, sk, ta=True)           # dV = P^T dO
        dp = ops.dropout(dpd, p_drop, seed) if p_drop > 0 else dpd
        ds = ops.softmax_bwd(prob, dp)
        scale = hd ** -0.5
        ops.gemm_batched2(ds[0, 0], kv[0, :, 0:hd], dq[0, :, 0:hd], B, heads, sp, sk, sq, alpha=scale)              # dQ = dS K
        ops.gemm_batched2(ds[0, 0], q[0, :, 0:hd], dkv[0, :, 0:hd], B, heads, sp, sq, sk, ta=True, alpha=scale)     # dK = dS^T Q
        return dq, dkv, None, None, None, None


def attention(q, kv, key_ids, heads, p_drop, training):
    p = p_drop if training else 0.0
    return _AttentionFn.apply(q, kv, key_ids, heads, p, _seed(training, p))


class _MaskedTimeEncodeFn(torch.autograd.Function):
    """cos(dt w + b), zero where the slot is padding (models/DyGFormer.py:263-266)"""

    @staticmethod
    def forward(ctx, dt, mask_ids, w, b):
        out = ops.time_encode_masked(dt, mask_ids, w.reshape(-1), b)
        ctx.save_for_backward(dt, mask_ids, w, b)
        return out

    @staticmethod
    def backward(ctx, g):
        dt, mask_ids, w, b = ctx.saved_tensors
        dw, db = ops.time_encode_bwd(dt, mask_ids.contiguous().reshape(-1), w.reshape(-1), b, g)
        return None, None, dw.reshape(w.shape), db


def masked_time_encode(dt, mask_ids, w, b):
    return _MaskedTimeEncodeFn.apply(dt, mask_ids, w, b)


class _EncoderBlockFn(torch.autograd.Function):
    """One pre-LN transformer block of DyGFormer (models/DyGFormer.py:418-461) as ONE autograd node with a hand-written backward:
        y1 = LN1(x); a = out_proj(self_attention(in_proj(y1))); o1 = x + drop(a); y2 = LN2(o1); out = o1 + drop(W2 drop(gelu(W1 y2)))
    The op-by-op form was ~17 autograd nodes per block (each with its Python and its zero fills / gradient-accumulation adds) and took
    the four weight gradients and four bias sums of a block as eight launches; here they are one grouped launch (tg_wgrad_group)
    into one zero fill."""

    @staticmethod
    def forward(ctx, x, g1, b1, Wqkv, bqkv, Wo, bo, g2, b2, W1, bf1, W2, bf2, heads, p, seeds):
        B, S, d = x.shape
        n = B * S
        x2 = x.reshape(n, d).contiguous()
        dev = x.device
        y1, m1, r1 = ops.add_layernorm_fwd(x2, None, g1, b1)
        qkv = torch.empty((n, 3 * d), device=dev)
        ops.gemm(y1, Wqkv, qkv, tb=True, bias=bqkv)
        qkv3 = qkv.view(B, S, 3 * d)
        att, prob, pd = _self_attn_fwd(qkv3, heads, p, seeds[0])
        att2 = att.view(n, d)
        ao = torch.empty((n, d), device=dev)
        ops.gemm(att2, Wo, ao, tb=True, bias=bo)
        o1 = ops.dropout_add(ao, x2, p, seeds[1]) if p > 0 else x2 + ao
        y2, m2, r2 = ops.add_layernorm_fwd(o1, None, g2, b2)
        h = torch.empty((n, W1.shape[0]), device=dev)
        ops.gemm(y2, W1, h, tb=True, bias=bf1)
        hgd = ops.gelu_dropout_fwd(h, p, seeds[2]) if p > 0 else ops.gelu_fwd(h)
        f = torch.empty((n, d), device=dev)
        ops.gemm(hgd, W2, f, tb=True, bias=bf2)
        out = ops.dropout_add(f, o1, p, seeds[3]) if p > 0 else o1 + f
        ctx.save_for_backward(x2, g1, m1, r1, y1, qkv, prob, pd, att2, Wqkv, Wo, o1, g2, m2, r2, y2, h, hgd, W1, W2)
        ctx.cfg = (B, S, d, heads, p, seeds)
        return out.view(B, S, d)

    @staticmethod
    def backward(ctx, dout):
        from ._lib import TgShapeNotCovered
        x2, g1, m1, r1, y1, qkv, prob, pd, att2, Wqkv, Wo, o1, g2, m2, r2, y2, h, hgd, W1, W2 = ctx.saved_tensors
        B, S, d, heads, p, seeds = ctx.cfg
        n = B * S
        dev = dout.device
        do = dout.reshape(n, d).contiguous()
        d_f = ops.dropout(do, p, seeds[3]) if p > 0 else do
        d_hgd = torch.empty_like(h)
        ops.gemm(d_f, W2, d_hgd)
        d_h = ops.gelu_dropout_bwd(h, d_hgd, p, seeds[2]) if p > 0 else ops.gelu_bwd(h, d_hgd)
        d_y2 = torch.empty((n, d), device=dev)
        ops.gemm(d_h, W1, d_y2)
        d_o1b, dg2, db2 = ops.add_layernorm_bwd(o1, None, d_y2, g2, m2, r2)
        d_o1 = do + d_o1b
        d_ao = ops.dropout(d_o1, p, seeds[1]) if p > 0 else d_o1
        d_att = torch.empty((n, d), device=dev)
        ops.gemm(d_ao, Wo, d_att)
        dqkv = _self_attn_bwd(qkv.view(B, S, 3 * d), prob, pd, d_att.view(B, S, d), heads, p, seeds[0]).view(n, 3 * d)
        d_y1 = torch.empty((n, d), device=dev)
        ops.gemm(dqkv, Wqkv, d_y1)
        d_xb, dg1, db1 = ops.add_layernorm_bwd(x2, None, d_y1, g1, m1, r1)
        dx = d_o1 + d_xb
        # the four weight gradients and bias sums: one zero fill, one grouped launch (shapes it does not cover: a product + a column sum each)
        jobs = [(d_f, hgd, W2), (d_h, y2, W1), (d_ao, att2, Wo), (dqkv, y1, Wqkv)]
        sizes = [w.numel() + w.shape[0] for _, _, w in jobs]
        zb = torch.zeros(sum(sizes), device=dev)
        grads, off = [], 0
        for (_, _, w), sz in zip(jobs, sizes):
            grads.append((zb[off:off + w.numel()].view(w.shape), zb[off + w.numel():off + sz]))
            off += sz
        ok = all(w.shape[0] % 4 == 0 and w.shape[1] % 4 == 0 for _, _, w in jobs) and n >= 256
        if ok:
            try:
                ops.wgrad_group([(a, b_, gw, gb) for (a, b_, _), (gw, gb) in zip(jobs, grads)])
            except TgShapeNotCovered:
                ok = False
                zb.zero_()
        if not ok:
            for (a, b_, _), (gw, gb) in zip(jobs, grads):
                ops.gemm(a, b_, gw, ta=True)
                ops.colsum(a, out=gb)
        (dW2, dbf2), (dW1, dbf1), (dWo, dbo), (dWqkv, dbqkv) = grads
        return (dx.view(B, S, d), dg1, db1, dWqkv, dbqkv, dWo, dbo, dg2, db2, dW1, dbf1, dW2, dbf2, None, None, None)


def encoder_block(x, ln1, in_w, in_b, out_w, out_b, ln2, fc1, fc2, heads, p, training):
    p = p if training else 0.0
    seeds = tuple(int(v) for v in torch.randint(0, 2 ** 62, (4,)).tolist()) if p > 0 else (0, 0, 0, 0)
    return _EncoderBlockFn.apply(x, ln1.weight, ln1.bias, in_w, in_b, out_w, out_b, ln2.weight, ln2.bias, fc1.weight, fc1.bias,
                                 fc2.weight, fc2.bias, heads, p, seeds)


class _PatchProjFn(torch.autograd.Function):
    """The four channel projections of DyGFormer at patch size 1 (models/DyGFormer.py:148-157: node / edge / time / co-occurrence
    features of every sequence position -> 4 x C channels) as ONE product against the block-diagonal weight: X = [node row | edge row |
    time encoding | co-occurrence encoding | pad] (n, Kp) is assembled once (the gathers write straight into it), Y = X Wbd^T + b lands
    in the (B, S, 4 C) layout the transformer reads -- no torch.stack copy -- and the backward is one grouped weight-gradient launch
    (the diagonal blocks are the four gradients) + the two input gradients that exist (time encoder, co-occurrence encoder).  Eight
    products with 50-column outputs (half of every 128 x 96 tile idle), eight weight / bias gradient launches and two stack copies
    before."""

    @staticmethod
    def forward(ctx, X, tf, cf, Wn, bn, We, be, Wt, bt, Wc, bc):
        n, Kp = X.shape
        dn, de, T, Cc = Wn.shape[1], We.shape[1], Wt.shape[1], Wc.shape[1]
        C = Wn.shape[0]
        o_e, o_t, o_c = dn, dn + de, dn + de + T
        X[:, o_t:o_t + T] = tf.reshape(n, T)
        X[:, o_c:o_c + Cc] = cf.reshape(n, Cc)
        Wbd = torch.zeros((4 * C, Kp), device=X.device)
        Wbd[0:C, 0:dn] = Wn
        Wbd[C:2 * C, o_e:o_e + de] = We
        Wbd[2 * C:3 * C, o_t:o_t + T] = Wt
        Wbd[3 * C:, o_c:o_c + Cc] = Wc
        bias = torch.cat([bn, be, bt, bc])
        Y = torch.empty((n, 4 * C), device=X.device)
        ops.gemm(X, Wbd, Y, tb=True, bias=bias)
        ctx.save_for_backward(X, Wt, Wc)
        ctx.dims = (dn, de, T, Cc, C, tf.shape, cf.shape)
        return Y

    @staticmethod
    def backward(ctx, dY):
        from ._lib import TgShapeNotCovered
        X, Wt, Wc = ctx.saved_tensors
        dn, de, T, Cc, C, tf_shape, cf_shape = ctx.dims
        n, Kp = X.shape
        o_e, o_t, o_c = dn, dn + de, dn + de + T
        dY = dY.contiguous()
        zb = torch.zeros(4 * C * Kp + 4 * C, device=X.device)
        dW, db = zb[:4 * C * Kp].view(4 * C, Kp), zb[4 * C * Kp:]
        done = False
        if n >= 256 and (4 * C) % 4 == 0 and Kp % 4 == 0:
            try:
                ops.wgrad_group([(dY, X, dW, db)])
                done = True
            except TgShapeNotCovered:
                zb.zero_()
        if not done:
            ops.gemm(dY, X, dW, ta=True)
            ops.colsum(dY, out=db)
        d_tf = torch.empty((n, T), device=X.device)
        ops.gemm(dY[:, 2 * C:3 * C], Wt, d_tf)
        d_cf = torch.empty((n, Cc), device=X.device)
        ops.gemm(dY[:, 3 * C:], Wc, d_cf)
        return (None, d_tf.view(tf_shape), d_cf.view(cf_shape),
                dW[0:C, 0:dn].contiguous(), db[0:C].clone(), dW[C:2 * C, o_e:o_e + de].contiguous(), db[C:2 * C].clone(),
                dW[2 * C:3 * C, o_t:o_t + T].contiguous(), db[2 * C:3 * C].clone(), dW[3 * C:, o_c:o_c + Cc].contiguous(), db[3 * C:].clone())


def patch_projection(X, tf, cf, proj):
    n, e, t, c = proj['node'], proj['edge'], proj['time'], proj['neighbor_co_occurrence']
    return _PatchProjFn.apply(X, tf, cf, n.weight, n.bias, e.weight, e.bias, t.weight, t.bias, c.weight, c.bias)
