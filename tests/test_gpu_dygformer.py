"""GPU parity: flid_amd.models.DyGFormer against the reference's golden vectors (padded sequences, co-occurrence counts,
the FLiD-specific edge_id-1 gather, patch sizes 1 and 2, embeddings and all parameter gradients) and against the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden, assert_grads_match
from oracle import flid_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


class _Data:
    def __init__(self, g):
        self.src_node_ids, self.dst_node_ids, self.edge_ids, self.node_interact_times = g["src"], g["dst"], g["eid"], g["t"]


def _model(g, dropout=0.0):
    from flid_amd.models.DyGFormer import DyGFormer
    from flid_amd.utils.utils import get_neighbor_sampler
    dn, de, dt, c, patch, layers, heads, max_len = [int(v) for v in g["dims"]]
    sampler = get_neighbor_sampler(_Data(g), "recent", seed=0)
    m = DyGFormer(g["node_feat"], g["edge_feat"], sampler, time_feat_dim=dt, channel_embedding_dim=c, patch_size=patch,
                  num_layers=layers, num_heads=heads, dropout=dropout, max_input_sequence_length=max_len, device="cuda:0")
    assert sorted(m.state_dict().keys()) == list(g["keys"])
    m.load_state_dict(O.seeded_like(O.dyg_shapes(dn, de, dt, c, patch, layers), int(g["seed"]), float(g["scale"])))
    return m.to("cuda:0")


@pytest.mark.parametrize("name", ["dyg_p1", "dyg_p2"])
def test_dygformer_matches_reference_golden(name):
    from flid_amd import ops
    g = load_golden(name)
    m = _model(g).train()
    dev = torch.device("cuda:0")
    # device-built sequences and counts vs what the reference's pad_sequences / count_nodes_appearances produced
    t_dev = torch.from_numpy(g["bt"]).to(dev)
    sn, se, st, sl = m._windows(torch.from_numpy(g["bs"].astype(np.int32)).to(dev), t_dev)
    dn_, de_, dt_, dl = m._windows(torch.from_numpy(g["bd"].astype(np.int32)).to(dev), t_dev)
    ws, wd = g["pn"].shape[1], g["qn"].shape[1]
    assert np.array_equal(sn.cpu().numpy()[:, :ws], g["pn"]) and np.array_equal(se.cpu().numpy()[:, :ws], g["pe"])
    assert np.array_equal(st.cpu().numpy()[:, :ws], g["pt"]) and np.array_equal(dn_.cpu().numpy()[:, :wd], g["qn"])
    sc, dc = ops.cooccurrence(sn[:, :ws].contiguous(), dn_[:, :wd].contiguous())
    assert np.array_equal(sc.cpu().numpy(), g["sc"]) and np.array_equal(dc.cpu().numpy(), g["dc"])
    s, d = m.compute_src_dst_node_temporal_embeddings(src_node_ids=g["bs"], dst_node_ids=g["bd"], node_interact_times=g["bt"])
    np.testing.assert_allclose(s.detach().cpu().numpy(), g["s_emb"], atol=TOL)
    np.testing.assert_allclose(d.detach().cpu().numpy(), g["d_emb"], atol=TOL)
    r = torch.from_numpy(g["r"]).cuda()
    ((s * r[0]).sum() + (d * r[1]).sum()).backward()
    assert_grads_match(g, {k_: v.grad.cpu().numpy() for k_, v in m.named_parameters()}, atol=1e-4, rtol=1e-3)


def test_dygformer_reddit_shape_against_oracle():
    """config-4 shape (172/172/100, C=50, P=1, L=2, H=2, max_len 32) on a reduced Reddit-like graph, 64 edges vs the oracle"""
    from flid_amd.synth import reddit_like
    from flid_amd.models.DyGFormer import DyGFormer
    from flid_amd.utils.utils import get_neighbor_sampler
    data = reddit_like(num_edges=20000, num_users=1500, num_items=200, seed=4)
    sampler = get_neighbor_sampler(data, "recent", seed=0)
    m = DyGFormer(data.node_raw_features, data.edge_raw_features, sampler, 100, 50, 1, 2, 2, 0.0, 32, "cuda:0").to("cuda:0").eval()
    p = O.seeded_like(O.dyg_shapes(172, 172, 100, 50, 1, 2), 91, 0.04)
    p["time_encoder.w.bias"].zero_()
    m.load_state_dict(p)
    adj = O.build_adjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    orc = O.DyGFormerOracle(torch.from_numpy(data.node_raw_features), torch.from_numpy(data.edge_raw_features), adj, p, 50, 1, 2, 2, 32)
    sl = slice(15000, 15064)
    args = (data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl])
    with torch.no_grad():
        s, d = m.compute_src_dst_node_temporal_embeddings(*args)
        os_, od_ = orc.src_dst(*args)
    np.testing.assert_allclose(s.cpu().numpy(), os_.numpy(), atol=TOL)
    np.testing.assert_allclose(d.cpu().numpy(), od_.numpy(), atol=TOL)
    # train-mode dropout runs and is stochastic
    m.train()
    m.dropout = 0.1
    for blk in m.transformers:
        blk.p = 0.1
    a, _ = m.compute_src_dst_node_temporal_embeddings(*args)
    b, _ = m.compute_src_dst_node_temporal_embeddings(*args)
    assert torch.isfinite(a).all() and not torch.equal(a, b)


def test_sequence_ops_vs_torch():
    from flid_amd import seqops
    torch.manual_seed(0)
    dev = "cuda:0"
    x = torch.randn(37, 24, device=dev, requires_grad=True)
    ref = torch.nn.functional.gelu(x.double())
    y = seqops.gelu(x)
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.detach().cpu().numpy(), atol=1e-6)
    g = torch.randn_like(y)
    y.backward(g)
    (gx,) = torch.autograd.grad(ref, x, g.double(), retain_graph=False, allow_unused=True) if False else (None,)
    xr = x.detach().double().requires_grad_(True)
    torch.nn.functional.gelu(xr).backward(g.double())
    np.testing.assert_allclose(x.grad.cpu().numpy(), xr.grad.cpu().numpy(), atol=1e-5)
    # self-attention vs nn.MultiheadAttention math
    B, S, d, H = 5, 11, 24, 2
    qkv = torch.randn(B, S, 3 * d, device=dev, requires_grad=True)
    out = seqops.self_attention(qkv, H, 0.0, False)
    q, k, v = qkv.detach().double().split(d, dim=2)
    sh = lambda t: t.reshape(B, S, H, d // H).permute(0, 2, 1, 3)
    qd = qkv.detach().double().requires_grad_(True)
    q, k, v = qd.split(d, dim=2)
    a = torch.softmax(sh(q) @ sh(k).transpose(2, 3) * (d // H) ** -0.5, dim=-1)
    ref = (a @ sh(v)).permute(0, 2, 1, 3).reshape(B, S, d)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), atol=1e-5)
    go = torch.randn_like(out)
    out.backward(go)
    ref.backward(go.double())
    np.testing.assert_allclose(qkv.grad.cpu().numpy(), qd.grad.cpu().numpy(), atol=1e-4)


def test_fused_encoder_block_equals_op_by_op():
    """TransformerEncoder as one autograd node with a hand-written backward (seqops.encoder_block: grouped weight gradients, one zero
    fill) against the op-by-op autograd form, values and every gradient; with dropout: finite and reproducible under one torch seed"""
    from flid_amd.models.DyGFormer import TransformerEncoder
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    blk = TransformerEncoder(200, 2, dropout=0.0).to(dev).train()
    x0 = torch.randn(300, 12, 200, device=dev)
    r = torch.randn(300, 12, 200, device=dev)
    res = []
    try:
        for fused in (True, False):
            TransformerEncoder.FUSED = fused
            for q in blk.parameters():
                q.grad = None
            x = x0.clone().requires_grad_(True)
            y = blk(x)
            (y * r).sum().backward()
            res.append((y.detach().clone(), x.grad.clone(), {n: q.grad.clone() for n, q in blk.named_parameters()}))
    finally:
        TransformerEncoder.FUSED = True
    assert float((res[0][0] - res[1][0]).abs().max()) <= 1e-5 * float(res[1][0].abs().max())
    assert float((res[0][1] - res[1][1]).abs().max()) <= 1e-4 * float(res[1][1].abs().max())
    for n in res[1][2]:
        a, b = res[0][2][n], res[1][2][n]
        assert float((a - b).abs().max()) <= 1e-4 * max(1e-6, float(b.abs().max())), n
    blk.p = 0.2
    outs = []
    for _ in range(2):
        torch.manual_seed(9)
        x = x0.clone().requires_grad_(True)
        y = blk(x)
        (y * r).sum().backward()
        outs.append((y.detach().clone(), x.grad.clone()))
    assert torch.isfinite(outs[0][0]).all() and torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float((outs[0][0] - res[0][0]).abs().max()) > 1e-3          # the masks are on


def test_fused_elementwise_passes_equal_their_compositions():
    """tg_gelu_dropout_fwd / _bwd and tg_dropout_add against gelu, dropout and add as separate launches (same hash masks)"""
    from flid_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    x, dy, res = (torch.randn(1000, 333, device=dev) for _ in range(3))
    p, seed = 0.25, 123456789
    a = ops.gelu_dropout_fwd(x, p, seed)
    b = ops.dropout(ops.gelu_fwd(x), p, seed)
    assert float((a - b).abs().max()) <= 1e-6 and 0.2 < float((a == 0).float().mean()) < 0.3
    a = ops.gelu_dropout_bwd(x, dy, p, seed)
    b = ops.gelu_bwd(x, ops.dropout(dy, p, seed))
    assert float((a - b).abs().max()) <= 1e-6
    a = ops.dropout_add(x, res, p, seed)
    b = res + ops.dropout(x, p, seed)
    assert float((a - b).abs().max()) <= 1e-6


def test_block_diagonal_patch_projection_equals_per_channel_products():
    """DyGFormer.FUSED_PROJECTION (patch size 1: node / edge / time / co-occurrence projections of both sides as one product against
    the block-diagonal weight, seqops.patch_projection) against the per-channel, per-side products: embeddings and every gradient"""
    from flid_amd.models.DyGFormer import DyGFormer
    g = load_golden("dyg_p1")
    r = torch.from_numpy(g["r"]).cuda()
    res = []
    try:
        for fused in (True, False):
            DyGFormer.FUSED_PROJECTION = fused
            m = _model(g).train()
            s, d = m.compute_src_dst_node_temporal_embeddings(g["bs"], g["bd"], g["bt"])
            ((s * r[0]).sum() + (d * r[1]).sum()).backward()
            res.append((torch.cat([s, d]).detach().clone(), {k_: v.grad.clone() for k_, v in m.named_parameters()}))
    finally:
        DyGFormer.FUSED_PROJECTION = True
    assert float((res[0][0] - res[1][0]).abs().max()) <= 2e-5
    for k_ in res[1][1]:
        a, b = res[0][1][k_], res[1][1][k_]
        assert float((a - b).abs().max()) <= 1e-4 * max(1e-6, float(b.abs().max())), k_


@pytest.mark.parametrize("B,S,d,heads,p", [(7, 64, 200, 2, 0.0), (5, 37, 200, 2, 0.3), (3, 5, 16, 2, 0.0), (4, 64, 100, 1, 0.2), (9, 33, 48, 4, 0.1),
                                            (11, 32, 200, 2, 0.0), (6, 32, 200, 2, 0.25), (5, 17, 100, 1, 0.0), (4, 31, 48, 4, 0.1)])
def test_fused_self_attention_equals_batched_products(B, S, d, heads, p):
    """tg_seq_attn_fwd / _bwd (the attention core of a (sequence, head) as one launch per direction, tg_seqattn.hip) against the batched
    exact-fp32 products + softmax + dropout passes they replace (same seed, same mask), and against float64"""
    from flid_amd import seqops
    dev = torch.device("cuda:0")
    torch.manual_seed(B * S + d)
    qkv = torch.randn(B, S, 3 * d, device=dev)
    dout = torch.randn(B, S, d, device=dev)
    seed = 123456789
    res = []
    for fused in (True, False):
        seqops.FUSED_ATTENTION = fused
        try:
            out, prob, pd = seqops._self_attn_fwd(qkv, heads, p, seed)
            dqkv = seqops._self_attn_bwd(qkv, prob, pd, dout, heads, p, seed)
        finally:
            seqops.FUSED_ATTENTION = True
        res.append((out, prob, dqkv))
    for a, b_ in zip(*res):
        assert float((a - b_).abs().max()) <= 2e-5 * max(1.0, float(b_.abs().max())), float((a - b_).abs().max())
    if p == 0.0:
        hd = d // heads
        q, k, v = (qkv[..., i * d:(i + 1) * d].double().view(B, S, heads, hd).transpose(1, 2).requires_grad_(True) for i in range(3))
        pr = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1)
        o = (pr @ v).transpose(1, 2).reshape(B, S, d)
        o.backward(dout.double())
        assert float((res[0][0].double() - o).abs().max()) <= 1e-5 and float((res[0][1].double() - pr).abs().max()) <= 1e-6
        want = torch.cat([g_.grad.transpose(1, 2).reshape(B, S, d) for g_ in (q, k, v)], dim=-1)
        assert float((res[0][2].double() - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))


def _flat_grads(m, st):
    """per-tensor views of the native step's gradient block (laid out like the flat parameter)"""
    base = st.flat.data_ptr()
    out = {}
    for k_, p in m.named_parameters():
        o = (p.data_ptr() - base) // 4
        out[k_] = st.grad[o:o + p.numel()].view(p.shape).cpu().numpy()
    return out


def test_dygformer_native_step_matches_reference_golden():
    """the native step (tg_dyg_forward / tg_dyg_backward, every launch issued by the library) against the reference's golden vectors:
    embeddings and every parameter gradient"""
    g = load_golden("dyg_p1")
    m = _model(g).train()
    m.flatten_parameters()
    st = m.enable_native_step(len(g["bs"]))
    B = len(g["bs"])
    emb = st.forward(g["bs"], g["bd"], g["bt"])
    np.testing.assert_allclose(emb[:B].cpu().numpy(), g["s_emb"], atol=TOL)
    np.testing.assert_allclose(emb[B:].cpu().numpy(), g["d_emb"], atol=TOL)
    r = torch.from_numpy(g["r"]).cuda()
    st.backward(r.reshape(2 * B, -1).contiguous())
    assert_grads_match(g, _flat_grads(m, st), atol=1e-4, rtol=1e-3)
    assert m._flat_pack[0].grad.data_ptr() == st.grad.data_ptr()
    # a smaller batch through the same object; an id outside the graph; a batch larger than the object was sized for
    emb2 = st.forward(g["bs"][:4], g["bd"][:4], g["bt"][:4]).clone()
    s4, d4 = m.compute_src_dst_node_temporal_embeddings(g["bs"][:4], g["bd"][:4], g["bt"][:4])
    # (the native step's d-deep products are split-bf16 against pre-split weights; the autograd path's few-row products exact fp32)
    assert float((emb2 - torch.cat([s4, d4]).detach()).abs().max()) <= TOL
    bad = g["bs"].copy()
    bad[0] = 10 ** 6
    with pytest.raises(IndexError):
        st.forward(bad, g["bd"], g["bt"])
    with pytest.raises(Exception):
        st.forward(np.tile(g["bs"], 2), np.tile(g["bd"], 2), np.tile(g["bt"], 2))


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_dygformer_native_step_equals_autograd_step(dropout):
    """config-4 shape: the native train_step against the autograd path on the same batch, weights and seeds -- embeddings, loss, every
    gradient; then two optimizer steps each (torch.optim.Adam there, the library's update here)"""
    from flid_amd import ops
    from flid_amd.optim import FlatAdam
    from flid_amd.synth import reddit_like
    from flid_amd.models.DyGFormer import DyGFormer
    from flid_amd.utils.utils import get_neighbor_sampler
    data = reddit_like(num_edges=20000, num_users=1500, num_items=200, seed=4)
    sampler = get_neighbor_sampler(data, "recent", seed=0)
    p = O.seeded_like(O.dyg_shapes(172, 172, 100, 50, 1, 2), 91, 0.04)
    ms = []
    for _ in range(2):
        m = DyGFormer(data.node_raw_features, data.edge_raw_features, sampler, 100, 50, 1, 2, 2, dropout, 32, "cuda:0").to("cuda:0").train()
        m.load_state_dict(p)
        ms.append(m)
    ma, mn = ms
    flat = mn.flatten_parameters()
    mn.enable_native_step(200)
    B = 200
    torch.manual_seed(3)
    rw = torch.randn(2 * B, 172, device="cuda:0")
    rw_grad = rw / float(2 * B)

    def loss_fn(emb):
        return ops.weighted_sum(emb, rw, 1.0 / (2 * B)), rw_grad

    opt_a = torch.optim.Adam(ma.parameters(), lr=1e-4)
    opt_n = FlatAdam([flat], lr=1e-4)
    for it, lo in enumerate((15000, 15200)):
        sl = slice(lo, lo + B)
        args = (data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl])
        torch.manual_seed(11 + it)
        s, d = ma.compute_src_dst_node_temporal_embeddings(*args)
        ea = torch.cat([s, d])
        la = (ea * rw).sum() / (2 * B)
        opt_a.zero_grad(set_to_none=True)
        la.backward()
        torch.manual_seed(11 + it)
        opt_n.zero_grad(set_to_none=True)
        en, ln = mn.train_step(*args, loss_fn, optimizer=opt_n if it == 1 else None)
        tol_e = 5e-5 if it == 0 else 2e-3                   # (two split-bf16 kernels; step 1 runs on weights two different Adam kernels produced)
        assert float((ea.detach() - en).abs().max()) <= tol_e, (it, float((ea.detach() - en).abs().max()))
        assert abs(float(la) - float(ln)) <= 1e-5 * max(1.0, abs(float(la))) * (1 if it == 0 else 100)
        if it == 0:
            gn = _flat_grads(mn, mn._stepper)
            for k_, q in ma.named_parameters():
                ga = q.grad.cpu().numpy()
                big = max(1e-6, float(np.abs(ga).max()))
                assert float(np.abs(gn[k_] - ga).max()) <= 2e-4 * big, (k_, float(np.abs(gn[k_] - ga).max()), big)
            opt_n.step()                                    # the library's stand-alone update on the native gradient
        opt_a.step()
    for (k_, qa), (_, qn) in zip(ma.named_parameters(), mn.named_parameters()):
        # two Adam steps of lr 1e-4 each: entries whose gradient is rounding-level may differ by the whole step
        assert float((qa.detach() - qn.detach()).abs().max()) <= 4.1e-4, k_


def test_dygformer_native_step_refuses_uncovered_shapes():
    from flid_amd._lib import TgShapeNotCovered
    g = load_golden("dyg_p2")
    m = _model(g).train()
    m.flatten_parameters()
    with pytest.raises(NotImplementedError):
        m.enable_native_step(16)                             # patch size 2
    from flid_amd.synth import reddit_like
    from flid_amd.models.DyGFormer import DyGFormer
    from flid_amd.utils.utils import get_neighbor_sampler
    data = reddit_like(num_edges=2000, num_users=150, num_items=20, seed=4)
    sampler = get_neighbor_sampler(data, "recent", seed=0)
    m = DyGFormer(data.node_raw_features, data.edge_raw_features, sampler, 100, 50, 1, 2, 2, 0.1, 64, "cuda:0").to("cuda:0")
    m.flatten_parameters()
    with pytest.raises(TgShapeNotCovered):
        m.enable_native_step(16)                             # two sides of 64 positions
    with pytest.raises(RuntimeError):
        DyGFormer(data.node_raw_features, data.edge_raw_features, sampler, 100, 50, 1, 2, 2, 0.1, 32, "cuda:0").to("cuda:0").train_step(
            data.src_node_ids[:4], data.dst_node_ids[:4], data.node_interact_times[:4], None)


def test_dygformer_native_step_on_empty_and_short_histories():
    """the stream's first edges: no node has a history (both sides one position wide: the node itself), then a few edges later (ragged,
    short windows) -- native forward / backward against the autograd path"""
    from flid_amd.synth import reddit_like
    from flid_amd.models.DyGFormer import DyGFormer
    from flid_amd.utils.utils import get_neighbor_sampler
    data = reddit_like(num_edges=3000, num_users=300, num_items=40, seed=7)
    sampler = get_neighbor_sampler(data, "recent", seed=0)
    p = O.seeded_like(O.dyg_shapes(172, 172, 100, 50, 1, 2), 17, 0.05)
    ma = DyGFormer(data.node_raw_features, data.edge_raw_features, sampler, 100, 50, 1, 2, 2, 0.0, 32, "cuda:0").to("cuda:0").train()
    mn = DyGFormer(data.node_raw_features, data.edge_raw_features, sampler, 100, 50, 1, 2, 2, 0.0, 32, "cuda:0").to("cuda:0").train()
    ma.load_state_dict(p)
    mn.load_state_dict(p)
    mn.flatten_parameters()
    st = mn.enable_native_step(64)
    t0 = float(data.node_interact_times[0])
    cases = [(data.src_node_ids[:5], data.dst_node_ids[:5], np.full(5, t0)),           # nothing strictly before the first time stamp
             (data.src_node_ids[40:104], data.dst_node_ids[40:104], data.node_interact_times[40:104]),
             (data.src_node_ids[2990:2991], data.dst_node_ids[2990:2991], data.node_interact_times[2990:2991])]   # one edge, long histories
    for src, dst, t in cases:
        B = len(src)
        s, d = ma.compute_src_dst_node_temporal_embeddings(src, dst, t)
        ea = torch.cat([s, d])
        torch.manual_seed(B)
        r = torch.randn(2 * B, 172, device="cuda:0")
        for q in ma.parameters():
            q.grad = None
        (ea * r).sum().backward()
        st.flat.grad = None                      # (the block is rewritten by every step: a backward into it without an update needs this)
        en = st.forward(src, dst, t)
        assert float((ea.detach() - en).abs().max()) <= TOL, B
        st.backward(r.contiguous())
        gn = _flat_grads(mn, st)
        for k_, q in ma.named_parameters():
            ga = q.grad.cpu().numpy()
            big = max(1e-6, float(np.abs(ga).max()))
            assert float(np.abs(gn[k_] - ga).max()) <= 5e-4 * big + 1e-6, (B, k_, float(np.abs(gn[k_] - ga).max()), big)
