"""CPU: the oracle (oracle/flid_oracle.py) against the golden vectors produced by the reference itself."""
import numpy as np
import pytest
import torch

from conftest import load_golden, assert_grads_match
from oracle import flid_oracle as O

TOL = 1e-4   # north_star: embeddings within 1e-4 fp32; sampled indices bit-exact


def _adj(g):
    return O.build_adjacency(g["src"], g["dst"], g["eid"], g["t"], int(g["num_rows"]))


def test_sampler_recent_bit_exact():
    g = load_golden("sampler")
    adj = _adj(g)
    for k in (1, 3, 20):
        a, b, c = O.sample_recent(adj, g["ids"], g["qt64"], k)
        assert np.array_equal(a, g[f"k{k}_n"]) and np.array_equal(b, g[f"k{k}_e"])
        assert c.dtype == np.float32 and np.array_equal(c, g[f"k{k}_t"])
        a2, b2, c2 = O.sample_recent(adj, a.reshape(-1), c.reshape(-1), k)       # float32 hop-2 query times
        assert np.array_equal(a2, g[f"k{k}_n2"]) and np.array_equal(b2, g[f"k{k}_e2"]) and np.array_equal(c2, g[f"k{k}_t2"])


def test_sampler_first_hop_and_random():
    g = load_golden("sampler")
    adj = _adj(g)
    la, lb, lc = O.first_hop_all(adj, g["ids"], g["qt64"])
    assert np.array_equal(np.array([len(x) for x in la]), g["fh_len"])
    assert np.array_equal(np.concatenate(la), g["fh_n"]) and np.array_equal(np.concatenate(lb), g["fh_e"])
    assert np.array_equal(np.concatenate(lc), g["fh_t"])
    for strat, tsf in (("uniform", None), ("time_interval_aware", 1e-4)):
        rng = np.random.RandomState(1)
        for call in (0, 1):
            a, b, c = O.sample_random(adj, g["ids"], g["qt64"], 5, rng, tsf)
            assert np.array_equal(a, g[f"{strat}{call}_n"]), (strat, call)
            assert np.array_equal(b, g[f"{strat}{call}_e"]) and np.array_equal(c, g[f"{strat}{call}_t"])
        a, _, _ = O.sample_random(adj, g["ids"], g["qt64"], 5, np.random.RandomState(1), tsf)
        assert np.array_equal(a, g[f"{strat}R_n"])


def test_time_encoder():
    g = load_golden("time_encoder")
    for tag in ("b0", "b1"):
        p = {"w.weight": torch.from_numpy(g[tag + "_w"]), "w.bias": torch.from_numpy(g[tag + "_b"])}
        grid = torch.from_numpy(g["grid"])
        assert np.array_equal(O.time_encode(p, "", grid).numpy(), g[tag + "_bk"])
        assert np.array_equal(O.time_encode(p, "", grid.reshape(-1, 1)).numpy(), g[tag + "_b1"])


def test_attention_forward_backward():
    g = load_golden("attention")
    dn, de, dt, heads = [int(v) for v in g["dims"]]
    dq, dk = dn + dt, dn + de + dt
    shapes = {"query_projection.weight": (dq, dq), "key_projection.weight": (dq, dk), "value_projection.weight": (dq, dk),
              "layer_norm.weight": (dq,), "layer_norm.bias": (dq,), "residual_fc.weight": (dq, dq), "residual_fc.bias": (dq,)}
    p = {k: v.requires_grad_(True) for k, v in O.seeded_like(shapes, int(g["seed"]), float(g["scale"])).items()}
    ins = {k: torch.from_numpy(g[k]).requires_grad_(True) for k in ("node", "ntime", "nbr", "nbrt", "nbre")}
    out, sc = O.temporal_attention(p, "", heads, ins["node"], ins["ntime"], ins["nbr"], ins["nbrt"], ins["nbre"], g["ids"])
    np.testing.assert_allclose(out.detach().numpy(), g["out"], atol=1e-5)
    np.testing.assert_allclose(sc.detach().numpy(), g["scores"], atol=1e-6)
    assert np.allclose(sc.detach().numpy()[2], 1.0 / g["ids"].shape[1])          # all-padded row: uniform attention
    (out * torch.from_numpy(g["r"])).sum().backward()
    for k, v in ins.items():
        np.testing.assert_allclose(v.grad.numpy(), g["gi:" + k], atol=1e-5, err_msg=k)
    assert_grads_match(g, {k: v.grad.numpy() for k, v in p.items()}, atol=1e-5)


TGAT_CASES = ["tgat_L1_K2", "tgat_L2_K2", "tgat_L2_K20", "tgat_L2_K20_full", "tgat_L1_K20_full_bias"]


@pytest.mark.parametrize("name", TGAT_CASES)
def test_tgat(name):
    g = load_golden(name)
    dn, de, dt, layers, k = [int(v) for v in g["dims"]]
    shapes = O.tgat_shapes(dn, de, dt, layers)
    assert sorted(shapes) == list(g["keys"])
    p = O.seeded_like(shapes, int(g["seed"]), float(g["scale"]))
    if not bool(g["bias_te"]):
        p["time_encoder.w.bias"].zero_()
    p = {k_: v.requires_grad_(True) for k_, v in p.items()}
    m = O.TGATOracle(torch.from_numpy(g["node_feat"]), torch.from_numpy(g["edge_feat"]), _adj(g), p, layers, 2)
    s, d = m.src_dst(g["bs"], g["bd"], g["bt"], k)
    np.testing.assert_allclose(s.detach().numpy(), g["s_emb"], atol=TOL)
    np.testing.assert_allclose(d.detach().numpy(), g["d_emb"], atol=TOL)
    r = torch.from_numpy(g["r"])
    ((s * r[0]).sum() + (d * r[1]).sum()).backward()
    assert_grads_match(g, {k_: v.grad.numpy() for k_, v in p.items()}, atol=2e-4, rtol=1e-3)


def test_tgn_sequence():
    g = load_golden("tgn_small")
    dn, de, dt, layers, k = [int(v) for v in g["dims"]]
    p = {k_: v.requires_grad_(True) for k_, v in O.seeded_like(O.tgn_shapes(dn, de, dt, layers), int(g["seed"]), float(g["scale"])).items()}
    m = O.TGNOracle(torch.from_numpy(g["node_feat"]), torch.from_numpy(g["edge_feat"]), _adj(g), p, layers, 2)
    bsz, backup = 12, None
    for b in range(7):
        sl = slice(b * bsz, (b + 1) * bsz)
        bs, bd, bt, be = g["src"][sl], g["dst"][sl], g["t"][sl], g["eid"][sl]
        ns_, nd_ = m.src_dst(bs, g[f"neg{b}"], bt, None, False, k)
        ps_, pd_ = m.src_dst(bs, bd, bt, be, True, k)
        for mine, key in ((ns_, "ns"), (nd_, "nd"), (ps_, "ps"), (pd_, "pd")):
            np.testing.assert_allclose(mine.detach().numpy(), g[f"{key}{b}"], atol=TOL, err_msg=f"{key}{b}")
        if b == 3:
            r = torch.from_numpy(g["r3"])
            sum((e * r[i]).sum() for i, e in enumerate((ns_, nd_, ps_, pd_))).backward()
            assert_grads_match(g, {k_: v.grad.numpy() for k_, v in p.items() if v.grad is not None}, atol=1e-4, rtol=1e-3)
        m.detach()
        np.testing.assert_allclose(m.memory.numpy(), g[f"mem{b}"], atol=TOL)
        assert np.array_equal(m.last_update.numpy(), g[f"lu{b}"])
        has = np.zeros(int(g["num_rows"]), dtype=bool)
        has[list(m.pending)] = True
        assert np.array_equal(has, g[f"has{b}"])
        for nid, (msg, ts) in m.pending.items():
            np.testing.assert_allclose(msg.numpy(), g[f"pm{b}"][nid], atol=TOL)
            assert ts == g[f"pt{b}"][nid]
        if b == 4:
            backup = m.backup()
    m.reload(backup)
    np.testing.assert_allclose(m.memory.numpy(), g["mem4"], atol=TOL)
    sl = slice(5 * bsz, 6 * bsz)
    with torch.no_grad():
        a, b_ = m.src_dst(g["src"][sl], g["dst"][sl], g["t"][sl], g["eid"][sl], True, k)
    np.testing.assert_allclose(a.numpy(), g["reload_ps5"], atol=TOL)
    np.testing.assert_allclose(m.memory.numpy(), g["reload_mem5"], atol=TOL)
    assert bool(g["past_assert"])
    with pytest.raises(AssertionError, match="time in the past"):
        with torch.no_grad():
            m.src_dst(g["src"][:bsz], g["dst"][:bsz], g["t"][:bsz] * 0.0 - 5.0, g["eid"][:bsz], True, k)
            m.src_dst(g["src"][:bsz], g["dst"][:bsz], g["t"][:bsz] * 0.0 - 9.0, g["eid"][:bsz], True, k)


@pytest.mark.parametrize("name", ["dyg_p1", "dyg_p2"])
def test_dygformer(name):
    g = load_golden(name)
    dn, de, dt, c, patch, layers, heads, max_len = [int(v) for v in g["dims"]]
    shapes = O.dyg_shapes(dn, de, dt, c, patch, layers)
    assert sorted(shapes) == list(g["keys"])
    p = {k_: v.requires_grad_(True) for k_, v in O.seeded_like(shapes, int(g["seed"]), float(g["scale"])).items()}
    m = O.DyGFormerOracle(torch.from_numpy(g["node_feat"]), torch.from_numpy(g["edge_feat"]), _adj(g), p,
                          c, patch, layers, heads, max_len)
    pn, pe, pt = m.sequences(g["bs"], g["bt"])
    qn, qe, qt = m.sequences(g["bd"], g["bt"])
    for mine, key in ((pn, "pn"), (pe, "pe"), (pt, "pt"), (qn, "qn"), (qe, "qe"), (qt, "qt")):
        assert np.array_equal(mine, g[key]), key
    sc, dc = O.cooccurrence_counts(pn, qn)
    assert np.array_equal(sc, g["sc"]) and np.array_equal(dc, g["dc"])
    with torch.no_grad():
        _, ef, tf = m.features(g["bt"], pn, pe, pt)
    np.testing.assert_allclose(ef.numpy(), g["ef"], atol=0)                       # the eid-1 gather quirk
    np.testing.assert_allclose(tf.numpy(), g["tf"], atol=1e-6)
    s, d = m.src_dst(g["bs"], g["bd"], g["bt"])
    np.testing.assert_allclose(s.detach().numpy(), g["s_emb"], atol=TOL)
    np.testing.assert_allclose(d.detach().numpy(), g["d_emb"], atol=TOL)
    r = torch.from_numpy(g["r"])
    ((s * r[0]).sum() + (d * r[1]).sum()).backward()
    assert_grads_match(g, {k_: v.grad.numpy() for k_, v in p.items()}, atol=1e-4, rtol=1e-3)


def test_state_dict_contract():
    g = load_golden("state_dict_keys")
    assert int(g["tgat_nparams"]) == 993432 and int(g["dyg_nparams"]) == 1027522
    tg = dict(zip(g["tgat_keys"], g["tgat_shapes"]))
    mine = O.tgat_shapes(172, 172, 100, 2)
    assert set(tg) == set(mine)
    for k, v in mine.items():
        assert ",".join(map(str, v)) == tg[k], k
    dy = dict(zip(g["dyg_keys"], g["dyg_shapes"]))
    mine = O.dyg_shapes(172, 172, 100, 50, 1, 2)
    assert set(dy) == set(mine)
    for k, v in mine.items():
        assert ",".join(map(str, v)) == dy[k], k
    tn = dict(zip(g["tgn_keys"], g["tgn_shapes"]))
    for k, v in O.tgn_shapes(172, 172, 100, 1).items():
        assert ",".join(map(str, v)) == tn[k], k


# ---- full size (B = 600, BASELINE dims): the oracle against what the reference produced on the same batches -----------------------
import fullsize  # noqa: E402


@pytest.mark.parametrize("name", ["tgat_B600_full", "tgat_B600_kinkfree"])
def test_tgat_full_size(name):
    g = load_golden(name)
    data, p, (bs, bd, bt), r = fullsize.tgat_case(g)
    p = {k_: v.requires_grad_(True) for k_, v in p.items()}
    adj = O.build_adjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    m = O.TGATOracle(torch.from_numpy(data.node_raw_features), torch.from_numpy(data.edge_raw_features), adj, p, 2, 2)
    s, d = m.src_dst(bs, bd, bt, 20)
    np.testing.assert_allclose(s.detach().numpy(), g["s_emb"], atol=TOL)
    np.testing.assert_allclose(d.detach().numpy(), g["d_emb"], atol=TOL)
    rr = torch.from_numpy(r)
    ((s * rr[0]).sum() + (d * rr[1]).sum()).backward()
    assert_grads_match(g, {k_: v.grad.numpy() for k_, v in p.items()}, atol=1e-4, rtol=1e-3, strict=bool(g["kink_free"]))


def test_tgn_full_size_sequence():
    g = load_golden("tgn_B600x3")
    data, p = fullsize.tgn_case(g)
    p = {k_: v.requires_grad_(True) for k_, v in p.items()}
    adj = O.build_adjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    m = O.TGNOracle(torch.from_numpy(data.node_raw_features), torch.from_numpy(data.edge_raw_features), adj, p, 1, 2)
    step = int(g["step"])
    for j, (bs, bd, bt, be), neg, r in fullsize.tgn_batches(g, data):
        if j is None:
            with torch.no_grad():
                m.src_dst(bs, bd, bt, be, True, 20)
            continue
        for v in p.values():
            v.grad = None
        ns_, nd_ = m.src_dst(bs, neg, bt, None, False, 20)
        ps_, pd_ = m.src_dst(bs, bd, bt, be, True, 20)
        rr = torch.from_numpy(r)
        sum((e * rr[i]).sum() for i, e in enumerate((ns_, nd_, ps_, pd_))).backward()
        assert_grads_match(fullsize.tgn_grads_view(g, j), {k_: v.grad.numpy() for k_, v in p.items() if v.grad is not None},
                           atol=1e-4, strict=True)
        m.detach()
        for mine, key in ((ns_, "ns"), (nd_, "nd"), (ps_, "ps"), (pd_, "pd")):
            np.testing.assert_allclose(mine.detach().numpy()[::step], g[f"{key}{j}"], atol=TOL, err_msg=f"{key}{j}")
        np.testing.assert_allclose(m.memory.numpy()[g[f"touched{j}"]], g[f"mem{j}"], atol=TOL)
        assert abs(m.memory.double().sum().item() - g[f"memsum{j}"][0]) < 1e-2
        assert np.array_equal(m.last_update.numpy(), g[f"lu{j}"])
        has = np.zeros(m.num_rows, dtype=bool)
        has[list(m.pending)] = True
        assert np.array_equal(has, g[f"has{j}"])
        for nid, (msg, ts) in m.pending.items():
            assert ts == g[f"pt{j}"][nid]
            assert abs(msg.double().sum().item() - g[f"pmsum{j}"][nid]) < 1e-3


def test_dygformer_full_size():
    g = load_golden("dyg_B600")
    data, p, (bs, bd, bt), r = fullsize.dyg_case(g)
    p = {k_: v.requires_grad_(True) for k_, v in p.items()}
    adj = O.build_adjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    m = O.DyGFormerOracle(torch.from_numpy(data.node_raw_features), torch.from_numpy(data.edge_raw_features), adj, p, 50, 1, 2, 2, 32)
    s, d = m.src_dst(bs, bd, bt)
    np.testing.assert_allclose(s.detach().numpy(), g["s_emb"], atol=TOL)
    np.testing.assert_allclose(d.detach().numpy(), g["d_emb"], atol=TOL)
    rr = torch.from_numpy(r)
    ((s * rr[0]).sum() + (d * rr[1]).sum()).backward()
    assert_grads_match(g, {k_: v.grad.numpy() for k_, v in p.items()}, atol=1e-4, rtol=1e-3)


# ---- the two further backbones: oracle vs vectors the reference produced -------------------------------------------
def _toy_sampler_fn(g, adj):
    """`recent`: the oracle's default; `uniform`: numpy's stream, consumed in the reference's call order (seed 3 in make_golden.py)"""
    if str(g["strategy"]) == "recent":
        return None
    rng = np.random.RandomState(3)
    return lambda ids, times, k: O.sample_random(adj, ids, times, k, rng)


@pytest.mark.parametrize("name", ["tcl_K5", "tcl_K3_uniform"])
def test_tcl(name):
    g = load_golden(name)
    dn, de, dt, layers, heads, k = [int(v) for v in g["dims"]]
    shapes = O.tcl_shapes(dn, de, dt, layers, k + 1)
    assert sorted(shapes) == list(g["keys"])
    p = {k_: v.requires_grad_(True) for k_, v in O.seeded_like(shapes, int(g["seed"]), float(g["scale"])).items()}
    adj = _adj(g)
    m = O.TCLOracle(torch.from_numpy(g["node_feat"]), torch.from_numpy(g["edge_feat"]), adj, p, layers, heads, sampler_fn=_toy_sampler_fn(g, adj))
    s, d = m.src_dst(g["bs"], g["bd"], g["bt"], k)
    np.testing.assert_allclose(s.detach().numpy(), g["s_emb"], atol=TOL)
    np.testing.assert_allclose(d.detach().numpy(), g["d_emb"], atol=TOL)
    r = torch.from_numpy(g["r"])
    ((s * r[0]).sum() + (d * r[1]).sum()).backward()
    assert_grads_match(g, {k_: v.grad.numpy() for k_, v in p.items()}, atol=1e-4, rtol=1e-3)


@pytest.mark.parametrize("name", ["mixer_K6", "mixer_K4_uniform"])
def test_graphmixer(name):
    g = load_golden(name)
    dn, dt, layers, k, gap = [int(v) for v in g["dims"]]
    shapes = O.mixer_shapes(dn, dt, k, layers)
    assert sorted(shapes) == list(g["keys"])
    p = {k_: v.requires_grad_(True) for k_, v in O.seeded_like(shapes, int(g["seed"]), float(g["scale"])).items()}
    adj = _adj(g)
    m = O.GraphMixerOracle(torch.from_numpy(g["node_feat"]), adj, p, layers, sampler_fn=_toy_sampler_fn(g, adj))
    s, d = m.src_dst(g["bs"], g["bd"], g["bt"], k, gap)
    np.testing.assert_allclose(s.detach().numpy(), g["s_emb"], atol=TOL)
    np.testing.assert_allclose(d.detach().numpy(), g["d_emb"], atol=TOL)
    r = torch.from_numpy(g["r"])
    ((s * r[0]).sum() + (d * r[1]).sum()).backward()
    assert_grads_match(g, {k_: v.grad.numpy() for k_, v in p.items() if v.grad is not None}, atol=1e-4, rtol=1e-3)


def test_tcl_full_dims():
    g = load_golden("tcl_full")
    data, p, (bs, bd, bt), r = fullsize.backbone_case(g, O.tcl_shapes(172, 172, 100, 2, 21))
    p = {k_: v.requires_grad_(True) for k_, v in p.items()}
    adj = O.build_adjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    m = O.TCLOracle(torch.from_numpy(data.node_raw_features), torch.from_numpy(data.edge_raw_features), adj, p, 2, 2)
    s, d = m.src_dst(bs, bd, bt, 20)
    np.testing.assert_allclose(s.detach().numpy(), g["s_emb"], atol=TOL)
    np.testing.assert_allclose(d.detach().numpy(), g["d_emb"], atol=TOL)
    rr = torch.from_numpy(r)
    ((s * rr[0]).sum() + (d * rr[1]).sum()).backward()
    assert_grads_match(g, {k_: v.grad.numpy() for k_, v in p.items()}, atol=1e-4, rtol=1e-3)


def test_graphmixer_full_dims():
    g = load_golden("mixer_full")
    data, p, (bs, bd, bt), r = fullsize.backbone_case(g, O.mixer_shapes(172, 100, 20, 2))
    p = {k_: v.requires_grad_(True) for k_, v in p.items()}
    adj = O.build_adjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    m = O.GraphMixerOracle(torch.from_numpy(data.node_raw_features), adj, p, 2)
    s, d = m.src_dst(bs, bd, bt, 20, 2000)
    np.testing.assert_allclose(s.detach().numpy(), g["s_emb"], atol=TOL)
    np.testing.assert_allclose(d.detach().numpy(), g["d_emb"], atol=TOL)
    rr = torch.from_numpy(r)
    ((s * rr[0]).sum() + (d * rr[1]).sum()).backward()
    assert_grads_match(g, {k_: v.grad.numpy() for k_, v in p.items() if v.grad is not None}, atol=1e-4, rtol=1e-3)
