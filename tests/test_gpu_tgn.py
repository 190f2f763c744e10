"""GPU parity: flid_amd.models.MemoryModel (TGN) against the reference's golden sequence: 7 chronological batches in the
link-prediction call order (negatives first, no state change; then positives), embeddings, memory table, last-update times,
pending last messages after every batch, GRU/attention gradients on one batch, backup -> reload, the past-time assertion."""
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


def _model(g):
    from flid_amd.models.MemoryModel import MemoryModel
    from flid_amd.utils.utils import get_neighbor_sampler
    dn, de, dt, layers, k = [int(v) for v in g["dims"]]
    sampler = get_neighbor_sampler(_Data(g), "recent", seed=0)
    m = MemoryModel(g["node_feat"], g["edge_feat"], sampler, time_feat_dim=dt, model_name="TGN", num_layers=layers, num_heads=2,
                    dropout=0.0, device="cuda:0")
    assert sorted(m.state_dict().keys()) == list(g["keys"])                 # incl. the duplicated memory_updater.memory_bank.*
    p = O.seeded_like(O.tgn_shapes(dn, de, dt, layers), int(g["seed"]), float(g["scale"]))
    sd = dict(p)
    sd["embedding_module.time_encoder.w.weight"] = p["time_encoder.w.weight"]
    sd["embedding_module.time_encoder.w.bias"] = p["time_encoder.w.bias"]
    missing = m.load_state_dict(sd, strict=False)
    assert all("memory_bank" in x for x in missing.missing_keys)
    return m.train(), p, k


def test_tgn_matches_reference_golden_sequence():
    g = load_golden("tgn_small")
    m, p, k = _model(g)
    bank = m.memory_bank
    bank.__init_memory_bank__()
    bsz, backup = 12, None
    for b in range(7):
        sl = slice(b * bsz, (b + 1) * bsz)
        bs, bd, bt, be = g["src"][sl], g["dst"][sl], g["t"][sl], g["eid"][sl]
        ns_, nd_ = m.compute_src_dst_node_temporal_embeddings(src_node_ids=bs, dst_node_ids=g[f"neg{b}"], node_interact_times=bt,
                                                              edge_ids=None, edges_are_positive=False, num_neighbors=k)
        ps_, pd_ = m.compute_src_dst_node_temporal_embeddings(src_node_ids=bs, dst_node_ids=bd, node_interact_times=bt,
                                                              edge_ids=be, edges_are_positive=True, num_neighbors=k)
        for mine, key in ((ns_, "ns"), (nd_, "nd"), (ps_, "ps"), (pd_, "pd")):
            np.testing.assert_allclose(mine.detach().cpu().numpy(), g[f"{key}{b}"], atol=TOL, err_msg=f"{key}{b}")
        if b == 3:
            r = torch.from_numpy(g["r3"]).cuda()
            m.zero_grad()
            sum((e * r[i]).sum() for i, e in enumerate((ns_, nd_, ps_, pd_))).backward()
            grads = {k_: v.grad.cpu().numpy() for k_, v in m.named_parameters() if v.grad is not None}
            assert "memory_updater.memory_updater.weight_ih" in grads
            assert_grads_match(g, grads, atol=1e-4, rtol=1e-3)
        bank.detach_memory_bank()
        np.testing.assert_allclose(bank.node_memories.detach().cpu().numpy(), g[f"mem{b}"], atol=TOL)
        assert np.array_equal(bank.node_last_updated_times.detach().cpu().numpy(), g[f"lu{b}"])
        raw = bank.node_raw_messages                      # the reference's dict-of-lists view
        has = np.zeros(int(g["num_rows"]), dtype=bool)
        has[[nid for nid, lst in raw.items() if len(lst)]] = True
        assert np.array_equal(has, g[f"has{b}"])
        for nid, lst in raw.items():
            msg, ts = lst[-1]
            np.testing.assert_allclose(msg.cpu().numpy(), g[f"pm{b}"][nid], atol=TOL)
            assert isinstance(ts, np.float64) and ts == g[f"pt{b}"][nid]
        if b == 4:
            backup = bank.backup_memory_bank()
    bank.reload_memory_bank(backup)
    np.testing.assert_allclose(bank.node_memories.detach().cpu().numpy(), g["mem4"], atol=TOL)
    sl = slice(5 * bsz, 6 * bsz)
    with torch.no_grad():
        a, b_ = m.compute_src_dst_node_temporal_embeddings(g["src"][sl], g["dst"][sl], g["t"][sl], g["eid"][sl], True, k)
    np.testing.assert_allclose(a.cpu().numpy(), g["reload_ps5"], atol=TOL)
    np.testing.assert_allclose(b_.cpu().numpy(), g["reload_pd5"], atol=TOL)
    np.testing.assert_allclose(bank.node_memories.detach().cpu().numpy(), g["reload_mem5"], atol=TOL)
    # the reference's dict form round-trips through the setter (what EarlyStopping.load_checkpoint does)
    snap = bank.node_raw_messages
    bank.node_raw_messages = {k_: [(v[0][0].clone(), v[0][1])] for k_, v in snap.items()}
    again = bank.node_raw_messages
    assert sorted(again) == sorted(snap)
    assert bool(g["past_assert"])
    with pytest.raises(AssertionError, match="time in the past"):
        with torch.no_grad():
            m.compute_src_dst_node_temporal_embeddings(g["src"][:bsz], g["dst"][:bsz], g["t"][:bsz] * 0.0 - 5.0, g["eid"][:bsz], True, k)
            m.compute_src_dst_node_temporal_embeddings(g["src"][:bsz], g["dst"][:bsz], g["t"][:bsz] * 0.0 - 9.0, g["eid"][:bsz], True, k)


def test_tgn_reddit_shape_against_oracle():
    """Reddit-shape synthetic (config 3, reduced edge count), L=1, K=20, full dims: 5 batches of 200 against the oracle"""
    from flid_amd.synth import reddit_like
    from flid_amd.models.MemoryModel import MemoryModel
    from flid_amd.utils.utils import get_neighbor_sampler
    data = reddit_like(num_edges=20000, num_users=1500, num_items=200, seed=2)
    sampler = get_neighbor_sampler(data, "recent", seed=0)
    m = MemoryModel(data.node_raw_features, data.edge_raw_features, sampler, 100, "TGN", 1, 2, 0.0, device="cuda:0").eval()
    p = O.seeded_like(O.tgn_shapes(172, 172, 100, 1), 77, 0.05)
    p["time_encoder.w.bias"].zero_()
    sd = dict(p)
    sd["embedding_module.time_encoder.w.weight"], sd["embedding_module.time_encoder.w.bias"] = p["time_encoder.w.weight"], p["time_encoder.w.bias"]
    m.load_state_dict(sd, strict=False)
    adj = O.build_adjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    orc = O.TGNOracle(torch.from_numpy(data.node_raw_features), torch.from_numpy(data.edge_raw_features), adj, p, 1, 2)
    m.memory_bank.__init_memory_bank__()
    for b in range(5):
        sl = slice(10000 + b * 200, 10000 + (b + 1) * 200)
        args = (data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl], data.edge_ids[sl], True, 20)
        with torch.no_grad():
            s, d = m.compute_src_dst_node_temporal_embeddings(*args)
            os_, od_ = orc.src_dst(*args)
        np.testing.assert_allclose(s.cpu().numpy(), os_.numpy(), atol=TOL, err_msg=f"batch {b}")
        np.testing.assert_allclose(d.cpu().numpy(), od_.numpy(), atol=TOL, err_msg=f"batch {b}")
        np.testing.assert_allclose(m.memory_bank.node_memories.cpu().numpy(), orc.memory.numpy(), atol=TOL)


def test_tgn_sharded_step_keeps_replicas_identical():
    """data-parallel form: two 'ranks' embed halves of each batch and advance the state with the whole batch; embeddings and
    state must equal the single-process run"""
    g = load_golden("tgn_small")
    ref, p, k = _model(g)
    r0, _, _ = _model(g)
    r1, _, _ = _model(g)
    for m in (ref, r0, r1):
        m.eval()
        m.memory_bank.__init_memory_bank__()
    bsz = 12
    for b in range(5):
        sl = slice(b * bsz, (b + 1) * bsz)
        args = (g["src"][sl], g["dst"][sl], g["t"][sl], g["eid"][sl])
        with torch.no_grad():
            s, d = ref.compute_src_dst_node_temporal_embeddings(*args, True, k)
            s0, d0 = r0.compute_shard_embeddings_and_advance(*args, (0, 5), True, k)
            s1, d1 = r1.compute_shard_embeddings_and_advance(*args, (5, bsz), True, k)
        np.testing.assert_allclose(torch.cat([s0, s1]).cpu().numpy(), s.cpu().numpy(), atol=1e-6)
        np.testing.assert_allclose(torch.cat([d0, d1]).cpu().numpy(), d.cpu().numpy(), atol=1e-6)
        for m in (r0, r1):
            assert torch.equal(m.memory_bank.node_memories, ref.memory_bank.node_memories)
            assert torch.equal(m.memory_bank._msg, ref.memory_bank._msg)
            assert np.array_equal(m.memory_bank._has, ref.memory_bank._has)


def test_tgn_uniform_sampling_follows_the_reference_rng_stream():
    """TGN with the `uniform` strategy: host RandomState draws (one call per embedding, sources and destinations together) feed the
    device engine; against the oracle driven by an equally seeded stream, over three state-advancing batches"""
    from flid_amd.synth import reddit_like
    from flid_amd.models.MemoryModel import MemoryModel
    from flid_amd.utils.utils import get_neighbor_sampler
    data = reddit_like(num_edges=6000, num_users=300, num_items=50, feat_dim=12, seed=8, zero_node_feat=False)
    sampler = get_neighbor_sampler(data, "uniform", seed=4)
    m = MemoryModel(data.node_raw_features, data.edge_raw_features, sampler, 8, "TGN", 1, 2, 0.0, device="cuda:0").eval()
    p = O.seeded_like(O.tgn_shapes(12, 12, 8, 1), 5, 0.2)
    p["time_encoder.w.bias"].zero_()
    sd = dict(p)
    sd["embedding_module.time_encoder.w.weight"], sd["embedding_module.time_encoder.w.bias"] = p["time_encoder.w.weight"], p["time_encoder.w.bias"]
    m.load_state_dict(sd, strict=False)
    m.set_neighbor_sampler(sampler)
    adj = O.build_adjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    orc = O.TGNOracle(torch.from_numpy(data.node_raw_features), torch.from_numpy(data.edge_raw_features), adj, p, 1, 2)
    rng = np.random.RandomState(4)
    orc.g.sampler_fn = lambda i, t, kk: O.sample_random(adj, i, t, kk, rng, None)
    m.memory_bank.__init_memory_bank__()
    for b in range(3):
        sl = slice(3000 + b * 40, 3000 + (b + 1) * 40)
        args = (data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl], data.edge_ids[sl], True, 5)
        with torch.no_grad():
            s, d = m.compute_src_dst_node_temporal_embeddings(*args)
            os_, od_ = orc.src_dst(*args)
        np.testing.assert_allclose(s.cpu().numpy(), os_.numpy(), atol=TOL, err_msg=f"batch {b}")
        np.testing.assert_allclose(d.cpu().numpy(), od_.numpy(), atol=TOL, err_msg=f"batch {b}")


def test_tgn_lazy_memory_equals_full_table_update():
    """MemoryModel.LAZY (GRU only on the memory rows a call touches, compact base table, device-side persist + last-message
    scatter) against the full-table form (every pending node updated before every call, as the reference does): embeddings,
    gradients and the whole state after each of 6 neg-then-pos batches"""
    from flid_amd.models.MemoryModel import MemoryModel
    g = load_golden("tgn_small")
    models = []
    for lazy in (True, False):
        m, p, k = _model(g)
        m.LAZY = lazy
        m.memory_bank.__init_memory_bank__()
        models.append(m)
    bsz = 12
    for b in range(6):
        sl = slice(b * bsz, (b + 1) * bsz)
        outs = []
        for m in models:
            m.zero_grad(set_to_none=True)
            ns_, nd_ = m.compute_src_dst_node_temporal_embeddings(g["src"][sl], g[f"neg{b}"], g["t"][sl], None, False, k)
            ps_, pd_ = m.compute_src_dst_node_temporal_embeddings(g["src"][sl], g["dst"][sl], g["t"][sl], g["eid"][sl], True, k)
            (ns_.sum() + 2 * nd_.sum() + 3 * ps_.sum() - pd_.sum()).backward()
            m.memory_bank.detach_memory_bank()
            outs.append((torch.cat([ns_, nd_, ps_, pd_]).detach(), {n_: q.grad.clone() for n_, q in m.named_parameters() if q.grad is not None}))
        assert float((outs[0][0] - outs[1][0]).abs().max()) < 2e-6
        # (with nothing pending the full-table form never calls the GRU: its parameters then have no gradient at all)
        assert set(outs[1][1]) <= set(outs[0][1]) and all(float(outs[0][1][n_].abs().max()) == 0.0 for n_ in set(outs[0][1]) - set(outs[1][1]))
        for n_ in outs[1][1]:
            ref = outs[1][1][n_]
            assert float((outs[0][1][n_] - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), (b, n_)
        a, c = models[0].memory_bank, models[1].memory_bank
        assert torch.allclose(a.node_memories, c.node_memories, atol=1e-6) and torch.equal(a.node_last_updated_times, c.node_last_updated_times)
        assert np.array_equal(a._has, c._has) and np.array_equal(a._msg_time, c._msg_time)
        idx = torch.from_numpy(np.nonzero(a._has)[0]).cuda()
        assert torch.allclose(a._msg[idx], c._msg[idx], atol=1e-6)
        assert np.array_equal(a._has, a._has_dev.cpu().numpy().astype(bool))


def test_tgn_fused_train_step_equals_autograd_path():
    """MemoryModel.flatten_parameters() + train_step (no autograd graph) against compute_src_dst... + loss.backward() on two
    identically initialised models over 4 positive batches: embeddings, loss, every gradient, and the advanced state"""
    from flid_amd import engine, ops
    g = load_golden("tgn_small")
    ma, p, k = _model(g)
    mf, _, _ = _model(g)
    flat = mf.flatten_parameters()
    assert sorted(mf.state_dict().keys()) == sorted(ma.state_dict().keys())
    for m in (ma, mf):
        m.memory_bank.__init_memory_bank__()
    bsz = 12
    w = torch.from_numpy(np.random.RandomState(1).standard_normal((2 * bsz, g["node_feat"].shape[1])).astype(np.float32)).cuda()
    for b in range(4):
        sl = slice(b * bsz, (b + 1) * bsz)
        args = (g["src"][sl], g["dst"][sl], g["t"][sl])
        ma.zero_grad(set_to_none=True)
        s, d = ma.compute_src_dst_node_temporal_embeddings(*args, g["eid"][sl], True, k)
        ea = torch.cat([s, d])
        la = (ea * w).sum() * 0.5
        la.backward()
        flat.grad = None
        job = mf.prepare_batch_finish(mf.prepare_batch_begin(*args, k))
        ef, lf = mf.train_step(job, g["eid"][sl], lambda e: (ops.weighted_sum(e, w, 0.5), 0.5 * w), k)
        assert float((ea.detach() - ef).abs().max()) < 2e-6 and abs(float(la) - float(lf)) <= 1e-5 * max(1.0, abs(float(la)))
        named = mf._trainable()
        offs, _ = engine.block_layout(named)
        ref = dict(ma.named_parameters())
        for (name, q), o in zip([(n_, q_) for n_, q_ in mf.named_parameters() if any(q_ is t for t in named)], offs):
            pass
        by_id = {id(q): flat.grad[o:o + q.numel()].view(q.shape) for o, q in zip(offs, named)}
        for name, q in mf.named_parameters():
            if id(q) not in by_id:
                continue
            ga = ref[name].grad
            if ga is None:                       # nothing pending yet: the autograd path never called the GRU
                assert float(by_id[id(q)].abs().max()) == 0.0, name
                continue
            assert float((by_id[id(q)] - ga).abs().max()) <= 2e-5 * max(1.0, float(ga.abs().max())), (b, name)
        a, c = ma.memory_bank, mf.memory_bank
        assert torch.allclose(a.node_memories, c.node_memories, atol=1e-6) and torch.equal(a.node_last_updated_times, c.node_last_updated_times)
        assert np.array_equal(a._has, c._has) and torch.allclose(a._msg, c._msg, atol=1e-6)


def test_tgn_native_step_equals_python_fused_step():
    """the native stepper's TGN step (csrc/tg_step.hip: preparation, GRU on the touched rows + layer, backward + state advance + Adam as
    one C call each) against MemoryModel.train_step's Python form on two identically initialised models over 6 positive batches with a
    two-stage prefetch: embeddings, loss, gradient, the advanced state and the parameters after the update"""
    from flid_amd import ops
    from flid_amd.optim import FlatAdam
    g = load_golden("tgn_small")
    res = []
    bsz = 12
    w = torch.from_numpy(np.random.RandomState(1).standard_normal((2 * bsz, g["node_feat"].shape[1])).astype(np.float32)).cuda()
    loss_fn = lambda e: (ops.weighted_sum(e, w, 0.5), 0.5 * w)
    for native in (False, True):
        m, p, k = _model(g)
        flat = m.flatten_parameters()
        # (a tiny learning rate: Adam turns rounding-level gradients into +-lr steps of either sign, which must not separate the two runs)
        opt = FlatAdam([flat], lr=1e-7)
        p0 = flat.detach().clone()
        m.memory_bank.__init_memory_bank__()
        if native:
            m.enable_native_step(bsz, k)
        batch = lambda b: (g["src"][b * bsz:(b + 1) * bsz], g["dst"][b * bsz:(b + 1) * bsz], g["t"][b * bsz:(b + 1) * bsz])
        eids = lambda b: g["eid"][b * bsz:(b + 1) * bsz]
        rec = []
        jobs = {0: m.prepare_batch_begin(*batch(0), k, edge_ids=eids(0)), 1: m.prepare_batch_begin(*batch(1), k, edge_ids=eids(1))}
        for b in range(6):
            opt.zero_grad(set_to_none=True)
            job = m.prepare_batch_finish(jobs.pop(b))
            if b + 2 < 6:
                jobs[b + 2] = m.prepare_batch_begin(*batch(b + 2), k, edge_ids=eids(b + 2))
            if native:
                emb, loss = m.train_step(job, eids(b), loss_fn, k, optimizer=opt)
            else:
                emb, loss = m.train_step(job, eids(b), loss_fn, k)
                opt.step()
            bank = m.memory_bank
            rec.append((emb.clone(), float(loss), flat.grad.clone(), flat.detach().clone(), bank.node_memories.data.clone(),
                        bank.node_last_updated_times.data.clone(), bank._msg.clone(), bank._has.copy(), bank._msg_time.copy(), bank._h_last.copy()))
        res.append(rec)
    for b, (x, y) in enumerate(zip(*res)):
        assert float((x[0] - y[0]).abs().max()) <= 1e-5, (b, float((x[0] - y[0]).abs().max()))
        assert abs(x[1] - y[1]) <= 1e-5 * max(1.0, abs(x[1]))
        scale = max(1.0, float(x[2].abs().max()))
        assert float((x[2] - y[2]).abs().max()) <= 2e-5 * scale, b
        assert torch.allclose(x[4], y[4], atol=1e-5) and torch.equal(x[5], y[5]) and torch.allclose(x[6], y[6], atol=1e-5)
        assert np.array_equal(x[7], y[7]) and np.array_equal(x[8], y[8]) and np.array_equal(x[9], y[9])
    a, c = res[0][0], res[1][0]              # the first update, where the gradient is significant: the same +-lr step in both
    sig = a[2].abs() > 1e-3 * float(a[2].abs().max())
    assert int(sig.sum()) > 100
    assert float((((a[3] - p0) - (c[3] - p0)).abs() * sig).max()) <= 0.05e-7 and float(((c[3] - p0).abs() * sig).max()) >= 0.9e-7


def test_tgn_native_warmup_step_negative_then_positive_equals_autograd():
    """the warm-up's link-prediction step on a memory model (PTCL/EM_warmup.py:159-231): negatives first (edge_ids None, no state
    advance), then positives, ONE BCE over both through the MergeLayer head, backward, Adam.  Native stepper: two calls that add into one
    gradient block (PairLinkLoss halves) + the update inside the second.  Against the autograd path with torch's head, BCELoss and the
    same FlatAdam: embeddings, loss, backbone + head gradients, advanced state, over 4 batches."""
    from flid_amd import engine
    from flid_amd.heads import PairLinkLoss
    from flid_amd.models.modules import MergeLayer
    from flid_amd.optim import FlatAdam
    g = load_golden("tgn_small")
    bsz = 12
    rs = np.random.RandomState(3)
    n_nodes = g["node_feat"].shape[0]
    negs = [rs.randint(1, n_nodes, bsz).astype(np.int64) for _ in range(4)]
    out = []
    for native in (False, True):
        m, p, k = _model(g)
        D = g["node_feat"].shape[1]
        torch.manual_seed(5)
        head = MergeLayer(D, D, D, 1).cuda()
        m.memory_bank.__init_memory_bank__()
        if native:
            # (the update's learning rate is negligible: the autograd twin below takes no update)
            flat = m.flatten_parameters()
            opt = FlatAdam([flat], lr=1e-9)
            m.enable_native_step(bsz, k)
            offs, _ = engine.block_layout(m._trainable())
        rec = []
        for b in range(4):
            sl = slice(b * bsz, (b + 1) * bsz)
            src, dst, t, eid = g["src"][sl], g["dst"][sl], g["t"][sl], g["eid"][sl]
            m.zero_grad(set_to_none=True)
            head.zero_grad(set_to_none=True)
            if native:
                opt.zero_grad(set_to_none=True)
                jn = m.prepare_batch_finish(m.prepare_batch_begin(src, negs[b], t, k))
                jp = m.prepare_batch_finish(m.prepare_batch_begin(src, dst, t, k, edge_ids=eid))
                en, ln = m.train_step(jn, None, PairLinkLoss(head, False), k, edges_are_positive=False, more=True)
                en = en.clone()
                ep, lp = m.train_step(jp, eid, PairLinkLoss(head, True), k, optimizer=opt, accumulate=True)
                loss = float(ln) + float(lp)
                gflat = torch.cat([flat.grad[o:o + q.numel()] for o, q in zip(offs, m._trainable())])
            else:
                ns_, nd_ = m.compute_src_dst_node_temporal_embeddings(src, negs[b], t, None, False, k)
                ps_, pd_ = m.compute_src_dst_node_temporal_embeddings(src, dst, t, eid, True, k)
                prob = torch.cat([head(ps_, pd_), head(ns_, nd_)]).squeeze(1).sigmoid()
                lab = torch.cat([torch.ones(bsz), torch.zeros(bsz)]).cuda()
                l_ = torch.nn.BCELoss()(prob, lab)
                l_.backward()
                loss = float(l_)
                gflat = torch.cat([(q.grad if q.grad is not None else torch.zeros_like(q)).reshape(-1) for q in m._trainable()])
                m.memory_bank.detach_memory_bank()
                en, ep = torch.cat([ns_, nd_]).detach(), torch.cat([ps_, pd_]).detach()
            bank = m.memory_bank
            rec.append((en.clone(), ep.clone(), loss, gflat, [q.grad.clone() for q in head.parameters()], bank.node_memories.data.clone(),
                        bank._msg.clone(), bank._has.copy()))
        out.append(rec)
    for b, (x, y) in enumerate(zip(*out)):
        assert float((x[0] - y[0]).abs().max()) <= 1e-5 and float((x[1] - y[1]).abs().max()) <= 1e-5, b
        assert abs(x[2] - y[2]) <= 1e-5 * max(1.0, abs(x[2])), (b, x[2], y[2])
        scale = max(1e-3, float(x[3].abs().max()))
        # (2e-4 of the block's largest entry: the time encoder's weight gradient -- sums of (time interval x d phase) over all slots of both calls -- is the largest and the worst conditioned)
        assert float((x[3] - y[3]).abs().max()) <= 2e-4 * scale, (b, float((x[3] - y[3]).abs().max()), scale)
        for ga, gb in zip(x[4], y[4]):
            assert float((ga - gb).abs().max()) <= 5e-5 * max(1e-3, float(ga.abs().max()))
        assert torch.allclose(x[5], y[5], atol=1e-5) and torch.allclose(x[6], y[6], atol=1e-5) and np.array_equal(x[7], y[7])


def test_tgn_regeneration_sweep_equals_sequential_positive_calls():
    """TGN sweep (M_step.py:456-509 with the memory bank reset first): stores == the loop of positive calls; state advanced alike"""
    from flid_amd.sweep import regenerate_embeddings
    g = load_golden("tgn_small")
    ma, p, k = _model(g)
    mb, _, _ = _model(g)
    for m in (ma, mb):
        m.eval()
        m.memory_bank.__init_memory_bank__()
    d = _Data(g)
    n = 84
    d.src_node_ids, d.dst_node_ids, d.node_interact_times, d.edge_ids = g["src"][:n], g["dst"][:n], g["t"][:n], g["eid"][:n]
    d.num_interactions = n
    s_store, d_store = regenerate_embeddings(ma, d, batch_size=12, num_neighbors=k)
    with torch.no_grad():
        for lo in range(0, n, 12):
            a, b = mb.compute_src_dst_node_temporal_embeddings(g["src"][lo:lo + 12], g["dst"][lo:lo + 12], g["t"][lo:lo + 12], g["eid"][lo:lo + 12], True, k)
            assert float((a - s_store[lo:lo + 12]).abs().max()) < 2e-6 and float((b - d_store[lo:lo + 12]).abs().max()) < 2e-6
    assert torch.allclose(ma.memory_bank.node_memories, mb.memory_bank.node_memories, atol=1e-6)


def test_prepare_batch_rejects_ids_beyond_the_graph_and_counts_history_on_the_host():
    """tg_tgn_prepare_batch raises the reference's IndexError for an id beyond the graph (before anything is launched);
    TemporalGraph.count_before_host (the host-side binary searches DyGFormer shapes its windows with) equals the oracle's history_end"""
    from flid_amd.models.MemoryModel import MemoryModel
    from flid_amd.synth import reddit_like
    from flid_amd.utils.utils import get_neighbor_sampler
    data = reddit_like(num_edges=5000, seed=1)
    sampler = get_neighbor_sampler(data, "recent", seed=0)
    m = MemoryModel(data.node_raw_features, data.edge_raw_features, sampler, 100, "TGN", 1, 2, 0.0, device="cuda:0").to("cuda:0").train()
    m.memory_bank.__init_memory_bank__()
    sl = slice(3000, 3100)
    bad = data.src_node_ids[sl].copy()
    bad[7] = m.num_nodes + 3
    with pytest.raises(IndexError):
        m.prepare_batch_begin(bad, data.dst_node_ids[sl], data.node_interact_times[sl], 20, edge_ids=data.edge_ids[sl])
    job = m.prepare_batch_finish(m.prepare_batch_begin(data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl], 20,
                                                       edge_ids=data.edge_ids[sl]))
    torch.cuda.synchronize()
    both = np.concatenate([data.src_node_ids[sl], data.dst_node_ids[sl]])
    assert sorted(job["u"].tolist()) == sorted(set(both.tolist()))
    last = {int(v): t for v, t in zip(both, np.concatenate([data.node_interact_times[sl]] * 2))}
    assert all(last[int(v)] == t for v, t in zip(job["u"], job["new_t"]))
    uniq = set(job["uniq"].cpu().tolist())
    assert set(both.tolist()) <= uniq and set(job["S"][0].cpu().numpy().reshape(-1).tolist()) <= uniq
    adj = O.build_adjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    ids = data.dst_node_ids[2000:2600]
    times = data.node_interact_times[2000:2600]
    want = np.array([O.history_end(adj, int(v), t) for v, t in zip(ids, times)])
    assert np.array_equal(sampler.graph.count_before_host(ids, times), want)


def test_tgn_native_step_ragged_batch_sizes():
    """the same two paths over chronological batches of DIFFERENT sizes through one native object (12, 3, 12, 5, 1, 9 edges: every slab and
    scratch region is sized for the largest; a smaller batch must not read what its launches did not write)"""
    from flid_amd import ops
    from flid_amd.optim import FlatAdam
    g = load_golden("tgn_small")
    sizes = [12, 3, 12, 5, 1, 9]
    bounds = np.concatenate([[0], np.cumsum(sizes)])
    assert bounds[-1] <= len(g["src"])
    dn = g["node_feat"].shape[1]
    res = []
    for native in (False, True):
        m, p, k = _model(g)
        flat = m.flatten_parameters()
        opt = FlatAdam([flat], lr=1e-7)
        m.memory_bank.__init_memory_bank__()
        if native:
            m.enable_native_step(max(sizes), k)
        rec = []
        for b in range(len(sizes)):
            sl = slice(int(bounds[b]), int(bounds[b + 1]))
            w = torch.from_numpy(np.random.RandomState(b).standard_normal((2 * sizes[b], dn)).astype(np.float32)).cuda()
            loss_fn = lambda e, w=w: (ops.weighted_sum(e, w, 0.5), 0.5 * w)
            opt.zero_grad(set_to_none=True)
            job = m.prepare_batch_finish(m.prepare_batch_begin(g["src"][sl], g["dst"][sl], g["t"][sl], k, edge_ids=g["eid"][sl]))
            if native:
                emb, loss = m.train_step(job, g["eid"][sl], loss_fn, k, optimizer=opt)
            else:
                emb, loss = m.train_step(job, g["eid"][sl], loss_fn, k)
                opt.step()
            bank = m.memory_bank
            rec.append((emb.clone(), float(loss), flat.grad.clone(), bank.node_memories.data.clone(), bank._msg.clone(), bank._has.copy()))
        res.append(rec)
    for b, (x, y) in enumerate(zip(*res)):
        assert float((x[0] - y[0]).abs().max()) <= 1e-5, (b, float((x[0] - y[0]).abs().max()))
        assert abs(x[1] - y[1]) <= 1e-5 * max(1.0, abs(x[1])), b
        scale = max(1.0, float(x[2].abs().max()))
        assert float((x[2] - y[2]).abs().max()) <= 2e-5 * scale, (b, float((x[2] - y[2]).abs().max()), scale)
        assert torch.allclose(x[3], y[3], atol=1e-5) and torch.allclose(x[4], y[4], atol=1e-5) and np.array_equal(x[5], y[5]), b
