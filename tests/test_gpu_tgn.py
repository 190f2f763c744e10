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
