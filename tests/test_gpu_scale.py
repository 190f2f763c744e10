"""BASELINE config 5 (10 M nodes / 100 M edges, feature tables hashed into HBM) at a size that CROSSES the 32-bit
limits the reference's python-list adjacency (utils/utils.py:297-300) never meets: an edge table of 13 M rows x 172 fp32 =
2.24e9 elements (> 2^31) = 8.9 GB (> 2^32 bytes), roots at the END of the stream so that the sampled edge rows lie beyond both
limits.  Sampled ids bit-exact against the oracle's adjacency; embeddings within 1e-4 of the oracle fed host-recomputed rows."""
import numpy as np
import pytest
import torch

import oracle.flid_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


class HashedTable:
    """stand-in for a (rows, cols) feature tensor: any row recomputed on the host (synth.hash_features_host)"""

    def __init__(self, cols, seed):
        self.cols, self.seed = cols, seed

    def __getitem__(self, idx):
        from flid_amd.synth import hash_features_host
        idx = idx.numpy() if torch.is_tensor(idx) else np.asarray(idx)
        rows = hash_features_host(idx.reshape(-1), self.cols, self.seed)
        return torch.from_numpy(rows).reshape(tuple(idx.shape) + (self.cols,))


def test_tables_past_2_31_elements_ids_bit_exact_and_embeddings_match_oracle():
    from flid_amd import ops
    from flid_amd.models.TGAT import TGAT
    from flid_amd.synth import hash_features_host, scale_like
    from flid_amd.utils.utils import get_neighbor_sampler
    free, _ = torch.cuda.mem_get_info()
    if free < 16 * 2 ** 30:
        pytest.skip("needs 16 GB of free HBM")
    U, I, E, D, K = 1_200_000, 100_000, 13_000_000, 172, 20
    data = scale_like(num_users=U, num_items=I, num_edges=E, seed=5, chunk=4_000_000)
    dev = torch.device("cuda:0")
    node_tab = ops.hash_features(U + I + 1, D, 1, dev)
    edge_tab = ops.hash_features(E + 1, D, 2, dev)
    assert edge_tab.numel() > 2 ** 31 and edge_tab.numel() * 4 > 2 ** 32
    # rows on both sides of the element-index and byte-offset limits, and the last one
    r31, r32 = 2 ** 31 // D, 2 ** 32 // (4 * D)
    erows = np.array([1, r32 - 1, r32, r32 + 1, r31 - 1, r31, r31 + 1, E - 1, E])
    assert np.array_equal(edge_tab[torch.from_numpy(erows).to(dev)].cpu().numpy(), hash_features_host(erows, D, 2))

    sampler = get_neighbor_sampler(data, "recent", seed=0)
    sl = slice(E - 16, E)
    bs, bd, bt = data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl]
    adj = O.build_adjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    ids, ts = np.concatenate([bs, bd]), np.concatenate([bt, bt])
    want = O.sample_recent(adj, ids, ts, K)
    got = sampler.get_historical_neighbors(ids, ts, K)
    for g_, w_ in zip(got, want):
        assert np.array_equal(np.asarray(g_), w_)                       # neighbor ids, edge ids, float32 times: bit-exact
    assert int(want[1].max()) > r31, "the sampled edge rows must lie past 2^31 elements"
    # hop 2: the float32 neighbor times fed back as query times (models/TGAT.py:110-111)
    want2 = O.sample_recent(adj, want[0].reshape(-1), want[2].reshape(-1), K)
    got2 = sampler.get_historical_neighbors(want[0].reshape(-1), want[2].reshape(-1), K)
    for g_, w_ in zip(got2, want2):
        assert np.array_equal(np.asarray(g_), w_)
    assert int(want2[1].max()) > r32

    torch.manual_seed(0)
    m = TGAT(node_tab, edge_tab, sampler, 100, 2, 2, 0.0, "cuda:0").to(dev).eval()
    with torch.no_grad():
        for prm in m.parameters():
            if prm.dim() > 1 and prm.shape[1] > 1:
                prm.copy_(torch.randn_like(prm) * 0.05)
        s, d = m.compute_src_dst_node_temporal_embeddings(bs, bd, bt, K)
    p = {k_: v.detach().cpu() for k_, v in m.state_dict().items()}
    orc = O.TGATOracle(HashedTable(D, 1), HashedTable(D, 2), adj, p, 2, 2)
    with torch.no_grad():
        os_, od_ = orc.src_dst(bs, bd, bt, K)
    np.testing.assert_allclose(s.cpu().numpy(), os_.numpy(), atol=TOL)
    np.testing.assert_allclose(d.cpu().numpy(), od_.numpy(), atol=TOL)
    # and the backward through the same rows: gradient of a scalar of the embeddings, finite and reproducible
    m.train()
    s1, d1 = m.compute_src_dst_node_temporal_embeddings(bs, bd, bt, K)
    (s1.sum() - d1.sum()).backward()
    g1 = {n: prm.grad.clone() for n, prm in m.named_parameters()}
    for prm in m.parameters():
        prm.grad = None
    s2, d2 = m.compute_src_dst_node_temporal_embeddings(bs, bd, bt, K)
    (s2.sum() - d2.sum()).backward()
    for n, prm in m.named_parameters():
        assert torch.isfinite(prm.grad).all()
        assert float((prm.grad - g1[n]).abs().max()) <= 1e-4 * max(1.0, float(g1[n].abs().max())), n
