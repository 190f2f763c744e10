"""Ingest: the reference's on-disk layout round-trips through flid_amd.ingest; the device CSR built from it equals the
oracle's adjacency (GPU part)."""
import numpy as np
import pytest

from flid_amd import ingest
from flid_amd.synth import wikipedia_like


def _tiny():
    d = wikipedia_like(num_edges=500, num_users=40, num_items=12, feat_dim=172, seed=2, zero_node_feat=False)
    d.labels_time = d.node_interact_times - 1.0
    return d


def test_write_then_load_round_trip_with_zero_padding(tmp_path):
    d = _tiny()
    ingest.write_dataset(str(tmp_path), "toy", d, node_feat_cols=5, edge_feat_cols=9)       # narrow tables on disk
    got = ingest.load_dataset(str(tmp_path), "toy")
    assert np.array_equal(got.src_node_ids, d.src_node_ids) and got.src_node_ids.dtype == np.longlong
    assert np.array_equal(got.dst_node_ids, d.dst_node_ids) and np.array_equal(got.edge_ids, d.edge_ids)
    assert np.array_equal(got.node_interact_times, d.node_interact_times) and got.node_interact_times.dtype == np.float64
    assert np.array_equal(got.labels, d.labels) and np.allclose(got.labels_time, d.labels_time)
    assert got.node_raw_features.shape == (53, 172) and got.edge_raw_features.shape == (501, 172) and got.node_raw_features.dtype == np.float32
    assert np.array_equal(got.node_raw_features[:, :5], d.node_raw_features[:, :5]) and not got.node_raw_features[:, 5:].any()
    assert np.array_equal(got.edge_raw_features[:, :9], d.edge_raw_features[:, :9]) and not got.edge_raw_features[:, 9:].any()
    tr, va, te = ingest.chronological_split(got, 0.15, 0.15)
    assert tr.num_interactions + va.num_interactions + te.num_interactions == 500
    assert tr.node_interact_times.max() < va.node_interact_times.min() <= va.node_interact_times.max() < te.node_interact_times.min()
    assert got.num_unique_nodes == len(set(d.src_node_ids) | set(d.dst_node_ids))


def test_too_wide_features_are_refused(tmp_path):
    d = _tiny()
    d.edge_raw_features = np.zeros((501, 200), dtype=np.float32)
    ingest.write_dataset(str(tmp_path), "wide", d)
    with pytest.raises(AssertionError, match="Edge feature dimension in dataset wide is bigger than 172"):
        ingest.load_dataset(str(tmp_path), "wide")


@pytest.mark.gpu
def test_device_graph_from_loaded_dataset_matches_oracle_adjacency(tmp_path):
    import torch
    from oracle import flid_oracle as O
    d = _tiny()
    perm = np.random.RandomState(0).permutation(500)          # a non-chronological file: rows are re-sorted per node, stably
    d2 = ingest.Data(d.src_node_ids[perm], d.dst_node_ids[perm], d.node_interact_times[perm], d.edge_ids[perm], d.labels[perm], None,
                     d.node_raw_features, d.edge_raw_features)
    ingest.write_dataset(str(tmp_path), "toy", d2)
    got = ingest.load_dataset(str(tmp_path), "toy")
    g, node, edge = ingest.to_device(got, torch.device("cuda:0"))
    rp, nb, ei, tt = g.host_csr()
    adj = O.build_adjacency(got.src_node_ids, got.dst_node_ids, got.edge_ids, got.node_interact_times, got.node_raw_features.shape[0])
    assert np.array_equal(rp, adj.row_ptr) and np.array_equal(nb, adj.nbr) and np.array_equal(ei, adj.eid) and np.array_equal(tt, adj.t)
    assert node.shape == (53, 172) and edge.is_cuda


@pytest.mark.gpu
def test_threaded_graph_build_equals_single_threaded_result():
    """>= 2^20 edges take the multi-threaded host build (disjoint node ranges per thread): same CSR as the oracle's"""
    from flid_amd.graph import TemporalGraph
    from oracle import flid_oracle as O
    rs = np.random.RandomState(5)
    E, N = (1 << 20) + 777, 5000
    src = rs.randint(1, N, E).astype(np.int64)
    dst = rs.randint(1, N, E).astype(np.int64)
    t = np.sort(rs.uniform(0, 1e6, E)).round(1)
    t[1000:1010] = t[1000]
    eid = np.arange(1, E + 1, dtype=np.int64)
    for times in (t, t[rs.permutation(E)]):                    # chronological and shuffled streams
        g = TemporalGraph(src, dst, eid, times, num_rows=N)
        rp, nb, ei, tt = g.host_csr()
        adj = O.build_adjacency(src, dst, eid, times, N)
        assert np.array_equal(rp, adj.row_ptr) and np.array_equal(nb, adj.nbr) and np.array_equal(ei, adj.eid) and np.array_equal(tt, adj.t)
