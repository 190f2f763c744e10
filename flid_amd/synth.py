"""Synthetic temporal interaction graphs of the Wikipedia / Reddit shape (the reference's
processed_data/ blobs are not shipped, so every measurement runs on these).  Deterministic in `seed`."""
from dataclasses import dataclass

import numpy as np


@dataclass
class Data:
    """Same fields as the reference's utils/DataLoader.py:46-65 `Data`, plus the two feature tables."""
    src_node_ids: np.ndarray
    dst_node_ids: np.ndarray
    node_interact_times: np.ndarray
    edge_ids: np.ndarray
    labels: np.ndarray
    node_raw_features: np.ndarray = None
    edge_raw_features: np.ndarray = None

    @property
    def num_interactions(self):
        return len(self.src_node_ids)

    def slice(self, lo, hi):
        return Data(self.src_node_ids[lo:hi], self.dst_node_ids[lo:hi], self.node_interact_times[lo:hi], self.edge_ids[lo:hi],
                    self.labels[lo:hi], self.node_raw_features, self.edge_raw_features)


def _bipartite(num_users, num_items, num_edges, t_max, decimals, feat_dim, seed, zero_node_feat=True):
    rs = np.random.RandomState(seed)
    pu = np.arange(1, num_users + 1, dtype=np.float64) ** -0.8
    pi = np.arange(1, num_items + 1, dtype=np.float64) ** -0.8
    src = rs.choice(num_users, size=num_edges, p=pu / pu.sum()).astype(np.int64) + 1
    dst = rs.choice(num_items, size=num_edges, p=pi / pi.sum()).astype(np.int64) + 1 + num_users
    t = np.sort(rs.uniform(0.0, t_max, size=num_edges)).round(decimals).astype(np.float64)
    eid = np.arange(1, num_edges + 1, dtype=np.int64)
    n_rows = num_users + num_items + 1
    node = np.zeros((n_rows, feat_dim), dtype=np.float32)
    if not zero_node_feat:
        node[1:] = rs.standard_normal((n_rows - 1, feat_dim)).astype(np.float32)
    edge = np.zeros((num_edges + 1, feat_dim), dtype=np.float32)
    edge[1:] = rs.standard_normal((num_edges, feat_dim)).astype(np.float32)
    labels = (rs.uniform(size=num_edges) < 0.002).astype(np.int64)
    # make sure the largest item id occurs so that num_rows = max id + 1 as the reference sizes it
    dst[-1] = num_users + num_items
    return Data(src, dst, t, eid, labels, node, edge)


def wikipedia_like(num_edges=157474, num_users=8227, num_items=1000, feat_dim=172, seed=0, zero_node_feat=True):
    """9 227 nodes / 157 474 edges (reference README.md:41-42), integer timestamps over one month."""
    return _bipartite(num_users, num_items, num_edges, 2.678e6, 0, feat_dim, seed, zero_node_feat)


def reddit_like(num_edges=672447, num_users=10000, num_items=984, feat_dim=172, seed=0, zero_node_feat=True):
    """10 984 nodes / 672 447 edges, timestamps to 3 decimals (exercises float32-rounded hop times)."""
    return _bipartite(num_users, num_items, num_edges, 2.678e6, 3, feat_dim, seed, zero_node_feat)
