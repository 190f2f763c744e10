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


def hash_features_host(rows: np.ndarray, cols: int, seed: int) -> np.ndarray:
    """Host mirror of tg_hash_features (flid_amd/csrc/tg_rowops.hip): the given rows of a hashed feature table, bit-exact."""
    rows = np.asarray(rows, dtype=np.uint64)
    with np.errstate(over="ignore"):
        idx = rows[:, None] * np.uint64(cols) + np.arange(cols, dtype=np.uint64)[None, :]
        x = np.uint64(seed) ^ (idx * np.uint64(0x9E3779B97F4A7C15))
        x ^= x >> np.uint64(33); x *= np.uint64(0xff51afd7ed558ccd); x ^= x >> np.uint64(33)
        x *= np.uint64(0xc4ceb9fe1a85ec53); x ^= x >> np.uint64(33)
    h = ((x >> np.uint64(11)) & np.uint64(0xFFFFFFFF)).astype(np.uint32) & np.uint32(0xFFFFFF)
    u = h.astype(np.float32) * np.float32(1.0 / 16777216.0)
    out = (u - np.float32(0.5)) * np.float32(3.4641016)
    out[rows == 0] = 0.0
    return out.astype(np.float32)


def scale_like(num_users=9_000_000, num_items=1_000_000, num_edges=100_000_000, seed=0, chunk=10_000_000) -> Data:
    """bipartite 9 M / 1 M nodes, 100 M edges, popularity ~ rank^-0.8, integer timestamps, chronological
    edge ids.  Only the interaction arrays are built here (3.2 GB); the feature tables are hashed into HBM (ops.hash_features)."""
    rs = np.random.RandomState(seed)

    def draw(n_items):
        p = np.arange(1, n_items + 1, dtype=np.float64) ** -0.8
        cdf = np.cumsum(p)
        cdf /= cdf[-1]
        out = np.empty(num_edges, dtype=np.int64)
        for lo in range(0, num_edges, chunk):
            hi = min(num_edges, lo + chunk)
            out[lo:hi] = np.searchsorted(cdf, rs.random_sample(hi - lo), side="right")
        np.minimum(out, n_items - 1, out=out)
        return out

    src = draw(num_users) + 1
    dst = draw(num_items) + 1 + num_users
    # sorted event times without a 100 M-element sort: normalised running sum of exponential gaps (a Poisson stream)
    t = np.cumsum(rs.exponential(size=num_edges))
    t *= 2.678e6 / t[-1]
    np.round(t, 0, out=t)
    dst[-1] = num_users + num_items
    eid = np.arange(1, num_edges + 1, dtype=np.int64)
    return Data(src, dst, t, eid, np.zeros(num_edges, dtype=np.int64), None, None)
