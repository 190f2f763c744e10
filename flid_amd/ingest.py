"""Graph ingest: the reference's on-disk dataset format -> host arrays + device-resident tables and CSR.

replaces the loading half of utils/DataLoader.py:229-278 (get_PTCL_data): `processed_data/{name}/ml_{name}.csv` with columns
u, i, ts, label, last_ts, idx (utils/DataLoader.py:262-278; `label_u/label_i/last_u_ts/last_i_ts` for the double-way datasets),
`ml_{name}.npy` = edge features (E + 1, De) and `ml_{name}_node.npy` = node features (N + 1, Dn), both zero-padded on the right to
172 columns (:253-258; 384 for 'oag').  Row 0 of both tables is the padding row.  The shipped utils/DataLoader.py cannot be imported
(SyntaxError at :239) and no dataset ships with the reference (.MISSING_LARGE_BLOBS): `write_dataset` produces files of the same
layout from any Data object, for tests and for exporting the synthetic workloads."""
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

FEAT_DIM = {"oag": 384}          # every other dataset: 172 (utils/DataLoader.py:240-246)
DOUBLE_WAY = ("arxiv", "oag")    # label per endpoint (utils/DataLoader.py:265)


@dataclass
class Data:
    """the reference's Data record (utils/DataLoader.py:46-65) plus the two feature tables"""
    src_node_ids: np.ndarray
    dst_node_ids: np.ndarray
    node_interact_times: np.ndarray
    edge_ids: np.ndarray
    labels: object
    labels_time: object = None
    node_raw_features: Optional[np.ndarray] = None
    edge_raw_features: Optional[np.ndarray] = None

    @property
    def num_interactions(self):
        return len(self.src_node_ids)

    @property
    def unique_node_ids(self):
        return set(self.src_node_ids) | set(self.dst_node_ids)

    @property
    def num_unique_nodes(self):
        return len(self.unique_node_ids)

    def select(self, mask):
        pick = lambda x: None if x is None else ([y[mask] for y in x] if isinstance(x, list) else x[mask])
        return Data(self.src_node_ids[mask], self.dst_node_ids[mask], self.node_interact_times[mask], self.edge_ids[mask],
                    pick(self.labels), pick(self.labels_time), self.node_raw_features, self.edge_raw_features)


def _pad_columns(x: np.ndarray, width: int, what: str, name: str) -> np.ndarray:
    assert width >= x.shape[1], f'{what} feature dimension in dataset {name} is bigger than {width}!'      # DataLoader.py:251-252
    out = np.zeros((x.shape[0], width), dtype=np.float32)
    out[:, :x.shape[1]] = x
    return out


def _read_csv(path: str):
    """header + numeric columns; pandas when present (fast C parser), numpy otherwise"""
    try:
        import pandas as pd
        df = pd.read_csv(path)
        return {c: df[c].values for c in df.columns}
    except ImportError:
        with open(path) as f:
            cols = f.readline().strip().split(",")
        arr = np.loadtxt(path, delimiter=",", skiprows=1, ndmin=2)
        return {c: arr[:, j] for j, c in enumerate(cols)}


def load_dataset(root: str, name: str) -> Data:
    """`root`/{name}/ml_{name}.csv + .npy + _node.npy -> Data with float32 tables padded to the dataset's feature width"""
    base = os.path.join(root, name, f"ml_{name}")
    cols = _read_csv(base + ".csv")
    width = FEAT_DIM.get(name, 172)
    edge = _pad_columns(np.load(base + ".npy"), width, "Edge", name)
    node = _pad_columns(np.load(base + "_node.npy"), width, "Node", name)
    src = cols["u"].astype(np.longlong)
    dst = cols["i"].astype(np.longlong)
    ts = cols["ts"].astype(np.float64)
    eid = cols["idx"].astype(np.longlong)
    if name in DOUBLE_WAY:
        labels, labels_time = [cols["label_u"], cols["label_i"]], [cols["last_u_ts"], cols["last_i_ts"]]
    else:
        labels, labels_time = cols["label"], cols.get("last_ts")
    assert node.shape[0] > max(int(src.max()), int(dst.max())) and edge.shape[0] > int(eid.max()), "feature tables shorter than the id range"
    return Data(src, dst, ts, eid, labels, labels_time, node, edge)


def write_dataset(root: str, name: str, data, node_feat_cols: Optional[int] = None, edge_feat_cols: Optional[int] = None):
    """files in the reference's layout from a Data-like object (tables optionally cut to their first k columns: the loader pads back)"""
    os.makedirs(os.path.join(root, name), exist_ok=True)
    base = os.path.join(root, name, f"ml_{name}")
    n = len(data.src_node_ids)
    labels = np.asarray(data.labels) if getattr(data, "labels", None) is not None else np.zeros(n, dtype=np.int64)
    last = getattr(data, "labels_time", None)
    last = np.asarray(last) if last is not None else np.asarray(data.node_interact_times)
    with open(base + ".csv", "w") as f:
        f.write("u,i,ts,label,last_ts,idx\n")
        for k in range(n):
            f.write(f"{int(data.src_node_ids[k])},{int(data.dst_node_ids[k])},{float(data.node_interact_times[k])!r},{int(labels[k])},"
                    f"{float(last[k])!r},{int(data.edge_ids[k])}\n")
    np.save(base + ".npy", np.asarray(data.edge_raw_features)[:, :edge_feat_cols])
    np.save(base + "_node.npy", np.asarray(data.node_raw_features)[:, :node_feat_cols])


def chronological_split(data: Data, val_ratio: float, test_ratio: float):
    """train / val / test by the timestamp quantiles of utils/DataLoader.py:405-410"""
    t = data.node_interact_times
    val_time, test_time = list(np.quantile(t, [1 - val_ratio - test_ratio, 1 - test_ratio]))
    return data.select(t <= val_time), data.select(np.logical_and(t <= test_time, t > val_time)), data.select(t > test_time)


def to_device(data: Data, device):
    """(TemporalGraph, node table, edge table) resident in HBM: what the backbones' constructors and get_neighbor_sampler build
    (utils/utils.py:283-302, models/TGAT.py:26-29), in one place -- the CSR is built by tg_graph_create (multi-threaded host
    counting sort, one H2D copy)"""
    import torch
    from .graph import TemporalGraph
    g = TemporalGraph(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times, num_rows=data.node_raw_features.shape[0])
    return g, torch.from_numpy(data.node_raw_features).to(device), torch.from_numpy(data.edge_raw_features).to(device)
