"""TCL backbone -- drop-in for the reference class (models/TCL.py): same constructor, forward surface, parameter names / shapes.
The neighbor lookups run in the device sampler (tg_sample_recent; the host mirror for the random strategies), the row gathers, the
time encoding, every product, the key-masked attention, LayerNorm and dropout in libflid_tg (modules.TransformerEncoder, seqops)."""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..utils.utils import NeighborSampler
from .modules import TimeEncoder, TransformerEncoder, linear


class TCL(nn.Module):

    def __init__(self, node_raw_features: np.ndarray, edge_raw_features: np.ndarray, neighbor_sampler: NeighborSampler,
                 time_feat_dim: int, num_layers: int = 2, num_heads: int = 2, num_depths: int = 20, dropout: float = 0.1, device: str = 'cpu'):
        super().__init__()
        if torch.device(device).type != "cuda":
            raise RuntimeError("flid_amd.TCL runs on a ROCm device only; there is no CPU path")
        self.node_raw_features = torch.from_numpy(node_raw_features.astype(np.float32)).to(device).contiguous()
        self.edge_raw_features = torch.from_numpy(edge_raw_features.astype(np.float32)).to(device).contiguous()
        self.neighbor_sampler = neighbor_sampler
        self.node_feat_dim, self.edge_feat_dim = self.node_raw_features.shape[1], self.edge_raw_features.shape[1]
        self.time_feat_dim, self.num_layers, self.num_heads = time_feat_dim, num_layers, num_heads
        self.num_depths, self.dropout, self.device = num_depths, dropout, device
        self.time_encoder = TimeEncoder(time_dim=time_feat_dim)
        self.depth_embedding = nn.Embedding(num_embeddings=num_depths, embedding_dim=self.node_feat_dim)
        self.projection_layer = nn.ModuleDict({
            'node': nn.Linear(self.node_feat_dim, self.node_feat_dim, bias=True),
            'edge': nn.Linear(self.edge_feat_dim, self.node_feat_dim, bias=True),
            'time': nn.Linear(self.time_feat_dim, self.node_feat_dim, bias=True)})
        self.transformers = nn.ModuleList([TransformerEncoder(attention_dim=self.node_feat_dim, num_heads=self.num_heads, dropout=self.dropout)
                                           for _ in range(self.num_layers)])
        self.output_layer = nn.Linear(self.node_feat_dim, self.node_feat_dim, bias=True)

    # ----------------------------------------------------------------------------------------------------------------------
    def _sequences(self, node_ids: np.ndarray, times: np.ndarray, k: int):
        """[the node itself | its k sampled neighbors] per root (models/TCL.py:75-109): ids, edge ids, query time - slot time"""
        dev = self.node_raw_features.device
        sampler = self.neighbor_sampler
        n = len(node_ids)
        if sampler.sample_neighbor_strategy == "recent" or getattr(sampler, "device_random", False):
            ids_d, t_d = ops.h2d([np.ascontiguousarray(node_ids, dtype=np.int32), np.ascontiguousarray(times, dtype=np.float64)], dev)
            nbr, eid, _, dt = (sampler.graph.sample_recent(ids_d, t_d, k) if sampler.sample_neighbor_strategy == "recent"
                               else sampler.sample_on_device(ids_d, t_d, k))
        else:                                   # uniform / time_interval_aware: the host mirror consumes numpy's stream as the reference does
            nb, ne, nt = sampler.get_historical_neighbors(node_ids, times, k)
            ids_d, nbr, eid, dt = ops.h2d([np.ascontiguousarray(node_ids, dtype=np.int32), nb.astype(np.int32), ne.astype(np.int32),
                                           (np.asarray(times, dtype=np.float64)[:, None] - nt).astype(np.float32)], dev)
        zi = torch.zeros((n, 1), dtype=torch.int32, device=dev)
        seq_ids = torch.cat([ids_d.view(n, 1), nbr], dim=1)
        seq_eid = torch.cat([zi, eid], dim=1)
        seq_dt = torch.cat([torch.zeros((n, 1), dtype=torch.float32, device=dev), dt], dim=1)      # the node itself: t - t = 0
        return seq_ids, seq_eid, seq_dt

    def _features(self, seq_ids, seq_eid, seq_dt):
        """get_features + the three projections + the depth embedding (models/TCL.py:111-142, 190-220)"""
        n, s = seq_ids.shape
        assert s == self.depth_embedding.weight.shape[0]
        nf = ops.gather_rows(self.node_raw_features, seq_ids.reshape(-1)).view(n, s, -1)
        ef = ops.gather_rows(self.edge_raw_features, seq_eid.reshape(-1)).view(n, s, -1)
        tf = self.time_encoder(seq_dt)
        pl = self.projection_layer
        return (linear(nf, pl['node'].weight, pl['node'].bias, exact=True) + linear(ef, pl['edge'].weight, pl['edge'].bias, exact=True) +
                linear(tf, pl['time'].weight, pl['time'].bias, exact=True) + self.depth_embedding.weight.unsqueeze(0))

    def compute_src_dst_node_temporal_embeddings(self, src_node_ids: np.ndarray, dst_node_ids: np.ndarray,
                                                 node_interact_times: np.ndarray, num_neighbors: int = 20):
        src_node_ids, dst_node_ids = np.asarray(src_node_ids), np.asarray(dst_node_ids)
        node_interact_times = np.asarray(node_interact_times)
        g = getattr(self.neighbor_sampler, "graph", None)
        for ids in (src_node_ids, dst_node_ids):
            if g is not None and len(ids) and (int(ids.max()) >= g.num_rows or int(ids.min()) < 0):
                raise IndexError("list index out of range")
        # (the reference samples the sources' neighbors, then the destinations': the order matters for the random strategies)
        s_ids, s_eid, s_dt = self._sequences(src_node_ids, node_interact_times, num_neighbors)
        d_ids, d_eid, d_dt = self._sequences(dst_node_ids, node_interact_times, num_neighbors)
        src_x, dst_x = self._features(s_ids, s_eid, s_dt), self._features(d_ids, d_eid, d_dt)
        src_e = dst_e = None
        for transformer in self.transformers:
            src_x = transformer(inputs_query=src_x, inputs_key=src_x, inputs_value=src_x, neighbor_masks=s_ids)      # self-attention
            dst_x = transformer(inputs_query=dst_x, inputs_key=dst_x, inputs_value=dst_x, neighbor_masks=d_ids)
            src_e = transformer(inputs_query=src_x, inputs_key=dst_x, inputs_value=dst_x, neighbor_masks=d_ids)      # cross-attention
            dst_e = transformer(inputs_query=dst_x, inputs_key=src_x, inputs_value=src_x, neighbor_masks=s_ids)
            src_x, dst_x = src_e, dst_e
        ol = self.output_layer
        return linear(src_e[:, 0, :], ol.weight, ol.bias, exact=True), linear(dst_e[:, 0, :], ol.weight, ol.bias, exact=True)

    def set_neighbor_sampler(self, neighbor_sampler: NeighborSampler):
        self.neighbor_sampler = neighbor_sampler
        if self.neighbor_sampler.sample_neighbor_strategy in ['uniform', 'time_interval_aware']:
            assert self.neighbor_sampler.seed is not None
            self.neighbor_sampler.reset_random_state()
