"""GraphMixer backbone -- drop-in for the reference class (models/GraphMixer.py): same constructor, forward surface, parameter
names / shapes.  Link encoder: device sampler + masked time encoding + MLP-Mixer blocks on the HIP products / LayerNorm / GELU /
dropout kernels; node encoder: the mean over the `time_gap` most recent neighbors' rows is one kernel (tg_recent_window_mean) instead
of a (batch, time_gap, node_feat_dim) gather."""
import numpy as np
import torch
import torch.nn as nn

from .. import ops, seqops
from ..utils.utils import NeighborSampler
from .modules import TimeEncoder, linear


class FeedForwardNet(nn.Module):
    """Linear -> GELU -> Dropout -> Linear -> Dropout (models/GraphMixer.py:169-196)"""

    def __init__(self, input_dim: int, dim_expansion_factor: float, dropout: float = 0.0):
        super().__init__()
        self.input_dim, self.dim_expansion_factor, self.dropout = input_dim, dim_expansion_factor, dropout
        self.ffn = nn.Sequential(nn.Linear(input_dim, int(dim_expansion_factor * input_dim)), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(int(dim_expansion_factor * input_dim), input_dim), nn.Dropout(dropout))

    def forward(self, x: torch.Tensor):
        p, tr = self.dropout, self.training
        h = seqops.dropout(seqops.gelu(linear(x, self.ffn[0].weight, self.ffn[0].bias)), p, tr)
        return seqops.dropout(linear(h, self.ffn[3].weight, self.ffn[3].bias), p, tr)


class MLPMixer(nn.Module):
    """token mixing over the neighbor axis, then channel mixing, both pre-LN with residuals (models/GraphMixer.py:199-246)"""

    def __init__(self, num_tokens: int, num_channels: int, token_dim_expansion_factor: float = 0.5,
                 channel_dim_expansion_factor: float = 4.0, dropout: float = 0.0):
        super().__init__()
        self.token_norm = nn.LayerNorm(num_tokens)
        self.token_feedforward = FeedForwardNet(num_tokens, token_dim_expansion_factor, dropout)
        self.channel_norm = nn.LayerNorm(num_channels)
        self.channel_feedforward = FeedForwardNet(num_channels, channel_dim_expansion_factor, dropout)

    def forward(self, input_tensor: torch.Tensor):
        h = seqops.layer_norm(input_tensor.permute(0, 2, 1).contiguous(), self.token_norm.weight, self.token_norm.bias)
        output_tensor = self.token_feedforward(h).permute(0, 2, 1) + input_tensor
        h = seqops.layer_norm(output_tensor, self.channel_norm.weight, self.channel_norm.bias)
        return self.channel_feedforward(h) + output_tensor


class GraphMixer(nn.Module):

    def __init__(self, node_raw_features: np.ndarray, edge_raw_features: np.ndarray, neighbor_sampler: NeighborSampler,
                 time_feat_dim: int, num_tokens: int, num_layers: int = 2, token_dim_expansion_factor: float = 0.5,
                 channel_dim_expansion_factor: float = 4.0, dropout: float = 0.1, device: str = 'cpu'):
        super().__init__()
        if torch.device(device).type != "cuda":
            raise RuntimeError("flid_amd.GraphMixer runs on a ROCm device only; there is no CPU path")
        self.node_raw_features = torch.from_numpy(node_raw_features.astype(np.float32)).to(device).contiguous()
        self.neighbor_sampler = neighbor_sampler
        self.node_feat_dim = self.node_raw_features.shape[1]
        self.time_feat_dim, self.num_tokens, self.num_layers = time_feat_dim, num_tokens, num_layers
        self.token_dim_expansion_factor, self.channel_dim_expansion_factor = token_dim_expansion_factor, channel_dim_expansion_factor
        self.dropout, self.device = dropout, device
        self.num_channels = 100
        self.time_encoder = TimeEncoder(time_dim=time_feat_dim, parameter_requires_grad=False)      # not trainable in GraphMixer
        self.projection_layer = nn.Linear(time_feat_dim, self.num_channels)
        self.mlp_mixers = nn.ModuleList([MLPMixer(self.num_tokens, self.num_channels, self.token_dim_expansion_factor,
                                                  self.channel_dim_expansion_factor, self.dropout) for _ in range(self.num_layers)])
        self.output_layer = nn.Linear(self.num_channels + self.node_feat_dim, self.node_feat_dim, bias=True)

    def compute_src_dst_node_temporal_embeddings(self, src_node_ids: np.ndarray, dst_node_ids: np.ndarray,
                                                 node_interact_times: np.ndarray, num_neighbors: int = 20, time_gap: int = 2000):
        return (self.compute_node_temporal_embeddings(src_node_ids, node_interact_times, num_neighbors, time_gap),
                self.compute_node_temporal_embeddings(dst_node_ids, node_interact_times, num_neighbors, time_gap))

    def compute_node_temporal_embeddings(self, node_ids: np.ndarray, node_interact_times: np.ndarray,
                                         num_neighbors: int = 20, time_gap: int = 2000):
        dev = self.node_raw_features.device
        node_ids, times = np.asarray(node_ids), np.asarray(node_interact_times, dtype=np.float64)
        sampler = self.neighbor_sampler
        g = getattr(sampler, "graph", None)
        if g is not None and len(node_ids) and (int(node_ids.max()) >= g.num_rows or int(node_ids.min()) < 0):
            raise IndexError("list index out of range")
        ids_d, t_d = ops.h2d([np.ascontiguousarray(node_ids, dtype=np.int32), np.ascontiguousarray(times)], dev)
        recent = sampler.sample_neighbor_strategy == "recent"
        # ---- link encoder (:97-121): time encodings of the sampled neighbors, zero for padded slots, projection, mixers, token mean
        if recent:
            nbr, _, _, dt = g.sample_recent(ids_d, t_d, num_neighbors)
        else:
            nb, _, nt = sampler.get_historical_neighbors(node_ids, times, num_neighbors)
            nbr, dt = ops.h2d([nb.astype(np.int32), (times[:, None] - nt).astype(np.float32)], dev)
        enc = self.time_encoder
        x = seqops.masked_time_encode(dt, nbr, enc.w.weight, enc.w.bias)
        x = linear(x, self.projection_layer.weight, self.projection_layer.bias)
        for mixer in self.mlp_mixers:
            x = mixer(x)
        combined = seqops.segment_mean(x.contiguous(), 0, x.shape[1])
        # ---- node encoder (:123-155): mean over the time_gap most recent neighbors' rows + the node's own row
        if recent:
            agg = g.recent_window_mean(ids_d, t_d, time_gap, self.node_raw_features)
        else:                                   # the reference draws a SECOND random sample of time_gap neighbors here
            nb2, _, _ = sampler.get_historical_neighbors(node_ids, times, time_gap)
            nb2_d, = ops.h2d([nb2.astype(np.int32)], dev)
            rows = ops.gather_rows(self.node_raw_features, nb2_d.reshape(-1)).view(len(node_ids), time_gap, -1)
            mask = (nb2_d > 0).float()
            mask[mask == 0] = -1e10
            agg = torch.mean(rows * torch.softmax(mask, dim=1).unsqueeze(-1), dim=1)
        own = ops.gather_rows(self.node_raw_features, ids_d)
        ol = self.output_layer
        return linear(torch.cat([combined, agg + own], dim=1), ol.weight, ol.bias)

    def set_neighbor_sampler(self, neighbor_sampler: NeighborSampler):
        self.neighbor_sampler = neighbor_sampler
        if self.neighbor_sampler.sample_neighbor_strategy in ['uniform', 'time_interval_aware']:
            assert self.neighbor_sampler.seed is not None
            self.neighbor_sampler.reset_random_state()
