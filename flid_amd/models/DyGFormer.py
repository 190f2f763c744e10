"""DyGFormer backbone -- drop-in for the reference class (models/DyGFormer.py): same constructor, forward surface,
parameter names / shapes.  Sequences are built on the device (tg_first_hop_window, tg_cooccurrence), every product runs on
the fp32 MFMA GEMM (plain, or two-level batched for the per-(sequence, head) attention products)."""
import numpy as np
import torch
import torch.nn as nn

from .. import ops, seqops
from ..utils.utils import NeighborSampler
from .modules import TimeEncoder, linear


class NeighborCooccurrenceEncoder(nn.Module):
    """parameter holder (models/DyGFormer.py:320-335): Linear(1, C) -> ReLU -> Linear(C, C) applied to each of the two counts"""

    def __init__(self, neighbor_co_occurrence_feat_dim: int, device: str = 'cpu'):
        super().__init__()
        self.neighbor_co_occurrence_feat_dim = neighbor_co_occurrence_feat_dim
        self.device = device
        self.neighbor_co_occurrence_encode_layer = nn.Sequential(
            nn.Linear(in_features=1, out_features=neighbor_co_occurrence_feat_dim), nn.ReLU(),
            nn.Linear(in_features=neighbor_co_occurrence_feat_dim, out_features=neighbor_co_occurrence_feat_dim))

    def encode(self, counts: torch.Tensor):
        """(B, W, 2) counts -> (B, W, C): MLP on each count, summed over the two (:409-411)"""
        l0, l2 = self.neighbor_co_occurrence_encode_layer[0], self.neighbor_co_occurrence_encode_layer[2]
        h = linear(counts.unsqueeze(-1), l0.weight, l0.bias, relu=True)
        return linear(h, l2.weight, l2.bias).sum(dim=2)


class TransformerEncoder(nn.Module):
    """pre-LN block (models/DyGFormer.py:418-461); nn.MultiheadAttention is kept as the parameter holder"""

    def __init__(self, attention_dim: int, num_heads: int, dropout: float = 0.1):
        super().__init__()
        self.multi_head_attention = nn.MultiheadAttention(embed_dim=attention_dim, num_heads=num_heads, dropout=dropout)
        self.num_heads, self.p = num_heads, dropout
        self.dropout = nn.Dropout(dropout)
        self.linear_layers = nn.ModuleList([nn.Linear(attention_dim, 4 * attention_dim), nn.Linear(4 * attention_dim, attention_dim)])
        self.norm_layers = nn.ModuleList([nn.LayerNorm(attention_dim), nn.LayerNorm(attention_dim)])

    FUSED = True      # one autograd node per block (seqops.encoder_block); False: the op-by-op form below (A/B, tests)

    def forward(self, inputs: torch.Tensor):
        mha, tr, p = self.multi_head_attention, self.training, self.p
        if self.FUSED and inputs.dim() == 3:
            return seqops.encoder_block(inputs, self.norm_layers[0], mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight,
                                        mha.out_proj.bias, self.norm_layers[1], self.linear_layers[0], self.linear_layers[1],
                                        self.num_heads, p, tr)
        y = seqops.layer_norm(inputs, self.norm_layers[0].weight, self.norm_layers[0].bias)
        qkv = linear(y, mha.in_proj_weight, mha.in_proj_bias)
        att = seqops.self_attention(qkv, self.num_heads, p, tr)
        att = linear(att, mha.out_proj.weight, mha.out_proj.bias)
        outputs = inputs + seqops.dropout(att, p, tr)
        y = seqops.layer_norm(outputs, self.norm_layers[1].weight, self.norm_layers[1].bias)
        h = seqops.gelu(linear(y, self.linear_layers[0].weight, self.linear_layers[0].bias))
        h = linear(seqops.dropout(h, p, tr), self.linear_layers[1].weight, self.linear_layers[1].bias)
        return outputs + seqops.dropout(h, p, tr)


class DyGFormer(nn.Module):
    FUSED_PROJECTION = True       # patch size 1: the four channel projections as one block-diagonal product (False: per channel and side)

    def __init__(self, node_raw_features: np.ndarray, edge_raw_features: np.ndarray, neighbor_sampler: NeighborSampler,
                 time_feat_dim: int, channel_embedding_dim: int, patch_size: int = 1, num_layers: int = 2, num_heads: int = 2,
                 dropout: float = 0.1, max_input_sequence_length: int = 512, device: str = 'cpu'):
        super().__init__()
        if torch.device(device).type != "cuda":
            raise RuntimeError("flid_amd.DyGFormer runs on a ROCm device only; there is no CPU path")
        self.node_raw_features = torch.from_numpy(node_raw_features.astype(np.float32)).to(device).contiguous()
        self.edge_raw_features = torch.from_numpy(edge_raw_features.astype(np.float32)).to(device).contiguous()
        self.neighbor_sampler = neighbor_sampler
        self.node_feat_dim, self.edge_feat_dim = self.node_raw_features.shape[1], self.edge_raw_features.shape[1]
        self.time_feat_dim, self.channel_embedding_dim, self.patch_size = time_feat_dim, channel_embedding_dim, patch_size
        self.num_layers, self.num_heads, self.dropout = num_layers, num_heads, dropout
        self.max_input_sequence_length, self.device = max_input_sequence_length, device
        self.time_encoder = TimeEncoder(time_dim=time_feat_dim)
        self.neighbor_co_occurrence_feat_dim = self.channel_embedding_dim
        self.neighbor_co_occurrence_encoder = NeighborCooccurrenceEncoder(self.neighbor_co_occurrence_feat_dim, device=device)
        self.projection_layer = nn.ModuleDict({
            'node': nn.Linear(self.patch_size * self.node_feat_dim, self.channel_embedding_dim, bias=True),
            'edge': nn.Linear(self.patch_size * self.edge_feat_dim, self.channel_embedding_dim, bias=True),
            'time': nn.Linear(self.patch_size * self.time_feat_dim, self.channel_embedding_dim, bias=True),
            'neighbor_co_occurrence': nn.Linear(self.patch_size * self.neighbor_co_occurrence_feat_dim, self.channel_embedding_dim, bias=True)})
        self.num_channels = 4
        self.transformers = nn.ModuleList([TransformerEncoder(self.num_channels * self.channel_embedding_dim, self.num_heads, self.dropout)
                                           for _ in range(self.num_layers)])
        self.output_layer = nn.Linear(self.num_channels * self.channel_embedding_dim, self.node_feat_dim, bias=True)

    # ----------------------------------------------------------------------------------------------------------------------
    def _native_param_order(self):
        """the parameter tensors in the order of tg_dyg_cfg.poff (include/flid_tg.h)"""
        co = self.neighbor_co_occurrence_encoder.neighbor_co_occurrence_encode_layer
        out = [self.time_encoder.w.weight, self.time_encoder.w.bias, co[0].weight, co[0].bias, co[2].weight, co[2].bias]
        for name in ("node", "edge", "time", "neighbor_co_occurrence"):
            out += [self.projection_layer[name].weight, self.projection_layer[name].bias]
        for blk in self.transformers:
            mha = blk.multi_head_attention
            out += [mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias,
                    blk.norm_layers[0].weight, blk.norm_layers[0].bias, blk.norm_layers[1].weight, blk.norm_layers[1].bias,
                    blk.linear_layers[0].weight, blk.linear_layers[0].bias, blk.linear_layers[1].weight, blk.linear_layers[1].bias]
        return out + [self.output_layer.weight, self.output_layer.bias]

    def flatten_parameters(self) -> nn.Parameter:
        """Opt-in (not in the reference): re-home every parameter in ONE flat nn.Parameter and return it -- the named parameters stay
        (same state_dict keys) as views of the flat buffer with requires_grad off; the trainer hands the returned parameter to its
        optimizer (flid_amd.optim.FlatAdam).  As TGAT.flatten_parameters; needed by enable_native_step()."""
        from .. import engine
        params = self._native_param_order()
        assert len(params) == len(list(self.parameters()))
        offs, total = engine.block_layout(params)
        flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
        views = []
        with torch.no_grad():
            for o, p in zip(offs, params):
                v = flat[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                p.requires_grad_(False)
                views.append(v)
        flat_param = nn.Parameter(flat)
        self._flat_pack = [flat_param, views]          # a list: nn.Module must not register it (state_dict stays the reference's)
        return flat_param

    def enable_native_step(self, max_batch_edges: int):
        """Opt-in (not in the reference; needs flatten_parameters()): train_step goes through ONE native object
        (flid_amd.stepper.DygStepper, csrc/tg_dyg.hip) -- forward and backward (+ update) a C call each, every launch of a step issued by
        the library out of a pre-sized arena.  Patch size 1, two sides of at most 32 positions; raises for other shapes (the autograd
        path takes those)."""
        from ..stepper import DygStepper
        self._stepper = DygStepper(self, max_batch_edges)
        return self._stepper

    def train_step(self, src_node_ids: np.ndarray, dst_node_ids: np.ndarray, node_interact_times: np.ndarray, loss_fn, optimizer=None):
        """One training step without an autograd graph: forward of compute_src_dst_node_temporal_embeddings, `loss_fn(emb)` ->
        (loss, d loss / d emb) on the (2 B, dn) block [source rows | destination rows], backward into the flat parameter's .grad and, with
        optimizer (a FlatAdam over the flat parameter), its update -- what the trainers' loss.backward(); optimizer.step() do
        (PTCL/M_step.py:297-325).  Returns (embeddings, loss)."""
        st = getattr(self, "_stepper", None)
        if st is None:
            raise RuntimeError("DyGFormer.train_step: call flatten_parameters() and enable_native_step() first")
        return st.step(src_node_ids, dst_node_ids, node_interact_times, loss_fn, optimizer=optimizer)

    def _windows(self, ids_dev, t_dev):
        """device restatement of get_all_first_hop_neighbors + pad_sequences (utils/utils.py:254-273, DyGFormer.py:196-245)"""
        P, L = self.patch_size, self.max_input_sequence_length
        wmax = (L + P - 1) // P * P
        return self.neighbor_sampler.graph.first_hop_window(ids_dev, t_dev, L, wmax)

    def compute_src_dst_node_temporal_embeddings(self, src_node_ids: np.ndarray, dst_node_ids: np.ndarray, node_interact_times: np.ndarray):
        dev = self.node_raw_features.device
        g = self.neighbor_sampler.graph
        src_node_ids, dst_node_ids = np.asarray(src_node_ids), np.asarray(dst_node_ids)
        for ids in (src_node_ids, dst_node_ids):
            if len(ids) and (int(ids.max()) >= g.num_rows or int(ids.min()) < 0):
                raise IndexError("list index out of range")
        B, P = len(src_node_ids), self.patch_size
        t_dev, src_dev, dst_dev = ops.h2d([np.ascontiguousarray(node_interact_times, dtype=np.float64),
                                           np.ascontiguousarray(src_node_ids, dtype=np.int32),
                                           np.ascontiguousarray(dst_node_ids, dtype=np.int32)], dev)
        sides = [self._windows(ids_dev, t_dev) for ids_dev in (src_dev, dst_dev)]
        # the reference pads every side to ITS OWN longest sequence of the batch (+ the node itself, rounded up to a patch
        # multiple): the unmasked transformer sees the padded positions, so the width is part of the result.  The lengths come from
        # the host copy of the adjacency (binary searches in C, ~10 us): reading the device kernel's lengths back made the host wait
        # for the whole previous step (1.1 of 5.5 ms)
        L = self.max_input_sequence_length
        lens = [int(np.minimum(g.count_before_host(ids, node_interact_times), L - 1).max()) + 1 for ids in (src_node_ids, dst_node_ids)] if B else [1, 1]
        widths = [(int(l) + P - 1) // P * P for l in lens]
        seqs = [tuple(t[:, :w].contiguous() for t in side[:3]) for side, w in zip(sides, widths)]
        counts = ops.cooccurrence(seqs[0][0], seqs[1][0])                                 # DyGFormer.py:104-106
        num_edges_rows = self.edge_raw_features.shape[0]
        enc = self.time_encoder
        if self.FUSED_PROJECTION and P == 1 and B > 0:
            # both sides' positions in the order the transformer reads them ([source positions | destination positions] per edge), all
            # four channels from ONE product against the block-diagonal projection weight (seqops.patch_projection)
            nbr = torch.cat([seqs[0][0], seqs[1][0]], dim=1).contiguous()
            eid = torch.cat([seqs[0][1], seqs[1][1]], dim=1)
            tt = torch.cat([seqs[0][2], seqs[1][2]], dim=1)
            cnt = torch.cat([counts[0], counts[1]], dim=1)
            S_tot = nbr.shape[1]
            n_pos = B * S_tot
            dn, de, T = self.node_feat_dim, self.edge_feat_dim, self.time_feat_dim
            Kx = dn + de + T + self.neighbor_co_occurrence_feat_dim
            Kp = (Kx + 3) // 4 * 4
            X = torch.empty((n_pos, Kp), device=dev)
            if Kp > Kx:
                X[:, Kx:] = 0.0
            ops.gather_rows(self.node_raw_features, nbr.reshape(-1), out=X[:, 0:dn])                       # :259
            eidx = torch.remainder(eid.reshape(-1) - 1, num_edges_rows).to(torch.int32)                     # :261 (edge_ids - 1 wraps)
            ops.gather_rows(self.edge_raw_features, eidx, out=X[:, dn:dn + de])
            dt = (t_dev.unsqueeze(1) - tt.double()).float()
            tf = seqops.masked_time_encode(dt, nbr, enc.w.weight, enc.w.bias)
            cf = self.neighbor_co_occurrence_encoder.encode(cnt)
            x = seqops.patch_projection(X, tf, cf, self.projection_layer).view(B, S_tot, -1)
            ns = widths[0]
            for block in self.transformers:
                x = block(x)
            nt = x.shape[1]
            s_out, d_out = seqops.segment_mean(x, 0, ns), seqops.segment_mean(x, ns, nt)
            return (linear(s_out, self.output_layer.weight, self.output_layer.bias),
                    linear(d_out, self.output_layer.weight, self.output_layer.bias))
        chans_per_side = []
        for (nbr, eid, tt), cnt, w in zip(seqs, counts, widths):
            flat = nbr.reshape(-1)
            nf = ops.gather_rows(self.node_raw_features, flat).reshape(B, w, -1)           # :259
            # FLiD-specific `edge_ids - 1` gather (:261): slot / edge id 0 wraps to the LAST edge row
            eidx = torch.remainder(eid.reshape(-1) - 1, num_edges_rows).to(torch.int32)
            ef = ops.gather_rows(self.edge_raw_features, eidx).reshape(B, w, -1)
            dt = (t_dev.unsqueeze(1) - tt.double()).float()                                # :263 float64 - float32 -> float32
            tf = seqops.masked_time_encode(dt, nbr, enc.w.weight, enc.w.bias)              # :263-266
            cf = self.neighbor_co_occurrence_encoder.encode(cnt)
            chans = []
            for name, x in (("node", nf), ("edge", ef), ("time", tf), ("neighbor_co_occurrence", cf)):
                lin = self.projection_layer[name]
                chans.append(linear(x.reshape(B, w // P, -1), lin.weight, lin.bias))       # get_patches + projection (:270-306, :148-157)
            chans_per_side.append(torch.stack(chans, dim=2).reshape(B, w // P, -1))
        ns = chans_per_side[0].shape[1]
        x = torch.cat(chans_per_side, dim=1).contiguous()                                  # :164-174
        for block in self.transformers:
            x = block(x)
        nt = x.shape[1]
        s_out, d_out = seqops.segment_mean(x, 0, ns), seqops.segment_mean(x, ns, nt)       # :185-187
        return (linear(s_out, self.output_layer.weight, self.output_layer.bias),
                linear(d_out, self.output_layer.weight, self.output_layer.bias))

    def set_neighbor_sampler(self, neighbor_sampler: NeighborSampler):
        self.neighbor_sampler = neighbor_sampler
        if self.neighbor_sampler.sample_neighbor_strategy in ['uniform', 'time_interval_aware']:
            assert self.neighbor_sampler.seed is not None
            self.neighbor_sampler.reset_random_state()
        st = getattr(self, "_stepper", None)          # the native step follows the sampler (PTCL/M_step.py:34, :200)
        if st is not None:
            st.rebind(neighbor_sampler.graph)
