"""TGN backbone (MemoryModel, model_name='TGN') -- drop-in for the reference class (models/MemoryModel.py).

Same constructor, methods, attribute names and state_dict keys (memory_bank.*, the duplicated memory_updater.memory_bank.*,
memory_updater.memory_updater.{weight_ih,...}, embedding_module.*).  Differences are internal:

  * the pending raw messages live in a device table (one row per node: only the LAST message of a node is ever consumed,
    reference :312-320) instead of a python dict of lists of (Tensor, float) -- `memory_bank.node_raw_messages` still reads and
    writes that dict form for the reference's checkpoint code (utils/EarlyStopping.py:85-98);
  * "updated memories of all nodes" (reference :117, a python loop over every node plus a full-table GRU) is one gathered GRU
    over the nodes that have a pending message, on the MFMA GEMM + gate kernel;
  * the per-edge python loops that build and file new messages (:233-278, :425-444) are one kernel and one ordered scatter.
DyRep / JODIE branches of the reference are unreachable from its own CLI (utils/load_configs.py:104-105) and are not provided.
"""
from collections import defaultdict

import numpy as np
import torch
import torch.nn as nn

from .. import engine, ops
from ..utils.utils import NeighborSampler
from .modules import MergeLayer, MultiHeadAttention, TimeEncoder


class MessageAggregator(nn.Module):
    """keeps the reference's module slot (no parameters); aggregation = 'last message wins' is built into the table layout"""

    def __init__(self):
        super().__init__()


class MemoryBank(nn.Module):

    def __init__(self, num_nodes: int, memory_dim: int, message_dim: int = 0):
        super().__init__()
        self.num_nodes = num_nodes
        self.memory_dim = memory_dim
        self.message_dim = message_dim
        # Parameters (requires_grad=False) so that they ride in the state_dict, as in the reference (:347-351)
        self.node_memories = nn.Parameter(torch.zeros((self.num_nodes, self.memory_dim)), requires_grad=False)
        self.node_last_updated_times = nn.Parameter(torch.zeros(self.num_nodes), requires_grad=False)
        self.__init_memory_bank__()
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._refresh_host_mirror())

    def _refresh_host_mirror(self):
        self._h_last = self.node_last_updated_times.detach().cpu().numpy().astype(np.float32)

    # ---- device message table + host mirrors of the scalars needed for the reference's assertion ----------------------
    def _alloc(self):
        dev = self.node_memories.device
        self._msg = torch.zeros((self.num_nodes, self.message_dim), device=dev)
        self._has = np.zeros(self.num_nodes, dtype=bool)
        self._msg_time = np.zeros(self.num_nodes, dtype=np.float64)
        self._h_last = np.zeros(self.num_nodes, dtype=np.float32)

    def __init_memory_bank__(self):
        """zero memories / last-update times and drop all pending messages (start of every epoch, reference :357-364)"""
        self.node_memories.data.zero_()
        self.node_last_updated_times.data.zero_()
        self._alloc()

    def _apply(self, fn, *a, **kw):          # keep the message table on the module's device across .to()/.cuda()
        out = super()._apply(fn, *a, **kw)
        if getattr(self, "_msg", None) is not None and self._msg.device != self.node_memories.device:
            self._msg = self._msg.to(self.node_memories.device)
        return out

    def get_memories(self, node_ids: np.ndarray):
        return self.node_memories[torch.from_numpy(np.asarray(node_ids)).to(self.node_memories.device)]

    def get_node_last_updated_times(self, unique_node_ids: np.ndarray):
        return self.node_last_updated_times[torch.from_numpy(np.asarray(unique_node_ids)).to(self.node_memories.device)]

    def pending_ids(self):
        return np.nonzero(self._has)[0]

    # ---- reference-compatible dict view (checkpointing) ---------------------------------------------------------------------
    @property
    def node_raw_messages(self):
        d = defaultdict(list)
        ids = self.pending_ids()
        if len(ids):
            rows = self._msg[torch.from_numpy(ids).to(self._msg.device)]
            for i, nid in enumerate(ids):
                d[int(nid)].append((rows[i], np.float64(self._msg_time[nid])))
        return d

    @node_raw_messages.setter
    def node_raw_messages(self, value):
        self._alloc()
        for nid, lst in value.items():
            if len(lst):
                msg, ts = lst[-1]
                self._msg[int(nid)] = msg.detach().to(self._msg.device)
                self._has[int(nid)] = True
                self._msg_time[int(nid)] = float(ts)
        self._h_last = self.node_last_updated_times.detach().cpu().numpy().astype(np.float32)

    def backup_memory_bank(self):
        """(memories, last-update times, messages) copies -- reference :383-393; the third item is this class's table form"""
        return (self.node_memories.data.clone(), self.node_last_updated_times.data.clone(),
                (self._msg.clone(), self._has.copy(), self._msg_time.copy(), self._h_last.copy()))

    def reload_memory_bank(self, backup_memory_bank: tuple):
        self.node_memories.data, self.node_last_updated_times.data = backup_memory_bank[0].clone(), backup_memory_bank[1].clone()
        third = backup_memory_bank[2]
        if isinstance(third, dict):
            self.node_raw_messages = third
        else:
            self._msg, self._has, self._msg_time, self._h_last = third[0].clone(), third[1].copy(), third[2].copy(), third[3].copy()

    def detach_memory_bank(self):
        """reference :409-423.  State here is always stored detached (gradient flows only through the same-call GRU)."""
        self.node_memories.detach_()

    def extra_repr(self):
        return 'num_nodes={}, memory_dim={}'.format(self.node_memories.shape[0], self.node_memories.shape[1])


class MemoryUpdater(nn.Module):
    def __init__(self, memory_bank: MemoryBank):
        super().__init__()
        self.memory_bank = memory_bank


class GRUMemoryUpdater(MemoryUpdater):
    def __init__(self, memory_bank: MemoryBank, message_dim: int, memory_dim: int):
        super().__init__(memory_bank)
        self.memory_updater = nn.GRUCell(input_size=message_dim, hidden_size=memory_dim)   # parameter holder (weight_ih, ...)


class GraphAttentionEmbedding(nn.Module):
    """parameter holder with the reference's layout (models/MemoryModel.py:592-630); compute runs in flid_amd.engine"""

    def __init__(self, node_raw_features, edge_raw_features, neighbor_sampler, time_encoder, node_feat_dim, edge_feat_dim,
                 time_feat_dim, num_layers=2, num_heads=2, dropout=0.1):
        super().__init__()
        self.node_raw_features, self.edge_raw_features = node_raw_features, edge_raw_features
        self.neighbor_sampler = neighbor_sampler
        self.time_encoder = time_encoder
        self.node_feat_dim, self.edge_feat_dim, self.time_feat_dim = node_feat_dim, edge_feat_dim, time_feat_dim
        self.num_layers, self.num_heads, self.dropout = num_layers, num_heads, dropout
        self.temporal_conv_layers = nn.ModuleList([
            MultiHeadAttention(node_feat_dim=node_feat_dim, edge_feat_dim=edge_feat_dim, time_feat_dim=time_feat_dim,
                               num_heads=num_heads, dropout=dropout) for _ in range(num_layers)])
        self.merge_layers = nn.ModuleList([
            MergeLayer(input_dim1=node_feat_dim + time_feat_dim, input_dim2=node_feat_dim, hidden_dim=node_feat_dim,
                       output_dim=node_feat_dim) for _ in range(num_layers)])

    def layer_params(self):
        out = []
        for conv, merge in zip(self.temporal_conv_layers, self.merge_layers):
            out += conv.fused_params() + [merge.fc1.weight, merge.fc1.bias, merge.fc2.weight, merge.fc2.bias]
        return out


class _GRURowsFn(torch.autograd.Function):
    """rows of GRU(msg, mem) for the nodes with a pending message (reference :501-528 / nn.GRUCell), on the MFMA GEMM + gate
    kernel.  State and messages are detached inputs (reference :409-423), so backward yields parameter gradients only."""

    @staticmethod
    def forward(ctx, msg_rows, h_rows, w_ih, w_hh, b_ih, b_hh):
        new, gi, gh = ops.gru_cell_fwd(msg_rows, h_rows, w_ih, w_hh, b_ih, b_hh)
        ctx.save_for_backward(msg_rows, h_rows, gi, gh, w_ih, w_hh)
        return new

    @staticmethod
    def backward(ctx, dout):
        msg_rows, h_rows, gi, gh, w_ih, w_hh = ctx.saved_tensors
        dw_ih, dw_hh, db_ih, db_hh = ops.gru_cell_bwd(msg_rows, h_rows, gi, gh, dout, w_ih, w_hh)
        return None, None, dw_ih, dw_hh, db_ih, db_hh


class MemoryModel(torch.nn.Module):

    def __init__(self, node_raw_features: np.ndarray, edge_raw_features: np.ndarray, neighbor_sampler: NeighborSampler,
                 time_feat_dim: int, model_name: str = 'TGN', num_layers: int = 2, num_heads: int = 2, dropout: float = 0.1,
                 src_node_mean_time_shift: float = 0.0, src_node_std_time_shift: float = 1.0, dst_node_mean_time_shift_dst: float = 0.0,
                 dst_node_std_time_shift: float = 1.0, device: str = 'cpu'):
        super().__init__()
        if model_name != 'TGN':
            raise ValueError(f'Not implemented error for model_name {model_name}!')       # reference :74 (DyRep/JODIE unreachable)
        if torch.device(device).type != "cuda":
            raise RuntimeError("flid_amd.MemoryModel runs on a ROCm device only; there is no CPU path")
        self.node_raw_features = torch.from_numpy(node_raw_features.astype(np.float32)).to(device).contiguous()
        self.edge_raw_features = torch.from_numpy(edge_raw_features.astype(np.float32)).to(device).contiguous()
        self.node_feat_dim = self.node_raw_features.shape[1]
        self.edge_feat_dim = self.edge_raw_features.shape[1]
        self.time_feat_dim = time_feat_dim
        self.num_layers, self.num_heads, self.dropout, self.device = num_layers, num_heads, dropout, device
        self.src_node_mean_time_shift, self.src_node_std_time_shift = src_node_mean_time_shift, src_node_std_time_shift
        self.dst_node_mean_time_shift_dst, self.dst_node_std_time_shift = dst_node_mean_time_shift_dst, dst_node_std_time_shift
        self.model_name = model_name
        self.num_nodes = self.node_raw_features.shape[0]
        self.memory_dim = self.node_feat_dim
        self.message_dim = self.memory_dim + self.memory_dim + self.time_feat_dim + self.edge_feat_dim
        self.time_encoder = TimeEncoder(time_dim=time_feat_dim)
        self.message_aggregator = MessageAggregator()
        self.memory_bank = MemoryBank(num_nodes=self.num_nodes, memory_dim=self.memory_dim, message_dim=self.message_dim)
        self.memory_updater = GRUMemoryUpdater(memory_bank=self.memory_bank, message_dim=self.message_dim, memory_dim=self.memory_dim)
        self.embedding_module = GraphAttentionEmbedding(self.node_raw_features, self.edge_raw_features, neighbor_sampler,
                                                        self.time_encoder, self.node_feat_dim, self.edge_feat_dim,
                                                        self.time_feat_dim, num_layers, num_heads, dropout)
        self.to(device)

    # ------------------------------------------------------------------------------------------------------------------
    def _check_not_in_the_past(self, ids):
        bank = self.memory_bank
        if len(ids):
            ok = bool(np.all(bank._h_last[ids] <= bank._msg_time[ids].astype(np.float32)))
            assert ok, "Trying to update memory to time in the past!"                      # reference :485-486, :515-516

    def _updated_table(self):
        """base table (N, D) = raw features + memories with every pending message applied (not persisted): reference :117,
        :191-212, :654-655.  Also returns the pending node ids (host, ascending) and their freshly computed memory rows."""
        bank, gru = self.memory_bank, self.memory_updater.memory_updater
        ids = bank.pending_ids()
        self._check_not_in_the_past(ids)
        base = bank.node_memories.detach() + self.node_raw_features
        new_rows = None
        if len(ids):
            dev = self.node_raw_features.device
            idx, = ops.h2d([ids.astype(np.int32)], dev)
            new_rows = _GRURowsFn.apply(ops.gather_rows(bank._msg, idx), ops.gather_rows(bank.node_memories.detach(), idx),
                                        gru.weight_ih, gru.weight_hh, gru.bias_ih, gru.bias_hh)
            base = base.index_copy(0, idx.long(), new_rows + ops.gather_rows(self.node_raw_features, idx))
        return base, ids, new_rows

    def compute_src_dst_node_temporal_embeddings(self, src_node_ids: np.ndarray, dst_node_ids: np.ndarray, node_interact_times: np.ndarray,
                                                 edge_ids: np.ndarray, edges_are_positive: bool = True, num_neighbors: int = 20):
        return self._step(src_node_ids, dst_node_ids, node_interact_times, edge_ids, edges_are_positive, num_neighbors, None)

    def compute_shard_embeddings_and_advance(self, src_node_ids, dst_node_ids, node_interact_times, edge_ids, shard,
                                             edges_are_positive: bool = True, num_neighbors: int = 20):
        """Data-parallel form (one process per GPU): embed only edges [lo, hi) = `shard` of the global batch, but advance the
        memory / pending-message state with the WHOLE batch.  Within a batch every root reads the same pre-batch state and the
        update depends only on that state and the batch's (src, dst, t, eid), all of which are replicated -- so every rank
        computes the identical update locally and the replicas stay bit-identical with no message exchange; the only
        collective of a TGN step is the gradient all-reduce (flid_amd.dist.GradAllReducer)."""
        return self._step(src_node_ids, dst_node_ids, node_interact_times, edge_ids, edges_are_positive, num_neighbors, shard)

    def _step(self, src_node_ids, dst_node_ids, node_interact_times, edge_ids, edges_are_positive, num_neighbors, shard):
        src_node_ids = np.asarray(src_node_ids)
        dst_node_ids = np.asarray(dst_node_ids)
        node_interact_times = np.asarray(node_interact_times, dtype=np.float64)
        node_ids = np.concatenate([src_node_ids, dst_node_ids])
        bank = self.memory_bank
        base, pend_ids, new_rows = self._updated_table()                                   # reference :117
        lo, hi = (0, len(src_node_ids)) if shard is None else shard
        emb_ids = np.concatenate([src_node_ids[lo:hi], dst_node_ids[lo:hi]])
        emb_t = np.concatenate([node_interact_times[lo:hi], node_interact_times[lo:hi]])
        smp = self.embedding_module.neighbor_sampler
        emb = engine.embed(smp.graph, base, self.edge_raw_features,
                           self.time_encoder.w.weight, self.time_encoder.w.bias, self.embedding_module.layer_params(),
                           emb_ids, emb_t, num_neighbors, self.num_layers,
                           self.num_heads, self.dropout, self.training, table_requires_grad=torch.is_grad_enabled(),
                           host_sampler=(None if smp.sample_neighbor_strategy == "recent" else smp),      # random strategies: host RNG
                           groups=None)
        src_emb, dst_emb = engine.split_rows(emb, hi - lo)
        if edges_are_positive:
            assert edge_ids is not None
            dev = self.node_raw_features.device
            # (1) persist the GRU update for the batch nodes that had a pending message (reference :158, :472-499).
            #     Same inputs, same kernel as the view above: its rows are reused.
            uniq = np.unique(node_ids)
            upd = uniq[bank._has[uniq]]
            n = len(src_node_ids)
            rev_nodes = node_ids[::-1]
            u, first_rev = np.unique(rev_nodes, return_index=True)
            last_pos = (2 * n - 1 - first_rev).astype(np.int64)
            if len(upd):
                self._check_not_in_the_past(upd)
            # every host array of the state update goes over in one pinned, asynchronous copy
            ui, where, new_t, a, b, t32, e, u_dev, last_dev = ops.h2d(
                [upd.astype(np.int64), np.searchsorted(pend_ids, upd).astype(np.int64), bank._msg_time[upd].astype(np.float32),
                 np.concatenate([src_node_ids, dst_node_ids]).astype(np.int32), np.concatenate([dst_node_ids, src_node_ids]).astype(np.int32),
                 np.concatenate([node_interact_times, node_interact_times]).astype(np.float32),
                 np.concatenate([edge_ids, edge_ids]).astype(np.int32), u.astype(np.int64), last_pos], dev)
            if len(upd):
                with torch.no_grad():
                    bank.node_memories.data.index_copy_(0, ui, new_rows.detach()[where])     # `where` = rows of the view's GRU output
                    bank.node_last_updated_times.data.index_copy_(0, ui, new_t)
                bank._h_last[upd] = bank._msg_time[upd].astype(np.float32)
            # (2) clear the batch nodes' pending messages (:162)
            bank._has[uniq] = False
            # (3) new raw messages from the post-update state, source role then destination role (:165-180); per node the
            #     last one in that order is the one that will ever be read
            with torch.no_grad():
                msgs = ops.build_messages(bank.node_memories.data, bank.node_last_updated_times.data, a, b, t32,
                                          self.edge_raw_features, e, self.time_encoder.w.weight.detach().reshape(-1),
                                          self.time_encoder.w.bias.detach())
                bank._msg.index_copy_(0, u_dev, msgs[last_dev])
            bank._has[u] = True
            bank._msg_time[u] = node_interact_times[last_pos % n]
        return src_emb, dst_emb

    # kept for callers / tests that poke the reference's helper methods
    def get_updated_memories(self, node_ids=None, node_raw_messages=None):
        base, _, _ = self._updated_table()
        lu = self.memory_bank.node_last_updated_times.data.clone()
        ids = self.memory_bank.pending_ids()
        if len(ids):
            lu[torch.from_numpy(ids).to(lu.device)] = torch.from_numpy(self.memory_bank._msg_time[ids].astype(np.float32)).to(lu.device)
        return base - self.node_raw_features, lu

    def set_neighbor_sampler(self, neighbor_sampler: NeighborSampler):
        assert self.model_name in ['TGN', 'DyRep'], f'Neighbor sampler is not defined in model {self.model_name}!'
        self.embedding_module.neighbor_sampler = neighbor_sampler
        if self.embedding_module.neighbor_sampler.sample_neighbor_strategy in ['uniform', 'time_interval_aware']:
            assert self.embedding_module.neighbor_sampler.seed is not None
            self.embedding_module.neighbor_sampler.reset_random_state()


def compute_src_dst_node_time_shifts(src_node_ids: np.ndarray, dst_node_ids: np.ndarray, node_interact_times: np.ndarray):
    """mean / std of per-node inter-event times (reference :718-751; used by JODIE only, computed and ignored for TGN)"""
    def shifts(ids):
        last, out = {}, np.empty(len(ids), dtype=np.float64)
        for k, (v, t) in enumerate(zip(ids, node_interact_times)):
            out[k] = t - last.get(v, 0)
            last[v] = t
        return out
    s, d = shifts(src_node_ids), shifts(dst_node_ids)
    return np.mean(s), np.std(s), np.mean(d), np.std(d)
