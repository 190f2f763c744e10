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
from .._lib import TgShapeNotCovered
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
        # device copies of the flags / message times (float32, what the memory's last-update time becomes) for the lazy update path,
        # the all -1 workspace of tg_msg_scatter_last, and a latch for the reference's "time in the past" assertion
        self._has_dev = torch.zeros(self.num_nodes, dtype=torch.int32, device=dev)
        self._msg_time_dev = torch.zeros(self.num_nodes, dtype=torch.float32, device=dev)
        self._last_idx_ws = torch.full((self.num_nodes,), -1, dtype=torch.int32, device=dev)
        self._past_violation = False

    def __init_memory_bank__(self):
        """zero memories / last-update times and drop all pending messages (start of every epoch, reference :357-364)"""
        self.node_memories.data.zero_()
        self.node_last_updated_times.data.zero_()
        self._alloc()

    def _apply(self, fn, *a, **kw):          # keep the message table on the module's device across .to()/.cuda()
        out = super()._apply(fn, *a, **kw)
        if getattr(self, "_msg", None) is not None and self._msg.device != self.node_memories.device:
            dev = self.node_memories.device
            self._msg, self._has_dev = self._msg.to(dev), self._has_dev.to(dev)
            self._msg_time_dev, self._last_idx_ws = self._msg_time_dev.to(dev), self._last_idx_ws.to(dev)
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
        self._sync_device_flags()

    def _sync_device_flags(self):
        dev = self._msg.device
        self._has_dev = torch.from_numpy(self._has.astype(np.int32)).to(dev)
        self._msg_time_dev = torch.from_numpy(self._msg_time.astype(np.float32)).to(dev)

    def backup_memory_bank(self):
        """(memories, last-update times, messages) copies -- reference :383-393; the third item is this class's table form"""
        return (self.node_memories.data.clone(), self.node_last_updated_times.data.clone(),
                (self._msg.clone(), self._has.copy(), self._msg_time.copy(), self._h_last.copy(), self._past_violation))

    def reload_memory_bank(self, backup_memory_bank: tuple):
        self.node_memories.data, self.node_last_updated_times.data = backup_memory_bank[0].clone(), backup_memory_bank[1].clone()
        third = backup_memory_bank[2]
        if isinstance(third, dict):
            self.node_raw_messages = third
        else:
            self._msg, self._has, self._msg_time, self._h_last = third[0].clone(), third[1].copy(), third[2].copy(), third[3].copy()
            self._past_violation = bool(third[4]) if len(third) > 4 else False
            self._sync_device_flags()

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


class _GRURowsMaskedFn(torch.autograd.Function):
    """lazy form: rows of the touched nodes; out = GRU(msg, mem) where the node has a pending message, mem otherwise.  Parameter
    gradients (both weight matrices and both biases) leave in ONE grouped launch (tg_wgrad_group)."""

    @staticmethod
    def forward(ctx, msg_rows, h_rows, has, w_ih, w_hh, b_ih, b_hh):
        new, gi, gh = ops.gru_cell_fwd(msg_rows, h_rows, w_ih, w_hh, b_ih, b_hh)
        mask = has.view(-1, 1)
        out = torch.where(mask > 0, new, h_rows)
        ctx.save_for_backward(msg_rows, h_rows, gi, gh, mask, w_ih, w_hh)
        return out

    @staticmethod
    def backward(ctx, dout):
        msg_rows, h_rows, gi, gh, mask, w_ih, w_hh = ctx.saved_tensors
        n, d = h_rows.shape
        dgi, dgh = torch.empty_like(gi), torch.empty_like(gh)
        dm = (dout * mask.to(dout.dtype)).contiguous()
        from .._lib import check, lib
        check(lib().tg_gru_gates_bwd(ops._p(gi), ops._p(gh), ops._p(h_rows), ops._p(dm), n, d, ops._p(dgi), ops._p(dgh), ops._p(None), ops._stream()),
              "tg_gru_gates_bwd")
        dw_ih, dw_hh = torch.zeros_like(w_ih), torch.zeros_like(w_hh)
        db_ih = torch.zeros(w_ih.shape[0], device=dout.device)
        db_hh = torch.zeros(w_hh.shape[0], device=dout.device)
        try:
            ops.wgrad_group([(dgi, msg_rows, dw_ih, db_ih), (dgh, h_rows, dw_hh, db_hh)])
        except TgShapeNotCovered:              # widths that are not multiples of 4: one exact product + one column sum each
            ops.gemm(dgi, msg_rows, dw_ih, ta=True)
            ops.gemm(dgh, h_rows, dw_hh, ta=True)
            db_ih, db_hh = ops.colsum(dgi), ops.colsum(dgh)
        return None, None, None, dw_ih, dw_hh, db_ih, db_hh


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

    LAZY = True      # one-layer 'recent' calls update only the memory rows the call touches (False: the full-table path below)

    def _step(self, src_node_ids, dst_node_ids, node_interact_times, edge_ids, edges_are_positive, num_neighbors, shard):
        smp = self.embedding_module.neighbor_sampler
        if isinstance(src_node_ids, dict) or (self.LAZY and self.num_layers == 1 and smp.sample_neighbor_strategy == "recent" and len(src_node_ids) > 0):
            return self._step_lazy(src_node_ids, dst_node_ids, node_interact_times, edge_ids, edges_are_positive, num_neighbors, shard)
        return self._step_full(src_node_ids, dst_node_ids, node_interact_times, edge_ids, edges_are_positive, num_neighbors, shard)

    # ---- flat-parameter mode + fused step (not in the reference) ------------------------------------------------------
    def _trainable(self):
        g = self.memory_updater.memory_updater
        return [self.time_encoder.w.weight, self.time_encoder.w.bias] + self.embedding_module.layer_params() + \
               [g.weight_ih, g.weight_hh, g.bias_ih, g.bias_hh]

    def flatten_parameters(self) -> nn.Parameter:
        """Opt-in, as TGAT.flatten_parameters(): every trainable tensor (time encoder, attention + merge layers, GRU) becomes a view
        of ONE flat nn.Parameter, returned for the optimizer; state_dict keys are unchanged.  Needed by train_step()."""
        params = self._trainable()
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
        self._flat_pack = [flat_param, views]
        return flat_param

    def enable_native_step(self, max_batch_edges: int, num_neighbors: int = 20, slots: int = 4):
        """Opt-in (not in the reference; needs flatten_parameters()): prepare_batch_begin(edge_ids=...) / prepare_batch_finish / train_step go
        through ONE native object (flid_amd.stepper.Stepper, csrc/tg_step.hip) -- the graph-only preparation, the forward (GRU on the
        touched rows + the layer) and the backward + state advance + Adam are a C call each, every launch issued by the library out of a
        pre-sized arena.  A prepared batch of this mode is consumed by train_step."""
        from ..stepper import Stepper
        self._stepper = Stepper(self, 2 * int(max_batch_edges), num_neighbors, slots)
        return self._stepper

    def train_step(self, prepared, edge_ids, loss_fn, num_neighbors: int = 20, optimizer=None, edges_are_positive: bool = True,
                   accumulate: bool = False, more: bool = False, grad_ready=None):
        """Fused-trainer step on a prepared (prepare_batch_begin / _finish) POSITIVE batch: updated memory rows of the touched
        nodes, embeddings, `loss_fn(emb) -> (loss, d_emb)` on the detached (2 B, D) block [src rows | dst rows], backward into the
        flat parameter's .grad (added, as autograd accumulates), and the state advance (persist, new messages, last-message-wins
        scatter) -- no autograd graph, ~40 launches.  Same kernels and numbers as compute_src_dst_node_temporal_embeddings +
        loss.backward()."""
        from .._lib import check, lib
        from ..stepper import StepJob
        if isinstance(prepared, StepJob):
            # (optimizer: a FlatAdam over the flat parameter -- its update is issued behind the state advance, in the backward's call)
            assert prepared.finished and prepared.k == int(num_neighbors) and prepared.stepper is getattr(self, "_stepper", None), \
                "prepared by another stepper / not finished"
            # The warm-up's link-prediction step (PTCL/EM_warmup.py:159-175, :212-231) embeds the negative pairs first (edge_ids None,
            # edges_are_positive False: no state advance), then the positive ones, and backpropagates ONE loss over both: two calls here,
            # train_step(neg, ..., edges_are_positive=False, more=True) then train_step(pos, ..., accumulate=True, optimizer=opt) -- the
            # loss is a mean over independent samples, so each call backpropagates its own samples' share into the same gradient block
            # grad_ready(segment): the layer's gradient block, handed over while the GRU's backward and the state advance are being issued
            # (flid_amd.dist.GradAllReducer.segment_ready starts its all-reduce there; reducer.finish() reduces the rest)
            return prepared.stepper.step_tgn(prepared, loss_fn, positive=edges_are_positive, optimizer=optimizer, accumulate=accumulate, more=more,
                                             grad_ready=grad_ready)
        if optimizer is not None or not edges_are_positive or accumulate or more or grad_ready is not None:
            raise NotImplementedError("train_step(optimizer / edges_are_positive=False / accumulate / more) are the native step's (enable_native_step())")
        flat = getattr(self, "_flat_pack", None)
        if flat is None:
            raise RuntimeError("MemoryModel.train_step needs the flat-parameter mode: call flatten_parameters() first")
        if self.num_layers != 1 or self.num_heads > 2 or not engine.NATIVE:
            raise NotImplementedError("train_step covers the one-layer TGN of the reference's CLI (1 or 2 heads)")
        job = prepared
        assert isinstance(job, dict) and job.get("finished") and job["k"] == int(num_neighbors), "pass a finished prepare_batch_begin() job"
        bank, dev = self.memory_bank, self.node_raw_features.device
        views = flat[1]
        te_w, te_b, layer_params, (w_ih, w_hh, b_ih, b_hh) = views[0], views[1], views[2:13], views[13:17]
        torch.cuda.current_stream().wait_event(job["ready"])
        n, m, k = job["n"], job["m"], job["k"]
        S, uniq, rowmap, batch_d = job["S"], job["uniq"], job["rowmap"], job["batch_d"]
        assert not bank._past_violation, "Trying to update memory to time in the past!"
        with torch.no_grad():
            mem = bank.node_memories.data
            pending = bool(bank._has.any())
            # ONE call: gathers of the touched rows (memory, pending messages), GRU products + gates, selection by has-message, and the
            # layer-0 table `updated memory + raw features` (one allocation for everything it leaves)
            U, D, MD = uniq.numel(), self.memory_dim, self.message_dim
            sizes = (U * D, U * MD if pending else 0, U * 3 * D if pending else 0, U * 3 * D if pending else 0, U * D, U * D)
            offs_, tot_ = [], 0
            for sz in sizes:
                offs_.append(tot_)
                tot_ += (sz + 3) // 4 * 4
            buf = torch.empty(max(tot_, 4), dtype=torch.float32, device=dev)
            bp = buf.data_ptr()
            h_rows, msg_rows, gi, gh, rows, base = (buf[o:o + sz].view(U, -1) if sz else None for o, sz in zip(offs_, sizes))
            ptr = lambda i: (bp + 4 * offs_[i]) if sizes[i] else None
            check(lib().tg_tgn_rows_fwd(mem.data_ptr(), mem.stride(0), bank._msg.data_ptr(), bank._msg.stride(0),
                                        self.node_raw_features.data_ptr(), self.node_raw_features.stride(0), uniq.data_ptr(), U,
                                        bank._has_dev.data_ptr(), D, MD, w_ih.data_ptr(), w_hh.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr(),
                                        int(pending), ptr(0), ptr(1), ptr(2), ptr(3), ptr(4), ptr(5), ops._stream()), "tg_tgn_rows_fwd")
            fr = engine.Frontier(counts=[m], ids_all=job["row_roots"], S=S, child=None, pad_rows=[], feat_idx0=job["row_nbrs"], pad_row0=job["pad"])
            cfg = dict(n=m, k=k, num_layers=1, num_heads=self.num_heads, dropout=float(self.dropout), training=bool(self.training),
                       edge_table=self.edge_raw_features, table_grad=pending)
            emb, saved = engine._native_forward(cfg, fr, base, te_w, te_b, layer_params)
        loss, d_emb = loss_fn(emb)
        with torch.no_grad():
            gru_offs, gru_len = engine.block_layout([w_ih, w_hh, b_ih, b_hh])
            d_table, zeroed, offs, npar = engine._native_backward(cfg, fr, base, te_w, te_b, layer_params, saved, d_emb, extra_floats=gru_len)
            if pending:
                tail = zeroed[npar:npar + gru_len]
                gv = [tail[o:o + t.numel()].view(t.shape) for o, t in zip(gru_offs, (w_ih, w_hh, b_ih, b_hh))]
                dgi, dgh = torch.empty_like(gi), torch.empty_like(gh)
                check(lib().tg_gru_gates_bwd_masked(ops._p(gi), ops._p(gh), ops._p(h_rows), ops._p(d_table), ops._p(uniq), ops._p(bank._has_dev),
                                                    d_table.shape[0], d_table.shape[1], ops._p(dgi), ops._p(dgh), ops._stream()),
                      "tg_gru_gates_bwd_masked")
                ops.wgrad_group([(dgi, msg_rows, gv[0], gv[2]), (dgh, h_rows, gv[1], gv[3])])
            g = zeroed[:npar + gru_len]
            if flat[0].grad is None:
                flat[0].grad = g
            else:
                flat[0].grad.add_(g)
        self._advance_state(job, rows, edge_ids)
        return emb, loss

    def _advance_state(self, job, rows, edge_ids):
        """positive call: persist the batch nodes' GRU rows, build the new raw messages from the post-update state, file them
        last-message-wins, update the host mirrors (reference :155-180)"""
        from .._lib import check, lib
        bank, dev = self.memory_bank, self.node_raw_features.device
        n, m, k = job["n"], job["m"], job["k"]
        node_ids, times, rowmap, batch_d, b_d, t32_d = job["node_ids"], job["times"], job["rowmap"], job["batch_d"], job["b_d"], job["t32_d"]
        e_d = job["e_d"]
        if e_d is None:
            assert edge_ids is not None
            e_d, = ops.h2d([np.concatenate([np.asarray(edge_ids), np.asarray(edge_ids)]).astype(np.int32)], dev)
        u, new_t = job["u"], job["new_t"]
        # host mirrors first (one C call; raises the reference's assertion of :485-486 before anything changes): pending messages of
        # the batch nodes applied, the new ones filed at the time of each node's last occurrence
        import ctypes as C
        viol = C.c_int(0)
        check(lib().tg_tgn_host_advance(u.ctypes.data, new_t.ctypes.data, len(u), bank._has.ctypes.data, bank._msg_time.ctypes.data,
                                        bank._h_last.ctypes.data, len(bank._has), C.addressof(viol)), "tg_tgn_host_advance")
        if viol.value:
            bank._past_violation = True         # the reference's next get_updated_memories would raise on this message
        with torch.no_grad():
            check(lib().tg_tgn_persist(ops._p(rows.detach()), rows.stride(0), ops._p(job["row_batch"]), ops._p(batch_d), ops._p(bank._has_dev),
                                       ops._p(bank._msg_time_dev), ops._p(bank.node_memories.data), bank.node_memories.stride(0),
                                       ops._p(bank.node_last_updated_times.data), 2 * n, self.memory_dim, ops._stream()), "tg_tgn_persist")
            msgs = ops.build_messages(bank.node_memories.data, bank.node_last_updated_times.data, batch_d, b_d, t32_d,
                                      self.edge_raw_features, e_d, self.time_encoder.w.weight.detach().reshape(-1),
                                      self.time_encoder.w.bias.detach())
            check(lib().tg_msg_scatter_last(ops._p(batch_d), ops._p(msgs), msgs.stride(0), ops._p(t32_d), 2 * n, self.message_dim,
                                            ops._p(bank._msg), bank._msg.stride(0), ops._p(bank._has_dev), ops._p(bank._msg_time_dev),
                                            ops._p(bank._last_idx_ws), ops._stream()), "tg_msg_scatter_last")

    # ---- lazy path: graph-only part (prefetchable) ----------------------------------------------------------------------------
    def prepare_batch_begin(self, src_node_ids, dst_node_ids, node_interact_times, num_neighbors: int = 20, shard=None, edge_ids=None):
        """Optional prefetch (not in the reference): the part of a call that depends on the graph and the batch only -- the H2D copy of
        the ids, the neighbor lookups, and the hash set of distinct touched nodes -- issued on a side stream; the count of distinct
        nodes travels to pinned memory.  prepare_batch_finish(job) (a step later: no wait) yields the object to pass as
        `src_node_ids` of compute_src_dst_node_temporal_embeddings / compute_shard_embeddings_and_advance (dst / times then None)."""
        import ctypes as C
        from .._lib import check, lib
        st = getattr(self, "_stepper", None)
        if st is not None and self.training and int(num_neighbors) == st.k:
            assert int(num_neighbors) > 0, 'Number of sampled neighbors for each node should be greater than 0!'
            return st.begin_tgn(src_node_ids, dst_node_ids, node_interact_times, edge_ids, shard)
        dev = self.node_raw_features.device
        src = np.ascontiguousarray(src_node_ids, dtype=np.int64)
        dst = np.ascontiguousarray(dst_node_ids, dtype=np.int64)
        times = np.ascontiguousarray(node_interact_times, dtype=np.float64)
        eid = None if edge_ids is None else np.ascontiguousarray(edge_ids, dtype=np.int64)
        n, k = len(src), int(num_neighbors)
        assert k > 0, 'Number of sampled neighbors for each node should be greater than 0!'
        lo, hi = (0, n) if shard is None else shard
        m = 2 * (hi - lo)                                      # embedded roots: both roles of the shard's edges
        graph = self.embedding_module.neighbor_sampler.graph
        off = (C.c_int64 * 8)()
        check(lib().tg_tgn_prepare_layout(n, hi - lo, k, off), "tg_tgn_prepare_layout")
        o_t, o_b, o_e, o_t32, o_root, o_batch, o_nbr, o_end = (int(v) for v in off)
        total = o_end - o_root
        side, main = engine._side_stream(), torch.cuda.current_stream()
        # ONE C call on the side stream: pinned staging of the ids / times, one H2D copy, neighbor lookup, distinct touched nodes, count
        # to pinned memory.  The device buffer is allocated while the side stream is current (the caching allocator then never hands
        # out a block that queued main-stream kernels still read); set_stream costs 0.4 us, the `with torch.cuda.stream` form 6.
        stage = torch.empty(4 * o_nbr + 16, dtype=torch.uint8, pin_memory=True)             # (+ the 2 count words behind the staged blob)
        torch.cuda.set_stream(side)
        try:
            # one allocation: [blob (o_end) | slot edge ids, slot times, query - slot times (3 m k) | distinct ids, their times, row of every entry (3 total) | count, pad]
            dev_all = torch.empty(o_end + 3 * m * k + 3 * total + 4, dtype=torch.int32, device=dev)
            zt = getattr(self, "_zero_t", None)
            if zt is None or zt.numel() < total:
                zt = self._zero_t = torch.zeros(total, dtype=torch.float32, device=dev)
            cap, ws = graph._dedupe_ws_for(total, dev)
        finally:
            torch.cuda.set_stream(main)
        blob = dev_all[:o_end]
        S3 = dev_all[o_end:o_end + 3 * m * k].view(3, m, k)
        ded = dev_all[o_end + 3 * m * k:o_end + 3 * m * k + 3 * total].view(3, total)
        base = dev_all.data_ptr()
        p_S, p_ded = base + 4 * o_end, base + 4 * (o_end + 3 * m * k)
        count_ptr = stage.data_ptr() + 4 * o_nbr
        u_buf, t_buf, n_u = np.empty(2 * n, dtype=np.int64), np.empty(2 * n, dtype=np.float64), C.c_int64(0)
        check(lib().tg_tgn_prepare_batch(graph._h, src.ctypes.data, dst.ctypes.data, times.ctypes.data, None if eid is None else eid.ctypes.data,
                                         n, lo, hi, k, self.num_nodes, stage.data_ptr(), base, p_S, p_S + 4 * m * k, p_S + 8 * m * k,
                                         zt.data_ptr(), cap, ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(),
                                         p_ded, p_ded + 4 * total, p_ded + 8 * total, p_ded + 12 * total, count_ptr,
                                         u_buf.ctypes.data, t_buf.ctypes.data, C.addressof(n_u), side.cuda_stream),
              "tg_tgn_prepare_batch")
        ev = engine._ring_event()            # (a new torch.cuda.Event costs 9 us: hipEventCreate)
        ev.record(side)
        S = (blob[o_nbr:o_end].view(m, k), S3[0], S3[1].view(torch.float32), S3[2].view(torch.float32))
        rowmap = ded[2]
        node_ids = np.concatenate([src, dst])
        # host side of the state advance (filled by the C call): the distinct batch nodes and, per node, the time of its LAST
        # occurrence in [src role | dst role] order
        u, new_t = u_buf[:n_u.value], t_buf[:n_u.value]
        return dict(n=n, m=m, k=k, lo=lo, hi=hi, node_ids=node_ids, times=times, S=S, uniq=ded[0], rowmap=rowmap,
                    row_roots=rowmap[:m], row_batch=rowmap[m:m + 2 * n], row_nbrs=rowmap[m + 2 * n:],
                    batch_d=blob[o_batch:o_batch + 2 * n], b_d=blob[o_b:o_b + 2 * n], t32_d=blob[o_t32:o_t32 + 2 * n].view(torch.float32),
                    e_d=(blob[o_e:o_e + 2 * n] if eid is not None else None), u=u, new_t=new_t,
                    count_host=stage[4 * o_nbr:4 * o_nbr + 8].view(torch.int32), ready=ev, main=main, graph=graph,
                    keep=(stage, dev_all, src, dst, times, eid))

    def prepare_batch_finish(self, job):
        from ..stepper import StepJob
        if isinstance(job, StepJob):
            return job.stepper.finish(job)
        job["ready"].synchronize()
        count, pad = job["count_host"].tolist()
        job["uniq"] = job["uniq"][:count]
        job["pad"] = pad
        job["keep"][1].record_stream(job["main"])          # the one device allocation every tensor of the job is a view of
        job["finished"] = True
        return job

    def _step_lazy(self, src_node_ids, dst_node_ids, node_interact_times, edge_ids, edges_are_positive, num_neighbors, shard):
        """Lazy memory: the reference applies every pending message of the WHOLE graph before each call
        (get_updated_memories over all nodes, :117) although a one-layer embedding reads only the rows of the batch nodes and of
        their sampled neighbors.  Here the distinct touched nodes are found on the device (hash set), the GRU runs on exactly those
        rows, `memory' + raw` is a compact gathered table the attention kernels index through a row map, and the state update
        (persist, new messages, last-message-wins scatter) is three kernels with no host-side unique / index_copy.  Results are the
        reference's: a node's updated memory depends only on its own pending message."""
        from .._lib import check, lib
        bank, gru, dev = self.memory_bank, self.memory_updater.memory_updater, self.node_raw_features.device
        if isinstance(src_node_ids, dict):
            job = src_node_ids
            assert job.get("finished") and job["k"] == int(num_neighbors) and job["graph"] is self.embedding_module.neighbor_sampler.graph, \
                "prepared for a different sampler / num_neighbors"
        else:
            job = self.prepare_batch_finish(self.prepare_batch_begin(src_node_ids, dst_node_ids, node_interact_times, num_neighbors, shard))
        torch.cuda.current_stream().wait_event(job["ready"])
        n, m, k, lo, hi = job["n"], job["m"], job["k"], job["lo"], job["hi"]
        node_ids, times, S, uniq, rowmap, batch_d = job["node_ids"], job["times"], job["S"], job["uniq"], job["rowmap"], job["batch_d"]
        assert not bank._past_violation, "Trying to update memory to time in the past!"        # reference :515-516 (raised by the view)
        mem = bank.node_memories.detach()
        if bank._has.any():
            has_rows = bank._has_dev[uniq.long()]
            rows = _GRURowsMaskedFn.apply(ops.gather_rows(bank._msg, uniq), ops.gather_rows(mem, uniq), has_rows,
                                          gru.weight_ih, gru.weight_hh, gru.bias_ih, gru.bias_hh)
        else:
            rows = ops.gather_rows(mem, uniq)             # nothing pending anywhere (first batch of an epoch): the GRU is not called (:203-212)
        base = rows + ops.gather_rows(self.node_raw_features, uniq)                                # reference :654-655 on the touched rows
        fr = engine.Frontier(counts=[m], ids_all=job["row_roots"], S=S, child=None, pad_rows=[], feat_idx0=job["row_nbrs"], pad_row0=job["pad"])
        cfg = dict(n=m, k=k, num_layers=1, num_heads=self.num_heads, dropout=float(self.dropout), training=bool(self.training),
                   edge_table=self.edge_raw_features, table_grad=bool(torch.is_grad_enabled() and rows.requires_grad))
        emb = engine._apply(cfg, fr, base, self.time_encoder.w.weight, self.time_encoder.w.bias, self.embedding_module.layer_params(), None)
        src_emb, dst_emb = engine.split_rows(emb, hi - lo)
        if edges_are_positive:
            self._advance_state(job, rows, edge_ids)
        return src_emb, dst_emb

    def _step_full(self, src_node_ids, dst_node_ids, node_interact_times, edge_ids, edges_are_positive, num_neighbors, shard):
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
            bank._sync_device_flags()
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
        st = getattr(self, "_stepper", None)          # the native step follows the sampler (PTCL/EM_warmup.py:118, :296)
        if st is not None:
            if neighbor_sampler.sample_neighbor_strategy == "recent":
                st.rebind(neighbor_sampler.graph)
            else:
                self._stepper = None


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
