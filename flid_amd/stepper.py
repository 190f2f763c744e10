"""The trainer's step through ONE native object (tg_stepper, csrc/tg_step.hip): a batch is prepared on the object's own side stream
(one C call per half), the forward of every layer is one C call, the backward of every layer + the optimizer's update another.

replaces the host side of models/TGAT.py:50-144 and of the loss.backward() / optimizer.step() sequence around it
(PTCL/EM_warmup.py:126-238, PTCL/M_step.py:209-325) for the fused trainers; flid_amd/engine.py's Python form of the same work stays as
the autograd-facing path (and as the test oracle of this one: same kernels, same seeds, same numbers).

PyTorch supplies the arena (one allocation, sized by the library) and the views a caller sees (embeddings, gradient); every launch of
a step is issued by the library."""
import ctypes as C

import numpy as np
import torch

from . import engine, ops
from ._lib import AdamArgs, DygCfg, GRAD_READY_FN, StepperCfg, check, lib


class StepJob:
    """a batch in preparation / prepared (a slot of the native object)"""
    __slots__ = ("stepper", "slot", "n", "nsrc", "finished", "rows", "k", "num_layers")

    def __init__(self, stepper, slot, n, nsrc):
        self.stepper, self.slot, self.n, self.nsrc, self.finished, self.rows = stepper, slot, n, nsrc, False, None
        self.k, self.num_layers = stepper.k, stepper.num_layers


class Stepper:
    """model: a TGAT or a MemoryModel (TGN, one layer) in flat-parameter mode (flatten_parameters()).  max_roots: roots of one prepared
    batch at most (TGN: 2 x the edges of a batch)."""

    def __init__(self, model, max_roots: int, num_neighbors: int = 20, slots: int = 4, dedupe=None):
        flat = getattr(model, "_flat_pack", None)
        if flat is None:
            raise RuntimeError("Stepper needs the flat-parameter mode: call flatten_parameters() first")
        self.tgn = hasattr(model, "memory_bank")
        sampler = model.embedding_module.neighbor_sampler if self.tgn else model.neighbor_sampler
        if sampler.sample_neighbor_strategy != "recent":
            raise NotImplementedError("the native step samples on the device ('recent')")
        self.model, self.flat = model, flat[0]
        self.k, self.num_layers = int(num_neighbors), int(model.num_layers)
        self.graph = sampler.graph
        node, edge = model.node_raw_features, model.edge_raw_features
        dev = node.device
        cfg = StepperCfg()
        cfg.graph = self.graph.handle
        cfg.d_node, cfg.node_ld = node.data_ptr(), node.stride(0)
        cfg.d_edge, cfg.edge_ld = edge.data_ptr(), edge.stride(0)
        cfg.dn, cfg.de, cfg.dt_dim = node.shape[1], edge.shape[1], model.time_feat_dim
        cfg.heads, cfg.layers, cfg.k = model.num_heads, model.num_layers, self.k
        cfg.max_roots, cfg.slots = int(max_roots), int(slots)
        cfg.d_param, cfg.param_floats = self.flat.data_ptr(), self.flat.numel()
        cfg.dropout_p = float(model.dropout)
        cfg.dedupe = int(engine.DEDUPE if dedupe is None else dedupe)
        cfg.extra_grad_floats = 0
        cfg.tgn = int(self.tgn)
        need = int(lib().tg_stepper_param_floats(C.byref(cfg)))
        if need != self.flat.numel():
            raise RuntimeError(f"flat parameter holds {self.flat.numel()} floats, the native layout {need}")
        total = int(lib().tg_stepper_arena_floats(C.byref(cfg)))
        if total <= 0:
            check(-1, "tg_stepper_arena_floats")
        self.arena = torch.empty(total + 64, dtype=torch.float32, device=dev)
        base = self.arena.data_ptr()
        shift = (-base // 4) % 64                                   # 256-byte aligned start
        self.arena = self.arena[shift:shift + total]
        self.cfg = cfg
        h = C.c_void_p()
        check(lib().tg_stepper_create(C.byref(cfg), self.arena.data_ptr(), total, C.byref(h)), "tg_stepper_create")
        self._h = h
        off = (C.c_int64 * 4)()
        check(lib().tg_stepper_regions(h, self.arena.data_ptr(), off), "tg_stepper_regions")
        self.grad = self.arena[off[0]:off[0] + self.flat.numel()]        # parameter gradients, laid out like the flat parameter
        self._emb_off, self.dn, self.max_roots, self.nslots = int(off[2]), int(cfg.dn), int(max_roots), int(slots)
        self._next = 0
        self._grad_pending = False           # a backward without an update left its gradient in the block
        self._seeds = (C.c_uint64 * (2 * self.num_layers))()
        self._rows = (C.c_int64 * 2)()
        self._emb_views = {}
        self.keep = (node, edge, self.flat)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().tg_stepper_destroy(h)
            except Exception:
                pass

    def rebind(self, graph):
        """the model's neighbor sampler was swapped (set_neighbor_sampler: the trainers alternate between the train-graph and the
        full-graph sampler every epoch, PTCL/EM_warmup.py:118, :296): drop every prepared batch and sample from `graph` from now on"""
        if graph is self.graph:
            return
        self.reset()
        check(lib().tg_stepper_set_graph(self._h, graph.handle), "tg_stepper_set_graph")
        self.graph = graph
        self.cfg.graph = graph.handle

    def _check_graph(self):
        m = self.model
        sampler = m.embedding_module.neighbor_sampler if self.tgn else m.neighbor_sampler
        if sampler.graph is not self.graph:
            raise RuntimeError("the native stepper samples from another graph than the model's current neighbor sampler "
                               "(set_neighbor_sampler() rebinds it; the sampler object was replaced behind the model's back)")

    # ---- preparation (side stream of the native object) ---------------------------------------------------------------------------
    def begin(self, id_lists, node_interact_times) -> StepJob:
        """id_lists: host int64 arrays that share `node_interact_times` ([src, dst], [src], [src, dst, negative dst]); the embedding block
        of the step holds their rows one after the other"""
        self._check_graph()
        ids = np.ascontiguousarray(np.concatenate(id_lists) if len(id_lists) > 1 else id_lists[0], dtype=np.int64)
        t = np.ascontiguousarray(node_interact_times, dtype=np.float64)
        times = np.ascontiguousarray(np.tile(t, len(id_lists))) if len(id_lists) > 1 else t
        slot = self._next
        check(lib().tg_stepper_prepare_begin(self._h, slot, ids.ctypes.data, times.ctypes.data, len(ids)), "tg_stepper_prepare_begin")
        self._next = (slot + 1) % self.nslots
        return StepJob(self, slot, len(ids), len(id_lists[0]))

    def finish(self, job: StepJob) -> StepJob:
        check(lib().tg_stepper_prepare_finish(self._h, job.slot, self._rows), "tg_stepper_prepare_finish")
        job.finished, job.rows = True, (int(self._rows[0]), int(self._rows[1]))
        return job

    def reset(self):
        """drop every batch in preparation (a trainer that restarts its prefetch pipeline)"""
        for slot in range(self.nslots):
            check(lib().tg_stepper_release(self._h, slot), "tg_stepper_release")
        self._next = 0

    def release(self, job: StepJob):
        check(lib().tg_stepper_release(self._h, job.slot), "tg_stepper_release")

    # ---- the step -----------------------------------------------------------------------------------------------------------------
    def forward(self, job: StepJob) -> torch.Tensor:
        training = bool(self.model.training)
        if training and self.cfg.dropout_p > 0:
            for i, v in enumerate(engine._next_seeds(2 * self.num_layers)):
                self._seeds[i] = v
        check(lib().tg_stepper_forward(self._h, job.slot, int(training), self._seeds, ops._stream(), None), "tg_stepper_forward")
        emb = self._emb_views.get(job.n)
        if emb is None:
            emb = self._emb_views[job.n] = self.arena[self._emb_off:self._emb_off + job.n * self.dn].view(job.n, self.dn)
        return emb

    def _ready_callback(self, grad_ready):
        """the C callback around `grad_ready`: an exception raised inside it (an RCCL / gloo error in segment_ready) cannot cross the C
        frame -- ctypes would print and swallow it -- so it is kept and re-raised once the C call has returned"""
        if grad_ready is None:
            return GRAD_READY_FN(), None
        g, base, box = self.grad, self.grad.data_ptr(), []

        def _cb(_user, seg_ptr, floats):
            if box:
                return
            try:
                o = (seg_ptr - base) // 4
                grad_ready(g[o:o + floats])
            except BaseException as e:          # noqa: BLE001 (relayed below)
                box.append(e)
        return GRAD_READY_FN(_cb), box

    def _earlier_grad(self, optimizer, cb_given, accumulate, who):
        """the flat parameter's .grad as the caller left it: None / this object's block (-> None) / another tensor (the step's gradient is
        added to it afterwards, as autograd accumulates).  The block is rewritten by every step (its zero fill rides in the forward's
        first launch), so a second backward into it without an update in between would silently drop the first: refuse that."""
        prev = self.flat.grad
        if prev is not None and prev.data_ptr() == self.grad.data_ptr():
            if optimizer is None and not accumulate and self._grad_pending:
                raise RuntimeError(f"{who}: the flat parameter's .grad still holds the previous step's gradient block, which this step "
                                   "rewrites -- zero_grad(set_to_none=True) after the optimizer's step (or pass optimizer= so that the "
                                   "update runs inside the call); gradients of several batches add up only with accumulate=True (TGN)")
            prev = None
        if prev is not None and (optimizer is not None or cb_given):
            raise RuntimeError(f"{who}: zero_grad(set_to_none=True) first (the update / the reduction runs on this step's gradient block)")
        return prev

    def backward(self, job: StepJob, d_emb: torch.Tensor, grad_ready=None, optimizer=None):
        """leaves the step's gradient in `self.grad` (and in the flat parameter's .grad); optimizer (a FlatAdam over the flat parameter):
        its update is issued right behind the last layer's backward, in the same call"""
        assert d_emb.is_contiguous() and d_emb.dtype == torch.float32 and d_emb.numel() == job.n * self.dn
        cb, box = self._ready_callback(grad_ready)
        prev = self._earlier_grad(optimizer, grad_ready is not None, False, "Stepper.backward")
        adam = None if optimizer is None else optimizer.native_args(self.flat)
        try:
            check(lib().tg_stepper_backward(self._h, job.slot, d_emb.data_ptr(), ops._stream(), cb, None, None if adam is None else C.byref(adam),
                                            None), "tg_stepper_backward")
        except BaseException:
            if optimizer is not None:
                optimizer.native_rollback(self.flat)            # the update was not issued: the bias correction's step count goes back
            raise
        if box:
            raise box[0]
        self._grad_pending = optimizer is None
        if prev is None:
            self.flat.grad = self.grad
        else:
            prev.add_(self.grad)                                    # as autograd accumulates

    # ---- TGN: the memory stage around the layer (tg_stepper_tgn_*) -------------------------------------------------------------------
    def begin_tgn(self, src, dst, t, edge_ids=None, shard=None) -> StepJob:
        self._check_graph()
        src, dst = np.ascontiguousarray(src, dtype=np.int64), np.ascontiguousarray(dst, dtype=np.int64)
        t = np.ascontiguousarray(t, dtype=np.float64)
        eid = None if edge_ids is None else np.ascontiguousarray(edge_ids, dtype=np.int64)
        n = len(src)
        lo, hi = (0, n) if shard is None else shard
        slot = self._next
        check(lib().tg_stepper_tgn_prepare_begin(self._h, slot, src.ctypes.data, dst.ctypes.data, t.ctypes.data,
                                                 None if eid is None else eid.ctypes.data, n, lo, hi), "tg_stepper_tgn_prepare_begin")
        self._next = (slot + 1) % self.nslots
        return StepJob(self, slot, 2 * (hi - lo), hi - lo)

    def _bank(self):
        from ._lib import TgnBank
        b = self.model.memory_bank
        mem = b.node_memories.data
        if b._has.dtype != np.bool_ or b._msg_time.dtype != np.float64 or b._h_last.dtype != np.float32:
            raise RuntimeError("memory bank host mirrors changed type")
        return TgnBank(mem.data_ptr(), mem.stride(0), b.node_last_updated_times.data.data_ptr(), b._msg.data_ptr(), b._msg.stride(0),
                       b._has_dev.data_ptr(), b._msg_time_dev.data_ptr(), b._last_idx_ws.data_ptr(), b._has.ctypes.data, b._msg_time.ctypes.data,
                       b._h_last.ctypes.data, len(b._has), int(bool(b._past_violation)))

    def forward_tgn(self, job: StepJob, keep_grad: bool = False) -> torch.Tensor:
        training = bool(self.model.training)
        if training and self.cfg.dropout_p > 0:
            for i, v in enumerate(engine._next_seeds(2 * self.num_layers)):
                self._seeds[i] = v
        bank = self._bank()
        check(lib().tg_stepper_tgn_forward(self._h, job.slot, C.byref(bank), int(training), self._seeds, ops._stream(), None, int(keep_grad)),
              "tg_stepper_tgn_forward")
        emb = self._emb_views.get(job.n)
        if emb is None:
            emb = self._emb_views[job.n] = self.arena[self._emb_off:self._emb_off + job.n * self.dn].view(job.n, self.dn)
        return emb

    def backward_tgn(self, job: StepJob, d_emb: torch.Tensor, positive: bool = True, optimizer=None, accumulate: bool = False, more: bool = False,
                     grad_ready=None):
        """accumulate: add to the gradients an earlier backward of this step left in the block; more: another backward follows;
        grad_ready(segment): called with the layer's finished gradient block while the GRU's backward and the state advance are still
        being issued (a data-parallel caller starts its reduction there)"""
        assert d_emb.is_contiguous() and d_emb.dtype == torch.float32 and d_emb.numel() == job.n * self.dn
        cb, box = self._ready_callback(grad_ready)
        prev = self._earlier_grad(optimizer, grad_ready is not None, accumulate, "Stepper.backward_tgn")
        adam = None if optimizer is None else optimizer.native_args(self.flat)
        bank = self._bank()
        try:
            check(lib().tg_stepper_tgn_backward(self._h, job.slot, C.byref(bank), d_emb.data_ptr(),
                                                int(bool(positive)) | (2 if accumulate else 0) | (4 if more else 0), ops._stream(),
                                                None if adam is None else C.byref(adam), None, cb, None), "tg_stepper_tgn_backward")
        except BaseException:
            if optimizer is not None:
                optimizer.native_rollback(self.flat)
            raise
        finally:
            if bank.past_violation:
                self.model.memory_bank._past_violation = True
        if box:
            raise box[0]
        self._grad_pending = optimizer is None
        if prev is None:
            self.flat.grad = self.grad
        else:
            prev.add_(self.grad)

    def step_tgn(self, job: StepJob, loss_fn, positive: bool = True, optimizer=None, accumulate: bool = False, more: bool = False,
                 grad_ready=None):
        try:
            emb = self.forward_tgn(job, keep_grad=accumulate)
            loss, d_emb = loss_fn(emb)
            self.backward_tgn(job, d_emb, positive=positive, optimizer=optimizer, accumulate=accumulate, more=more, grad_ready=grad_ready)
        except BaseException:
            self._release_quietly(job)              # (a slot only becomes free behind its backward: do not leak it)
            raise
        return emb, loss

    def step(self, job: StepJob, loss_fn, grad_ready=None, optimizer=None):
        """forward, `loss_fn(emb) -> (loss, d loss / d emb)`, backward (+ update): (embeddings, loss)"""
        try:
            emb = self.forward(job)
            loss, d_emb = loss_fn(emb)
            self.backward(job, d_emb, grad_ready=grad_ready, optimizer=optimizer)
        except BaseException:
            self._release_quietly(job)
            raise
        return emb, loss

    def _release_quietly(self, job: StepJob):
        try:
            self.release(job)
        except Exception:
            pass


class DygStepper:
    """DyGFormer's step through ONE native object (tg_dyg, csrc/tg_dyg.hip): forward = one C call from the host id arrays to the
    (2 B, dn) embeddings, backward (+ the optimizer's update) another; model: a flid_amd DyGFormer at patch size 1 in flat-parameter mode.

    replaces the host side of models/DyGFormer.py:60-194 and the loss.backward() / optimizer.step() around it (PTCL/M_step.py:209-325);
    the autograd form (DyGFormer.compute_src_dst_node_temporal_embeddings + seqops) stays as the reference-facing path and this one's
    test oracle."""

    def __init__(self, model, max_batch_edges: int):
        flat = getattr(model, "_flat_pack", None)
        if flat is None:
            raise RuntimeError("DygStepper needs the flat-parameter mode: call flatten_parameters() first")
        if model.patch_size != 1:
            raise NotImplementedError("the native DyGFormer step covers patch_size 1 (the autograd path takes other patch sizes)")
        self.model, self.flat = model, flat[0]
        self.graph = model.neighbor_sampler.graph
        node, edge = model.node_raw_features, model.edge_raw_features
        cfg = DygCfg()
        cfg.graph = self.graph.handle
        cfg.d_node, cfg.node_ld = node.data_ptr(), node.stride(0)
        cfg.d_edge, cfg.edge_ld, cfg.num_edge_rows = edge.data_ptr(), edge.stride(0), edge.shape[0]
        cfg.d_param, cfg.param_floats = self.flat.data_ptr(), self.flat.numel()
        base = self.flat.data_ptr()
        tensors = model._native_param_order()
        assert len(tensors) <= 64
        for i, t in enumerate(tensors):
            assert t.is_contiguous() and (t.data_ptr() - base) % 16 == 0
            cfg.poff[i] = (t.data_ptr() - base) // 4
        cfg.dn, cfg.de, cfg.dt_dim, cfg.channel = node.shape[1], edge.shape[1], model.time_feat_dim, model.channel_embedding_dim
        cfg.layers, cfg.heads, cfg.max_len, cfg.max_edges = model.num_layers, model.num_heads, model.max_input_sequence_length, int(max_batch_edges)
        total = int(lib().tg_dyg_arena_floats(C.byref(cfg)))
        if total <= 0:
            check(-5 if "native step covers" in (lib().tg_last_error() or b"").decode() else -1, "tg_dyg_arena_floats")
        self.arena = torch.empty(total + 64, dtype=torch.float32, device=node.device)
        shift = (-self.arena.data_ptr() // 4) % 64
        self.arena = self.arena[shift:shift + total]
        self.cfg = cfg
        h = C.c_void_p()
        check(lib().tg_dyg_create(C.byref(cfg), self.arena.data_ptr(), total, C.byref(h)), "tg_dyg_create")
        self._h = h
        off = (C.c_int64 * 2)()
        check(lib().tg_dyg_regions(h, self.arena.data_ptr(), off), "tg_dyg_regions")
        self.grad = self.arena[off[0]:off[0] + self.flat.numel()]
        self._emb_off, self.dn, self.max_edges = int(off[1]), int(cfg.dn), int(max_batch_edges)
        self._seeds = (C.c_uint64 * (4 * model.num_layers))()
        self._emb_views = {}
        self._last_B = 0
        self._grad_pending = False
        self.keep = (node, edge, self.flat)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().tg_dyg_destroy(h)
            except Exception:
                pass

    def rebind(self, graph):
        """the model's neighbor sampler was swapped (set_neighbor_sampler): later batches read their histories from `graph`"""
        if graph is self.graph:
            return
        check(lib().tg_dyg_set_graph(self._h, graph.handle), "tg_dyg_set_graph")
        self.graph = graph
        self.cfg.graph = graph.handle

    def forward(self, src_node_ids, dst_node_ids, node_interact_times) -> torch.Tensor:
        """(2 B, dn): source embeddings, then destination embeddings"""
        m = self.model
        if m.neighbor_sampler.graph is not self.graph:
            raise RuntimeError("the native DyGFormer step reads another graph than the model's current neighbor sampler (set_neighbor_sampler() rebinds it)")
        src = np.ascontiguousarray(src_node_ids, dtype=np.int64)
        dst = np.ascontiguousarray(dst_node_ids, dtype=np.int64)
        t = np.ascontiguousarray(node_interact_times, dtype=np.float64)
        B, L = len(src), m.max_input_sequence_length
        self._last_B = B
        # every side is as wide as ITS longest sequence of the batch (the node itself + its history, at most L): host binary searches
        ws, wd = (int(np.minimum(self.graph.count_before_host(ids, t), L - 1).max()) + 1 for ids in (src, dst))
        p = float(m.dropout) if m.training else 0.0
        if p > 0:                                               # four per block, drawn block by block as seqops.encoder_block draws them
            for l in range(m.num_layers):
                for i, v in enumerate(engine._next_seeds(4)):
                    self._seeds[4 * l + i] = v
        check(lib().tg_dyg_forward(self._h, src.ctypes.data, dst.ctypes.data, t.ctypes.data, B, ws, wd, p, self._seeds, ops._stream(), None),
              "tg_dyg_forward")
        emb = self._emb_views.get(B)
        if emb is None:
            emb = self._emb_views[B] = self.arena[self._emb_off:self._emb_off + 2 * B * self.dn].view(2 * B, self.dn)
        return emb

    def backward(self, d_emb: torch.Tensor, optimizer=None):
        """leaves the step's gradient in `self.grad` (= the flat parameter's .grad); optimizer (a FlatAdam over the flat parameter): its
        update is issued right behind the backward, in the same call"""
        assert d_emb.is_contiguous() and d_emb.dtype == torch.float32 and d_emb.numel() == 2 * self._last_B * self.dn, \
            "d_emb must be the (2 B, dn) gradient of the last forward's embedding block"
        prev = self.flat.grad
        if prev is not None and prev.data_ptr() == self.grad.data_ptr():
            if optimizer is None and self._grad_pending:
                raise RuntimeError("DygStepper.backward: the flat parameter's .grad still holds the previous step's gradient block, which this "
                                   "step rewrites -- zero_grad(set_to_none=True) after the optimizer's step, or pass optimizer=")
            prev = None
        if prev is not None and optimizer is not None:
            raise RuntimeError("DygStepper.backward: zero_grad(set_to_none=True) first (the update runs on this step's gradient block)")
        adam = None if optimizer is None else optimizer.native_args(self.flat)
        try:
            check(lib().tg_dyg_backward(self._h, d_emb.data_ptr(), ops._stream(), None if adam is None else C.byref(adam), None), "tg_dyg_backward")
        except BaseException:
            if optimizer is not None:
                optimizer.native_rollback(self.flat)
            raise
        self._grad_pending = optimizer is None
        if prev is None:
            self.flat.grad = self.grad
        else:
            prev.add_(self.grad)

    def step(self, src_node_ids, dst_node_ids, node_interact_times, loss_fn, optimizer=None):
        """forward, `loss_fn(emb) -> (loss, d loss / d emb)`, backward (+ update): (embeddings, loss)"""
        emb = self.forward(src_node_ids, dst_node_ids, node_interact_times)
        loss, d_emb = loss_fn(emb)
        self.backward(d_emb, optimizer=optimizer)
        return emb, loss
