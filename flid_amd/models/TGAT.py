"""TGAT backbone -- drop-in for the reference class (models/TGAT.py): same constructor, methods, parameter names.

compute_src_dst_node_temporal_embeddings() takes host numpy (int64 ids, float64 times) and returns two device fp32
tensors connected by autograd to this module's parameters; the work in between runs in libflid_tg.so."""
import numpy as np
import torch
import torch.nn as nn

from .. import engine
from ..utils.utils import NeighborSampler
from .modules import MergeLayer, MultiHeadAttention, TimeEncoder


class TGAT(nn.Module):

    def __init__(self, node_raw_features: np.ndarray, edge_raw_features: np.ndarray, neighbor_sampler: NeighborSampler,
                 time_feat_dim: int, num_layers: int = 2, num_heads: int = 2, dropout: float = 0.1, device: str = 'cpu'):
        super().__init__()
        if torch.device(device).type != "cuda":
            raise RuntimeError("flid_amd.TGAT runs on a ROCm device only (device='cuda' / 'cuda:N'); there is no CPU path")
        # plain tensors, not parameters/buffers: not in state_dict, no grad (reference TGAT.py:26-29)
        # (tables already resident in HBM -- flid_amd.ops.hash_features at SURVEY 8d config-5 sizes -- are taken as they are)
        as_dev = lambda x: (x.to(device=device, dtype=torch.float32) if torch.is_tensor(x)
                            else torch.from_numpy(x.astype(np.float32)).to(device)).contiguous()
        self.node_raw_features = as_dev(node_raw_features)
        self.edge_raw_features = as_dev(edge_raw_features)
        self.neighbor_sampler = neighbor_sampler
        self.node_feat_dim = self.node_raw_features.shape[1]
        self.edge_feat_dim = self.edge_raw_features.shape[1]
        self.time_feat_dim = time_feat_dim
        self.num_layers = num_layers
        self.num_heads = num_heads
        self.dropout = dropout
        self.time_encoder = TimeEncoder(time_dim=time_feat_dim)
        self.temporal_conv_layers = nn.ModuleList([
            MultiHeadAttention(node_feat_dim=self.node_feat_dim, edge_feat_dim=self.edge_feat_dim, time_feat_dim=self.time_feat_dim,
                               num_heads=self.num_heads, dropout=self.dropout) for _ in range(num_layers)])
        self.merge_layers = nn.ModuleList([
            MergeLayer(input_dim1=self.node_feat_dim + self.time_feat_dim, input_dim2=self.node_feat_dim,
                       hidden_dim=self.node_feat_dim, output_dim=self.node_feat_dim) for _ in range(num_layers)])

    def _layer_params(self):
        out = []
        for conv, merge in zip(self.temporal_conv_layers, self.merge_layers):
            out += conv.fused_params() + [merge.fc1.weight, merge.fc1.bias, merge.fc2.weight, merge.fc2.bias]
        return out

    def flatten_parameters(self) -> nn.Parameter:
        """Opt-in (not in the reference): re-home every parameter of the backbone in ONE flat nn.Parameter and return it.
        The named parameters stay where they are -- same state_dict keys, load_state_dict keeps working -- but become views of
        the flat buffer with requires_grad off; the trainer hands the returned parameter to its optimizer instead of
        model.parameters().  The backward pass then delivers one gradient tensor (its gradient block already has this layout,
        engine.block_layout): one AccumulateGrad, one optimizer kernel, one all-reduce operand instead of 24.  Call after the
        module is on its device."""
        params = [self.time_encoder.w.weight, self.time_encoder.w.bias] + self._layer_params()
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

    def enable_native_step(self, max_batch_roots: int, num_neighbors: int = 20, slots: int = 4):
        """Opt-in (not in the reference; needs flatten_parameters()): prepare_batch_begin / prepare_roots_begin / prepare_batch_finish /
        train_step go through ONE native object (flid_amd.stepper.Stepper, csrc/tg_step.hip) -- a C call each, every launch of a step
        issued by the library out of a pre-sized arena.  max_batch_roots: roots of a prepared batch at most (2 B; 3 B for the
        link-prediction warm-up's [src | dst | negative dst]).  Host numpy batches only; a prepared batch of this mode is consumed by
        train_step (the autograd-facing calls keep the Python engine)."""
        from ..stepper import Stepper
        self._stepper = Stepper(self, max_batch_roots, num_neighbors, slots)
        return self._stepper

    def compute_src_dst_node_temporal_embeddings(self, src_node_ids: np.ndarray, dst_node_ids: np.ndarray,
                                                 node_interact_times: np.ndarray, num_neighbors: int = 20, roots: str = "both"):
        # both sides share one device pass: rows are independent (reference computes them one after the other, :61-65)
        # roots = "src" (not in the reference): only the source embeddings are wanted (PTCL/M_step.py:285) -> (src_emb, None)
        if roots == "src" and not isinstance(src_node_ids, engine.PreparedFrontier):
            return self.compute_node_temporal_embeddings(src_node_ids, node_interact_times, self.num_layers, num_neighbors), None
        if isinstance(src_node_ids, engine.PreparedFrontier):
            emb = self.compute_node_temporal_embeddings(src_node_ids, None, self.num_layers, num_neighbors)
            return engine.split_rows(emb, src_node_ids.nsrc)
        nsrc = len(src_node_ids)
        if torch.is_tensor(src_node_ids):          # ids/times already in HBM (bench, fused trainers)
            ids = torch.cat([src_node_ids, dst_node_ids])
            times = torch.cat([node_interact_times, node_interact_times])
        else:
            ids = np.concatenate([src_node_ids, dst_node_ids])
            times = np.concatenate([node_interact_times, node_interact_times])
        emb = self.compute_node_temporal_embeddings(ids, times, self.num_layers, num_neighbors, _groups=[(0, nsrc), (nsrc, len(ids))])
        return engine.split_rows(emb, nsrc)

    def prepare_batch(self, src_node_ids, dst_node_ids, node_interact_times, num_neighbors: int = 20):
        """Optional prefetch (not in the reference): do the sampler work of a FUTURE batch now, on a side stream.  Takes
        device tensors (int32 ids, float64 times); pass the result as `src_node_ids` of compute_src_dst_node_temporal_embeddings
        (dst / times may then be None)."""
        pf = engine.prepare_frontier(self.neighbor_sampler.graph, [src_node_ids, dst_node_ids], [node_interact_times, node_interact_times],
                                     num_neighbors, self.num_layers)
        pf.nsrc = src_node_ids.numel()
        return pf

    def prepare_batch_begin(self, src_node_ids, dst_node_ids, node_interact_times, num_neighbors: int = 20, roots: str = "both"):
        """First half of prepare_batch for a trainer that knows its batches two steps ahead: issues the level-0 lookups and the
        row sharing on the side stream and returns at once; prepare_batch_finish(job) one step later reads the (by then
        complete) distinct-row count without waiting and issues the rest."""
        # roots = "src": embed the source nodes only -- single-way datasets' M-step classifies `batch_src_node_embeddings` alone
        # (PTCL/M_step.py:285: the destination rows the reference computes there are never read), half the roots of a step
        assert roots in ("both", "src")
        st = getattr(self, "_stepper", None)
        # (the native object serves training steps; an eval-time prefetch takes the engine's path, whose job the autograd-facing call accepts)
        if st is not None and self.training and not torch.is_tensor(src_node_ids) and num_neighbors == st.k:
            return st.begin([np.asarray(src_node_ids)] if roots == "src" else [np.asarray(src_node_ids), np.asarray(dst_node_ids)],
                            node_interact_times)
        if not torch.is_tensor(src_node_ids):
            # host numpy int64 ids / float64 times, as the reference's trainers hand them over (PTCL/EM_warmup.py:128-130): one pinned
            # staging block, one asynchronous copy ON THE SIDE STREAM (the main stream is a step behind and must not be waited for)
            from .. import ops
            ids = [np.asarray(src_node_ids), np.asarray(dst_node_ids)]
            graph = self.neighbor_sampler.graph
            for a in ids:
                if len(a) and (int(a.max()) >= graph.num_rows or int(a.min()) < 0):
                    raise IndexError("list index out of range")                      # what utils/utils.py:141 raises
            with torch.cuda.stream(engine._side_stream()):
                src_node_ids, dst_node_ids, node_interact_times = ops.h2d(
                    [np.ascontiguousarray(ids[0], dtype=np.int32), np.ascontiguousarray(ids[1], dtype=np.int32),
                     np.ascontiguousarray(node_interact_times, dtype=np.float64)], self.node_raw_features.device)
        if roots == "src":
            job = engine.prepare_begin(self.neighbor_sampler.graph, [src_node_ids], [node_interact_times], num_neighbors, self.num_layers)
        else:
            job = engine.prepare_begin(self.neighbor_sampler.graph, [src_node_ids, dst_node_ids], [node_interact_times, node_interact_times],
                                       num_neighbors, self.num_layers)
        job.nsrc = src_node_ids.numel()
        return job

    def prepare_roots_begin(self, id_lists, node_interact_times, num_neighbors: int = 20):
        """as prepare_batch_begin for any number of root lists that share the batch's interaction times -- the link-prediction warm-up's
        [src, dst, negative dst] (PTCL/EM_warmup.py:128-153; the reference embeds the sources twice, once per pair).  Host numpy
        int64 / float64 in; the embedding block of the prepared batch is the lists' rows one after the other."""
        from .. import ops
        st = getattr(self, "_stepper", None)
        if st is not None and self.training and num_neighbors == st.k:
            return st.begin([np.asarray(a) for a in id_lists], node_interact_times)
        graph = self.neighbor_sampler.graph
        arrs = [np.asarray(a) for a in id_lists]
        for a in arrs:
            if len(a) and (int(a.max()) >= graph.num_rows or int(a.min()) < 0):
                raise IndexError("list index out of range")
        with torch.cuda.stream(engine._side_stream()):
            dev = ops.h2d([np.ascontiguousarray(a, dtype=np.int32) for a in arrs] + [np.ascontiguousarray(node_interact_times, dtype=np.float64)],
                          self.node_raw_features.device)
        job = engine.prepare_begin(graph, dev[:-1], [dev[-1]] * len(arrs), num_neighbors, self.num_layers)
        job.nsrc = len(arrs[0])
        return job

    def prepare_batch_finish(self, job):
        from ..stepper import StepJob
        if isinstance(job, StepJob):
            return job.stepper.finish(job)
        pf = engine.prepare_finish(job)
        pf.nsrc = job.nsrc
        return pf

    def train_step(self, prepared, loss_fn, num_neighbors: int = 20, grad_ready=None, optimizer=None):
        """Fused-trainer step (not in the reference): forward of the prepared batch, `loss_fn(emb) -> (loss, d_emb)` on
        the detached (2 B, Dn) embedding block [src rows | dst rows], backward -- without an autograd graph.  Needs
        flatten_parameters(); the gradient lands in the flat parameter's .grad exactly as loss.backward() would leave it.
        optimizer (a FlatAdam over the flat parameter; native step only, not together with grad_ready): its update is issued in the
        same native call as the backward -- the caller then does not call optimizer.step()."""
        from ..stepper import StepJob
        if isinstance(prepared, StepJob):
            assert prepared.finished and prepared.k == num_neighbors and prepared.stepper is getattr(self, "_stepper", None), \
                "prepared by another stepper / not finished"
            return prepared.stepper.step(prepared, loss_fn, grad_ready=grad_ready, optimizer=optimizer)
        if optimizer is not None:
            raise NotImplementedError("train_step(optimizer=...) is the native step's (enable_native_step())")
        flat = getattr(self, "_flat_pack", None)
        if flat is None:
            raise RuntimeError("TGAT.train_step needs the flat-parameter mode: call flatten_parameters() first")
        if self.neighbor_sampler.sample_neighbor_strategy != "recent":
            raise NotImplementedError("train_step takes a prepared (device-sampled, 'recent') batch")
        pf = prepared
        assert pf.k == num_neighbors and pf.num_layers == self.num_layers and pf.graph is self.neighbor_sampler.graph, \
            "prepared for a different sampler / k / depth"
        torch.cuda.current_stream().wait_event(pf.ready)
        cfg = dict(n=pf.n, k=num_neighbors, num_layers=self.num_layers, num_heads=self.num_heads, dropout=float(self.dropout),
                   training=bool(self.training), edge_table=self.edge_raw_features, table_grad=False)
        if self.num_heads > 2 or not engine.NATIVE:
            raise NotImplementedError("train_step runs the one-call-per-layer path (1 or 2 heads)")
        return engine.forward_backward(cfg, pf.frontier, self.node_raw_features, flat, loss_fn, grad_ready=grad_ready)

    def compute_node_temporal_embeddings(self, node_ids: np.ndarray, node_interact_times: np.ndarray,
                                         current_layer_num: int, num_neighbors: int = 20, _groups=None):
        assert current_layer_num >= 0
        # 'recent' (the FLiD default, load_configs.py:115) is sampled on the device.  'uniform' / 'time_interval_aware' draw from
        # numpy's RandomState on the host in the reference's order (bit-exact neighbor choice), the kernels are the same.
        host_sampler = None if self.neighbor_sampler.sample_neighbor_strategy == "recent" else self.neighbor_sampler
        if host_sampler is not None and isinstance(node_ids, engine.PreparedFrontier):
            raise NotImplementedError("prepared batches are device-sampled ('recent')")
        flat = getattr(self, "_flat_pack", None)
        if flat is not None and current_layer_num != self.num_layers and torch.is_grad_enabled():
            # the named parameters are views with requires_grad off in this mode: a partial-depth call would train nothing, silently
            raise RuntimeError("flatten_parameters(): only full-depth calls (current_layer_num == num_layers) are differentiable; "
                               "use torch.no_grad() for partial depths or do not flatten")
        if flat is not None and current_layer_num == self.num_layers and torch.is_grad_enabled():
            views = flat[1]
            return engine.embed(self.neighbor_sampler.graph, self.node_raw_features, self.edge_raw_features, views[0], views[1],
                                views[2:], node_ids, node_interact_times, num_neighbors, current_layer_num, self.num_heads,
                                self.dropout, self.training, flat=flat, host_sampler=host_sampler, groups=_groups)
        params = self._layer_params()[:11 * current_layer_num]
        return engine.embed(self.neighbor_sampler.graph, self.node_raw_features, self.edge_raw_features,
                            self.time_encoder.w.weight, self.time_encoder.w.bias, params, node_ids, node_interact_times,
                            num_neighbors, current_layer_num, self.num_heads, self.dropout, self.training,
                            host_sampler=host_sampler, groups=_groups)

    def set_neighbor_sampler(self, neighbor_sampler: NeighborSampler):
        self.neighbor_sampler = neighbor_sampler
        if self.neighbor_sampler.sample_neighbor_strategy in ['uniform', 'time_interval_aware']:
            assert self.neighbor_sampler.seed is not None
            self.neighbor_sampler.reset_random_state()
        # the native step follows the sampler (the trainers alternate train-graph / full-graph samplers every epoch, PTCL/EM_warmup.py:118, :296)
        st = getattr(self, "_stepper", None)
        if st is not None:
            if neighbor_sampler.sample_neighbor_strategy == "recent":
                st.rebind(neighbor_sampler.graph)
            else:
                self._stepper = None
