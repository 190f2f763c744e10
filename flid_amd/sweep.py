"""Embedding-regeneration sweep: forward-only pass of the backbone over EVERY edge of a dataset into the
`(num_edges, node_feat_dim)` source / destination stores the E-step then trains its decoder on.

replaces: PTCL/M_step.py:456-509 (and PTCL/EM_warmup.py:291-347) -- a python loop of `compute_src_dst_node_temporal_embeddings` over
the full index loader under torch.no_grad(), `torch.cat` of the per-batch results, copy into `src_node_embeddings` /
`dst_node_embeddings`; the E-step reads rows `edge_ids - 1` of them (PTCL/E_step.py:58-59, :169-170).

What changes:
  * stateless backbones (TGAT, DyGFormer): an embedding is a function of (node, time) and the graph only, so the sweep runs in
    CHUNKS far larger than the trainer's batch (default 4 096 edges): one launch sequence per chunk instead of per 200 edges, row
    sharing (engine.DEDUPE) across the whole chunk -- the (node, float32 time) pairs sampled for one edge recur for the next edges
    of the same users, so a chunk's ~164 k neighbor slots collapse to a fraction of distinct layer-1 rows -- and the sampler work
    of the next chunk prefetched on a side stream (no host wait).  Results are written straight into the stores.
  * TGN: the memory makes the walk sequential and the batch boundaries part of the result, so the trainer's batch size is kept; every
    rank advances the replicated state with the whole batch and embeds only its shard (MemoryModel.compute_shard_embeddings_and_advance).
  * data parallel: rank r of `world` takes chunks r, r + world, ... (TGN: its shard of every batch) and ONE all-reduce (sum of
    disjointly filled, zero-initialised stores == all-gather) completes the stores on every rank.
"""
from typing import Optional, Tuple

import numpy as np
import torch


def regenerate_embeddings(backbone, data, batch_size: int = 200, num_neighbors: int = 20, chunk_edges: int = 4096,
                          out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, first_edge: int = 0, num_edges: Optional[int] = None,
                          rank: int = 0, world: int = 1, time_gap: int = 2000):
    """Fill (and return) the two stores with the embeddings of edges [first_edge, first_edge + num_edges) of `data` (chronological
    order, row = position in `data`, as the reference's full_idx_data_loader walks it).  The backbone's current neighbor sampler is
    used (the callers set the full-graph sampler first, M_step.py:458); TGN's memory bank must have been reset by the caller
    (M_step.py:460) and is advanced by the sweep."""
    from .models.DyGFormer import DyGFormer
    from .models.MemoryModel import MemoryModel
    from .models.TGAT import TGAT
    E = data.num_interactions if hasattr(data, "num_interactions") else len(data.src_node_ids)
    num_edges = E - first_edge if num_edges is None else num_edges
    dev = backbone.node_raw_features.device
    D = backbone.node_feat_dim
    if out is None:
        out = (torch.zeros((E, D), device=dev), torch.zeros((E, D), device=dev))
    src_store, dst_store = out
    was_training = backbone.training
    backbone.eval()
    src, dst, t = data.src_node_ids, data.dst_node_ids, data.node_interact_times
    try:
        with torch.no_grad():
            if isinstance(backbone, MemoryModel):
                B = int(batch_size)
                for lo in range(first_edge, first_edge + num_edges, B):
                    hi = min(lo + B, first_edge + num_edges)
                    n = hi - lo
                    a, b = (rank * n) // world, ((rank + 1) * n) // world
                    s, d = backbone.compute_shard_embeddings_and_advance(src[lo:hi], dst[lo:hi], t[lo:hi], data.edge_ids[lo:hi], (a, b), True,
                                                                         num_neighbors)
                    src_store[lo + a:lo + b].copy_(s)
                    dst_store[lo + a:lo + b].copy_(d)
            else:
                C = max(int(chunk_edges), 1)
                bounds = [(lo, min(lo + C, first_edge + num_edges)) for lo in range(first_edge, first_edge + num_edges, C)]
                mine = bounds[rank::world]
                if isinstance(backbone, TGAT) and backbone.neighbor_sampler.sample_neighbor_strategy == "recent":
                    ids32 = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.int32)).to(dev)
                    def begin(i):
                        lo, hi = mine[i]
                        return backbone.prepare_batch_begin(ids32(src[lo:hi]), ids32(dst[lo:hi]), torch.from_numpy(np.ascontiguousarray(t[lo:hi])).to(dev),
                                                            num_neighbors)
                    jobs = {0: begin(0)} if mine else {}
                    ready = {}
                    for i, (lo, hi) in enumerate(mine):
                        if i not in ready:
                            ready[i] = backbone.prepare_batch_finish(jobs.pop(i))
                        if i + 1 < len(mine):
                            if i + 1 not in jobs and i + 1 not in ready:
                                jobs[i + 1] = begin(i + 1)
                            ready[i + 1] = backbone.prepare_batch_finish(jobs.pop(i + 1))
                        if i + 2 < len(mine):
                            jobs[i + 2] = begin(i + 2)
                        s, d = backbone.compute_src_dst_node_temporal_embeddings(ready.pop(i), None, None, num_neighbors)
                        src_store[lo:hi].copy_(s)
                        dst_store[lo:hi].copy_(d)
                elif isinstance(backbone, DyGFormer):
                    # DyGFormer pads both sides to the longest history IN THE BATCH, its transformer is unmasked and the mean runs
                    # over the padded patches (models/DyGFormer.py:196-245, :185-187): an edge's embedding depends on which edges
                    # share its batch.  The stores must therefore be filled batch by batch exactly as the reference's loop walks them
                    # (`batch_size` edges, M_step.py:456-509) -- ranks take whole batches, never a different grouping.
                    B = int(batch_size)
                    batches = [(lo, min(lo + B, first_edge + num_edges)) for lo in range(first_edge, first_edge + num_edges, B)]
                    for lo, hi in batches[rank::world]:
                        s, d = backbone.compute_src_dst_node_temporal_embeddings(src[lo:hi], dst[lo:hi], t[lo:hi])
                        src_store[lo:hi].copy_(s)
                        dst_store[lo:hi].copy_(d)
                else:
                    from .models.GraphMixer import GraphMixer
                    for lo, hi in mine:
                        if isinstance(backbone, GraphMixer):        # (its node encoder looks `time_gap` neighbors back: args.time_gap, M_step.py:487)
                            s, d = backbone.compute_src_dst_node_temporal_embeddings(src[lo:hi], dst[lo:hi], t[lo:hi], num_neighbors, time_gap)
                        else:
                            s, d = backbone.compute_src_dst_node_temporal_embeddings(src[lo:hi], dst[lo:hi], t[lo:hi], num_neighbors)
                        src_store[lo:hi].copy_(s)
                        dst_store[lo:hi].copy_(d)
            if world > 1:
                import torch.distributed as dist
                dist.all_reduce(src_store[first_edge:first_edge + num_edges])
                dist.all_reduce(dst_store[first_edge:first_edge + num_edges])
    finally:
        backbone.train(was_training)
    return src_store, dst_store
