"""One rank of tests/test_00_gpu_dist.py (started by torch.distributed.run; also usable by hand on one GPU for plumbing:
FLID_BENCH_SHARE_GPU=1 FLID_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 tests/dist_gpu_worker.py)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from flid_amd import dist as fdist                                   # noqa: E402
from flid_amd import ops                                             # noqa: E402
from flid_amd._lib import lib                                        # noqa: E402


def flat_grad(model, flat, data, sl, k, weight, reducer=None):
    """one fused step's gradient on edges `sl` (loss = sum over the slice of emb . r / n_global)"""
    n = sl.stop - sl.start
    rs = np.random.RandomState(5)
    r_all = torch.from_numpy(rs.standard_normal((2, 600, data.node_raw_features.shape[1])).astype(np.float32)).cuda()
    lo = sl.start - 2000
    r = torch.cat([r_all[0, lo:lo + n], r_all[1, lo:lo + n]]).contiguous()

    def loss_fn(emb):
        return (emb * r).sum() / n, r / n                           # mean over the LOCAL slice (the reducer's weight rescales it)
    flat.grad = None
    pf = model.prepare_batch_finish(model.prepare_batch_begin(data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl], k))
    model.train_step(pf, loss_fn, k, grad_ready=(lambda seg: reducer.segment_ready(seg, weight)) if reducer is not None else None)
    if reducer is not None:
        reducer.finish(weight)
    torch.cuda.synchronize()
    return flat.grad.clone()


def main():
    rank, world, local = fdist.init_from_env()
    if os.environ.get("FLID_BENCH_SHARE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    from flid_amd.models.MemoryModel import MemoryModel
    from flid_amd.models.TGAT import TGAT
    from flid_amd.sweep import regenerate_embeddings
    from flid_amd.synth import wikipedia_like
    from flid_amd.utils.utils import get_neighbor_sampler
    dev = f"cuda:{local}"
    data = wikipedia_like(num_edges=6000, seed=0, zero_node_feat=False)
    sampler = get_neighbor_sampler(data, "recent", seed=0)

    # ---- (i) TGAT gradient: 2 shards of 300 edges + weighted all-reduce == one process on the 600 edges
    for mode, tol in ((0, 3e-6), (1, 1e-4)):
        lib().tg_set_gemm_mode(mode)
        torch.manual_seed(0)
        model = TGAT(data.node_raw_features, data.edge_raw_features, sampler, time_feat_dim=100, num_layers=2, num_heads=2, dropout=0.0,
                     device=dev).to(dev).train()
        fdist.broadcast_parameters(model)
        flat = model.flatten_parameters()
        reducer = fdist.GradAllReducer([flat])
        lo, hi = fdist.shard_bounds(600, rank, world)
        g_dp = flat_grad(model, flat, data, slice(2000 + lo, 2000 + hi), 10, (hi - lo) / 600.0, reducer)
        g_one = flat_grad(model, flat, data, slice(2000, 2600), 10, 1.0)
        err = float((g_dp - g_one).abs().max()) / max(1e-12, float(g_one.abs().max()))
        assert err <= tol, f"rank {rank} mode {mode}: reduced gradient differs from the full-batch gradient by {err:.2e} of its largest entry"
        # the same through the native stepper (csrc/tg_step.hip): its backward hands the root layer's block to the reducer through the
        # C callback (tg_grad_ready_fn) while the lower layer's backward is still being issued
        model.enable_native_step(1200, 10)
        g_dp_n = flat_grad(model, flat, data, slice(2000 + lo, 2000 + hi), 10, (hi - lo) / 600.0, reducer)
        err = float((g_dp_n - g_one).abs().max()) / max(1e-12, float(g_one.abs().max()))
        assert err <= tol, f"rank {rank} mode {mode}: native stepper, reduced gradient differs from the full-batch gradient by {err:.2e}"
    lib().tg_set_gemm_mode(1)

    # ---- (ii) TGN: replicas stay bit-identical over 5 sharded steps
    torch.manual_seed(0)
    tgn = MemoryModel(data.node_raw_features, data.edge_raw_features, sampler, time_feat_dim=100, model_name="TGN", num_layers=1,
                      num_heads=2, dropout=0.0, device=dev).to(dev).train()
    fdist.broadcast_parameters(tgn)
    tgn.memory_bank.__init_memory_bank__()
    for b in range(5):
        sl = slice(b * 200, (b + 1) * 200)
        with torch.no_grad():
            tgn.compute_shard_embeddings_and_advance(data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl],
                                                     data.edge_ids[sl], fdist.shard_bounds(200, rank, world), True, 10)
    state = torch.cat([tgn.memory_bank.node_memories.data.reshape(-1), tgn.memory_bank.node_last_updated_times.data.reshape(-1)])
    both = [torch.empty_like(state) for _ in range(world)]
    dist.all_gather(both, state)
    assert all(torch.equal(both[0], b_) for b_ in both[1:]), "TGN replicas diverged"

    # ---- (ii-b) TGN fused step through the native stepper under data parallelism: every rank embeds its shard of the 200-edge batch with
    # the layer's gradient block handed to the reducer from C (grad_ready) and advances the replicated state with the whole batch; the
    # reduced gradient == the single-process gradient of the whole batch, the replicas' states stay identical
    torch.manual_seed(0)
    tn = MemoryModel(data.node_raw_features, data.edge_raw_features, sampler, time_feat_dim=100, model_name="TGN", num_layers=1,
                     num_heads=2, dropout=0.0, device=dev).to(dev).train()
    fdist.broadcast_parameters(tn)
    t1 = MemoryModel(data.node_raw_features, data.edge_raw_features, sampler, time_feat_dim=100, model_name="TGN", num_layers=1,
                     num_heads=2, dropout=0.0, device=dev).to(dev).train()
    t1.load_state_dict(tn.state_dict())
    fl_n, fl_1 = tn.flatten_parameters(), t1.flatten_parameters()
    red_n = fdist.GradAllReducer([fl_n])
    tn.enable_native_step(200, 10)
    t1.enable_native_step(200, 10)
    tn.memory_bank.__init_memory_bank__()
    t1.memory_bank.__init_memory_bank__()
    rs = np.random.RandomState(11)
    for b in range(4):
        sl = slice(b * 200, (b + 1) * 200)
        args_ = (data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl])
        lo, hi = fdist.shard_bounds(200, rank, world)
        r_full = torch.from_numpy(rs.standard_normal((2, 200, data.node_raw_features.shape[1])).astype(np.float32)).cuda()
        r_loc = torch.cat([r_full[0, lo:hi], r_full[1, lo:hi]]).contiguous()
        r_all = torch.cat([r_full[0], r_full[1]]).contiguous()
        w_ = (hi - lo) / 200.0
        fl_n.grad = None
        job = tn.prepare_batch_finish(tn.prepare_batch_begin(*args_, 10, (lo, hi), edge_ids=data.edge_ids[sl]))
        tn.train_step(job, data.edge_ids[sl], lambda e: ((e * r_loc).sum() / (hi - lo), r_loc / (hi - lo)), 10,
                      grad_ready=lambda seg: red_n.segment_ready(seg, w_))
        red_n.finish(w_)
        fl_1.grad = None
        job = t1.prepare_batch_finish(t1.prepare_batch_begin(*args_, 10, edge_ids=data.edge_ids[sl]))
        t1.train_step(job, data.edge_ids[sl], lambda e: ((e * r_all).sum() / 200.0, r_all / 200.0), 10)
        torch.cuda.synchronize()
        err = float((fl_n.grad - fl_1.grad).abs().max()) / max(1e-12, float(fl_1.grad.abs().max()))
        assert err <= 1e-4, f"rank {rank} batch {b}: TGN reduced gradient differs from the whole-batch gradient by {err:.2e}"
        assert torch.allclose(tn.memory_bank.node_memories.data, t1.memory_bank.node_memories.data, atol=1e-6), "TGN shard step: state differs"

    # ---- (iii) regeneration sweep: rank-interleaved chunks + all-gather == one rank
    torch.manual_seed(0)
    m2 = TGAT(data.node_raw_features, data.edge_raw_features, sampler, time_feat_dim=100, num_layers=2, num_heads=2, dropout=0.1,
              device=dev).to(dev)
    fdist.broadcast_parameters(m2)
    s_dp, d_dp = regenerate_embeddings(m2, data, 200, 10, chunk_edges=512, rank=rank, world=world)
    s_1, d_1 = regenerate_embeddings(m2, data, 200, 10, chunk_edges=512, rank=0, world=1)
    assert float((s_dp - s_1).abs().max()) < 2e-5 and float((d_dp - d_1).abs().max()) < 2e-5, "sharded sweep != single-rank stores"

    # ---- (iv) DyGFormer: data parallelism runs WHOLE batches per rank (the reference pads every batch to its own longest sequence,
    # models/DyGFormer.py:196-245, and the unmasked transformer sees the padding: splitting one batch over ranks would change the result).
    # The weighted all-reduce of the ranks' gradients == the mean of the two batches' gradients computed by one process.
    from flid_amd.models.DyGFormer import DyGFormer
    torch.manual_seed(0)
    dyg = DyGFormer(data.node_raw_features, data.edge_raw_features, sampler, 100, 50, 1, 2, 2, 0.0, 32, dev).to(dev).train()
    fdist.broadcast_parameters(dyg)
    params = [p_ for p_ in dyg.parameters() if p_.requires_grad]
    rs = np.random.RandomState(9)
    r_all = torch.from_numpy(rs.standard_normal((world, 2, 200, data.node_raw_features.shape[1])).astype(np.float32)).cuda()

    def dyg_grads(b):
        sl = slice(3000 + 200 * b, 3200 + 200 * b)
        for p_ in params:
            p_.grad = None
        se, de = dyg.compute_src_dst_node_temporal_embeddings(data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl])
        ((se * r_all[b, 0]).sum() + (de * r_all[b, 1]).sum()).div(200.0).backward()
        return [p_.grad.clone() for p_ in params]
    mine = dyg_grads(rank)
    for p_, g_ in zip(params, mine):
        p_.grad = g_
    red = fdist.GradAllReducer(params)
    red.reduce(weight=1.0 / world)
    g_dp = [p_.grad.clone() for p_ in params]
    every = [dyg_grads(b) for b in range(world)]
    for i, g_ in enumerate(g_dp):
        want = sum(e[i] for e in every) / world
        err = float((g_ - want).abs().max()) / max(1e-12, float(want.abs().max()))
        assert err <= 2e-4, f"rank {rank}: DyGFormer reduced gradient of parameter {i} off by {err:.2e}"

    # ---- (iv-b) the same through the native step (csrc/tg_dyg.hip): every rank's train_step leaves its batch's gradient in the flat
    # parameter's block, one in-place all-reduce of that block, then the library's Adam -- replicas stay identical
    from flid_amd import ops
    from flid_amd.optim import FlatAdam
    flat = dyg.flatten_parameters()
    stp = dyg.enable_native_step(200)
    opt = FlatAdam([flat], lr=1e-4)
    red_n = fdist.GradAllReducer([flat])

    def native_grad(b):
        sl = slice(3000 + 200 * b, 3200 + 200 * b)
        rb = torch.cat([r_all[b, 0], r_all[b, 1]]).contiguous()
        opt.zero_grad(set_to_none=True)
        dyg.train_step(data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl],
                       lambda emb: (ops.weighted_sum(emb, rb, 1.0 / 200.0), rb / 200.0))
        return stp.grad.clone()
    every_n = [native_grad(b) for b in range(world)]
    native_grad(rank)
    red_n.reduce(weight=1.0 / world)
    want = sum(every_n) / world
    err = float((flat.grad - want).abs().max()) / max(1e-12, float(want.abs().max()))
    assert err <= 2e-4, f"rank {rank}: DyGFormer native step, reduced gradient off by {err:.2e}"
    # (against the autograd path's gradients of (iv), tensor by tensor)
    base = flat.data_ptr()
    for i, (p_, e0) in enumerate(zip(params, every[rank])):
        o = (p_.data_ptr() - base) // 4
        gn = every_n[rank][o:o + p_.numel()].view(p_.shape)
        err = float((gn - e0).abs().max()) / max(1e-12, float(e0.abs().max()))
        assert err <= 2e-4, f"rank {rank}: DyGFormer native gradient of parameter {i} differs from autograd by {err:.2e}"
    opt.step()
    both = [torch.empty_like(flat.data) for _ in range(world)]
    dist.all_gather(both, flat.data.contiguous())
    assert all(torch.equal(both[0], b_) for b_ in both[1:]), "DyGFormer replicas diverged after the native step"

    dist.barrier()
    if rank == 0:
        print("DIST-GPU-OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
