#!/usr/bin/env python3
"""Headline benchmark: edges/sec through TGAT.compute_src_dst_node_temporal_embeddings fwd + bwd (+ Adam step)
on the Wikipedia-shape synthetic graph (BASELINE.json configs[1]: batch 600, 20 temporal neighbors, 2 attention layers).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the hot path over one batch of 600 edges per GPU (weak scaling: rank r of N takes batch step*N + r of
the chronological stream), a scalar loss on both outputs, backward to every backbone parameter, the RCCL gradient all-reduce
when N > 1, and the Adam update.  The feature tables and the graph are resident in HBM; a batch's ids / times enter each call as the
reference's trainers hand them over -- host numpy int64 / float64 -- and their host-side cast, pinned staging and H2D copy are INSIDE
the timed region (on the prefetch stream, like the rest of the sampler work).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK = 8.0e12            # B/s, MI355X_MICROARCH.md chip table
DTYPE = "f32 (products as split-bf16x3 MFMA: hi*hi + hi*lo + lo*hi, fp32 accumulate; everything else fp32)"
MFMA_F32_PEAK = 157.3e12     # FLOP/s dense f32-input MFMA
BF16_PEAK = 2.5e15           # FLOP/s dense bf16 MFMA (MI355X_MICROARCH.md)
BATCH, K, L, H, DN, DE, DT = 600, 20, 2, 2, 172, 172, 100


def kernel_src_sha(files=("tg_attn_fast.hip", "tg_attn.hip", "tg_common.h")):
    """identity of the attention kernels' SOURCE: the PMC traffic figures under profiles/ are valid for the build they were taken on"""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(REPO, "flid_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def tgat_bytes_per_edge(k=K, layers=L, dn=DN, de=DE):
    """de-duplicated minimum bytes per root, forward: node rows + edge rows + sampler + output row;
    an edge has 2 roots; backward re-reads the gathered inputs once."""
    node = sum(k ** l for l in range(layers + 1)) * 4 * dn
    edge = sum(k ** l for l in range(1, layers + 1)) * 4 * de
    samp = sum(k ** l for l in range(layers)) * (16 + 64 + 16 * k)
    root = node + edge + samp + 4 * dn
    return 2 * root * 2


def attn_bytes_per_instance(k=K, heads=H, dn=DN, de=DE, dt=DT, backward=False, activations=False):
    """HBM bytes of ONE attention instance in tg_attn_fwd / tg_attn_bwd.
    activations=False: the ALGORITHMIC bytes of SURVEY.md 8(d) -- what any implementation of the reference's gather has to move:
    k neighbor rows (node + edge, fp32) + k x 16 B of slot lists (feat idx, edge idx, nbr id, dt) = 27 840 B at k = 20, 172 / 172.
    activations=True: what this design moves on top (its own intermediates, not part of the algorithmic figure): u in + agg out
    (+ prob); backward: u, agg, dagg in, du out (+ prob)."""
    dk = dn + de + dt
    rows = k * 4 * (dn + de) + k * 16
    if not activations:
        return rows
    vec = heads * dk * 4
    return (4 if backward else 2) * vec + heads * k * 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--roofline-kernel", default="attn_bwd", choices=["attn_fwd", "attn_bwd", "gemm"])
    ap.add_argument("--chunk-edges", type=int, default=4096, help="--mode sweep: edges per chunk of the regeneration sweep")
    ap.add_argument("--mode", default="train", choices=["train", "fwd", "lp", "sweep"],
                    help="train = fwd+bwd+Adam (the headline metric); fwd = eval-mode embedding regeneration sweep (M_step.py:456-509); "
                         "lp = link-prediction train step of EM_warmup.py:126-231: src, dst and a random negative dst embedded in ONE "
                         "call (3 roots per edge), MergeLayer head, BCE loss, Adam on backbone + head; "
                         "sweep = the whole embedding-regeneration sweep of M_step.py:456-509 over EVERY edge of the graph into the (E, 172) "
                         "stores (flid_amd.sweep.regenerate_embeddings; full-graph sampler, chunked, prefetched)")
    ap.add_argument("--model", default="tgat", choices=["tgat", "tgn", "dygformer", "tcl", "graphmixer"],
                    help="tgat = BASELINE configs[1] (the headline) / configs[4]; tgn = configs[2] (Reddit-shape, memory + GRU update + "
                         "message scatter); dygformer = configs[3] (Reddit-shape, first-hop sequence transformer)")
    ap.add_argument("--workload", default=None, choices=["wikipedia", "reddit", "scale"],
                    help="wikipedia = BASELINE configs[1] (default for tgat); reddit = configs[2]/[3] (default for tgn / dygformer); "
                         "scale = SURVEY 8d config 5 (10 M nodes / 100 M edges, tables hashed into HBM: 79 GB resident)")
    ap.add_argument("--scale-users", type=int, default=9_000_000)
    ap.add_argument("--scale-items", type=int, default=1_000_000)
    ap.add_argument("--scale-edges", type=int, default=100_000_000)
    ap.add_argument("--no-flat", action="store_true", help="per-tensor parameters / gradients / Adam as in the reference's trainers")
    ap.add_argument("--gemm-mode", type=int, default=None, help="tg_set_gemm_mode override (experiments)")
    ap.add_argument("--trace-steps", action="store_true", help="per-step GPU times (events) to stderr")
    ap.add_argument("--event-every", type=int, default=10, help="HIP events around the roofline kernel's launches in every n-th timed step "
                    "(an event pair costs ~10 us of stream idle time per launch -- round 5, kernel trace: +20 us on a step that carries them, "
                    "whatever n; 1 = every step)")
    ap.add_argument("--host-profile", action="store_true", help="host issue cost per phase of a fused step (GPU idle at each step start) to stderr")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true", help="skip the untimed per-family profiling pass")
    ap.add_argument("--cpu-sample-edges", type=int, default=1800)
    ap.add_argument("--autograd", action="store_true",
                    help="train mode: loss and backward through torch autograd (loss.backward()) instead of the fused step "
                         "TGAT.train_step (same kernels, no autograd graph, loss scalar from one HIP reduction)")
    ap.add_argument("--python-step", action="store_true",
                    help="fused step issued by the Python engine (flid_amd/engine.py) instead of the native stepper (csrc/tg_step.hip): A/B")
    ap.add_argument("--simulate-world", type=int, default=0,
                    help="--model tgn on ONE GPU: one rank's step of an N-GPU job (global batch N x 600, this rank embeds its 600-edge shard and "
                         "advances the replicated state with the whole batch; no collective): what the replicated advance costs as N grows")
    ap.add_argument("--master-port", type=int, default=29533)
    ap.add_argument("--no-merged", action="store_true", help="tg_set_layer_merged(0): the reference's four separate projections per layer")
    ap.add_argument("--no-grouped", action="store_true", help="tg_set_wgrad_grouped(0): one exact product + one column sum per weight gradient")
    ap.add_argument("--merged-min-rows", type=int, default=None, help="tg_set_merged_min_rows override (experiments)")
    ap.add_argument("--overlap", type=int, default=None, help="tg_set_overlap override (experiments): 0 = weight gradients on the main stream")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # launched plainly with --gpus N: become the launcher.  The ranks are CHILD processes started before this process has
        # touched the GPU (nothing above imports torch.cuda state); their exit code is ours.
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))

    if args.workload is None:
        args.workload = "wikipedia" if args.model == "tgat" else "reddit"
    if args.model != "tgat":
        return bench_memory_or_sequence_model(args)

    from flid_amd import dist as fdist
    from flid_amd import ops
    from flid_amd.models.TGAT import TGAT
    from flid_amd.synth import scale_like, wikipedia_like
    from flid_amd.utils.utils import get_neighbor_sampler

    if args.gemm_mode is not None:
        from flid_amd._lib import lib as _l
        _l().tg_set_gemm_mode(args.gemm_mode)
    if args.no_merged or args.no_grouped:
        from flid_amd._lib import lib as _l
        if args.no_merged:
            _l().tg_set_layer_merged(0)
        if args.no_grouped:
            _l().tg_set_wgrad_grouped(0)
    if args.merged_min_rows is not None:
        from flid_amd._lib import lib as _l
        _l().tg_set_merged_min_rows(args.merged_min_rows)
    if args.overlap is not None:
        from flid_amd._lib import lib as _l
        _l().tg_set_overlap(args.overlap)
    rank, world, local = fdist.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a ROCm device (the product has no CPU path)"
    if os.environ.get("FLID_BENCH_SHARE_GPU"):      # plumbing check of the N > 1 path on a 1-GPU box (with FLID_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if args.workload == "scale":
        t_gen = time.perf_counter()
        data = scale_like(args.scale_users, args.scale_items, args.scale_edges, seed=0)
        node_tab = ops.hash_features(args.scale_users + args.scale_items + 1, DN, 1, dev)
        edge_tab = ops.hash_features(args.scale_edges + 1, DE, 2, dev)
        args.no_cpu_baseline = True          # the oracle's python-list adjacency is infeasible at 2 x 10^8 entries
        workload = (f"scale synthetic ({args.scale_users + args.scale_items} nodes, {args.scale_edges} edges, hashed {DN}-d node / "
                    f"{DE}-d edge tables resident in HBM: {(node_tab.numel() + edge_tab.numel()) * 4 / 1e9:.1f} GB)")
    else:
        data = wikipedia_like(seed=0)
        node_tab, edge_tab = data.node_raw_features, data.edge_raw_features
        workload = "Wikipedia-shape synthetic (9227 nodes, 157474 edges, 172-d edge feats)"
    n_train = int(0.7 * data.num_interactions)
    if args.mode == "sweep":
        return bench_sweep(args, data, node_tab, edge_tab, workload, dev, rank, world)
    sampler = get_neighbor_sampler(data.slice(0, n_train), "recent", seed=0)       # train graph, as EM_warmup.py:71-76
    if args.workload == "scale" and rank == 0:
        print(f"[bench] scale workload built in {time.perf_counter() - t_gen:.1f} s", file=sys.stderr, flush=True)
    torch.manual_seed(0)
    model = TGAT(node_tab, edge_tab, sampler, time_feat_dim=DT, num_layers=L, num_heads=H,
                 dropout=args.dropout, device=str(dev)).to(dev).train()
    fdist.broadcast_parameters(model)
    # all 24 parameter tensors live in one flat parameter (state_dict unchanged): one gradient tensor per step, one Adam kernel,
    # one all-reduce operand.  --no-flat keeps the reference's per-tensor parameters.
    train_params = [model.flatten_parameters()] if not args.no_flat else list(model.parameters())
    if args.no_flat:
        opt = torch.optim.Adam(train_params, lr=1e-4, fused=True)                    # load_configs.py:119,123
    else:
        from flid_amd.optim import FlatAdam
        opt = FlatAdam(train_params, lr=1e-4)                                        # same update rule, one kernel (tg_adam_f32)
    reducer = fdist.GradAllReducer(train_params) if world > 1 else None

    total_steps = args.warmup + args.steps
    n_prefetch = 2                      # batches prepared ahead of the one being computed (two-stage sampler prefetch)
    n_batches = n_train // BATCH
    first = n_batches // 2                                                           # mid-stream: histories are populated
    span = n_batches - first            # rank r takes batches r, r + N, ... of the second half of the train stream, cyclically:
                                        # long runs / many ranks re-visit batches (same shapes, same graph; synthetic input)

    def batch_slice(step):
        b = first + (step * world + rank) % span
        return slice(b * BATCH, (b + 1) * BATCH)

    # a batch enters the timed region as host numpy (int64 ids, float64 times), as the reference's trainers pass it
    # (PTCL/EM_warmup.py:128-130); --mode lp keeps its three id lists resident (it concatenates them on the device)
    host_batches, dev_batches = [], []
    for s in range(total_steps + n_prefetch):          # the last timed step still prefetches like every other
        sl = batch_slice(s)
        host_batches.append((np.ascontiguousarray(data.src_node_ids[sl], dtype=np.int64), np.ascontiguousarray(data.dst_node_ids[sl], dtype=np.int64),
                             np.ascontiguousarray(data.node_interact_times[sl], dtype=np.float64)))
        if args.mode == "lp":
            dev_batches.append((torch.from_numpy(data.src_node_ids[sl].astype(np.int32)).to(dev),
                                torch.from_numpy(data.dst_node_ids[sl].astype(np.int32)).to(dev),
                                torch.from_numpy(data.node_interact_times[sl]).to(dev)))
    rw = torch.randn(2, BATCH, DN, device=dev)
    # the scalar loss of a step: mean over the batch of (src_emb . rw[0] + dst_emb . rw[1]) -- as a fused-step loss function
    # (value from one HIP reduction, gradient w.r.t. the embedding block = rw / (B Dn), a constant)
    rw_flat = rw.reshape(2 * BATCH, DN).contiguous()
    rw_grad = rw_flat / float(BATCH * DN)
    loss_out = torch.zeros(1, device=dev)

    def mean_loss(emb):
        return ops.weighted_sum(emb, rw_flat, 1.0 / (BATCH * DN), out=loss_out), rw_grad

    prepared, jobs = {}, {}
    lp_fused = False
    if args.mode == "lp":
        from flid_amd import engine
        from flid_amd.models.modules import MergeLayer
        torch.manual_seed(1)
        head = MergeLayer(DN, DN, DN, 1).to(dev)                                     # the reference's link predictor (EM_init.py)
        fdist.broadcast_parameters(head)
        head_opt = torch.optim.Adam(head.parameters(), lr=1e-4, fused=True)
        head_reducer = fdist.GradAllReducer(head.parameters()) if world > 1 else None
        n_items = 1000 if args.workload == "wikipedia" else args.scale_items
        first_item = int(data.dst_node_ids.min())
        rs_neg = np.random.RandomState(7 + rank)
        neg_batches = [torch.from_numpy(rs_neg.randint(first_item, first_item + n_items, BATCH).astype(np.int32)).to(dev)
                       for _ in range(total_steps + n_prefetch)]
        labels = torch.cat([torch.ones(BATCH, device=dev), torch.zeros(BATCH, device=dev)])
        bce = torch.nn.BCELoss()
        args.no_cpu_baseline = True

        lp_fused = not args.autograd and not args.no_flat
        if lp_fused:
            from flid_amd.heads import LinkPredictionLoss
            lp_loss = LinkPredictionLoss(head)

        def begin_lp(s_):
            src, dst, t = dev_batches[s_]
            return engine.prepare_begin(sampler.graph, [src, dst, neg_batches[s_]], [t, t, t], K, L)

    if args.mode == "fwd":
        model.eval()
        args.no_cpu_baseline = True
        args.roofline_kernel = "attn_fwd"

    fused = args.mode == "train" and not args.autograd and not args.no_flat
    native = (fused or lp_fused) and not args.python_step
    if native:
        # every launch of a step issued by the library (one C call per half of the preparation, one for the forward, one for backward + Adam)
        model.enable_native_step((3 if args.mode == "lp" else 2) * BATCH, K)
        if args.mode == "lp":
            neg_host = [b.cpu().numpy().astype(np.int64) for b in neg_batches]

            def begin_lp(s_):
                src, dst, t = host_batches[s_]
                return model.prepare_roots_begin([src, dst, neg_host[s_]], t, K)

    def drop_prepared():
        prepared.clear()
        jobs.clear()
        if native:
            model._stepper.reset()

    def begin(s_):
        return begin_lp(s_) if args.mode == "lp" else model.prepare_batch_begin(*host_batches[s_], K)

    def finish(job):
        from flid_amd import engine as _e
        return _e.prepare_finish(job) if (args.mode == "lp" and not native) else model.prepare_batch_finish(job)

    def prefetch(s):
        """sampler work of FUTURE batches on the side stream (graph only, weight-independent -- the data-loader style prefetch):
        batch s+1's second half (its distinct-row count was copied to pinned memory a step ago: no wait), batch s+2's first half"""
        if s not in prepared:                                      # cold start
            prepared[s] = finish(jobs.pop(s) if s in jobs else begin(s))
        if s + 1 < len(host_batches) and s + 1 not in prepared:
            prepared[s + 1] = finish(jobs.pop(s + 1) if s + 1 in jobs else begin(s + 1))
        if s + 2 < len(host_batches) and s + 2 not in jobs:
            jobs[s + 2] = begin(s + 2)

    def step_lp(s):
        prefetch(s)
        opt.zero_grad(set_to_none=True)
        head_opt.zero_grad(set_to_none=True)
        if lp_fused:
            # head + sigmoid + BCE with explicit forward / backward on the HIP kernels (flid_amd.heads.LinkPredictionLoss), the backbone
            # through train_step: no autograd graph anywhere in the step
            if native and reducer is None:
                model.train_step(prepared.pop(s), lp_loss, K, optimizer=opt)             # Adam on the backbone inside the backward's call
                head_opt.step()
                return
            model.train_step(prepared.pop(s), lp_loss, K, grad_ready=reducer.segment_ready if reducer is not None else None)
            if reducer is not None:
                reducer.finish()
                head_reducer.reduce()
            opt.step()
            head_opt.step()
            return
        emb = model.compute_node_temporal_embeddings(prepared.pop(s), None, L, K)    # (3 B, Dn): src | dst | negative dst
        se, de_, ne = emb[:BATCH], emb[BATCH:2 * BATCH], emb[2 * BATCH:]
        prob = torch.cat([head(se, de_), head(se, ne)]).squeeze(1).sigmoid()         # EM_warmup.py:178-186
        loss = bce(prob, labels)
        loss.backward()
        if reducer is not None:
            reducer.reduce()
            head_reducer.reduce()
        opt.step()
        head_opt.step()

    def step(s):
        if args.mode == "lp":
            return step_lp(s)
        prefetch(s)
        if args.mode == "fwd":
            with torch.no_grad():
                model.compute_src_dst_node_temporal_embeddings(prepared.pop(s), None, None, K)
            return
        opt.zero_grad(set_to_none=True)
        if fused:
            # forward, loss, backward: no autograd graph; N > 1: the root layer's gradient block starts its all-reduce as soon as that
            # layer's backward is queued, under the lower layer's backward
            if native and reducer is None:
                model.train_step(prepared.pop(s), mean_loss, K, optimizer=opt)           # Adam inside the backward's call
                return
            model.train_step(prepared.pop(s), mean_loss, K, grad_ready=reducer.segment_ready if reducer is not None else None)
            if reducer is not None:
                reducer.finish()
            opt.step()
            return
        else:
            se, de_ = model.compute_src_dst_node_temporal_embeddings(prepared.pop(s), None, None, K)
            loss = torch.addcmul(se * rw[0], de_, rw[1]).mean()      # the same scalar through torch autograd
            loss.backward()
        if reducer is not None:
            reducer.reduce()
        opt.step()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for s in range(args.warmup):
        step(s)
    torch.cuda.synchronize()
    # timed region: HIP events (C side, on the launch stream) around the roofline kernel's launches only -- 2 per step for the
    # attention kernels; timing every GEMM launch as well costs ~15 % wall, so the per-family breakdown is a second, untimed pass
    # the events sit on the launch stream and each costs a few us of idle time around the launch they bracket: every
    # --event-every-th step of the timed region carries them, the others run as the product does
    ops.profile_enable(args.roofline_kernel)
    ops.profile_collect(args.roofline_kernel)
    ops.profile_enable(False)
    barrier()
    marks = []
    ev_every = max(1, args.event_every)
    t0 = time.perf_counter()
    for s in range(args.warmup, total_steps):
        if ev_every == 1 or (s - args.warmup) % ev_every == 0:
            ops.profile_enable(args.roofline_kernel)
            step(s)
            ops.profile_enable(False)
        else:
            step(s)
        if args.trace_steps:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((ev, time.perf_counter() - t0))
    host_issue = time.perf_counter() - t0          # the host has ISSUED every timed step (it runs ahead of the GPU when it can)
    barrier()
    elapsed = time.perf_counter() - t0
    if args.trace_steps and rank == 0:
        print(f"[bench] host issue time {host_issue / args.steps * 1e3:.3f} ms/step, wall {elapsed / args.steps * 1e3:.3f} ms/step", file=sys.stderr)
        gpu = [marks[i - 1][0].elapsed_time(marks[i][0]) for i in range(1, len(marks))]
        host = [marks[i][1] - marks[i - 1][1] for i in range(1, len(marks))]
        print("[bench] per-step GPU ms :", " ".join(f"{x:.2f}" for x in gpu), file=sys.stderr)
        print("[bench] per-step host ms:", " ".join(f"{x * 1e3:.2f}" for x in host), file=sys.stderr)
    if args.host_profile and fused and rank == 0:
        # what the HOST spends issuing one step when nothing is queued ahead of it (the GPU can only be as fast as this)
        acc = {"prefetch": 0.0, "train_step": 0.0, "adam": 0.0}
        drop_prepared()
        for s in range(args.warmup, total_steps):
            torch.cuda.synchronize()
            a = time.perf_counter()
            prefetch(s)
            b = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            model.train_step(prepared.pop(s), mean_loss, K, optimizer=opt if native else None)
            c = time.perf_counter()
            if not native:
                opt.step()
            d = time.perf_counter()
            if s > args.warmup:
                acc["prefetch"] += b - a; acc["train_step"] += c - b; acc["adam"] += d - c
        print("[bench] host issue us/step (idle GPU): " + ", ".join(f"{k} {v / (args.steps - 1) * 1e6:.0f}" for k, v in acc.items()), file=sys.stderr)
    fam = {args.roofline_kernel: ops.profile_collect(args.roofline_kernel)}
    others = [t for t in ("attn_fwd", "attn_bwd", "gemm") if t != args.roofline_kernel]
    if not args.no_breakdown:
        ops.profile_enable(others)
        drop_prepared()
        for s in range(args.warmup, total_steps):          # same batches again (weights have moved on; shapes are identical)
            step(s)
        fam.update({tag: ops.profile_collect(tag) for tag in others})
    ops.profile_enable(False)
    dist_info = None
    if world > 1:
        mine = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(every, mine)
        per_rank = [float(t.item()) for t in every]
        elapsed = max(per_rank)
        # the collective alone: the flat gradient (one bucket) reduced 20 times back to back
        gbuf = torch.zeros(sum(p.numel() for p in train_params), device=dev)
        for _ in range(3):
            torch.distributed.all_reduce(gbuf)
        torch.cuda.synchronize()
        t_ar = time.perf_counter()
        for _ in range(20):
            torch.distributed.all_reduce(gbuf)
        torch.cuda.synchronize()
        dist_info = {"rccl_ranks": world, "backend": torch.distributed.get_backend(),
                     "per_rank_ms_per_step": [round(e / args.steps * 1e3, 4) for e in per_rank],
                     "allreduce_ms": round((time.perf_counter() - t_ar) / 20 * 1e3, 4), "allreduce_floats": int(gbuf.numel()),
                     "overlap": "root-layer block reduced under the layer-1 backward (GradAllReducer.segment_ready)" if fused else "none"}
    breakdown = {k_: round(v[0] / (len(range(0, args.steps, max(1, args.event_every))) if k_ == args.roofline_kernel else args.steps), 4)
                 for k_, v in fam.items()}

    edges = args.steps * BATCH * world
    value = edges / elapsed
    from flid_amd._lib import lib as _lib_
    gemm_mode_now = int(_lib_().tg_get_gemm_mode())
    # the reference's own row-for-row recursion (no sharing of repeated (node, time) rows, so every occurrence draws its own dropout
    # mask): a short extra run, reported beside the headline
    dedupe_off = None
    if args.mode == "train" and world == 1 and not args.no_breakdown:
        from flid_amd import engine as _eng
        _eng.DEDUPE = False
        try:
            drop_prepared()
            if native:
                model.enable_native_step(2 * BATCH, K)      # (row sharing is a property of the stepper, read at its creation)
            n_off = min(20, args.steps)
            for s in range(args.warmup, args.warmup + 3):
                step(s)
            torch.cuda.synchronize()
            drop_prepared()
            t1 = time.perf_counter()
            for s in range(args.warmup, args.warmup + n_off):
                step(s)
            torch.cuda.synchronize()
            dt_off = time.perf_counter() - t1
            dedupe_off = {"value": round(n_off * BATCH / dt_off, 1), "ms_per_step": round(dt_off / n_off * 1e3, 4), "steps": n_off,
                          "note": "engine.DEDUPE = False: 24 000 layer-1 instances per step instead of the ~13 k distinct ones"}
        finally:
            _eng.DEDUPE = True
            drop_prepared()
            if native:
                model.enable_native_step(2 * BATCH, K)
    # the same step with the reference's plain fp32 products (models/modules.py:190-245 are fp32 mm: tg_set_gemm_mode(0), exact f32-input
    # MFMA everywhere -- chains, packed weights and the second weight-gradient form are bypassed), and with BOTH relaxations off
    # (exact products + the reference's row-for-row recursion): short extra runs beside the headline
    exact_f32 = strict = None
    if args.mode == "train" and world == 1 and not args.no_breakdown and gemm_mode_now != 0 and fused:
        from flid_amd import engine as _eng

        def short_leg(n_leg=min(20, args.steps)):
            drop_prepared()
            if native:
                model.enable_native_step(2 * BATCH, K)
            for s in range(args.warmup, args.warmup + 3):
                step(s)
            torch.cuda.synchronize()
            drop_prepared()
            t1 = time.perf_counter()
            for s in range(args.warmup, args.warmup + n_leg):
                step(s)
            torch.cuda.synchronize()
            dt_ = time.perf_counter() - t1
            return {"value": round(n_leg * BATCH / dt_, 1), "ms_per_step": round(dt_ / n_leg * 1e3, 4), "steps": n_leg}
        _lib_().tg_set_gemm_mode(0)
        try:
            exact_f32 = dict(short_leg(), note="tg_set_gemm_mode(0): every forward and input-gradient product exact fp32 (f32-input MFMA), as the reference's mm; "
                                                  "the weight gradients leave in the grouped split-bf16x3 launch in this mode too (as in earlier rounds: "
                                                  "tg_set_wgrad_grouped(0) would make them exact products as well)")
            _eng.DEDUPE = False
            strict = dict(short_leg(), note="exact fp32 products AND engine.DEDUPE = False (24 000 layer-1 instances, per-root dropout "
                                            "masks): the reference's arithmetic and recursion, row for row")
        finally:
            _eng.DEDUPE = True
            _lib_().tg_set_gemm_mode(gemm_mode_now)
            drop_prepared()
            if native:
                model.enable_native_step(2 * BATCH, K)
    # fwd = half of fwd+bwd; the link-prediction step embeds 3 roots per edge instead of 2
    bpe = {"train": tgat_bytes_per_edge(), "fwd": tgat_bytes_per_edge() // 2, "lp": tgat_bytes_per_edge() * 3 // 2}[args.mode]

    # roofline of the selected kernel family: HIP events on its launch stream over the timed region
    ms, units, cnt = fam[args.roofline_kernel]
    secs = max(ms * 1e-3, 1e-12)
    sampled_steps = len(range(0, args.steps, ev_every))
    if args.roofline_kernel == "gemm":
        roof = {"bound": "mfma", "kernel": "gemm_tile_kernel (tg_gemm_f32*, all shapes of a step)",
                "achieved": round(units / secs / 1e12, 3), "peak": round(MFMA_F32_PEAK / 1e12, 1), "unit": "TFLOP/s",
                "frac": round(units / secs / MFMA_F32_PEAK, 4), "traffic": None, "launches": cnt,
                "avg_launch_ms": round(ms / max(1, cnt), 4), "flops_per_step": units / sampled_steps}
    else:
        bwd = args.roofline_kernel == "attn_bwd"
        # `units` counts instances x (algorithmic + activation) bytes on the C side; the roofline is priced on the ALGORITHMIC bytes
        # of SURVEY 8(d) alone (27 840 B per instance), the activation traffic of this design is listed beside it
        alg, act = attn_bytes_per_instance(backward=bwd), attn_bytes_per_instance(backward=bwd, activations=True)
        inst = units / (alg + act)
        roof = {"bound": "hbm", "kernel": ("attn_bwd_fast_kernel<2,2,*> (tg_attn_fast.hip; layer-1 + root launch)" if bwd else
                           "attn_fwd_fast_kernel<2,2> (layer 1) + attn_fwd_kernel (root)"),
                "achieved": round(inst * alg / secs / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(inst * alg / secs / HBM_PEAK, 4), "traffic": None, "launches": cnt,
                "avg_launch_ms": round(ms / max(1, cnt), 4), "bytes_per_instance": alg,
                "bytes_per_launch_avg": round(inst * alg / max(1, cnt), 1),
                "activation_bytes_per_instance": act, "achieved_incl_activations": round(units / secs / 1e9, 1),
                "instances_per_launch_avg": round(inst / max(1, cnt), 1)}
        # PMC passes of the same command (tools/traffic_from_pmc.py, separate rocprofv3 --pmc runs): reported only when they were
        # taken on THIS source of the attention kernels; a file from another build is named, not quoted
        tr = os.path.join(REPO, "profiles", "traffic_r05.json")
        if os.path.exists(tr) and args.workload == "wikipedia" and args.mode == "train":
            try:
                tj = json.load(open(tr))
                if tj.get("_kernel_src_sha") == kernel_src_sha():
                    # `traffic` = HBM bytes per launch, averaged over the family's launches as `achieved` is (the contract's figure);
                    # the pass's details (raw counters, the layer-1 launch alone, the profiled command) beside it
                    ent = tj.get(args.roofline_kernel) or {}
                    roof["traffic"] = ent.get("hbm_bytes_per_launch")
                    roof["traffic_detail"] = ent
                else:
                    roof["traffic_stale"] = {"file": "profiles/traffic_r05.json", "kernel_src_sha": tj.get("_kernel_src_sha"),
                                             "current_kernel_src_sha": kernel_src_sha(),
                                             "note": "PMC passes taken on another build of the attention kernels: not quoted"}
            except Exception:
                pass

    out = {
        "metric": "edges/sec (link-prediction train step: 3 roots/edge + MergeLayer head + BCE), TGAT" if args.mode == "lp" else
                  ("edges/sec (temporal-embedding fwd+bwd), TGAT Wikipedia, 1/2/4/8 MI355X" if args.mode == "train" else
                   "edges/sec (temporal-embedding fwd only, eval), TGAT Wikipedia") if args.workload == "wikipedia" else
                  "edges/sec (temporal-embedding %s), TGAT 10M-node / 100M-edge scale graph" % ("fwd+bwd" if args.mode == "train" else "fwd only, eval"),
        "value": round(value, 1), "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE if gemm_mode_now != 0 else "f32 (exact f32-input MFMA products)", "data": "synthetic",
        "config": {"workload": workload + " + TGAT L=2 H=2 T=100, batch 600 edges/GPU, 20 recent neighbors, dropout %.2f, %s"
                               % (args.dropout, {"train": "fwd+bwd+Adam (%s)" % (("fused step, native stepper" if native else "fused step") if fused else "autograd"), "fwd": "fwd (eval)",
                                                 "lp": "link-prediction step (%s)" % ("fused head + loss, no autograd graph" if args.mode == "lp" and lp_fused else "autograd")}[args.mode]),
                   "batch_per_gpu": BATCH, "global_batch": BATCH * world, "num_neighbors": K, "num_layers": L,
                   "parallelism": f"dp{world}"},
        "path_roofline": {"bytes_per_edge_fwd_bwd": bpe, "hbm_frac": round(value / world * bpe / HBM_PEAK, 4),
                          "edges_per_s_at_100pct": round(HBM_PEAK / bpe, 1)},
        "roofline": dict(roof, timed_steps=f"{sampled_steps} of {args.steps} (every {ev_every}th step of the timed region carries the HIP events)"),
        "breakdown_ms": breakdown,
    }
    if dedupe_off is not None:
        out["row_sharing_off"] = dedupe_off
    if exact_f32 is not None:
        out["exact_f32"] = exact_f32
    if strict is not None:
        out["strict"] = strict
    if dist_info is not None:
        out["distributed"] = dist_info
    if "gemm" in fam and args.roofline_kernel != "gemm" and fam["gemm"][2] > 0:
        # the dense projections are priced against the f32-input MFMA peak "alongside" (untimed second pass; all
        # product launches of a step: split-bf16, direct and tiled kernels; flops = 2 M N K as the reference's fp32 mm would do)
        g_ms, g_fl, g_cnt = fam["gemm"]
        out["roofline_mfma"] = {"bound": "mfma", "kernel": "tg_gemm_f32* (all product launches of a step)",
                                "achieved": round(g_fl / (g_ms * 1e-3) / 1e12, 2), "peak": round(MFMA_F32_PEAK / 1e12, 1),
                                "unit": "TFLOP/s", "frac": round(g_fl / (g_ms * 1e-3) / MFMA_F32_PEAK, 4), "launches": g_cnt,
                                "gflop_per_step": round(g_fl / args.steps / 1e9, 2),
                                # what the split-bf16 kernels actually issue: 3 (chains: 4) bf16 MFMAs per fp32 product term
                                "peak_bf16x3_equivalent": round(BF16_PEAK / 3 / 1e12, 1),
                                "frac_bf16x3_equivalent": round(g_fl / (g_ms * 1e-3) / (BF16_PEAK / 3), 4)}
        # the dense side (row-block chains, products, weight gradients) is the largest share of the step, not the named HBM-bound
        # attention kernel: its time, work and rate against what the split-bf16 kernels could issue
        out["dominant_family"] = {"family": "dense side: chain_fwd/bwd + product + weight-gradient launches (HIP events, untimed second pass)",
                                  "us_per_step": round(g_ms / args.steps * 1e3, 1),
                                  "share_of_step": round(g_ms / args.steps / (elapsed / args.steps * 1e3), 3),
                                  "gflop_per_step": round(g_fl / args.steps / 1e9, 2),
                                  "tflops": round(g_fl / (g_ms * 1e-3) / 1e12, 2),
                                  "frac_of_bf16x3_peak": round(g_fl / (g_ms * 1e-3) / (BF16_PEAK / 3), 4),
                                  "frac_of_f32_mfma_peak": round(g_fl / (g_ms * 1e-3) / MFMA_F32_PEAK, 4)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:        # reported at N = 1 only
        out["cpu_baseline"] = cpu_baseline(data, n_train, model, batch_slice(args.warmup), args.cpu_sample_edges)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def bench_sweep(args, data, node_tab, edge_tab, workload, dev, rank, world):
    """--mode sweep: M_step.py:456-509 over the whole stream (full-graph sampler, eval mode), timed end to end"""
    from flid_amd.models.TGAT import TGAT
    from flid_amd.sweep import regenerate_embeddings
    from flid_amd.utils.utils import get_neighbor_sampler
    sampler = get_neighbor_sampler(data, "recent", seed=1)                         # full graph, as train.py:707
    torch.manual_seed(0)
    model = TGAT(node_tab, edge_tab, sampler, time_feat_dim=DT, num_layers=L, num_heads=H, dropout=args.dropout, device=str(dev)).to(dev)
    E = data.num_interactions
    stores = (torch.zeros((E, DN), device=dev), torch.zeros((E, DN), device=dev))
    warm = min(E, args.warmup * args.chunk_edges)
    regenerate_embeddings(model, data, 200, K, args.chunk_edges, out=stores, first_edge=0, num_edges=warm, rank=rank, world=world)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    regenerate_embeddings(model, data, 200, K, args.chunk_edges, out=stores, rank=rank, world=world)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    chunks = -(-E // args.chunk_edges)
    bpe = tgat_bytes_per_edge() // 2
    out = {"metric": "edges/sec (embedding-regeneration sweep, fwd only, eval), TGAT Wikipedia", "value": round(E / elapsed, 1), "unit": "edges/s",
           "n_gpus": world, "steps": chunks, "warmup": args.warmup, "ms_per_step": round(elapsed / chunks * 1e3, 4), "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
           "config": {"workload": workload + f" + TGAT L=2 H=2 T=100, 20 recent neighbors, whole stream ({E} edges) in chunks of {args.chunk_edges}, "
                                             "full-graph sampler, eval mode, stores (E, 172) x 2 resident", "parallelism": f"dp{world}"},
           "path_roofline": {"bytes_per_edge_fwd": bpe, "hbm_frac": round(E / elapsed / world * bpe / HBM_PEAK, 4),
                             "edges_per_s_at_100pct": round(HBM_PEAK / bpe, 1)}, "sweep_seconds": round(elapsed, 4)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def bench_memory_or_sequence_model(args):
    """BASELINE configs[2] / configs[3]: TGN (MemoryModel) and DyGFormer on the Reddit-shape synthetic graph through the class API
    the reference's trainers call (host numpy ids per call: the 14-KB pinned H2D copy is inside the timed call), fwd + bwd + Adam per
    600-edge batch per GPU; TGN in the M-step order (positive batches, PTCL/M_step.py:224-257, detach_memory_bank after every step)."""
    from flid_amd import dist as fdist
    from flid_amd import ops
    from flid_amd.synth import reddit_like
    from flid_amd.utils.utils import get_neighbor_sampler
    rank, world, local = fdist.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a ROCm device (the product has no CPU path)"
    if os.environ.get("FLID_BENCH_SHARE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.gemm_mode is not None:
        from flid_amd._lib import lib as _l
        _l().tg_set_gemm_mode(args.gemm_mode)
    assert args.workload == "reddit", "tgn / dygformer are benchmarked on the Reddit-shape graph (BASELINE configs[2], [3])"
    data = reddit_like(seed=0)
    n_train = int(0.7 * data.num_interactions)
    sampler = get_neighbor_sampler(data.slice(0, n_train), "recent", seed=0)
    torch.manual_seed(0)
    if args.model == "tgn":
        from flid_amd.models.MemoryModel import MemoryModel
        model = MemoryModel(data.node_raw_features, data.edge_raw_features, sampler, DT, "TGN", 1, H, args.dropout, device=str(dev))
        model.memory_bank.__init_memory_bank__()
        desc = "TGN (MemoryModel) L=1 H=2 T=100, 20 recent neighbors, GRU memory, last-message aggregation"
    elif args.model == "tcl":
        from flid_amd.models.TCL import TCL
        model = TCL(data.node_raw_features, data.edge_raw_features, sampler, DT, 2, H, K + 1, args.dropout, str(dev))
        desc = "TCL L=2 H=2 T=100, 20 recent neighbors + the node (depth 21), exact-fp32 products"
    elif args.model == "graphmixer":
        from flid_amd.models.GraphMixer import GraphMixer
        model = GraphMixer(data.node_raw_features, data.edge_raw_features, sampler, DT, K, 2, 0.5, 4.0, args.dropout, str(dev))
        desc = "GraphMixer L=2, 20 tokens, 100 channels, time_gap 2000"
    else:
        from flid_amd.models.DyGFormer import DyGFormer
        model = DyGFormer(data.node_raw_features, data.edge_raw_features, sampler, DT, 50, 1, 2, 2, args.dropout, 32, str(dev))
        desc = "DyGFormer L=2 H=2 C=50 patch 1, max sequence 32"
    model = model.to(dev).train()
    fdist.broadcast_parameters(model)
    # TGN and DyGFormer: flat parameter + the native stepper (DyGFormer's --python-step is its autograd path)
    fused = ((args.model == "tgn") or (args.model == "dygformer" and not args.python_step)) and not args.autograd and not args.no_flat
    if fused:
        from flid_amd.optim import FlatAdam
        params = [model.flatten_parameters()]
        opt = FlatAdam(params, lr=1e-4)
    else:
        params = [p for p in model.parameters() if p.requires_grad]
        opt = torch.optim.Adam(params, lr=1e-4, fused=True)
    reducer = fdist.GradAllReducer(params) if world > 1 else None
    native = fused and not args.python_step
    wsim = args.simulate_world if (args.simulate_world > 1 and args.model == "tgn" and world == 1) else world
    if native and args.model == "dygformer":
        model.enable_native_step(BATCH)
    elif native:
        model.enable_native_step(BATCH * wsim, K)            # (the state advance covers the whole global batch on every rank)
    total_steps = args.warmup + args.steps
    n_batches = n_train // BATCH
    first = n_batches // 2
    span = n_batches - first
    rw = torch.randn(2, BATCH, DN, device=dev)
    rw_flat = rw.reshape(2 * BATCH, DN).contiguous()
    rw_grad = rw_flat / float(BATCH * DN)
    loss_out = torch.zeros(1, device=dev)

    def mean_loss(emb):
        return ops.weighted_sum(emb, rw_flat, 1.0 / (BATCH * DN), out=loss_out), rw_grad

    def batch(step):
        # TGN's state makes the call order significant: every rank walks the SAME chronological stream (rank r embeds its shard of
        # the global batch of world x 600 edges and advances the replicated state with the whole batch); DyGFormer: rank r takes
        # batch step * world + r
        if args.model == "tgn":
            n_gb = n_train // (BATCH * wsim)                 # global batches in the train stream; the second half is walked, cyclically
            b = n_gb // 2 + step % max(1, n_gb - n_gb // 2)
            return slice(b * BATCH * wsim, (b + 1) * BATCH * wsim)
        b = first + (step * world + rank) % span
        return slice(b * BATCH, (b + 1) * BATCH)

    jobs, prepared = {}, {}
    tgn_lp = args.model == "tgn" and args.mode == "lp"
    if tgn_lp:
        # the warm-up's link-prediction step on the memory model (PTCL/EM_warmup.py:126-238): negatives first (no state advance), then
        # positives, one BCE over both through the MergeLayer head, Adam on backbone + head
        assert native and world == 1, "--model tgn --mode lp runs the native stepper on one GPU"
        from flid_amd.heads import PairLinkLoss
        from flid_amd.models.modules import MergeLayer
        torch.manual_seed(1)
        head = MergeLayer(DN, DN, DN, 1).to(dev)
        head_opt = torch.optim.Adam(head.parameters(), lr=1e-4, fused=True)
        loss_neg, loss_pos = PairLinkLoss(head, False), PairLinkLoss(head, True)
        first_item, n_items = int(data.dst_node_ids.min()), int(data.dst_node_ids.max() - data.dst_node_ids.min() + 1)
        rs_neg = np.random.RandomState(7)
        neg_ids = [rs_neg.randint(first_item, first_item + n_items, BATCH).astype(np.int64) for _ in range(total_steps + 3)]
        args.no_cpu_baseline = True

        def lp_begin(s_):
            sl_ = batch(s_)
            t_ = data.node_interact_times[sl_]
            return (model.prepare_batch_begin(data.src_node_ids[sl_], neg_ids[s_], t_, K),
                    model.prepare_batch_begin(data.src_node_ids[sl_], data.dst_node_ids[sl_], t_, K, edge_ids=data.edge_ids[sl_]))

        def lp_step(s):
            # two batches are in preparation per step (4 slots): this step's pair was begun a step ago
            if s not in jobs:
                jobs[s] = lp_begin(s)
            jn, jp = (model.prepare_batch_finish(j_) for j_ in jobs.pop(s))
            opt.zero_grad(set_to_none=True)
            head_opt.zero_grad(set_to_none=True)
            model.train_step(jn, None, loss_neg, K, edges_are_positive=False, more=True)
            model.train_step(jp, data.edge_ids[batch(s)], loss_pos, K, optimizer=opt, accumulate=True)
            head_opt.step()
            if s + 1 < total_steps + 3:
                jobs[s + 1] = lp_begin(s + 1)

    def tgn_begin(s_):
        sl_ = batch(s_)
        return model.prepare_batch_begin(data.src_node_ids[sl_], data.dst_node_ids[sl_], data.node_interact_times[sl_], K,
                                         None if wsim == 1 else (rank * BATCH, (rank + 1) * BATCH), edge_ids=data.edge_ids[sl_])

    def step(s):
        if args.model == "tgn":
            n_gb_ = n_train // (BATCH * wsim)
            if s > 0 and s % max(1, n_gb_ - n_gb_ // 2) == 0:
                # the walk over the stream's second half starts over: the memory must not see time run backwards (the reference resets it
                # at every epoch start, PTCL/EM_warmup.py:121)
                model.memory_bank.__init_memory_bank__()
        if tgn_lp:
            return lp_step(s)
        sl = batch(s)
        a = (data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl])
        opt.zero_grad(set_to_none=True)
        if args.model == "tgn":
            # graph-only part of the NEXT batches on the side stream (ids, neighbor lookups, distinct touched nodes): two-stage, no wait
            if s not in prepared:
                prepared[s] = model.prepare_batch_finish(jobs.pop(s) if s in jobs else tgn_begin(s))
            if s + 1 not in prepared:
                prepared[s + 1] = model.prepare_batch_finish(jobs.pop(s + 1) if s + 1 in jobs else tgn_begin(s + 1))
            if s + 2 not in jobs:
                jobs[s + 2] = tgn_begin(s + 2)
            pf = prepared.pop(s)
            if native and reducer is None:                   # ... and Adam, every launch issued by the library
                model.train_step(pf, data.edge_ids[sl], mean_loss, K, optimizer=opt)
                return
            if native:                                       # N > 1: the layer's block starts its all-reduce under the GRU's backward + the state advance
                model.train_step(pf, data.edge_ids[sl], mean_loss, K, grad_ready=reducer.segment_ready)
                reducer.finish()
                opt.step()
                return
            if fused:                                        # forward, loss, backward, state advance: no autograd graph
                model.train_step(pf, data.edge_ids[sl], mean_loss, K)
                if reducer is not None:
                    reducer.reduce()
                opt.step()
                return
            if world > 1:
                se, de = model.compute_shard_embeddings_and_advance(pf, None, None, data.edge_ids[sl], (rank * BATCH, (rank + 1) * BATCH), True, K)
            else:
                se, de = model.compute_src_dst_node_temporal_embeddings(pf, None, None, data.edge_ids[sl], True, K)
        elif native:                                         # DyGFormer: forward + backward (+ Adam) as two library calls, no autograd graph
            if reducer is None:
                model.train_step(*a, mean_loss, optimizer=opt)
            else:
                model.train_step(*a, mean_loss)
                reducer.reduce()
                opt.step()
            return
        else:
            se, de = model.compute_src_dst_node_temporal_embeddings(*a)
        loss = torch.addcmul(se * rw[0], de, rw[1]).mean()
        loss.backward()
        if reducer is not None:
            reducer.reduce()
        opt.step()
        if args.model == "tgn":
            model.memory_bank.detach_memory_bank()                                    # M_step.py:325

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for s in range(args.warmup):
        step(s)
    torch.cuda.synchronize()
    fam_name = "attn_bwd" if args.model == "tgn" else "gemm"
    # the roofline family's launches are timed with HIP events on their stream: one or two launches of a TGN step (inside the timed
    # region), but 32 of a DyGFormer step -- two event creations each, ~1 ms of host time per step -- so the other models take the
    # family's times from `prof_steps` extra steps behind the timed region
    prof_inside = args.model == "tgn"
    prof_steps = 0 if (prof_inside or args.no_breakdown) else min(10, args.steps)
    ev_every = max(1, args.event_every)
    if prof_inside:
        ops.profile_enable(fam_name)
        ops.profile_collect(fam_name)
        ops.profile_enable(False)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.warmup, total_steps):
        # TGN: the family's HIP events ride in every `--event-every`-th timed step, as on the headline line (an event pair costs the
        # stream ~12 us of idle time per launch: in every step they were 18 of its 362 us)
        if prof_inside and (ev_every == 1 or (s - args.warmup) % ev_every == 0):
            ops.profile_enable(fam_name)
            step(s)
            ops.profile_enable(False)
        else:
            step(s)
    host_issue = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    if not prof_inside:
        ops.profile_enable(fam_name)
        ops.profile_collect(fam_name)
        for s in range(total_steps, total_steps + prof_steps):
            step(s)
        torch.cuda.synchronize()
    ms, units, cnt = ops.profile_collect(fam_name)
    ops.profile_enable(False)
    unit_steps = max(1, args.steps if prof_inside else prof_steps)
    dist_info = None
    if world > 1:
        mine = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(every, mine)
        per_rank = [float(t.item()) for t in every]
        elapsed = max(per_rank)
        # the collective alone: the flat gradient (one bucket) reduced 20 times back to back
        gbuf = torch.zeros(sum(p.numel() for p in params), device=dev)
        for _ in range(3):
            torch.distributed.all_reduce(gbuf)
        torch.cuda.synchronize()
        t_ar = time.perf_counter()
        for _ in range(20):
            torch.distributed.all_reduce(gbuf)
        torch.cuda.synchronize()
        dist_info = {"rccl_ranks": world, "backend": torch.distributed.get_backend(),
                     "per_rank_ms_per_step": [round(e / args.steps * 1e3, 4) for e in per_rank],
                     "allreduce_ms": round((time.perf_counter() - t_ar) / 20 * 1e3, 4), "allreduce_floats": int(gbuf.numel()),
                     "overlap": ("attention + merge layer block reduced under the GRU backward and the state advance (GradAllReducer.segment_ready)"
                                 if (args.model == "tgn" and native) else "none (one flat-bucket all-reduce between backward and optimizer)")}
    if args.model == "tgn" and native and (world > 1 or wsim > 1) and not tgn_lp:
        # every rank advances the REPLICATED memory / message state with the whole global batch (no exchange step) -- what
        # that costs a rank, from HIP events around the advance's launches in a few extra steps behind the timed region
        ops.profile_enable("tgn_advance")
        ops.profile_collect("tgn_advance")
        n_adv = min(10, args.steps)
        for s in range(total_steps, total_steps + n_adv):
            step(s)
        adv_ms, adv_edges, adv_cnt = ops.profile_collect("tgn_advance")
        ops.profile_enable(False)
        adv = {"state_advance_ms_per_step": round(adv_ms / max(1, adv_cnt), 4), "edges_filed_per_step": int(adv_edges / max(1, adv_cnt)),
               "state_advance_share_of_step": round(adv_ms / max(1, adv_cnt) / (elapsed / args.steps * 1e3), 4),
               "note": "replicated state advance over the whole global batch on every rank (no memory / message exchange); HIP events, untimed extra steps"}
        if dist_info is not None:
            dist_info["tgn_state_advance"] = adv
        else:
            dist_info = {"rccl_ranks": world, "tgn_state_advance": adv}
    value = args.steps * BATCH * world / elapsed
    secs = max(ms * 1e-3, 1e-12)
    if args.model == "tgn":
        # SURVEY 8d "Algorithmic bytes (TGN, L=1)": per edge fwd, 2 roots: node rows (raw + memory) 2 x 2 (1 + K) x 688, edge rows 2 K x 688,
        # sampler 2 x 400, lazy memory update <= 2 (1 + K) (616 + 172) x 4; the positive step adds 2 x (616 x 4 written + 172 x 4 r/w + 8);
        # backward re-reads the gathered inputs once
        fwd = 2 * 2 * (1 + K) * 4 * DN + 2 * K * 4 * DE + 2 * 400 + 2 * (1 + K) * (616 + 172) * 4
        bpe = 2 * fwd + 2 * (616 * 4 + 172 * 4 * 2 + 8)
        roof = {"bound": "hbm", "kernel": "attn_bwd_fast_kernel<2,2,true,false> (fused attention backward, compact memory table)",
                "achieved": round(units / secs / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(units / secs / HBM_PEAK, 4),
                "traffic": None, "launches": cnt, "avg_launch_ms": round(ms / max(1, cnt), 4),
                "bytes_per_instance": attn_bytes_per_instance(backward=True),
                "timed_steps": f"{len(range(0, args.steps, ev_every))} of {args.steps} (every {ev_every}th step of the timed region carries the HIP events)"}
        path = {"bytes_per_edge_fwd_bwd": bpe, "hbm_frac": round(value / world * bpe / HBM_PEAK, 4), "edges_per_s_at_100pct": round(HBM_PEAK / bpe, 1)}
        metric = ("edges/sec (link-prediction warm-up step: negatives then positives, MergeLayer head + BCE, memory update + message scatter), TGN Reddit"
                  if tgn_lp else "edges/sec (temporal-embedding fwd+bwd, memory update + message scatter), TGN Reddit, 1/2/4/8 MI355X")
    else:
        roof = {"bound": "mfma", "kernel": "tg_gemm_f32* (all product launches of a step: projections, feed-forward, attention products)",
                "achieved": round(units / secs / 1e12, 2), "peak": round(MFMA_F32_PEAK / 1e12, 1), "unit": "TFLOP/s",
                "frac": round(units / secs / MFMA_F32_PEAK, 4), "traffic": None, "launches": cnt, "avg_launch_ms": round(ms / max(1, cnt), 4),
                "gflop_per_step": round(units / unit_steps / 1e9, 2), "measured_over": f"{prof_steps} steps behind the timed region"}
        if args.gemm_mode != 0:
            # `frac` prices the reference's fp32 flops against the f32-input MFMA peak (what a plain fp32 mm could reach); the kernels
            # that ran issue three bf16 MFMAs per product term set, so the peak THEY could reach is 2.5 PFLOP/s / 3
            roof["frac_bf16x3_equivalent"] = round(units / secs / (BF16_PEAK / 3), 4)
            roof["peak_bf16x3_equivalent"] = round(BF16_PEAK / 3 / 1e12, 1)
        flops_edge = units / (unit_steps * BATCH)
        path = {"flops_per_edge_fwd_bwd": round(flops_edge, 1), "mfma_frac": round(value / world * flops_edge / MFMA_F32_PEAK, 4),
                "edges_per_s_at_100pct": round(MFMA_F32_PEAK / max(flops_edge, 1.0), 1)}
        if args.gemm_mode != 0:
            path["mfma_frac_bf16x3_equivalent"] = round(value / world * flops_edge / (BF16_PEAK / 3), 4)
        metric = "edges/sec (temporal-embedding fwd+bwd), %s Reddit, 1/2/4/8 MI355X" % {"dygformer": "DyGFormer", "tcl": "TCL", "graphmixer": "GraphMixer"}[args.model]
    out = {"metric": metric, "value": round(value, 1), "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": DTYPE if args.gemm_mode != 0 else "f32 (exact f32-input MFMA products)", "data": "synthetic",
           "config": {"workload": f"Reddit-shape synthetic (10984 nodes, 672447 edges, 172-d edge feats) + {desc}, batch 600 edges/GPU, "
                                  f"dropout {args.dropout:.2f}, host numpy ids per call, fwd+bwd+Adam ({('fused step, native stepper' if native else 'fused step') if fused else 'autograd'}{', neg-then-pos warm-up step' if tgn_lp else ''})",
                      "batch_per_gpu": BATCH, "global_batch": BATCH * wsim, "parallelism": f"dp{world}"},
           "path_roofline": path, "roofline": roof,
           # wall time until the host had ISSUED the timed steps: with a GPU-bound step this is queue back-pressure (~ the step time), not
           # the host's own cost -- that is the idle-GPU figure of tools/*_host_prof.py (profiles/*_host_issue.txt)
           "host_issue_wall_ms_per_step_incl_queue_backpressure": round(host_issue / args.steps * 1e3, 4)}
    if dist_info is not None:
        out["distributed"] = dist_info
    if wsim != world:
        out["simulated_world"] = {"ranks": wsim, "note": "ONE rank's step of a %d-GPU data-parallel TGN job on one GPU: its 600-edge shard embedded, "
                                  "the replicated memory / message state advanced with the whole %d-edge global batch, no collective; `value` "
                                  "counts this rank's 600 edges per step" % (wsim, BATCH * wsim)}
        args.no_cpu_baseline = True
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_model(args.model, data, n_train, model, batch(args.warmup), args.dropout)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def cpu_baseline_model(name, data, n_train, model, sl, dropout):
    """the oracle (kind "port") of TGN / DyGFormer on this box's host cores: same graph and weights; TGN: 3 chronological positive
    batches of 600 from a fresh memory (the oracle walks every node per call, as the reference does); DyGFormer: one batch; fwd + bwd"""
    from oracle import flid_oracle as O
    threads = torch.get_num_threads()
    sd = {k_: v.detach().cpu().clone() for k_, v in model.state_dict().items()}
    adj = O.build_adjacency(data.src_node_ids[:n_train], data.dst_node_ids[:n_train], data.edge_ids[:n_train], data.node_interact_times[:n_train])
    nt, et = torch.from_numpy(data.node_raw_features), torch.from_numpy(data.edge_raw_features)
    t0 = time.perf_counter()
    if name == "tgn":
        p = {k_: v.requires_grad_(True) for k_, v in sd.items() if v.is_floating_point() and "memory_bank" not in k_ and not k_.startswith("embedding_module.time_encoder")}
        orc = O.TGNOracle(nt, et, adj, p, 1, H, dropout=dropout, training=True)
        edges = 0
        for b in range(3):
            s_ = slice(sl.start + b * BATCH, sl.start + (b + 1) * BATCH)
            a, c = orc.src_dst(data.src_node_ids[s_], data.dst_node_ids[s_], data.node_interact_times[s_], data.edge_ids[s_], True, K)
            (a.mean() + c.mean()).backward()
            orc.detach()
            edges += BATCH
        what = "3 chronological positive batches of 600 from a fresh memory"
    else:
        p = {k_: (v.requires_grad_(True) if not (name == "graphmixer" and k_.startswith("time_encoder")) else v) for k_, v in sd.items()}
        s_ = slice(sl.start, sl.start + BATCH)
        if name == "tcl":
            orc = O.TCLOracle(nt, et, adj, p, 2, H, dropout=dropout, training=True)
            a, c = orc.src_dst(data.src_node_ids[s_], data.dst_node_ids[s_], data.node_interact_times[s_], K)
        elif name == "graphmixer":
            orc = O.GraphMixerOracle(nt, adj, p, 2, dropout=dropout, training=True)
            a, c = orc.src_dst(data.src_node_ids[s_], data.dst_node_ids[s_], data.node_interact_times[s_], K, 2000)
        else:
            orc = O.DyGFormerOracle(nt, et, adj, p, 50, 1, 2, 2, 32, dropout=dropout, training=True)
            a, c = orc.src_dst(data.src_node_ids[s_], data.dst_node_ids[s_], data.node_interact_times[s_])
        (a.mean() + c.mean()).backward()
        edges = BATCH
        what = "one batch of 600"
    dt = time.perf_counter() - t0
    return {"value": round(edges / dt, 2), "unit": "edges/s", "cores": threads, "kind": "port",
            "sample": f"{what}, fwd+bwd, oracle/flid_oracle.py on torch CPU ({threads} threads), {dt:.1f} s"}


def cpu_baseline(data, n_train, model, sl, sample_edges):
    """The oracle (CPU restatement of the reference path, parity-pinned) timed on this box's host cores on the first
    `sample_edges` edges of the first timed batch: same graph, same weights, fwd + bwd.  Baseline only."""
    from oracle import flid_oracle as O
    threads = torch.get_num_threads()
    p = {k_: v.detach().cpu().clone().requires_grad_(True) for k_, v in model.state_dict().items()}
    adj = O.build_adjacency(data.src_node_ids[:n_train], data.dst_node_ids[:n_train], data.edge_ids[:n_train],
                            data.node_interact_times[:n_train])
    orc = O.TGATOracle(torch.from_numpy(data.node_raw_features), torch.from_numpy(data.edge_raw_features), adj, p, L, H,
                       dropout=0.1, training=True)
    def run(lo, hi):
        s, d = orc.src_dst(data.src_node_ids[lo:hi], data.dst_node_ids[lo:hi], data.node_interact_times[lo:hi], K)
        (s.mean() + d.mean()).backward()
    run(sl.start, sl.start + 4)
    t0 = time.perf_counter()
    done = 0
    while done < sample_edges:                       # consecutive batches of the timed stream, BATCH edges per call as on the GPU
        nb = min(BATCH, sample_edges - done)
        run(sl.start + done, sl.start + done + nb)
        done += nb
    dt = time.perf_counter() - t0
    return {"value": round(sample_edges / dt, 2), "unit": "edges/s", "cores": threads, "kind": "port",
            "sample": f"{sample_edges} edges ({-(-sample_edges // BATCH)} batches of <= {BATCH}) of the timed stream, fwd+bwd, "
                      f"oracle/flid_oracle.py on torch CPU ({threads} threads), {dt:.1f} s"}


if __name__ == "__main__":
    main()
