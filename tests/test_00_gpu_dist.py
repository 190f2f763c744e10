"""Multi-rank GPU test of the data-parallel path: 2 RCCL ranks, one process per GPU, against the single-process result.

Runs FIRST in the GPU suite (file name) and only looks at torch.cuda.device_count() -- which does not initialise the GPU -- before it
starts the ranks as CHILD processes, so nothing here execs from a process that holds a GPU context.  On a 1-GPU box the two ranks SHARE
the GPU and reduce over gloo (FLID_BENCH_SHARE_GPU / FLID_DIST_BACKEND: the same worker, the same assertions, every rank-level code path
except RCCL's own transport); on a multi-GPU node they are one process per GPU over RCCL.  It checks
  (i)   TGAT: the flat gradient after GradAllReducer (segment_ready + finish, weight = local / global edges) == the single-process
        full-batch gradient, in the exact-product mode (tight) and in the default split-bf16 dispatch (1e-4 of the largest entry);
  (ii)  TGN: the replicated memory / message state is bit-identical on both ranks after 5 sharded steps;
  (iii) regenerate_embeddings(world=2) fills the same stores as a single rank;
  (iv)  DyGFormer (whole batches per rank: its padding is per batch): the reduced gradient == the mean of the batches' gradients;
  and (i) again through the native stepper, whose backward hands the root layer's gradient block to the reducer from C."""
import os
import subprocess
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(REPO, "tests", "dist_gpu_worker.py")


@pytest.mark.gpu
def test_two_rccl_ranks_match_single_process():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if torch.cuda.device_count() < 2:
        env.update(FLID_BENCH_SHARE_GPU="1", FLID_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", WORKER]
    r = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + "\n" + r.stderr[-4000:]
    assert "DIST-GPU-OK" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["tgat", "tgn", "dygformer"])
def test_bench_two_ranks_prints_one_line(model):
    """the program the driver launches at N > 1: `bench.py --gpus 2` as a child (it spawns torch.distributed.run itself, before any GPU call),
    two ranks -- sharing the one GPU over gloo on a 1-GPU box, as above -- through the native steps with the gradient hand-over from C
    (PTCL/EM_warmup.py:126-238 sharded by edges).  ONE JSON line on stdout: whole-job edges/s, every rank's ms per step, the
    collective's own time; TGN additionally the share of a step that the replicated state advance takes."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if torch.cuda.device_count() < 2:
        env.update(FLID_BENCH_SHARE_GPU="1", FLID_DIST_BACKEND="gloo")
    port = {"tgat": 29611, "tgn": 29612, "dygformer": 29613}[model]
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-breakdown",
           "--model", model, "--master-port", str(port)]
    r = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + "\n" + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    dist = d["distributed"]
    assert dist["rccl_ranks"] == 2 and len(dist["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in dist["per_rank_ms_per_step"])
    assert dist["allreduce_ms"] > 0 and dist["allreduce_floats"] > 900_000
    # whole-job value = the edges all ranks processed / the slowest rank's time
    per_gpu = d["config"]["batch_per_gpu"] if "batch_per_gpu" in d["config"] else 600
    assert abs(d["value"] - 2 * per_gpu * 3 / (max(dist["per_rank_ms_per_step"]) * 3e-3)) <= 0.02 * d["value"]
    if model == "tgn":
        adv = dist["tgn_state_advance"]
        assert adv["edges_filed_per_step"] == 2 * 600 and 0 < adv["state_advance_share_of_step"] < 1
