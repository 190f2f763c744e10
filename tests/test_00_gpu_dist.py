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
