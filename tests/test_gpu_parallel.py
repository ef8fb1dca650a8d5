"""-m gpu: the RCCL plumbing of the data-parallel exchange on ONE GPU (world_size 1): the torch view aliases the
library's gradient buffer (zero copy), all_reduce runs on it in place and the update consumes what it left.
Runs in a fresh interpreter because torch has to be imported before libeesen_hip.so (one shared HIP runtime)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_grad_view_is_zero_copy_and_allreduce_in_place(gpu):
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rccl_plumbing_check.py")
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_PLUMBING_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_runs_under_torchrun_with_one_rank(gpu):
    """The driver's multi-GPU launch line with N = 1 (same code path as N > 1 minus the collective)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--T", "100",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and '"metric"' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
