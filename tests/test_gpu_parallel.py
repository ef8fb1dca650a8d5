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


def test_bench_with_two_ranks_on_one_gpu_through_the_stand_in(gpu):
    """bench.py's world > 1 path itself -- self-launch, TCP rendezvous, the library's communicator attached to the Net, the
    barrier's all-reduce of dt / padded / real frames over the ranks, rank 0's one JSON line with the exchange report -- executed
    with TWO ranks before an 8-GPU node ever sees it.  Both ranks share GPU 0 (EESEN_BENCH_SHARE_GPU), the
    collective is the test-only stand-in for librccl.so (tests/native/fake_rccl.hip), and two processes cannot both hold a
    persistent recurrence grid on one GPU, so the per-step kernels run: this is plumbing, never a measurement."""
    import json
    from tests.test_gpu_multirank import fake_rccl_path
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EESEN_RCCL_LIBRARY=fake_rccl_path(), EESEN_PERSISTENT="0", FAKE_RCCL_QUIET="1", EESEN_BENCH_SHARE_GPU="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--main-only", "--T", "120"],
                       capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch_utterances"] == 64 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["padded_frames_per_step"] == 2 * 32 * 120           # the ranks' frames were summed over the communicator
    assert d["value"] == pytest.approx(d["config"]["padded_frames_per_step"] * 1e3 / d["ms_per_step"], rel=1e-6)
    ex = d["roofline"]["exchange"]
    assert ex["bound"] == "xgmi" and ex["peak"] == pytest.approx(7 * 153.0)
    assert len(ex["buckets"]) == 5 and [b["layer"] for b in ex["buckets"]] == [4, 3, 2, 1, 0]     # affine, then the LSTM layers top-down
    assert abs(sum(b["MB"] for b in ex["buckets"]) - 84.8) < 0.2                                   # 84.8 MB at cfg2
    assert all(b["ms"] > 0 for b in ex["buckets"]) and ex["ms_per_step"] > 0 and ex["exposed_ms_per_step"] >= 0
    assert d["phase_ms_per_step"]["allreduce"] == pytest.approx(ex["ms_per_step"]) and "allreduce_exposed" in d["phase_ms_per_step"]
