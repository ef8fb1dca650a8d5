"""-m gpu: the LIBRARY's data-parallel path (eesen_amd/csrc/comm.cpp: communicator, per-layer gradient buckets on the
communication stream, zero-gradient protocol with the liveness word, watchdog) executed with MORE THAN ONE RANK -- on the one
GPU a box of this pool has, through a test-only stand-in for librccl.so (tests/native/fake_rccl.hip, loaded by the library
itself via EESEN_RCCL_LIBRARY; its all-reduce is a real kernel on the stream it is given, ranks are separate processes).

  (i)   2 ranks x S = 16 == 1 process x S = 32 (SURVEY.md section 8e's parity statement), 3 steps, momentum 0.9 + <MaxGrad>;
  (ii)  uneven shards: ranks that run out of minibatches keep stepping with zero gradients until no rank is live, and the
        closing round does not move the model: equal to one process on the union of the minibatches;
  (iii) a rank that dies before its first collective does not hang the other: the watchdog aborts, EESEN_ERR_COMM;
  (iv)  one process, PERSISTENT recurrence kernels on, the stand-in's all-reduce kernels (32 workgroups x 512 threads, 25 MB
        buckets through PCIe) running on the communication stream under every backward recurrence, 200 steps at cfg2: no spin
        time-out, and what the overlap costs per step.
  (v)   (round 5) TWO ranks, each holding persistent grids: a shape whose grids are co-resident (H = 256, S = 32 per rank), every
        process sizing against half the device (EESEN_GPU_SHARE=2) -- the product's N > 1 configuration.
The other multi-process cases run the one-launch-per-step kernels (EESEN_PERSISTENT=0: at their shapes two processes' grids
would each want every CU); (iv) and (v) are where the co-residency hazard is exercised.
"""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

from eesen_amd import synth
from tests.util import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_multirank_worker.py")


def fake_rccl_path() -> str:
    """tests/native/libfake_rccl.so, built in-tree with hipcc (cross-compiles without a GPU; travels to the GPU box)."""
    src = os.path.join(ROOT, "tests", "native", "fake_rccl.hip")
    lib = os.path.join(ROOT, "tests", "native", "libfake_rccl.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        from eesen_amd import build
        r = subprocess.run([build.hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC", src, "-o", lib, "-lrt"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
    return lib


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def launch(mode, world, tmp_path, opts=None, env_extra=None, timeout=300):
    """Starts `world` worker processes (one rank each, all on GPU 0); returns ([npz dict | None per rank], [returncode], [stderr])."""
    port = _free_port()
    procs, outs = [], []
    for r in range(world):
        out = str(tmp_path / f"{mode}_rank{r}.npz")
        outs.append(out)
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   EESEN_RCCL_LIBRARY=fake_rccl_path(), EESEN_PERSISTENT="0", FAKE_RCCL_QUIET="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(env_extra or {})
        args = [sys.executable, WORKER, mode, out] + [f"{k}={v}" for k, v in (opts or {}).items()]
        procs.append(subprocess.Popen(args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    res, rcs, errs = [], [], []
    deadline = time.time() + timeout
    for p, out in zip(procs, outs):
        try:
            _, err = p.communicate(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            _, err = p.communicate()
            err += "\n[killed: timeout]"
        rcs.append(p.returncode)
        errs.append(err)
        res.append(dict(np.load(out)) if os.path.exists(out) else None)
    return res, rcs, errs


def _single_process(cfg, batches, lr=1e-3, momentum=0.9, max_grad=0.05, persistent="0"):
    """The same steps in ONE process without any exchange: the arbiter of the N-rank runs."""
    from eesen_amd.api import Net, Ctc
    old = os.environ.get("EESEN_PERSISTENT")
    os.environ["EESEN_PERSISTENT"] = persistent
    try:
        net = Net.from_layers(synth.make_model(max_grad=max_grad, **cfg))
    finally:
        if old is None:
            del os.environ["EESEN_PERSISTENT"]
        else:
            os.environ["EESEN_PERSISTENT"] = old
    net.SetTrainOptions(lr, momentum)
    ctc = Ctc()
    for b in batches:
        net.SetSeqLengths(b.lens)
        o = net.Propagate(b.feats)
        d = ctc.EvalParallel(b.lens, o, b.labels, want_pzx=False)
        ctc.ErrorRateMSeq(b.lens, o, b.labels, deferred=True)
        net.Backpropagate(d)
    net.Synchronize()
    return net.GetParams(), ctc.stats()


def _merge(batches):
    """One minibatch holding the utterances of several (each re-padded to the longest): what a single process with
    --num-sequence = the sum would have assembled."""
    D = batches[0].feats.shape[1]
    mats, labels = [], []
    for b in batches:
        f3 = b.feats.reshape(b.T, b.S, D)
        for s in range(b.S):
            mats.append(f3[: b.lens[s], s, :])
            labels.append(b.labels[s])
    lens = np.array([m.shape[0] for m in mats], np.int32)
    T, S = int(lens.max()), len(mats)
    feats = np.zeros((T, S, D), np.float32)
    for s, m in enumerate(mats):
        feats[: m.shape[0], s] = m
    return synth.Batch(feats=feats.reshape(T * S, D), lens=lens, labels=labels, T=T, S=S)


@pytest.mark.parametrize("defer", ["0", "1"])
def test_two_ranks_on_one_gpu_equal_one_process_on_the_whole_batch(gpu, tmp_path, defer):
    """(defer = "1": EESEN_COMM_DEFER -- the buckets issued behind the backward pass's last recurrence; same sums.)"""
    cfg = synth.config("small_bi"); cfg.update(S=32, T=60)
    res, rcs, errs = launch("parity", 2, tmp_path, dict(cfg="small_bi", S=32, T=60, steps=3), env_extra={"EESEN_COMM_DEFER": defer})
    assert rcs == [0, 0] and all(r is not None for r in res), errs
    assert not str(res[0]["error"]) and not str(res[1]["error"]), (res[0]["error"], res[1]["error"])
    assert np.array_equal(res[0]["params"], res[1]["params"])            # the ranks hold the same model, bit for bit
    full = synth.make_batch(**cfg)
    want, st = _single_process(cfg, [full] * 3)
    assert rel_err(res[0]["params"], want) < 1e-5
    # the merged statistics (what comm_touch_done sums over the done-files, communicator.h:121-170)
    assert abs(float(res[0]["obj_sum"]) - st["obj_sum"]) < 1e-5 * abs(st["obj_sum"])
    assert int(res[0]["ref"]) == st["ref_tokens"] and abs(int(res[0]["err"]) - st["err_tokens"]) <= 1 and int(res[0]["frames"]) == st["frames"]


def test_two_ranks_each_holding_persistent_grids_on_one_gpu(gpu, tmp_path):
    """The product's N > 1 configuration -- PERSISTENT recurrence kernels in every rank + the communicator's per-layer buckets under the
    backward pass -- with two ranks, before an 8-GPU box runs it first.  Shape chosen so that both processes'
    grids are co-resident on the one GPU: 2 x BiLSTM of 256 cells, S = 32 per rank (forward 32 x 2 x 2 = 128, backward 8 x 2 x 8 = 128
    workgroups per process, 256 CUs), every process sizing its grids against HALF the device (EESEN_GPU_SHARE=2: fits() and the tile
    choices divide the CU count instead of relying on the spin time-outs).  Asserted: both ranks really ran every layer pass on the
    persistent kernels; 2 x 32 == ONE process x 64 over 3 steps with momentum 0.9 and <MaxGrad> -- against this library in one process
    AND against the reference's own Net (oracle/_ref, `--num-sequence 64`: SURVEY.md section 8e's parity statement) --; the ranks hold
    the same model bit for bit, also after a soak of further steps with zero recoveries; what the exchange cost goes on record."""
    from oracle import refbind
    from eesen_amd import nnet_io
    nsoak = int(os.environ.get("EESEN_SOAK_STEPS", "200"))
    over = dict(S=64, T=80, H=256, layers=2)
    # first the same run with the buckets DEFERRED behind the backward pass (EESEN_COMM_DEFER=1), no soak: same model bit for bit
    (tmp_path / "defer").mkdir()
    res_d, rcs_d, errs_d = launch("persist", 2, tmp_path / "defer", dict(cfg="cfg2", steps=3, soak=0, **over),
                                  env_extra={"EESEN_PERSISTENT": "1", "EESEN_GPU_SHARE": "2", "EESEN_COMM_DEFER": "1"}, timeout=300)
    assert rcs_d == [0, 0] and all(r is not None and not str(r["error"]) for r in res_d), [e[-2000:] for e in errs_d]
    cfg = synth.config("cfg2"); cfg.update(over)
    res, rcs, errs = launch("persist", 2, tmp_path, dict(cfg="cfg2", steps=3, soak=nsoak, **over),
                            env_extra={"EESEN_PERSISTENT": "1", "EESEN_GPU_SHARE": "2"}, timeout=600)
    assert rcs == [0, 0] and all(r is not None for r in res), [e[-2000:] for e in errs]
    for r in res:
        assert not str(r["error"]), r["error"]
        assert list(r["recurrence_steps"]) == [2, 2, 2] and list(r["recurrence"]) == [2, 2, 2], (r["recurrence_steps"], r["recurrence"])   # {LSTM layers, fwd persistent, bwd persistent}
        assert int(r["recoveries"]) == 0 and int(r["dropped"]) == 0
    assert np.array_equal(res[0]["params_steps"], res[1]["params_steps"]) and np.array_equal(res[0]["params"], res[1]["params"])
    assert np.array_equal(res_d[0]["params_steps"], res[0]["params_steps"]) and np.array_equal(res_d[1]["params_steps"], res[0]["params_steps"])
    layers = synth.make_model(max_grad=0.05, **cfg)
    full = synth.make_batch(**cfg)
    want, _ = _single_process(cfg, [full] * 3, persistent="1")     # one process, S = 64, the whole device
    from eesen_amd.api import Net
    p0 = Net.from_layers(layers).GetParams()
    got = res[0]["params_steps"]
    rep = dict(config="2 x BiLSTM(256) + affine + softmax + CTC, S = 32 per rank x 2 ranks on ONE GPU, T = 80, EESEN_GPU_SHARE=2, persistent kernels, stand-in collective",
               params_vs_one_process=rel_err(got, want), update_vs_one_process=rel_err(got - p0, want - p0))
    assert rep["params_vs_one_process"] < 1e-5 and rep["update_vs_one_process"] < 5e-3, rep
    if refbind.available():   # the reference itself as ONE process with --num-sequence 64
        path = str(tmp_path / "m.nnet")
        nnet_io.write_nnet(path, layers, binary=True)
        ref = refbind.RefNet(path)
        ref.set_train_options(1e-3, 0.9)
        for _ in range(3):
            ref.set_seq_lengths(full.lens)
            o = ref.propagate(full.feats)
            c = refbind.cuda_ctc_eval_parallel(o, full.T, full.S, full.lens, full.label_ids, full.label_off)
            ref.backpropagate(c["diff"], False)
        rp = ref.get_params()
        rep.update(params_vs_reference=rel_err(got, rp), update_vs_reference=rel_err(got - p0, rp - p0))
        assert rep["params_vs_reference"] < 1e-5 and rep["update_vs_reference"] < 5e-3, rep
    if nsoak:
        rep.update(soak_steps=nsoak, ms_per_step=[float(r["soak_ms_per_step"]) for r in res],
                   recurrence_fwd_ms=[float(r["soak_ms_recurrence_fwd"]) for r in res], recurrence_bwd_ms=[float(r["soak_ms_recurrence_bwd"]) for r in res],
                   allreduce_ms=[float(r["soak_ms_allreduce"]) for r in res], allreduce_exposed_ms=[float(r["soak_ms_allreduce_exposed"]) for r in res],
                   recoveries=[int(r["recoveries"]) for r in res])
    out_dir = os.environ.get("EESEN_PARITY_OUT", os.path.join(ROOT, "gpurun_out"))
    os.makedirs(out_dir, exist_ok=True)
    json.dump(rep, open(os.path.join(out_dir, "multirank_persistent.json"), "w"), indent=1)


def test_three_ranks_and_odd_shard_sizes(gpu, tmp_path):
    cfg = synth.config("small_bi"); cfg.update(S=10, T=40)              # 10 utterances over 3 ranks: shards of 4, 3, 3
    res, rcs, errs = launch("parity", 3, tmp_path, dict(cfg="small_bi", S=10, T=40, steps=2))
    assert rcs == [0, 0, 0] and all(r is not None and not str(r["error"]) for r in res), errs
    want, _ = _single_process(cfg, [synth.make_batch(**cfg)] * 2)
    for r in res:
        assert np.array_equal(r["params"], res[0]["params"])
    assert rel_err(res[0]["params"], want) < 1e-5


@pytest.mark.parametrize("defer", ["0", "1"])
def test_uneven_shards_zero_gradient_protocol(gpu, tmp_path, defer):
    cfg = synth.config("small_bi"); cfg.update(S=6, T=40)
    steps, world = 4, 2
    res, rcs, errs = launch("uneven", world, tmp_path, dict(cfg="small_bi", S=6, T=40, steps=steps, fewer=2), env_extra={"EESEN_COMM_DEFER": defer})
    assert rcs == [0, 0] and all(r is not None for r in res), errs
    # rank 0 trained 4 minibatches, rank 1 only 2 and then followed with zero gradients for exactly the 2 steps it lacked
    assert [int(r["real_steps"]) for r in res] == [4, 2] and [int(r["zero_steps"]) for r in res] == [0, 2]
    assert np.array_equal(res[0]["params"], res[1]["params"])
    # one process on the union: step k = the minibatches the ranks held at step k; the closing all-zero round is a no-op
    batches = []
    for k in range(steps):
        held = [synth.make_batch(**{**cfg, "seed": 1000 + 10 * k + r}) for r in range(world) if k < steps - 2 * r]
        batches.append(_merge(held))
    want, st = _single_process(cfg, batches)
    assert rel_err(res[0]["params"], want) < 1e-5
    assert int(res[0]["ref"]) == st["ref_tokens"] and int(res[0]["frames"]) == st["frames"]


def test_a_rank_that_dies_before_the_first_collective_does_not_hang_the_other(gpu, tmp_path):
    t0 = time.time()
    res, rcs, errs = launch("die", 2, tmp_path, dict(cfg="small_bi", victim=1, steps=2), env_extra={"EESEN_COMM_TIMEOUT_S": "4"}, timeout=120)
    took = time.time() - t0
    assert rcs[1] == 0 and res[1] is None                                 # the victim left without a word
    assert res[0] is not None, errs[0][-3000:]
    assert "communicator aborted" in str(res[0]["error"]) and int(res[0]["code"]) == -5, res[0]["error"]   # EESEN_ERR_COMM
    assert took < 90, f"the survivor needed {took:.0f} s to give up"


def test_standin_allreduce_under_every_persistent_backward_recurrence(gpu, tmp_path):
    """(iv): the co-residency hazard.  The persistent recurrence grids need every workgroup resident (one per CU at cfg2); with a
    communicator attached, the stand-in's all-reduce kernels -- 32 workgroups x 512 threads per 8 MB chunk, four chunks per
    25 MB bucket, each moving its payload to host memory and back -- run on the high-priority communication stream exactly
    when the next-lower layer's backward recurrence is on the chip.  200 steps: no spin time-out (a time-out under a
    communicator is fatal by design), the model stays bit-identical to the run without the exchange (one rank: the sum is the
    identity), and the per-step cost of the overlap goes on record."""
    steps = int(os.environ.get("EESEN_SOAK_STEPS", "200"))
    res, rcs, errs = launch("soak", 1, tmp_path, dict(cfg="cfg2", steps=steps), env_extra={"EESEN_PERSISTENT": "1"}, timeout=600)
    assert rcs == [0] and res[0] is not None, errs[0][-3000:]
    r = res[0]
    assert bool(r["standin"]), "EESEN_RCCL_LIBRARY was not honoured"
    ms0, ms1 = float(r["ms0"]), float(r["ms1"])
    # {lstm layers, forward persistent, backward persistent}: every layer pass ran as ONE launch, with and without the exchange
    assert list(r["info0"]) == [4, 4, 4] and list(r["info1"]) == [4, 4, 4], (r["info0"], r["info1"])
    assert int(r["rec0"]) == 0 and int(r["rec1"]) == 0 and int(r["dropped"]) == 0, "a persistent recurrence kernel timed out beside the all-reduce kernels"
    assert bool(r["identical"])       # one rank: the sum is the identity, so the model must not move by a bit
    rep = dict(config="cfg2", steps=steps, ms_per_step_without_exchange=ms0, ms_per_step_with_standin_allreduce=ms1,
               overlap_cost_ms=ms1 - ms0, buckets_mb=[0.19, 25.2, 25.2, 25.2, 9.1],
               standin="tests/native/fake_rccl.hip: 32 workgroups x 512 threads per 8 MB chunk, payload through host memory (PCIe) and back")
    out_dir = os.environ.get("EESEN_PARITY_OUT", os.path.join(ROOT, "gpurun_out"))
    os.makedirs(out_dir, exist_ok=True)
    json.dump(rep, open(os.path.join(out_dir, "multirank_overlap.json"), "w"), indent=1)
    assert ms1 < 2.0 * ms0, rep
