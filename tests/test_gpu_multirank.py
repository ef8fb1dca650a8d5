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


def _with_params(layers, flat):
    """The same topology carrying the parameters of a flat Net::GetParams-ordered vector."""
    out, i = [], 0
    for L in layers:
        L = dict(L)
        ps = []
        for p in L["params"]:
            ps.append(np.asarray(flat[i:i + p.size], np.float32).reshape(p.shape)); i += p.size
        L["params"] = ps
        out.append(L)
    assert i == len(flat)
    return out


def _gradient_arbiters(layers, full, params_before, persistent):
    """The gradient of the WHOLE minibatch at the given parameters, two ways: this library in ONE process, and the reference's own
    Net (oracle/_ref; an lr = 1, momentum 0, <MaxGrad> 0 step's parameter delta IS the gradient, bilstm-layer.h:846-883).
    Returns (one_process, reference | None), flat in Net::GetParams order."""
    from eesen_amd.api import Net, Ctc
    from eesen_amd import nnet_io
    from oracle import refbind
    import tempfile
    lay = _with_params(layers, params_before)
    for L in lay:
        L["max_grad"] = 0.0
    old = os.environ.get("EESEN_PERSISTENT")
    os.environ["EESEN_PERSISTENT"] = persistent
    try:
        net = Net.from_layers(lay)
    finally:
        if old is None:
            del os.environ["EESEN_PERSISTENT"]
        else:
            os.environ["EESEN_PERSISTENT"] = old
    net.SetTrainOptions(1.0, 0.0)
    ctc = Ctc()
    net.SetSeqLengths(full.lens)
    o = net.Propagate(full.feats)
    d = ctc.EvalParallel(full.lens, o, full.labels, want_pzx=False)
    net.BackpropagateNoUpdate(d)
    one = net.GetGrads()
    ref = None
    if refbind.available():
        path = tempfile.mktemp(suffix=".nnet")
        nnet_io.write_nnet(path, lay, binary=True)
        try:
            rn = refbind.RefNet(path)
        finally:
            os.unlink(path)
        before = rn.get_params().astype(np.float64)
        rn.set_train_options(1.0, 0.0)
        rn.set_seq_lengths(full.lens)
        ro = rn.propagate(full.feats)
        c = refbind.cuda_ctc_eval_parallel(ro, full.T, full.S, full.lens, full.label_ids, full.label_off)
        rn.backpropagate(c["diff"], False)
        ref = (before - rn.get_params().astype(np.float64)).astype(np.float32)
    return one, ref


def _per_tensor(layers, a, b):
    from tests.util import split_params
    return {f"L{li}.{nm}": rel_err(x, y) for (li, nm, x), (_, _, y) in zip(split_params(layers, a), split_params(layers, b))}


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
    res, rcs, errs = launch("parity", 2, tmp_path, dict(cfg="small_bi", S=32, T=60, steps=3, grads=1), env_extra={"EESEN_COMM_DEFER": defer})
    assert rcs == [0, 0] and all(r is not None for r in res), errs
    assert not str(res[0]["error"]) and not str(res[1]["error"]), (res[0]["error"], res[1]["error"])
    assert np.array_equal(res[0]["params"], res[1]["params"])            # the ranks hold the same model, bit for bit
    assert np.array_equal(res[0]["grads_steps"], res[1]["grads_steps"])  # ... because they applied the same SUM, bit for bit
    full = synth.make_batch(**cfg)
    # What the exchange delivers, per step: the all-reduced fresh gradient (read between Backpropagate and Update) against the gradient of
    # the WHOLE minibatch at the very parameters the ranks held -- this library in one process, and the reference's own Net -- per
    # tensor at north_star's 1e-4 (this, not a clipped three-step update, is the bar of the data-parallel step)
    layers0 = synth.make_model(max_grad=0.05, **cfg)
    from eesen_amd.api import Net as _Net
    before = [_Net.from_layers(layers0).GetParams()] + list(res[0]["params_each_step"][:-1])
    for k in range(3):
        one, ref = _gradient_arbiters(layers0, full, before[k], "0")
        e1 = _per_tensor(layers0, res[0]["grads_steps"][k], one)
        assert max(e1.values()) < 1e-4, (k, e1)
        if ref is not None:
            e2 = _per_tensor(layers0, res[0]["grads_steps"][k], ref)
            assert max(e2.values()) < 1e-4, (k, e2)
    want, st = _single_process(cfg, [full] * 3)
    assert rel_err(res[0]["params"], want) < 1e-5
    # the merged statistics (what comm_touch_done sums over the done-files, communicator.h:121-170)
    assert abs(float(res[0]["obj_sum"]) - st["obj_sum"]) < 1e-5 * abs(st["obj_sum"])
    assert int(res[0]["ref"]) == st["ref_tokens"] and abs(int(res[0]["err"]) - st["err_tokens"]) <= 1 and int(res[0]["frames"]) == st["frames"]


RCCL_SHAPED = {"FAKE_RCCL_SHAPE": "rccl", "FAKE_RCCL_BLOCKS": "16"}   # two ranks x 16 = the 32 workgroups of ONE RCCL kernel on the device


def test_two_ranks_each_holding_persistent_grids_on_one_gpu(gpu, tmp_path):
    """The product's N > 1 configuration -- PERSISTENT recurrence kernels in every rank + the communicator's per-layer buckets --
    with two ranks, before an 8-GPU box runs it first.  Shape chosen so that both processes' grids are co-resident on the one GPU:
    2 x BiLSTM of 256 cells, S = 32 per rank (forward 32 x 2 x 2 = 128, backward 8 x 2 x 8 = 128 workgroups per process, 256 CUs),
    every process sizing its grids against HALF the device (EESEN_GPU_SHARE=2).  Round 6: the stand-in's all-reduce kernels have
    RCCL's FOOTPRINT (FAKE_RCCL_SHAPE=rccl: 512 threads x 256 VGPRs, 37.7 KB LDS -- the whole register file of the CU they land on),
    so what is resident beside what is what it will be under RCCL.  Runs: both schedules (overlapped = what the plan rule picks at
    256 cells, deferred), and the overlapped one again with rank 1 LATE by 3 ms every step (the straggler case of DESIGN.md section 7).
    Asserted: every layer pass on the persistent kernels, zero recoveries in all four runs; ranks bit-identical to each other and
    across schedules; per step the ALL-REDUCED gradient == the gradient of the 64 utterances at the same parameters, per tensor at
    1e-4, against this library in one process AND the reference's own Net (`--num-sequence 64`: SURVEY.md section 8e's parity
    statement).  What the exchange cost in each run goes on record (profiles/r06_rccl_shaped_soak.json)."""
    nsoak = int(os.environ.get("EESEN_SOAK_STEPS", "200"))
    over = dict(S=64, T=80, H=256, layers=2)
    cfg = synth.config("cfg2"); cfg.update(over)
    base = {"EESEN_PERSISTENT": "1", "EESEN_GPU_SHARE": "2", **RCCL_SHAPED}
    runs = {}
    for name, extra, opts in (("overlapped", {"EESEN_COMM_DEFER": "0"}, {}), ("deferred", {"EESEN_COMM_DEFER": "1"}, {}),
                              ("auto", {}, {}),
                              ("overlapped_straggler_3ms", {"EESEN_COMM_DEFER": "0"}, dict(straggle_ms=3, straggle_rank=1)),
                              ("deferred_straggler_3ms", {"EESEN_COMM_DEFER": "1"}, dict(straggle_ms=3, straggle_rank=1))):
        (tmp_path / name).mkdir()
        res, rcs, errs = launch("persist", 2, tmp_path / name, dict(cfg="cfg2", steps=3, soak=nsoak if name != "auto" else 0, grads=1, **over, **opts),
                                env_extra={**base, **extra}, timeout=600)
        assert rcs == [0, 0] and all(r is not None for r in res), (name, [e[-2000:] for e in errs])
        for r in res:
            assert not str(r["error"]), (name, r["error"])
            assert list(r["recurrence_steps"]) == [2, 2, 2] and list(r["recurrence"]) == [2, 2, 2], (name, r["recurrence_steps"], r["recurrence"])   # {LSTM layers, fwd persistent, bwd persistent}
            assert int(r["recoveries"]) == 0 and int(r["dropped"]) == 0, name
        assert np.array_equal(res[0]["params_steps"], res[1]["params_steps"]) and np.array_equal(res[0]["params"], res[1]["params"]), name
        assert np.array_equal(res[0]["grads_steps"], res[1]["grads_steps"]), name
        runs[name] = res
    for name in runs:   # the schedule and a late peer change WHEN the buckets travel, never what they sum to
        assert np.array_equal(runs[name][0]["params_steps"], runs["overlapped"][0]["params_steps"]), name
    plan = json.loads(str(runs["auto"][0]["plan"]))
    assert plan["exchange"].startswith("overlapped") and plan["gpu_share"] == 2, plan    # 256 cells: q4<4,4>, 122 registers -> 256 free per SIMD lane
    assert plan["layers"][0]["backward"]["kernel"] == "lstm_bwd_persistent_q4_kernel<4,4>" and plan["layers"][0]["backward"]["free_vgprs_per_simd_lane"] == 256, plan

    res = runs["overlapped"]
    layers = synth.make_model(max_grad=0.05, **cfg)
    full = synth.make_batch(**cfg)
    from eesen_amd.api import Net
    p0 = Net.from_layers(layers).GetParams()
    before = [p0] + list(res[0]["params_each_step"][:-1])
    rep = dict(config="2 x BiLSTM(256) + affine + softmax + CTC, S = 32 per rank x 2 ranks on ONE GPU, T = 80, EESEN_GPU_SHARE=2, persistent kernels, "
                      "stand-in collective in RCCL's footprint (512 threads x 256 VGPRs, 37.7 KB LDS, 2 x 16 workgroups)", steps=[])
    for k in range(3):
        one, ref = _gradient_arbiters(layers, full, before[k], "1")
        e1 = _per_tensor(layers, res[0]["grads_steps"][k], one)
        st = dict(step=k + 1, summed_gradient_vs_one_process_worst_tensor=max(e1.values()))
        assert max(e1.values()) < 1e-4, (k, e1)
        if ref is not None:
            e2 = _per_tensor(layers, res[0]["grads_steps"][k], ref)
            st["summed_gradient_vs_reference_worst_tensor"] = max(e2.values())
            assert max(e2.values()) < 1e-4, (k, e2)
        rep["steps"].append(st)
    want, _ = _single_process(cfg, [full] * 3, persistent="1")     # one process, S = 64, the whole device: the parameters after three steps
    rep["params_vs_one_process"] = rel_err(res[0]["params_steps"], want)
    assert rep["params_vs_one_process"] < 1e-5, rep
    if nsoak:
        rep["soak_steps"] = nsoak
        for name, rr in runs.items():
            if name == "auto":
                continue
            rep[name] = dict(ms_per_step=[float(r["soak_ms_per_step"]) for r in rr],
                             recurrence_bwd_ms=[float(r["soak_ms_recurrence_bwd"]) for r in rr], recurrence_bwd_max_ms=[float(r["soak_max_ms_recurrence_bwd"]) for r in rr],
                             recurrence_fwd_max_ms=[float(r["soak_max_ms_recurrence_fwd"]) for r in rr],
                             allreduce_ms=[float(r["soak_ms_allreduce"]) for r in rr], allreduce_max_ms=[float(r["soak_max_ms_allreduce"]) for r in rr],
                             allreduce_exposed_ms=[float(r["soak_ms_allreduce_exposed"]) for r in rr], recoveries=[int(r["recoveries"]) for r in rr])
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


@pytest.mark.parametrize("shape", ["plain", "rccl"])
def test_standin_allreduce_under_every_persistent_backward_recurrence(gpu, tmp_path, shape):
    """(iv): the co-residency hazard at the HEADLINE shape.  cfg2's persistent grids hold one workgroup on every CU; with a communicator
    attached the stand-in's all-reduce kernels (32 workgroups x 512 threads per 8 MB chunk, four chunks per 25 MB bucket, payload to
    host memory and back) run on the high-priority communication stream.
      shape = "plain": the ~30-register kernel of rounds 3-5, which fits beside every tile -- the overlapped schedule forced
                       (EESEN_COMM_DEFER=0), as those rounds ran it;
      shape = "rccl":  RCCL's footprint (256 VGPRs x 512 threads, 37.7 KB LDS: fits beside NO backward tile of cfg2) -- under BOTH
                       schedules: overlapped (the all-reduce workgroups become resident when a recurrence retires and hold their CUs
                       against the next one) and deferred, which is what the plan rule picks here (q4<8,4>: 176 registers free < 256).
    200 steps each: no spin time-out (a time-out under a communicator is fatal by design), the model bit-identical to the run
    without the exchange (one rank: the sum is the identity), and the per-step cost goes on record (profiles/r06_rccl_shaped_soak.json)."""
    steps = int(os.environ.get("EESEN_SOAK_STEPS", "200"))
    arms = [("overlapped", "0")] if shape == "plain" else [("overlapped", "0"), ("deferred", "1"), ("auto", None)]
    rep = dict(config="cfg2", steps=steps, shape=shape, buckets_mb=[0.19, 25.2, 25.2, 25.2, 9.1],
               standin="tests/native/fake_rccl.hip: 32 workgroups x 512 threads per 8 MB chunk, payload through host memory (PCIe) and back"
                       + ("; RCCL's footprint: 256 VGPRs per lane, 37 664 B LDS" if shape == "rccl" else "; ~30 VGPRs"))
    for name, defer in arms:
        (tmp_path / name).mkdir()
        env = {"EESEN_PERSISTENT": "1"}
        if defer is not None:
            env["EESEN_COMM_DEFER"] = defer
        if shape == "rccl":
            env["FAKE_RCCL_SHAPE"] = "rccl"
        res, rcs, errs = launch("soak", 1, tmp_path / name, dict(cfg="cfg2", steps=steps if name != "auto" else 20), env_extra=env, timeout=600)
        assert rcs == [0] and res[0] is not None, (name, errs[0][-3000:])
        r = res[0]
        assert bool(r["standin"]), "EESEN_RCCL_LIBRARY was not honoured"
        ms0, ms1 = float(r["ms0"]), float(r["ms1"])
        ex = json.loads(str(r["exchange"]))
        # {lstm layers, forward persistent, backward persistent}: every layer pass ran as ONE launch, with and without the exchange
        assert list(r["info0"]) == [4, 4, 4] and list(r["info1"]) == [4, 4, 4], (name, r["info0"], r["info1"])
        assert int(r["rec0"]) == 0 and int(r["rec1"]) == 0 and int(r["dropped"]) == 0, f"{name}: a persistent recurrence kernel timed out beside the all-reduce kernels"
        assert bool(r["identical"]), name       # one rank: the sum is the identity, so the model must not move by a bit
        assert ex["schedule"].startswith("deferred" if name in ("deferred", "auto") else "overlapped"), (name, ex)
        rep[name] = dict(ms_per_step_without_exchange=ms0, ms_per_step_with_standin_allreduce=ms1, cost_ms=ms1 - ms0, **ex)
        assert ms1 < 2.0 * ms0, rep
    out_dir = os.environ.get("EESEN_PARITY_OUT", os.path.join(ROOT, "gpurun_out"))
    os.makedirs(out_dir, exist_ok=True)
    json.dump(rep, open(os.path.join(out_dir, f"multirank_overlap_{shape}.json"), "w"), indent=1)
