#!/usr/bin/env python3
"""bench.py -- CTC training frames/sec of the MI355X path on BASELINE.json's configuration.

  python bench.py --gpus N --steps K --warmup W
  N > 1: either under a launcher that exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
  (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...), or on its own: without
  WORLD_SIZE in the environment `bench.py --gpus N` starts its N ranks itself (one process per GPU) and still prints
  exactly one JSON line.

A "step" is one pass of train-ctc-parallel's inner loop (/root/reference/src/netbin/train-ctc-parallel.cc:195-207)
over one synthetic utterance mini-batch per GPU, as SURVEY.md section 8(d) defines it -- H2D -> forward -> CTC -> backward ->
[all-reduce] -> update: the minibatch's S HOST matrices go through the device feeder (pinned pack, one PCIe copy, time-major
interleave on the device: what train-ctc-parallel.cc:186-198 does on the host every minibatch) -> SetSeqLengths -> Propagate ->
Ctc::EvalParallel -> Ctc::ErrorRateMSeq -> Backpropagate (+ per-layer RCCL gradient all-reduce when N > 1, issued by the library
under the backward pass) -> Update.  Workload at N = 1 = BASELINE.json configs[1]:
4 x BiLSTM (512 cells/direction), 40-d input, 46 classes, 32 utterances, T_max = 1000, fp32.
Weak scaling: every rank runs its own 32-utterance shard (global batch 32 N = configs[2] at N = 8).
`value` = padded frames/s of the whole job (the reference's own fps counts padded frames,
train-ctc-parallel.cc:215,247-252) with the per-step H2D INSIDE the timed step (round 5; the copy of step n + 1 travels under
step n's backward pass); `config.device_resident_frames_per_s` is the same loop with the features already resident in HBM.

Rank 0 prints ONE JSON line (see the task contract) with `roofline` (dominant kernel, live HIP-event timing)
and, at N = 1, `cpu_baseline` (the reference's own CPU code from oracle/_ref when present, else the C port).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from eesen_amd import synth  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: fp32-input MFMA dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0   # same guide: bf16 MFMA dense peak (the sparsity-inflated headline figure is twice that)
PEAK_HBM_GBS = 8000.0
PEAK_XGMI_GBS = 7 * 153.0          # same guide: 7 xGMI links x ~153 GB/s per GPU (point-to-point: a ring step is bound by ONE link)


def flops_per_frame(cfg) -> float:
    """sum over layers of 2 dirs * 24 H (D_in + H), + 6 D_last K (+ 6 * 2H * P per projection)."""
    nd = 2 if cfg["kind"].startswith("BiLstm") else 1
    H, D, K = cfg["H"], cfg["D"], cfg["K"]
    total, din = 0.0, D
    for li in range(cfg["layers"]):
        total += nd * 24.0 * H * (din + H)
        din = nd * H
        if cfg.get("proj") and li < cfg["layers"] - 1:
            total += 6.0 * din * cfg["proj"]
            din = cfg["proj"]
    total += 6.0 * din * K
    return total


def gemm_products() -> int:
    """Matrix-pipe products per fp32 product of the dense GEMMs in the library's current arithmetic mode (gemm.hip): 0 = the
    f32-input MFMA (EESEN_GEMM_MODE=f32), 6 = three bf16 planes per operand (split), 3 = two fp16 planes per operand (half: the
    default since round 6)."""
    import ctypes as C
    from eesen_amd import _lib
    m = C.c_int(-1)
    _lib.check(_lib.load().eesen_get_gemm_mode(C.byref(m)))
    return {0: 0, 1: 6, 2: 3}[m.value]


def fwd_rec_products(cfg, S: int, forward_mode: int = 0) -> int:
    """How the FORWARD recurrent product of this configuration is executed (lstm_persistent.hip: bf_plan): 0 = on the fp32-input
    MFMA; n > 0 = on the bf16 pipe as n bf16 products per fp32 product -- 6 for the fp32-class 3-way split the narrow tile takes
    (H <= 512, S > 16; EESEN_FWD_SPLIT), 2 (m_t one plane, W_m as hi + lo planes) for BASELINE config 4's bf16
    forward (--forward-precision bf16) on layers of 256 .. 1024 cells."""
    if os.environ.get("EESEN_PERSISTENT", "1") == "0" or cfg["H"] % 32 != 0:
        return 0
    H = cfg["H"]
    if forward_mode == 1 and H % 256 == 0 and H <= 1024:
        return 2
    if os.environ.get("EESEN_FWD_SPLIT", "1") == "0":
        return 0
    f16 = os.environ.get("EESEN_FWD_F16", "1") != "0"   # round 6: two fp16 planes per operand, three products -- narrow AND wide tiles
    narrow = H % 8 == 0 and S > 16 and (H // 32 + 7) // 8 <= 2
    if narrow:
        return 3 if f16 else 6
    if f16 and H % 256 == 0 and H <= 1024:
        return 3
    return 0


def bwd_rec_products(cfg) -> int:
    """How the BACKWARD recurrent product runs: 0 = on the fp32-input MFMA (narrow layers: the 4 x 32 tile; any layer with
    EESEN_BWD_F16=0 / EESEN_FWD_SPLIT=0); 3 = the K-split tile of wide layers (768 < H <= 1024, H % 256 == 0) on two fp16 planes per
    operand (lstm_bwd_persistent_ksplit_h_kernel, round 6)."""
    if os.environ.get("EESEN_PERSISTENT", "1") == "0" or os.environ.get("EESEN_FWD_SPLIT", "1") == "0" or os.environ.get("EESEN_BWD_F16", "1") == "0":
        return 0
    H = cfg["H"]
    return 3 if H == 1024 and os.environ.get("EESEN_BWD_KSPLIT", "1") != "0" else 0


def pipe_bound(cfg, gemm_prod: int = 3, fwd_products: int = 0, bwd_products: int = 0) -> dict:
    """The step's flops by matrix pipe and the time the two pipes need for them at their peaks.  Per layer and direction the
    recurrence kernels execute the forward and the backward recurrent product (8 H^2 per frame each).  The backward one runs on
    the fp32 pipe (v_mfma_f32_16x16x4_f32 / 4x4x1); the forward one too, unless `fwd_products` says it runs on the bf16 pipe as that
    many 16-bit products per fp32 product (fwd_rec_products).  Everything else is GEMM: on the 16-bit pipe as `gemm_prod` products per
    fp32 product (3: two fp16 planes per operand, the default; 6: three bf16 planes), or on the fp32 pipe (0: EESEN_GEMM_MODE=f32).
    The bf16 and fp16 MFMA forms have the same dense peak.  frac = bound / measured time is <= 1 by construction."""
    nd = 2 if cfg["kind"].startswith("BiLstm") else 1
    rec = float(cfg["layers"]) * nd * 16.0 * cfg["H"] * cfg["H"]
    gemm = flops_per_frame(cfg) - rec
    rec_f32 = (0.0 if fwd_products else rec / 2) + (0.0 if bwd_products else rec / 2)
    rec_bf16 = (fwd_products + bwd_products) * rec / 2
    split_gemm = gemm_prod > 0
    if split_gemm:
        bf16 = float(gemm_prod) * gemm + rec_bf16
        sec = rec_f32 / (PEAK_F32_MFMA_TFLOPS * 1e12) + bf16 / (PEAK_BF16_MFMA_TFLOPS * 1e12)
    else:
        bf16 = rec_bf16
        sec = (rec_f32 + gemm) / (PEAK_F32_MFMA_TFLOPS * 1e12) + bf16 / (PEAK_BF16_MFMA_TFLOPS * 1e12)
    return {"f32_pipe_flops_per_frame": rec_f32 + (0.0 if split_gemm else gemm), "gemm_flops_per_frame_fp32_equivalent": gemm,
            "bf16_pipe_executed_flops_per_frame": bf16, "gemm_products": gemm_prod, "forward_recurrence_bf16_products": fwd_products, "backward_recurrence_f16_products": bwd_products,
            "bound_us_per_frame": 1e6 * sec}


def ctc_block(cfg, batch, ctc_ph: dict, K: int) -> dict:
    """CTC against the HBM roofline.  Whole CTC: 4 * (3K + 2L') algorithmic bytes per padded frame.  Per part as
    well: the lattice sweep is a 2T-step dependency chain of S independent lattices (not bandwidth-shaped); the bulk pass (gamma,
    softmax Jacobian: reads alpha_t, beta_t and y_t, writes diff_t) is -- quoted on its algorithmic bytes (the utterance's own L'_s
    positions) AND on the bytes it actually moves: the kernel reads whole padded lane slots (64 x positions-per-lane of the longest
    lattice), of real frames only."""
    T, S, Kc = batch.T, batch.S, cfg["K"]
    Lp = 2 * max(len(l) for l in batch.labels) + 1
    ctc_bytes = 4.0 * (3 * Kc + 2 * Lp) * T * S
    ctc_s = (ctc_ph["alpha_beta"] + ctc_ph["error_diff"] + ctc_ph["log"]) / K
    bulk_bytes = 4.0 * sum(int(batch.lens[s]) * (2 * (2 * len(batch.labels[s]) + 1) + 2 * Kc) for s in range(S))
    maxp = next(m for m in (2, 4, 6, 8, 12, 16, 24, 32, 48, 64) if 64 * m >= Lp)      # ctc.hip: ctc_error_diff's positions per lane
    lpad = 64
    while lpad < Lp:
        lpad *= 2
    moved = 4.0 * float(np.sum(batch.lens)) * (2 * 64 * maxp + 2 * Kc)
    bulk_s = ctc_ph["error_diff"] / K
    sweep_s = ctc_ph["alpha_beta"] / K
    sweep_moved = 4.0 * float(np.sum(batch.lens)) * (2 * lpad + 2 * Kc)   # alpha + beta rows written at the padded width; the frame's log-probability row read once per direction
    return {"bound": "hbm", "achieved": ctc_bytes / ctc_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": ctc_bytes / ctc_s / 1e9 / PEAK_HBM_GBS, "ms": 1e3 * ctc_s, "bytes": ctc_bytes, "bytes_per_frame": 4.0 * (3 * Kc + 2 * Lp),
            "lattice_positions": Lp, "padded_row": lpad,
            "sweep": {"ms": 1e3 * sweep_s, "us_per_lattice_step": 1e6 * sweep_s / T, "bytes_moved": sweep_moved, "moved_GBps": sweep_moved / sweep_s / 1e9,
                      "note": "bounded by the T-step dependency chain (alpha and beta sweeps of all S lattices run concurrently: 2S workgroups), not by HBM (SURVEY.md 8d caveat)"},
            "bulk": {"ms": 1e3 * bulk_s, "bytes": bulk_bytes, "achieved": bulk_bytes / bulk_s / 1e9, "peak": PEAK_HBM_GBS,
                     "unit": "GB/s", "frac": bulk_bytes / bulk_s / 1e9 / PEAK_HBM_GBS,
                     "bytes_moved": moved, "moved_GBps": moved / bulk_s / 1e9, "moved_frac": moved / bulk_s / 1e9 / PEAK_HBM_GBS,
                     "positions_per_lane": maxp}}


def cpu_baseline(cfg, seconds_budget: float = 25.0) -> dict:
    """The reference's CPU path (oracle/_ref: src/net + src/cpucompute compiled unmodified; its CUDA-only CTC
    kernels run through the CPU shim) or, when that library is absent, the C port; same model, a bounded
    sample of the same workload.  Reported, never the target."""
    from oracle import refbind, net as onet
    from eesen_amd import nnet_io
    ncores = os.cpu_count() or 1
    sc = dict(cfg)
    sc["T"] = 250  # 32 utterances x 250 frames = 8000 padded frames of the same 4x512 BiLSTM: ~20 s over the three thread settings
    layers = synth.make_model(max_grad=50.0, **sc)
    batch = synth.make_batch(**sc)
    frames = batch.T * batch.S
    out = {"unit": "frames/s", "sample": f"1 step of the same model on S={batch.S} utterances x T={batch.T} frames ({frames} padded frames)"}
    if refbind.available():
        path = tempfile.mktemp(suffix=".nnet")
        nnet_io.write_nnet(path, layers, binary=True)
        res = {}
        for thr in sorted({1, min(16, ncores), ncores}):
            refbind.set_blas_threads(thr)
            r = refbind.RefNet(path)
            r.set_train_options(4e-5, 0.9)
            r.set_seq_lengths(batch.lens)
            t0 = time.perf_counter()
            o = r.propagate(batch.feats)
            c = refbind.cuda_ctc_eval_parallel(o, batch.T, batch.S, batch.lens, batch.label_ids, batch.label_off)
            r.backpropagate(c["diff"], False)
            res[thr] = frames / (time.perf_counter() - t0)
        os.unlink(path)
        best = max(res, key=res.get)   # the reference's default build is single-threaded (src/configure:76); report its best
        out.update(kind="reference", value=res[best], cores=best, by_blas_threads={str(k): v for k, v in res.items()},
                   host_cores=ncores,
                   note="reference src/net+src/cpucompute (OpenBLAS sgemm); CTC = reference CUDA kernel bodies run on the CPU, "
                        "the reference has no CPU CTC")
    else:
        ora = onet.OracleNet(layers, "f32")
        ora.set_train_options(4e-5, 0.9)
        sc["T"] = 16
        batch = synth.make_batch(**sc)
        t0 = time.perf_counter()
        onet.train_step(ora, batch, "f32")
        out.update(kind="port", value=batch.T * batch.S / (time.perf_counter() - t0), cores=1,
                   sample=f"1 step, S={batch.S} x T={batch.T} (scalar C port, naive GEMM loops)")
    return out


def frontend_leg(dev: int, S: int = 32, T: int = 1000, D: int = 40, iters: int = 20) -> dict:
    """Not the headline: the feature front end in front of the path on one cfg2-shaped minibatch of RAW
    features -- the wsj recipe's `apply-cmvn --norm-vars=true ... | add-deltas` (train_ctc_parallel.sh:95-110), 40 -> 120 columns.
    `device`: raw matrices over PCIe, CMVN + deltas + interleave on the GPU (eesen_feeder_submit_raw); `host_filtered`: what the
    reference pipeline hands over, 120-column matrices, interleave only.  `cpu_baseline`: the reference's own filter processes
    (oracle/_ref/featbin, when present) on the same table."""
    import subprocess
    import tempfile
    from eesen_amd import _lib, frontend as fe, kaldi_io
    from eesen_amd.api import Feeder
    rng = np.random.default_rng(777)
    lens = np.sort(rng.integers(int(0.8 * T), T + 1, size=S)); lens[-1] = T
    utts = [(f"spk{s % 4}_utt{s:03d}", (rng.standard_normal((int(lens[s]), D)) * 2 + 0.5).astype(np.float32)) for s in range(S)]
    stats = {}
    for k, m in utts:
        st = stats.setdefault(k.split("_")[0], np.zeros((2, D + 1)))
        st[0, :D] += m.sum(0, dtype=np.float64); st[1, :D] += (m.astype(np.float64) ** 2).sum(0); st[0, D] += m.shape[0]
    cm = [fe.cmvn_norm(stats[k.split("_")[0]], True) for k, _ in utts]
    raw = [m for _, m in utts]
    lib = _lib.load()

    def timed(feeder, submit):
        for _ in range(3):
            slot = submit(); feeder.acquire(slot); feeder.release(slot)
        _lib.check(lib.eesen_device_synchronize(dev))
        t0 = time.perf_counter()
        for _ in range(iters):
            slot = submit(); feeder.acquire(slot); feeder.release(slot)
        _lib.check(lib.eesen_device_synchronize(dev))
        return (time.perf_counter() - t0) / iters

    f1 = Feeder(dev, slots=2)
    f1.set_pipeline([(fe.CMVN, 1, 0), (fe.DELTAS, 2, 2)])
    t_dev = timed(f1, lambda: f1.submit_raw(raw, cm))
    got = f1.acquire(f1.submit_raw(raw, cm)).numpy()
    filtered = [np.ascontiguousarray(got.reshape(T, S, 3 * D)[: lens[s], s, :]) for s in range(S)]
    f2 = Feeder(dev, slots=2)
    t_host = timed(f2, lambda: f2.submit(filtered))
    frames = float(lens.sum())
    # algorithmic bytes per real frame: CMVN reads and writes D floats, the deltas read D and write 3D, the interleave moves 3D
    alg = 4.0 * (2 * D + 4 * D + 6 * D)
    out = {"workload": f"S={S} utterances x ~{T} frames, {D} -> {3 * D} columns (apply-cmvn --norm-vars=true | add-deltas)",
           "device": {"ms_per_batch": 1e3 * t_dev, "frames_per_s": frames / t_dev, "pcie_bytes_per_batch": int(frames * D * 4),
                      "note": "pinned pack + H2D of the RAW features + 2 stage kernels + interleave, end to end per batch"},
           "host_filtered": {"ms_per_batch": 1e3 * t_host, "frames_per_s": frames / t_host, "pcie_bytes_per_batch": int(frames * 3 * D * 4)},
           "algorithmic_bytes_per_frame": alg}
    bindir = os.path.join(ROOT, "oracle", "_ref", "featbin")
    if os.path.isfile(os.path.join(bindir, "apply-cmvn")):
        with tempfile.TemporaryDirectory() as tmp:
            ark, scp = os.path.join(tmp, "raw.ark"), os.path.join(tmp, "raw.scp")
            kaldi_io.write_mat_ark(ark, utts * 8, scp_path=None)      # 8 x the batch: ~0.25 M frames
            with open(os.path.join(tmp, "utt2spk"), "w") as f:
                for k, _ in utts:
                    f.write(f"{k} {k.split('_')[0]}\n")
            import struct
            with open(os.path.join(tmp, "cmvn.ark"), "wb") as f:
                for spk, st in stats.items():
                    f.write(spk.encode() + b" \x00BDM \x04" + struct.pack("<i", 2) + b"\x04" + struct.pack("<i", D + 1) + st.astype("<f8").tobytes())
            env = dict(os.environ, PATH=bindir + os.pathsep + os.environ.get("PATH", ""))
            cmd = (f"apply-cmvn --norm-vars=true --utt2spk=ark:{tmp}/utt2spk ark:{tmp}/cmvn.ark ark:{ark} ark:- 2>/dev/null | "
                   f"add-deltas ark:- ark:/dev/null 2>/dev/null")
            t0 = time.perf_counter()
            rc = subprocess.run(cmd, shell=True, env=env).returncode
            el = time.perf_counter() - t0
            if rc == 0:
                out["cpu_baseline"] = {"value": 8 * frames / el, "unit": "frames/s", "cores": 2, "kind": "reference",
                                       "sample": f"{int(8 * frames)} frames through the reference's apply-cmvn | add-deltas processes (one core each)"}
    return out


def secondary_leg(name: str, dev: int, steps: int = 3, warmup: int = 1, forward_bf16=False, reps: int = 3, over=None) -> dict:
    """Not the headline: one of the other single-GPU BASELINE.json configurations (configs[3] = cfg4: 5x1024 BiLSTM + 512-d
    projections; configs[4] = cfg5: 6x1024, S = 64 per GPU, T = 3000; cfg2 at --num-sequence 64), the same loop body on features
    resident in HBM, `reps` repetitions of `steps` timed steps (ms_per_step = the median repetition; min / max beside it), so that
    the driver's record holds a driver-timed number for every configuration -- and the CTC against its HBM roofline at the
    configuration BASELINE.json says it binds at (configs[4])."""
    from eesen_amd.api import Net, Ctc, CuMatrix, Feeder
    cfg = synth.config(name)
    cfg.update(over or {})
    layers = synth.make_model(max_grad=50.0, **cfg)
    batch = synth.make_batch(**cfg)
    net = Net.from_layers(layers, device=dev)
    net.SetTrainOptions(4e-5, 0.9)
    net.SetForwardPrecision(int(forward_bf16))    # 1 / True: forward GEMMs + forward recurrence on bf16 operands; 2: the GEMMs only
    ctc = Ctc(device=dev)
    ctc.SetGuard(net)
    feats = CuMatrix.from_numpy(batch.feats, dev)
    diff = CuMatrix(batch.T * batch.S, cfg["K"], dev)
    # round 6: like the headline, the timed step INCLUDES the minibatch's H2D (S host matrices -> pinned slot -> one PCIe copy ->
    # interleave on the device, double-buffered under the previous step's backward pass); the device-resident loop is kept beside it
    f3 = batch.feats.reshape(batch.T, batch.S, cfg["D"])
    mats = [np.ascontiguousarray(f3[: batch.lens[s], s, :]) for s in range(batch.S)]
    feeder = Feeder(dev, slots=2)
    pending = [feeder.submit(mats)]

    def step():
        slot = pending.pop()
        net.SetSeqLengths(batch.lens)
        out = net.Propagate(feeder.acquire(slot))
        feeder.release(slot)
        ctc.EvalParallel(batch.lens, out, batch.labels, diff, want_pzx=False)
        ctc.ErrorRateMSeq(batch.lens, out, batch.labels, deferred=True)
        net.Backpropagate(diff)
        pending.append(feeder.submit(mats))

    def step_resident():
        net.SetSeqLengths(batch.lens)
        out = net.Propagate(feats)
        ctc.EvalParallel(batch.lens, out, batch.labels, diff, want_pzx=False)
        ctc.ErrorRateMSeq(batch.lens, out, batch.labels, deferred=True)
        net.Backpropagate(diff)
    for _ in range(warmup):
        step()
    net.Synchronize()
    plan = net.Plan()
    ctc.SetProfiling(True)
    net.SetProfiling(True, accumulate=True)
    dts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        net.Synchronize()
        dts.append((time.perf_counter() - t0) / steps)
    ctc_ph = ctc.PhaseTimes()
    phases = net.PhaseTimes()
    ctc.SetProfiling(False)
    net.SetProfiling(False)
    step_resident()
    net.Synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_resident()
    net.Synchronize()
    dt_res = (time.perf_counter() - t0) / steps
    dt = float(np.median(dts))
    info = net.RecurrenceInfo()
    fpf = flops_per_frame(cfg)
    frames = float(batch.T * batch.S)
    nd = 2 if cfg["kind"].startswith("BiLstm") else 1
    # the leg's dominant kernel -- the backward recurrence -- against its roofline, live (HIP-event spans of this run) + the committed
    # counters of the same instantiation where they exist (profiles/pmc_traffic.json "wide", scripts/collect_profiles_wide.sh)
    leg_roof = None
    try:
        nl, n_timed = cfg["layers"], reps * steps
        rec_flops = 2.0 * batch.S * 4 * cfg["H"] * cfg["H"] * nd * batch.T
        us = 1e6 * phases["recurrence_bwd"] / (nl * n_timed)
        bk = sorted({L["backward"]["kernel"] for L in plan["layers"]})
        leg_roof = {"bound": "mfma", "kernel": bk[0] if len(bk) == 1 else bk, "avg_launch_us": us, "flops_per_launch": rec_flops,
                    "achieved": rec_flops / (us * 1e-6) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": rec_flops / (us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                    "launches_per_layer_pass": plan["layers"][-1]["backward"].get("launches"), "traffic": None, "mfma_busy": None}
        pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        key = {("cfg4", False): "cfg4", ("cfg4", True): "cfg4_bf16_forward", ("cfg5", False): "cfg5"}.get((name, bool(forward_bf16))) if not over else None
        w = pt.get("wide", {}).get(key) if key else None
        if w:
            from eesen_amd.build import csrc_digest
            for kn, e in w["kernels"].items():
                if kn.replace(" ", "") == (bk[0] if len(bk) == 1 else ""):
                    leg_roof.update(traffic=e["bytes_per_launch"], algorithmic_bytes=e["algorithmic_bytes_per_launch"], traffic_over_algorithmic=e["traffic_over_algorithmic"],
                                    mfma_busy=e.get("mfma_busy"), traffic_source={"file": "profiles/pmc_traffic.json (wide)", "tables": pt.get("wide_source"),
                                                                                  "csrc_sha": pt.get("csrc_sha"), "stale": pt.get("csrc_sha") != csrc_digest()})
    except Exception:   # noqa: BLE001 -- a leg's roofline block never takes the leg down
        pass
    return {"roofline": leg_roof, "workload": f"{name}: {cfg['layers']}x{cfg['H']} {'Bi' if nd == 2 else ''}LSTM{' + ' + str(cfg['proj']) + '-d projections' if cfg.get('proj') else ''}, "
                        f"K={cfg['K']}, S={batch.S} utterances/GPU, T_max={batch.T}",
            "steps": steps, "warmup": warmup, "repetitions": reps, "ms_per_step": 1e3 * dt, "ms_per_step_min_median_max": [1e3 * min(dts), 1e3 * dt, 1e3 * max(dts)],
            "frames_per_s": frames / dt, "input": "host matrices through the feeder, H2D inside the timed step (like the headline)",
            "device_resident_ms_per_step": 1e3 * dt_res, "device_resident_frames_per_s": frames / dt_res,
            "kernels": plan_summary(plan),
            "phase_ms_per_step": {k: 1e3 * v / (reps * steps) for k, v in phases.items()},
            "ctc": ctc_block(cfg, batch, ctc_ph, reps * steps),
            "dtype": ("bf16-fwd/f32" if int(forward_bf16) == 1 else "bf16-fwd-gemm-only/f32") if forward_bf16 else "f32",
            "bf16_recurrence_layers": net.Bf16RecurrenceLayers(),
            "whole_step_tflops_fp32_equivalent": fpf * frames / dt / 1e12,
            "whole_step_frac_of_pipe_roofline": pipe_bound(cfg, gemm_products(), fwd_rec_products(cfg, batch.S, int(forward_bf16)), bwd_rec_products(cfg))["bound_us_per_frame"] * 1e-6 * frames / dt,   # both pipes at peak, <= 1 (pipe_bound)
            "flops_per_frame": fpf, "persistent_layers": {"fwd": info["fwd_persistent"], "bwd": info["bwd_persistent"], "of": info["lstm_layers"]},
            "recoveries": net.recoveries, "ctc_minibatches_dropped": ctc.Dropped()}


def plan_summary(plan: dict) -> dict:
    """eesen_net_plan_string boiled down to what a bench leg should say about itself: the recurrence instantiations the library's
    one selection function chose for this shape (the launchers execute exactly these), and the schedule decisions."""
    fw = sorted({L["forward"]["kernel"] for L in plan["layers"]})
    bw = sorted({L["backward"]["kernel"] for L in plan["layers"]})
    one = plan["layers"][-1] if plan["layers"] else None
    out = {"forward": fw, "backward": bw, "weight_gradient_gemms": plan["weight_gradient_gemms"], "exchange": plan["exchange"]}
    if one and one["backward"].get("persistent"):
        b, f = one["backward"], one["forward"]
        out["backward_tile"] = {k: b[k] for k in ("sequences_per_workgroup", "units_per_workgroup", "workgroups", "launches", "vgprs", "lds_bytes", "free_vgprs_per_simd_lane")}
        if f.get("persistent"):
            out["forward_tile"] = {k: f[k] for k in ("sequences_per_workgroup", "units_per_workgroup", "workgroups", "workgroups_per_cu", "launches", "vgprs", "lds_bytes")}
    return out


def recipe_leg(dev: int, num_sequence: int, n_utts: int = 120, frame_limit: int = 25000, reps: int = 3) -> dict:
    """Not the headline: the reference's OWN recipe shape -- 4 x 320 BiLSTM on 120-d features (40 fbanks + deltas), ~45 phone
    targets, --num-sequence 10 (20 as the second point) --frame-num-limit 25000, utterances of a length-sorted list with WSJ-like
    durations (asr_egs/wsj/run_ctc_phn.sh:65-85, steps/train_ctc_parallel.sh:13-21) -- through the trainer's own path: greedy
    grouping, device feeder (every minibatch has its own T and S), Propagate / CTC / Backpropagate.  (H = 320 is not a multiple of
    128: up to round 4 the backward pass fell to the 8-sequence tile here; since round 5 it takes the 4 x 32 tile with K = 4H not
    filling the waves' chunk pairs -- lstm_bwd_persistent_q4_kernel<6, .> -- wherever S is a multiple of 4.)"""
    from eesen_amd.api import Net, Ctc, CuMatrix, Feeder
    from eesen_amd.batching import assemble
    cfg = dict(kind="BiLstmParallel", layers=4, H=320, D=120, K=46)
    rng = np.random.default_rng(777)
    lens = np.sort(np.clip(rng.gamma(6.0, 130.0, size=n_utts), 150, 1600).astype(int))        # ~7.8 s mean, sorted as the recipes do
    utts = [(f"utt{i:04d}", rng.standard_normal((int(n), cfg["D"])).astype(np.float32)) for i, n in enumerate(lens)]
    labs = {k: rng.integers(1, cfg["K"], size=max(1, m.shape[0] // 10)).astype(np.int32) for k, m in utts}
    groups = list(assemble(iter(utts), labs, num_sequence, frame_limit, cfg["D"], interleaved=False))
    net = Net.from_layers(synth.make_model(max_grad=50.0, **cfg), device=dev)
    net.SetTrainOptions(4e-5, 0.9)
    ctc = Ctc(device=dev)
    ctc.SetGuard(net)
    feeder = Feeder(dev, slots=2)
    diffs = {}

    def epoch():
        pers = [0, 0, 0]
        slot = feeder.submit(groups[0].mats)
        for i, mb in enumerate(groups):
            net.SetSeqLengths(mb.lens)
            out = net.Propagate(feeder.acquire(slot))
            feeder.release(slot)
            if out.rows not in diffs:
                diffs[out.rows] = CuMatrix(out.rows, cfg["K"], dev, zero=False)
            d = diffs[out.rows]
            ctc.EvalParallel(mb.lens, out, mb.labels, d, want_pzx=False)
            ctc.ErrorRateMSeq(mb.lens, out, mb.labels, deferred=True)
            net.Backpropagate(d)
            if i + 1 < len(groups):
                slot = feeder.submit(groups[i + 1].mats)
            info = net.RecurrenceInfo()
            pers[0] += info["fwd_persistent"]; pers[1] += info["bwd_persistent"]; pers[2] += info["lstm_layers"]
        net.Synchronize()
        return pers
    epoch()                                   # warm-up: allocations, every distinct shape once
    dts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        pers = epoch()
        dts.append(time.perf_counter() - t0)
    dt = float(np.median(dts))
    padded = float(sum(g.T * g.S for g in groups)); real = float(sum(int(g.lens.sum()) for g in groups))
    fpf = flops_per_frame(cfg)
    return {"workload": f"4x320 BiLSTM, D=120, K=46, --num-sequence {num_sequence} --frame-num-limit {frame_limit}, {n_utts} length-sorted utterances "
                        f"of {int(lens.min())}-{int(lens.max())} frames in {len(groups)} minibatches (S = {min(g.S for g in groups)}-{max(g.S for g in groups)})",
            "minibatches": len(groups), "repetitions": reps, "ms_per_minibatch": 1e3 * dt / len(groups),
            "ms_per_minibatch_min_median_max": [1e3 * min(dts) / len(groups), 1e3 * dt / len(groups), 1e3 * max(dts) / len(groups)],
            "padded_frames_per_s": padded / dt, "real_frames_per_s": real / dt,
            "whole_step_tflops_fp32_equivalent": fpf * padded / dt / 1e12, "flops_per_frame": fpf,
            "persistent_layer_passes": {"fwd": pers[0], "bwd": pers[1], "of": pers[2]}, "recoveries": net.recoveries}


def secondary_legs(dev: int) -> dict:
    """name -> callable: the other single-GPU BASELINE configurations and the reference's own recipe shape (config.secondary)."""
    # (10 / 5 timed steps: three were inside the box-to-box noise for comparing the cfg4 legs with each other)
    return {"cfg2_S64": lambda: secondary_leg("cfg2", dev, steps=10, warmup=2, over=dict(S=64)),
            "cfg4": lambda: secondary_leg("cfg4", dev, steps=10, warmup=2),
            "cfg4_bf16_forward": lambda: secondary_leg("cfg4", dev, steps=10, warmup=2, forward_bf16=True),
            "cfg5": lambda: secondary_leg("cfg5", dev, steps=4, warmup=1),
            "wsj_recipe_shape_S10": lambda: recipe_leg(dev, 10), "wsj_recipe_shape_S20": lambda: recipe_leg(dev, 20),
            # the same shape with the minibatch this part wants (INTEGRATION.md "Which --num-sequence"): the frame limit raised
            # so that --num-sequence is what bounds a minibatch
            "wsj_recipe_shape_S32": lambda: recipe_leg(dev, 32, 256, 100000), "wsj_recipe_shape_S64": lambda: recipe_leg(dev, 64, 256, 100000)}


def check_full_cfg3(comm, dev: int, rank: int, world: int, all_reduce) -> dict:
    """--check full_cfg3 (N = 8): SURVEY.md section 8e's parity statement through the REAL exchange.  Rank r runs shard r of the global
    minibatch of BASELINE configs[2] (256 utterances, T = 1000, 4 x 512; the interleaved deal of eesen_amd.parallel.shard_batch) with
    lr = 1, momentum 0, no clipping, and the communicator attached; what eesen_net_get_grads returns after Backpropagate is then the
    all-reduced gradient of the 256 utterances.  It is held against the committed fixture tests/golden/full_cfg3.npz -- ONE process of
    the reference itself at --num-sequence 256 (its CUDA CTC bodies executed on the CPU; made by oracle/fullsize.py where
    /root/reference exists) -- on the fixture's sample (every 1009th element + every tensor of <= 8192 elements whole) and its
    per-tensor sums, at the fixture's own bars: per tensor max(1e-4, 3 x the reference's fp32-vs-fp64-CTC floor).  Data only is read
    here (the .npz); nothing under oracle/ is imported.  Every rank calls this (collectives inside); rank 0's dict is the result."""
    from eesen_amd.api import Net, Ctc
    from eesen_amd.parallel import shard_batch, deal_shards
    W, STRIDE, SMALL, TOL = 8, 1009, 8192, 1e-4
    if world != W:
        return {"skipped": f"needs 8 ranks (the fixture is the 8-shard global minibatch); this run has {world}"}
    fx = np.load(os.path.join(ROOT, "tests", "golden", "full_cfg3.npz"))
    cfg = synth.config("cfg2"); cfg["S"] = 256
    layers = synth.make_model(**cfg)                       # no <MaxGrad>: the delta of an lr = 1 step IS the gradient
    glob = synth.make_batch(**cfg)
    sb = shard_batch(glob, rank, W)
    net = Net.from_layers(layers, device=dev)
    net.SetTrainOptions(1.0, 0.0)
    net.SetComm(comm)
    ctc = Ctc(device=dev)
    net.SetSeqLengths(sb.lens)
    out = net.Propagate(sb.feats)
    diff = ctc.EvalParallel(sb.lens, out, sb.labels)
    net.BackpropagateNoUpdate(diff)
    g = net.GetGrads().astype(np.float64)                  # waits for the buckets: the SUM over the eight ranks
    plan = net.Plan()
    info = net.RecurrenceInfo()
    net.SetComm(None)
    # ln p of this rank's utterances against the fixture's, worst over the ranks
    mine = deal_shards(glob.S, W)[rank]
    e_pzx = float(np.max(np.abs(ctc.pzx.astype(np.float64) - fx["pzx"][mine]) / np.maximum(np.abs(fx["pzx"][mine]), 1e-30)))
    e_pzx = all_reduce([e_pzx], 1)[0]
    # the fixture's gradient sample and per-tensor statistics, in Net::GetParams order
    bounds, i = [], 0
    for L in layers:
        for prm in L["params"]:
            bounds.append((i, i + prm.size)); i += prm.size
    m = np.zeros(i, bool); m[::STRIDE] = True
    for a, b in bounds:
        if b - a <= SMALL:
            m[a:b] = True
    idx = np.flatnonzero(m)
    fs, floors = fx["grad_stats"], fx["floor_grads"]
    worst, worst_ratio, bad = 0.0, 0.0, []
    for t, (a, b) in enumerate(bounds):
        gt = g[a:b]
        e = abs(np.max(np.abs(gt)) - fs[t, 0]) / fs[t, 0]
        e = max(e, abs(gt.sum() - fs[t, 1]) / fs[t, 2], abs(np.abs(gt).sum() - fs[t, 2]) / fs[t, 2])
        lo, hi = np.searchsorted(idx, [a, b])
        if hi > lo:
            e = max(e, float(np.max(np.abs(g[idx[lo:hi]] - fx["grad_sample"][lo:hi].astype(np.float64))) / fs[t, 0]))
        bar = max(TOL, 3.0 * float(floors[t]))
        worst, worst_ratio = max(worst, float(e)), max(worst_ratio, float(e) / bar)
        if not (e < bar and e < max(3 * TOL, float(floors[t]))):
            bad.append(t)
    # every rank must have computed the same verdict from the same summed gradient
    v = all_reduce([worst, -worst], 1)
    same = v[0] == -v[1]
    ok = bool(not bad and e_pzx < TOL and same)
    return {"ok": ok, "fixture": "tests/golden/full_cfg3.npz: one reference process, --num-sequence 256 (made by oracle/fullsize.py)",
            "shards": W, "utterances_per_shard": sb.S, "ln_p_worst_rel_err_per_sequence": e_pzx,
            "summed_gradient_worst_tensor_error": worst, "worst_error_over_its_bar": worst_ratio, "tensors": len(bounds), "tensors_over_bar": bad,
            "bar": "per tensor max(1e-4, 3 x the reference's own fp32-vs-fp64-CTC floor), as tests/test_gpu_reference_fullsize.py",
            "every_rank_sees_the_same_sum": bool(same), "exchange": plan["exchange"],
            "persistent_layers": {"fwd": info["fwd_persistent"], "bwd": info["bwd_persistent"], "of": info["lstm_layers"]}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip config.secondary (cfg4 / cfg5 / the recipe shape, a few timed steps each)")
    ap.add_argument("--main-only", action="store_true",
                    help="profiling runs: skip the secondary legs (f32-MFMA-GEMM comparison, PCIe-inclusive loop, standalone GEMM, CPU baseline, config.secondary)")
    ap.add_argument("--T", type=int, default=0, help="override T_max (debug)")
    ap.add_argument("--H", type=int, default=0, help="override cells per direction (debug)")
    ap.add_argument("--S", type=int, default=0, help="override utterances per GPU (debug)")
    ap.add_argument("--layers", type=int, default=0, help="override layer count (debug)")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-GPU code path (RCCL communicator, per-layer gradient all-reduce) even with one rank (debug)")
    ap.add_argument("--forward-precision", choices=["f32", "bf16", "bf16-gemm"], default="f32",
                    help="bf16: BASELINE config 4's variant -- forward GEMMs and the forward time recurrence on bf16 operands, fp32 accumulate "
                         "(bf16-gemm: the GEMM operands only, the round-3 variant); never the headline (reported with dtype bf16-fwd/f32)")
    ap.add_argument("--comm", choices=["native", "bulk", "torch"], default="native",
                    help="gradient exchange: native = the library's RCCL communicator, one bucket per layer overlapped with the backward "
                         "pass (default); bulk = the same communicator, one all-reduce after the backward pass; torch = torch.distributed")
    ap.add_argument("--check", choices=["full_cfg3"], default=None,
                    help="N = 8 only: before the timed steps, the eight shards of BASELINE configs[2]'s global minibatch (256 utterances) through the "
                         "communicator -- the all-reduced gradient against tests/golden/full_cfg3.npz (ONE reference process at --num-sequence "
                         "256) at the fixture's bars; the result goes into the line as config.check_full_cfg3")
    ap.add_argument("--leg", default=None, help="(internal) run ONE leg of config.secondary in this process and print its record as a JSON line")
    args = ap.parse_args()
    if args.leg:
        print(json.dumps(secondary_legs(0)[args.leg]()), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    # stdout carries exactly ONE JSON line: everything else that C libraries print there (RCCL prints its version banner on
    # stdout, buffered until exit) is rerouted to stderr by swapping the file descriptors for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    multi = world > 1 or args.force_dist
    from eesen_amd.api import Net, Ctc, CuMatrix, Comm
    from eesen_amd import _lib
    dist = None
    comm = None
    if multi and args.comm != "torch":
        try:
            comm = Comm.from_env(device=local)   # the library's own RCCL communicator: no torch in the process
        except Exception as e:  # rendezvous port taken, librccl not loadable, ...: fails on every rank alike, so every rank takes the same fallback
            print(f"[bench] rank {rank}: native communicator unavailable ({e}); falling back to torch.distributed", file=sys.stderr)
            args.comm = "torch"
    if multi and args.comm == "torch":
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if args.force_dist and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    def all_reduce(values, op=Comm.SUM):
        """Host scalars over the ranks (sum / max), whatever the transport."""
        if comm is not None:
            return comm.allreduce(values, op)
        if dist is not None:
            t = torch.tensor(list(values), dtype=torch.float64, device=f"cuda:{local}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX if op == Comm.MAX else dist.ReduceOp.SUM)
            return t.tolist()
        return list(values)

    # What the exchange really is: the library the linker resolved, the ranks and devices IT reports, every
    # rank's PCI bus id.  n_gpus of the line = the DISTINCT devices the ranks sit on, not the number of processes.
    comm_desc = None
    if comm is not None:
        comm_desc = comm.Describe()                     # collective
    elif dist is not None:
        props = torch.cuda.get_device_properties(local)
        busid = f"{getattr(props, 'pci_domain_id', 0):04x}:{getattr(props, 'pci_bus_id', local):02x}:{getattr(props, 'pci_device_id', 0):02x}.0"
        allb = [None] * dist.get_world_size()
        dist.all_gather_object(allb, (os.uname().nodename, busid))
        comm_desc = {"library": "torch.distributed (backend nccl = RCCL on ROCm)", "rccl_version": ".".join(str(x) for x in torch.cuda.nccl.version()),
                     "stand_in": False, "rank": dist.get_rank(), "world": world, "world_seen": dist.get_world_size(),
                     "devices": [f"{h}/{b}" for h, b in allb], "distinct_devices": len(set(allb)), "ranks_share_devices": len(set(allb)) < dist.get_world_size()}

    check = None
    if args.check == "full_cfg3":
        if comm is None:
            check = {"skipped": "needs the library's communicator (--comm native / bulk) and 8 ranks"}
        else:
            try:
                check = check_full_cfg3(comm, local, rank, world, all_reduce)
            except Exception as e:   # noqa: BLE001 -- the check must never take the measurement down (it fails on every rank alike or not at all: same code, same data)
                check = {"ok": False, "error": str(e)}

    cfg = synth.config(args.config)
    for k in ("T", "H", "S", "layers"):
        if getattr(args, k):
            cfg[k] = getattr(args, k)
    layers = synth.make_model(max_grad=50.0, **cfg)                 # recipe settings: model_topo.py:90, run_ctc_phn.sh:84-85
    batch = synth.make_batch(**{**cfg, "seed": 777 + rank})         # every rank its own shard of the global batch
    dev = local if multi else 0

    def make_net(attach=True):
        n = Net.from_layers(layers, device=dev)
        n.SetTrainOptions(4e-5, 0.9)
        n.SetProfiling(True)
        n.SetForwardPrecision({"f32": 0, "bf16": 1, "bf16-gemm": 2}[args.forward_precision])
        if not attach:
            return n
        if comm is not None and args.comm == "native":
            n.SetComm(comm)                                   # per-layer buckets, overlapped with the backward pass
        elif comm is not None:
            n.grad_hook = lambda nn: nn.AllReduceGrads(comm)  # one bulk all-reduce between backprop and update
        elif dist is not None:
            from eesen_amd.parallel import GradAllReducer
            n.grad_hook = GradAllReducer(n)
        return n

    net = make_net(attach=False)      # the probing step below runs WITHOUT the exchange: a rank that fails must not strand the others in a collective
    ctc = Ctc(device=dev)
    feats_dev = CuMatrix.from_numpy(batch.feats, dev)             # the device-resident comparison loop's input
    diff = CuMatrix(batch.T * batch.S, cfg["K"], dev)
    # the minibatch as the trainer holds it: S host matrices [T_s x D] (train-ctc-parallel.cc:149-193 reads them from the table)
    from eesen_amd.api import Feeder
    f3 = batch.feats.reshape(batch.T, batch.S, cfg["D"])
    mats = [np.ascontiguousarray(f3[: batch.lens[s], s, :]) for s in range(batch.S)]
    feeder = Feeder(dev, slots=2)
    pending = [feeder.submit(mats)]

    def step():   # the reference trainer's loop body, train-ctc-parallel.cc:186-207, the H2D of :198 included
        slot = pending.pop()
        net.SetSeqLengths(batch.lens)
        out = net.Propagate(feeder.acquire(slot))                               # :198 net.Propagate(CuMatrix<BaseFloat>(feat_mat_host), ...)
        feeder.release(slot)
        ctc.EvalParallel(batch.lens, out, batch.labels, diff, want_pzx=False)   # like the reference's call: accumulates the objective
        ctc.ErrorRateMSeq(batch.lens, out, batch.labels, deferred=True)        # :202: greedy decode + edit distance, every minibatch
        net.Backpropagate(diff)
        pending.append(feeder.submit(mats))     # the NEXT minibatch: packed, copied and interleaved while this one's backward pass runs
        return out

    def step_resident():   # the same loop body on features that are already in HBM (config.device_resident_*)
        net.SetSeqLengths(batch.lens)
        out = net.Propagate(feats_dev)
        ctc.EvalParallel(batch.lens, out, batch.labels, diff, want_pzx=False)
        ctc.ErrorRateMSeq(batch.lens, out, batch.labels, deferred=True)
        net.Backpropagate(diff)
        return out

    def barrier():
        net.Synchronize()
        if multi:
            all_reduce([0.0])                                  # every rank has drained its own work
        _lib.check(_lib.load().eesen_device_synchronize(dev))  # = torch.cuda.synchronize() on this rank's GPU

    # One probing step first: if the cooperative (persistent) recurrence kernels cannot run on this box (the library then
    # raises instead of hanging), fall back to the one-launch-per-step kernels -- on EVERY rank, so all ranks time the same code.
    ok = 1.0
    try:
        step()
        net.Synchronize()
    except Exception as e:   # noqa: BLE001
        print(f"bench: persistent path failed on rank {rank} ({e}); falling back to per-step kernels", file=sys.stderr)
        ok = 0.0
    if multi:
        ok = 1.0 - all_reduce([1.0 - ok], Comm.MAX)[0]        # any rank failed -> every rank falls back
    if ok == 0.0:
        os.environ["EESEN_PERSISTENT"] = "0"
    net = make_net()                  # same initial weights on every rank (seed 777), now with the gradient exchange attached
    for _ in range(args.warmup):
        step()
    barrier()
    # The per-phase HIP-event timers (recorded on the library's own streams) cover exactly the K timed steps: the library
    # accumulates their spans and they are read once, after the closing barrier -- no host synchronisation inside the region
    # beyond what a training step does itself (the CTC returns ln p to the host every step).
    net.SetProfiling(True, accumulate=True)
    ctc.SetProfiling(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    spans = net.PhaseSpans()           # every timed span of the K steps, in record order (before PhaseTimes clears them)
    main_plan = net.Plan()             # what the library's selection functions chose for this shape -- with the exchange schedule
    # Do the ranks still hold the SAME model after warm-up + K steps?  Every rank hashes its parameters (48 bits of sha1: exact in the
    # doubles the communicator's host all-reduce carries); identical iff max == min over the ranks.
    ranks_identical = None
    if multi:
        import hashlib
        hv = float(int.from_bytes(hashlib.sha1(net.GetParams().tobytes()).digest()[:6], "big"))
        mx = all_reduce([hv, -hv], Comm.MAX)
        ranks_identical = bool(mx[0] == -mx[1])
    phases = net.PhaseTimes()
    ctc_ph = ctc.PhaseTimes()
    ctc.SetProfiling(False)
    net.SetProfiling(True)
    if multi:
        dt = all_reduce([dt], Comm.MAX)[0]                                       # the slowest rank's clock
        padded, real = all_reduce([float(batch.T * batch.S), float(batch.real_frames)])
    else:
        padded, real = float(batch.T * batch.S), float(batch.real_frames)

    # Not the headline either: the same K steps with every GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32, an exact fp32 fmaf
    # chain) instead of the default 3-way bf16 split (fp32-class accuracy on the bf16 matrix pipe, tests/test_gpu_gemm.py) -- so
    # that both arithmetic modes are on record from the same box and process.
    f32_only = None
    bf16_split = None
    if world == 1 and not args.main_only and gemm_products() == 3:
        # Each in its OWN process (this script with --main-only and EESEN_GEMM_MODE set): a second Net of another arithmetic inside this
        # process measured 40.1-41.3 ms for the six-product mode where a process of its own measures 35.8 (the step time of that mode is
        # bimodal with where the Net's buffers land, scripts/leg_artifact.py) -- a comparison must not depend on that draw.  This process
        # idles meanwhile.
        import subprocess
        for mode, env_mode, label in ((0, "f32", "v_mfma_f32_32x32x2_f32 (EESEN_GEMM_MODE=f32)"),
                                      (1, "split", "three bf16 planes per operand, six products on v_mfma_f32_32x32x16_bf16 (EESEN_GEMM_MODE=split: the default of "
                                                   "rounds 2-5; the recurrences as in the headline)")):
            try:
                cmd = [sys.executable, os.path.abspath(__file__), "--main-only", "--steps", str(args.steps), "--warmup", str(max(2, args.warmup)), "--config", args.config,
                       "--forward-precision", args.forward_precision]
                for k in ("T", "H", "S", "layers"):
                    if getattr(args, k):
                        cmd += ["--" + k, str(getattr(args, k))]
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env={**os.environ, "EESEN_GEMM_MODE": env_mode})
                d2 = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
                rec = {"ms_per_step": d2["ms_per_step"], "frames_per_s": d2["value"], "gemm": label, "how": "its own process: " + " ".join(cmd[1:]) + f" with EESEN_GEMM_MODE={env_mode}"}
            except Exception as e:   # noqa: BLE001 -- a comparison leg must never take the measurement down
                rec = {"error": str(e)}
            if mode == 0:
                f32_only = rec
            else:
                bf16_split = rec

    # Not the headline: the same K steps with the features already RESIDENT in HBM (what rounds 1-4 reported as `value`): what the
    # per-step H2D costs the step is the difference (the copy and the interleave of minibatch n + 1 run on the feeder's stream under
    # step n's backward pass; what is left is the host's pack into the pinned slot on the enqueueing thread -- hidden while the host
    # runs ahead of the device -- and the interleave kernel's share of the chip).  Same warm-up, same process, same Net.
    resident = None
    if world == 1 and not args.main_only:
        for _ in range(max(2, args.warmup)):
            step_resident()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step_resident()
        barrier()
        dt1 = time.perf_counter() - t1
        resident = {"ms_per_step": 1e3 * dt1 / args.steps, "frames_per_s": float(batch.T * batch.S) * args.steps / dt1}

    if rank == 0:
        K = args.steps
        value = padded * K / dt
        fpf = flops_per_frame(cfg)
        fwd_mode = {"f32": 0, "bf16": 1, "bf16-gemm": 2}[args.forward_precision]
        fprod = fwd_rec_products(cfg, batch.S, fwd_mode)
        gprod = gemm_products()
        pb = pipe_bound(cfg, gprod, fprod, bwd_rec_products(cfg))
        nd = 2 if cfg["kind"].startswith("BiLstm") else 1
        H, S, T, nl = cfg["H"], batch.S, batch.T, cfg["layers"]
        # per-launch algorithmic work of the three kernels that carry the step (DESIGN.md "kernels")
        rec_flops = 2.0 * S * 4 * H * H * nd                              # one recurrence step, all directions
        n_rec = T * nl * K
        gemm_flops = sum(2.0 * T * S * nd * 4 * H * (cfg["D"] if li == 0 else nd * H) for li in range(nl)) / nl
        persistent = os.environ.get("EESEN_PERSISTENT", "1") != "0"   # one cooperative launch per layer pass (default)
        if persistent:
            n_rec, rec_flops = nl * K, rec_flops * T
        kn = "persistent_kernel" if persistent else "step_kernel"
        # the backward pass takes the 4-sequence x 32-unit tile where lstm_bwd_persistent (lstm_persistent.hip) does: the 8-sequence
        # tile would be chosen (16-sequence tiles leave half the CUs idle), H a multiple of 128 up to 512, whole 4-sequence tiles
        q4 = (persistent and H % 128 == 0 and H <= 512 and S % 4 == 0 and S > 8 and 2 * ((H + 15) // 16) * nd * ((S + 15) // 16) <= 256
              and os.environ.get("EESEN_BWD_Q4", "1") != "0")
        split = gprod > 0
        gk = "gemm_f32_split_f16" if gprod == 3 else "gemm_f32_split_bf16"
        bf16_fwd = args.forward_precision != "f32"
        # the main-stream input->gates GEMMs take the 256 x 256-tile flavour when the shape holds >= 16 whole big tiles (gemm.hip)
        # (layer 1's K = 40 is not a multiple of 16 and takes the 128 x 128 flavour; the name is that of the layers that dominate)
        big = not bf16_fwd and (T * S) % 256 == 0 and (nd * 4 * H) % 256 == 0 and ((T * S) // 256) * ((nd * 4 * H) // 256) >= 16
        # "the middle first" (net.cpp): with the persistent kernels every upper layer's input GEMM is three launches -- the middle half
        # of the rows on the 128 x 128 flavour under the end of the recurrence below, the two ends on the main stream -- and the
        # figure below is their summed time per layer (the middle part's duration is that of a kernel sharing the chip)
        mid_first = persistent and nd == 2 and nl > 1 and H <= 512 and os.environ.get("EESEN_FWD_MID", "1") != "0" and os.environ.get("EESEN_OVERLAP", "1") != "0"
        gemm_name = ("gemm_f32_mfma_kernel" if not split and not bf16_fwd else
                     ((gk + "_big_kernel") if big else ("gemm_f32_split_bf16_kernel" if bf16_fwd else gk + "_kernel"))) + \
                    (f"(input->gates: 2 ends + {gk}_kernel middle under the recurrence)" if mid_first and split and not bf16_fwd else "(input->gates)")
        bwd_name = "lstm_bwd_persistent_q4_kernel" if q4 else ("lstm_bwd_persistent_ksplit_h_kernel (fp16 planes)" if persistent and bwd_rec_products(cfg) else "lstm_bwd_" + kn)
        # the forward recurrence on the bf16 pipe (lstm_fwd_persistent_bf_kernel): the fp32-class 3-way split of the narrow tile (six
        # products), or config 4's bf16 forward (m_t one plane, W_m hi + lo: two products)
        fwd_name = "lstm_fwd_" + kn if not fprod else {6: "lstm_fwd_persistent_bf_kernel<AP=3, WP=3> (bf16 planes)", 3: "lstm_fwd_persistent_bf_kernel<AP=2, WP=2, F16> (fp16 planes)",
                                                        2: "lstm_fwd_persistent_bf_kernel<AP=1, WP=2> (bf16 forward)"}[fprod]
        kern = {
            fwd_name: dict(total_s=phases["recurrence_fwd"], launches=n_rec, flops=rec_flops, pipe="f32" if not fprod else "bf16", products=fprod or 1),
            bwd_name: dict(total_s=phases["recurrence_bwd"], launches=n_rec, flops=rec_flops, pipe="f32" if not bwd_rec_products(cfg) else "bf16", products=bwd_rec_products(cfg) or 1),
            gemm_name: dict(total_s=phases["input_gemm"], launches=nl * K, flops=gemm_flops, pipe="f32" if (not split and not bf16_fwd) else "bf16"),
        }
        for k in kern.values():
            k["avg_us"] = 1e6 * k["total_s"] / k["launches"]
            k["achieved"] = k["flops"] / (k["total_s"] / k["launches"]) / 1e12      # fp32-equivalent (algorithmic) TFLOP/s
            # what the matrix pipe EXECUTES: the split GEMM runs six bf16 products per fp32 product (one with bf16-rounded forward operands)
            k["executed"] = k["achieved"] * (1 if k["pipe"] == "f32" else k.get("products", 1 if bf16_fwd else gprod))
            k["peak"] = PEAK_F32_MFMA_TFLOPS if k["pipe"] == "f32" else PEAK_BF16_MFMA_TFLOPS
            k["frac"] = k["executed"] / k["peak"]                                     # never above 1: executed flops over that pipe's peak
        dom = max(kern, key=lambda n: kern[n]["total_s"])
        # HBM traffic per launch and matrix-pipe occupancy of the dominant kernel, from the committed PMC passes (rocprofv3 --pmc cannot
        # run inside this process); only quoted when the workload is the one the counters were collected on
        traffic = mfma_busy = traffic_source = None
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pt["config"] == {"config": args.config, "T": T, "S": S} and not (args.H or args.layers):
                traffic = pt["bytes_per_launch"].get(dom)
                mfma_busy = pt.get("mfma_busy", {}).get(dom)
                # where the counters come from: the tables, the commit they were collected at, and the switches of that run
                from eesen_amd.build import csrc_digest
                # stale: the library's sources are not the ones the counters were collected on (sha1 over csrc/ + include/: the GPU box has no .git)
                traffic_source = {"file": "profiles/pmc_traffic.json", "tables": pt.get("source"), "commit": pt.get("commit"),
                                  "csrc_sha": pt.get("csrc_sha"), "stale": pt.get("csrc_sha") != csrc_digest(),
                                  "switches": pt.get("switches"), "kernel": pt.get("kernels", {}).get(dom)}
        except Exception:
            pass
        # The launch that runs ALONE on the chip: in the backward pass the top LSTM layer's recurrence starts when no side-stream
        # GEMM is in flight (the weight-gradient GEMMs of a layer start behind its recurrence), the others share the chip with the
        # layer above's gradient GEMMs.  Spans of one phase arrive in launch order: top layer first, every step.
        alone = None
        try:
            ph_name = "recurrence_bwd" if dom.startswith("lstm_bwd") else ("recurrence_fwd" if dom.startswith("lstm_fwd") else None)
            if ph_name and persistent:
                sp = [sec for name, sec in spans if name == ph_name]
                if len(sp) == nl * K:
                    first = sp[0::nl] if ph_name == "recurrence_bwd" else sp[0::nl]
                    us = 1e6 * float(np.mean(first))
                    alone = {"avg_launch_us": us, "achieved": kern[dom]["flops"] / (us * 1e-6) / 1e12,
                             "frac": kern[dom]["flops"] / (us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                             "which": ("the top layer's launch of every step: no side-stream GEMM is in flight beside it" if ph_name == "recurrence_bwd"
                                       else "the lowest layer's launch of every step")}
        except Exception:
            pass
        roofline = {"bound": "mfma", "kernel": dom, "achieved": kern[dom]["executed"], "peak": kern[dom]["peak"],
                    "unit": "TFLOP/s", "frac": kern[dom]["frac"], "traffic": traffic, "traffic_source": traffic_source, "mfma_busy": mfma_busy, "alone": alone,
                    "avg_launch_us": kern[dom]["avg_us"], "flops_per_launch": kern[dom]["flops"],
                    "note": ("one launch = the whole T-step recurrence of a layer; its duration is set by the per-step chain -- a 32 KB "
                             "operand fetch through one CU's L1 beside 0.9 us of MFMA (4 x 32 tile on v_mfma_f32_4x4x1_16B_f32 with the "
                             "A-operand broadcast), the cell update, and the hand-off (counter-increment flight + poll round trip + drain "
                             "of the write-through stores), DESIGN.md sections 4, 9 and 10 -- and is measured while the launch shares the "
                             "chip with the overlapped GEMMs (`alone`: the launch that does not); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES over "
                             "the kernel's SIMD-cycles from the committed PMC pass; whole_step is the step's total FLOPs over its time"),
                    "whole_step": {"achieved": fpf * value / world / 1e12, "unit": "TFLOP/s fp32-equivalent", "flops_per_frame": fpf,
                                   **pb, "frac": pb["bound_us_per_frame"] * 1e-6 * value / world,
                                   "note": "frac = (fp32-pipe recurrence flops / fp32-MFMA peak + products x (GEMM + 16-bit-pipe recurrence) flops / 16-bit MFMA peak) / measured time: the step against BOTH matrix pipes at their peaks, <= 1 by construction (gemm_products per fp32 product: 3 = two fp16 planes, 6 = three bf16 planes); see config.f32_mfma_gemm_only / config.bf16_split_gemm for the same step in the other arithmetic modes"},
                    "other_kernels": {n: ({"achieved_fp32_equivalent": v["achieved"], "executed": v["executed"], "pipe": v["pipe"], "peak": v["peak"],
                                           "frac": v["frac"], "avg_launch_us": v["avg_us"]})
                                      for n, v in kern.items() if n != dom}}
        # The gate GEMM alone (same kernel, same shape, HIP events around 5 back-to-back launches): inside the step its launches
        # are GATED on the recurrence's arrival counters and run under it, so their in-step duration says nothing about the
        # matrix pipe.  This is the figure to hold against the >= 60 % MFMA target on the gate GEMMs.
        try:
            if args.main_only:
                raise RuntimeError("skipped (--main-only)")
            import ctypes as C
            Mg, Ng, Kg = T * S, nd * 4 * H, nd * H
            rg = np.random.default_rng(1)   # random operands: all-zero inputs would flatter the clocks
            A_ = CuMatrix.from_numpy(rg.uniform(-1, 1, (Mg, Kg)).astype(np.float32), dev)
            B_ = CuMatrix.from_numpy(rg.uniform(-0.1, 0.1, (Ng, Kg)).astype(np.float32), dev)
            C_ = CuMatrix(Mg, Ng, dev, zero=False)
            lib = _lib.load()
            gg = {}
            for mode, name in ((0, "f32_mfma"), (1, "bf16_split"), (2, "f16_planes")):
                _lib.check(lib.eesen_set_gemm_mode(mode))
                ms = C.c_float()
                _lib.check(lib.eesen_op_gemm_bench(dev, 1, 1, Mg, Ng, Kg, C.c_void_p(A_.ptr), A_.stride, C.c_void_p(B_.ptr), B_.stride,
                                                   C.c_void_p(C_.ptr), C_.stride, 5, C.byref(ms)))
                tf = 2.0 * Mg * Ng * Kg / ms.value / 1e9
                if mode == 0:
                    gg[name] = {"kernel": "gemm_f32_mfma_kernel (v_mfma_f32_32x32x2_f32)", "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS,
                                "unit": "TFLOP/s", "frac": tf / PEAK_F32_MFMA_TFLOPS, "avg_launch_us": 1e3 * ms.value}
                elif mode == 1:   # six bf16 MFMA products per fp32 product: the matrix pipe executes 6x the useful flops
                    gg[name] = {"kernel": "gemm_f32_split_bf16_kernel (6 x v_mfma_f32_32x32x16_bf16 per 32x32x16 block)",
                                "achieved_fp32_equivalent": tf, "executed_bf16": 6 * tf, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                                "frac": 6 * tf / PEAK_BF16_MFMA_TFLOPS, "vs_f32_mfma_peak": tf / PEAK_F32_MFMA_TFLOPS, "avg_launch_us": 1e3 * ms.value}
                else:             # two fp16 planes per operand, three products (the default): 3x the useful flops
                    gg[name] = {"kernel": "gemm_f32_split_f16_kernel (3 x v_mfma_f32_32x32x16_f16 per 32x32x16 block; operand bounds measured outside the timed launches)",
                                "achieved_fp32_equivalent": tf, "executed_f16": 3 * tf, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                                "frac": 3 * tf / PEAK_BF16_MFMA_TFLOPS, "vs_f32_mfma_peak": tf / PEAK_F32_MFMA_TFLOPS, "avg_launch_us": 1e3 * ms.value}
            _lib.check(lib.eesen_set_gemm_mode(-1))
            roofline["gate_gemm_standalone"] = dict(shape=[Mg, Ng, Kg], **gg)
            del A_, B_, C_
        except Exception as e:  # noqa: BLE001
            roofline["gate_gemm_standalone"] = {"error": str(e)}
        roofline["ctc"] = ctc_block(cfg, batch, ctc_ph, K)
        line = {
            "metric": "CTC training frames/sec (whole node), 4x512 BiLSTM",
            "value": value, "unit": "frames/s", "n_gpus": (comm_desc["distinct_devices"] if comm_desc else world), "steps": K, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.forward_precision == "f32" else "bf16-fwd/f32", "data": "synthetic (seed 777: N(0,1) 40-d features, lengths U{0.8T..T}, T/10 labels per utterance, U(-0.1,0.1) weights)",
            "config": {"workload": f"{args.config}: {nl}x{H} {'Bi' if nd == 2 else ''}LSTM + affine + softmax + CTC, D={cfg['D']}, K={cfg['K']}, "
                                   f"S={S} utterances/GPU, T_max={T}, SGD lr=4e-5 momentum=0.9 max_grad=50",
                       "global_batch_utterances": S * world, "parallelism": f"dp{world}", "ranks": world,
                       # flat on purpose (a reader that keeps only scalars still sees them): did N ranks on N distinct GPUs meet through RCCL, and did they stay identical
                       "ranks_share_devices": (comm_desc["ranks_share_devices"] if comm_desc else None),
                       "distinct_devices": (comm_desc["distinct_devices"] if comm_desc else None),
                       "comm_stand_in": (comm_desc["stand_in"] if comm_desc else None),
                       "comm_library": (comm_desc["library"] if comm_desc else None),
                       "comm_world_seen": (comm_desc["world_seen"] if comm_desc else None),
                       "ranks_bit_identical": ranks_identical,
                       "check_full_cfg3_ok": (check.get("ok") if check else None),
                       "vs_1gpu": None,     # for the driver to fill: value / (N x the N = 1 value of the same round)
                       "comm": comm_desc,
                       "check_full_cfg3": check,
                       "gradient_exchange": (None if not multi else
                                             ("TEST STAND-IN (tests/native/libfake_rccl.so, NOT RCCL): " if comm_desc and comm_desc["stand_in"] else "RCCL ") +
                                             {"native": "all-reduce(sum, fp32) per layer bucket on a communication stream, issued by libeesen_hip.so",
                                              "bulk": "one all-reduce of the whole gradient buffer after the backward pass (libeesen_hip.so)",
                                              "torch": "one torch.distributed all-reduce of the whole gradient buffer"}[args.comm]),
                       "exchange_schedule": (None if not multi or args.comm != "native" else main_plan["exchange"]),
                       "exchange_rule": (None if not multi or args.comm != "native" else main_plan["exchange_rule"]),
                       "kernels": plan_summary(main_plan),
                       "real_frames_per_s": real * K / dt, "padded_frames_per_step": padded, "real_frames_per_step": real,
                       "h2d": "inside the timed step: S host matrices -> pinned slot -> one PCIe copy -> time-major interleave on the device (feeder), double-buffered",
                       "device_resident_frames_per_s": resident["frames_per_s"] if resident else None,
                       "device_resident_ms_per_step": resident["ms_per_step"] if resident else None,
                       "gemm_arithmetic": {0: "f32-input MFMA (exact fp32 fmaf chain)",
                                           6: "fp32 operands split exactly into 3 bf16 terms, 6 of the 9 cross products on v_mfma_f32_32x32x16_bf16 with fp32 "
                                              "accumulation (error <= 2^-23 |ab| per product = one fp32 rounding; measured against fp64 equal to the fp32 chain, "
                                              "tests/test_gpu_gemm.py); recurrence, CTC and update in plain fp32",
                                           3: "fp32 operands held as TWO fp16 planes (round to nearest at both levels: hi + lo = the value to within 2^-22), each row of "
                                              "op(A) / column of op(B) times the power of two its own largest magnitude asks for (measured on the device), 3 of the 4 "
                                              "cross products on v_mfma_f32_32x32x16_f16 with fp32 accumulation (error <= 3 * 2^-22 |ab| per product, the '3xTF32' arithmetic; measured against "
                                              "fp64 equal to the fp32 chain on every shape of tests/test_gpu_gemm.py); the forward recurrence on the same planes; "
                                              "backward recurrence, CTC and update in plain fp32"}[gprod],
                       "f32_mfma_gemm_only": f32_only, "bf16_split_gemm": bf16_split},
            "phase_ms_per_step": {k: 1e3 * v / K for k, v in {**phases, **{'ctc_' + a: b for a, b in ctc_ph.items()}}.items()},
            "roofline": roofline,
        }
        if comm is not None and args.comm == "native":
            # The exchange (replaces communicator.h:39-170), from the library's own events (eesen_net_get_phase_spans,
            # phases 6 and 7), rank 0's view: per gradient bucket the all-reduce on the communication stream -- its duration, the bus
            # bandwidth 2 (N-1)/N x bytes / time a ring moves per GPU, against the xGMI peak of one GPU (7 links x 153 GB/s) -- and what of
            # it the backward pass did NOT hide (the time eesen_net_update's stream waited for buckets).
            order = net.BucketOrder()
            nb = max(1, len(order))
            ar = [sec for nm, sec in spans if nm == "allreduce"]
            ex = [sec for nm, sec in spans if nm == "allreduce_exposed"]
            bytes_of = {li: 4.0 * sum(int(np.size(p)) for p in layers[li]["params"]) for li in order}
            ring = 2.0 * (world - 1) / world
            buckets = []
            for bi, li in enumerate(order):
                tt = ar[bi::nb]
                sec = sum(tt) / max(1, len(tt))
                gbs = ring * bytes_of[li] / sec / 1e9 if sec > 0 else 0.0
                buckets.append({"layer": li, "MB": bytes_of[li] / 1e6, "ms": 1e3 * sec, "bus_GBps": gbs, "frac_of_xgmi_peak": gbs / PEAK_XGMI_GBS})
            tot_b, tot_s = sum(bytes_of.values()), sum(ar) / K
            line["phase_ms_per_step"]["allreduce"] = 1e3 * tot_s
            line["phase_ms_per_step"]["allreduce_exposed"] = 1e3 * sum(ex) / K
            roofline["exchange"] = {"bound": "xgmi", "achieved": ring * tot_b / tot_s / 1e9 if tot_s > 0 else 0.0, "peak": PEAK_XGMI_GBS, "unit": "GB/s",
                                    "frac": (ring * tot_b / tot_s / 1e9 / PEAK_XGMI_GBS) if tot_s > 0 else 0.0, "bytes_per_step": tot_b,
                                    "ms_per_step": 1e3 * tot_s, "exposed_ms_per_step": 1e3 * sum(ex) / K, "buckets": buckets,
                                    "note": "bus bandwidth of a ring all-reduce, 2 (N-1)/N x bytes / time, on rank 0; peak = 7 xGMI links x 153 GB/s per GPU"
                                            + ("; ONE rank: nothing crosses a link, the figure is the collective's launch + copy cost" if world == 1 else "")}
        if world == 1 and not args.no_cpu_baseline and not args.main_only:
            try:
                line["cpu_baseline"] = cpu_baseline(cfg)
            except Exception as e:  # the baseline leg must never take the GPU number down with it
                line["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        if world == 1 and not args.main_only:
            try:
                line["frontend"] = frontend_leg(dev)
            except Exception as e:  # never takes the headline down
                line["frontend"] = {"error": str(e)}
        if world == 1 and not args.main_only and not args.no_secondary and args.config == "cfg2" and not (args.T or args.H or args.S or args.layers):
            # the other single-GPU BASELINE configurations and the reference's own recipe shape, driver-timed in the same run
            del net, feats_dev, diff, feeder
            sec = {}
            # Every leg in a process of its OWN (this script with --leg NAME): a Net's step time depends on what the process allocated and
            # freed before it -- measured: cfg4 71.1-72.1 ms after one allocation history, 77.5 after another; the recipe leg at
            # --num-sequence 10 19.5 / 23.7 -- and a trainer is one Net per process.  This process has released its own Net and idles meanwhile.
            import subprocess
            for name in secondary_legs(dev):
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg", name], capture_output=True, text=True, timeout=900)
                    sec[name] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                except Exception as e:  # noqa: BLE001
                    sec[name] = {"error": str(e)}
            line["config"]["secondary"] = sec
            # every leg's figure twice more, where a truncated or scalars-only copy of the line still has it: flat in config, and as
            # the LAST object of the line (the driver's stored tail is the last 9 KB)
            legs = {}
            for name, r in sec.items():
                v = r.get("ms_per_step", r.get("ms_per_minibatch")) if isinstance(r, dict) else None
                legs[name] = v if v is not None else (r.get("error") if isinstance(r, dict) else None)
                line["config"]["leg_" + name + ("_ms_per_step" if "ms_per_step" in r else "_ms_per_minibatch")] = v
            line["legs_ms"] = {"headline_cfg2": line["ms_per_step"], **legs,
                               "note": "ms per step (cfg legs, H2D inside the step) / per minibatch (recipe legs); details in config.secondary"}
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if multi:
        all_reduce([0.0])
    if dist is not None:
        dist.destroy_process_group()


def self_launch(n: int) -> int:
    """`bench.py --gpus N` outside any launcher: start the N ranks (one process per GPU, same argv), hand rank 0's stdout --
    the one JSON line -- through, return the worst exit code."""
    import socket
    import subprocess
    with socket.socket() as s0:
        s0.bind(("127.0.0.1", 0))
        port = s0.getsockname()[1]
    procs = []
    for r in range(n):
        # EESEN_BENCH_SHARE_GPU=<index> (tests): every rank on that ONE device -- the multi-rank path of this file through the test-only
        # RCCL stand-in (tests/test_gpu_parallel.py); never a measurement
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=os.environ.get("EESEN_BENCH_SHARE_GPU", str(r)), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


if __name__ == "__main__":
    main()
