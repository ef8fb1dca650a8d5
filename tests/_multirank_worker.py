"""One rank of a multi-rank run of the LIBRARY's data-parallel path on ONE GPU (started by tests/test_gpu_multirank.py).

Every rank is its own process with its own libeesen_hip.so, its own HIP context on device 0 and the library's own
communicator (eesen_comm_create_tcp -> comm.cpp); what stands in for librccl.so is tests/native/libfake_rccl.so
(EESEN_RCCL_LIBRARY), whose all-reduce kernels exchange through host shared memory because real RCCL wants one device per
rank.  TEST INFRASTRUCTURE.

argv: MODE OUT.npz [options as k=v]; environment: RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT, EESEN_RCCL_LIBRARY, ...
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from eesen_amd import synth  # noqa: E402
from eesen_amd.api import Comm, Ctc, EesenError, Net  # noqa: E402
from eesen_amd.parallel import shard_batch  # noqa: E402


def soak(out, opt):
    """One process, persistent recurrence kernels ON, the stand-in's all-reduce kernels under every backward recurrence."""
    from eesen_amd.api import CuMatrix
    steps = int(opt.get("steps", 200))
    cfg = synth.config(opt.get("cfg", "cfg2"))
    layers = synth.make_model(max_grad=50.0, **cfg)
    batch = synth.make_batch(**cfg)
    feats = CuMatrix.from_numpy(batch.feats)
    diff = CuMatrix(batch.T * batch.S, cfg["K"])

    def run(comm):
        net = Net.from_layers(layers); net.SetTrainOptions(4e-5, 0.9)
        ctc = Ctc(); ctc.SetGuard(net)
        if comm is not None:
            net.SetComm(comm)

        def step():
            net.SetSeqLengths(batch.lens)
            o = net.Propagate(feats)
            ctc.EvalParallel(batch.lens, o, batch.labels, diff, want_pzx=False)
            ctc.ErrorRateMSeq(batch.lens, o, batch.labels, deferred=True)
            net.Backpropagate(diff)
        for _ in range(3):
            step()
        net.Synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        net.Synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        info = net.RecurrenceInfo()
        p = net.GetParams()
        # a second, shorter pass with the library's span timers on: what the exchange itself took and what of it was exposed
        ex = {}
        if comm is not None:
            net.SetProfiling(True, accumulate=True)
            n2 = max(1, min(steps, 50))
            for _ in range(n2):
                step()
            net.Synchronize()
            spans = net.PhaseSpans(); net.PhaseTimes(); net.SetProfiling(False)
            for nm in ("recurrence_bwd", "allreduce", "allreduce_exposed"):
                v = [sec for n_, sec in spans if n_ == nm]
                ex[nm] = 1e3 * sum(v) / n2
                ex["max_" + nm] = 1e3 * max(v or [0.0])
            ex["schedule"] = net.Plan()["exchange"]
            net.RecurrenceInfo()
            net.SetComm(None)
        return ms, info, net.recoveries, p, ctc.Dropped(), ex

    ms0, info0, rec0, p0, _, _ = run(None)
    comm = Comm.from_env(device=0, timeout_s=60)
    standin = any("libfake_rccl.so" in l for l in open("/proc/self/maps"))   # not the real RCCL, whose one-rank all-reduce is a no-op
    ms1, info1, rec1, p1, dropped, ex = run(comm)
    np.savez(out, ms0=ms0, ms1=ms1, info0=list(info0.values()), info1=list(info1.values()), rec0=rec0, rec1=rec1, dropped=dropped,
             identical=np.array_equal(p0, p1), standin=standin, steps=steps, exchange=np.array(__import__("json").dumps(ex)))


def main():
    mode, out = sys.argv[1], sys.argv[2]
    opt = dict(a.split("=", 1) for a in sys.argv[3:])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if mode == "soak":
        return soak(out, opt)
    comm = Comm.from_env(device=0, timeout_s=60)
    assert (comm.rank, comm.world) == (rank, world)
    cfg = synth.config(opt.get("cfg", "small_bi"))
    for k in ("S", "T", "H", "layers"):
        if k in opt:
            cfg[k] = int(opt[k])
    layers = synth.make_model(max_grad=float(opt.get("max_grad", 0.05)), **cfg)
    full = synth.make_batch(**cfg)
    steps = int(opt.get("steps", 3))

    if mode == "die":
        # joined the communicator (rendezvous + bootstrap barrier are through), then gone before the first collective
        if rank == int(opt.get("victim", 1)):
            os._exit(0)

    net = Net.from_layers(layers)
    net.SetTrainOptions(float(opt.get("lr", 1e-3)), float(opt.get("momentum", 0.9)))
    net.SetComm(comm)
    ctc = Ctc()
    ctc.SetGuard(net)

    # straggle_ms=3 straggle_rank=1: that rank's host is late by 3 ms EVERY step -- its collectives arrive late, which is when a
    # resident all-reduce kernel of the punctual rank holds its CUs longest (the straggler case)
    straggle_s = float(opt.get("straggle_ms", 0)) * 1e-3 if rank == int(opt.get("straggle_rank", -1)) else 0.0
    captured = {"grads": [], "params": []}

    def real_step(batch, capture=False):
        if straggle_s:
            time.sleep(straggle_s)
        net.SetSeqLengths(batch.lens)
        o = net.Propagate(batch.feats)
        d = ctc.EvalParallel(batch.lens, o, batch.labels, want_pzx=False)
        ctc.ErrorRateMSeq(batch.lens, o, batch.labels, deferred=True)
        if capture:   # the exchange's result itself: the all-reduced fresh gradient, read between Backpropagate and Update
            net.BackpropagateNoUpdate(d)
            captured["grads"].append(net.GetGrads())     # (eesen_net_get_grads waits for the buckets)
            net.Update()
            captured["params"].append(net.GetParams())
        else:
            net.Backpropagate(d)

    res = {}
    t0 = time.time()
    if mode in ("parity", "die"):
        mine = shard_batch(full, rank, world)     # utterance s -> rank s mod N, re-padded to the shard's own T_max
        try:
            for _ in range(steps):
                real_step(mine, capture=opt.get("grads") == "1")
            net.Synchronize()
            if captured["grads"]:
                res["grads_steps"] = np.stack(captured["grads"]); res["params_each_step"] = np.stack(captured["params"])
            res["error"] = np.array("")
        except EesenError as e:
            res["error"] = np.array(str(e))
            res["code"] = np.array(e.code)
            res["seconds"] = np.array(time.time() - t0)
            np.savez(out, **res)
            os._exit(0)     # the communicator is dead: no orderly teardown through it
    elif mode == "persist":
        # the PRODUCT configuration at N > 1: persistent recurrence kernels in EVERY rank's process + the communicator's per-layer
        # buckets under the backward pass -- two processes on one GPU, each sizing its grids against half of it (EESEN_GPU_SHARE=2).
        # 3 steps whose result goes to the arbiters, then a soak of `soak` more steps: per-step time, recoveries, the exchange's spans.
        mine = shard_batch(full, rank, world)
        for _ in range(steps):
            real_step(mine, capture=opt.get("grads") == "1")
        net.Synchronize()
        if captured["grads"]:
            res["grads_steps"] = np.stack(captured["grads"]); res["params_each_step"] = np.stack(captured["params"])
        res["plan"] = np.array(__import__("json").dumps(net.Plan()))
        res["params_steps"] = net.GetParams()
        res["recurrence_steps"] = np.array(list(net.RecurrenceInfo().values()))
        nsoak = int(opt.get("soak", 0))
        if nsoak:
            comm.barrier()
            net.SetProfiling(True, accumulate=True)
            t1 = time.perf_counter()
            for _ in range(nsoak):
                real_step(mine)
            net.Synchronize()
            comm.barrier()
            res["soak_ms_per_step"] = np.array(1e3 * (time.perf_counter() - t1) / nsoak)
            spans = net.PhaseSpans()
            net.PhaseTimes()
            net.SetProfiling(False)
            for nm in ("recurrence_fwd", "recurrence_bwd", "allreduce", "allreduce_exposed"):
                res["soak_ms_" + nm] = np.array(1e3 * sum(sec for n_, sec in spans if n_ == nm) / nsoak)
                res["soak_max_ms_" + nm] = np.array(1e3 * max([sec for n_, sec in spans if n_ == nm] or [0.0]))   # the longest single span: a recurrence held up by a resident all-reduce shows here
            res["soak_steps"] = np.array(nsoak)
        res["error"] = np.array("")
        net.RecurrenceInfo()
        res["recoveries"] = np.array(net.recoveries)
        res["dropped"] = np.array(ctc.Dropped())
    elif mode == "uneven":
        # rank r holds steps - r minibatches (each a different one): the others keep going, r drains with zero gradients
        n_mine = max(0, steps - rank * int(opt.get("fewer", 1)))
        for k in range(n_mine):
            real_step(synth.make_batch(**{**cfg, "seed": 1000 + 10 * k + rank}))
        zero = 0
        while True:
            net.BackpropagateZero()
            net.Update()
            if net.LiveRanks() == 0:
                break
            zero += 1
        net.Synchronize()
        res["zero_steps"] = np.array(zero)
        res["real_steps"] = np.array(n_mine)
    else:
        raise SystemExit(f"unknown mode {mode}")
    st = ctc.stats()
    tot = comm.allreduce([st["obj_sum"], st["err_tokens"], st["ref_tokens"], st["frames"]])
    res.update(params=net.GetParams(), obj_sum=np.array(tot[0]), err=np.array(tot[1]), ref=np.array(tot[2]), frames=np.array(tot[3]),
               recurrence=np.array(list(net.RecurrenceInfo().values())), seconds=np.array(time.time() - t0))
    np.savez(out, **res)
    comm.barrier()
    net.SetComm(None)


if __name__ == "__main__":
    main()
