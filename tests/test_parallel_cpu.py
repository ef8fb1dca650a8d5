"""CPU, world_size 2: the data-parallel recipe (shard deal, all-reduce(SUM) of FRESH gradients, then momentum / clip /
update on every rank) equals one process on the whole minibatch — the parity statement of SURVEY.md section 8(e):
N ranks x S == --num-sequence = N*S.  The per-rank arithmetic is the oracle's; the exchange is the PRODUCT's
(eesen_amd.parallel.GradAllReducer over gloo, installed as grad_hook exactly as the trainers install it), and so are the
sharding helpers and the library's TCP rendezvous (eesen_comm_exchange), which need no GPU."""
import os
import socket

import numpy as np
import pytest

from eesen_amd import synth
from eesen_amd.parallel import deal_shards, shard_batch
from tests.util import rel_err


def test_deal_is_a_balanced_partition():
    for n, w in [(32, 8), (33, 4), (5, 8), (256, 8)]:
        sh = deal_shards(n, w)
        assert sorted(i for s in sh for i in s) == list(range(n))
        assert max(map(len, sh)) - min(map(len, sh)) <= 1


def test_shard_repads_to_its_own_tmax():
    cfg = synth.config("small_bi")
    b = synth.make_batch(**cfg)
    parts = [shard_batch(b, r, 3) for r in range(3)]
    assert sum(p.S for p in parts) == b.S and sum(p.real_frames for p in parts) == b.real_frames
    for r, p in enumerate(parts):
        idx = deal_shards(b.S, 3)[r]
        assert p.T == b.lens[idx].max() and p.feats.shape == (p.T * p.S, cfg["D"])
        f_all = b.feats.reshape(b.T, b.S, -1)
        assert np.array_equal(p.feats.reshape(p.T, p.S, -1), f_all[:p.T, idx])


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _HostNet:
    """The reference operator interface (Propagate / BackpropagateNoUpdate / grad_hook / Update, eesen_amd.api.Net) over the CPU
    oracle, so that the PRODUCT's exchange (`eesen_amd.parallel.GradAllReducer` installed as `grad_hook`) can run without a
    GPU: the per-rank arithmetic is the oracle's, the data-parallel step is the product's."""

    def __init__(self, layers, lr, mmt):
        import torch
        from oracle import net as onet
        self.o = onet.OracleNet(layers, "f32")
        self.o.set_train_options(lr, mmt)
        self.n = int(sum(p.size for L in self.o.layers for p in L["params"]))
        self._g = torch.zeros(self.n, dtype=torch.float32)          # the contiguous fresh-gradient buffer
        self.grad_hook = None

    def grad_tensor(self):
        return self._g

    def step(self, batch):
        from oracle import net as onet
        o = self.o
        o.set_seq_lengths(batch.lens)
        out = o.propagate(batch.feats)
        c = onet.ctc_eval_parallel(out, batch.T, batch.S, batch.lens, batch.label_ids, batch.label_off, "f32")
        # BackpropagateNoUpdate: fresh gradients only (momentum term kept out), into the contiguous buffer
        mmt, o.momentum = o.momentum, 0.0
        self.saved = [[x.copy() for x in L["corr"]] for L in o.layers]
        for L in o.layers:
            for x in L["corr"]: x[...] = 0
        o.backpropagate(c["diff"], update=False)
        o.momentum = mmt
        self._g.numpy()[:] = np.concatenate([x.ravel() for L in o.layers for x in L["corr"]])
        if self.grad_hook is not None:
            self.grad_hook(self)                                    # the one exchange step
        g = self._g.numpy(); i = 0                                  # Update: corr = momentum * corr + summed gradient, clip, step
        for L, sv in zip(o.layers, self.saved):
            for x, s0 in zip(L["corr"], sv):
                x[...] = mmt * s0 + g[i:i + x.size].reshape(x.shape); i += x.size
        for li, L in enumerate(o.layers):
            if L["params"]: o.update_layer(li)
        return float(c["pzx"].sum())


def _worker(rank, world, port, cfg, steps, q):
    import torch.distributed as dist
    from eesen_amd.parallel import GradAllReducer, allreduce_stats
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    layers = synth.make_model(max_grad=0.05, **cfg)
    full = synth.make_batch(**cfg)
    mine = shard_batch(full, rank, world)
    net = _HostNet(layers, 1e-3, 0.9)
    net.grad_hook = GradAllReducer(net)                             # what bench.py --comm torch and the trainers install
    lnp = 0.0
    for _ in range(steps):
        lnp = allreduce_stats([net.step(mine)])[0]
    if rank == 0:
        q.put((net.o.get_params(), lnp))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_process_on_the_whole_batch():
    import torch.multiprocessing as mp
    from oracle import net as onet
    cfg = synth.config("tiny_bi"); cfg.update(S=6, T=16)
    steps = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cfg, steps, q)) for r in range(2)]
    for p in procs: p.start()
    params_dp, lnp_dp = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60); assert p.exitcode == 0
    layers = synth.make_model(max_grad=0.05, **cfg)
    full = synth.make_batch(**cfg)
    net = onet.OracleNet(layers, "f32")
    net.set_train_options(1e-3, 0.9)
    for _ in range(steps):
        o = onet.train_step(net, full, "f32")
    assert rel_err(params_dp, net.get_params()) < 1e-5
    assert abs(lnp_dp - float(o["pzx"].sum())) / abs(float(o["pzx"].sum())) < 1e-5


def test_minibatch_sharding_is_a_partition_and_job_substitution():
    from eesen_amd.parallel import shard_minibatches, minibatch_owner, job_rspecifier
    batches = [f"mb{i}" for i in range(11)]
    for world in (1, 2, 3, 8):
        parts = [list(shard_minibatches(iter(batches), r, world)) for r in range(world)]
        assert sorted(x for p in parts for x in p) == sorted(batches)
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
        # consecutive minibatches (neighbours in the length-sorted list) train in the same synchronous step
        for i in range(len(batches)):
            assert batches[i] in parts[minibatch_owner(i, world)]
    assert job_rspecifier("scp:feats.JOB.scp", 2) == "scp:feats.3.scp" and job_rspecifier("ark:x.ark", 5) == "ark:x.ark"
    # only a JOB that stands alone, and only with several jobs (a path or a command that merely CONTAINS the letters is left alone)
    assert job_rspecifier("scp:exp/JOBS/feats.scp", 2) == "scp:exp/JOBS/feats.scp"
    assert job_rspecifier("ark:cat $JOBDIR/f.JOB.ark |", 0) == "ark:cat $JOBDIR/f.1.ark |"
    assert job_rspecifier("scp:feats.JOB.scp", 0, world=1) == "scp:feats.JOB.scp"


def test_python_trainer_treats_its_list_as_its_own_shard(monkeypatch, tmp_path):
    """queue.pl substitutes JOB before the trainer starts, so a job sees `feats_tr.3.scp --job-id=3` and no
    literal JOB: that list is the job's shard and must be trained IN FULL; dealing minibatches of a shared list is opt-in."""
    import inspect
    from eesen_amd import train_ctc_parallel as t
    src = inspect.getsource(t.main)
    assert "if world > 1 and o.shard_shared_list:" in src and '"JOB" in feature_rspecifier' not in src
    o = t.build_parser().read(["--print-args=false", "--num-jobs=4", "--job-id=3", "scp:feats_tr.3.scp", "ark:l", "m", "o"])
    assert o.shard_shared_list is False and o.num_jobs == 4 and o.job_id == 3 and o.args == ["scp:feats_tr.3.scp", "ark:l", "m", "o"]
    assert t.build_parser().read(["--print-args=false", "--shard-shared-list=true", "a", "b", "c", "d"]).shard_shared_list is True
    # node-local shards under one path are not "the same list": the explicit override, off by default; the check's collective
    # is entered by EVERY job whatever its switches (they travel with the hash), and both trainers hash with FNV-1a
    assert o.allow_identical_lists is False
    assert t.build_parser().read(["--print-args=false", "--allow-identical-lists=true", "a", "b", "c", "d"]).allow_identical_lists is True
    assert "1469598103934665603" in src and "if comm is not None:" in src and "crc32" not in src
    native = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "eesen_amd", "csrc", "tools", "train_ctc_parallel.cc")).read()
    assert "1469598103934665603ull" in native and "allow-identical-lists" in native


def test_rendezvous_survives_stray_and_half_open_peers():
    """A port scanner or a peer that hangs up in the middle of the hand-out must not take the rendezvous down."""
    import multiprocessing as mp
    import struct
    import time
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    p0 = ctx.Process(target=_rdv, args=(0, 2, port, q)); p0.start()
    time.sleep(0.5)
    for payload in (b"GET / HTTP/1.0\r\n\r\n", struct.pack("<II", 0x45534e31, 1)):     # garbage; a valid hello from "rank 1" that hangs up at once
        for _ in range(50):
            try:
                c = socket.create_connection(("127.0.0.1", port), timeout=2)
                break
            except OSError:
                time.sleep(0.1)
        c.sendall(payload)
        c.shutdown(socket.SHUT_RDWR) if payload.startswith(b"GET") else None
        c.close()
    p1 = ctx.Process(target=_rdv, args=(1, 2, port, q)); p1.start()
    got = sorted(q.get(timeout=60) for _ in range(2))
    for p in (p0, p1):
        p.join(timeout=30); assert p.exitcode == 0
    # the half-open "rank 1" never confirmed the blob, so it does not count as served: the real rank 1 still gets it
    assert got == [(0, 0, True), (1, 0, True)]


def _rdv(rank, world, port, q):
    import ctypes as C
    from eesen_amd import _lib
    lib = _lib.load()
    buf = C.create_string_buffer(128)
    if rank == 0:
        buf.raw = bytes(range(128))
    rc = lib.eesen_comm_exchange(b"127.0.0.1", port, rank, world, buf, 128, 30)
    q.put((rank, rc, buf.raw == bytes(range(128))))


@pytest.mark.parametrize("world", [3, 8])
def test_library_tcp_rendezvous_hands_the_id_to_every_rank(world):
    """eesen_comm_exchange (the hand-out of the RCCL unique id, include/eesen_hip.h) between `world` processes -- 8 is the node
    the scaling bench runs on; the late starter is rank 0, so the others exercise their connect-retry loop."""
    import multiprocessing as mp
    import time
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rdv, args=(r, world, port, q)) for r in range(1, world)]
    for p in procs: p.start()
    time.sleep(0.5)
    p0 = ctx.Process(target=_rdv, args=(0, world, port, q)); p0.start(); procs.append(p0)
    got = sorted(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(timeout=30); assert p.exitcode == 0
    assert got == [(r, 0, True) for r in range(world)]


def test_bench_pipe_bound_keeps_every_fraction_at_or_below_one():
    """bench.py's whole-step roofline: recurrent products on the fp32 matrix pipe, every other flop as six bf16 products on the
    bf16 pipe, both at peak (a driver record must never carry a fraction above 1).  The bound is a sum of two times,
    so no measured step can beat it; the flops it splits must add up to SURVEY.md section 8(d)'s per-frame figure."""
    import bench
    from eesen_amd import synth
    for name in ("cfg1", "cfg2", "cfg4", "cfg5"):
        cfg = synth.config(name)
        for prod in (3, 6):   # two fp16 planes (the default since round 6) / three bf16 planes per operand
            pb = bench.pipe_bound(cfg, prod)
            assert pb["f32_pipe_flops_per_frame"] + pb["gemm_flops_per_frame_fp32_equivalent"] == pytest.approx(bench.flops_per_frame(cfg))
            assert pb["bf16_pipe_executed_flops_per_frame"] == pytest.approx(prod * pb["gemm_flops_per_frame_fp32_equivalent"])
            lower = pb["f32_pipe_flops_per_frame"] / (bench.PEAK_F32_MFMA_TFLOPS * 1e12) + \
                pb["bf16_pipe_executed_flops_per_frame"] / (bench.PEAK_BF16_MFMA_TFLOPS * 1e12)
            assert pb["bound_us_per_frame"] == pytest.approx(1e6 * lower)
            # the all-f32 arithmetic (EESEN_GEMM_MODE=f32) is bounded by the fp32 pipe alone, and more tightly
            assert bench.pipe_bound(cfg, 0)["bound_us_per_frame"] > pb["bound_us_per_frame"]
        # the forward recurrence on the 16-bit pipe moves its half of the recurrent flops there
        assert bench.pipe_bound(cfg, 3, 3)["bound_us_per_frame"] < bench.pipe_bound(cfg, 3, 0)["bound_us_per_frame"]
