"""-m gpu: the library's RCCL communicator at world size 1 (the only size a single-GPU box offers): creation through the TCP
rendezvous path, the per-layer bucket order, and bit-equality of the bucketed / bulk / detached gradient paths.
N > 1 is covered on CPU by tests/test_parallel_cpu.py (product reducer over gloo, sharding, rendezvous between processes)."""
import numpy as np
import pytest

from eesen_amd import synth
from tests.util import rel_err

pytestmark = pytest.mark.gpu


def _steps(net, ctc, batch, n=3):
    out = []
    for _ in range(n):
        net.SetSeqLengths(batch.lens)
        o = net.Propagate(batch.feats)
        d = ctc.EvalParallel(batch.lens, o, batch.labels)
        net.BackpropagateNoUpdate(d)
        if net.grad_hook is not None:
            net.grad_hook(net)
        g = net.GetGrads()
        net.Update()
        out.append((g, net.GetParams()))
    return out


def _steps_reference(layers, batch, n):
    """n plain steps without any exchange."""
    from eesen_amd.api import Net, Ctc
    net = Net.from_layers(layers); net.SetTrainOptions(1e-3, 0.9); ctc = Ctc()
    return _steps(net, ctc, batch, n)[-1][1]


@pytest.fixture(scope="module")
def comm(gpu):
    from eesen_amd.api import Comm
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    c = Comm(0, 0, 1, "127.0.0.1", port, 30)
    assert (c.rank, c.world) == (0, 1)
    return c


def test_host_scalars_and_barrier(comm):
    assert comm.allreduce([1.5, -2.0, 7.0]) == [1.5, -2.0, 7.0]
    assert comm.allreduce([3.0, 4.0], comm.MAX) == [3.0, 4.0]
    comm.barrier()


@pytest.mark.parametrize("defer", ["0", "1"])
@pytest.mark.parametrize("cfg_name,over", [("small_bi", {}), ("cfg2", dict(T=40, S=16, layers=3)), ("tiny_bi", dict(layers=3, proj=12, H=12, T=20, S=4))])
def test_bucketed_bulk_and_detached_paths_are_bit_identical(gpu, comm, cfg_name, over, defer, monkeypatch):
    """defer = "1": EESEN_COMM_DEFER (round 5) -- the same buckets in the same order, issued when the backward pass's last recurrence
    has run instead of as each layer's gradients are enqueued: bit-identical to every other path, the zero-gradient protocol included."""
    from eesen_amd.api import Net, Ctc
    cfg = synth.config(cfg_name); cfg.update(over)
    layers = synth.make_model(max_grad=0.5, **cfg)
    batch = synth.make_batch(**cfg)
    monkeypatch.setenv("EESEN_COMM_DEFER", defer)     # read when a Net is created

    def run(mode):
        net = Net.from_layers(layers); net.SetTrainOptions(1e-3, 0.9); ctc = Ctc()
        if mode == "bucket":
            net.SetComm(comm)
        elif mode == "bulk":
            net.grad_hook = lambda n: n.AllReduceGrads(comm)
        r = _steps(net, ctc, batch)
        return r, net

    ref, _ = run("none")
    buck, nb = run("bucket")
    bulk, _ = run("bulk")
    for (g0, p0), (g1, p1), (g2, p2) in zip(ref, buck, bulk):
        assert np.array_equal(g0, g1) and np.array_equal(g0, g2)
        assert np.array_equal(p0, p1) and np.array_equal(p0, p2)
    # one bucket per TRAINABLE layer, top-down: the order in which Backpropagate completes their gradients, i.e. where the
    # reference calls Update per layer (net.cc:98-104)
    trainable = [i for i, L in enumerate(layers) if L["params"]]
    assert nb.BucketOrder() == trainable[::-1]
    # a rank without a minibatch: zero gradient through the same collectives.  At world size 1 that is a step in which NO rank
    # was live -- the closing round of the zero-gradient protocol -- which must not move the model (the liveness word that
    # rides with the top bucket is 0, the update kernels read it on the device); with a live peer the step would be applied
    # (tests/test_gpu_multirank.py::test_uneven_shards_zero_gradient_protocol)
    before = nb.GetParams()
    nb.BackpropagateZero()
    assert nb.BucketOrder() == trainable[::-1]
    assert not np.any(nb.GetGrads())
    nb.Update()
    assert nb.LiveRanks() == 0
    assert np.array_equal(before, nb.GetParams())
    # ... and the next real step counts this rank again and carries the momentum on as if the closing round had not happened
    nb.SetSeqLengths(batch.lens)
    o = nb.Propagate(batch.feats)
    from eesen_amd.api import Ctc as _Ctc
    d = _Ctc().EvalParallel(batch.lens, o, batch.labels)
    nb.BackpropagateNoUpdate(d)
    nb.Update()
    assert nb.LiveRanks() == 1
    assert rel_err(nb.GetParams(), _steps_reference(layers, batch, 4)) < 1e-6
    nb.SetComm(None)


def test_timed_out_recurrence_contributes_zero_in_a_data_parallel_run(gpu, comm, monkeypatch, capfd):
    """With a communicator attached the other ranks DO apply the step a failed rank could not compute, so the failed rank can
    neither put its garbage gradient into the sum nor skip the update (the ranks would diverge).  Its gradients enter the
    all-reduce as ZEROS -- decided on the device, where the error word is -- its liveness word with them, and it applies the summed
    update like everybody else; the host notices later, warns, and carries on (round 3 raised an error here, also for the harmless
    value 2 of a milestone waiter that gave up: VERDICT r3 item 4c).  One rank: the sum is zero, liveness 0 -> the model does not
    move; the next step runs on the per-step kernels (value 1) / without the early input GEMM (value 2) and equals a clean step."""
    from eesen_amd.api import Net, Ctc
    cfg = synth.config("small_bi"); cfg.update(S=32, T=64)        # (the middle-first schedule needs T >= 32 and a second LSTM layer)
    layers = synth.make_model(**cfg); batch = synth.make_batch(**cfg)

    def one_step(net, ctc):
        net.SetSeqLengths(batch.lens)
        out = net.Propagate(batch.feats)
        d = ctc.EvalParallel(batch.lens, out, batch.labels, want_pzx=False)
        net.Backpropagate(d)

    clean = Net.from_layers(layers); clean.SetTrainOptions(1e-3, 0.9); clean.SetComm(comm)
    one_step(clean, Ctc()); clean.Synchronize()
    want = clean.GetParams()
    clean.SetComm(None)
    for value, spin in ((1, "0"), (2, None)):
        if value == 1:
            monkeypatch.setenv("EESEN_SPIN_LIMIT", spin)          # every bounded spin of the recurrence kernels gives up at once: value 1
        net = Net.from_layers(layers)
        net.SetTrainOptions(1e-3, 0.9)
        net.SetComm(comm)                  # (an explicit EESEN_SPIN_LIMIT is respected; otherwise attaching raises the bound)
        if value == 1:
            monkeypatch.delenv("EESEN_SPIN_LIMIT")
        ctc = Ctc(); ctc.SetGuard(net)
        before = net.GetParams()
        if value == 2:
            net._raise_error_word(2)       # what wait_for_word_kernel stores when it gives up (lstm_persistent.hip)
        one_step(net, ctc)
        net.Synchronize()                  # the host notices here: a WARNING, not an exception
        err = capfd.readouterr().err
        assert "contributed a ZERO gradient to the data-parallel sum" in err, err
        assert ("continuing with the one-launch-per-step kernels" in err) == (value == 1)
        assert ("continuing without the early input GEMM" in err) == (value == 2)
        assert np.array_equal(net.GetParams(), before)            # zero sum, liveness 0: the step did not move the model
        assert net.RecurrenceInfo() is not None and net.recoveries == 1
        one_step(net, ctc); net.Synchronize()                     # the next step is a clean one
        ri = net.RecurrenceInfo()
        assert (ri["fwd_persistent"] == 0) == (value == 1), ri
        assert rel_err(net.GetParams(), want) < 1e-6
        assert net.LiveRanks() == 1
        net.SetComm(None)


@pytest.mark.parametrize("defer", ["0", "1"])
def test_backpropagate_that_throws_still_completes_the_steps_collectives(gpu, comm, defer, monkeypatch, capfd):
    """a Backpropagate that throws half-way (here: a dropout layer asked to backpropagate in test mode,
    bilstm-parallel-layer.h:425 -- the LOWEST layer, so the upper layers' gradients are already complete) on a rank whose peers are
    in the same step.  The peers issue every bucket of the step; the failing rank must too, or they spin until the watchdog's
    EESEN_COMM_TIMEOUT_S -- in the overlapped schedule for the lower buckets, in the deferred one for ALL of them.  The library
    completes the sequence before the exception leaves it: complete gradients as they are, the rest as zeros.  Seen here at world
    size 1 through what is observable: the exception still arrives, the bucket log holds every trainable layer top-down, the
    communicator is alive, and the next (clean) step equals a clean net's."""
    from eesen_amd.api import Net, Ctc, EesenError
    cfg = synth.config("small_bi")
    layers = synth.make_model(max_grad=0.5, **cfg)
    batch = synth.make_batch(**cfg)
    monkeypatch.setenv("EESEN_COMM_DEFER", defer)
    net = Net.from_layers(layers); net.SetTrainOptions(1e-3, 0.9); ctc = Ctc()
    net.SetComm(comm)
    trainable = [i for i, L in enumerate(layers) if L["params"]]
    net.SetLayerDropout(0, {"forward": 0.2})
    net.SetTestMode()
    net.SetSeqLengths(batch.lens)
    o = net.Propagate(batch.feats)
    d = ctc.EvalParallel(batch.lens, o, batch.labels)
    before = net.GetParams()
    with pytest.raises(EesenError, match="test mode"):
        net.BackpropagateNoUpdate(d)
    assert net.BucketOrder() == trainable[::-1]                     # the whole sequence went out, in the peers' order
    assert "all-reduced as zeros" in capfd.readouterr().err
    g = net.GetGrads()                                              # (waits for the buckets: none is stuck)
    from tests.util import split_params
    assert not np.any(split_params(layers, g)[0][2])                # the failed layer's bucket carried zeros
    assert comm.allreduce([2.0]) == [2.0]                           # the communicator is alive
    assert np.array_equal(net.GetParams(), before)
    # the same handle, made sane again, takes a clean step equal to a fresh net's
    net.SetLayerDropout(0, {})
    net.SetTrainMode()
    got = _steps(net, ctc, batch, 1)[-1][1]
    net.SetComm(None)
    assert rel_err(got, _steps_reference(layers, batch, 1)) < 1e-6
