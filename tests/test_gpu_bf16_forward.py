"""-m gpu: BASELINE config 4's "bf16 forward" -- eesen_net_set_forward_precision(1): forward GEMMs AND the forward time recurrence on
bf16 operands with fp32 accumulation (lstm_fwd_persistent_bf16_kernel) -- against its arbiter, the reference's forward equations
with the variant's roundings applied where include/eesen_hip.h says they are (oracle/bf16_forward.py, pinned to the reference's
golden fixtures on the CPU by tests/test_bf16_forward_oracle.py).

What is asserted: mode 1 equals the restatement with BOTH roundings, mode 2 (GEMM operands only, the round-3 variant) the one with
the GEMM rounding only, each to 2e-4 of the largest softmax output -- and they differ from each other by far more, so the test can
tell the two apart; the recurrence really ran on the bf16 kernel (eesen_net_bf16_recurrence_layers); the backward pass, which stays
fp32, runs on the bf16 forward's activations and lands within what eight-bit operands allow of the fp32 path's gradients."""
import numpy as np
import pytest

from eesen_amd import synth
from oracle import bf16_forward as bf
from tests.util import rel_err, valid_mask

pytestmark = pytest.mark.gpu


def _forward(layers, batch, mode):
    from eesen_amd.api import Net
    net = Net.from_layers(layers)
    net.SetTrainOptions(1.0, 0.0)
    net.SetForwardPrecision(mode)
    net.SetSeqLengths(batch.lens)
    out = net.Propagate(batch.feats).numpy()
    net.Synchronize()
    ri = net.RecurrenceInfo()
    return net, out, ri


# H per direction (CPW = H / 256 = 1..4), S (whole tiles, a ragged last tile), T
SHAPES = [(256, 32, 40), (512, 20, 25), (768, 16, 12), (1024, 32, 10)]


@pytest.mark.parametrize("H,S,T", SHAPES)
def test_bf16_recurrence_step_by_step_against_the_reference_equations_with_its_roundings(gpu, H, S, T):
    """One BiLSTM layer, its output m compared ONE STEP DEEP: the restatement takes the library's own m_{t-1} as the recurrent
    input of step t (teacher forcing), so an m that lands on the other side of a bf16 rounding boundary in one implementation
    cannot grow through the chain, and the bar is the fp32 one (2e-5 of the largest |m|: v_exp_f32 / v_rcp_f32 in the cell).
    The same comparison WITHOUT the recurrence's roundings in the restatement misses by an order of magnitude more, and so does the
    restatement with W_m as ONE bf16 plane (round 4's arm, removed from the library in round 5): the kernel really multiplies bf16(m)
    with W_m held as hi + lo bf16 planes (17 bits), and nothing else is rounded."""
    cfg = dict(kind="BiLstmParallel", layers=1, H=H, D=40, K=30, S=S, T=T, seed=1234 + H)
    layers = synth.make_model(**cfg)[:1]
    batch = synth.make_batch(**cfg)
    vm = valid_mask(batch.lens, batch.T, batch.S)
    for wp in (2,):
        net, out, ri = _forward(layers, batch, 1)
        assert ri["fwd_persistent"] == ri["lstm_layers"] == 1 and net.Bf16RecurrenceLayers() == 1, ri
        with_r = bf.forward(layers, batch.feats, batch.lens, T, S, bf16_gemm=True, bf16_rec=True, teacher=out, w_planes=wp)
        other = bf.forward(layers, batch.feats, batch.lens, T, S, bf16_gemm=True, bf16_rec=True, teacher=out, w_planes=3 - wp)
        without = bf.forward(layers, batch.feats, batch.lens, T, S, bf16_gemm=True, bf16_rec=False, teacher=out)
        e, e0, e_other = rel_err(out[vm], with_r[vm]), rel_err(out[vm], without[vm]), rel_err(out[vm], other[vm])
        assert e < 2e-5, (wp, e)
        assert e0 > 10 * max(e, 1e-6) and e_other > 5 * max(e, 1e-6), (wp, e, e0, e_other)
        assert np.all(out[~vm] == 0)
    # mode 2 keeps the fp32 recurrence
    net2, out2, _ = _forward(layers, batch, 2)
    assert net2.Bf16RecurrenceLayers() == 0
    plain = bf.forward(layers, batch.feats, batch.lens, T, S, bf16_gemm=True, bf16_rec=False, teacher=out2)
    assert rel_err(out2[vm], plain[vm]) < 2e-5


@pytest.mark.parametrize("H,S,T,proj", [(256, 32, 40, 128), (1024, 32, 10, 256)])
def test_bf16_forward_whole_net_against_the_restatement(gpu, H, S, T, proj):
    """The free-running stack (two layers, projection, softmax) against the free-running restatement.  Here single bf16 rounding
    flips -- an operand 1e-7 apart in the two implementations that falls on the other side of a rounding boundary, in a GEMM
    operand or in the recurrence -- DO travel down the chain, so the bar is what a handful of 2^-9 perturbations can do (measured
    6e-4 .. 1.5e-3 of the largest softmax output; bar 4e-3); the step-by-step test above is the sharp one.  The GEMM-only mode
    still sits nearer its own arbiter than the other mode's."""
    cfg = dict(kind="BiLstmParallel", layers=2, H=H, D=40, K=30, S=S, T=T, proj=proj, seed=4321 + H)
    layers = synth.make_model(**cfg)
    batch = synth.make_batch(**cfg)
    vm = valid_mask(batch.lens, batch.T, batch.S)
    ora = {1: bf.forward(layers, batch.feats, batch.lens, T, S, bf16_gemm=True, bf16_rec=True),
           2: bf.forward(layers, batch.feats, batch.lens, T, S, bf16_gemm=True, bf16_rec=False)}
    err = {}
    for mode in (1, 2):
        net, out, ri = _forward(layers, batch, mode)
        assert ri["fwd_persistent"] == ri["lstm_layers"] == 2, ri
        assert net.Bf16RecurrenceLayers() == (2 if mode == 1 else 0)
        err[mode] = {k: rel_err(out[vm], ora[k][vm]) for k in (1, 2)}
    assert err[1][1] < 4e-3 and err[2][2] < 4e-3, err
    assert err[2][2] < err[2][1], err


def test_fp32_backward_on_the_bf16_forward(gpu):
    """The backward pass of the variant is the fp32 one, fed with the activations the bf16 forward stored (G, C, Y in fp32): its
    gradients sit within eight-bit-operand distance of the all-fp32 step, and the fp32 path itself is untouched by the option
    (a Net switched back to mode 0 reproduces a never-switched Net bit for bit)."""
    from eesen_amd.api import Net, Ctc
    cfg = dict(kind="BiLstmParallel", layers=2, H=256, D=40, K=30, S=32, T=60, proj=128)
    layers = synth.make_model(**cfg)
    batch = synth.make_batch(**cfg)

    def step(net):
        ctc = Ctc()
        net.SetSeqLengths(batch.lens)
        out = net.Propagate(batch.feats)
        diff = ctc.EvalParallel(batch.lens, out, batch.labels)
        net.BackpropagateNoUpdate(diff)
        net.Synchronize()
        return out.numpy(), ctc.pzx.copy(), net.GetGrads()

    plain = Net.from_layers(layers); plain.SetTrainOptions(1.0, 0.0)
    r0 = step(plain)
    net = Net.from_layers(layers); net.SetTrainOptions(1.0, 0.0)
    net.SetForwardPrecision(1)
    r1 = step(net)
    assert net.Bf16RecurrenceLayers() == 2
    net.SetForwardPrecision(0)
    r2 = step(net)
    assert net.Bf16RecurrenceLayers() == 0
    assert np.array_equal(r2[0], r0[0]) and np.array_equal(r2[1], r0[1])
    assert np.all(np.isfinite(r1[2]))
    assert rel_err(r1[1], r0[1]) < 1e-2 and rel_err(r1[2], r0[2]) < 0.15 and not np.array_equal(r1[0], r0[0])
