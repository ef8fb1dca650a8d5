"""numpy restatement of Net::Propagate with the roundings of BASELINE config 4's "bf16 forward" variant.  TEST INFRASTRUCTURE.

The reference has no bf16 arithmetic (BaseFloat = float, /root/reference/src/base/kaldi-types.h:26-30), so the variant's arbiter
is the reference's OWN forward equations with the variant's roundings applied at exactly the places include/eesen_hip.h
(eesen_net_set_forward_precision) names:
  * (gemm)  every forward GEMM -- x W_x^T of an LSTM layer (bilstm-parallel-layer.h:109-110,163-164), an <AffineTransform>
            (affine-trans-layer.h:161-166) -- rounds BOTH operands to nearest-even bf16, products and sums in fp32 or wider;
  * (rec)   the recurrent product m_{t-1} W_m^T (bilstm-parallel-layer.h:116-118,171-173) rounds m_{t-1} to ONE bf16 value and holds
            W_m as `w_planes` bf16 planes (round to nearest even, round the exact remainder again: 2 planes = 17 significant bits,
            the library's default; 1 = the single-plane arm); the gate pre-activations, the cell state and the activations stay fp32.
With both switched off this is the plain forward pass and must equal the C oracle / the golden fixtures (tests/test_bf16_forward_oracle.py
pins it), which is what makes the roundings the only difference.  Cell equations: bilstm-parallel-layer.h:120-148 (g = tanh, i / f with
the peepholes on c_{t-1}, c = g i + c_{t-1} f, o with the peephole on c_t, m = tanh(c) o).  Padding frames (t >= len_s) carry zero state in
both directions, as in the HIP library.  Sums are formed in fp64: the order of an fp32 sum is not part of the
statement, and at the variant's tolerance (1e-4 and up) it does not matter.
"""
from __future__ import annotations

import numpy as np


def round_bf16(a):
    """fp32 -> nearest-even bf16 -> fp32 (the bit trick of gemm.hip: rne_bf16_bits; no NaN handling needed here)."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).astype(np.uint32)
    return r.view(np.float32).reshape(np.shape(a))


def bf16_planes(a, planes):
    """The value `planes` successive bf16 roundings keep of a (each plane rounds the exact remainder of the ones before): what
    lstm_fwd_persistent_bf_kernel multiplies with when it holds an operand as that many planes."""
    a = np.ascontiguousarray(a, np.float32)
    kept = np.zeros_like(a)
    rem = a.copy()
    for _ in range(planes):
        p = round_bf16(rem)
        kept = kept + p          # exact: the planes do not overlap
        rem = rem - p
    return kept


def _mm(a, b_t, bf16):
    """a [n x k] times b_t [m x k] transposed, fp64 accumulation; operands rounded to bf16 on request."""
    if bf16:
        a, b_t = round_bf16(a), round_bf16(b_t)
    return (np.asarray(a, np.float64) @ np.asarray(b_t, np.float64).T).astype(np.float32)


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x.astype(np.float64)))


def _lstm_direction(gx, Wm, peep, lens, T, S, H, reverse, bf16_rec, teacher=None, w_planes=2):
    """gx [T, S, 4H] = x W_x^T + bias in the FILE's gate order g | i | f | o; returns m [T, S, H] (zero on padding).
    teacher [T, S, H]: another implementation's m; when given, step t takes ITS m of the previous step as the recurrent input
    (the cell state is still carried here), so that a comparison of the two outputs is one step deep everywhere: an m that
    falls on the other side of a bf16 rounding boundary in one of the two cannot amplify through the chain.
    All sequences advance together (step n of sequence s is frame n, or len_s - 1 - n in the reverse direction)."""
    p_i, p_f, p_o = [np.asarray(p, np.float64) for p in peep]
    lens = np.asarray(lens, np.int64)
    m_out = np.zeros((T, S, H), np.float32)
    Wt = (bf16_planes(Wm, w_planes) if bf16_rec else np.asarray(Wm, np.float32)).astype(np.float64).T      # [H x 4H]
    c = np.zeros((S, H), np.float64)
    m = np.zeros((S, H), np.float32)
    sidx = np.arange(S)
    for n in range(int(lens.max())):
        act = n < lens
        t = np.where(reverse, lens - 1 - n, n)
        t = np.where(act, t, 0)
        if teacher is not None and n > 0:
            tp = np.where(act, t + 1 if reverse else t - 1, 0)
            m = np.asarray(teacher[tp, sidx], np.float32)
        mp = round_bf16(m) if bf16_rec else m
        pre = gx[t, sidx].astype(np.float64) + (mp.astype(np.float64) @ Wt).astype(np.float32)
        g = np.tanh(pre[:, 0:H])
        i = _sig((pre[:, H:2 * H] + p_i * c).astype(np.float32))
        f = _sig((pre[:, 2 * H:3 * H] + p_f * c).astype(np.float32))
        cn = g * i + c * f
        o = _sig((pre[:, 3 * H:4 * H] + p_o * cn).astype(np.float32))
        mn = (np.tanh(cn) * o).astype(np.float32)
        c = np.where(act[:, None], cn, c)
        m = np.where(act[:, None], mn, m)
        m_out[t[act], sidx[act]] = mn[act]
    return m_out


def forward(layers, feats, lens, T, S, bf16_gemm=False, bf16_rec=False, teacher=None, w_planes=2):
    """layers: eesen_amd.synth.make_model / nnet_io.read_nnet dicts (parameters in the file's order); feats [T*S x D] time-major
    interleaved.  Returns net_out [T*S x K].  teacher: for a net that is ONE (Bi)LSTM layer, another implementation's output
    [T*S x ndir*H] to take the recurrent inputs from (see _lstm_direction)."""
    x = np.asarray(feats, np.float32)
    assert teacher is None or len(layers) == 1
    for L in layers:
        t = L["type"]
        if t in ("BiLstmParallel", "LstmParallel"):
            nd = 2 if t == "BiLstmParallel" else 1
            H = L["output_dim"] // nd
            outs = []
            for d in range(nd):
                Wx, Wm, bias, pi, pf, po = L["params"][6 * d: 6 * d + 6]
                gx = (_mm(x, Wx, bf16_gemm) + np.asarray(bias, np.float32)).reshape(T, S, 4 * H)
                tch = None if teacher is None else np.asarray(teacher, np.float32).reshape(T, S, nd * H)[:, :, d * H:(d + 1) * H]
                outs.append(_lstm_direction(gx, Wm, (pi, pf, po), lens, T, S, H, d == 1, bf16_rec, tch, w_planes))
            x = np.concatenate(outs, axis=2).reshape(T * S, nd * H)
        elif t == "AffineTransform":
            W, b = L["params"]
            x = _mm(x, W, bf16_gemm) + np.asarray(b, np.float32)
        elif t == "Softmax":
            z = x.astype(np.float64)
            z = np.exp(z - z.max(axis=1, keepdims=True))
            x = (z / z.sum(axis=1, keepdims=True)).astype(np.float32)
        elif t == "Sigmoid":
            x = _sig(x).astype(np.float32)
        elif t == "Tanh":
            x = np.tanh(x.astype(np.float64)).astype(np.float32)
        else:
            raise ValueError(t)
    return x
