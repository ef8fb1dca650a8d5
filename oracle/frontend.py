"""CPU restatement of the reference's feature filters that sit in front of the trainer.  TEST INFRASTRUCTURE: only tests/
may import this; the product runs these stages on the device (eesen_amd/csrc/feeder.hip, eesen_feeder_set_pipeline).

Restated, function by function, from
  ApplyCmvn            /root/reference/src/feat/cmvn.cc:64-118          (featbin/apply-cmvn.cc)
  SpliceFrames         /root/reference/src/feat/feature-functions.cc:391-412 (featbin/splice-feats.cc)
  subsample-feats      /root/reference/src/featbin/subsample-feats.cc:77-108
  DeltaFeatures        /root/reference/src/feat/feature-functions.cc:210-267 (ComputeDeltas :319-330, featbin/add-deltas.cc)
in fp32, every multiply and add rounded separately (numpy float32 arithmetic never fuses them).  The reference's
`output.AddVec(scale, row)` goes through BLAS saxpy, whose rounding (fused or not) depends on the BLAS build; the other three
are exact copies / a scalar `a + x * b`, so they are bit-exact against the reference's binaries and the deltas agree to an ulp
or two (tests/test_frontend.py pins all four against oracle/_ref/featbin where /root/reference exists, and against
tests/golden/frontend.npz everywhere).
"""
from __future__ import annotations

import numpy as np

F = np.float32


def cmvn_norm(stats: np.ndarray, norm_vars: bool) -> np.ndarray:
    """cmvn.cc:78-108: stats [1|2 x dim+1] doubles -> [2 x dim] floats (offset row, scale row)."""
    stats = np.asarray(stats, np.float64)
    dim = stats.shape[1] - 1
    if stats.shape[0] not in (1, 2):
        raise ValueError("Dim mismatch in ApplyCmvn")
    if stats.shape[0] == 1 and norm_vars:
        raise ValueError("You requested variance normalization but no variance stats are supplied.")
    count = stats[0, dim]
    if count < 1.0:
        raise ValueError("Insufficient stats for cepstral mean and variance normalization")
    norm = np.zeros((2, dim), F)
    for d in range(dim):
        mean = stats[0, d] / count
        if not norm_vars:
            scale, offset = 1.0, -mean
        else:
            var = stats[1, d] / count - mean * mean
            if var < 1.0e-20:
                var = 1.0e-20
            scale = 1.0 / np.sqrt(var)
            offset = -(mean * scale)
        norm[0, d] = offset
        norm[1, d] = scale
    return norm


def apply_cmvn(stats: np.ndarray, norm_vars: bool, feats: np.ndarray) -> np.ndarray:
    """cmvn.cc:110-117: f = norm(0, d) + f * norm(1, d), in float."""
    norm = cmvn_norm(stats, norm_vars)
    feats = np.asarray(feats, F)
    if feats.shape[1] != norm.shape[1]:
        raise ValueError("Dim mismatch in ApplyCmvn")
    return (norm[0][None, :] + (feats * norm[1][None, :]).astype(F)).astype(F)


def splice_frames(feats: np.ndarray, left: int, right: int) -> np.ndarray:
    """feature-functions.cc:395-411."""
    feats = np.asarray(feats, F)
    T, D = feats.shape
    if T == 0 or D == 0:
        raise ValueError("SpliceFrames: empty input")
    N = 1 + left + right
    out = np.empty((T, D * N), F)
    for j in range(N):
        t2 = np.clip(np.arange(T) + j - left, 0, T - 1)
        out[:, j * D:(j + 1) * D] = feats[t2]
    return out


def subsample(feats: np.ndarray, n: int, offset: int = 0):
    """subsample-feats.cc:77-108.  Returns None where the tool writes no output (no frame survives)."""
    feats = np.asarray(feats, F)
    assert n != 0
    if n > 0:
        idx = np.arange(offset, feats.shape[0], n)
        if idx.size == 0:
            return None
        return feats[idx].copy()
    assert offset == 0
    return feats[np.arange(feats.shape[0] * -n) // -n].copy()


def delta_scales(order: int, window: int):
    """DeltaFeatures::DeltaFeatures, feature-functions.cc:216-241: float accumulation, Scale(float(1.0 / normalizer))."""
    scales = [np.array([1.0], F)]
    for i in range(1, order + 1):
        prev = scales[i - 1]
        prev_offset = (prev.size - 1) // 2
        cur_offset = prev_offset + window
        cur = np.zeros(prev.size + 2 * window, F)
        normalizer = F(0.0)
        for j in range(-window, window + 1):
            normalizer = F(normalizer + F(j * j))
            for k in range(-prev_offset, prev_offset + 1):
                cur[j + k + cur_offset] = F(cur[j + k + cur_offset] + F(F(j) * prev[k + prev_offset]))
        alpha = F(1.0 / np.float64(normalizer))
        scales.append((cur * alpha).astype(F))
    return scales


def add_deltas(feats: np.ndarray, order: int = 2, window: int = 2) -> np.ndarray:
    """DeltaFeatures::Process, feature-functions.cc:252-266 for every frame (ComputeDeltas :319-330)."""
    feats = np.asarray(feats, F)
    T, D = feats.shape
    scales = delta_scales(order, window)
    out = np.zeros((T, D * (order + 1)), F)
    for i, sc in enumerate(scales):
        max_offset = (sc.size - 1) // 2
        acc = np.zeros((T, D), F)
        for j in range(-max_offset, max_offset + 1):
            w = sc[j + max_offset]
            if w != 0.0:
                rows = feats[np.clip(np.arange(T) + j, 0, T - 1)]
                acc = (acc + (w * rows).astype(F)).astype(F)
        out[:, i * D:(i + 1) * D] = acc
    return out


def run_pipeline(stages, feats: np.ndarray, stats=None):
    """stages: [("cmvn", norm_vars), ("splice", L, R), ("subsample", n, offset), ("deltas", order, window)] in order."""
    x = np.asarray(feats, F)
    for st in stages:
        if st[0] == "cmvn":
            x = apply_cmvn(stats, bool(st[1]), x)
        elif st[0] == "splice":
            x = splice_frames(x, st[1], st[2])
        elif st[0] == "subsample":
            x = subsample(x, st[1], st[2])
            if x is None:
                return None
        elif st[0] == "deltas":
            x = add_deltas(x, st[1], st[2])
        else:
            raise ValueError(st[0])
    return x
