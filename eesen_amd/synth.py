"""Seeded synthetic utterance batches and model files for the BASELINE.json configurations.

Follows SURVEY.md §8(d): N(0,1) fp32 features (CMVN'd fbank surrogate), lengths U{0.8 T_max .. T_max} sorted
ascending with the last = T_max (recipes sort by length, /root/reference/asr_egs/wsj/steps/train_ctc_parallel.sh:84-89),
U_s = floor(T_s/10) labels uniform in 1..K-1 with ~10 % forced adjacent repeats, weights U(-0.1, 0.1)
(/root/reference/asr_egs/wsj/utils/model_topo.py:80), seed 777 (/root/reference/src/netbin/net-initialize.cc:41).
Batch layout is the reference trainer's: time-major interleaved rows t*S+s, zero beyond len_s
(/root/reference/src/netbin/train-ctc-parallel.cc:187-193).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np

from . import nnet_io

CONFIGS = {
    # name: layers spec (type, H per direction), D, K, S, T
    "cfg1": dict(kind="LstmParallel", layers=1, H=128, D=40, K=31, S=8, T=200),
    "cfg2": dict(kind="BiLstmParallel", layers=4, H=512, D=40, K=46, S=32, T=1000),
    "cfg4": dict(kind="BiLstmParallel", layers=5, H=1024, D=40, K=51, S=32, T=1000, proj=512),
    "cfg5": dict(kind="BiLstmParallel", layers=6, H=1024, D=40, K=51, S=64, T=3000),
    # small shapes for parity tests
    "tiny_bi": dict(kind="BiLstmParallel", layers=2, H=16, D=8, K=7, S=3, T=12),
    "small_bi": dict(kind="BiLstmParallel", layers=2, H=64, D=40, K=46, S=8, T=60),
    "small_uni": dict(kind="LstmParallel", layers=1, H=32, D=40, K=31, S=4, T=50),
}


@dataclass
class Batch:
    feats: np.ndarray      # [T*S, D] float32, row t*S+s
    lens: np.ndarray       # [S] int32 (frame_num_utt)
    labels: List[np.ndarray]  # S arrays of int32 label ids (no blanks)
    T: int
    S: int

    @property
    def label_ids(self) -> np.ndarray:
        return np.concatenate(self.labels).astype(np.int32) if self.labels else np.zeros(0, np.int32)

    @property
    def label_off(self) -> np.ndarray:
        return np.concatenate([[0], np.cumsum([len(l) for l in self.labels])]).astype(np.int32)

    @property
    def real_frames(self) -> int:
        return int(self.lens.sum())


def make_model(kind: str, layers: int, H: int, D: int, K: int, seed: int = 777, max_grad: float = 0.0,
               learn_rate_coef: float = 1.0, proj: int = 0, param_range: float = 0.1, proj_act=None, **_) -> List[dict]:
    """proj_act: None, "Sigmoid", "Tanh" or a list cycled over the projections: an activation layer after each projection."""
    rng = np.random.default_rng(seed)
    u = lambda *sh: rng.uniform(-param_range, param_range, size=sh).astype(np.float32)
    out, din = [], D
    ndir = 2 if kind.startswith("BiLstm") else 1
    for li in range(layers):
        params = []
        for _ in range(ndir):
            params += [u(4 * H, din), u(4 * H, H), u(4 * H), u(H), u(H), u(H)]
        out.append(dict(type=kind, input_dim=din, output_dim=ndir * H, learn_rate_coef=learn_rate_coef,
                        max_grad=max_grad, params=params))
        din = ndir * H
        if proj and li < layers - 1:
            out.append(dict(type="AffineTransform", input_dim=din, output_dim=proj, learn_rate_coef=learn_rate_coef,
                            max_grad=max_grad, params=[u(proj, din), u(proj)]))
            din = proj
            if proj_act:
                acts = [proj_act] if isinstance(proj_act, str) else list(proj_act)
                out.append(dict(type=acts[li % len(acts)], input_dim=din, output_dim=din, params=[]))
    out.append(dict(type="AffineTransform", input_dim=din, output_dim=K, learn_rate_coef=learn_rate_coef,
                    max_grad=max_grad, params=[u(K, din), u(K)]))
    out.append(dict(type="Softmax", input_dim=K, output_dim=K, params=[]))
    return out


def make_batch(S: int, T: int, D: int, K: int, seed: int = 777, min_frac: float = 0.8, label_div: int = 10,
               repeat_frac: float = 0.1, **_) -> Batch:
    rng = np.random.default_rng(seed + 1)
    lens = np.sort(rng.integers(int(np.ceil(min_frac * T)), T + 1, size=S)).astype(np.int32)
    lens[-1] = T
    feats = np.zeros((T, S, D), np.float32)
    labels = []
    for s in range(S):
        feats[: lens[s], s, :] = rng.standard_normal((lens[s], D)).astype(np.float32)
        U = max(1, int(lens[s]) // label_div)
        lab = rng.integers(1, K, size=U).astype(np.int32)
        rep = rng.random(U) < repeat_frac
        for i in range(1, U):
            if rep[i]: lab[i] = lab[i - 1]
        labels.append(lab)
    return Batch(feats=feats.reshape(T * S, D), lens=lens, labels=labels, T=T, S=S)


def config(name: str) -> dict:
    return dict(CONFIGS[name])


def write_model(path: str, layers: List[dict], binary: bool = False):
    nnet_io.write_nnet(path, layers, binary=binary)
