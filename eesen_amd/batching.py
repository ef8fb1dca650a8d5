"""Minibatch assembly of `train-ctc-parallel` (/root/reference/src/netbin/train-ctc-parallel.cc:144-193), row a1 of the
hot-path table: greedy grouping of up to --num-sequence utterances whose padded size stays within --frame-limit, then
zero-padding to the group's longest utterance and time-major interleaving (row t*S + s = frame t of sequence s)."""
from __future__ import annotations

import sys
from dataclasses import dataclass, field
from typing import Dict, Iterable, Iterator, List, Tuple

import numpy as np


@dataclass
class Minibatch:
    feats: np.ndarray            # [T*S x D] float32, time-major interleaved, zero beyond each utterance's length
                                 # (None when assembled with interleaved=False: the device Feeder builds it from `mats`)
    lens: np.ndarray             # frame_num_utt, int32 [S]
    labels: List[np.ndarray]     # S int32 label vectors
    keys: List[str]
    T: int
    S: int
    mats: List[np.ndarray] = None    # the S utterance matrices [T_s x D] (kept only with interleaved=False)


MAX_LABELS = 2047   # include/eesen_hip.h, eesen_ctc_eval_parallel: lattices of up to 4096 positions (511 / 1024 up to round 4)


@dataclass
class AssemblyStats:
    num_no_tgt_mat: int = 0      # utterances without targets (train-ctc-parallel.cc:152-156)
    num_too_long: int = 0        # utterances above the frame limit (:161-164)
    num_other_error: int = 0     # utterances the CTC cannot take: empty transcript (the reference reads alpha column -1 there,
                                 # ctc-loss.cc:151) or more than 2047 labels (expanded length above the 4096 lattice positions a sweep holds)
    warnings: List[str] = field(default_factory=list)


def interleave(mats: List[np.ndarray], feat_dim: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """train-ctc-parallel.cc:186-193: feat_mat_host(cur_sequence_num * max_frame_num, feat_dim, kSetZero), row r*S + s."""
    S = len(mats)
    lens = np.array([m.shape[0] for m in mats], np.int32)
    T = int(lens.max()) if S else 0
    out = np.zeros((T, S, feat_dim), np.float32)
    for s, m in enumerate(mats):
        if m.shape[1] != feat_dim:
            raise ValueError(f"feature dimension {m.shape[1]} does not match the net's InputDim {feat_dim}")
        out[: m.shape[0], s, :] = m
    return out.reshape(T * S, feat_dim), lens, T


def assemble(features: Iterable[Tuple[str, np.ndarray]], targets: Dict[str, np.ndarray], num_sequence: int, frame_limit: float,
             feat_dim: int, stats: AssemblyStats = None, interleaved: bool = True) -> Iterator[Minibatch]:
    """The while(1) loop of train-ctc-parallel.cc:144-183.  An utterance that does not fit the current group
    (new_max_len * (n + 1) > frame_limit) opens the next group (:170-172, the reader is not advanced); a full group
    (n == num_sequence) advances the reader and closes (:179-182)."""
    stats = stats if stats is not None else AssemblyStats()
    it = iter(features)
    pending = None
    done = False
    while not done:
        mats, labs, keys = [], [], []
        max_frame_num = 0
        while True:
            if pending is None:
                try:
                    pending = next(it)
                except StopIteration:
                    done = True
                    break
            utt, mat = pending
            if utt not in targets:
                stats.num_no_tgt_mat += 1
                stats.warnings.append(f"{utt}, missing targets")
                pending = None
                continue
            n_lab = len(targets[utt])
            if n_lab == 0 or n_lab > MAX_LABELS:
                stats.num_other_error += 1
                stats.warnings.append(f"{utt}, {'empty transcript' if n_lab == 0 else f'{n_lab} labels exceed the {MAX_LABELS} a lattice sweep holds'}; ignoring")
                pending = None
                continue
            if mat.shape[0] > frame_limit:
                stats.num_too_long += 1
                stats.warnings.append(f"{utt}, has too many frames; ignoring: {mat.shape[0]} > {frame_limit:g}")
                pending = None
                continue
            new_max = max(max_frame_num, mat.shape[0])
            if new_max * (len(mats) + 1) > frame_limit:
                break                                   # does not fit: keep `pending` for the next group
            max_frame_num = new_max
            mats.append(mat); labs.append(np.asarray(targets[utt], np.int32)); keys.append(utt)
            pending = None
            if len(mats) == num_sequence:
                break
        if mats and interleaved:
            feats, lens, T = interleave(mats, feat_dim)
            yield Minibatch(feats=feats, lens=lens, labels=labs, keys=keys, T=T, S=len(mats))
        elif mats:  # padding + interleave happen on the device (eesen_amd.api.Feeder)
            for m in mats:
                if m.shape[1] != feat_dim:
                    raise ValueError(f"feature dimension {m.shape[1]} does not match the net's InputDim {feat_dim}")
            lens = np.array([m.shape[0] for m in mats], np.int32)
            yield Minibatch(feats=None, lens=lens, labels=labs, keys=keys, T=int(lens.max()), S=len(mats), mats=mats)
        elif not done and pending is not None:
            # a single utterance within frame_limit always fits an empty group, so this cannot loop
            raise RuntimeError("batch assembly made no progress")
