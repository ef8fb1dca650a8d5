"""Host side of the device feature front end (include/eesen_hip.h `eesen_feeder_set_pipeline`).

The recipes hand `train-ctc-parallel` / `net-output-extract` their features as an rspecifier that is a pipe of host filters
(/root/reference/asr_egs/wsj/steps/train_ctc_parallel.sh:95-110, decode_ctc_lat.sh:92-95,
librispeech/steps/train_ctc_parallel_mult.sh:110-133), e.g.

    ark,s,cs:apply-cmvn --norm-vars=true --utt2spk=ark:data/utt2spk scp:data/cmvn.scp scp:exp/train.scp ark:- | \
             splice-feats --left-context=1 --right-context=1 ark:- ark:- | subsample-feats --n=3 --offset=0 ark:- ark:- | \
             add-deltas ark:- ark:- |

`parse_feature_pipeline` recognises exactly such command lines (the option sets of the reference's featbin tools): the
trainers then read the RAW table themselves (here `scp:exp/train.scp`), look the CMVN statistics up per utterance, and
the filters run on the GPU inside the batch assembly -- same command line, no filter processes.  Anything it does not
recognise (another tool, an option it does not implement such as --skip-dims) returns None and the rspecifier is opened
as the pipe it is.
"""
from __future__ import annotations

import os
import shlex
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

from . import kaldi_io

CMVN, SPLICE, SUBSAMPLE, DELTAS = 1, 2, 3, 4    # EESEN_FEAT_* of include/eesen_hip.h


def _bool(v: str) -> Optional[bool]:
    """ParseOptions::ToBool (/root/reference/src/util/parse-options.cc): true/t/1 and false/f/0, and the empty string = true."""
    v = v.strip().lower()
    if v in ("", "true", "t", "1"):
        return True
    if v in ("false", "f", "0"):
        return False
    return None


def _options(argv: List[str], known: Dict[str, type]):
    """Splits `--name=value` options from positional arguments; None if an option is unknown or malformed."""
    opts, pos = {}, []
    for a in argv:
        if a.startswith("--"):
            name, _, val = a[2:].partition("=")
            name = name.replace("_", "-")
            if name not in known:
                return None
            if known[name] is bool:
                b = _bool(val)
                if b is None:
                    return None
                opts[name] = b
            elif known[name] is int:
                try:
                    opts[name] = int(val)
                except ValueError:
                    return None
            else:
                opts[name] = val
        else:
            pos.append(a)
    return opts, pos


@dataclass
class FeaturePipeline:
    source: str                                   # rspecifier of the raw features
    stages: List[Tuple[int, int, int]] = field(default_factory=list)   # (EESEN_FEAT_*, a, b) in order
    cmvn: Optional[str] = None                    # stats rspecifier (per utterance / speaker) or rxfilename (global)
    utt2spk: Optional[str] = None
    norm_vars: bool = False

    def out_dim(self, D: int) -> int:
        for k, a, b in self.stages:
            if k == SPLICE:
                D *= 1 + a + b
            elif k == DELTAS:
                D *= 1 + a
        return D

    def out_frames(self, T: int) -> int:
        for k, a, b in self.stages:
            if k == SUBSAMPLE:
                T = T * -a if a < 0 else (max(0, (T - b + a - 1) // a) if T > b else 0)
        return T


def parse_feature_pipeline(rspecifier: str) -> Optional[FeaturePipeline]:
    head, sep, cmd = rspecifier.partition(":")
    if not sep or head.split(",")[0] != "ark" or not cmd.rstrip().endswith("|"):
        return None
    segs = [s.strip() for s in cmd.rstrip().rstrip("|").split("|")]
    if not segs or any(not s for s in segs):
        return None
    pipe = None
    for i, seg in enumerate(segs):
        try:
            argv = shlex.split(seg)
        except ValueError:
            return None
        tool, argv = os.path.basename(argv[0]), argv[1:]
        if i == 0:
            if tool == "apply-cmvn":        # featbin/apply-cmvn.cc:36-47
                r = _options(argv, {"utt2spk": str, "norm-vars": bool, "norm-means": bool})
                if r is None or len(r[1]) != 3 or r[1][2] != "ark:-":
                    return None
                o, (stats, src, _) = r
                norm_means, norm_vars = o.get("norm-means", True), o.get("norm-vars", False)
                if norm_vars and not norm_means:
                    return None             # the tool itself refuses this (:55-56): let it say so
                pipe = FeaturePipeline(source=src)
                if norm_means:
                    pipe.cmvn, pipe.utt2spk, pipe.norm_vars = stats, o.get("utt2spk") or None, norm_vars
                    pipe.stages.append((CMVN, int(norm_vars), 0))
            elif tool == "copy-feats":
                r = _options(argv, {})
                if r is None or len(r[1]) != 2 or r[1][1] != "ark:-":
                    return None
                pipe = FeaturePipeline(source=r[1][0])
            else:
                return None
            if pipe.source.partition(":")[0].split(",")[0] not in ("ark", "scp") or pipe.source.rstrip().endswith("|"):
                return None
            continue
        if tool == "splice-feats":          # featbin/splice-feats.cc:36-40: both contexts default to 4
            r = _options(argv, {"left-context": int, "right-context": int})
            if r is None or r[1] != ["ark:-", "ark:-"]:
                return None
            L, R = r[0].get("left-context", 4), r[0].get("right-context", 4)
            if L < 0 or R < 0:
                return None
            pipe.stages.append((SPLICE, L, R))
        elif tool == "subsample-feats":     # featbin/subsample-feats.cc:46-55
            r = _options(argv, {"n": int, "offset": int})
            if r is None or r[1] != ["ark:-", "ark:-"]:
                return None
            n, off = r[0].get("n", 1), r[0].get("offset", 0)
            if n == 0 or off < 0 or (n < 0 and off != 0):
                return None
            pipe.stages.append((SUBSAMPLE, n, off))
        elif tool == "add-deltas":          # featbin/add-deltas.cc:33-38, DeltaFeaturesOptions feature-functions.h: order 2, window 2
            r = _options(argv, {"delta-order": int, "delta-window": int, "truncate": int})
            if r is None or r[1] != ["ark:-", "ark:-"] or r[0].get("truncate", 0) != 0:
                return None
            order, window = r[0].get("delta-order", 2), r[0].get("delta-window", 2)
            if not (0 <= order <= 8 and 0 < window <= 16):
                return None
            pipe.stages.append((DELTAS, order, window))
        else:
            return None
    return pipe


def cmvn_norm(stats: np.ndarray, norm_vars: bool) -> np.ndarray:
    """[2 x dim] float32 (offsets, scales) from a CMVN statistics matrix: eesen_cmvn_norm, i.e. ApplyCmvn's own arithmetic
    (/root/reference/src/feat/cmvn.cc:78-108)."""
    import ctypes as C
    from . import _lib
    stats = np.ascontiguousarray(stats, np.float64)
    if stats.ndim != 2:
        raise _lib.EesenError(-1, "CMVN statistics must be a matrix")
    out = np.empty((2, max(stats.shape[1] - 1, 0)), np.float32)
    _lib.check(_lib.load().eesen_cmvn_norm(stats.ctypes.data_as(C.POINTER(C.c_double)), stats.shape[0], stats.shape[1], int(norm_vars),
                                           out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


class RawUtt:
    """An utterance as the feeder takes it with a pipeline set: the raw matrix, its CMVN vectors, and -- what batch
    assembly and the CTC see -- the shape BEHIND the pipeline."""
    __slots__ = ("raw", "cmvn", "shape")

    def __init__(self, raw: np.ndarray, cmvn: Optional[np.ndarray], shape: Tuple[int, int]):
        self.raw, self.cmvn, self.shape = raw, cmvn, shape


class CmvnTable:
    """RandomAccessDoubleMatrixReaderMapped(cmvn_rspecifier, utt2spk_rspecifier) of apply-cmvn.cc:80-81, or the single
    matrix of its rxfilename form (:115-122); normalisers are computed once per speaker."""

    def __init__(self, spec: str, utt2spk: Optional[str], norm_vars: bool):
        self.norm_vars = norm_vars
        self.global_norm = None
        self.map = None
        self.cache: Dict[str, np.ndarray] = {}
        if spec.partition(":")[0].split(",")[0] in ("ark", "scp") and ":" in spec:
            self.stats = dict(kaldi_io.read_mat64_table(spec))
            if utt2spk:
                self.map = kaldi_io.read_token_map(utt2spk)
        else:
            if utt2spk:
                raise kaldi_io.KaldiIOError("--utt2spk option not compatible with rxfilename as input (did you forget ark:?)")
            self.global_norm = cmvn_norm(kaldi_io.read_mat64_file(spec), norm_vars)

    def lookup(self, utt: str) -> Optional[np.ndarray]:
        if self.global_norm is not None:
            return self.global_norm
        key = utt
        if self.map is not None:
            if utt not in self.map:
                return None
            key = self.map[utt]
        if key not in self.stats:
            return None
        if key not in self.cache:
            self.cache[key] = cmvn_norm(self.stats[key], self.norm_vars)
        return self.cache[key]


def read_raw(pipe: FeaturePipeline, source: Optional[str] = None, warn=None) -> Iterator[Tuple[str, RawUtt]]:
    """The raw table as (key, RawUtt).  Utterances the reference's filters would not have passed on are dropped the same
    way: no CMVN statistics (apply-cmvn.cc:87-92), no rows (add-deltas.cc:55-58), no frame left by the subsampling
    (subsample-feats.cc:87-92)."""
    warn = warn or (lambda msg: None)
    table = CmvnTable(pipe.cmvn, pipe.utt2spk, pipe.norm_vars) if pipe.cmvn else None
    for key, mat in kaldi_io.read_mat_table(source or pipe.source):
        vec = None
        if table is not None:
            vec = table.lookup(key)
            if vec is None:
                warn(f"No normalization statistics available for key {key}, producing no output for this utterance")
                continue
            if vec.shape[1] != mat.shape[1]:
                raise kaldi_io.KaldiIOError(f"Dim mismatch in ApplyCmvn: cmvn 2x{vec.shape[1] + 1}, feats {mat.shape[0]}x{mat.shape[1]}")
        if mat.shape[0] == 0:
            warn(f"Empty feature matrix for key {key}")
            continue
        frames = pipe.out_frames(mat.shape[0])
        if frames == 0:
            warn(f"For utterance {key}, output would have no rows, producing no output.")
            continue
        yield key, RawUtt(mat, vec, (frames, pipe.out_dim(mat.shape[1])))
