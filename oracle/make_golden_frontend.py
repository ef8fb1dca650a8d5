"""Generates tests/golden/frontend.npz: the outputs of THE REFERENCE'S OWN feature tools (oracle/_ref/featbin, compiled unmodified
from /root/reference/src/featbin by oracle/ref_build/Makefile target `featbin`) on a small seeded table, for the pipelines the
recipes use.  TEST INFRASTRUCTURE; runs only where /root/reference exists.

    python -m oracle.make_golden_frontend
"""
from __future__ import annotations

import os
import subprocess
import tempfile

import numpy as np

from eesen_amd import kaldi_io

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "_ref", "featbin")
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden", "frontend.npz")

# name -> (reference command line behind `apply-cmvn ... ark:- |`, stages as oracle/frontend.py takes them)
PIPELINES = {
    "wsj_train": ("add-deltas ark:- ark:- |", True, [("cmvn", True), ("deltas", 2, 2)]),                         # train_ctc_parallel.sh:95-110
    "wsj_x3": ("splice-feats --left-context=1 --right-context=1 ark:- ark:- | subsample-feats --n=3 --offset=1 ark:- ark:- |",
               True, [("cmvn", True), ("splice", 1, 1), ("subsample", 3, 1)]),                                   # train_ctc_parallel_x3.sh:112-135
    "wsj_decode": ("splice-feats --left-context=1 --right-context=1 ark:- ark:- | subsample-feats --n=2 --offset=0 ark:- ark:- | add-deltas ark:- ark:- |",
                   False, [("cmvn", False), ("splice", 1, 1), ("subsample", 2, 0), ("deltas", 2, 2)]),           # decode_ctc_lat.sh:92-95
    "libri_mult": ("splice-feats --left-context=2 --right-context=1 ark:- ark:- | add-deltas --delta-order=1 --delta-window=3 ark:- ark:- | "
                   "subsample-feats --n=3 --offset=2 ark:- ark:- |",
                   True, [("cmvn", True), ("splice", 2, 1), ("deltas", 1, 3), ("subsample", 3, 2)]),             # train_ctc_parallel_mult.sh:121-133
    "repeat": ("subsample-feats --n=-2 ark:- ark:- |", False, [("cmvn", False), ("subsample", -2, 0)]),
}


def table(seed=777, D=13):
    """Six utterances of three speakers; lengths 1, 2 and 3 exercise the clamped edges and a subsampling that leaves nothing."""
    rng = np.random.default_rng(seed)
    lens = [1, 2, 3, 17, 40, 64]
    utts = [(f"spk{i % 3}_utt{i}", (rng.standard_normal((T, D)) * rng.uniform(0.5, 3.0, D) + rng.uniform(-2, 2, D)).astype(np.float32))
            for i, T in enumerate(lens)]
    return utts


def run_reference(utts, tail: str, norm_vars: bool, tmp: str):
    ark, scp = os.path.join(tmp, "raw.ark"), os.path.join(tmp, "raw.scp")
    kaldi_io.write_mat_ark(ark, utts, scp_path=scp)
    with open(os.path.join(tmp, "utt2spk"), "w") as f:
        for k, _ in utts:
            f.write(f"{k} {k.split('_')[0]}\n")
    spk2utt = {}
    for k, _ in utts:
        spk2utt.setdefault(k.split("_")[0], []).append(k)
    with open(os.path.join(tmp, "spk2utt"), "w") as f:
        for s, ks in spk2utt.items():
            f.write(s + " " + " ".join(ks) + "\n")
    env = dict(os.environ, PATH=BIN + os.pathsep + os.environ["PATH"])
    cmvn_ark = os.path.join(tmp, "cmvn.ark")
    subprocess.run(f"compute-cmvn-stats --spk2utt=ark:{tmp}/spk2utt scp:{scp} ark:{cmvn_ark}", shell=True, check=True, env=env,
                   stderr=subprocess.DEVNULL)
    rspec = (f"ark:apply-cmvn --norm-vars={'true' if norm_vars else 'false'} --utt2spk=ark:{tmp}/utt2spk ark:{cmvn_ark} scp:{scp} ark:- | " + tail)
    out = os.path.join(tmp, "out.ark")
    subprocess.run(f"copy-feats '{rspec}' ark:{out}", shell=True, check=True, env=env, stderr=subprocess.DEVNULL)
    stats = dict(kaldi_io.read_mat64_table(f"ark:{cmvn_ark}"))
    return dict(kaldi_io.read_mat_table(f"ark:{out}")), stats, rspec


def main():
    assert os.path.isfile(os.path.join(BIN, "apply-cmvn")), "build oracle/_ref first: make -C oracle/ref_build featbin"
    utts = table()
    blob = {}
    for i, (k, m) in enumerate(utts):
        blob[f"raw/{k}"] = m
    for name, (tail, norm_vars, _) in PIPELINES.items():
        with tempfile.TemporaryDirectory() as tmp:
            outs, stats, _ = run_reference(utts, tail, norm_vars, tmp)
        for k, m in outs.items():
            blob[f"out/{name}/{k}"] = m
        for s, m in stats.items():
            blob[f"stats/{s}"] = m
        print(name, {k: v.shape for k, v in outs.items()})
    np.savez_compressed(GOLDEN, **blob)
    print("wrote", GOLDEN, os.path.getsize(GOLDEN), "bytes")


if __name__ == "__main__":
    main()
