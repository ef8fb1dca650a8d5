#!/usr/bin/env python3
"""End-to-end throughput of the TRAINER BINARIES on a synthetic Kaldi table (what a recipe actually runs: archive reading, grouping,
feeder, the whole loop, model writing) next to bench.py's recipe legs, which drive the same library from resident numpy matrices.
4 x 320 BiLSTM, D = 120, 512 length-sorted utterances of 210-1600 frames, ~45 phone targets.

    python scripts/trainer_throughput.py [--num-sequence 32] [--frame-limit 100000]

Prints one JSON object: per trainer (native C++, Python) the wall seconds of the process, the fps line the trainer logs
(train-ctc-parallel.cc:247-252 counts padded frames), and the padded frames/s over the whole process."""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eesen_amd import kaldi_io, nnet_io, synth   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--num-sequence", type=int, default=32)
ap.add_argument("--frame-limit", type=int, default=100000)
ap.add_argument("--utts", type=int, default=512)
ap.add_argument("--extract", action="store_true", help="also time net-output-extract over the table (--num-sequence 1 and 32)")
args = ap.parse_args()
cfg = dict(kind="BiLstmParallel", layers=4, H=320, D=120, K=46)
rng = np.random.default_rng(777)
lens = np.sort(np.clip(rng.gamma(6.0, 130.0, size=args.utts), 150, 1600).astype(int))
res = {"workload": f"4x320 BiLSTM, D=120, {args.utts} utterances of {lens.min()}-{lens.max()} frames ({int(lens.sum())} real frames), --num-sequence {args.num_sequence} --frame-limit {args.frame_limit}"}
with tempfile.TemporaryDirectory() as tmp:
    feats = [(f"utt{i:04d}", rng.standard_normal((int(n), cfg["D"])).astype(np.float32)) for i, n in enumerate(lens)]
    labs = {k: rng.integers(1, cfg["K"], size=max(1, m.shape[0] // 10)).astype(np.int32) for k, m in feats}
    ark, scp, lab = os.path.join(tmp, "feats.ark"), os.path.join(tmp, "feats.scp"), os.path.join(tmp, "labels.ark")
    kaldi_io.write_mat_ark(ark, feats, scp_path=scp)
    kaldi_io.write_vec_int_ark(lab, labs.items())
    m_in = os.path.join(tmp, "nnet.init")
    nnet_io.write_nnet(m_in, synth.make_model(max_grad=50.0, **cfg), binary=True)
    res["archive_MB"] = os.path.getsize(ark) / 1e6
    opts = ["--learn-rate=4e-5", "--momentum=0.9", f"--num-sequence={args.num_sequence}", f"--frame-limit={args.frame_limit}", "--report-step=100000"]
    for name, cmd in (("native", [os.path.join(ROOT, "eesen_amd", "bin", "train-ctc-parallel")]), ("python", [sys.executable, "-m", "eesen_amd.train_ctc_parallel"])):
        for rep in range(2):      # the second run has the archive in the page cache and the code objects loaded once before
            t0 = time.perf_counter()
            r = subprocess.run(cmd + opts + ["scp:" + scp, "ark:" + lab, m_in, os.path.join(tmp, f"nnet.{name}")], capture_output=True, text=True, cwd=ROOT,
                               env=dict(os.environ, PYTHONPATH=ROOT))
            wall = time.perf_counter() - t0
        m = re.search(r"fps\s*([0-9.e+]+)", r.stderr)
        done = re.search(r"Done (\d+) files", r.stderr)
        pad = re.search(r"\[TRAINING, ([0-9.e+-]+) min", r.stderr)
        res[name] = {"rc": r.returncode, "wall_s": wall, "logged_fps": float(m.group(1)) if m else None, "files": int(done.group(1)) if done else None,
                     "tail": r.stderr.strip().splitlines()[-2:] if r.returncode else None}
    # the step AFTER the path: net-output-extract over the same table, one utterance at a time (the reference's way,
    # the tool's default) and 32 together (--num-sequence, an extension: padding is masked, every utterance's output is its own)
    if args.extract:
        exe = os.path.join(ROOT, "eesen_amd", "bin", "net-output-extract")
        for ns in (1, 32):
            for rep in range(2):
                t0 = time.perf_counter()
                r = subprocess.run([exe, "--apply-log=true", f"--num-sequence={ns}", os.path.join(tmp, "nnet.native"), "scp:" + scp, "ark:/dev/null"],
                                   capture_output=True, text=True, cwd=ROOT)
                wall = time.perf_counter() - t0
            res[f"extract_num_sequence_{ns}"] = {"rc": r.returncode, "wall_s": wall, "real_frames_per_s": float(lens.sum()) / wall,
                                                 "tail": r.stderr.strip().splitlines()[-2:] if r.returncode else None}
print(json.dumps(res))
