"""-m gpu: the device feature front end (eesen_feeder_set_pipeline / eesen_feeder_submit_raw: apply-cmvn, splice-feats,
subsample-feats, add-deltas run on the packed utterances in HBM inside the batch assembly).

* bit-exact against the oracle (oracle/frontend.py: the reference's arithmetic, operation for operation, no FMA);
* against the committed outputs of the reference's own tools (tests/golden/frontend.npz): exact for the stages that copy or
  compute a + x*b, within the BLAS-saxpy rounding for the deltas;
* end to end: both trainers and both extractors, handed the recipes' pipe rspecifier, equal themselves handed an archive that
  went through the filters beforehand -- byte-identical models / outputs."""
import os
import subprocess
import sys

import numpy as np
import pytest

from eesen_amd import frontend as fe, kaldi_io, nnet_io, synth
from eesen_amd.batching import interleave
from oracle import frontend as ofe
from oracle import make_golden_frontend as mg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(mg.GOLDEN)
KIND = {"cmvn": fe.CMVN, "splice": fe.SPLICE, "subsample": fe.SUBSAMPLE, "deltas": fe.DELTAS}


def _stages(named):
    return [(KIND[s[0]], int(s[1]), int(s[2]) if len(s) > 2 else 0) for s in named]


def _gold_utts():
    return [(k[4:], GOLD[k]) for k in GOLD.files if k.startswith("raw/")]


def _run_device(named, utts, stats, norm_vars):
    """All utterances the pipeline leaves frames to, as ONE batch through the feeder -> per-utterance matrices."""
    from eesen_amd.api import Feeder
    f = Feeder()
    f.set_pipeline(_stages(named))
    has_cmvn = any(s[0] == "cmvn" for s in named)
    keep = [(k, m) for k, m in utts if f.pipeline_shape(m.shape[1], m.shape[0])[0] > 0]
    cm = [fe.cmvn_norm(stats[k.split("_")[0]], norm_vars) for k, _ in keep] if has_cmvn else None
    slot = f.submit_raw([m for _, m in keep], cm)
    got = f.acquire(slot)
    frames = [f.pipeline_shape(m.shape[1], m.shape[0])[0] for _, m in keep]
    T, S = max(frames), len(keep)
    D = f.pipeline_shape(keep[0][1].shape[1], 1)[1]
    assert (got.rows, got.cols) == (T * S, D)
    host = got.numpy().reshape(T, S, D)
    f.release(slot)
    out = {}
    for s, (k, _) in enumerate(keep):
        out[k] = host[:frames[s], s, :].copy()
        assert not host[frames[s]:, s, :].any(), "rows beyond an utterance's end are zero"
    return out


@pytest.mark.parametrize("name", sorted(mg.PIPELINES))
def test_device_pipeline_equals_oracle_and_reference_tools(gpu, name):
    _, norm_vars, named = mg.PIPELINES[name]
    utts = _gold_utts()
    stats = {k[6:]: GOLD[k] for k in GOLD.files if k.startswith("stats/")}
    got = _run_device(named, utts, stats, norm_vars)
    ref = {k.split("/", 2)[2]: GOLD[k] for k in GOLD.files if k.startswith(f"out/{name}/")}
    assert set(got) == set(ref), "the utterances the reference tools wrote, no more, no fewer"
    for k, raw in utts:
        want = ofe.run_pipeline(named, raw, stats[k.split("_")[0]])
        if want is None:
            assert k not in got
            continue
        assert np.array_equal(got[k], want), f"{name}/{k}: device vs oracle must be bit-exact"
        if any(s[0] == "deltas" for s in named):
            assert np.max(np.abs(got[k] - ref[k])) <= 4e-7 * (np.max(np.abs(ref[k])) + 1e-30)
        else:
            assert np.array_equal(got[k], ref[k])


@pytest.mark.parametrize("D,lens", [(40, [300, 211, 7, 1]), (3, [5, 2]), (16, [64])])
def test_orders_shapes_and_edges(gpu, D, lens):
    """Stage orders the recipes do not use, float4 and scalar interleave paths, single-frame utterances (every clamp active)."""
    rng = np.random.default_rng(D)
    utts = [(f"s_u{i}", rng.standard_normal((t, D)).astype(np.float32)) for i, t in enumerate(lens)]
    def stats_for(named):      # statistics of the dimension the CMVN stage sees (in front of it the features may have grown)
        Dc = D
        for s_ in named:
            if s_[0] == "cmvn":
                break
            Dc *= (1 + s_[1] + s_[2]) if s_[0] == "splice" else (1 + s_[1]) if s_[0] == "deltas" else 1
        st = np.zeros((2, Dc + 1)); st[0, :Dc] = rng.standard_normal(Dc) * 50; st[1, :Dc] = 100 + rng.random(Dc) * 400; st[0, Dc] = 57
        return st
    for named in ([("deltas", 2, 2), ("cmvn", True), ("splice", 0, 3)],
                  [("subsample", 2, 1), ("deltas", 3, 1), ("subsample", -2, 0)],
                  [("splice", 2, 0), ("splice", 0, 1)],
                  [("deltas", 0, 2)],
                  [("cmvn", False)]):
        st = stats_for(named)
        stats = {"s": st}
        got = _run_device(named, utts, stats, bool(dict((s[0], s[1]) for s in named).get("cmvn", False)))
        for k, raw in utts:
            want = ofe.run_pipeline(named, raw, st)
            assert (want is None) == (k not in got)
            if want is not None:
                assert np.array_equal(got[k], want), (named, k)


def test_front_end_argument_checks(gpu):
    from eesen_amd.api import Feeder, EesenError
    f = Feeder()
    with pytest.raises(EesenError, match="n must not be 0"):
        f.set_pipeline([(fe.SUBSAMPLE, 0, 0)])
    with pytest.raises(EesenError, match="offset"):
        f.set_pipeline([(fe.SUBSAMPLE, -2, 1)])                 # subsample-feats.cc:53-55
    with pytest.raises(EesenError, match="order"):
        f.set_pipeline([(fe.DELTAS, 2, 0)])                     # feature-functions.cc:213
    with pytest.raises(EesenError, match="unknown front-end stage"):
        f.set_pipeline([(9, 0, 0)])
    with pytest.raises(EesenError, match="at most one CMVN"):
        f.set_pipeline([(fe.CMVN, 0, 0), (fe.CMVN, 0, 0)])
    f.set_pipeline([(fe.CMVN, 1, 0), (fe.DELTAS, 2, 2)])
    m = np.ones((4, 5), np.float32)
    with pytest.raises(EesenError, match="CMVN"):
        f.submit_raw([m], None)                                 # the stage needs its vectors
    f.set_pipeline([(fe.SUBSAMPLE, 3, 2)])
    with pytest.raises(EesenError, match="empty after the front end"):
        f.submit_raw([m[:2]], None)
    f.set_pipeline([])                                          # back to the plain assembly
    slot = f.submit_raw([m], None)
    assert np.array_equal(f.acquire(slot).numpy(), interleave([m], 5)[0])


def _dataset(tmp_path, D=8, K=7, n=12, seed=9):
    """Raw table + CMVN statistics per speaker + utt2spk + labels sized for the SUBSAMPLED length."""
    rng = np.random.default_rng(seed)
    utts = [(f"spk{i % 3}_utt{i:02d}", (rng.standard_normal((int(rng.integers(20, 70)), D)) * 3 + 1).astype(np.float32)) for i in range(n)]
    utts.sort(key=lambda kv: kv[1].shape[0])
    ark, scp = str(tmp_path / "raw.ark"), str(tmp_path / "raw.scp")
    kaldi_io.write_mat_ark(ark, utts, scp_path=scp)
    stats = {}
    for k, m in utts:
        s = stats.setdefault(k.split("_")[0], np.zeros((2, D + 1)))
        s[0, :D] += m.sum(0, dtype=np.float64); s[1, :D] += (m.astype(np.float64) ** 2).sum(0); s[0, D] += m.shape[0]
    with open(tmp_path / "cmvn.ark", "wb") as f:       # Matrix<double>, binary, as compute-cmvn-stats writes it
        for spk, s in stats.items():
            import struct
            f.write(spk.encode() + b" \x00BDM \x04" + struct.pack("<i", 2) + b"\x04" + struct.pack("<i", D + 1) + s.astype("<f8").tobytes())
    with open(tmp_path / "utt2spk", "w") as f:
        for k, _ in utts:
            f.write(f"{k} {k.split('_')[0]}\n")
    labs = {k: rng.integers(1, K, size=max(1, m.shape[0] // 12)).astype(np.int32) for k, m in utts}
    lab = str(tmp_path / "labels.ark")
    kaldi_io.write_vec_int_ark(lab, labs.items())
    return utts, stats, scp, lab


def test_trainers_and_extractors_take_the_recipes_pipe_rspecifier(gpu, tmp_path):
    """`ark,s,cs:apply-cmvn ... | splice-feats ... | subsample-feats ... | add-deltas ... |` handed to train-ctc-parallel /
    net-output-extract (native binaries and Python mirrors) == the same tools on an archive that went through the filters
    beforehand (here: the oracle's, which the device reproduces bit for bit)."""
    D, K = 8, 7
    utts, stats, scp, lab = _dataset(tmp_path, D=D, K=K)
    named = [("cmvn", True), ("splice", 1, 1), ("subsample", 2, 1), ("deltas", 2, 2)]
    pre = [(k, ofe.run_pipeline(named, m, stats[k.split("_")[0]])) for k, m in utts]
    pre_ark = str(tmp_path / "pre.ark")
    kaldi_io.write_mat_ark(pre_ark, pre)
    Dn = pre[0][1].shape[1]
    assert Dn == D * 3 * 3
    cfg = synth.config("tiny_bi"); cfg.update(D=Dn, K=K)
    m_in = str(tmp_path / "init.nnet")
    nnet_io.write_nnet(m_in, synth.make_model(max_grad=50.0, **cfg), binary=True)
    pipe = (f"ark,s,cs:apply-cmvn --norm-vars=true --utt2spk=ark:{tmp_path}/utt2spk ark:{tmp_path}/cmvn.ark scp:{scp} ark:- | "
            "splice-feats --left-context=1 --right-context=1 ark:- ark:- | subsample-feats --n=2 --offset=1 ark:- ark:- | add-deltas ark:- ark:- |")
    assert fe.parse_feature_pipeline(pipe) is not None
    opts = ["--learn-rate=0.01", "--momentum=0.9", "--num-sequence=4", "--frame-limit=120", "--report-step=4", "--verbose=1"]
    exe_t = os.path.join(ROOT, "eesen_amd", "bin", "train-ctc-parallel")
    exe_x = os.path.join(ROOT, "eesen_amd", "bin", "net-output-extract")
    py_t = [sys.executable, "-m", "eesen_amd.train_ctc_parallel"]
    py_x = [sys.executable, "-m", "eesen_amd.net_output_extract"]
    # an environment without any featbin tool on PATH: the pipe can only have run on the device
    env = dict(os.environ, PATH="/usr/bin:/bin")
    models = {}
    for tag, cmd in (("cc", [exe_t]), ("py", py_t)):
        for src_tag, spec in (("pipe", pipe), ("pre", "ark:" + pre_ark)):
            out = str(tmp_path / f"{tag}_{src_tag}.nnet")
            r = subprocess.run(cmd + opts + [spec, "ark:" + lab, m_in, out], capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
            assert r.returncode == 0, r.stderr[-2000:]
            assert f"Done {len(utts)} files" in r.stderr
            models[(tag, src_tag)] = open(out, "rb").read()
    assert models[("cc", "pipe")] == models[("cc", "pre")] == models[("py", "pipe")] == models[("py", "pre")]
    outs = {}
    for tag, cmd in (("cc", [exe_x]), ("py", py_x)):
        for src_tag, spec in (("pipe", pipe), ("pre", "ark:" + pre_ark)):
            out = str(tmp_path / f"{tag}_{src_tag}.ark")
            r = subprocess.run(cmd + ["--num-sequence=3", str(tmp_path / "cc_pipe.nnet"), spec, "ark:" + out], capture_output=True, text=True,
                               cwd=ROOT, timeout=600, env=env)
            assert r.returncode == 0, r.stderr[-2000:]
            outs[(tag, src_tag)] = open(out, "rb").read()
    assert outs[("cc", "pipe")] == outs[("cc", "pre")] == outs[("py", "pipe")] == outs[("py", "pre")]
    # an utterance without statistics and one that the subsampling empties are dropped with the reference tools' warnings
    with open(tmp_path / "utt2spk", "a") as f:
        f.write("orphan_utt99 nobody\n")
    extra = [("orphan_utt99", utts[0][1]), ("spk0_short", utts[0][1][:1])]
    kaldi_io.write_mat_ark(str(tmp_path / "raw2.ark"), utts + extra, scp_path=str(tmp_path / "raw2.scp"))
    with open(tmp_path / "utt2spk", "a") as f:
        f.write("spk0_short spk0\n")
    pipe2 = pipe.replace(scp, str(tmp_path / "raw2.scp"))
    r = subprocess.run([exe_x, str(tmp_path / "cc_pipe.nnet"), pipe2, "ark:" + str(tmp_path / "o2.ark")], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and f"Done {len(utts)} files" in r.stderr
    assert "No normalization statistics available for key orphan_utt99" in r.stderr and "spk0_short, output would have no rows" in r.stderr
