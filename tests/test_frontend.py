"""CPU: the feature front end's host logic and its oracle.

* oracle/frontend.py (restatement of apply-cmvn / splice-feats / subsample-feats / add-deltas) against the committed outputs of
  the reference's own tools (tests/golden/frontend.npz, oracle/make_golden_frontend.py) -- and, where oracle/_ref/featbin exists,
  against those tools run live on fresh seeds;
* the rspecifier-pipeline parser of eesen_amd/frontend.py against the recipes' command lines;
* eesen_cmvn_norm (host-only arithmetic of the C-ABI library) against the oracle.
The device kernels are pinned to the oracle in tests/test_gpu_frontend.py."""
import os
import tempfile

import numpy as np
import pytest

from eesen_amd import frontend as fe
from oracle import frontend as ofe
from oracle import make_golden_frontend as mg

GOLD = np.load(mg.GOLDEN)


def _gold_utts():
    return [(k[4:], GOLD[k]) for k in GOLD.files if k.startswith("raw/")]


def _check(name, stages, outs, utts, stats):
    """outs: key -> matrix as the reference tools wrote it."""
    for key, raw in utts:
        mine = ofe.run_pipeline(stages, raw, stats[key.split("_")[0]])
        if key not in outs:
            assert mine is None, f"{name}/{key}: the reference wrote nothing, the oracle {None if mine is None else mine.shape}"
            continue
        ref = outs[key]
        assert mine is not None and mine.shape == ref.shape, f"{name}/{key}"
        if not any(s[0] == "deltas" for s in stages):
            assert np.array_equal(mine, ref), f"{name}/{key}: copies and a + x*b are bit-exact"
        else:
            # AddVec = BLAS saxpy: fused or not depends on the BLAS build (this OpenBLAS fuses); a few ulp of the largest term
            scale = np.max(np.abs(ref)) + 1e-30
            assert np.max(np.abs(mine - ref)) <= 4e-7 * scale, f"{name}/{key}: {np.max(np.abs(mine - ref)) / scale:.2e}"


@pytest.mark.parametrize("name", sorted(mg.PIPELINES))
def test_oracle_against_the_reference_tools_golden(name):
    _, _, stages = mg.PIPELINES[name]
    outs = {k.split("/", 2)[2]: GOLD[k] for k in GOLD.files if k.startswith(f"out/{name}/")}
    stats = {k[6:]: GOLD[k] for k in GOLD.files if k.startswith("stats/")}
    assert outs
    _check(name, stages, outs, _gold_utts(), stats)


@pytest.mark.skipif(not os.path.isfile(os.path.join(mg.BIN, "apply-cmvn")), reason="oracle/_ref/featbin not built (needs /root/reference)")
@pytest.mark.parametrize("seed", [1, 2])
def test_oracle_against_the_reference_tools_live(seed):
    utts = mg.table(seed=seed, D=7 + seed)
    for name, (tail, norm_vars, stages) in mg.PIPELINES.items():
        with tempfile.TemporaryDirectory() as tmp:
            outs, stats, _ = mg.run_reference(utts, tail, norm_vars, tmp)
        _check(name, stages, outs, utts, stats)


def test_delta_window_coefficients():
    """The textbook regression windows (DeltaFeatures::DeltaFeatures): order 1, window 2 = [-2 -1 0 1 2] / 10."""
    sc = ofe.delta_scales(2, 2)
    assert np.allclose(sc[1], np.array([-2, -1, 0, 1, 2]) / 10.0, atol=1e-7)
    assert np.allclose(sc[2], np.convolve(sc[1], sc[1]), atol=1e-7) and sc[2].size == 9
    assert abs(sc[2].sum()) < 1e-6


RECIPE_LINES = {
    # asr_egs/wsj/steps/train_ctc_parallel.sh:95,109
    "ark,s,cs:apply-cmvn --norm-vars=true --utt2spk=ark:data/train/utt2spk scp:data/train/cmvn.scp scp:exp/train.scp ark:- | add-deltas ark:- ark:- |":
        ("scp:exp/train.scp", [(fe.CMVN, 1, 0), (fe.DELTAS, 2, 2)], "scp:data/train/cmvn.scp", "ark:data/train/utt2spk"),
    # decode_ctc_lat.sh:92-95
    "ark,s,cs:apply-cmvn --norm-vars=false --utt2spk=ark:d/utt2spk scp:d/cmvn.scp scp:d/feats.scp ark:- | splice-feats --left-context=1 --right-context=1 ark:- ark:- | "
    "subsample-feats --n=2 --offset=0 ark:- ark:- | add-deltas ark:- ark:- |":
        ("scp:d/feats.scp", [(fe.CMVN, 0, 0), (fe.SPLICE, 1, 1), (fe.SUBSAMPLE, 2, 0), (fe.DELTAS, 2, 2)], "scp:d/cmvn.scp", "ark:d/utt2spk"),
    # train_ctc_parallel.sh:103 (features copied to local disk first)
    "ark,s,cs:copy-feats scp:exp/train_local.scp ark:- |": ("scp:exp/train_local.scp", [], None, None),
    # defaults of the tools: splice 4 + 4, subsample n = 1, deltas 2 / 2; apply-cmvn without mean normalisation passes through
    "ark:apply-cmvn --norm-means=false g.cmvn ark:raw.ark ark:- | splice-feats ark:- ark:- | subsample-feats ark:- ark:- |":
        ("ark:raw.ark", [(fe.SPLICE, 4, 4), (fe.SUBSAMPLE, 1, 0)], None, None),
}


@pytest.mark.parametrize("line", sorted(RECIPE_LINES))
def test_pipeline_parser_on_recipe_lines(line):
    src, stages, cmvn, u2s = RECIPE_LINES[line]
    p = fe.parse_feature_pipeline(line)
    assert p is not None and p.source == src and p.stages == stages and p.cmvn == cmvn and p.utt2spk == u2s


@pytest.mark.parametrize("line", [
    "scp:feats.scp", "ark:feats.ark",                                                     # not a pipe at all
    "ark:apply-cmvn --skip-dims=0:1 scp:c.scp scp:f.scp ark:- |",                          # an option that is not implemented
    "ark:apply-cmvn scp:c.scp scp:f.scp ark:- | paste-feats ark:- scp:x.scp ark:- |",      # another tool
    "ark:add-deltas --truncate=13 scp:f.scp ark:- |",                                     # a filter cannot head the pipe
    "ark:apply-cmvn scp:c.scp scp:f.scp ark:- | add-deltas --truncate=13 ark:- ark:- |",
    "ark:apply-cmvn --norm-vars=true --norm-means=false scp:c.scp scp:f.scp ark:- |",      # the tool itself refuses this
    "ark:apply-cmvn scp:c.scp 'ark:gunzip -c f.gz |' ark:- |",                             # the source is itself a pipe
    "ark:apply-cmvn scp:c.scp scp:f.scp ark:- | subsample-feats --n=0 ark:- ark:- |",
])
def test_pipeline_parser_leaves_the_rest_to_the_shell(line):
    assert fe.parse_feature_pipeline(line) is None


def test_shapes_behind_the_pipeline():
    p = fe.parse_feature_pipeline("ark:copy-feats scp:f.scp ark:- | splice-feats --left-context=1 --right-context=1 ark:- ark:- | "
                                  "subsample-feats --n=3 --offset=1 ark:- ark:- | add-deltas ark:- ark:- |")
    assert p.out_dim(40) == 360
    for T in range(0, 12):
        assert p.out_frames(T) == len(range(1, T, 3))
    q = fe.parse_feature_pipeline("ark:copy-feats scp:f.scp ark:- | subsample-feats --n=-3 ark:- ark:- |")
    assert q.out_frames(5) == 15


def test_cmvn_normaliser_of_the_library_equals_the_oracle():
    """eesen_cmvn_norm is host arithmetic (no device needed): ApplyCmvn's offset / scale from the statistics matrix."""
    stats = {k[6:]: GOLD[k] for k in GOLD.files if k.startswith("stats/")}
    for s, m in stats.items():
        for nv in (False, True):
            assert np.array_equal(fe.cmvn_norm(m, nv), ofe.cmvn_norm(m, nv)), (s, nv)
    from eesen_amd._lib import EesenError
    with pytest.raises(EesenError, match="variance"):
        fe.cmvn_norm(stats["spk0"][:1], True)            # cmvn.cc:74-76
    bad = stats["spk0"].copy(); bad[0, -1] = 0.5
    with pytest.raises(EesenError, match="Insufficient stats"):
        fe.cmvn_norm(bad, False)                          # cmvn.cc:81-83


def test_raw_reader_drops_what_the_reference_tools_drop(tmp_path):
    """No statistics for the speaker -> apply-cmvn skips the utterance (apply-cmvn.cc:87-92); no frame behind the subsampling ->
    subsample-feats skips it (subsample-feats.cc:87-92)."""
    from eesen_amd import kaldi_io
    utts = _gold_utts()
    ark, scp = str(tmp_path / "raw.ark"), str(tmp_path / "raw.scp")
    kaldi_io.write_mat_ark(ark, utts, scp_path=scp)
    stats = {k[6:]: GOLD[k] for k in GOLD.files if k.startswith("stats/")}
    # a text-form double table, two of the three speakers only
    with open(tmp_path / "cmvn.ark", "w") as f:
        for s in ("spk0", "spk1"):
            f.write(s + "  [\n" + "\n".join("  " + " ".join(repr(float(v)) for v in row) for row in stats[s]) + " ]\n")
    with open(tmp_path / "utt2spk", "w") as f:
        for k, _ in utts:
            f.write(f"{k} {k.split('_')[0]}\n")
    line = (f"ark:apply-cmvn --norm-vars=true --utt2spk=ark:{tmp_path}/utt2spk ark:{tmp_path}/cmvn.ark scp:{scp} ark:- | "
            "subsample-feats --n=3 --offset=1 ark:- ark:- |")
    p = fe.parse_feature_pipeline(line)
    warnings = []
    got = dict(fe.read_raw(p, warn=warnings.append))
    want = {k for k, m in utts if not k.startswith("spk2") and m.shape[0] > 1}
    assert set(got) == want
    assert sum("No normalization statistics" in w for w in warnings) == 2 and sum("no rows" in w for w in warnings) == 1
    for k, u in got.items():
        raw = dict(utts)[k]
        assert u.shape == (len(range(1, raw.shape[0], 3)), raw.shape[1]) and np.array_equal(u.raw, raw)
        assert np.array_equal(u.cmvn, ofe.cmvn_norm(stats[k.split("_")[0]], True))


REFUSED = [
    "scp:feats.scp", "ark:feats.ark",
    "ark:apply-cmvn --skip-dims=0:1 scp:c.scp scp:f.scp ark:- |",
    "ark:apply-cmvn scp:c.scp scp:f.scp ark:- | paste-feats ark:- scp:x.scp ark:- |",
    "ark:add-deltas --truncate=13 scp:f.scp ark:- |",
    "ark:apply-cmvn scp:c.scp scp:f.scp ark:- | add-deltas --truncate=13 ark:- ark:- |",
    "ark:apply-cmvn --norm-vars=true --norm-means=false scp:c.scp scp:f.scp ark:- |",
    "ark:apply-cmvn scp:c.scp scp:f.scp ark:- | subsample-feats --n=0 ark:- ark:- |",
    "ark:apply-cmvn scp:c.scp scp:f.scp ark:- | splice-feats --left-context=-1 ark:- ark:- |",
    "ark:apply-cmvn scp:c.scp scp:f.scp ark:- | add-deltas --delta-window=0 ark:- ark:- |",
]


def test_native_parser_agrees_with_the_python_one(tmp_path):
    """csrc/tools/feat_pipeline.h (what the native trainer / extractor use) and eesen_amd/frontend.py must recognise exactly the
    same pipelines: a small g++ harness prints the C++ parse of every line."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "parse_pipeline")
    subprocess.run(["g++", "-O1", "-std=c++17", os.path.join(root, "tests", "native", "parse_pipeline.cc"), "-o", exe,
                    "-L" + os.path.join(root, "eesen_amd", "lib"), "-leesen_hip", "-Wl,-rpath," + os.path.join(root, "eesen_amd", "lib"),
                    "-Wl,--allow-shlib-undefined"], check=True)
    lines = sorted(RECIPE_LINES) + REFUSED + [
        "ark:copy-feats scp:f.scp ark:- | splice-feats --left_context=2 ark:- ark:- | subsample-feats --n=-2 ark:- ark:- | add-deltas --delta-order=1 --delta-window=3 ark:- ark:- |",
        "ark,s,cs:/opt/kaldi/bin/apply-cmvn --norm-vars=T --utt2spk=ark:u2s 'scp:c m.scp' scp:f.scp ark:- |",
    ]
    out = subprocess.run([exe] + lines, capture_output=True, text=True, check=True).stdout.splitlines()
    assert len(out) == len(lines)
    for line, got in zip(lines, out):
        p = fe.parse_feature_pipeline(line)
        want = "NONE" if p is None else "|".join([p.source, p.cmvn or "", p.utt2spk or "", str(int(p.norm_vars)),
                                                   ",".join(f"{k}:{a}:{b}" for k, a, b in p.stages)])
        assert got == want, (line, got, want)
