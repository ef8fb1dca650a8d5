"""CPU: Kaldi table I/O against the reference's own table classes (oracle/_ref, skipped where absent), and the
minibatch assembly rule of train-ctc-parallel.cc:144-193."""
import ctypes as C
import os

import numpy as np
import pytest

from eesen_amd import kaldi_io
from eesen_amd.batching import assemble, interleave, AssemblyStats
from oracle import refbind


def _utts(n, D=5, seed=0, lo=3, hi=12):
    rng = np.random.default_rng(seed)
    feats = [(f"utt{i:03d}", rng.standard_normal((int(rng.integers(lo, hi)), D)).astype(np.float32)) for i in range(n)]
    labs = {k: rng.integers(1, 9, size=max(1, m.shape[0] // 3)).astype(np.int32) for k, m in feats}
    return feats, labs


@pytest.mark.parametrize("text", [False, True])
def test_roundtrip_ark_and_scp(tmp_path, text):
    feats, labs = _utts(7)
    ark, scp, lab = str(tmp_path / "f.ark"), str(tmp_path / "f.scp"), str(tmp_path / "l.ark")
    kaldi_io.write_mat_ark(ark, feats, text=text, scp_path=None if text else scp)
    kaldi_io.write_vec_int_ark(lab, labs.items(), text=text)
    back = list(kaldi_io.read_mat_table(("ark,t:" if text else "ark:") + ark))
    assert [k for k, _ in back] == [k for k, _ in feats]
    for (_, a), (_, b) in zip(back, feats):
        assert np.array_equal(a, b)
    if not text:
        for (ka, a), (kb, b) in zip(kaldi_io.read_mat_table("scp:" + scp), feats):
            assert ka == kb and np.array_equal(a, b)
    lb = kaldi_io.read_vec_int_table(("ark,t:" if text else "ark:") + lab)
    assert set(lb) == set(labs) and all(np.array_equal(lb[k], labs[k]) for k in labs)


@pytest.mark.skipif(not refbind.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("text", [False, True])
def test_against_the_reference_table_classes(tmp_path, text):
    lib = refbind._load()
    feats, labs = _utts(5, D=4, seed=3)
    mode = "ark,t:" if text else "ark:"
    # (1) written by the reference's BaseFloatMatrixWriter / Int32VectorWriter, read by us
    ark, lab = str(tmp_path / "ref.ark"), str(tmp_path / "ref_lab.ark")
    keys = (C.c_char_p * len(feats))(*[k.encode() for k, _ in feats])
    mats = (C.c_void_p * len(feats))(*[m.ctypes.data for _, m in feats])
    rows = (C.c_int * len(feats))(*[m.shape[0] for _, m in feats])
    assert lib.ref_write_feats((mode + ark).encode(), len(feats), keys, mats, rows, 4) == 0
    lv = [np.ascontiguousarray(labs[k]) for k, _ in feats]
    lp = (C.c_void_p * len(lv))(*[v.ctypes.data for v in lv])
    ll = (C.c_int * len(lv))(*[v.size for v in lv])
    assert lib.ref_write_labels((mode + lab).encode(), len(lv), keys, lp, ll) == 0
    back = list(kaldi_io.read_mat_table(mode + ark))
    assert [k for k, _ in back] == [k for k, _ in feats]
    for (_, a), (_, b) in zip(back, feats):
        assert np.allclose(a, b, rtol=1e-6 if text else 0, atol=0)
    lb = kaldi_io.read_vec_int_table(mode + lab)
    assert all(np.array_equal(lb[k], labs[k]) for k in labs)
    # (2) written by us, read by the reference's SequentialBaseFloatMatrixReader / RandomAccessInt32VectorReader
    ours, ours_lab = str(tmp_path / "ours.ark"), str(tmp_path / "ours_lab.ark")
    kaldi_io.write_mat_ark(ours, feats, text=text)
    kaldi_io.write_vec_int_ark(ours_lab, labs.items(), text=text)
    n, tot, cs = C.c_int(), C.c_long(), C.c_double()
    kb = C.create_string_buffer(4096)
    assert lib.ref_read_feats_summary((mode + ours).encode(), C.byref(n), C.byref(tot), C.byref(cs), kb, 4096) == 0
    want = sum(float((m.astype(np.float64) * (1 + np.arange(m.shape[1]))).sum()) for _, m in feats)
    assert n.value == len(feats) and tot.value == sum(m.shape[0] for _, m in feats)
    assert abs(cs.value - want) < 1e-4 * max(1.0, abs(want))
    assert kb.value.decode().split() == [k for k, _ in feats]
    out = np.zeros(64, np.int32)
    for k in labs:
        cnt = lib.ref_read_labels((mode + ours_lab).encode(), k.encode(), out.ctypes.data_as(C.c_void_p), 64)
        assert cnt == labs[k].size and np.array_equal(out[:cnt], labs[k])


def test_interleave_is_time_major_zero_padded():
    mats = [np.full((2, 3), 1, np.float32), np.full((4, 3), 2, np.float32), np.full((1, 3), 3, np.float32)]
    feats, lens, T = interleave(mats, 3)
    assert T == 4 and lens.tolist() == [2, 4, 1] and feats.shape == (12, 3)
    f = feats.reshape(4, 3, 3)
    assert np.all(f[:2, 0] == 1) and np.all(f[2:, 0] == 0) and np.all(f[:, 1] == 2) and np.all(f[0, 2] == 3) and np.all(f[1:, 2] == 0)


def test_grouping_rule():
    """Lengths 10,10,10,30,5,5 with --num-sequence=3 --frame-limit=60: [10,10,10] (full) | 30 alone would be 30*1<=60 ok, then
    5: max 30 * 2 = 60 <= 60 fits, then 5: 30 * 3 = 90 > 60 opens the next group."""
    lens = [10, 10, 10, 30, 5, 5]
    feats = [(f"u{i}", np.zeros((n, 2), np.float32)) for i, n in enumerate(lens)]
    labs = {k: np.array([1], np.int32) for k, _ in feats}
    got = [mb.lens.tolist() for mb in assemble(iter(feats), labs, 3, 60, 2)]
    assert got == [[10, 10, 10], [30, 5], [5]]


def test_skips_and_counts():
    feats = [("a", np.zeros((4, 2), np.float32)), ("nolab", np.zeros((4, 2), np.float32)), ("huge", np.zeros((50, 2), np.float32)),
             ("b", np.zeros((6, 2), np.float32))]
    labs = {"a": np.array([1], np.int32), "huge": np.array([1], np.int32), "b": np.array([2, 3], np.int32)}
    st = AssemblyStats()
    mbs = list(assemble(iter(feats), labs, 5, 40, 2, st))
    assert [mb.keys for mb in mbs] == [["a", "b"]] and st.num_no_tgt_mat == 1 and st.num_too_long == 1
    assert mbs[0].T == 6 and mbs[0].feats.shape == (12, 2)


def test_every_utterance_lands_in_exactly_one_group():
    feats, labs = _utts(40, seed=9, lo=5, hi=60)
    for ns, fl in [(1, 100), (4, 100), (10, 250), (64, 1e5)]:
        mbs = list(assemble(iter(feats), labs, ns, fl, 5))
        assert [k for mb in mbs for k in mb.keys] == [k for k, _ in feats]
        assert all(mb.S <= ns and mb.T * mb.S <= fl for mb in mbs)


def test_compressed_feature_archive_golden():
    """`CM` / `CM2` archives (copy-feats --compress=true): the fixture was written by the reference's CompressedMatrixWriter and
    carries the reference's own CopyToMat output (oracle/make_golden.py compressed_feature_case); our reader must decode the
    same bits."""
    import os
    g = os.path.join(os.path.dirname(__file__), "golden")
    ref = np.load(os.path.join(g, "compressed_feats.npz"))
    got = list(kaldi_io.read_mat_table("ark:" + os.path.join(g, "compressed_feats.ark")))
    assert [k for k, _ in got] == list(ref["keys"])
    o = 0
    for (_, m), r in zip(got, ref["rows"]):
        assert m.dtype == np.float32 and m.shape == (r, 6)
        assert np.array_equal(m, ref["decoded"][o:o + r])
        # lossy but close: one byte per element over the column's range
        orig = ref["original"][o:o + r]
        assert np.abs(m - orig).max() <= 0.02 * max(1e-3, float(np.ptp(orig)))
        o += r


@pytest.mark.skipif(not refbind.available(), reason="oracle/_ref not built")
def test_compressed_feature_archive_against_reference(tmp_path):
    lib = refbind._load()
    rng = np.random.default_rng(21)
    mats = [(f"k{i}", (rng.standard_normal((r, 40)) * 3).astype(np.float32)) for i, r in enumerate([300, 7, 64, 2])]
    n = len(mats)
    keys = (C.c_char_p * n)(*[k.encode() for k, _ in mats])
    ptrs = (C.c_void_p * n)(*[m.ctypes.data for _, m in mats])
    rows = (C.c_int * n)(*[m.shape[0] for _, m in mats])
    decoded = np.zeros((sum(m.shape[0] for _, m in mats), 40), np.float32)
    path = str(tmp_path / "c.ark")
    lib.ref_write_compressed_feats.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_void_p]
    assert lib.ref_write_compressed_feats(("ark:" + path).encode(), n, keys, ptrs, rows, 40, decoded.ctypes.data_as(C.c_void_p)) == 0
    got = np.concatenate([m for _, m in kaldi_io.read_mat_table("ark:" + path)])
    assert np.array_equal(got, decoded)


def test_assembly_skips_utterances_the_ctc_cannot_take():
    """Empty transcripts (the reference reads alpha column -1 there, ctc-loss.cc:151) and transcripts beyond the 2047 labels a lattice of
    4096 positions holds are dropped with a warning and counted as `other errors` instead of aborting the run; 600 labels -- beyond
    the 511 of rounds 1-4 -- train."""
    from eesen_amd.batching import assemble, AssemblyStats
    rng = np.random.default_rng(3)
    feats = [(f"u{i}", rng.standard_normal((10 + i, 4)).astype(np.float32)) for i in range(5)]
    targets = {"u0": np.array([1, 2], np.int32), "u1": np.zeros(0, np.int32), "u2": np.array([3], np.int32),
               "u3": np.ones(2100, np.int32), "u4": np.array([2, 2], np.int32), "u5": np.ones(600, np.int32)}
    feats.append(("u5", rng.standard_normal((16, 4)).astype(np.float32)))
    st = AssemblyStats()
    got = list(assemble(iter(feats), targets, 2, 1e5, 4, st))
    assert [k for mb in got for k in mb.keys] == ["u0", "u2", "u4", "u5"]
    assert st.num_other_error == 2 and any("empty transcript" in w for w in st.warnings) and any("2047" in w for w in st.warnings)


def test_pipe_and_stdin_specifiers(tmp_path):
    """`cmd |` rspecifiers (what steps/train_ctc_parallel.sh:95-115 always passes) and `| cmd` wspecifiers."""
    from eesen_amd import kaldi_io
    rng = np.random.default_rng(4)
    mats = [(f"k{i}", rng.standard_normal((3 + i, 5)).astype(np.float32)) for i in range(4)]
    ark, out = str(tmp_path / "f.ark"), str(tmp_path / "o.ark")
    kaldi_io.write_mat_ark(ark, mats)
    got = list(kaldi_io.read_mat_table(f"ark:cat {ark} |"))
    assert [k for k, _ in got] == [k for k, _ in mats] and all(np.array_equal(a, b) for (_, a), (_, b) in zip(got, mats))
    kaldi_io.write_mat_ark(f"| cat > {out}", iter(mats))
    assert open(out, "rb").read() == open(ark, "rb").read()
    with pytest.raises(kaldi_io.KaldiIOError):
        list(kaldi_io.read_mat_table("ark:false |"))


def _read_tables(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "read_tables")
    subprocess.run(["g++", "-O1", "-std=c++17", os.path.join(root, "tests", "native", "read_tables.cc"), "-o", exe], check=True)
    return exe


def test_native_table_reader_scp_entries_and_failing_pipes(tmp_path):
    """eesen_amd/csrc/tools/kaldi_tables.h: an scp entry is everything behind the key -- also when its first
    word occurs inside the key (`cat_utt1 cat f |`) --, and a table pipe that ends with a non-zero status is an error, not a
    silently truncated table."""
    import subprocess
    from eesen_amd import kaldi_io
    exe = _read_tables(tmp_path)
    rng = np.random.default_rng(3)
    mats = [("cat_utt1", rng.standard_normal((5, 4)).astype(np.float32)), ("utt2", rng.standard_normal((7, 4)).astype(np.float32))]
    ark, scp = str(tmp_path / "f.ark"), str(tmp_path / "f.scp")
    kaldi_io.write_mat_ark(ark, mats, scp_path=scp)
    one = str(tmp_path / "one.ark"); kaldi_io.write_mat_ark(one, mats[:1])
    # entry 1: a command whose first word is a prefix of the key; entry 2: path:offset
    off = open(scp).read().splitlines()[1].split()[1]
    cmd_scp = str(tmp_path / "cmd.scp")
    open(cmd_scp, "w").write(f"cat_utt1 cat {one} | tail -c +10 |\nutt2\t{off}\n")   # (skips `cat_utt1 ` = 9 bytes: the bare matrix)
    r = subprocess.run([exe, "feats", "scp:" + cmd_scp], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.split() == ["cat_utt1", "5", "4", "utt2", "7", "4"], (r.stdout, r.stderr)
    r = subprocess.run([exe, "feats", f"ark:cat {ark} |"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.split() == ["cat_utt1", "5", "4", "utt2", "7", "4"]
    # a filter that dies after its first utterance: the reference warns, the Python reader raises; the native reader must not pass it off
    r = subprocess.run([exe, "feats", f"ark:cat {one}; exit 7 |"], capture_output=True, text=True)
    assert r.returncode == 3 and "truncated" in r.stderr, (r.returncode, r.stderr)
    lab = str(tmp_path / "l.ark"); kaldi_io.write_vec_int_ark(lab, [("a", np.arange(3, dtype=np.int32)), ("b", np.arange(5, dtype=np.int32))])
    r = subprocess.run([exe, "labels", f"ark:cat {lab} |"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.split() == ["a", "3", "b", "5"]
    r = subprocess.run([exe, "labels", f"ark:cat {lab}; false |"], capture_output=True, text=True)
    assert r.returncode == 3 and "truncated" in r.stderr
