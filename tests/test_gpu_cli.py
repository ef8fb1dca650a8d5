"""-m gpu: the command-line mirror of train-ctc-parallel end to end (Kaldi tables in, <Nnet> out, TOKEN_ACCURACY on
stderr) against the oracle driven through the same minibatch assembly."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from eesen_amd import kaldi_io, nnet_io, synth
from eesen_amd.batching import assemble
from tests.util import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dataset(tmp_path, n=14, D=8, K=7, seed=5):
    rng = np.random.default_rng(seed)
    feats = [(f"spk{i % 3}_utt{i:02d}", rng.standard_normal((int(rng.integers(8, 30)), D)).astype(np.float32)) for i in range(n)]
    feats.sort(key=lambda kv: kv[1].shape[0])                   # recipes sort by length (train_ctc_parallel.sh:84-89)
    labs = {k: rng.integers(1, K, size=max(1, m.shape[0] // 5)).astype(np.int32) for k, m in feats}
    ark, scp, lab = str(tmp_path / "feats.ark"), str(tmp_path / "feats.scp"), str(tmp_path / "labels.ark")
    kaldi_io.write_mat_ark(ark, feats, scp_path=scp)
    kaldi_io.write_vec_int_ark(lab, labs.items())
    return feats, labs, scp, lab


def _run(args):
    return subprocess.run([sys.executable, "-m", "eesen_amd.train_ctc_parallel"] + args, capture_output=True, text=True, cwd=ROOT, timeout=600)


def test_training_iteration_matches_oracle(gpu, tmp_path):
    from oracle import net as onet
    cfg = synth.config("tiny_bi")
    layers = synth.make_model(max_grad=50.0, **cfg)
    feats, labs, scp, lab = _dataset(tmp_path, D=cfg["D"], K=cfg["K"])
    m_in, m_out = str(tmp_path / "nnet.init"), str(tmp_path / "nnet.iter1")
    nnet_io.write_nnet(m_in, layers, binary=True)
    opts = ["--learn-rate=0.01", "--momentum=0.9", "--num-sequence=4", "--frame-limit=90", "--report-step=4", "--verbose=1"]
    r = _run(opts + ["scp:" + scp, "ark:" + lab, m_in, m_out])
    assert r.returncode == 0, r.stderr[-3000:]
    assert "TRAINING STARTED" in r.stderr and re.search(r"TOKEN_ACCURACY >> [-0-9.e]+% <<", r.stderr)
    assert re.search(r"Done 14 files, 0 with no targets", r.stderr) and "Obj(log[Pzx])" in r.stderr
    got = nnet_io.read_nnet(m_out)
    ora = onet.OracleNet(layers, "f32"); ora.set_train_options(0.01, 0.9)
    errs = refs = 0
    for mb in assemble(iter(feats), labs, 4, 90, cfg["D"]):
        ora.set_seq_lengths(mb.lens)
        out = ora.propagate(mb.feats)
        ids = np.concatenate(mb.labels); off = np.concatenate([[0], np.cumsum([len(l) for l in mb.labels])]).astype(np.int32)
        c = onet.ctc_eval_parallel(out, mb.T, mb.S, mb.lens, ids, off, "f32")
        e, n = onet.ctc_error_rate_mseq(out, mb.T, mb.S, mb.lens, ids, off); errs += e; refs += n
        ora.backpropagate(c["diff"])
    assert rel_err(nnet_io.flatten_params(got), ora.get_params()) < 1e-4
    acc = float(re.search(r"TOKEN_ACCURACY >> ([-0-9.e]+)% <<", r.stderr).group(1))
    assert abs(acc - 100.0 * (1.0 - errs / refs)) < 1e-3
    # cross-validation: no model written, no parameter change, same report line
    r2 = _run(["--cross-validate=true", "--num-sequence=4", "--frame-limit=90", "scp:" + scp, "ark:" + lab, m_out])
    assert r2.returncode == 0 and "CROSS-VALIDATION STARTED" in r2.stderr and "TOKEN_ACCURACY" in r2.stderr


def test_cli_error_contract(gpu, tmp_path):
    cfg = synth.config("tiny_bi")
    feats, labs, scp, lab = _dataset(tmp_path, D=cfg["D"], K=cfg["K"])
    r = _run(["scp:" + scp, "ark:" + lab])                       # wrong argument count: usage, exit 1 (:81-84)
    assert r.returncode == 1 and "usage" in r.stderr.lower()
    r = _run(["scp:" + scp, "ark:" + lab, str(tmp_path / "missing.nnet"), str(tmp_path / "o")])
    assert r.returncode == 255 and "cannot open model file" in r.stderr
    m_in = str(tmp_path / "nnet.init"); nnet_io.write_nnet(m_in, synth.make_model(**cfg), binary=False)
    r = _run(["--opt-algorithm=Adam", "scp:" + scp, "ark:" + lab, m_in, str(tmp_path / "o")])
    assert r.returncode == 255 and "unknown optimization algorithm" in r.stderr
    r = _run(["--opt-algorithm=Adagrad", "--learn-rate=0.01", "scp:" + scp, "ark:" + lab, m_in, str(tmp_path / "o")])
    assert r.returncode == 0 and any("accu" in L for L in nnet_io.read_nnet(str(tmp_path / "o")))


def test_config_file_runs_like_the_same_options_on_the_command_line(gpu, tmp_path):
    """`--config=conf/train.conf` (parse-options.cc:338-400,470-506): one --x=y per line, comments, names with `_`; options on the
    command line override it.  Both trainers; the models equal the run with the options spelled out, byte for byte."""
    cfg = synth.config("tiny_bi")
    feats, labs, scp, lab = _dataset(tmp_path, D=cfg["D"], K=cfg["K"])
    m_in = str(tmp_path / "nnet.init"); nnet_io.write_nnet(m_in, synth.make_model(max_grad=50.0, **cfg), binary=True)
    conf = tmp_path / "train.conf"
    conf.write_text("# conf/train.conf\n--learn-rate=0.5      # overridden on the command line\n--momentum=0.9\n\n--num_sequence=4\n--frame-limit=90\n")
    outs = {}
    for name, cmd in (("python", [sys.executable, "-m", "eesen_amd.train_ctc_parallel"]), ("native", [os.path.join(ROOT, "eesen_amd", "bin", "train-ctc-parallel")])):
        for how, opts in (("config", [f"--config={conf}", "--learn-rate=0.01"]),
                          ("spelled", ["--learn-rate=0.01", "--momentum=0.9", "--num-sequence=4", "--frame-limit=90"])):
            out = str(tmp_path / f"{name}_{how}.nnet")
            r = subprocess.run(cmd + opts + ["scp:" + scp, "ark:" + lab, m_in, out], capture_output=True, text=True, cwd=ROOT, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            assert r.stderr.splitlines()[0].rstrip().endswith(out)      # --print-args: the command line, first thing on stderr
            outs[name, how] = open(out, "rb").read()
    assert outs["python", "config"] == outs["python", "spelled"] == outs["native", "config"] == outs["native", "spelled"]


def test_net_output_extract(gpu, tmp_path):
    """net-output-extract (forward for decoding): log-posteriors minus scaled log-priors, per utterance, against the oracle
    at S = 1; batching several utterances must not change a single bit on valid frames."""
    from oracle import net as onet
    from eesen_amd.net_output_extract import class_log_priors
    cfg = synth.config("tiny_bi")
    layers = synth.make_model(**cfg)
    feats, labs, scp, lab = _dataset(tmp_path, n=6, D=cfg["D"], K=cfg["K"])
    model = str(tmp_path / "final.nnet"); nnet_io.write_nnet(model, layers, binary=True)
    counts = str(tmp_path / "label.counts")
    open(counts, "w").write("[ 1200 30 0 45.5 8 19 77 ]\n")            # one class below the cutoff
    out1, out4 = str(tmp_path / "o1.ark"), str(tmp_path / "o4.ark")
    base = [sys.executable, "-m", "eesen_amd.net_output_extract", "--class-frame-counts=" + counts, "--apply-log=true", "--prior-scale=0.8",
            "--blank-scale=0.5"]
    r = subprocess.run(base + [model, "scp:" + scp, "ark:" + out1], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and "Done 6 files" in r.stderr, r.stderr[-2000:]
    r = subprocess.run(base + ["--num-sequence=4", model, "scp:" + scp, "ark:" + out4], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got1 = dict(kaldi_io.read_mat_table("ark:" + out1)); got4 = dict(kaldi_io.read_mat_table("ark:" + out4))
    pri = class_log_priors(counts, 1e-10, 0.5)
    ora = onet.OracleNet(layers, "f32")
    for key, m in feats:
        ora.set_seq_lengths([m.shape[0]])
        want = np.log(ora.propagate(m)) - np.float32(0.8) * pri[None, :]
        assert got1[key].shape == want.shape
        ok = pri < 1e30                                               # the floored class carries -0.8 * FLT_MAX/2 on both sides
        assert rel_err(got1[key][:, ok], want[:, ok]) < 1e-4
        assert np.all(got1[key][:, ~ok] < -1e37)
        assert np.array_equal(got1[key], got4[key])


def test_native_trainer_binary_equals_python_mirror(gpu, tmp_path):
    """eesen_amd/bin/train-ctc-parallel (host C++ over the C-ABI, its own Kaldi table readers) against the Python mirror on the
    same archives: the same library calls in the same order, so the written models are byte-identical and the report lines
    agree; text, scp and compressed inputs included."""
    exe = os.path.join(ROOT, "eesen_amd", "bin", "train-ctc-parallel")
    assert os.path.exists(exe), "run python -m eesen_amd.build"
    cfg = synth.config("tiny_bi")
    layers = synth.make_model(max_grad=50.0, **cfg)
    layers[0]["dropout"] = dict(forward=0.2, fw_step=True)          # device-drawn masks are a function of (seed, counter): equal too
    feats, labs, scp, lab = _dataset(tmp_path, D=cfg["D"], K=cfg["K"])
    m_in = str(tmp_path / "nnet.init")
    nnet_io.write_nnet(m_in, layers, binary=True)
    ark_t, lab_t = str(tmp_path / "feats_t.ark"), str(tmp_path / "labels_t.ark")
    kaldi_io.write_mat_ark(ark_t, feats, text=True)
    kaldi_io.write_vec_int_ark(lab_t, labs.items(), text=True)
    opts = ["--learn-rate=0.01", "--momentum=0.9", "--num-sequence=4", "--frame-limit=90", "--report-step=4", "--verbose=1"]
    for fspec, lspec in (("scp:" + scp, "ark:" + lab), ("ark:" + str(tmp_path / "feats.ark"), "ark:" + lab), ("ark,t:" + ark_t, "ark,t:" + lab_t)):
        o_py, o_cc = str(tmp_path / "py.nnet"), str(tmp_path / "cc.nnet")
        r1 = _run(opts + [fspec, lspec, m_in, o_py])
        r2 = subprocess.run([exe] + opts + [fspec, lspec, m_in, o_cc], capture_output=True, text=True, timeout=600)
        assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr[-1500:], r2.stderr[-1500:])
        assert open(o_py, "rb").read() == open(o_cc, "rb").read()
        acc = [re.search(r"TOKEN_ACCURACY >> ([-0-9.e]+)% <<", r.stderr).group(1) for r in (r1, r2)]
        assert acc[0] == acc[1]
        assert "TRAINING STARTED" in r2.stderr and re.search(r"Done 14 files, 0 with no targets, 0 with other errors", r2.stderr)
        assert "Obj(log[Pzx])" in r2.stderr
    # the compressed archive of the reference's writer (tests/golden/compressed_feats.ark): cross-validation over it runs
    g = os.path.join(ROOT, "tests", "golden")
    ref = np.load(os.path.join(g, "compressed_feats.npz"))
    labs6 = {str(k): np.arange(1, 1 + max(1, int(r) // 6), dtype=np.int32) % 5 + 1 for k, r in zip(ref["keys"], ref["rows"])}
    lab6 = str(tmp_path / "lab6.ark"); kaldi_io.write_vec_int_ark(lab6, labs6.items())
    cfg6 = dict(cfg); cfg6.update(D=6)
    m6 = str(tmp_path / "m6.nnet"); nnet_io.write_nnet(m6, synth.make_model(**cfg6), binary=True)
    cv = ["--cross-validate=true", "--num-sequence=3", "ark:" + os.path.join(g, "compressed_feats.ark"), "ark:" + lab6, m6]
    r1 = _run(cv); r2 = subprocess.run([exe] + cv, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr[-1500:], r2.stderr[-1500:])
    assert re.search(r"TOKEN_ACCURACY >> ([-0-9.e]+)% <<", r1.stderr).group(1) == re.search(r"TOKEN_ACCURACY >> ([-0-9.e]+)% <<", r2.stderr).group(1)
    # error contract: usage -> 1, failures -> message on stderr + 255 (train-ctc-parallel.cc:81-84, 260-263)
    assert subprocess.run([exe, "scp:" + scp], capture_output=True).returncode == 1
    r = subprocess.run([exe, "scp:" + scp, "ark:" + lab, str(tmp_path / "missing"), str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 255 and "cannot open model file" in r.stderr


def test_native_net_output_extract_equals_python_mirror(gpu, tmp_path):
    exe = os.path.join(ROOT, "eesen_amd", "bin", "net-output-extract")
    assert os.path.exists(exe), "run python -m eesen_amd.build"
    cfg = synth.config("tiny_bi")
    feats, labs, scp, lab = _dataset(tmp_path, n=6, D=cfg["D"], K=cfg["K"])
    model = str(tmp_path / "final.nnet"); nnet_io.write_nnet(model, synth.make_model(**cfg), binary=True)
    counts = str(tmp_path / "label.counts")
    open(counts, "w").write("[ 1200 30 0 45.5 8 19 77 ]\n")
    for extra in ([], ["--num-sequence=4"], ["--class-frame-counts=" + counts, "--apply-log=true", "--prior-scale=0.8", "--blank-scale=0.5"]):
        o_py, o_cc = str(tmp_path / "py.ark"), str(tmp_path / "cc.ark")
        r1 = subprocess.run([sys.executable, "-m", "eesen_amd.net_output_extract"] + extra + [model, "scp:" + scp, "ark:" + o_py],
                            capture_output=True, text=True, cwd=ROOT, timeout=600)
        r2 = subprocess.run([exe] + extra + [model, "scp:" + scp, "ark:" + o_cc], capture_output=True, text=True, timeout=600)
        assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr[-1500:], r2.stderr[-1500:])
        assert "Done 6 files" in r2.stderr
        assert open(o_py, "rb").read() == open(o_cc, "rb").read()
    # text output parses back to the same values
    o_t = str(tmp_path / "cc_t.ark")
    assert subprocess.run([exe, model, "scp:" + scp, "ark,t:" + o_t], capture_output=True).returncode == 0
    a = dict(kaldi_io.read_mat_table("ark:" + o_cc)); b = dict(kaldi_io.read_mat_table("ark,t:" + o_t))
    r3 = subprocess.run([exe, model, "scp:" + scp, "ark:" + o_cc], capture_output=True)
    a = dict(kaldi_io.read_mat_table("ark:" + o_cc))
    assert r3.returncode == 0 and all(np.array_equal(a[k], b[k]) for k in a)
    assert subprocess.run([exe, model], capture_output=True).returncode == 1


def test_reference_trainer_source_compiled_against_the_seam(gpu, tmp_path):
    """The C++ seam, COMPILED: oracle/_ref/train-ctc-parallel-seam is the reference's OWN src/netbin/train-ctc-parallel.cc,
    unmodified, built by oracle/ref_build/Makefile against include/eesen_hip_net.h (eesen::Net / eesen::Ctc / CuMatrix over the
    C-ABI; the reference's base / util / cpucompute stay).  Its loop (:144-215: the reference's own table readers, greedy
    grouping, host padding + interleave) drives libeesen_hip.so; the native trainer of this repository must write the same
    model, byte for byte, and report the same accuracy -- pipes (`ark:cat ... |`), --sequence-out-file and cross-validation included."""
    seam = os.path.join(ROOT, "oracle", "_ref", "train-ctc-parallel-seam")
    exe = os.path.join(ROOT, "eesen_amd", "bin", "train-ctc-parallel")
    if not os.path.exists(seam):
        pytest.skip("oracle/_ref/train-ctc-parallel-seam is built where /root/reference exists (make -C oracle/ref_build seam)")
    cfg = synth.config("tiny_bi")
    layers = synth.make_model(max_grad=50.0, **cfg)
    feats, labs, scp, lab = _dataset(tmp_path, D=cfg["D"], K=cfg["K"])
    m_in = str(tmp_path / "nnet.init")
    nnet_io.write_nnet(m_in, layers, binary=True)
    opts = ["--learn-rate=0.01", "--momentum=0.9", "--num-sequence=4", "--frame-limit=90", "--report-step=4", "--verbose=1"]
    ark = str(tmp_path / "feats.ark")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "eesen_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    for fspec, lspec in (("scp:" + scp, "ark:" + lab), (f"ark:cat {ark} |", f"ark:cat {lab} |")):
        o_ref, o_cc = str(tmp_path / "seam.nnet"), str(tmp_path / "cc.nnet")
        s_ref, s_cc = str(tmp_path / "seq_seam.txt"), str(tmp_path / "seq_cc.txt")
        r1 = subprocess.run([seam] + opts + ["--sequence-out-file=" + s_ref, fspec, lspec, m_in, o_ref], capture_output=True, text=True, timeout=600, env=env)
        r2 = subprocess.run([exe] + opts + ["--sequence-out-file=" + s_cc, fspec, lspec, m_in, o_cc], capture_output=True, text=True, timeout=600)
        assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr[-2500:], r2.stderr[-1500:])
        assert open(o_ref, "rb").read() == open(o_cc, "rb").read()
        acc = [re.search(r"TOKEN_ACCURACY >> ([-0-9.e]+)% <<", r.stderr).group(1) for r in (r1, r2)]
        assert acc[0] == acc[1]
        assert "TRAINING STARTED" in r1.stderr and re.search(r"Done 14 files, 0 with no targets, 0 with other errors", r1.stderr)
        assert open(s_ref).read() == open(s_cc).read() and open(s_ref).read().count("utt") == 14
    cv = ["--cross-validate=true", "--num-sequence=4", "--frame-limit=90", "scp:" + scp, "ark:" + lab, o_ref]
    r1 = subprocess.run([seam] + cv, capture_output=True, text=True, timeout=600, env=env)
    r2 = subprocess.run([exe] + cv, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr[-1500:], r2.stderr[-1500:])
    assert re.search(r"TOKEN_ACCURACY >> ([-0-9.e]+)% <<", r1.stderr).group(1) == re.search(r"TOKEN_ACCURACY >> ([-0-9.e]+)% <<", r2.stderr).group(1)


def test_two_jobs_with_uneven_per_job_lists_through_the_rccl_standin(gpu, tmp_path):
    """The recipes' multi-job launch (`JOB=1:$nj ... --num-jobs=$nj --job-id=JOB scp:feats_tr.JOB.scp`, train_ctc_parallel_h.sh:
    96,141-143: queue.pl has substituted JOB before the process starts, so each job receives ITS OWN list and no literal JOB)
    with two jobs on the one GPU of the box, the library's communicator running over the test stand-in for librccl.so
    (tests/native/fake_rccl.hip).  The per-job lists are uneven (8 and 5 utterances): job 2 runs out first and follows with
    zero-gradient steps.  Three hosts -- the native trainer, the Python mirror, and the reference's OWN trainer source behind the
    seam -- must write the same model, equal to one process stepping through the union of the jobs' minibatches, and job 1 must
    print the merged `TOTAL TOKEN_ACCURACY` line the recipe greps in its log (:147)."""
    import socket
    from tests.test_gpu_multirank import fake_rccl_path, _merge
    from eesen_amd.api import Net, Ctc
    exe = os.path.join(ROOT, "eesen_amd", "bin", "train-ctc-parallel")
    seam = os.path.join(ROOT, "oracle", "_ref", "train-ctc-parallel-seam")
    cfg = synth.config("tiny_bi")
    layers = synth.make_model(max_grad=50.0, **cfg)
    feats, labs, scp, lab = _dataset(tmp_path, n=13, D=cfg["D"], K=cfg["K"])
    lines = open(scp).read().splitlines()
    shard = {1: lines[0:13:2] + [lines[11]], 2: lines[1:11:2]}           # 8 and 5 utterances, each list still sorted by length
    shard[1].sort(key=lambda l: [k for k, _ in feats].index(l.split()[0]))
    for j, ls in shard.items():
        open(str(tmp_path / f"feats.{j}.scp"), "w").write("\n".join(ls) + "\n")
    m_in = str(tmp_path / "nnet.init")
    nnet_io.write_nnet(m_in, layers, binary=True)
    opts = ["--learn-rate=0.01", "--momentum=0.9", "--num-sequence=2", "--frame-limit=90", "--num-jobs=2"]

    def two_jobs(cmd, out, tag):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        env = dict(os.environ, EESEN_RCCL_LIBRARY=fake_rccl_path(), EESEN_PERSISTENT="0", FAKE_RCCL_QUIET="1", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), EESEN_DEVICE="0", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
                   LD_LIBRARY_PATH=os.path.join(ROOT, "eesen_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
        env.pop("RANK", None); env.pop("WORLD_SIZE", None)
        ps = [subprocess.Popen(cmd + opts + [f"--job-id={j}", "scp:" + str(tmp_path / f"feats.{j}.scp"), "ark:" + lab, m_in, out],
                               env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT) for j in (1, 2)]
        errs = [p.communicate(timeout=600)[1] for p in ps]
        assert [p.returncode for p in ps] == [0, 0], (tag, errs[0][-2500:], errs[1][-2500:])
        return errs

    o_cc, o_py, o_seam = (str(tmp_path / f"{t}.nnet") for t in ("cc", "py", "seam"))
    e_cc = two_jobs([exe, "--device=0"], o_cc, "native")
    e_py = two_jobs([sys.executable, "-m", "eesen_amd.train_ctc_parallel", "--device=0"], o_py, "python")
    total = lambda err: re.search(r"TOTAL TOKEN_ACCURACY >> ([-0-9.e]+)% <<", err)
    assert total(e_cc[0]) and not total(e_cc[1]) and total(e_py[0]) and not total(e_py[1])       # job 1 reports the merged line, once
    assert total(e_cc[0]).group(1) == total(e_py[0]).group(1)
    assert open(o_cc, "rb").read() == open(o_py, "rb").read()
    assert "ran out of minibatches" in e_cc[1] and "ran out of minibatches" not in e_cc[0]
    # the reference's own Info() / InfoGradient() log (train-ctc-parallel.cc:236-240): layer markers and per-tensor moments
    assert "layer 1 : <BiLstmParallel>, input-dim" in e_cc[0] and "wei_gifo_x_fw_  " in e_cc[0] and "### Gradient stats :" in e_cc[0]
    assert re.search(r"linearity_corr_ \( min [-0-9.e+]+, max [-0-9.e+]+, mean ", e_cc[0])
    if os.path.exists(seam):
        e_seam = two_jobs([seam], o_seam, "seam")
        assert open(o_seam, "rb").read() == open(o_cc, "rb").read(), rel_err(nnet_io.flatten_params(nnet_io.read_nnet(o_seam)), nnet_io.flatten_params(nnet_io.read_nnet(o_cc)))
        assert total(e_seam[0]) and not total(e_seam[1])
        assert abs(float(total(e_seam[0]).group(1)) - float(total(e_cc[0]).group(1))) < 1e-4
        assert "layer 1 : <BiLstmParallel>, input-dim" in e_seam[0] and "### Gradient stats :" in e_seam[0]
    # one process on the union: step k = the k-th minibatches of the two lists together
    groups = {j: list(assemble(((l.split()[0], dict(feats)[l.split()[0]]) for l in shard[j]), labs, 2, 90, cfg["D"])) for j in (1, 2)}
    os.environ["EESEN_PERSISTENT"] = "0"
    try:
        net = Net.from_layers(layers)
    finally:
        del os.environ["EESEN_PERSISTENT"]
    net.SetTrainOptions(0.01, 0.9)
    ctc = Ctc()
    errs_tot = refs_tot = 0
    for k in range(max(len(g) for g in groups.values())):
        mb = _merge([synth.Batch(feats=g[k].feats, lens=g[k].lens, labels=g[k].labels, T=g[k].T, S=g[k].S) for g in groups.values() if k < len(g)])
        net.SetSeqLengths(mb.lens)
        o = net.Propagate(mb.feats)
        d = ctc.EvalParallel(mb.lens, o, mb.labels)
        e, n = ctc.ErrorRateMSeq(mb.lens, o, mb.labels); errs_tot += e; refs_tot += n
        net.Backpropagate(d)
    assert len(groups[1]) > len(groups[2])
    assert rel_err(nnet_io.flatten_params(nnet_io.read_nnet(o_cc)), net.GetParams()) < 1e-5
    assert abs(float(total(e_cc[0]).group(1)) - 100.0 * (1.0 - errs_tot / refs_tot)) < 1e-3


def test_shared_list_dealing_is_opt_in(gpu, tmp_path):
    """A job trains EVERY minibatch of the list it was handed (reference semantics: the list is the job's shard); with
    --shard-shared-list=true job J trains minibatches J-1, J-1+N, ... of a list all jobs read -- checked through the utterance
    counts of a one-job-of-two cross-validation run.  The SAME list on every job without that switch (a launcher that hands all
    ranks one command line and no JOB to substitute) would train N copies of every minibatch: refused, loudly."""
    import shutil
    import socket
    from tests.test_gpu_multirank import fake_rccl_path
    cfg = synth.config("tiny_bi")
    feats, labs, scp, lab = _dataset(tmp_path, n=12, D=cfg["D"], K=cfg["K"])
    for j in (1, 2):
        shutil.copy(scp, str(tmp_path / f"feats.{j}.scp"))
    per_job = "scp:" + str(tmp_path / "feats.JOB.scp")
    m_in = str(tmp_path / "nnet.init"); nnet_io.write_nnet(m_in, synth.make_model(**cfg), binary=True)
    for cmd in ([os.path.join(ROOT, "eesen_amd", "bin", "train-ctc-parallel")], [sys.executable, "-m", "eesen_amd.train_ctc_parallel"]):
        # (extra options of job 1, of job 2, rspecifier, utterances done per job | the message both jobs must die with)
        for extra1, extra2, rspec, want in (([], [], per_job, (12, 12)), (["--shard-shared-list=true"], ["--shard-shared-list=true"], "scp:" + scp, (6, 6)),
                                            ([], [], "scp:" + scp, "were given the same feature rspecifier"),
                                            # the same words, different data on every node (node-local shards): the explicit override
                                            (["--allow-identical-lists=true"], ["--allow-identical-lists=true"], "scp:" + scp, (12, 12)),
                                            # jobs that disagree on a switch meet in the SAME collective and get a message, not a hang
                                            (["--shard-shared-list=true"], [], "scp:" + scp, "disagree on --shard-shared-list")):
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            env = dict(os.environ, EESEN_RCCL_LIBRARY=fake_rccl_path(), EESEN_PERSISTENT="0", FAKE_RCCL_QUIET="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            ps = [subprocess.Popen(cmd + ["--device=0", "--cross-validate=true", "--num-sequence=2", "--num-jobs=2", f"--job-id={j}"] + (extra1 if j == 1 else extra2) +
                                   [rspec, "ark:" + lab, m_in], env=env, stderr=subprocess.PIPE, text=True, cwd=ROOT) for j in (1, 2)]
            errs = [p.communicate(timeout=600)[1] for p in ps]
            extra = extra1
            if isinstance(want, str):
                assert [p.returncode for p in ps] == [255, 255], errs
                assert all(want in e for e in errs), errs
                continue
            assert [p.returncode for p in ps] == [0, 0], errs
            got = tuple(int(re.search(r"Done (\d+) files", e).group(1)) for e in errs)
            assert got == want, (extra, got)


def test_seam2_layer_adaptor_inside_the_reference_net(gpu, tmp_path):
    """Seam 2, COMPILED: oracle/_ref/seam2_check links the reference's own Net / Layer code (CPU mode)
    with include/eesen_hip_layer.h, swaps every <BiLstmParallel> of a model for HipBiLstmParallel -- the reference's layer class
    with PropagateFnc / BackpropagateFnc / Update running in libeesen_hip.so -- and compares two momentum steps (outputs,
    in_diff, updated parameters, model file round trip) with the unmodified reference net on the same inputs."""
    exe = os.path.join(ROOT, "oracle", "_ref", "seam2_check")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/seam2_check is built where /root/reference exists (make -C oracle/ref_build seam2)")
    cfg = synth.config("small_bi")
    layers = synth.make_model(max_grad=0.5, learn_rate_coef=0.7, **cfg)
    model = str(tmp_path / "m.nnet")
    nnet_io.write_nnet(model, layers, binary=True)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "eesen_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""), OPENBLAS_NUM_THREADS="8")
    r = subprocess.run([exe, model, "8", "30", "2"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "SEAM2 OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    assert "BiLstmParallel layers running in libeesen_hip.so: 2" in r.stdout
