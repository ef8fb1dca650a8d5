"""CPU: the command-line conventions of the reference's tools (/root/reference/src/util/parse-options.cc:338-400,470-656) in the
native tools (eesen_amd/csrc/tools/parse_options.h) and the Python hosts (eesen_amd/parse_options.py): --config=<file>, --print-args,
--help, `--x=y` before the positional arguments, normalised names, "Invalid option" + usage for the rest, the exit codes of
train-ctc-parallel.cc:82-85,259-263.  Where the reference's own trainer is built (oracle/_ref/train-ctc-parallel-seam: its
unmodified main() and ParseOptions), its --help and its error output are the arbiter, line for line."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "eesen_amd", "bin", "train-ctc-parallel")
NATIVE_EXTRACT = os.path.join(ROOT, "eesen_amd", "bin", "net-output-extract")
SEAM = os.path.join(ROOT, "oracle", "_ref", "train-ctc-parallel-seam")
PY = [sys.executable, "-m", "eesen_amd.train_ctc_parallel"]


def _run(cmd, *args):
    r = subprocess.run(list(cmd) + list(args), capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT))
    return r.returncode, r.stderr


def _hosts():
    hosts = [("python", PY)]
    if os.path.exists(NATIVE):
        hosts.append(("native", [NATIVE]))
    return hosts


def _option_lines(text):
    return {l.split(":")[0].strip(): l.strip() for l in text.splitlines() if l.startswith("  --")}


@pytest.mark.parametrize("name,cmd", _hosts())
def test_help_prints_the_usage_and_exits_zero(name, cmd):
    rc, err = _run(cmd, "--help")
    assert rc == 0
    assert "Usage: train-ctc-parallel [options] <feature-rspecifier> <labels-rspecifier> <model-in> [<model-out>]" in err
    assert "Options:" in err and "Standard options:" in err
    opts = _option_lines(err)
    for o in ("--learn-rate", "--momentum", "--num-sequence", "--frame-limit", "--cross-validate", "--opt-algorithm", "--config", "--print-args", "--help", "--verbose"):
        assert o in opts, (o, sorted(opts))
    if os.path.exists(SEAM):      # the reference's own main(): every option it documents reads the same here
        rc_r, err_r = _run([SEAM], "--help")
        assert rc_r == 0
        for o, line in _option_lines(err_r).items():
            assert opts.get(o) == line, (o, opts.get(o), line)


@pytest.mark.parametrize("name,cmd", _hosts())
def test_config_file_normalised_names_and_argument_count(name, cmd, tmp_path):
    conf = tmp_path / "train.conf"
    conf.write_text("# recipe settings\n--learn-rate=0.002   # halved\n\n--num_sequence=0x10\n--Momentum=0.9\n")
    # too few positional arguments: usage, exit code 1 (train-ctc-parallel.cc:82-85) -- AFTER the config file has been accepted
    rc, err = _run(cmd, f"--config={conf}", "--cross-validate", "a", "b")
    assert rc == 1 and "Usage: train-ctc-parallel" in err
    assert "--cross-validate a b" in err.splitlines()[0]          # --print-args (default true): the command line is echoed first
    rc, err = _run(cmd, f"--config={conf}", "--print-args=false", "--cross-validate", "a", "b")
    assert rc == 1 and not err.splitlines()[0].strip().endswith("a b")
    # a line that is no option, an unknown option in the file, a missing file
    bad = tmp_path / "bad.conf"
    bad.write_text("learn_rate=0.1\n")
    rc, err = _run(cmd, f"--config={bad}", "a", "b", "c", "d")
    assert rc == 255 and "does not look like a line" in err
    bad.write_text("--no-such-option=1\n")
    rc, err = _run(cmd, f"--config={bad}", "a", "b", "c", "d")
    assert rc == 255 and "Invalid option --no-such-option=1 in config file" in err
    rc, err = _run(cmd, f"--config={tmp_path / 'missing.conf'}", "a", "b", "c", "d")
    assert rc == 255 and "Cannot open config file" in err


@pytest.mark.parametrize("name,cmd", _hosts())
def test_invalid_options_are_refused_like_the_reference(name, cmd):
    cases = [(["--bogus=1", "a", "b", "c", "d"], "Invalid option --bogus=1"),
             (["--learn-rate", "0.1", "a", "b", "c"], "Invalid floating-point option"),      # `--x y` is not the reference's syntax
             (["--num-sequence=ten", "a", "b", "c", "d"], "Invalid integer option"),
             (["--binary=maybe", "a", "b", "c", "d"], "Invalid format for boolean argument"),
             (["--binary=", "a", "b", "c", "d"], "Invalid option --binary="),
             (["--opt-algorithm", "a", "b", "c", "d"], "Invalid option --opt-algorithm"),
             (["--=3", "a", "b", "c", "d"], "Invalid option (no key)")]
    for args, msg in cases:
        rc, err = _run(cmd, *args)
        assert rc == 255 and msg in err, (args, rc, err[-300:])
        if os.path.exists(SEAM):
            rc_r, err_r = _run([SEAM], *args)
            assert rc_r == 255 and msg in err_r, (args, rc_r, err_r[-300:])
    # an option behind the first positional argument is a positional argument (parse-options.cc:358-389): five of them -> usage, 1
    rc, err = _run(cmd, "a", "--learn-rate=0.1", "b", "c", "d")
    assert rc == 1 and "Usage:" in err
    # a lone -- ends the named options and is dropped
    rc, err = _run(cmd, "--cross-validate", "--", "--weird-name", "b")
    assert rc == 1


@pytest.mark.skipif(not os.path.exists(NATIVE_EXTRACT), reason="native tools not built")
def test_native_extractor_follows_the_same_conventions():
    rc, err = _run([NATIVE_EXTRACT], "--help")
    assert rc == 0 and "Usage:  net-output-extract [options] <model-in> <feature-rspecifier> <feature-wspecifier>" in err
    assert "--class-frame-counts" in err and "--apply-log" in err and "--use-gpu" in err
    rc, err = _run([NATIVE_EXTRACT], "--nope=1", "m", "ark:a", "ark:b")
    assert rc == 255 and "Invalid option --nope=1" in err
    rc, err = _run([sys.executable, "-m", "eesen_amd.net_output_extract"], "--nope=1", "m", "ark:a", "ark:b")
    assert rc == 255 and "Invalid option --nope=1" in err


def test_python_numbers_parse_the_way_strtol_and_strtod_do():
    """The reference converts option values with strtol(base 0) / strtod and accepts any value with a numeric PREFIX
    (parse-options.cc:561-656; csrc/tools/parse_options.h does the same): the Python hosts must take and refuse the same strings
    and read the same values -- held here against the C library itself."""
    import ctypes
    import math
    from eesen_amd.parse_options import _strtol0, _strtod
    libc = ctypes.CDLL(None)
    libc.strtol.restype = ctypes.c_long
    libc.strtol.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int]
    libc.strtod.restype = ctypes.c_double
    libc.strtod.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p)]

    def c_long(text):
        buf = ctypes.create_string_buffer(text.encode())
        end = ctypes.c_char_p()
        v = libc.strtol(buf, ctypes.byref(end), 0)
        consumed = ctypes.cast(end, ctypes.c_void_p).value - ctypes.addressof(buf)
        return None if consumed == 0 else ctypes.c_int32(v & 0xFFFFFFFF).value

    def c_double(text):
        buf = ctypes.create_string_buffer(text.encode())
        end = ctypes.c_char_p()
        v = libc.strtod(buf, ctypes.byref(end))
        consumed = ctypes.cast(end, ctypes.c_void_p).value - ctypes.addressof(buf)
        return None if consumed == 0 else v

    for t in ["10", "010", "08", "0x1F", "0X1f", "12abc", " 42", "+7", "-7", "abc", "", "-", "0", "1_0", "0x", "4294967297", "99999999999999999999"]:
        assert _strtol0(t) == c_long(t), (t, _strtol0(t), c_long(t))
    for t in ["1.5", "1.5x", "1_0", "1e-3", ".5", "5.", "abc", "", "-inf", "INF", "0x1p3", "1e", "1e+", "-.5e2z", ".", "+", "1e400"]:
        a, b = _strtod(t), c_double(t)
        assert (a is None) == (b is None) and (a is None or a == b or (math.isnan(a) and math.isnan(b))), (t, a, b)
