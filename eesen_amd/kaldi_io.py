"""Kaldi tables for the trainer's two inputs: float feature matrices and int32 label vectors.

Host-side data loading for the CLI mirror (eesen_amd/train_ctc_parallel.py); formats follow the reference:
  * archive entry = `key` + space + object (/root/reference/src/util/kaldi-table-inl.h TableWriterArchiveImpl::Write);
  * binary objects start with `\\0B` (src/base/io-funcs-inl.h:183-187);
  * float matrix: `FM ` + int32 rows + int32 cols (each preceded by a size byte 4) + raw little-endian fp32
    (src/cpucompute/matrix.cc:968-994); text: ` [` newline-separated rows ` ]`;
  * int32 vector (BasicVectorHolder, src/util/kaldi-holder-inl.h:190-260): binary = sized int32 count, then every
    element as a sized int32; text = `1 2 3\\n`;
  * script files: `key path[:byte_offset]` per line, the offset pointing at the object (src/util/kaldi-table.h).
  * compressed float matrix (`copy-feats --compress=true`, src/cpucompute/compressed-matrix.cc:437-470): token `CM `
    (one byte per element, per-column 16-bit percentile header) or `CM2 ` (uint16 per element) + GlobalHeader without its
    format field {float min_value, float range, int32 rows, int32 cols}; decoded here with the reference's own fp32/fp64
    expression order (`Uint16ToFloat` :244-250, `CharToFloat` :363-373) -- bit-exact against its CopyToMat (tests).
Supported specifiers: `ark:path`, `ark,t:path`, `scp:path`, and the "extended filenames" of src/util/kaldi-io.cc the recipes
use: `-` (stdin for readers, stdout for writers), `cmd |` (read the command's output: steps/train_ctc_parallel.sh:95-115
always passes `ark,s,cs:apply-cmvn ... |` and `ark:gunzip -c labels.tr.gz|`) and `| cmd` (write into the command).
Double-precision matrices are outside the hot path and raise.
"""
from __future__ import annotations

import io
import struct
import subprocess
import sys
from typing import BinaryIO, Dict, Iterator, List, Tuple

import numpy as np


class KaldiIOError(RuntimeError):
    pass


def _parse_specifier(spec: str) -> Tuple[str, str, bool]:
    if ":" not in spec:
        raise KaldiIOError(f"bad table specifier '{spec}' (expected ark:... or scp:...)")
    head, path = spec.split(":", 1)
    opts = head.split(",")
    kind = opts[0]
    if kind not in ("ark", "scp"):
        raise KaldiIOError(f"unsupported table kind in '{spec}'")
    return kind, path, "t" in opts[1:]


def _read_token(f: BinaryIO) -> str:
    """Key of the next archive entry: bytes up to the first space; '' at EOF. Leading newlines (text mode) are skipped."""
    out = bytearray()
    while True:
        c = f.read(1)
        if not c:
            return out.decode()
        if c in b" \t":
            if out:
                return out.decode()
            continue
        if c in b"\r\n" and not out:
            continue
        out += c


def _read_sized_int(f: BinaryIO) -> int:
    b = f.read(5)
    if len(b) != 5 or b[0] != 4:
        raise KaldiIOError("bad binary int32")
    return struct.unpack("<i", b[1:])[0]


def _u16_to_float(min_value: np.float32, rng: np.float32, v: np.ndarray) -> np.ndarray:
    """CompressedMatrix::Uint16ToFloat (compressed-matrix.cc:244-250): min + range * (1/65535)f * value, all in fp32, left to right."""
    return (np.float32(min_value) + (np.float32(rng) * np.float32(1.52590218966964e-05)) * v.astype(np.float32)).astype(np.float32)


def _read_compressed(f: BinaryIO, fmt: int) -> np.ndarray:
    """CompressedMatrix::Read + CopyToMat (compressed-matrix.cc:437-470, 485-520)."""
    hdr = f.read(16)
    if len(hdr) != 16:
        raise KaldiIOError("truncated compressed-matrix header")
    min_value, rng = np.frombuffer(hdr, "<f4", 2)
    rows, cols = (int(x) for x in np.frombuffer(hdr, "<i4", 2, 8))
    if cols == 0:
        return np.zeros((0, 0), np.float32)
    if fmt == 2:
        buf = f.read(2 * rows * cols)
        if len(buf) != 2 * rows * cols:
            raise KaldiIOError("truncated compressed matrix")
        return _u16_to_float(min_value, rng, np.frombuffer(buf, "<u2").reshape(rows, cols))
    buf = f.read(cols * (8 + rows))
    if len(buf) != cols * (8 + rows):
        raise KaldiIOError("truncated compressed matrix")
    pc = _u16_to_float(min_value, rng, np.frombuffer(buf, "<u2", 4 * cols).reshape(cols, 4))   # percentiles 0, 25, 75, 100 per column
    v = np.frombuffer(buf, np.uint8, rows * cols, 8 * cols).reshape(cols, rows)                # column-major bytes
    p0, p25, p75, p100 = (pc[:, i:i + 1] for i in range(4))
    vf = v.astype(np.float32)

    def seg(lo, hi, x, scale):      # float diff * float value -> fp32; * double constant and + lo in fp64; rounded once to fp32
        return (lo.astype(np.float64) + ((hi - lo).astype(np.float32) * x).astype(np.float32).astype(np.float64) * scale).astype(np.float32)

    out = np.where(v <= 64, seg(p0, p25, vf, 1 / 64.0),
                   np.where(v <= 192, seg(p25, p75, vf - np.float32(64), 1 / 128.0), seg(p75, p100, vf - np.float32(192), 1 / 63.0)))
    return np.ascontiguousarray(out.T, np.float32)


def _read_matrix(f: BinaryIO) -> np.ndarray:
    hdr = f.read(2)
    if hdr == b"\x00B":
        tok = f.read(3)
        if tok == b"CM ":
            return _read_compressed(f, 1)
        if tok == b"CM2":
            if f.read(1) != b" ":
                raise KaldiIOError("malformed CM2 token")
            return _read_compressed(f, 2)
        if tok == b"DM ":
            raise KaldiIOError("double-precision matrices are not supported (the path is BaseFloat = float)")
        if tok != b"FM ":
            raise KaldiIOError(f"expected FM, got {tok!r}")
        rows, cols = _read_sized_int(f), _read_sized_int(f)
        buf = f.read(4 * rows * cols)
        if len(buf) != 4 * rows * cols:
            raise KaldiIOError("truncated matrix")
        return np.frombuffer(buf, dtype="<f4").reshape(rows, cols).astype(np.float32)
    # text: "[" rows "]" ; hdr holds the first two characters already
    txt = bytearray(hdr)
    while b"]" not in txt:
        chunk = f.readline()
        if not chunk:
            raise KaldiIOError("unterminated text matrix")
        txt += chunk
    s = txt.decode()
    body = s[s.index("[") + 1: s.index("]")]
    rows = [r.split() for r in body.strip().split("\n") if r.strip()]
    if not rows:
        return np.zeros((0, 0), np.float32)
    return np.array(rows, dtype=np.float32)


def _read_matrix64(f: BinaryIO) -> np.ndarray:
    """Matrix<double>::Read (cpucompute/matrix.cc:1012-1100): `DM`, or `FM` converted, or text.  CMVN statistics are doubles
    (featbin/compute-cmvn-stats.cc writes Matrix<double>)."""
    hdr = f.read(2)
    if hdr == b"\x00B":
        tok = f.read(3)
        if tok not in (b"DM ", b"FM "):
            raise KaldiIOError(f"expected DM or FM, got {tok!r}")
        rows, cols = _read_sized_int(f), _read_sized_int(f)
        w = 8 if tok == b"DM " else 4
        buf = f.read(w * rows * cols)
        if len(buf) != w * rows * cols:
            raise KaldiIOError("truncated matrix")
        return np.frombuffer(buf, dtype="<f8" if w == 8 else "<f4").reshape(rows, cols).astype(np.float64)
    txt = bytearray(hdr)
    while b"]" not in txt:
        chunk = f.readline()
        if not chunk:
            raise KaldiIOError("unterminated text matrix")
        txt += chunk
    s = txt.decode()
    body = s[s.index("[") + 1: s.index("]")]
    rows = [r.split() for r in body.strip().split("\n") if r.strip()]
    if not rows:
        return np.zeros((0, 0), np.float64)
    return np.array(rows, dtype=np.float64)


def _read_int_vector(f: BinaryIO) -> np.ndarray:
    c0 = f.read(1)
    if c0 == b"\n" or not c0:       # text mode, empty vector
        return np.zeros(0, np.int32)
    hdr = c0 + (f.read(1) if c0 == b"\x00" else b"")
    if hdr == b"\x00B":
        n = _read_sized_int(f)
        buf = f.read(5 * n)
        if len(buf) != 5 * n:
            raise KaldiIOError("truncated int vector")
        a = np.frombuffer(buf, dtype=np.uint8).reshape(n, 5)
        if n and not np.all(a[:, 0] == 4):
            raise KaldiIOError("bad element size in int vector")
        return np.ascontiguousarray(a[:, 1:]).view("<i4").reshape(n).astype(np.int32)
    line = hdr + f.readline()
    return np.array(line.split(), dtype=np.int32)


class _PipeIn:
    """`cmd |`: the command's stdout as a binary stream; a non-zero exit status is an error at close, as in the reference
    (src/util/kaldi-io.cc PipeInputImpl::Close)."""

    def __init__(self, cmd: str):
        self.cmd = cmd
        self.p = subprocess.Popen(cmd, shell=True, stdout=subprocess.PIPE)
        self.f = self.p.stdout

    def __enter__(self):
        return self.f

    def __exit__(self, *exc):
        self.f.close()
        rc = self.p.wait()
        if rc != 0 and exc[0] is None:
            raise KaldiIOError(f"command '{self.cmd}' exited with status {rc}")
        return False


class _Keep:
    """stdin / stdout as a context manager that does not close them."""

    def __init__(self, f):
        self.f = f

    def __enter__(self):
        return self.f

    def __exit__(self, *exc):
        self.f.flush() if self.f.writable() else None
        return False


def _open(path: str):
    path = path.strip()
    if path == "-":
        return _Keep(sys.stdin.buffer)
    if path.endswith("|"):
        return _PipeIn(path[:-1])
    return open(path, "rb")


class _PipeOut:
    def __init__(self, cmd: str):
        self.cmd = cmd
        self.p = subprocess.Popen(cmd, shell=True, stdin=subprocess.PIPE)
        self.f = self.p.stdin

    def __enter__(self):
        return self.f

    def __exit__(self, *exc):
        self.f.close()
        rc = self.p.wait()
        if rc != 0 and exc[0] is None:
            raise KaldiIOError(f"command '{self.cmd}' exited with status {rc}")
        return False


def _open_out(path: str):
    path = path.strip()
    if path == "-":
        return _Keep(sys.stdout.buffer)
    if path.startswith("|"):
        return _PipeOut(path[1:])
    return open(path, "wb")


def _iter_table(spec: str, read_obj) -> Iterator[Tuple[str, np.ndarray]]:
    kind, path, _ = _parse_specifier(spec)
    if kind == "ark":
        with _open(path) as f:
            while True:
                key = _read_token(f)
                if not key:
                    return
                yield key, read_obj(f)
    else:
        with open(path) as scp:
            for line in scp:
                line = line.strip()
                if not line:
                    continue
                key, loc = line.split(None, 1)
                if loc.rstrip().endswith("|"):      # a script entry may itself be a command
                    with _open(loc) as f:
                        yield key, read_obj(f)
                    continue
                off = 0
                if ":" in loc and loc.rsplit(":", 1)[1].isdigit():
                    loc, o = loc.rsplit(":", 1)
                    off = int(o)
                with open(loc, "rb") as f:
                    f.seek(off)
                    yield key, read_obj(f)


def read_mat_table(spec: str) -> Iterator[Tuple[str, np.ndarray]]:
    """SequentialBaseFloatMatrixReader (train-ctc-parallel.cc:124)."""
    return _iter_table(spec, _read_matrix)


def read_mat64_table(spec: str) -> Iterator[Tuple[str, np.ndarray]]:
    """RandomAccessDoubleMatrixReader (featbin/apply-cmvn.cc:80): CMVN statistics per utterance or speaker."""
    return _iter_table(spec, _read_matrix64)


def read_mat64_file(rxfilename: str) -> np.ndarray:
    """Input ki(rxfilename); Matrix<double>::Read (apply-cmvn.cc:118-122): `file`, `file:offset` or `cmd |`."""
    loc, off = rxfilename, 0
    if not loc.rstrip().endswith("|") and ":" in loc and loc.rsplit(":", 1)[1].isdigit():
        loc, o = loc.rsplit(":", 1)
        off = int(o)
    with _open(loc) as f:
        if off:
            f.seek(off)
        return _read_matrix64(f)


def read_token_map(spec: str) -> Dict[str, str]:
    """The utt2spk map behind RandomAccessTableReaderMapped (util/kaldi-table.h): a text archive of `utt spk` lines."""
    kind, path, _ = _parse_specifier(spec)
    if kind != "ark":
        raise KaldiIOError("utt2spk: only ark: tables are supported")
    out = {}
    with _open(path) as f:
        for line in f:
            parts = line.decode().split()
            if len(parts) >= 2:
                out[parts[0]] = parts[1]
    return out


def read_vec_int_table(spec: str) -> Dict[str, np.ndarray]:
    """RandomAccessInt32VectorReader (train-ctc-parallel.cc:125): the whole table as a dict."""
    return dict(_iter_table(spec, _read_int_vector))


# ---------------------------------------------------------------------------------------------- writers
def write_mat_ark(path: str, items, text: bool = False, scp_path: str = None):
    """BaseFloatMatrixWriter to `ark:path` (optionally also an scp with byte offsets, like ark,scp:).  `path` may be `-`
    (stdout) or `| cmd`; `items` may be a generator: every entry is written (and flushed) as it is produced, so a
    downstream tool of a pipe starts on the first utterance."""
    scp = open(scp_path, "w") if scp_path else None
    with _open_out(path) as f:
        pos = 0
        for key, m in items:
            m = np.ascontiguousarray(m, np.float32)
            f.write(key.encode() + b" ")
            pos += len(key.encode()) + 1
            if scp:
                scp.write(f"{key} {path}:{pos}\n")
            if text:
                body = b" [" + b"".join(b"\n  " + " ".join(repr(float(np.float32(v))) for v in r).encode() + b" " for r in m) + b"]\n"
            else:
                body = b"\x00BFM " + b"\x04" + struct.pack("<i", m.shape[0]) + b"\x04" + struct.pack("<i", m.shape[1]) + m.tobytes()
            f.write(body)
            pos += len(body)
            f.flush()
    if scp:
        scp.close()


def write_vec_int_ark(path: str, items, text: bool = False):
    """Int32VectorWriter to `ark:path` / `ark,t:path`."""
    with _open_out(path) as f:
        for key, v in items:
            v = np.ascontiguousarray(v, np.int32)
            f.write(key.encode() + b" ")
            if text:
                f.write((" ".join(str(int(x)) for x in v) + " \n").encode())
            else:
                f.write(b"\x00B\x04" + struct.pack("<i", v.size))
                out = np.empty((v.size, 5), np.uint8)
                out[:, 0] = 4
                out[:, 1:] = v.astype("<i4").view(np.uint8).reshape(v.size, 4)
                f.write(out.tobytes())
