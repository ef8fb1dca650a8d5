"""ctypes binding to oracle/_ref/libeesen_ref.so: the REAL reference (src/net + src/cpucompute compiled
unmodified, CPU mode) plus the reference's CUDA CTC kernels emulated on the CPU.  TEST INFRASTRUCTURE.

Build with `make -C oracle/ref_build` where /root/reference exists; elsewhere the prebuilt .so is used.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_DIR, "_ref", "libeesen_ref.so")
REFERENCE_ROOT = "/root/reference"


def build_if_possible() -> bool:
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "src")):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(_DIR, "ref_build")])
    return os.path.exists(LIB)


def available() -> bool:
    if not os.path.exists(LIB):
        return False
    try:
        _load()
        return True
    except OSError:
        return False


_lib = None


def _load():
    global _lib
    if _lib is None:
        lib = C.CDLL(LIB)
        lib.ref_net_read.restype = C.c_void_p
        lib.ref_last_error.restype = C.c_char_p
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def set_blas_threads(n: int):
    _load().ref_set_blas_threads(int(n))


class RefNet:
    """The reference's eesen::Net (+ eesen::Ctc for the error-rate path), CPU mode."""

    def __init__(self, model_path: str):
        self.lib = _load()
        self.h = C.c_void_p(self.lib.ref_net_read(model_path.encode()))
        if not self.h:
            raise RuntimeError("reference Net::Read failed: " + self.lib.ref_last_error().decode())
        self.din = self.lib.ref_net_input_dim(self.h)
        self.dout = self.lib.ref_net_output_dim(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_net_free(self.h)
            self.h = None

    def _ck(self, rc):
        if rc < 0:
            raise RuntimeError("reference error: " + self.lib.ref_last_error().decode())
        return rc

    def num_params(self) -> int:
        return self.lib.ref_net_num_params(self.h)

    def get_params(self) -> np.ndarray:
        out = np.empty(self.num_params(), np.float32)
        self._ck(self.lib.ref_net_get_params(self.h, _p(out)))
        return out

    def set_train_options(self, learn_rate: float, momentum: float):
        self._ck(self.lib.ref_net_set_train_options(self.h, C.c_float(learn_rate), C.c_float(momentum)))

    def set_seq_lengths(self, lens):
        lens = np.ascontiguousarray(lens, np.int32)
        self._ck(self.lib.ref_net_set_seq_lengths(self.h, _p(lens), len(lens)))

    def propagate(self, feats: np.ndarray) -> np.ndarray:
        feats = np.ascontiguousarray(feats, np.float32)
        out = np.empty((feats.shape[0], self.dout), np.float32)
        self._ck(self.lib.ref_net_propagate(self.h, _p(feats), feats.shape[0], _p(out)))
        return out

    def backpropagate(self, out_diff: np.ndarray, want_in_diff: bool = True, lowmem: bool = False) -> Optional[np.ndarray]:
        """Net::Backpropagate (net.cc:88-108).  lowmem: the same per-layer Backpropagate + Update loop driven from
        ref_driver.cc, which releases each BiLstm layer's state buffers once the layer is done (SGD only)."""
        out_diff = np.ascontiguousarray(out_diff, np.float32)
        in_diff = np.empty((out_diff.shape[0], self.din), np.float32) if want_in_diff else None
        fn = self.lib.ref_net_backpropagate_lowmem if lowmem else self.lib.ref_net_backpropagate
        self._ck(fn(self.h, _p(out_diff), out_diff.shape[0], _p(in_diff) if want_in_diff else None))
        return in_diff

    def write(self, path: str, binary: bool):
        self._ck(self.lib.ref_net_write(self.h, path.encode(), int(binary)))

    def set_mode(self, train: bool):
        """Net::SetTrainMode / SetTestMode (net.cc:396-412)."""
        self._ck(self.lib.ref_net_set_mode(self.h, int(bool(train))))

    def dropout_masks(self, layer: int) -> dict:
        """The masks the reference drew in its last Propagate for BiLstm(Parallel) layer `layer` (bilstm-parallel-layer.h:46-94):
        fwd [T*S x 2H] (empty if unused), rec_fw / rec_bw [rows x H] with rows = (T+2)*S (step) or S (sequence), and the
        twiddle coin.  Values are 0 or 1/(1-p)."""
        dims = (C.c_int * 5)()
        rc = self.lib.ref_net_get_dropout_masks(self.h, layer, None, None, None, dims)
        if rc == -2:
            return {}
        self._ck(rc)
        fwd = np.zeros((dims[0], dims[1]), np.float32); rf = np.zeros((dims[2], dims[3]), np.float32); rb = np.zeros_like(rf)
        self._ck(self.lib.ref_net_get_dropout_masks(self.h, layer, _p(fwd) if fwd.size else None, _p(rf) if rf.size else None,
                                                    _p(rb) if rb.size else None, dims))
        return dict(fwd=fwd, rec_fw=rf, rec_bw=rb, twiddle_apply_forward=bool(dims[4]))

    def error_rate_mseq(self, net_out, T, S, lens, label_ids, label_off):
        net_out = np.ascontiguousarray(net_out, np.float32)
        lens = np.ascontiguousarray(lens, np.int32)
        ids = np.ascontiguousarray(label_ids, np.int32); off = np.ascontiguousarray(label_off, np.int32)
        ne, nr = C.c_float(0), C.c_int(0)
        self._ck(self.lib.ref_ctc_error_rate_mseq(self.h, _p(net_out), T, S, net_out.shape[1], _p(lens), _p(ids), _p(off),
                                                  C.byref(ne), C.byref(nr)))
        return int(round(ne.value)), nr.value


def cuda_ctc_eval_parallel(probs: np.ndarray, T: int, S: int, lens, label_ids, label_off):
    """The reference's CTC (CUDA kernel bodies run on the CPU). Returns dict(alpha, beta, pzx, diff, ctc_err, L)."""
    lib = _load()
    probs = np.ascontiguousarray(probs, np.float32)
    K = probs.shape[1]
    lens = np.ascontiguousarray(lens, np.int32)
    ids = np.ascontiguousarray(label_ids, np.int32); off = np.ascontiguousarray(label_off, np.int32)
    L = 2 * int(np.max(np.diff(off))) + 1
    alpha = np.empty((T * S, L), np.float32); beta = np.empty((T * S, L), np.float32)
    pzx = np.empty(S, np.float32); diff = np.empty((T * S, K), np.float32); err = np.empty((T * S, K), np.float32)
    rc = lib.ref_cuda_ctc_eval_parallel(_p(probs), T, S, K, _p(lens), _p(ids), _p(off), _p(alpha), _p(beta), _p(pzx),
                                        _p(diff), _p(err), L)
    assert rc == L, rc
    return dict(alpha=alpha, beta=beta, pzx=pzx, diff=diff, ctc_err=err, L=L)


def cuda_ctc_eval(probs: np.ndarray, label):
    """The reference's SINGLE-sequence CTC, Ctc::Eval (ctc-loss.cc:28-75) through its one-sequence CUDA kernel bodies:
    what train-ctc computes per utterance.  probs [T x K], label [U]."""
    lib = _load()
    probs = np.ascontiguousarray(probs, np.float32)
    T, K = probs.shape
    label = np.ascontiguousarray(label, np.int32)
    L = 2 * label.size + 1
    alpha = np.empty((T, L), np.float32); beta = np.empty((T, L), np.float32)
    pzx = np.empty(1, np.float32); diff = np.empty((T, K), np.float32)
    rc = lib.ref_cuda_ctc_eval(_p(probs), T, K, _p(label), int(label.size), _p(alpha), _p(beta), _p(pzx), _p(diff))
    assert rc == L, rc
    return dict(alpha=alpha, beta=beta, pzx=float(pzx[0]), diff=diff)


def cuda_activation(name: str, x: np.ndarray) -> np.ndarray:
    lib = _load()
    x = np.ascontiguousarray(x, np.float32).reshape(1, -1)
    y = np.empty_like(x)
    getattr(lib, f"ref_cuda_{name}")(_p(y), _p(x), 1, x.shape[1])
    return y.ravel()


def cuda_adaptive_update(param: np.ndarray, corr: np.ndarray, accu: np.ndarray, lr: float, eps: float, rho: float, rmsprop: bool):
    """The reference's Adagrad / RMSProp tensor update through its own elementwise CUDA kernel bodies (in place on
    param and accu; corr must already hold the clipped momentum-folded gradient)."""
    lib = _load()
    assert param.dtype == np.float32 and accu.dtype == np.float32 and param.flags.c_contiguous and accu.flags.c_contiguous
    corr = np.ascontiguousarray(corr, np.float32)
    rows, cols = (param.shape if param.ndim == 2 else (1, param.size))
    r = np.float32(rho)
    lib.ref_cuda_adaptive_update(_p(param), _p(corr), _p(accu), rows, cols, C.c_float(lr), C.c_float(eps), C.c_float(r),
                                 C.c_float(np.float32(1.0) - r), int(rmsprop))
