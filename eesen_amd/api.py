"""Host-side mirror of the reference's operator interface for the hot path, over the C-ABI.

Same names, argument meaning and error behaviour as the classes `train-ctc-parallel` drives
(/root/reference/src/netbin/train-ctc-parallel.cc:111-119,195-207):

  Net  : Read, Write, SetTrainOptions, SetSeqLengths, InputDim, OutputDim, NumParams, GetParams, SetParams,
         Propagate, Backpropagate                      (/root/reference/src/net/net.h:48-161)
  Ctc  : EvalParallel, ErrorRateMSeq, Report, NumErrorTokens, NumRefTokens
                                                       (/root/reference/src/net/ctc-loss.h:40-63)
  Feeder : device-side minibatch assembly (train-ctc-parallel.cc:186-195), double-buffered
  CuMatrix : a [rows x cols] fp32 device matrix with a stride, the stand-in for CuMatrix<BaseFloat>
                                                       (/root/reference/src/gpucompute/cuda-matrix.h)

Errors surface as EesenError (the reference throws std::runtime_error from KALDI_ERR).  Everything here
is plumbing: all arithmetic happens in libeesen_hip.so.  No torch import — device memory comes from the
library's own allocator, so the product path has no framework dependency.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import EesenError, check


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class CuMatrix:
    """Row-major fp32 device matrix {rows, cols, stride} (/root/reference/src/gpucompute/cuda-matrixdim.h:52-56).

    Owns its memory when created with `CuMatrix(rows, cols)`; `CuMatrix.view(ptr, ...)` wraps memory owned
    by a Net (Propagate output).  Rows are padded to a multiple of 4 floats so they stay 16-byte aligned.
    """

    def __init__(self, rows: int, cols: int, device: int = 0, zero: bool = True):
        self.rows, self.cols, self.device = int(rows), int(cols), device
        self.stride = (self.cols + 3) & ~3
        self._own = True
        p = C.c_void_p()
        check(_lib.load().eesen_dev_alloc(device, max(1, self.rows * self.stride) * 4, C.byref(p)))
        self.ptr = p.value
        if zero and rows * cols:
            z = np.zeros(self.rows * self.stride, np.float32)
            check(_lib.load().eesen_dev_copy(device, C.c_void_p(self.ptr), _np_ptr(z), z.nbytes, 1))

    @classmethod
    def view(cls, ptr: int, rows: int, cols: int, stride: int, device: int = 0, keepalive=None) -> "CuMatrix":
        m = cls.__new__(cls)
        m.ptr, m.rows, m.cols, m.stride, m.device, m._own, m._keep = ptr, rows, cols, stride, device, False, keepalive
        return m

    @classmethod
    def from_numpy(cls, a: np.ndarray, device: int = 0) -> "CuMatrix":
        a = np.ascontiguousarray(a, np.float32)
        m = cls(a.shape[0], a.shape[1], device, zero=False)
        buf = np.zeros((m.rows, m.stride), np.float32)
        buf[:, : m.cols] = a
        check(_lib.load().eesen_dev_copy(device, C.c_void_p(m.ptr), _np_ptr(buf), buf.nbytes, 1))
        return m

    def NumRows(self) -> int:
        return self.rows

    def NumCols(self) -> int:
        return self.cols

    def Stride(self) -> int:
        return self.stride

    def numpy(self) -> np.ndarray:
        """CopyToMat: device -> host, dense [rows x cols]. Synchronous."""
        buf = np.empty((self.rows, self.stride), np.float32)
        if buf.size:
            check(_lib.load().eesen_dev_copy(self.device, _np_ptr(buf), C.c_void_p(self.ptr), buf.nbytes, 2))
        return np.ascontiguousarray(buf[:, : self.cols])

    def __del__(self):
        if getattr(self, "_own", False) and getattr(self, "ptr", None):
            try:
                _lib.load().eesen_dev_free(self.device, C.c_void_p(self.ptr))
            except Exception:
                pass
            self.ptr = None


class Comm:
    """One rank of the data-parallel job: an RCCL communicator owned by libeesen_hip.so (include/eesen_hip.h
    `eesen_comm_*`), the replacement of the reference's file-based comm_avg_weights / comm_touch_done
    (/root/reference/src/net/communicator.h:39-170).  No torch involved: rank 0 hands the RCCL unique id to the other
    ranks over a plain TCP connection on MASTER_ADDR : (EESEN_COMM_PORT | MASTER_PORT + 17)."""

    SUM, MAX = 0, 1

    def __init__(self, device: int, rank: int, world: int, addr: str = "127.0.0.1", port: int = 29517, timeout_s: int = 120):
        self.lib = _lib.load()
        self.device, self.rank, self.world = device, rank, world
        self.h = C.c_void_p()
        check(self.lib.eesen_comm_create_tcp(device, addr.encode(), int(port), rank, world, int(timeout_s), C.byref(self.h)))

    @classmethod
    def from_env(cls, device: Optional[int] = None, timeout_s: int = 120) -> "Comm":
        """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT as torch.distributed.run (or bench.py's own launcher) export them."""
        import os
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        dev = int(os.environ.get("LOCAL_RANK", "0")) if device is None else device
        port = int(os.environ.get("EESEN_COMM_PORT", 0)) or int(os.environ.get("MASTER_PORT", "29500")) + 17
        return cls(dev, rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"), port, timeout_s)

    def __del__(self):
        if getattr(self, "h", None) and self.h:
            try:
                self.lib.eesen_comm_destroy(self.h)
            except Exception:
                pass
            self.h = None

    def allreduce(self, values: Sequence[float], op: int = 0) -> List[float]:
        """Sum (or max) of up to 64 host scalars over the ranks; blocks (also serves as a barrier)."""
        a = (C.c_double * len(values))(*[float(v) for v in values])
        check(self.lib.eesen_comm_allreduce_host(self.h, a, len(values), int(op)))
        return list(a)

    def barrier(self):
        self.allreduce([0.0])

    def Describe(self) -> dict:
        """COLLECTIVE (every rank calls it): the library the linker resolved, its version, whether it is the tests' stand-in, the
        ranks and device the library itself reports, every rank's PCI bus id, distinct_devices (eesen_comm_describe)."""
        import json
        buf = C.create_string_buffer(16384)
        check(self.lib.eesen_comm_describe(self.h, buf, len(buf)))
        return json.loads(buf.value.decode())


class Net:
    """eesen::Net for BiLstmParallel / LstmParallel / AffineTransform / Softmax stacks."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.lib = _lib.load()
        self.device = device
        self._stream = stream
        self.h = C.c_void_p()
        check(self.lib.eesen_net_create(device, C.c_void_p(stream) if stream else None, C.byref(self.h)))
        self._out = None
        self.grad_hook = None  # callable(net) run between backprop and update: the data-parallel exchange

    def __del__(self):
        if getattr(self, "h", None) and self.h:
            try:
                self.lib.eesen_net_destroy(self.h)
            except Exception:
                pass
            self.h = None

    # ---- model I/O ---------------------------------------------------------------------------
    def Read(self, path: str):
        """Net::Read (net.cc:279-309). Learn rate is reset to 0: call SetTrainOptions afterwards."""
        check(self.lib.eesen_net_read(self.h, path.encode()))
        return self

    def Write(self, path: str, binary: bool = True):
        check(self.lib.eesen_net_write(self.h, path.encode(), int(binary)))

    @classmethod
    def from_layers(cls, layers: List[dict], device: int = 0, stream: Optional[int] = None) -> "Net":
        """Build from the dict form of eesen_amd.nnet_io (type, input_dim, output_dim, learn_rate_coef, max_grad, params)."""
        net = cls(device, stream)
        for L in layers:
            check(net.lib.eesen_net_add_layer(net.h, _lib.KIND_OF[L["type"]], int(L["input_dim"]), int(L["output_dim"]),
                                              float(L.get("learn_rate_coef", 1.0)), float(L.get("max_grad", 0.0))))
        check(net.lib.eesen_net_finalize(net.h))
        flat = [np.asarray(p, np.float32).ravel() for L in layers for p in L["params"]]
        if flat:
            net.SetParams(np.concatenate(flat))
        for i, L in enumerate(layers):
            if L.get("dropout"):
                net.SetLayerDropout(i, L["dropout"])
        return net

    # ---- dropout (bilstm-parallel-layer.h:46-94) ------------------------------------------------
    def SetTrainMode(self):
        """Net::SetTrainMode (net.cc:405-412): dropout layers draw and apply their masks (the default)."""
        check(self.lib.eesen_net_set_train_mode(self.h, 1))

    def SetTestMode(self):
        """Net::SetTestMode (net.cc:396-403): no dropout."""
        check(self.lib.eesen_net_set_train_mode(self.h, 0))

    def SetLayerDropout(self, layer: int, opts: dict):
        """opts: keys of eesen_amd.nnet_io.DROPOUT_KEYS (forward, fw_step, fw_seq, rec_step, rec_seq, rnndrop, nml, recurrent, twiddle)."""
        from .nnet_io import dropout_values
        v = np.array([float(x) for x in dropout_values({"dropout": opts})], np.float32)
        check(self.lib.eesen_net_set_layer_dropout(self.h, layer, v.ctypes.data_as(C.POINTER(C.c_float))))

    def GetLayerDropout(self, layer: int) -> dict:
        from .nnet_io import DROPOUT_KEYS
        v = np.zeros(9, np.float32)
        check(self.lib.eesen_net_get_layer_dropout(self.h, layer, v.ctypes.data_as(C.POINTER(C.c_float))))
        return {k: (float(x) if k in ("forward", "recurrent") else bool(x)) for k, x in zip(DROPOUT_KEYS, v) if x}

    def SetDropoutSeed(self, seed: int):
        check(self.lib.eesen_net_set_dropout_seed(self.h, C.c_ulonglong(seed)))

    def SetDropoutMasks(self, layer: int, fwd=None, rec=None, twiddle_coin: int = -1):
        """Masks for the NEXT Propagate (parity tests): fwd [T*S x 2H]; rec [(T+2)*S x 2H] or [S x 2H] (fw columns, then bw)."""
        f = None if fwd is None else np.ascontiguousarray(fwd, np.float32)
        r = None if rec is None else np.ascontiguousarray(rec, np.float32)
        check(self.lib.eesen_net_set_dropout_masks(self.h, layer, _np_ptr(f) if f is not None else None, 0 if f is None else f.size,
                                                   _np_ptr(r) if r is not None else None, 0 if r is None else r.shape[0],
                                                   0 if r is None else r.size, int(twiddle_coin)))

    def GetDropoutMasks(self, layer: int, T: int, S: int) -> dict:
        """What the last Propagate applied to `layer`: dict(fwd [T*S x 2H] | None, rec [(T+2)*S x 2H] | None, mode, coin)."""
        info = (C.c_int * 4)()
        check(self.lib.eesen_net_get_dropout_masks(self.h, layer, None, None, info))
        w = info[3]
        fwd = np.zeros((T * S, w), np.float32) if info[0] else None
        rec = np.zeros(((T + 2) * S, w), np.float32) if info[1] else None
        check(self.lib.eesen_net_get_dropout_masks(self.h, layer, _np_ptr(fwd) if fwd is not None else None,
                                                   _np_ptr(rec) if rec is not None else None, info))
        return dict(fwd=fwd, rec=rec, mode=int(info[1]), coin=bool(info[2]))

    def layers(self) -> List[dict]:
        n = C.c_int()
        check(self.lib.eesen_net_num_layers(self.h, C.byref(n)))
        out = []
        for i in range(n.value):
            k, di, do, cf, mg = C.c_int(), C.c_int(), C.c_int(), C.c_float(), C.c_float()
            check(self.lib.eesen_net_layer_info(self.h, i, C.byref(k), C.byref(di), C.byref(do), C.byref(cf), C.byref(mg)))
            out.append(dict(type=_lib.NAME_OF[k.value], input_dim=di.value, output_dim=do.value,
                            learn_rate_coef=cf.value, max_grad=mg.value))
            dr = self.GetLayerDropout(i)
            if dr:
                out[-1]["dropout"] = dr
        return out

    def InputDim(self) -> int:
        d = C.c_int()
        check(self.lib.eesen_net_input_dim(self.h, C.byref(d)))
        return d.value

    def OutputDim(self) -> int:
        d = C.c_int()
        check(self.lib.eesen_net_output_dim(self.h, C.byref(d)))
        return d.value

    def NumParams(self) -> int:
        n = C.c_long()
        check(self.lib.eesen_net_num_params(self.h, C.byref(n)))
        return n.value

    def GetParams(self) -> np.ndarray:
        out = np.empty(self.NumParams(), np.float32)
        check(self.lib.eesen_net_get_params(self.h, _np_ptr(out), out.size))
        return out

    def SetParams(self, flat: np.ndarray):
        flat = np.ascontiguousarray(flat, np.float32)
        check(self.lib.eesen_net_set_params(self.h, _np_ptr(flat), flat.size))

    def GetGrads(self) -> np.ndarray:
        """Fresh gradients of the last Backpropagate in GetParams order (parity accessor)."""
        out = np.empty(self.NumParams(), np.float32)
        check(self.lib.eesen_net_get_grads(self.h, _np_ptr(out), out.size))
        return out

    # ---- options -----------------------------------------------------------------------------
    def SetTrainOptions(self, learn_rate: float, momentum: float = 0.0):
        check(self.lib.eesen_net_set_train_options(self.h, float(learn_rate), float(momentum)))

    def SetUpdateAlgorithm(self, name: str, adagrad_epsilon: float = 1e-6, rmsprop_rho: float = 0.9):
        """Net::SetUpdateAlgorithm (net.cc:481-497) + the adaptive hyper-parameters of NetTrainOptions."""
        check(self.lib.eesen_net_set_update_algorithm(self.h, name.encode()))
        check(self.lib.eesen_net_set_adaptive_options(self.h, float(adagrad_epsilon), float(rmsprop_rho)))

    def GetAccumulators(self) -> np.ndarray:
        out = np.empty(self.NumParams(), np.float32)
        check(self.lib.eesen_net_get_accumulators(self.h, _np_ptr(out), out.size))
        return out

    def SetAccumulators(self, flat: np.ndarray):
        flat = np.ascontiguousarray(flat, np.float32)
        check(self.lib.eesen_net_set_accumulators(self.h, _np_ptr(flat), flat.size))

    def SetSeqLengths(self, lens: Sequence[int]):
        a = np.ascontiguousarray(lens, np.int32)
        check(self.lib.eesen_net_set_seq_lengths(self.h, _np_ptr(a), a.size))

    # ---- compute -----------------------------------------------------------------------------
    def Propagate(self, feats) -> CuMatrix:
        """Net::Propagate (net.cc:67-86). feats: host ndarray [T*S x InputDim] (uploaded, as the reference's
        CuMatrix ctor does) or a CuMatrix.  Returns a view of the net-owned output, valid until the next call."""
        p, co, ld = C.c_void_p(), C.c_int(), C.c_int()
        if isinstance(feats, CuMatrix):
            check(self.lib.eesen_net_propagate(self.h, C.c_void_p(feats.ptr), feats.rows, feats.stride, 1, C.byref(p), C.byref(co), C.byref(ld)))
            rows = feats.rows
        else:
            a = np.ascontiguousarray(feats, np.float32)
            if a.ndim != 2:
                raise EesenError(-1, "Propagate expects a [rows x dim] matrix")
            check(self.lib.eesen_net_propagate(self.h, _np_ptr(a), a.shape[0], a.shape[1], 0, C.byref(p), C.byref(co), C.byref(ld)))
            rows = a.shape[0]
        self._out = CuMatrix.view(p.value, rows, co.value, ld.value, self.device, keepalive=self)
        return self._out

    def BackpropagateNoUpdate(self, out_diff: CuMatrix, in_diff: Optional[CuMatrix] = None):
        check(self.lib.eesen_net_backpropagate(self.h, C.c_void_p(out_diff.ptr), out_diff.stride,
                                               C.c_void_p(in_diff.ptr) if in_diff is not None else None,
                                               in_diff.stride if in_diff is not None else 0))

    def Update(self):
        check(self.lib.eesen_net_update(self.h))

    def Backpropagate(self, out_diff: CuMatrix, in_diff: Optional[CuMatrix] = None):
        """Net::Backpropagate (net.cc:88-108): gradients, [data-parallel exchange], per-layer Update."""
        self.BackpropagateNoUpdate(out_diff, in_diff)
        if self.grad_hook is not None:
            self.grad_hook(self)
        self.Update()

    def SetComm(self, comm: Optional[Comm]):
        """Attach the data-parallel communicator: Backpropagate then all-reduces every layer's fresh gradients (one bucket per
        layer, on the communicator's stream, under the lower layers' backward pass) and Update waits bucket by bucket."""
        check(self.lib.eesen_net_set_comm(self.h, comm.h if comm is not None else None))
        self._comm = comm

    def AllReduceGrads(self, comm: Comm):
        """The bulk form: one all-reduce of the whole gradient buffer, between BackpropagateNoUpdate and Update."""
        check(self.lib.eesen_net_allreduce_grads(self.h, comm.h))

    def BackpropagateZero(self):
        """This rank has no minibatch this step (others do): zero gradient through the same collectives; then Update()."""
        check(self.lib.eesen_net_backpropagate_zero(self.h))

    def LiveRanks(self) -> int:
        """How many ranks had a minibatch in the step issued last (the liveness word that rides with the top layer's gradient
        bucket).  For a rank in the zero-gradient protocol: 0 = every rank is out of data.  Blocks until that bucket has arrived."""
        n = C.c_int()
        check(self.lib.eesen_net_live_ranks(self.h, C.byref(n)))
        return n.value

    def LayerMarker(self, idx: int) -> str:
        buf = C.create_string_buffer(64)
        check(self.lib.eesen_net_layer_marker(self.h, idx, buf, 64))
        return buf.value.decode()

    def TensorMoments(self, which: int, layer: int) -> np.ndarray:
        """[n_tensors x 6] = (min, max, mean, variance, skewness, kurtosis) per tensor of `layer`, in the order of the reference
        layer's Info(); which: 0 parameters, 1 momentum buffers (*_corr_), 2 adaptive accumulators."""
        n = C.c_int()
        check(self.lib.eesen_net_tensor_moments(self.h, which, layer, None, 0, C.byref(n)))
        out = np.zeros((n.value, 6), np.float64)
        if n.value:
            check(self.lib.eesen_net_tensor_moments(self.h, which, layer, out.ctypes.data_as(C.POINTER(C.c_double)), n.value, C.byref(n)))
        return out

    def Info(self) -> str:
        """Net::Info (net.cc:336-354): topology + MomentStatistics of every parameter tensor."""
        return _net_info(self, 0)

    def InfoGradient(self) -> str:
        """Net::InfoGradient (net.cc:356-366): MomentStatistics of every momentum buffer (*_corr_)."""
        return _net_info(self, 1)

    def BucketOrder(self) -> List[int]:
        buf = (C.c_int * 64)()
        n = C.c_int()
        check(self.lib.eesen_net_bucket_order(self.h, buf, 64, C.byref(n)))
        return list(buf[: n.value])

    def grad_buffer(self):
        """(device pointer, float count) of the contiguous fresh-gradient buffer (all-reduce payload)."""
        p, n = C.c_void_p(), C.c_long()
        check(self.lib.eesen_net_grad_buffer(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def SetForwardPrecision(self, bf16):
        """BASELINE config 4's bf16 forward (eesen_net_set_forward_precision): True / 1 = forward GEMMs AND the forward time
        recurrence on bf16 operands with fp32 accumulation; 2 = the GEMMs only; False / 0 = fp32.  Everything else stays fp32."""
        check(self.lib.eesen_net_set_forward_precision(self.h, int(bf16)))

    def Bf16RecurrenceLayers(self) -> int:
        """LSTM layers of the last Propagate whose recurrence ran on the bf16 kernel (debug accessor)."""
        n = C.c_int()
        check(self.lib.eesen_net_bf16_recurrence_layers(self.h, C.byref(n)))
        return n.value

    def RecurrenceInfo(self) -> dict:
        """Which recurrence kernels the last Propagate / Backpropagate used (debug accessor)."""
        a = (C.c_int * 4)()
        check(self.lib.eesen_net_recurrence_info(self.h, a))
        self.recoveries = a[3]
        return dict(lstm_layers=a[0], fwd_persistent=a[1], bwd_persistent=a[2])

    def Plan(self) -> dict:
        """What will run the current minibatch shape (after SetSeqLengths): the recurrence plans of every LSTM layer -- the very ones
        the launchers execute -- and the schedule decisions read from them (eesen_net_plan_string)."""
        import json
        buf = C.create_string_buffer(32768)
        check(self.lib.eesen_net_plan_string(self.h, buf, len(buf)))
        return json.loads(buf.value.decode())

    def _raise_error_word(self, value: int):
        """Test hook: what a recurrence kernel (1) / the milestone waiter (2) stores when it gives up."""
        check(self.lib.eesen_net_debug_set_error_word(self.h, int(value)))

    def Synchronize(self):
        check(self.lib.eesen_net_synchronize(self.h))

    def SetProfiling(self, on, accumulate: bool = False):
        """HIP-event phase timers: per step (read after every step), or accumulated over several steps until PhaseTimes()."""
        check(self.lib.eesen_net_set_profiling(self.h, 2 if (on and accumulate) else int(bool(on))))

    def PhaseSpans(self):
        """[(phase name, seconds)] of every timed span since the last PhaseTimes(), in record order (call before PhaseTimes)."""
        names = ["input_gemm", "recurrence_fwd", "affine_softmax", "recurrence_bwd", "grad_gemm", "update", "allreduce", "allreduce_exposed"]
        n = C.c_int()
        check(self.lib.eesen_net_get_phase_spans(self.h, None, None, 0, C.byref(n)))
        ph, se = (C.c_int * max(n.value, 1))(), (C.c_float * max(n.value, 1))()
        check(self.lib.eesen_net_get_phase_spans(self.h, ph, se, n.value, C.byref(n)))
        return [(names[ph[i]] if 0 <= ph[i] < len(names) else str(ph[i]), float(se[i])) for i in range(n.value)]

    def PhaseTimes(self) -> dict:
        out = np.zeros(6, np.float32)
        check(self.lib.eesen_net_get_phase_times(self.h, _np_ptr(out)))
        return dict(zip(["input_gemm", "recurrence_fwd", "affine_softmax", "recurrence_bwd", "grad_gemm", "update"], out.tolist()))


_LSTM_TENSORS = ["wei_gifo_x", "wei_gifo_m", "bias", "phole_i_c", "phole_f_c", "phole_o_c"]


def _g(v: float) -> str:
    return f"{v:g}"     # what operator<< prints for a float by default (6 significant digits)


def _net_info(net: "Net", which: int) -> str:
    """The strings of Net::Info / Net::InfoGradient (net.cc:336-366) with the per-layer parts of bilstm-layer.h:496-560,
    lstm-layer.h:175-196, affine-trans-layer.h:145-159."""
    layers = net.layers()
    out = []
    if which == 0:
        out += [f"num-layers {len(layers)}", f"input-dim {net.InputDim()}", f"output-dim {net.OutputDim()}",
                f"number-of-parameters {_g(net.NumParams() / 1e6)} millions"]
    else:
        out.append("### Gradient stats :")
    for i, L in enumerate(layers):
        m = net.TensorMoments(which, i)
        stats = [f" ( min {_g(r[0])}, max {_g(r[1])}, mean {_g(r[2])}, variance {_g(r[3])}, skewness {_g(r[4])}, kurtosis {_g(r[5])} ) "
                 for r in m]
        suf = "" if which == 0 else "corr_"
        body = ""
        if L["type"] in ("BiLstmParallel", "LstmParallel"):
            dirs = ["_fw_", "_bw_"] if L["type"] == "BiLstmParallel" else ["_"]
            names = [f"{t}{d}{suf}" for d in dirs for t in _LSTM_TENSORS]
            body = "    " + "".join(f"\n  {nm}  {st}" for nm, st in zip(names, stats))
        elif L["type"] == "AffineTransform":
            names = ["linearity", "bias"] if which == 0 else ["linearity_corr_", "bias_corr_"]
            body = "".join(f"\n  {nm}{st}" for nm, st in zip(names, stats))
        marker = net.LayerMarker(i)
        if which == 0:
            out.append(f"layer {i + 1} : {marker}, input-dim {L['input_dim']}, output-dim {L['output_dim']}, {body}")
        else:
            out.append(f"Layer {i + 1} : {marker}, {body}")
    return "\n".join(out) + "\n"


class Ctc:
    """eesen::Ctc (ctc-loss.h:31-90) for the multi-sequence path."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.lib = _lib.load()
        self.device = device
        self.h = C.c_void_p()
        check(self.lib.eesen_ctc_create(device, C.c_void_p(stream) if stream else None, C.byref(self.h)))
        self.pzx = None

    def __del__(self):
        if getattr(self, "h", None) and self.h:
            try:
                self.lib.eesen_ctc_destroy(self.h)
            except Exception:
                pass
            self.h = None

    @staticmethod
    def _csr(label: Sequence[Sequence[int]]):
        off = np.zeros(len(label) + 1, np.int32)
        off[1:] = np.cumsum([len(l) for l in label])
        ids = np.concatenate([np.asarray(l, np.int32) for l in label]) if off[-1] else np.zeros(1, np.int32)
        return np.ascontiguousarray(ids, np.int32), off

    def EvalParallel(self, frame_num_utt: Sequence[int], net_out: CuMatrix, label: Sequence[Sequence[int]],
                     diff: Optional[CuMatrix] = None, want_pzx: bool = True) -> CuMatrix:
        """Ctc::EvalParallel (ctc-loss.cc:101-194). Returns diff (allocated when not given); self.pzx = ln p per sequence.
        want_pzx=False: nothing waits for the device -- ln p joins the objective sum (stats()) when it has arrived, as the
        reference's own call returns nothing but `diff` and only accumulates obj_progress_ (:171-177); self.pzx is None then."""
        fn = np.ascontiguousarray(frame_num_utt, np.int32)
        ids, off = self._csr(label)
        if diff is None:
            diff = CuMatrix(net_out.rows, net_out.cols, self.device, zero=False)
        pzx = np.empty(fn.size, np.float32) if want_pzx else None
        check(self.lib.eesen_ctc_eval_parallel(self.h, _np_ptr(fn), fn.size, C.c_void_p(net_out.ptr), net_out.rows, net_out.cols,
                                               net_out.stride, _np_ptr(ids), _np_ptr(off), C.c_void_p(diff.ptr), diff.stride,
                                               _np_ptr(pzx) if want_pzx else None))
        self.pzx = pzx
        return diff

    def ErrorRateMSeq(self, frame_num_utt: Sequence[int], net_out: CuMatrix, label: Sequence[Sequence[int]], deferred: bool = False):
        """Ctc::ErrorRateMSeq (ctc-loss.cc:235-298): accumulates the error / reference token counts.  Returns this call's
        (errors, refs); with deferred=True only the argmax and the copy of the ids are enqueued and the host part (collapse +
        edit distance) runs at the next call / stats(), under the device's backward pass -- the reference's call returns void."""
        fn = np.ascontiguousarray(frame_num_utt, np.int32)
        ids, off = self._csr(label)
        ne, nr = C.c_int(), C.c_int()
        check(self.lib.eesen_ctc_error_rate_mseq(self.h, _np_ptr(fn), fn.size, C.c_void_p(net_out.ptr), net_out.rows, net_out.cols,
                                                 net_out.stride, _np_ptr(ids), _np_ptr(off), None if deferred else C.byref(ne),
                                                 None if deferred else C.byref(nr)))
        return None if deferred else (ne.value, nr.value)

    def stats(self) -> dict:
        o, s, f, e, r = C.c_double(), C.c_long(), C.c_long(), C.c_long(), C.c_long()
        check(self.lib.eesen_ctc_stats(self.h, C.byref(o), C.byref(s), C.byref(f), C.byref(e), C.byref(r)))
        return dict(obj_sum=o.value, sequences=s.value, frames=f.value, err_tokens=e.value, ref_tokens=r.value)

    def NumErrorTokens(self) -> int:
        return self.stats()["err_tokens"]

    def NumRefTokens(self) -> int:
        return self.stats()["ref_tokens"]

    def Report(self) -> str:
        """Ctc::Report (ctc-loss.cc:300-304); train_ctc_parallel.sh greps this line."""
        st = self.stats()
        acc = 100.0 * (1.0 - st["err_tokens"] / st["ref_tokens"]) if st["ref_tokens"] else float("nan")
        return f"\nTOKEN_ACCURACY >> {acc:g}% <<"

    def alpha_beta(self):
        """(alpha, beta) of the last EvalParallel in the reference's [T*S x L'] layout (parity accessor)."""
        L = C.c_int()
        check(self.lib.eesen_ctc_get_alpha_beta(self.h, None, None, C.byref(L)))
        rows = self._rows
        a = np.empty((rows, L.value), np.float32)
        b = np.empty((rows, L.value), np.float32)
        check(self.lib.eesen_ctc_get_alpha_beta(self.h, _np_ptr(a), _np_ptr(b), C.byref(L)))
        return a, b

    def SetGuard(self, net: Optional[Net]):
        """Drop minibatches computed from a timed-out persistent forward pass of `net` from the statistics (eesen_ctc_set_guard)."""
        check(self.lib.eesen_ctc_set_guard(self.h, net.h if net is not None else None))
        self._guard = net

    def Dropped(self) -> int:
        n = C.c_long()
        check(self.lib.eesen_ctc_dropped(self.h, C.byref(n)))
        return n.value

    def SetSequenceOutFile(self, path: Optional[str]):
        """--sequence-out-file (train-ctc-parallel.cc:53-54,134-137): ErrorRateMSeq appends `utt | label frame prob | ...` lines."""
        check(self.lib.eesen_ctc_set_sequence_out_file(self.h, path.encode() if path else None))

    def SetProfiling(self, accumulate: bool):
        """accumulate=True: PhaseTimes() returns the sums over all EvalParallel calls since the last read (no per-call sync)."""
        check(self.lib.eesen_ctc_set_profiling(self.h, 2 if accumulate else 0))

    def PhaseTimes(self) -> dict:
        out = np.zeros(3, np.float32)
        check(self.lib.eesen_ctc_get_phase_times(self.h, _np_ptr(out)))
        return dict(zip(["log", "alpha_beta", "error_diff"], out.tolist()))


class Feeder:
    """Device-side minibatch assembly, double-buffered (include/eesen_hip.h `eesen_feeder_*`; replaces the host padding +
    interleave + blocking copy of /root/reference/src/netbin/train-ctc-parallel.cc:186-195).

        slot = feeder.submit(mats)          # S host matrices [T_s x D]; returns at once, copy + interleave run async
        feats = feeder.acquire(slot)        # CuMatrix view [T*S x D], row t*S + s; the compute stream waits on-device
        out = net.Propagate(feats); feeder.release(slot)
    """

    def __init__(self, device: int = 0, stream: Optional[int] = None, slots: int = 2):
        self.lib = _lib.load()
        self.device = device
        self.h = C.c_void_p()
        check(self.lib.eesen_feeder_create(device, C.c_void_p(stream) if stream else None, slots, C.byref(self.h)))
        self._dims = {}

    def __del__(self):
        if getattr(self, "h", None) and self.h:
            try:
                self.lib.eesen_feeder_destroy(self.h)
            except Exception:
                pass
            self.h = None

    def set_pipeline(self, stages: Sequence[Tuple[int, int, int]]):
        """The feature filters of the recipes' rspecifier pipe, run on the device between the PCIe copy and the interleave
        (`eesen_feeder_set_pipeline`; stages = [(EESEN_FEAT_*, a, b)], see eesen_amd.frontend).  Empty list: none."""
        arr = (C.c_int * (3 * max(len(stages), 1)))()
        for i, (k, a, b) in enumerate(stages):
            arr[3 * i], arr[3 * i + 1], arr[3 * i + 2] = int(k), int(a), int(b)
        check(self.lib.eesen_feeder_set_pipeline(self.h, C.cast(arr, C.c_void_p), len(stages)))
        self._has_pipeline = len(stages) > 0

    def pipeline_shape(self, D_in: int, frames_in: int) -> Tuple[int, int]:
        """(frames, feature dimension) behind the pipeline for an utterance of [frames_in x D_in]."""
        d, t = C.c_int(), C.c_int()
        check(self.lib.eesen_feeder_pipeline_shape(self.h, int(D_in), int(frames_in), C.byref(d), C.byref(t)))
        return t.value, d.value

    def submit_raw(self, mats: Sequence[np.ndarray], cmvn: Optional[Sequence[np.ndarray]] = None) -> int:
        """RAW utterance matrices [T_s x D_in] (+ per utterance the [2 x Dc] CMVN offsets / scales when the pipeline has a
        CMVN stage); the assembled batch is [T*S x D_out] with T = the longest utterance behind the pipeline."""
        if not len(mats):
            raise EesenError(-1, "empty minibatch")
        mats = [np.ascontiguousarray(m, np.float32) for m in mats]
        D = mats[0].shape[1]
        for m in mats:
            if m.ndim != 2 or m.shape[1] != D:
                raise EesenError(-1, f"feature dimension {m.shape[1] if m.ndim == 2 else m.shape} does not match {D}")
        S = len(mats)
        ptrs = (C.c_void_p * S)(*[m.ctypes.data for m in mats])
        frames = np.array([m.shape[0] for m in mats], np.int32)
        cptr = None
        if cmvn is not None:
            if len(cmvn) != S:
                raise EesenError(-1, "one CMVN vector pair per utterance")
            cmvn = [np.ascontiguousarray(c, np.float32) for c in cmvn]
            for c in cmvn:
                if c.ndim != 2 or c.shape[0] != 2 or c.shape[1] != cmvn[0].shape[1]:
                    raise EesenError(-1, "CMVN vectors must be [2 x dim] (offsets, scales)")
            cptr = (C.c_void_p * S)(*[c.ctypes.data for c in cmvn])
        slot = C.c_int()
        check(self.lib.eesen_feeder_submit_raw(self.h, ptrs, frames.ctypes.data_as(C.POINTER(C.c_int)), None, cptr, S, D, C.byref(slot)))
        self._dims[slot.value] = self.pipeline_shape(D, 1)[1]
        return slot.value

    def submit(self, mats: Sequence[np.ndarray]) -> int:
        if not len(mats):
            raise EesenError(-1, "empty minibatch")
        if hasattr(mats[0], "raw"):     # eesen_amd.frontend.RawUtt: raw matrix + CMVN vectors, shape = behind the pipeline
            return self.submit_raw([m.raw for m in mats], [m.cmvn for m in mats] if mats[0].cmvn is not None else None)
        mats = [m if (m.dtype == np.float32 and m.ndim == 2 and m.strides[1] == 4 and m.strides[0] % 4 == 0 and m.strides[0] >= 4 * m.shape[1])
                else np.ascontiguousarray(m, np.float32) for m in mats]
        D = mats[0].shape[1]
        for m in mats:
            if m.ndim != 2 or m.shape[1] != D:
                raise EesenError(-1, f"feature dimension {m.shape[1] if m.ndim == 2 else m.shape} does not match {D}")
        S = len(mats)
        ptrs = (C.c_void_p * S)(*[m.ctypes.data for m in mats])
        frames = np.array([m.shape[0] for m in mats], np.int32)
        strides = np.array([m.strides[0] // 4 for m in mats], np.int32)
        slot = C.c_int()
        check(self.lib.eesen_feeder_submit(self.h, ptrs, frames.ctypes.data_as(C.POINTER(C.c_int)), strides.ctypes.data_as(C.POINTER(C.c_int)),
                                           S, D, C.byref(slot)))
        self._dims[slot.value] = D
        return slot.value

    def acquire(self, slot: int) -> CuMatrix:
        p, T, S, ld = C.c_void_p(), C.c_int(), C.c_int(), C.c_int()
        check(self.lib.eesen_feeder_acquire(self.h, slot, C.byref(p), C.byref(T), C.byref(S), C.byref(ld)))
        return CuMatrix.view(p.value, T.value * S.value, self._dims[slot], ld.value, self.device, keepalive=self)

    def release(self, slot: int):
        check(self.lib.eesen_feeder_release(self.h, slot))


def train_step(net: Net, ctc: Ctc, batch, error_rate: bool = False) -> dict:
    """One pass of the trainer's inner loop (train-ctc-parallel.cc:195-207) on a eesen_amd.synth.Batch."""
    net.SetSeqLengths(batch.lens)
    net_out = net.Propagate(batch.feats)
    diff = ctc.EvalParallel(batch.lens, net_out, batch.labels, getattr(ctc, "_diff", None)
                            if getattr(ctc, "_diff", None) is not None and ctc._diff.rows == net_out.rows and ctc._diff.cols == net_out.cols else None)
    ctc._diff = diff
    ctc._rows = net_out.rows
    res = dict(net_out=net_out, diff=diff, pzx=ctc.pzx)
    if error_rate:
        res["errors"] = ctc.ErrorRateMSeq(batch.lens, net_out, batch.labels)
    net.Backpropagate(diff)
    return res
