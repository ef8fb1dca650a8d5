"""ctypes binding of eesen_amd/lib/libeesen_hip.so (the C-ABI of include/eesen_hip.h).

There is NO CPU fallback: if the HIP library is missing or no GPU is visible, the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

_DIR = os.path.dirname(os.path.abspath(__file__))
# EESEN_HIP_LIBRARY: developer override used by scripts/build_variant.py to A/B kernel build flags on the GPU box
LIB_PATH = os.environ.get("EESEN_HIP_LIBRARY") or os.path.join(_DIR, "lib", "libeesen_hip.so")

OK = 0
LAYER_AFFINE, LAYER_SOFTMAX, LAYER_LSTM_PARALLEL, LAYER_BILSTM_PARALLEL, LAYER_SIGMOID, LAYER_TANH = 1, 2, 3, 4, 5, 6
KIND_OF = {"AffineTransform": LAYER_AFFINE, "Softmax": LAYER_SOFTMAX, "LstmParallel": LAYER_LSTM_PARALLEL,
           "BiLstmParallel": LAYER_BILSTM_PARALLEL, "Lstm": LAYER_LSTM_PARALLEL, "BiLstm": LAYER_BILSTM_PARALLEL,
           "Sigmoid": LAYER_SIGMOID, "Tanh": LAYER_TANH}
NAME_OF = {LAYER_AFFINE: "AffineTransform", LAYER_SOFTMAX: "Softmax", LAYER_LSTM_PARALLEL: "LstmParallel",
           LAYER_BILSTM_PARALLEL: "BiLstmParallel", LAYER_SIGMOID: "Sigmoid", LAYER_TANH: "Tanh"}

# every symbol include/eesen_hip.h declares: name -> (restype, argtypes)
_vp, _i, _f, _l = C.c_void_p, C.c_int, C.c_float, C.c_long
_pi, _pf, _pl, _pd = C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_long), C.POINTER(C.c_double)
SIGNATURES = {
    "eesen_last_error": (C.c_char_p, []),
    "eesen_version": (C.c_char_p, []),
    "eesen_device_count": (_i, [_pi]),
    "eesen_net_create": (_i, [_i, _vp, C.POINTER(_vp)]),
    "eesen_net_destroy": (_i, [_vp]),
    "eesen_net_add_layer": (_i, [_vp, _i, _i, _i, _f, _f]),
    "eesen_net_finalize": (_i, [_vp]),
    "eesen_net_read": (_i, [_vp, C.c_char_p]),
    "eesen_net_write": (_i, [_vp, C.c_char_p, _i]),
    "eesen_net_num_layers": (_i, [_vp, _pi]),
    "eesen_net_layer_info": (_i, [_vp, _i, _pi, _pi, _pi, _pf, _pf]),
    "eesen_net_input_dim": (_i, [_vp, _pi]),
    "eesen_net_output_dim": (_i, [_vp, _pi]),
    "eesen_net_num_params": (_i, [_vp, _pl]),
    "eesen_net_get_params": (_i, [_vp, _vp, _l]),
    "eesen_net_set_params": (_i, [_vp, _vp, _l]),
    "eesen_net_set_train_options": (_i, [_vp, _f, _f]),
    "eesen_net_set_update_algorithm": (_i, [_vp, C.c_char_p]),
    "eesen_net_set_adaptive_options": (_i, [_vp, _f, _f]),
    "eesen_net_get_accumulators": (_i, [_vp, _vp, _l]),
    "eesen_net_set_accumulators": (_i, [_vp, _vp, _l]),
    "eesen_net_set_seq_lengths": (_i, [_vp, _vp, _i]),
    "eesen_net_propagate": (_i, [_vp, _vp, _i, _i, _i, C.POINTER(_vp), _pi, _pi]),
    "eesen_net_get_output": (_i, [_vp, _vp, _l]),
    "eesen_net_backpropagate": (_i, [_vp, _vp, _i, _vp, _i]),
    "eesen_net_grad_buffer": (_i, [_vp, C.POINTER(_vp), _pl]),
    "eesen_net_get_grads": (_i, [_vp, _vp, _l]),
    "eesen_net_update": (_i, [_vp]),
    "eesen_net_set_forward_precision": (_i, [_vp, _i]),
    "eesen_net_bf16_recurrence_layers": (_i, [_vp, _vp]),
    "eesen_net_recurrence_info": (_i, [_vp, _pi]),
    "eesen_net_debug_set_error_word": (_i, [_vp, C.c_uint]),
    "eesen_net_synchronize": (_i, [_vp]),
    "eesen_net_set_profiling": (_i, [_vp, _i]),
    "eesen_net_get_phase_times": (_i, [_vp, _vp]),
    "eesen_net_get_phase_spans": (_i, [_vp, _pi, _pf, _i, _pi]),
    "eesen_device_synchronize": (_i, [_i]),
    "eesen_set_gemm_mode": (_i, [_i]),
    "eesen_get_gemm_mode": (_i, [_pi]),
    "eesen_comm_get_unique_id": (_i, [_vp]),
    "eesen_comm_exchange": (_i, [C.c_char_p, _i, _i, _i, _vp, _i, _i]),
    "eesen_comm_create": (_i, [_i, _vp, _i, _i, C.POINTER(_vp)]),
    "eesen_comm_create_tcp": (_i, [_i, C.c_char_p, _i, _i, _i, _i, C.POINTER(_vp)]),
    "eesen_comm_destroy": (_i, [_vp]),
    "eesen_comm_info": (_i, [_vp, _pi, _pi]),
    "eesen_comm_describe": (_i, [_vp, C.c_char_p, _i]),
    "eesen_net_plan_string": (_i, [_vp, C.c_char_p, _i]),
    "eesen_comm_allreduce_host": (_i, [_vp, _pd, _i, _i]),
    "eesen_net_set_comm": (_i, [_vp, _vp]),
    "eesen_net_allreduce_grads": (_i, [_vp, _vp]),
    "eesen_net_backpropagate_zero": (_i, [_vp]),
    "eesen_net_bucket_order": (_i, [_vp, _pi, _i, _pi]),
    "eesen_net_live_ranks": (_i, [_vp, _pi]),
    "eesen_net_layer_marker": (_i, [_vp, _i, C.c_char_p, _i]),
    "eesen_net_tensor_moments": (_i, [_vp, _i, _i, _pd, _i, _pi]),
    "eesen_ctc_set_guard": (_i, [_vp, _vp]),
    "eesen_ctc_dropped": (_i, [_vp, _pl]),
    "eesen_ctc_create": (_i, [_i, _vp, C.POINTER(_vp)]),
    "eesen_ctc_destroy": (_i, [_vp]),
    "eesen_ctc_eval_parallel": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "eesen_ctc_error_rate_mseq": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _pi, _pi]),
    "eesen_ctc_stats": (_i, [_vp, _pd, _pl, _pl, _pl, _pl]),
    "eesen_ctc_get_alpha_beta": (_i, [_vp, _vp, _vp, _pi]),
    "eesen_ctc_set_profiling": (_i, [_vp, _i]),
    "eesen_ctc_set_sequence_out_file": (_i, [_vp, C.c_char_p]),
    "eesen_ctc_get_phase_times": (_i, [_vp, _vp]),
    "eesen_net_set_train_mode": (_i, [_vp, _i]),
    "eesen_net_set_dropout_seed": (_i, [_vp, C.c_ulonglong]),
    "eesen_net_set_layer_dropout": (_i, [_vp, _i, _pf]),
    "eesen_net_get_layer_dropout": (_i, [_vp, _i, _pf]),
    "eesen_net_set_dropout_masks": (_i, [_vp, _i, _vp, _l, _vp, _i, _l, _i]),
    "eesen_net_get_dropout_masks": (_i, [_vp, _i, _vp, _vp, _pi]),
    "eesen_feeder_create": (_i, [_i, _vp, _i, C.POINTER(_vp)]),
    "eesen_feeder_destroy": (_i, [_vp]),
    "eesen_feeder_submit": (_i, [_vp, C.POINTER(_vp), _pi, _pi, _i, _i, _pi]),
    "eesen_feeder_acquire": (_i, [_vp, _i, C.POINTER(_vp), _pi, _pi, _pi]),
    "eesen_feeder_release": (_i, [_vp, _i]),
    "eesen_feeder_set_pipeline": (_i, [_vp, _vp, _i]),
    "eesen_feeder_pipeline_shape": (_i, [_vp, _i, _i, _pi, _pi]),
    "eesen_feeder_submit_raw": (_i, [_vp, C.POINTER(_vp), _pi, _pi, C.POINTER(_vp), _i, _i, _pi]),
    "eesen_cmvn_norm": (_i, [_pd, _i, _i, _i, _pf]),
    "eesen_dev_alloc": (_i, [_i, _l, C.POINTER(_vp)]),
    "eesen_dev_free": (_i, [_i, _vp]),
    "eesen_dev_copy": (_i, [_i, _vp, _vp, _l, _i]),
    "eesen_op_log_sub_prior": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _vp, _f]),
    "eesen_op_amax_rows_cols": (_i, [_i, _vp, C.c_long, _i, _i, _vp, _vp]),
    "eesen_op_gemm_bench": (_i, [_i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _pf]),
    "eesen_op_gemm": (_i, [_i, _vp, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _i, _f, _vp, _i, _vp]),
    "eesen_op_gemm_async": (_i, [_i, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, C.c_long, _i]),
}


class EesenError(RuntimeError):
    """Mirrors the std::runtime_error thrown by KALDI_ERR (/root/reference/src/base/kaldi-error.cc:168-182)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"[{code}] {msg}")
        self.code = code


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EesenError(-2, f"{LIB_PATH} is missing: build it with `python -m eesen_amd.build` "
                                 "(there is no CPU fallback for the HIP path)")
        # multi-process GPU work (RCCL peer access) needs dmabuf IPC on this driver stack; the ROCr runtime reads the variable when
        # it initialises, i.e. at the first HIP call made through this library
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError = a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc: int):
    if rc != OK:
        raise EesenError(rc, load().eesen_last_error().decode(errors="replace"))


def device_count() -> int:
    n = C.c_int(0)
    check(load().eesen_device_count(C.byref(n)))
    return n.value
