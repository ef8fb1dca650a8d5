"""OracleNet / oracle_ctc: CPU restatement of the reference's Net + Ctc orchestration over oracle/eesen_oracle.c.

TEST INFRASTRUCTURE — the checker for tests/, smoke() and bench.py's cpu_baseline; never the product path.

Orchestration follows /root/reference/src/net/net.cc:
  Propagate     :67-86    layer chain, every layer input kept (propagate_buf_)
  Backpropagate :88-108   top-down; per trainable layer BackpropagateFnc then Update (:101-104)
with the layer internals of bilstm-parallel-layer.h:379-420,881-913, lstm-parallel-layer.h:47-213,
affine-trans-layer.h:161-219, softmax-layer.h:44-57 (backward = copy: CTC already yields d/d(logits)).

Gradient bookkeeping mirrors the reference: every parameter has a `corr` buffer holding
`momentum * corr + fresh_gradient` (sum over frames, bilstm-parallel-layer.h:504-510); Update clips corr
(if max_grad > 0) and applies `param -= lr * learn_rate_coef * corr` (bilstm-layer.h:846-883).
For data-parallel parity `fresh` gradients (momentum 0 contribution) are also kept per step.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np

from . import cbind


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleNet:
    def __init__(self, layers: List[dict], prec: str = "f32"):
        """layers: as produced by eesen_amd.nnet_io.read_nnet / eesen_amd.synth.make_model."""
        self.prec = prec
        self.lib = cbind.load(prec)
        self.dt = cbind.dtype(prec)
        self.real = C.c_float if prec == "f32" else C.c_double
        self.layers = []
        for L in layers:
            self.layers.append(dict(type=L["type"], din=L["input_dim"], dout=L["output_dim"],
                                    coef=float(L.get("learn_rate_coef", 1.0)), max_grad=float(L.get("max_grad", 0.0)),
                                    dropout=dict(L.get("dropout") or {}),
                                    params=[np.array(p, dtype=self.dt, order="C") for p in L["params"]]))
        for L in self.layers:
            L["corr"] = [np.zeros_like(p) for p in L["params"]]     # zeroed at Read (bilstm-layer.h:405-410)
        self.rule = "SGD"          # Net::SetUpdateAlgorithm (net.cc:481-497)
        self.ada_eps, self.rms_rho = 1e-6, 0.9
        for L in self.layers:
            L["accu"] = [np.zeros_like(p) for p in L["params"]]
        self.learn_rate = 0.0      # Net::Read resets it (net.cc:294)
        self.momentum = 0.0
        self.lens = None
        self.bufs = None
        self.in_train = True       # BiLstm::in_train defaults to true (bilstm-layer.h:38)
        self.masks = {}            # layer index -> dict(fwd=[T*S x 2H], rec_fw/rec_bw=[(T+2)*S or S x H], twiddle_apply_forward)

    # -- dropout ---------------------------------------------------------------
    def set_mode(self, train: bool):
        """Net::SetTrainMode / SetTestMode (net.cc:396-412)."""
        self.in_train = bool(train)

    def set_dropout_masks(self, layer: int, fwd=None, rec_fw=None, rec_bw=None, twiddle_apply_forward: bool = False):
        """The masks the reference would draw from its host RNG (bilstm-parallel-layer.h:46-94) are INPUTS here."""
        cv = lambda a: None if a is None or not np.size(a) else np.ascontiguousarray(a, self.dt)
        self.masks[layer] = dict(fwd=cv(fwd), rec_fw=cv(rec_fw), rec_bw=cv(rec_bw), twiddle_apply_forward=bool(twiddle_apply_forward))

    def _dropout_plan(self, li: int):
        """(drop_mode, forward_dropout?) for layer li in the current mode: bilstm-parallel-layer.h:385-390."""
        o = self.layers[li]["dropout"]
        if not o or not self.in_train:
            return 0, False
        tw = bool(o.get("twiddle", False))
        coin = self.masks.get(li, {}).get("twiddle_apply_forward", False)
        rec = (o.get("rnndrop", False) or o.get("nml", False)) and (not tw or not coin)
        fwd = o.get("forward", 0.0) > 0.0 and (not tw or coin)
        mode = 0 if not rec else (2 if o.get("rnndrop", False) else 1)
        return mode, bool(fwd)

    # -- reference API mirror -------------------------------------------------------------------
    def set_train_options(self, learn_rate: float, momentum: float):
        self.learn_rate, self.momentum = float(learn_rate), float(momentum)

    def set_seq_lengths(self, lens):
        self.lens = np.ascontiguousarray(lens, np.int32)

    def get_params(self) -> np.ndarray:
        return np.concatenate([p.ravel() for L in self.layers for p in L["params"]])

    def get_corr(self) -> np.ndarray:
        return np.concatenate([c.ravel() for L in self.layers for c in L["corr"]])

    def propagate(self, feats: np.ndarray) -> np.ndarray:
        S = len(self.lens); rows = feats.shape[0]; T = rows // S
        x = np.ascontiguousarray(feats, self.dt)
        self.acts = [x]; self.state = []
        for li, L in enumerate(self.layers):
            t = L["type"]
            if t in ("BiLstmParallel", "LstmParallel"):
                ndir = 2 if t == "BiLstmParallel" else 1
                H = L["dout"] // ndir
                out = np.zeros((rows, L["dout"]), self.dt)
                bufs = []
                mode, fwd_drop = self._dropout_plan(li)
                mk = self.masks.get(li, {})
                step = bool(L["dropout"].get("rec_step", False))
                for d in range(ndir):
                    Wx, Wm, b, pi, pf, po = L["params"][6 * d: 6 * d + 6]
                    buf = np.empty(((T + 2) * S, 7 * H), self.dt)
                    rm = (mk.get("rec_bw") if d else mk.get("rec_fw")) if mode else None
                    assert not mode or (rm is not None and rm.shape == (((T + 2) * S if step else S), H)), "recurrent dropout needs its masks"
                    self.lib.orc_lstm_dir_forward(T, S, L["din"], H, d, _p(self.lens), _p(x), _p(Wx), _p(Wm), _p(b),
                                                  _p(pi), _p(pf), _p(po), _p(buf), _p(rm), int(step), mode)
                    out[:, d * H:(d + 1) * H] = buf[S:(T + 1) * S, 6 * H:7 * H]   # bilstm-parallel-layer.h:409-419
                    bufs.append(buf)
                if fwd_drop:                                                        # :414-417
                    assert mk.get("fwd") is not None and mk["fwd"].shape == out.shape, "forward dropout needs its mask"
                    out = out * mk["fwd"]
                self.state.append(bufs)
            elif t == "AffineTransform":
                out = np.empty((rows, L["dout"]), self.dt)
                self.lib.orc_affine_forward(rows, L["din"], L["dout"], _p(x), _p(L["params"][0]), _p(L["params"][1]), _p(out))
                self.state.append(None)
            elif t == "Softmax":
                out = np.empty((rows, L["dout"]), self.dt)
                self.lib.orc_softmax_rows(rows, L["dout"], _p(x), _p(out))
                self.state.append(None)
            elif t == "Sigmoid":       # sigmoid-layer.h:44-46 -> cuda-kernels.cu:687-697: 1 / (1 + exp(-x))
                out = (1.0 / (1.0 + np.exp(-x.astype(self.dt)))).astype(self.dt)
                self.state.append(None)
            elif t == "Tanh":          # tanh-layer.h:44-46 -> cuda-kernels.cu:713-727: (e^2x - 1) / (e^2x + 1), 1 where e^2x overflows
                with np.errstate(over="ignore", invalid="ignore"):
                    e = np.exp(self.dt(2.0) * x.astype(self.dt))
                    out = np.where(np.isinf(e), self.dt(1.0), (e - 1.0) / (e + 1.0)).astype(self.dt)
                self.state.append(None)
            else:
                raise NotImplementedError(t)
            x = out
            self.acts.append(x)
        return x

    def backpropagate(self, out_diff: np.ndarray, update: bool = True) -> np.ndarray:
        S = len(self.lens); rows = out_diff.shape[0]; T = rows // S
        d = np.ascontiguousarray(out_diff, self.dt)
        mmt = self.real(self.momentum)
        self.fresh = [None] * len(self.layers)
        for li in range(len(self.layers) - 1, -1, -1):
            L = self.layers[li]; t = L["type"]; x = self.acts[li]
            in_diff = np.zeros((rows, L["din"]), self.dt)
            before = [c.copy() for c in L["corr"]]
            if t in ("BiLstmParallel", "LstmParallel"):
                ndir = 2 if t == "BiLstmParallel" else 1
                H = L["dout"] // ndir
                mode, fwd_drop = self._dropout_plan(li)
                mk = self.masks.get(li, {})
                step = bool(L["dropout"].get("rec_step", False))
                if fwd_drop:
                    d = np.ascontiguousarray(d * mk["fwd"])                          # out_diff_drop, :892-896
                for dd in range(ndir):
                    Wx, Wm, b, pi, pf, po = L["params"][6 * dd: 6 * dd + 6]
                    cWx, cWm, cb, cpi, cpf, cpo = L["corr"][6 * dd: 6 * dd + 6]
                    dbuf = np.empty(((T + 2) * S, 7 * H), self.dt)
                    rm = (mk.get("rec_bw") if dd else mk.get("rec_fw")) if mode else None
                    self.lib.orc_lstm_dir_backward(T, S, L["din"], H, dd, _p(x), _p(self.state[li][dd]), _p(d), L["dout"], dd * H,
                                                   _p(Wx), _p(Wm), _p(pi), _p(pf), _p(po), _p(dbuf), _p(in_diff), dd, mmt,
                                                   _p(cWx), _p(cWm), _p(cb), _p(cpi), _p(cpf), _p(cpo), _p(rm), int(step), mode)
            elif t == "AffineTransform":
                self.lib.orc_affine_backward(rows, L["din"], L["dout"], _p(d), _p(L["params"][0]), _p(in_diff))
                # gradients are computed inside Update in the reference (affine-trans-layer.h:182-183)
                self.lib.orc_affine_grads(rows, L["din"], L["dout"], _p(x), _p(d), mmt, _p(L["corr"][0]), _p(L["corr"][1]))
            elif t == "Softmax":
                in_diff = d.copy()                                                   # softmax-layer.h:49-57
            elif t == "Sigmoid":
                y = self.acts[li + 1]
                in_diff = (y * (1.0 - y) * d).astype(self.dt)                        # sigmoid-layer.h:48-51 -> cuda-kernels.cu:699-709
            elif t == "Tanh":
                y = self.acts[li + 1]
                in_diff = ((1.0 - y * y) * d).astype(self.dt)                        # tanh-layer.h:48-51 -> cuda-kernels.cu:730-740
            # fresh gradient of this step = corr_after - momentum * corr_before (pre-clipping)
            self.fresh[li] = [c - self.momentum * b0 for c, b0 in zip(L["corr"], before)]
            if update and L["params"]:
                self.update_layer(li)
            d = in_diff
        return d

    def set_update_algorithm(self, rule: str, adagrad_epsilon: float = 1e-6, rmsprop_rho: float = 0.9):
        assert rule in ("SGD", "Adagrad", "RMSProp")
        self.rule, self.ada_eps, self.rms_rho = rule, adagrad_epsilon, rmsprop_rho

    def get_accu(self) -> np.ndarray:
        return np.concatenate([a.ravel() for L in self.layers for a in L["accu"]])

    def update_layer(self, li: int):
        L = self.layers[li]
        for p, c, a in zip(L["params"], L["corr"], L["accu"]):
            if self.rule == "SGD":
                self.lib.orc_sgd_update(C.c_long(p.size), _p(p), _p(c), self.real(self.learn_rate * L["coef"]), self.real(L["max_grad"]))
            else:
                rho = np.float32(self.rms_rho)
                self.lib.orc_adaptive_update(C.c_long(p.size), _p(p), _p(c), _p(a), self.real(self.learn_rate), self.real(L["max_grad"]),
                                             self.real(self.ada_eps), self.real(rho), self.real(np.float32(1.0) - rho),
                                             int(self.rule == "RMSProp"))

    def fresh_grads_flat(self) -> np.ndarray:
        return np.concatenate([g.ravel() for f in self.fresh if f for g in f])

    def to_layers(self) -> List[dict]:
        return [dict(type=L["type"], input_dim=L["din"], output_dim=L["dout"], learn_rate_coef=L["coef"], max_grad=L["max_grad"],
                     params=[p.astype(np.float32) for p in L["params"]]) for L in self.layers]


def ctc_eval_parallel(probs: np.ndarray, T: int, S: int, lens, label_ids, label_off, prec: str = "f32"):
    """Ctc::EvalParallel restated (oracle/eesen_oracle.c: orc_ctc_eval_parallel). Returns dict(alpha, beta, pzx, diff, L)."""
    lib = cbind.load(prec); dt = cbind.dtype(prec)
    probs = np.ascontiguousarray(probs, dt)
    K = probs.shape[1]
    lens = np.ascontiguousarray(lens, np.int32)
    ids = np.ascontiguousarray(label_ids, np.int32); off = np.ascontiguousarray(label_off, np.int32)
    L = 2 * int(np.max(np.diff(off))) + 1
    alpha = np.empty((T * S, L), dt); beta = np.empty((T * S, L), dt)
    pzx = np.empty(S, dt); diff = np.empty((T * S, K), dt)
    rc = lib.orc_ctc_eval_parallel(T, S, K, _p(probs), _p(lens), _p(ids), _p(off), _p(alpha), _p(beta), _p(pzx), _p(diff), L)
    assert rc == L, rc
    return dict(alpha=alpha, beta=beta, pzx=pzx, diff=diff, L=L)


def ctc_error_rate_mseq(net_out: np.ndarray, T: int, S: int, lens, label_ids, label_off, prec: str = "f32"):
    lib = cbind.load(prec); dt = cbind.dtype(prec)
    net_out = np.ascontiguousarray(net_out, dt)
    lens = np.ascontiguousarray(lens, np.int32)
    ids = np.ascontiguousarray(label_ids, np.int32); off = np.ascontiguousarray(label_off, np.int32)
    ne, nr = C.c_int(0), C.c_int(0)
    lib.orc_ctc_error_rate_mseq(T, S, net_out.shape[1], _p(net_out), _p(lens), _p(ids), _p(off), C.byref(ne), C.byref(nr))
    return ne.value, nr.value


def train_step(net: OracleNet, batch, prec: str = "f32"):
    """One pass of the reference trainer's inner loop (netbin/train-ctc-parallel.cc:195-207):
    SetSeqLengths → Propagate → EvalParallel → Backpropagate(+Update). Returns dict of everything checkable."""
    net.set_seq_lengths(batch.lens)
    net_out = net.propagate(batch.feats)
    ctc = ctc_eval_parallel(net_out, batch.T, batch.S, batch.lens, batch.label_ids, batch.label_off, prec)
    in_diff = net.backpropagate(ctc["diff"])
    return dict(net_out=net_out, pzx=ctc["pzx"], diff=ctc["diff"], in_diff=in_diff, alpha=ctc["alpha"], beta=ctc["beta"])
