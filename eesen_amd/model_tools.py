"""The reference's model-file tools around the trainer, host side only (no device): the recipes call them between training
iterations and before decoding.

    python -m eesen_amd.model_tools net-change-model [--binary=B] [--forwarddrop=F --forwardstep=B ...] <model-in> <model-out>
    python -m eesen_amd.model_tools net-copy [--binary=B] [--remove-first-layers=N] [--remove-last-layers=N] <model-in> <model-out>
    python -m eesen_amd.model_tools format-to-nonparallel [--binary=B] <model-in> <model-out>

and one tool the reference does not need (libeesen_hip.so pads LSTM cell counts to multiples of 4 internally; this writes the padded model
OUT, for the few places where the padding would show -- INTEGRATION.md "Restrictions"):

    python -m eesen_amd.model_tools pad-cells [--binary=B] [--multiple=4] <model-in> <model-out>
    python -m eesen_amd.model_tools unpad-cells [--binary=B] --cells=H1,H2,... <model-in> <model-out>

Same options, argument order, checks and exit codes as /root/reference/src/netbin/{net-change-model,net-copy,format-to-nonparallel}.cc;
binary output is byte-identical to the reference tools' (tests/test_model_tools.py pins it where oracle/_ref/netbin exists).
"""
from __future__ import annotations

import sys
from typing import List

from . import nnet_io


def _bool(v: str) -> bool:
    v = v.strip().lower()
    if v in ("", "true", "t", "1"):
        return True
    if v in ("false", "f", "0"):
        return False
    raise ValueError(f"Invalid format for boolean argument [expected true or false]: {v}")


def _parse(argv: List[str], spec: dict):
    """spec: option name -> (type, default).  Returns (options, positional)."""
    o = {k: d for k, (_, d) in spec.items()}
    pos = []
    for a in argv:
        if a.startswith("--"):
            name, _, val = a[2:].partition("=")
            name = name.replace("_", "-")
            if name not in spec:
                raise ValueError(f"Invalid option {a}")
            t = spec[name][0]
            o[name] = _bool(val) if t is bool else t(val)
        else:
            pos.append(a)
    return o, pos


def check_dropout(d: dict):
    """BiLstm::ChangeDropoutParameters, /root/reference/src/net/bilstm-layer.h:74-82, 101-112."""
    fwd, st, sq = d["forward"], d["fw_step"], d["fw_seq"]
    if fwd > 0.0 and not (sq or st):
        raise ValueError("ForwardDropoutFactor > 0 but ForwardTimeStepDropout and ForwardSequenceDropout are both false, One must be true.")
    if sq and st:
        raise ValueError("Both ForwardTimeStepDropout and ForwardSequenceDropout are true, Only one can be true.")
    if fwd == 0.0 and (sq or st):
        raise ValueError("ForwardDropoutFactor = 0 but ForwardTimeStepDropout and/or ForwardSequenceDropout is true, both must be false.")
    if d["rec_seq"] and d["rec_step"]:
        raise ValueError("RecurrentSequenceDropout and RecurrentTimeStepDropout cannot be true at the same time. Pick one.")
    if d["rnndrop"] and d["nml"]:
        raise ValueError("Only one of RNNDrop, NoMemLossDropout can be true. Pick one.")
    if d["recurrent"] == 0.0 and (d["nml"] or d["rnndrop"]):
        raise ValueError("RecurrentDropoutFactor must be nonzero if RNNDrop or NoMemLossDropout is true")
    if not (d["rec_step"] or d["rec_seq"]) and (d["rnndrop"] or d["nml"]):
        raise ValueError(" Either RecurrentSequenceDropout or RecurrentTimeStepDropout must be true if RNNDrop or NoMemLossDropout is true")


def net_change_model(argv: List[str]) -> int:
    """net-change-model.cc:38-92; Net::ChangeDropoutParameters net.cc:414-434: every BiLstm(Parallel) layer gets the SAME,
    fully specified option set (options not given fall back to off)."""
    spec = {"binary": (bool, True), "forwarddrop": (float, 0.0), "forwardstep": (bool, False), "forwardseq": (bool, False),
            "rnndrop": (bool, False), "nmldrop": (bool, False), "recurrentdrop": (float, 0.0), "recurrentstep": (bool, False),
            "recurrentseq": (bool, False), "twiddleforward": (bool, False)}
    o, pos = _parse(argv, spec)
    if len(pos) != 2:
        print("Usage:  net-change-model [options] <model-in> <model-out>", file=sys.stderr)
        return 1
    layers = nnet_io.read_nnet(pos[0])
    d = dict(forward=o["forwarddrop"], fw_step=o["forwardstep"], fw_seq=o["forwardseq"], rec_step=o["recurrentstep"],
             rec_seq=o["recurrentseq"], rnndrop=o["rnndrop"], nml=o["nmldrop"], recurrent=o["recurrentdrop"], twiddle=o["twiddleforward"])
    for i, L in enumerate(layers):
        if L["type"] in ("BiLstm", "BiLstmParallel"):
            print(f"LOG (net-change-model) Changing dropout params for layer {i}", file=sys.stderr)
            check_dropout(d)
            L["dropout"] = dict(d)
    nnet_io.write_nnet(pos[1], layers, binary=o["binary"])
    print(f"LOG (net-change-model) Written model to {pos[1]}", file=sys.stderr)
    return 0


def net_copy(argv: List[str]) -> int:
    """net-copy.cc:38-85."""
    o, pos = _parse(argv, {"binary": (bool, True), "remove-first-layers": (int, 0), "remove-last-layers": (int, 0)})
    if len(pos) != 2:
        print("Usage:  net-copy [options] <model-in> <model-out>", file=sys.stderr)
        return 1
    layers = nnet_io.read_nnet(pos[0])
    if o["remove-first-layers"] > 0:
        layers = layers[o["remove-first-layers"]:]
    if o["remove-last-layers"] > 0:
        layers = layers[: len(layers) - o["remove-last-layers"]]
    nnet_io.write_nnet(pos[1], layers, binary=o["binary"])
    print(f"LOG (net-copy) Written model to {pos[1]}", file=sys.stderr)
    return 0


NONPARALLEL = {"BiLstmParallel": "BiLstm", "LstmParallel": "Lstm"}     # Layer::GetTypeNonParal


def format_to_nonparallel(argv: List[str]) -> int:
    """format-to-nonparallel.cc:36-65 (Net::WriteNonParal: the layer markers change, nothing else)."""
    o, pos = _parse(argv, {"binary": (bool, True)})
    if len(pos) != 2:
        print("Usage:  format-to-nonparallel [options] <model-in> <model-out>", file=sys.stderr)
        return 1
    layers = nnet_io.read_nnet(pos[0])
    for L in layers:
        L["type"] = NONPARALLEL.get(L["type"], L["type"])
    nnet_io.write_nnet(pos[1], layers, binary=o["binary"])
    print(f"LOG (format-to-nonparallel) Written model to {pos[1]}", file=sys.stderr)
    return 0


# ---- cells per direction padded to a multiple of 4 ---------------------------------------------------------------------------------
# libeesen_hip.so fetches the recurrent state four cells at a time; the reference takes any cell count.  The library pads a layer whose
# count per direction is not a multiple of 4 internally (net.cpp: add_layer, for_each_param) by exactly the construction below, which
# this tool applies to the model FILE: <CellDim> 300 for a BiLstm (150 per direction) becomes 152 per direction, with cells that are
# identically zero and stay so:
#   a padded cell has zero W_x / W_m rows, zero bias, zero peepholes: g = tanh(0) = 0, i = f = o = 1/2, c_t = f c_{t-1} + i g = 0,
#   m_t = o tanh(c_t) = 0 for every t;  the columns that read its output -- W_m's own columns and the next layer's input columns --
#   are zero, so nothing downstream sees it: the padded net computes the original function.
#   Training keeps it so: the cell receives d_m = 0 (zero columns above it), hence zero gate gradients, hence zero gradients of its own
#   rows; and the gradient of a zero COLUMN is sum_t dG_t^T x_t over an input x that is identically 0.  SGD with momentum, the
#   clipping and Adagrad / RMSProp all map (parameter 0, gradient 0, accumulator 0) to 0.  (Not so behind a <Sigmoid> layer, whose
#   output for the padded cell is 1/2: refused, like a <Softmax> directly on a padded layer.)
# Every tool of either code base reads the padded file; unpad-cells cuts it back (and checks that what it cuts is zero).
def _lstm_dirs(t: str) -> int:
    return 2 if t.startswith("BiLstm") else 1


def _expand(a, axis: int, H: int, Hp: int, blocks: int):
    """Splits `axis` into `blocks` runs of H entries and pads each run with Hp - H zeros."""
    import numpy as np
    a = np.asarray(a, np.float32)
    sh = list(a.shape)
    assert sh[axis] == blocks * H, (sh, axis, blocks, H)
    sh[axis:axis + 1] = [blocks, H]
    a = a.reshape(sh)
    pad = [(0, 0)] * a.ndim
    pad[axis + 1] = (0, Hp - H)
    a = np.pad(a, pad)
    sh[axis:axis + 2] = [blocks * Hp]
    return np.ascontiguousarray(a.reshape(sh))


def _shrink(a, axis: int, H: int, Hp: int, blocks: int, what: str):
    import numpy as np
    a = np.asarray(a, np.float32)
    sh = list(a.shape)
    assert sh[axis] == blocks * Hp, (sh, axis, blocks, Hp)
    sh[axis:axis + 1] = [blocks, Hp]
    a = a.reshape(sh)
    keep = [slice(None)] * a.ndim; cut = list(keep)
    keep[axis + 1] = slice(0, H); cut[axis + 1] = slice(H, Hp)
    if np.any(a[tuple(cut)] != 0):
        raise ValueError(f"unpad-cells: {what} is not zero where it would be cut -- not a model padded by pad-cells (or --cells is wrong)")
    a = a[tuple(keep)]
    sh[axis:axis + 2] = [blocks * H]
    return np.ascontiguousarray(a.reshape(sh))


def _repad(layers, new_H, fn):
    """Shared walk of pad-cells / unpad-cells.  new_H(layer index, H) -> the layer's new cell count per direction;
    fn(array, axis, H_old, H_new, blocks, what) resizes one axis made of `blocks` runs of H_old entries."""
    out, col = [], None    # col = (H_old, H_new, blocks) of the columns the next layer reads, when they changed
    for i, L in enumerate(layers):
        L = dict(L); t = L["type"]
        tens = {k: [p for p in L[k]] for k in ("params", "accu") if L.get(k)}
        if L["type"] in ("Softmax", "Sigmoid") and col:
            raise ValueError(f"layer {i}: a <{t}> directly on a padded LSTM layer would see the padded cells (sigmoid(0) = 1/2); not supported")
        if col and t == "Tanh":                       # tanh(0) = 0: the padded columns pass through
            L["input_dim"] = L["output_dim"] = col[1] * col[2]
        elif nnet_io.is_lstm(t):
            nd = _lstm_dirs(t); H = L["output_dim"] // nd; Hn = new_H(i, H)
            for k, ts in tens.items():
                for d in range(nd):
                    wx, wm, b, pi, pf, po = ts[6 * d: 6 * d + 6]
                    if col: wx = fn(wx, 1, col[0], col[1], col[2], f"layer {i} W_x columns")
                    if Hn != H:
                        wx = fn(wx, 0, H, Hn, 4, f"layer {i} W_x rows")
                        wm = fn(fn(wm, 0, H, Hn, 4, f"layer {i} W_m rows"), 1, H, Hn, 1, f"layer {i} W_m columns")
                        b = fn(b, 0, H, Hn, 4, f"layer {i} bias")
                        pi, pf, po = (fn(v, 0, H, Hn, 1, f"layer {i} peephole") for v in (pi, pf, po))
                    ts[6 * d: 6 * d + 6] = [wx, wm, b, pi, pf, po]
            if col: L["input_dim"] = col[1] * col[2]
            L["output_dim"] = nd * Hn
            col = (H, Hn, nd) if Hn != H else None
        elif t == "AffineTransform":
            if col:
                for k, ts in tens.items():
                    ts[0] = fn(ts[0], 1, col[0], col[1], col[2], f"layer {i} weight columns")
                L["input_dim"] = col[1] * col[2]
            col = None
        for k, ts in tens.items():
            L[k] = ts
        out.append(L)
    if col:
        raise ValueError("the last layer is a padded LSTM layer: its output dimension would change; not supported")
    return out


def pad_cells_layers(layers, multiple: int = 4):
    return _repad(layers, lambda i, H: -(-H // multiple) * multiple, lambda a, ax, H, Hn, nb, what: _expand(a, ax, H, Hn, nb))


def unpad_cells_layers(layers, cells):
    idx = [i for i, L in enumerate(layers) if nnet_io.is_lstm(L["type"])]
    if len(cells) != len(idx):
        raise ValueError(f"unpad-cells: the model has {len(idx)} LSTM layers, --cells names {len(cells)}")
    want = dict(zip(idx, cells))

    def new_H(i, H):
        if want[i] > H or want[i] <= 0:
            raise ValueError(f"unpad-cells: layer {i} has {H} cells per direction, cannot cut to {want[i]}")
        return want[i]
    return _repad(layers, new_H, lambda a, ax, Hp, H, nb, what: _shrink(a, ax, H, Hp, nb, what))


def pad_cells(argv: List[str]) -> int:
    o, pos = _parse(argv, {"binary": (bool, True), "multiple": (int, 4)})
    if len(pos) != 2 or o["multiple"] <= 0:
        print("Usage:  pad-cells [--binary=true] [--multiple=4] <model-in> <model-out>", file=sys.stderr)
        return 1
    layers = nnet_io.read_nnet(pos[0])
    padded = pad_cells_layers(layers, o["multiple"])
    for i, (a, b) in enumerate(zip(layers, padded)):
        if a["output_dim"] != b["output_dim"]:
            print(f"LOG (pad-cells) layer {i} <{a['type']}>: <CellDim> {a['output_dim']} -> {b['output_dim']}", file=sys.stderr)
    nnet_io.write_nnet(pos[1], padded, binary=o["binary"])
    print(f"LOG (pad-cells) Written model to {pos[1]}", file=sys.stderr)
    return 0


def unpad_cells(argv: List[str]) -> int:
    o, pos = _parse(argv, {"binary": (bool, True), "cells": (str, "")})
    if len(pos) != 2 or not o["cells"]:
        print("Usage:  unpad-cells [--binary=true] --cells=H1,H2,... (cells per direction of every LSTM layer) <model-in> <model-out>", file=sys.stderr)
        return 1
    layers = unpad_cells_layers(nnet_io.read_nnet(pos[0]), [int(x) for x in o["cells"].split(",")])
    nnet_io.write_nnet(pos[1], layers, binary=o["binary"])
    print(f"LOG (unpad-cells) Written model to {pos[1]}", file=sys.stderr)
    return 0


TOOLS = {"net-change-model": net_change_model, "net-copy": net_copy, "format-to-nonparallel": format_to_nonparallel,
         "pad-cells": pad_cells, "unpad-cells": unpad_cells}


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] not in TOOLS:
        print("usage: python -m eesen_amd.model_tools {" + ",".join(TOOLS) + "} [options] <model-in> <model-out>", file=sys.stderr)
        return 1
    try:
        return TOOLS[argv[0]](argv[1:])
    except Exception as e:          # the reference tools print e.what() and return -1
        print(str(e), file=sys.stderr)
        return 255


if __name__ == "__main__":
    sys.exit(main())
