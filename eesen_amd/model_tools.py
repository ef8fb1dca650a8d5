"""The reference's model-file tools around the trainer, host side only (no device): the recipes call them between training
iterations and before decoding.

    python -m eesen_amd.model_tools net-change-model [--binary=B] [--forwarddrop=F --forwardstep=B ...] <model-in> <model-out>
    python -m eesen_amd.model_tools net-copy [--binary=B] [--remove-first-layers=N] [--remove-last-layers=N] <model-in> <model-out>
    python -m eesen_amd.model_tools format-to-nonparallel [--binary=B] <model-in> <model-out>

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


TOOLS = {"net-change-model": net_change_model, "net-copy": net_copy, "format-to-nonparallel": format_to_nonparallel}


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
