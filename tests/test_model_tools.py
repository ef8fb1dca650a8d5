"""CPU: eesen_amd/model_tools.py (net-change-model, net-copy, format-to-nonparallel) -- the model-file tools the recipes run
around the trainer.  Where the reference's own binaries exist (oracle/_ref/netbin, compiled unmodified by
oracle/ref_build/Makefile target `netbin`) the binary outputs must be byte-identical; everywhere, the effect on the file is
checked through eesen_amd/nnet_io.py."""
import os
import subprocess

import numpy as np
import pytest

from eesen_amd import model_tools, nnet_io, synth

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "netbin")
have_ref = os.path.isfile(os.path.join(BIN, "net-change-model"))


def _model(tmp_path, **over):
    cfg = synth.config("small_bi"); cfg.update(layers=2, proj=12, proj_act="Tanh", H=8, D=5, K=6); cfg.update(over)
    layers = synth.make_model(max_grad=50.0, **cfg)
    p = str(tmp_path / "in.nnet")
    nnet_io.write_nnet(p, layers, binary=True)
    return layers, p


CASES = [
    ("net-change-model", ["--forwarddrop=0.2", "--forwardstep=true", "--recurrentdrop=0.3", "--recurrentseq=true", "--nmldrop=true"]),
    ("net-change-model", ["--forwarddrop=0.1", "--forwardseq=true", "--twiddleforward=true", "--rnndrop=true", "--recurrentdrop=0.5", "--recurrentstep=true"]),
    ("net-change-model", []),                                   # options not given fall back to off
    ("net-copy", []),
    ("net-copy", ["--remove-last-layers=2"]),
    ("net-copy", ["--remove-first-layers=1", "--remove-last-layers=1"]),
    ("format-to-nonparallel", []),
]


@pytest.mark.parametrize("tool,opts", CASES)
def test_tools_against_the_reference_binaries(tmp_path, tool, opts):
    layers, p = _model(tmp_path)
    mine = str(tmp_path / "mine.nnet")
    assert model_tools.main([tool] + opts + [p, mine]) == 0
    back = nnet_io.read_nnet(mine)
    if tool == "net-change-model":
        want = dict(zip(nnet_io.DROPOUT_KEYS, [0.0, False, False, False, False, False, False, 0.0, False]))
        names = {"forwarddrop": "forward", "forwardstep": "fw_step", "forwardseq": "fw_seq", "recurrentstep": "rec_step", "recurrentseq": "rec_seq",
                 "rnndrop": "rnndrop", "nmldrop": "nml", "recurrentdrop": "recurrent", "twiddleforward": "twiddle"}
        for o in opts:
            k, v = o[2:].split("=")
            want[names[k]] = float(v) if "drop=" in o and k in ("forwarddrop", "recurrentdrop") else v == "true"
        for L in back:
            if L["type"].startswith("BiLstm"):
                got = dict(zip(nnet_io.DROPOUT_KEYS, nnet_io.dropout_values(L)))
                assert all(abs(float(got[k]) - float(want[k])) < 1e-7 for k in want), (got, want)
        assert np.array_equal(nnet_io.flatten_params(back), nnet_io.flatten_params(layers))
    elif tool == "format-to-nonparallel":
        assert [L["type"] for L in back] == [{"BiLstmParallel": "BiLstm"}.get(L["type"], L["type"]) for L in layers]
        assert np.array_equal(nnet_io.flatten_params(back), nnet_io.flatten_params(layers))
    else:
        first = int(dict(o[2:].split("=") for o in opts).get("remove-first-layers", 0))
        last = int(dict(o[2:].split("=") for o in opts).get("remove-last-layers", 0))
        assert [L["type"] for L in back] == [L["type"] for L in layers][first: len(layers) - last]
    if have_ref:
        ref = str(tmp_path / "ref.nnet")
        r = subprocess.run([os.path.join(BIN, tool)] + opts + [p, ref], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1500:]
        assert open(mine, "rb").read() == open(ref, "rb").read(), "binary output must equal the reference tool's byte for byte"


@pytest.mark.parametrize("opts,msg", [
    (["--forwarddrop=0.2"], "One must be true"),
    (["--forwarddrop=0.2", "--forwardstep=true", "--forwardseq=true"], "Only one can be true"),
    (["--forwardstep=true"], "both must be false"),
    (["--recurrentdrop=0.2", "--recurrentstep=true", "--recurrentseq=true"], "Pick one"),
    (["--rnndrop=true", "--nmldrop=true", "--recurrentdrop=0.2", "--recurrentstep=true"], "Only one of RNNDrop"),
    (["--rnndrop=true", "--recurrentstep=true"], "must be nonzero"),
    (["--rnndrop=true", "--recurrentdrop=0.2"], "must be true if RNNDrop"),
])
def test_net_change_model_refuses_what_the_reference_refuses(tmp_path, capsys, opts, msg):
    """bilstm-layer.h:74-82, 101-112."""
    _, p = _model(tmp_path)
    out = str(tmp_path / "o.nnet")
    assert model_tools.main(["net-change-model"] + opts + [p, out]) == 255
    assert msg in capsys.readouterr().err and not os.path.exists(out)
    if have_ref:
        r = subprocess.run([os.path.join(BIN, "net-change-model")] + opts + [p, str(tmp_path / "r.nnet")], capture_output=True, text=True)
        assert r.returncode != 0 and msg in r.stderr


def test_usage_and_text_output(tmp_path, capsys):
    _, p = _model(tmp_path)
    assert model_tools.main(["net-copy", p]) == 1 and "Usage" in capsys.readouterr().err
    assert model_tools.main(["no-such-tool"]) == 1
    t = str(tmp_path / "t.nnet")
    assert model_tools.main(["net-copy", "--binary=false", p, t]) == 0
    assert open(t).read().startswith("<Nnet>")
    assert np.array_equal(nnet_io.flatten_params(nnet_io.read_nnet(t)), nnet_io.flatten_params(nnet_io.read_nnet(p)))


@pytest.mark.parametrize("kind,H,extra", [("BiLstmParallel", 10, {}), ("LstmParallel", 7, {}), ("BiLstmParallel", 6, dict(proj=9)),
                                          ("BiLstmParallel", 5, dict(proj=6, proj_act="Tanh"))])
def test_pad_cells_is_function_and_training_preserving(tmp_path, kind, H, extra):
    """pad-cells (the construction libeesen_hip.so applies INSIDE to LSTM layers whose cells per direction are not a multiple of 4, here
    applied to the model file): the padded model -- a file every tool of either code base reads -- computes the ORIGINAL function bit
    for bit and stays padded with exact zeros through training (momentum, clipping, Adagrad), on the CPU restatement of the reference
    (oracle/); unpad-cells cuts it back and refuses a model whose padding is not zero.  (The device side of the same statement:
    tests/test_gpu_parity.py::test_cell_counts_that_are_not_multiples_of_4.)"""
    from oracle.net import OracleNet, train_step
    cfg = dict(kind=kind, layers=2, H=H, D=7, K=9, S=3, T=14); cfg.update(extra)
    layers = synth.make_model(max_grad=5.0, **cfg); batch = synth.make_batch(**cfg)
    src, dst, back = (str(tmp_path / n) for n in ("in.nnet", "pad.nnet", "back.nnet"))
    nnet_io.write_nnet(src, layers, binary=True)
    assert model_tools.main(["pad-cells", src, dst]) == 0
    padded = nnet_io.read_nnet(dst)
    nd = 2 if kind.startswith("Bi") else 1
    assert all(L["output_dim"] % (4 * nd) == 0 for L in padded if nnet_io.is_lstm(L["type"]))
    assert padded[-1]["output_dim"] == layers[-1]["output_dim"] and padded[0]["input_dim"] == layers[0]["input_dim"]
    assert model_tools.main(["unpad-cells", f"--cells={H},{H}", dst, back]) == 0
    assert open(back, "rb").read() == open(src, "rb").read()
    for rule in ("SGD", "Adagrad"):
        res = {}
        for name, ls in (("orig", layers), ("pad", padded)):
            net = OracleNet(ls); net.set_train_options(0.5, 0.9); net.set_update_algorithm(rule)
            steps = []
            for _ in range(3):
                r = train_step(net, batch)
                steps.append((r["net_out"].copy(), r["pzx"].copy(), r["in_diff"].copy()))
            res[name] = (steps, net.to_layers())
        for a, b in zip(res["orig"][0], res["pad"][0]):
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), rule
        cut = model_tools.unpad_cells_layers(res["pad"][1], [H, H])       # raises unless the padding is still exactly zero
        for La, Lb in zip(res["orig"][1], cut):
            assert all(np.array_equal(x, y) for x, y in zip(La["params"], Lb["params"])), rule
    bad = [dict(L) for L in padded]
    bad[0] = dict(bad[0], params=[p.copy() for p in bad[0]["params"]]); bad[0]["params"][2][-1] = 0.25     # a padded cell's bias
    with pytest.raises(ValueError, match="not zero"):
        model_tools.unpad_cells_layers(bad, [H, H])


def test_pad_cells_refuses_what_it_cannot_keep_equivalent(tmp_path):
    cfg = dict(kind="BiLstmParallel", layers=2, H=5, D=7, K=9, proj=6, proj_act="Tanh")
    layers = synth.make_model(**cfg)
    direct_sigmoid = [layers[0], dict(type="Sigmoid", input_dim=10, output_dim=10, params=[])] + [dict(type="AffineTransform", input_dim=10, output_dim=9,
                      params=[np.zeros((9, 10), np.float32), np.zeros(9, np.float32)])]
    with pytest.raises(ValueError, match="Sigmoid"):
        model_tools.pad_cells_layers(direct_sigmoid)
    with pytest.raises(ValueError, match="last layer"):
        model_tools.pad_cells_layers(layers[:1])
    assert model_tools.pad_cells_layers(synth.make_model(**dict(cfg, H=8))) [0]["output_dim"] == 16     # nothing to do: unchanged
