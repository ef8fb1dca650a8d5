"""CPU: the <Nnet> model-file tool (text + binary round trips, header handling, refusals)."""
import numpy as np
import pytest

from eesen_amd import nnet_io, synth


@pytest.mark.parametrize("cfg_name", ["tiny_bi", "small_uni"])
@pytest.mark.parametrize("binary", [False, True])
def test_roundtrip(tmp_path, cfg_name, binary):
    cfg = synth.config(cfg_name)
    layers = synth.make_model(max_grad=50.0, learn_rate_coef=0.5, **cfg)
    p = str(tmp_path / "m.nnet")
    nnet_io.write_nnet(p, layers, binary=binary)
    back = nnet_io.read_nnet(p)
    assert [(l["type"], l["input_dim"], l["output_dim"]) for l in back] == [(l["type"], l["input_dim"], l["output_dim"]) for l in layers]
    assert np.array_equal(nnet_io.flatten_params(back), nnet_io.flatten_params(layers))   # text uses shortest round-trip digits
    assert back[0]["max_grad"] == 50.0 and back[0]["learn_rate_coef"] == 0.5
    assert nnet_io.num_params(back) == nnet_io.num_params(layers)


def test_param_count_matches_reference_formula():
    # BiLSTM layer = 2 * (4H*D + 4H*H + 4H + 3H)  (/root/reference/src/net/bilstm-layer.h:991-998); cfg2 total 21 211 182
    cfg = synth.config("cfg2")
    H, D, K = cfg["H"], cfg["D"], cfg["K"]
    n = 2 * (4 * H * D + 4 * H * H + 7 * H) + 3 * 2 * (4 * H * 2 * H + 4 * H * H + 7 * H) + (K * 2 * H + K)
    assert n == 21211182


@pytest.mark.parametrize("binary", [False, True])
def test_dropout_options_roundtrip(tmp_path, binary):
    """The nine dropout tokens of BiLstm (bilstm-layer.h:331-373, :435-455) survive write -> read with their values."""
    cfg = synth.config("tiny_bi")
    layers = synth.make_model(**cfg)
    layers[0]["dropout"] = dict(forward=0.25, fw_step=True, recurrent=0.125, rec_seq=True, nml=True, twiddle=True)
    layers[1]["dropout"] = dict(recurrent=0.5, rec_step=True, rnndrop=True)
    p = str(tmp_path / "d.nnet")
    nnet_io.write_nnet(p, layers, binary=binary)
    back = nnet_io.read_nnet(p)
    assert back[0]["dropout"] == layers[0]["dropout"] and back[1]["dropout"] == layers[1]["dropout"]
    assert "dropout" not in back[2]
    assert np.array_equal(nnet_io.flatten_params(back), nnet_io.flatten_params(layers))


def test_headerless_layer_data_is_accepted(tmp_path):
    """All header tokens are optional on read (bilstm-layer.h:322-373)."""
    cfg = synth.config("tiny_bi")
    layers = synth.make_model(**cfg)
    p = str(tmp_path / "m.nnet")
    nnet_io.write_nnet(p, layers, binary=False, write_dropout_tokens=False)
    assert np.array_equal(nnet_io.flatten_params(nnet_io.read_nnet(p)), nnet_io.flatten_params(layers))


@pytest.mark.parametrize("binary", [False, True])
def test_accumulators_roundtrip(tmp_path, binary):
    """<BiLstmAccus> / <AffineAccus> precede the weights (bilstm-layer.h:376-395, affine-trans-layer.h:99-106)."""
    cfg = synth.config("tiny_bi")
    layers = synth.make_model(**cfg)
    rng = np.random.default_rng(1)
    for L in layers:
        if L["params"]:
            L["accu"] = [rng.random(p.shape).astype(np.float32) for p in L["params"]]
    p = str(tmp_path / "m.nnet")
    nnet_io.write_nnet(p, layers, binary=binary)
    back = nnet_io.read_nnet(p)
    for a, b in zip(back, layers):
        assert np.array_equal(nnet_io.flatten_params([a]), nnet_io.flatten_params([b]))
        if b["params"]:
            assert all(np.array_equal(x, y) for x, y in zip(a["accu"], b["accu"]))
