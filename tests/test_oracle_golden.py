"""CPU: the oracle (oracle/eesen_oracle.c via oracle/net.py) against the committed golden vectors, which
are outputs of the reference itself (oracle/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

from oracle import net as onet
from tests.util import GOLDEN, GOLDEN_DROPOUT, golden_masks, load_golden, rel_err, valid_mask


@pytest.mark.parametrize("name", GOLDEN)
def test_oracle_matches_reference_outputs(name):
    cfg, layers, batch, g = load_golden(name)
    ora = onet.OracleNet(layers, "f32")
    ora.set_train_options(1.0, 0.0)
    o = onet.train_step(ora, batch, "f32")
    # same algorithm, same fp32 type; only the GEMM summation order (OpenBLAS vs plain loops) differs
    assert rel_err(o["net_out"], g["net_out"]) < 2e-6
    for k in ("alpha", "beta"):
        assert np.array_equal(o[k] == -1e30, g[k] == -1e30), f"{k}: sentinel pattern differs"
        m = g[k] != -1e30
        assert np.max(np.abs(o[k][m] - g[k][m]) / np.maximum(1.0, np.abs(g[k][m]))) < 2e-6
    assert rel_err(o["pzx"], g["pzx"]) < 1e-6
    assert rel_err(o["diff"], g["diff"]) < 2e-5
    assert rel_err(o["in_diff"], g["in_diff"]) < 2e-5
    assert rel_err(ora.get_params(), g["params_after"]) < 2e-5
    grad_ref = g["params"].astype(np.float64) - g["params_after"]     # lr = 1, momentum = 0
    assert rel_err(ora.fresh_grads_flat(), grad_ref) < 5e-5
    ne, nr = onet.ctc_error_rate_mseq(g["net_out"], batch.T, batch.S, batch.lens, batch.label_ids, batch.label_off)
    assert [ne, nr] == g["errors"].tolist()


@pytest.mark.parametrize("name", GOLDEN)
def test_fp64_arbiter_agrees(name):
    """The double-precision build of the same restatement brackets both fp32 results."""
    cfg, layers, batch, g = load_golden(name)
    ora = onet.OracleNet(layers, "f64")
    ora.set_train_options(1.0, 0.0)
    o = onet.train_step(ora, batch, "f64")
    vm = valid_mask(batch.lens, batch.T, batch.S)
    assert rel_err(g["net_out"][vm], o["net_out"][vm]) < 1e-5
    assert rel_err(g["pzx"], o["pzx"]) < 1e-5
    assert rel_err(g["diff"], o["diff"]) < 1e-4
    assert rel_err(g["params_after"], ora.get_params()) < 1e-4


def test_padded_rows_of_diff_are_zero():
    cfg, layers, batch, g = load_golden("ragged_bi")
    assert np.all(g["diff"][~valid_mask(batch.lens, batch.T, batch.S)] == 0)
    assert np.all(g["in_diff"][~valid_mask(batch.lens, batch.T, batch.S)] == 0)


@pytest.mark.parametrize("name", GOLDEN_DROPOUT)
def test_oracle_matches_reference_dropout_outputs(name):
    """Dropout fixtures: outputs of the reference run with the masks its host RNG drew (stored alongside)."""
    cfg, layers, batch, g = load_golden(name)
    ora = onet.OracleNet(layers, "f32")
    ora.set_train_options(1.0, 0.0)
    H = cfg["H"]
    for li, fwd, rec, coin in golden_masks(layers, g):
        ora.set_dropout_masks(li, fwd=fwd, rec_fw=None if rec is None else rec[:, :H], rec_bw=None if rec is None else rec[:, H:],
                              twiddle_apply_forward=bool(coin))
    o = onet.train_step(ora, batch, "f32")
    vm = valid_mask(batch.lens, batch.T, batch.S)
    assert rel_err(o["net_out"][vm], g["net_out"][vm]) < 2e-6
    assert rel_err(o["in_diff"], g["in_diff"]) < 2e-5
    assert rel_err(ora.get_params(), g["params_after"]) < 2e-5
