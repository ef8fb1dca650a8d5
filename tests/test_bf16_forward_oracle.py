"""CPU: the numpy restatement of the "bf16 forward" variant (oracle/bf16_forward.py) is pinned to the reference.

With both roundings off it is the plain Net::Propagate and must reproduce the golden fixtures made from the reference itself
(tests/golden/*.npz, oracle/make_golden.py); the roundings are then the only thing the variant's arbiter adds."""
import numpy as np
import pytest

from oracle import bf16_forward as bf
from tests.util import load_golden, rel_err, valid_mask


@pytest.mark.parametrize("name", ["tiny_bi", "small_uni", "small_bi", "proj_bi", "ragged_bi"])
def test_plain_forward_equals_the_reference_fixture(name):
    cfg, layers, batch, g = load_golden(name)
    out = bf.forward(layers, batch.feats, batch.lens, batch.T, batch.S)
    vm = valid_mask(batch.lens, batch.T, batch.S)
    assert rel_err(out[vm], g["net_out"][vm]) < 2e-6


def test_round_bf16_is_round_to_nearest_even():
    x = np.array([1.0, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -7, 1.0 + 3 * 2.0 ** -8, -0.3, 1e-30, 65504.0, 0.0], np.float32)
    r = bf.round_bf16(x)
    assert r[0] == 1.0 and r[1] == 1.0                       # tie -> even mantissa (1.0)
    assert r[2] == np.float32(1.0 + 2.0 ** -7)
    assert r[3] == np.float32(1.0 + 2.0 ** -6)                # tie -> even (1 + 2/128)
    assert np.all((r.view(np.uint32) & 0xFFFF) == 0)
    assert np.all(np.abs(r - x) <= np.abs(x) * 2.0 ** -8)


def test_planes_carry_8_then_9_more_bits_each():
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(4096) * np.exp(rng.uniform(-8, 8, 4096))).astype(np.float32)
    e1 = np.max(np.abs(bf.bf16_planes(x, 1) - x) / np.abs(x))
    e2 = np.max(np.abs(bf.bf16_planes(x, 2) - x) / np.abs(x))
    assert 2.0 ** -10 < e1 <= 2.0 ** -8 and 2.0 ** -19 < e2 <= 2.0 ** -16      # half an ulp of 8, then of 8 + 9 = 17 significant bits
    assert np.array_equal(bf.bf16_planes(x, 3), x)                      # three planes: every fp32 value exactly
    assert np.array_equal(bf.bf16_planes(x, 1), bf.round_bf16(x))


def test_the_two_roundings_move_the_output_by_what_eight_bits_allow():
    cfg, layers, batch, g = load_golden("small_bi")
    vm = valid_mask(batch.lens, batch.T, batch.S)
    plain = bf.forward(layers, batch.feats, batch.lens, batch.T, batch.S)
    gemm = bf.forward(layers, batch.feats, batch.lens, batch.T, batch.S, bf16_gemm=True)
    both = bf.forward(layers, batch.feats, batch.lens, batch.T, batch.S, bf16_gemm=True, bf16_rec=True)
    e1, e2 = rel_err(gemm[vm], plain[vm]), rel_err(both[vm], gemm[vm])
    assert 1e-5 < e1 < 5e-2 and 1e-6 < e2 < 5e-2
