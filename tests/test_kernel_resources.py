"""CPU (no GPU): the register / LDS budgets the library's schedule decisions rest on, read from the BUILT libeesen_hip.so's gfx950
code objects (tools/kernel_resources.py: llvm-readelf --notes) and pinned.

Why: co-residency is arithmetic on these numbers -- a persistent recurrence grid holds one 512-thread workgroup
(two waves per SIMD) on every CU; what else fits on that CU is 512 registers per SIMD lane minus 2 x the tile's allocation
(blocks of 8), and 160 KB of LDS minus the tile's.  `q4<8,4>` at 164 (allocated 168) leaves 176: exactly one wave per SIMD of the
side-stream split GEMM (168).  One more block of 8 and the weight-gradient GEMMs no longer run under the recurrence; the only
symptom would be a slower step.  A compiler bump or a one-line edit that moves a kernel over its budget fails HERE instead.
The table the budgets were read from is committed as profiles/kernel_resources.md (regenerate: python tools/kernel_resources.py).
"""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

SIMD_VGPRS = 512          # registers per lane of one SIMD (gfx950: unified VGPR/AGPR file)
CU_LDS = 160 * 1024


def alloc(v):             # registers are allocated in blocks of 8
    return (v + 7) & ~7


@pytest.fixture(scope="module")
def table():
    import kernel_resources as kr
    from eesen_amd import build
    rows = kr.kernels(build.build())
    return {r["name"]: r for r in rows}


# kernel -> (max vgprs, max static LDS bytes).  The persistent tiles of every BASELINE shape + what has to fit beside them.
BUDGETS = {
    # cfg2 (4 x 512, S = 32): backward 4 x 32 tile, forward narrow bf16-pipe tile, the side-stream / early-middle GEMM flavour
    "lstm_bwd_persistent_q4_kernel<8,4>": (168, 74 * 1024),
    "lstm_fwd_persistent_bf_kernel<2,2,3,3,false>": (112, 19 * 1024),
    "lstm_fwd_persistent_bf_kernel<2,2,2,2,true>": (96, 19 * 1024),     # round 6: two fp16 planes, three products (the default narrow tile)
    # cfg2 at S = 64: two 4-sequence tiles per workgroup; recipe width 320: <6,4>; H = 256: <4,4> (the one tile RCCL fits beside)
    "lstm_bwd_persistent_q4_kernel<8,8>": (192, 81 * 1024),
    "lstm_bwd_persistent_q4_kernel<6,4>": (160, 9 * 1024),
    "lstm_bwd_persistent_q4_kernel<4,4>": (128, 9 * 1024),
    "lstm_bwd_persistent_kernel<8,16,false>": (184, 9 * 1024),
    # cfg4 / cfg5 (1024 cells): K-split backward (+ its time-multiplexed form at S = 64), wide forward fp32 / bf16 / multiplexed
    "lstm_bwd_persistent_ksplit_kernel<4>": (232, 34 * 1024),
    "lstm_bwd_persistent_ksplit_mux_kernel<4>": (248, 35 * 1024),
    "lstm_bwd_persistent_ksplit_h_kernel<4>": (256, 40 * 1024),     # round 6: the K-split tile on two fp16 planes (fills the register file)
    "lstm_fwd_persistent_kernel<4,1,4,false,true>": (208, 35 * 1024),
    "lstm_fwd_persistent_bf_kernel<4,4,1,2,false>": (192, 35 * 1024),
    "lstm_fwd_persistent_bf_kernel<4,4,2,2,true>": (216, 35 * 1024),    # round 6: the wide tile on two fp16 planes (fp32-class; replaces the fp32-input tile)
    "lstm_fwd_persistent_mux_kernel<4,4,true>": (224, 35 * 1024),
}


def test_no_kernel_spills_vector_registers_or_uses_scratch(table):
    """Zero spilled VGPRs and zero scratch in every kernel of the library -- with the exceptions written down here: two
    m/n-contiguous flavours of the 256 x 256 split GEMM spill 3 / 6 registers in their prologue (they run at the same 203-206 TF
    as the flavour without, profiles/r05_bench_line.json), and three guarded instantiations of the non-default f32-MFMA GEMM
    (EESEN_GEMM_MODE=f32) keep a small indexed array in scratch."""
    allowed_spill = {"gemm_f32_split_bf16_big_kernel<false,true>": 4, "gemm_f32_split_bf16_big_kernel<true,false>": 8}
    allowed_scratch = {"gemm_f32_split_bf16_big_kernel<false,true>": 32, "gemm_f32_split_bf16_big_kernel<true,false>": 32,
                       "gemm_f32_mfma_kernel<false,false,false>": 96, "gemm_f32_mfma_kernel<false,true,false>": 64, "gemm_f32_mfma_kernel<true,false,false>": 64}
    assert len(table) >= 140
    bad = []
    for n, r in table.items():
        if r["vgpr_spill"] > allowed_spill.get(n, 0) or r["scratch"] > allowed_scratch.get(n, 0):
            bad.append((n, r["vgpr_spill"], r["scratch"]))
    assert not bad, bad
    # every recurrence / CTC kernel: none at all
    for n, r in table.items():
        if n.startswith(("lstm_", "ctc_")):
            assert r["vgpr_spill"] == 0 and r["scratch"] == 0, (n, r)


def test_persistent_tiles_stay_inside_their_budgets(table):
    for n, (vmax, lmax) in BUDGETS.items():
        assert n in table, f"{n} is no longer instantiated: update BUDGETS and DESIGN.md's front page"
        r = table[n]
        assert r["max_threads"] == 512 and r["agprs"] == 0
        assert r["vgprs"] <= vmax, f"{n}: {r['vgprs']} VGPRs > budget {vmax}"
        assert r["lds"] <= lmax, f"{n}: {r['lds']} B LDS > budget {lmax}"


def test_side_stream_gemm_fits_beside_the_cfg2_backward_tile(table):
    """net.cpp's overlap rule: the weight-gradient GEMMs (gemm_f32_split_bf16_kernel, 256 threads = one wave per SIMD) run on the
    side stream UNDER the next-lower layer's backward recurrence (q4<8,4>, two waves per SIMD, one workgroup on every CU)."""
    q4 = table["lstm_bwd_persistent_q4_kernel<8,4>"]
    side = [r for n, r in table.items() if n.startswith(("gemm_f32_split_bf16_kernel<", "gemm_f32_split_f16_kernel<"))]
    assert len(side) == 16
    for g in side:
        assert g["max_threads"] == 256 and g["vgprs"] <= 168
        assert 2 * alloc(q4["vgprs"]) + alloc(g["vgprs"]) <= SIMD_VGPRS, (q4["vgprs"], g["vgprs"])
        assert q4["lds"] + g["lds"] <= CU_LDS
    # ... and NOT two of them (the side stream's occupancy cap asks for one per CU beside a recurrence: EESEN_SIDE_LDS_KB = 48)
    assert 2 * alloc(q4["vgprs"]) + 2 * alloc(min(g["vgprs"] for g in side)) > SIMD_VGPRS


def test_two_narrow_forward_workgroups_per_cu_and_the_early_gemm_beside_one(table):
    for name in ("lstm_fwd_persistent_bf_kernel<2,2,3,3,false>", "lstm_fwd_persistent_bf_kernel<2,2,2,2,true>"):
        bf = table[name]
        assert 4 * alloc(bf["vgprs"]) <= SIMD_VGPRS and 2 * bf["lds"] <= CU_LDS          # --num-sequence 64 at 512 cells: two per CU
        g = max(r["vgprs"] for n, r in table.items() if n.startswith(("gemm_f32_split_bf16_kernel<", "gemm_f32_split_f16_kernel<")))
        assert 2 * alloc(bf["vgprs"]) + alloc(g) <= SIMD_VGPRS                              # "the middle first": one GEMM workgroup beside ONE tile


def test_exchange_schedule_table(table):
    """DESIGN.md section 7's residency table: which persistent backward tile leaves the 256 registers per SIMD lane an RCCL all-reduce
    workgroup needs (Net::exchange_deferred_for_minibatch: deferred below that).  Every BASELINE shape must come out deferred, the
    256-cell tile overlapped -- from the numbers in the code object, not from a comment."""
    src = open(os.path.join(ROOT, "eesen_amd", "csrc", "net.h")).read()
    need = int(re.search(r"constexpr int kRcclVgprsPerSimdLane = (\d+);", src).group(1))
    assert need == 256
    free = lambda n: SIMD_VGPRS - 2 * alloc(table[n]["vgprs"])
    deferred = ["lstm_bwd_persistent_q4_kernel<8,4>", "lstm_bwd_persistent_q4_kernel<8,8>", "lstm_bwd_persistent_q4_kernel<6,4>",
                "lstm_bwd_persistent_kernel<8,16,false>", "lstm_bwd_persistent_ksplit_kernel<4>", "lstm_bwd_persistent_ksplit_mux_kernel<4>",
                "lstm_bwd_persistent_ksplit_h_kernel<4>"]
    for n in deferred:
        assert free(n) < need, (n, free(n))
    for n in ["lstm_bwd_persistent_q4_kernel<4,4>", "lstm_bwd_persistent_q4_kernel<2,4>"]:
        assert free(n) >= need, (n, free(n))


def test_the_stand_in_has_rccls_footprint():
    """tests/native/libfake_rccl.so, FAKE_RCCL_SHAPE=rccl: the kernel the residency tests run beside the persistent grids must be as
    large as ncclDevKernel_Generic_* of RCCL 2.27.7 (profiles/r05_rccl_kernel_descriptors.md: 512 threads, 248-256 VGPRs, 37 664 B LDS)."""
    import kernel_resources as kr
    from tests.test_gpu_multirank import fake_rccl_path
    rows = {r["name"]: r for r in kr.kernels(fake_rccl_path())}
    shaped = [r for n, r in rows.items() if n.startswith("fake_allreduce_rccl_shaped_kernel<")]
    plain = [r for n, r in rows.items() if n.startswith("fake_allreduce_kernel<")]
    assert len(shaped) == 3 and len(plain) == 3
    for r in shaped:
        assert r["max_threads"] == 512 and 248 <= r["vgprs"] <= 256 and r["lds"] == 37664 and r["vgpr_spill"] == 0
    for r in plain:
        assert r["vgprs"] <= 64 and r["lds"] < 1024


def test_committed_table_matches_the_built_library(table):
    """profiles/kernel_resources.md is what the budgets were read from: it must describe the library that is built."""
    text = open(os.path.join(ROOT, "profiles", "kernel_resources.md")).read()
    for n in BUDGETS:
        m = re.search(r"\| `" + re.escape(n) + r"` \| (\d+) \| (\d+) \| \d+ \| \d+ \| (\d+) \|", text)
        assert m, n
        assert (int(m.group(2)), int(m.group(3))) == (table[n]["vgprs"], table[n]["lds"]), f"{n}: regenerate with python tools/kernel_resources.py"
