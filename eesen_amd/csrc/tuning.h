// tuning.h -- EVERY run-time switch of libeesen_hip.so, in one place.
//
// The product path has no tuning to do: all defaults below are what bench.py measures.  The switches exist for three reasons --
// (i) fallbacks a deployment may need, (ii) A/B arms of the parity tests (a kernel against its slower twin), (iii) diagnostics --
// and are read from the environment ONCE PER NET, when it is created (tests create a fresh Net per arm).  Everything else that
// rounds 1 and 2 carried as experiment knobs (CU-masked side streams, cooperative launches, poll-wave / sleep / shard-count
// variants, the L2-local hand-off, synthetic interference GEMMs, timing probes; round 4's arms that lost their A/B: the 4 x 32
// forward tile, two chains per workgroup on narrow layers, W_m as one bf16 plane, the backward kernel's early operand request)
// has been measured, written up in DESIGN.md section 9 and REMOVED from the code.
//
//   variable                default  meaning
//   ---- fallbacks ---------------------------------------------------------------------------------------------------------
//   EESEN_PERSISTENT        1        0: one launch per recurrence step (lstm.hip) instead of one persistent launch per layer pass
//   EESEN_OVERLAP           auto     weight-gradient GEMMs on a side stream under the next recurrence (auto: on with persistent
//                                    kernels on layers of <= 512 cells whose backward pass takes a 4- / 8-sequence tile; off beside
//                                    the 16-sequence tiles, e.g. --num-sequence 64 at 512 cells)
//   EESEN_SPIN_LIMIT        400000   bound of the in-kernel hand-off spins (x10 with a communicator attached); 0 in tests forces a time-out
//   EESEN_GEMM_MODE         half     arithmetic of every dense GEMM (gemm.hip): half = two fp16 planes per operand, three products on
//                                    v_mfma_f32_32x32x16_f16, per-row / per-column power-of-two scales measured on the device (round 6);
//                                    split = three bf16 planes, six products (the default of rounds 2-5); f32 = v_mfma_f32_32x32x2_f32
//                                    (an exact fp32 fmaf chain).  All three fp32-class against fp64 (tests/test_gpu_gemm.py)
//   EESEN_HOST_FEATURE_PIPES unset   (trainers) run recognised feature pipes as host processes instead of on the device
//   ---- A/B arms of the tests --------------------------------------------------------------------------------------------
//   EESEN_BWD_Q4            1        0: 8-sequence backward tile instead of the 4 x 32 tile (H <= 512)
//   EESEN_BWD_Q4_ST8        1        0: the 16 x 16 backward tile instead of the 4 x 32 tile with TWO 4-sequence tiles per workgroup where
//                                    the one-tile grid does not fit (H <= 512 at --num-sequence 64); 2: the two-tile form wherever it
//                                    applies (bit-identical to the one-tile form: the tests' arm)
//   EESEN_BWD_KSPLIT        1        0: 16 x 16 backward tile instead of the K-split kernel (wide layers)
//   EESEN_FWD_MUX           1        0: two sequence windows instead of the time-multiplexed forward kernel (S = 64 at H = 1024)
//   EESEN_BWD_MUX           1        0: the same for the K-split backward kernel
//   EESEN_FWD_SPLIT         1        0: every recurrence on the fp32-input MFMA (the forward pass bit-identical to the per-step kernels)
//                                    instead of on 16-bit planes of both operands (fp32-class; which: EESEN_FWD_F16, EESEN_BWD_F16)
//   EESEN_FWD_F16           1        0: the narrow forward tile on three bf16 planes / six products (rounds 4-5) and the wide tile on the
//                                    fp32-input MFMA, instead of both on two fp16 planes / three products (round 6)
//   EESEN_BWD_F16           1        0: the K-split backward tile of wide layers on the fp32-input MFMA (and its time-multiplexed form at
//                                    S = 64) instead of on two fp16 planes per operand, the gate gradients published a second time as
//                                    planes with a per-(producer, sequence) power of two (round 6)
//   EESEN_XCD_MAP           1        0: plain workgroup -> role map instead of the XCD-aware one
//   EESEN_CTC_WAVES         0        n: the CTC lattice sweep as n wavefronts per lattice where that instantiation exists (read when a
//                                    Ctc is created; 0: one wave up to 256 lattice positions, 4 / 8 / 16 / 16 for rows of 512 / 1024 /
//                                    2048 / 4096; bit-identical)
//   EESEN_FWD_T16_SMALL     1        0: batches of <= 16 sequences on narrow layers take the 32 x 4 fp32 forward tile (twice the workgroups)
//                                    instead of the 16 x 8 tile and its bf16-pipe kernel (recipe shape, --num-sequence 10: 21.5 / 20.0 ms)
//   EESEN_FWD_NARROW2       1        0: the wide 16 x 16 forward tile instead of TWO workgroups of the narrow bf16-pipe tile per CU where
//                                    the one-per-CU grid is full (--num-sequence 64 at 512 cells: 512 workgroups, seen co-resident
//                                    by a one-time census)
//   EESEN_GPU_SHARE         1        n: the persistent grids are sized against 1/n of the device's CUs (n processes on one GPU); read once
//                                    per PROCESS
//   EESEN_GATE_FWD          auto     next layer's input GEMM gated under the forward recurrence (auto: f32 GEMM mode only)
//   EESEN_FWD_MID           1        0: the next layer's input GEMM waits for the whole forward recurrence (no early middle part):
//                                    what counter-collecting runs (rocprofv3 --pmc lets ONE kernel run at a time) must set
//   EESEN_SIDE_LDS_KB       auto     occupancy cap of the side-stream GEMMs (unused dynamic LDS; auto: 48 half and split / 32 f32)
//   ---- diagnostics -------------------------------------------------------------------------------------------------------
//   EESEN_TRACE             0        1: in-kernel s_memtime timeline of workgroup 0, printed when the Net is destroyed
//   EESEN_PRINT_FLIGHT      unset    print the measured increment flight and the first-poll delays in use
//   EESEN_POLL_NS           unset    "fwd,bwd": first-poll delays in ns instead of the tuned ones (400 / 280, 420 beside side-stream GEMMs; net.cpp)
//   ---- data-parallel exchange (comm.cpp) ----------------------------------------------------------------------------------
//   EESEN_RCCL_LIBRARY      librccl.so.1   library to dlopen for the nccl* entry points (tests: the stand-in)
//   EESEN_COMM_TIMEOUT_S    600      watchdog: seconds after which an unfinished collective aborts the communicator
//   EESEN_COMM_DEFER        auto     a backward pass's gradient buckets all-reduced when its LAST recurrence has run (1: top-down, one after
//                                    the other on the communication stream) or each as soon as its layer's gradients are enqueued (0).
//                                    auto (round 6): deferred whenever some layer's persistent backward grid leaves < 256 registers per
//                                    SIMD lane on its CUs -- RCCL's all-reduce workgroups (248-256 VGPRs x 512 threads) then cannot be
//                                    resident beside it, would run in the gaps between recurrences and hold CUs the next recurrence
//                                    wants while a peer is late (measured beside an RCCL-shaped stand-in: profiles/r06_rccl_shaped_soak.json).
//                                    Every BASELINE shape is deferred; layers of <= 256 cells overlap.  Same buckets, order and sums.
//   EESEN_COMM_PORT         MASTER_PORT+17  rendezvous port of the hosts that create the communicator from the environment
#pragma once
#include <cstdlib>

namespace eesen {

struct Tuning {
  int persistent = 1;
  int overlap = -1, gate_fwd = -1, side_lds_kb = -1;   // -1: decided by the Net (see above)
  int spin_limit = 400000;
  bool spin_limit_set = false;
  int bwd_q4 = 1, bwd_q4_st8 = 1, fwd_narrow2 = 1, fwd_t16_small = 1, bwd_ksplit = 1, fwd_mux = 1, bwd_mux = 1, xcd_map = 1, fwd_mid = 1, fwd_split = 1, fwd_f16 = 1, bwd_f16 = 1;
  int comm_defer = -1;   // -1: decided per minibatch from the backward plans (Net::exchange_deferred_for_minibatch)
  int trace = 0;
  bool print_flight = false;
  const char* poll_ns = nullptr;

  static Tuning from_env() {
    Tuning t;
    auto num = [](const char* name, int dflt) {
      const char* e = getenv(name);
      return e && *e ? atoi(e) : dflt;
    };
    t.persistent = num("EESEN_PERSISTENT", 1);
    t.overlap = num("EESEN_OVERLAP", -1);
    t.gate_fwd = num("EESEN_GATE_FWD", -1);
    t.fwd_mid = num("EESEN_FWD_MID", 1);
    t.side_lds_kb = num("EESEN_SIDE_LDS_KB", -1);
    t.spin_limit_set = getenv("EESEN_SPIN_LIMIT") != nullptr;
    t.spin_limit = num("EESEN_SPIN_LIMIT", 400000);
    t.bwd_q4 = num("EESEN_BWD_Q4", 1);
    t.bwd_q4_st8 = num("EESEN_BWD_Q4_ST8", 1);
    t.fwd_narrow2 = num("EESEN_FWD_NARROW2", 1);
    t.fwd_t16_small = num("EESEN_FWD_T16_SMALL", 1);
    t.bwd_ksplit = num("EESEN_BWD_KSPLIT", 1);
    t.fwd_mux = num("EESEN_FWD_MUX", 1);
    t.bwd_mux = num("EESEN_BWD_MUX", 1);
    t.fwd_split = num("EESEN_FWD_SPLIT", 1);
    t.fwd_f16 = num("EESEN_FWD_F16", 1);
    t.bwd_f16 = num("EESEN_BWD_F16", 1);
    t.xcd_map = num("EESEN_XCD_MAP", 1);
    t.comm_defer = num("EESEN_COMM_DEFER", -1);
    t.trace = num("EESEN_TRACE", 0);
    t.print_flight = getenv("EESEN_PRINT_FLIGHT") != nullptr;
    t.poll_ns = getenv("EESEN_POLL_NS");
    return t;
  }
};

}  // namespace eesen
