// lstm_persistent.hip -- the whole time recurrence of one (Bi-)LSTM layer in ONE launch, for gfx950.
//
// Why this exists (measured, profiles/r01b_pmc_T250.md + tools/l2_probe.hip): the XCD L2s are invalidated at every
// kernel boundary, so the one-launch-per-step kernels of lstm.hip re-stream the whole recurrent matrix W_m
// (2 x 4 MB at H = 512) through the Infinity Fabric on EVERY step (TCC_MISS ~ 115 k lines = 15 MB per step), which is
// 2-3 us of a 6.5 us forward step and ~5 us of a 10 us backward step.  Here each workgroup loads its slice of W_m
// ONCE into VGPRs (as MFMA B operands) and keeps it there for all T steps; per step only the new cell outputs
// m_t (forward, 64 KB per direction) or gate gradients DG_t (backward, 256 KB per direction) cross the chip.
//
// Cross-workgroup hand-off per step (the placement-independent recipe of the CDNA4 guide, section 6 G16, form R1):
//   producer: payload with write-through (sc1) stores -> every storing wave drains vmcnt -> workgroup barrier ->
//             one lane bumps its shard of the (direction, sequence-tile) group's arrival counters (8 shards on separate
//             128-byte lines, relaxed agent-scope atomic);
//   consumer: 8 lanes of ONE wave (wave 7, after a first-poll delay; see EESEN_POLL_WAVE below) poll the 8 shards (relaxed,
//             agent scope, s_sleep between polls, BOUNDED spin) ->
//             workgroup barrier -> every wave reads the payload with plain loads of lines nobody has read before (see below;
//             L1-bypassing sc1 loads were measured slower).
// Workgroup roles are laid out so that, with the observed block -> XCD round-robin, the workgroups of one group share an L2
// (struct Role); that only changes how much crosses the fabric, never correctness.
// Every step writes row blocks that no one has read before in this launch, so no cache can hold a stale copy -- PROVIDED
// a step's row block starts on a 128-byte line (the launchers check it and fall back otherwise): an unaligned block
// shares its first line with the previous block, which caches it one step early.
// A workgroup can run at most one step ahead of the slowest one of its (direction, sequence-tile) group, and step t
// writes row block t while laggards still read row block t-1, so there is no write-after-read hazard.
// All workgroups must be co-resident: the host launches only after an occupancy check with margin (see coop_launch for why the
// launch itself is an ordinary one) and otherwise falls back to the per-step kernels; a spin that exceeds its bound raises an
// error word instead of hanging.
//
// Arithmetic is that of lstm.hip (same cell equations, K split over the 8 waves the same way): the fp32-input forward kernel
// (lstm_fwd_persistent_kernel) is bit-identical to the per-step path; the DEFAULT forward kernel of narrow layers since round 4,
// lstm_fwd_persistent_bf_kernel<.,2,3,3> (the exact 3-way bf16 split, EESEN_FWD_SPLIT=1), is fp32-class but not bit-identical; the
// backward pass differs in the last bits (FMA contraction, and the full-line operand fetch consumes each 32-float chunk as two
// 16-float halves); tests assert exactly that.
#include "kernels.h"

#include <atomic>
#include <map>
#include <mutex>

namespace eesen {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NW = 8;
constexpr int kSc1 = 16;  // cache-policy bit of the raw-buffer builtins: sc1 = write-through store / L1-bypassing load

// v_exp_f32 / v_rcp_f32 (1 ulp each) instead of the ~40-instruction libm expf and IEEE division: the cell update is on the
// per-step critical path; the error (~2e-7 relative) is three orders below the parity bar
// __frcp_rn compiles to the IEEE division sequence (v_div_scale / v_rcp / 3 fma / v_div_fmas / v_div_fixup, ~10 dependent
// instructions); v_rcp_f32 alone is 1 ulp -- three orders below the parity bar, and the cell update is on the per-step critical path
#define EESEN_RCP(x) __builtin_amdgcn_rcpf(x)
__device__ __forceinline__ float sigmoidf_(float x) { return EESEN_RCP(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f * EESEN_RCP(1.f + __expf(2.f * x)); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
}
// one float through a buffer resource; an offset at or above num_records (0x80000000) reads as zero without touching memory
__device__ __forceinline__ float ld1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
// 8 consecutive floats at row[k..k+7] (k, kmax multiples of 4) through the sc1 path; zeros outside
// Tried: consumers reading the handed-off rows with sc1 (L1-bypassing) loads.  Measured on MI355X
// this makes every workgroup pull its own copy through the Infinity Fabric (16 MB per backward step, ~27 GB/s per CU).
// Default: plain loads.  They are safe HERE because (i) every step reads row blocks that were never read before in this
// launch, so neither the CU's L1 nor the XCD's L2 can hold an older copy, (ii) the producers' sc1 stores write through
// and drop the line from their L2, and (iii) no load is issued before the arrival counters say every producer has
// drained its stores -- and they let the 16 workgroups of an XCD share one fabric fetch through the L2.
// Polling wave and back-off (round 2, measured on cfg2 with nothing overlapped: forward recurrences 14.2 -> 12.9 ms, backward
// 17.7 -> 17.1 ms; step 45.8 -> 44.5 ms with the default overlap).  The poller is wave 7, which has NO other memory traffic in
// flight: waves 0-1 finish the cell, store, drain, publish and prefetch the next step's epilogue operands from HBM, and vmcnt
// returns in order -- a poll issued by wave 0 came back only after its own counter increment had been acknowledged (hipcc puts an
// s_waitcnt vmcnt(0) in front of the prefetch: 0.5 us) AND the HBM prefetch had landed.  A free wave polling at once is worse
// than that (forward 16.6 ms): 256 eager pollers crowd the counter lines and delay the very increments they wait for.  So the
// first poll is DELAYED by about the time the peers' increments need to land (s_sleep 20 = 1280 cycles forward, 14 backward:
// flat optimum 16-24 / 12-16; 12 forward or 8 backward give the gain away, 48 costs 10 %), and then usually succeeds at once.
constexpr int EESEN_POLL_WAVE = 7;   // the polling wave (see above)
constexpr int EESEN_POLL_SLEEP = 1;  // s_sleep between polls (0 ... 16: flat)
// Branch-free on purpose: a lane that has nothing to read points its offset past the descriptor's num_records and the
// buffer unit returns zeros.  With predicated loads (`if (ok) load`) the compiler cannot count outstanding loads and puts
// one `s_waitcnt vmcnt(0)` in front of the whole MFMA chain, which serialised operand fetch (~1.9 us per backward step)
// and MFMA (~1.7 us); straight-line loads get per-chunk `vmcnt(n)` waits and the two overlap.
__device__ __forceinline__ void ld8_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, int k, int kmax, bool ok, float (&v)[8]) {
  constexpr int pol = 0;   // plain loads (see above); sc1 loads were measured slower
  constexpr unsigned kOob = 0x80000000u;  // >= num_records (0x7fffffff): reads as zero
  const unsigned oa = (ok && k < kmax) ? byte_off : kOob;
  const unsigned ob = (ok && k + 4 < kmax) ? byte_off + 16 : kOob;
  const f32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, oa, 0, pol);
  const f32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, ob, 0, pol);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
  v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
__device__ __forceinline__ float ror8(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x128, 0xf, 0xf, true));
}
__device__ __forceinline__ void ld8_plain(const float* __restrict__ row, int k, int kmax, bool ok, float (&v)[8]) {
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (ok && k < kmax) a = *reinterpret_cast<const float4*>(row + k);
  if (ok && k + 4 < kmax) b = *reinterpret_cast<const float4*>(row + k + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// Workgroup -> (unit group, direction, sequence group).  The grid is 1-D; with `xcd` set the (direction, sequence group)
// pair varies FASTEST, so that -- with the dispatcher's observed block b -> XCD b % 8 placement -- all workgroups that
// exchange rows with each other (same direction, same sequence group) sit behind the same L2 when there are 8 such
// groups (2 directions x 4 sequence tiles at S = 32 backward), or behind two L2s when there are 4 (forward): each
// L2 then pulls only its own group's rows through the fabric instead of every group's.  Pure speed hint: the hand-off
// protocol does not depend on the placement.
struct Role {
  int nblk, ndir, nz, xcd;
  __device__ __forceinline__ int combo(int b) const { return xcd ? b % (ndir * nz) : b / nblk; }
  __device__ __forceinline__ int unit_group(int b) const { return xcd ? b / (ndir * nz) : b % nblk; }
  __device__ __forceinline__ int dir(int b) const { return combo(b) % ndir; }
  __device__ __forceinline__ int seq_group(int b) const { return combo(b) / ndir; }
};

// Arrival counters are sharded 8 ways (shard = unit group & 7, one 128-byte line each) so that the increments of a
// step do not serialise on one address.  Lanes 0-7 of wave 0 each poll one shard until it reaches its own target
// (workgroups in that shard x steps).  Returns false (and raises *err) when the bound is hit or a peer gave up.
//
// The first-poll delay.  A poll issued before the peers' increments have landed costs a whole extra round trip AND slows the
// very increments it waits for; one issued late is pure waiting.  Round 2 found the optimum by hand as s_sleep constants (20
// forward, 14 backward: about ONE increment flight) for one clock and one box of the pool.  It is a hardware latency, so it is
// now MEASURED: when a Net is created, a two-workgroup ping-pong (handoff_flight_ns below) times the flight of an agent-scope
// increment between two CUs of this device against the constant-rate wall clock, and each wait's delay is that flight times a
// dimensionless factor (net.cpp) -- passed to the kernels in wall-clock ticks and waited for on the wall clock, so neither the
// measurement nor the wait depends on where DVFS has the shader clock at the time.  What was tried instead and
// measured: controllers that steer the delay from first-poll misses.  Per workgroup they are unstable (a workgroup that
// lengthens its delay publishes late, its peers' polls then miss and lengthen theirs: the delays ran to the limit, cfg2 forward
// pass 12.3 -> 23.3 ms); with one common delay steered from the grid's miss rate they run away as well, because the workgroups
// that finish their step early miss at ANY delay (the step is paced by the slowest one, for which the right delay is exactly the
// flight).  The flight is the quantity to know; the miss rate says nothing about it.
// waits `ticks` of the constant-rate wall clock (wall_clock64: 100 MHz, 10 ns): independent of where DVFS has the shader clock
__device__ __forceinline__ void sleep_ticks(int ticks) {
  if (ticks <= 0) return;
  const unsigned long long t0 = wall_clock64();
  while ((long long)(wall_clock64() - t0) < (long long)ticks) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ bool wait_counters(unsigned* cnt, unsigned nblk, unsigned step, unsigned* err, int spin_limit, int lane, int delay) {
  const unsigned mine = lane < kShards ? ((nblk - lane + kShards - 1) / kShards) * step : 0u;
  sleep_ticks(delay);  // nobody can have arrived yet: the peers are still in their own step
  for (int spins = 0; spins < spin_limit; ++spins) {
    bool ok = true;
    if (lane < kShards) ok = __hip_atomic_load(cnt + lane * kShardStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= mine;
    if (__all(ok)) return true;
    if ((spins & 1023) == 1023 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    __builtin_amdgcn_s_sleep(EESEN_POLL_SLEEP);
  }
  if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}

// Two workgroups (two CUs) hand a counter back and forth `rounds` times with the same agent-scope relaxed atomics the hand-off
// uses; out[0] = wall-clock ticks (10 ns) of the whole exchange.  One round = two flights (A's store becomes visible to B's poll, B's to A's).
__global__ __launch_bounds__(64) void handoff_pingpong_kernel(unsigned* flags, unsigned long long* out, int rounds) {
  if (threadIdx.x != 0) return;
  unsigned* mine = flags + (blockIdx.x == 0 ? 0 : 32);
  unsigned* theirs = flags + (blockIdx.x == 0 ? 32 : 0);
  unsigned* dead = flags + 16;   // either side gave up: the other leaves at once instead of spinning through every remaining round
  const unsigned long long t0 = wall_clock64();
  bool ok = true;
  for (int i = 1; i <= rounds && ok; ++i) {
    if (blockIdx.x == 0) __hip_atomic_store(mine, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok = false;
    for (int spins = 0; spins < (1 << 20); ++spins) {
      if (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)i) { ok = true; break; }
      if ((spins & 255) == 255 && __hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    }
    if (!ok) { __hip_atomic_store(dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // (the two workgroups were not co-resident)
    if (blockIdx.x != 0) __hip_atomic_store(mine, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (blockIdx.x == 0) out[0] = ok ? wall_clock64() - t0 : 0ull;   // 0: no measurement (the host falls back to the default delays)
}

// (unsigned* err: this kernel may RAISE the error word too)
__global__ __launch_bounds__(64) void wait_for_word_kernel(const unsigned* word, unsigned target, unsigned* err, unsigned long long limit_ticks) {
  if (threadIdx.x != 0) return;
  // ~7 us per poll, bounded on the WALL CLOCK (limit_ticks of 10 ns; the host scales it with the recurrence it waits on and with the
  // spin limit, which a communicator raises tenfold: wait_for_word): orders of magnitude above any recurrence this waits on, and the
  // producer's stream raises the word itself behind that kernel.  A wait that does give up must not let the consumer behind it pass
  // for a success: it raises the error word (value 2), the step is dropped like one whose recurrence kernel gave up (in a
  // data-parallel run: contributes a zero gradient), and the host goes on WITHOUT the early GEMM (the persistent kernels stay).
  // The one known way to get there: a tool that lets only ONE kernel run at a time (rocprofv3 --pmc) and picks this one before the
  // recurrence -- collect counters with EESEN_FWD_MID=0 (scripts/collect_profiles.sh does).  A command-processor wait
  // (hipStreamWaitValue64) instead costs 2.2 ms per cfg2 step and DEAD-LOCKS under that tool, without a bound (net.cpp).
  const unsigned long long t0 = wall_clock64();
  for (unsigned spins = 0;; ++spins) {
    if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return;
    if ((spins & 7) == 7) {
      if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
      if (wall_clock64() - t0 > limit_ticks) break;
    }
    // ~7 us asleep per poll (round 6; round 5: ~1 us): the waiter is off the critical path -- what it releases is a GEMM that runs
    // UNDER the rest of the recurrence -- and it sits on a CU whose issue slots the recurrence wants for 2 ms, three times a step
    __builtin_amdgcn_s_sleep(127);
    __builtin_amdgcn_s_sleep(127);
  }
  __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // 2: "the milestone wait gave up" (net.cpp: check_device_error)
}

// LstmLayerDev::milestone: the first workgroup of every (direction, sequence tile) group reports once its group has published
// step milestone_step; the last of them raises the flag word the host's side stream waits for
__device__ __forceinline__ void report_milestone(unsigned* ms, unsigned ngroups) {
  if (__hip_atomic_fetch_add(ms, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == ngroups)
    __hip_atomic_store(ms + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Debug timeline (EESEN_TRACE=1): workgroup (0,0,0), thread 0 stamps the shader clock at 5 points of the first 128 steps.
#define EESEN_STAMP(i) do { if (trace && tid == 0 && blockIdx.x == 0 && step < 127) \
    trace[step * 5 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

// ------------------------------------------------------------------------------------------------
// forward: workgroup = (4*NT hidden units) x (16*MT sequences) x 1 direction, 512 threads, MT*NT = 2.
//   <MT=2, NT=1>: grid (H/4, ndir, ceil(S/32)) -- the decomposition of lstm_fwd_step_kernel
//   <MT=1, NT=2>: grid (H/8, ndir, ceil(S/16)) -- same workgroup count at S = 32, same MFMA count, but each workgroup
//                 fetches m_{t-1} of 16 sequences instead of 32: the per-CU fetch (~34 GB/s per CU for data that just
//                 crossed the fabric) is what bounds the step, so halving it is worth the second 16-row slice of W_m
//                 in registers.
// ------------------------------------------------------------------------------------------------
template <int CPW, int MT, int NT, bool DROP, bool XCHG = false>
__global__ __launch_bounds__(NW * 64) void lstm_fwd_persistent_kernel(LstmLayerDev L, unsigned* cnt, unsigned* err,
                                                                      int spin_limit, unsigned long long* trace, Role R) {
  constexpr int ST = 16 * MT, UB = 4 * NT, RW = 16 * NT + 4;  // sequences, units per workgroup; padded LDS row
  __shared__ __attribute__((aligned(16))) float red[NW][ST][RW];
  __shared__ int s_go;
  __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win issue arbitration against co-resident GEMM waves
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = L.H, S = L.S, T = L.T;
  const int ldY = L.ndir * H, ldG = L.ndir * 4 * H;
  const int bx = R.unit_group(blockIdx.x), dir = R.dir(blockIdx.x), bz = R.seq_group(blockIdx.x);
  const int u0 = bx * UB, s0 = L.s_begin + bz * ST;
  const int s_end = L.s_begin + (L.s_count ? L.s_count : S);   // sequence window of this launch
  unsigned* my_cnt = cnt + (size_t)(dir * R.nz + bz) * kShards * kShardStride;
  const unsigned nblk = R.nblk;

  const int li = lane & 15, kq = lane >> 4;
  // this wave's part of the workgroup's 16*NT gate rows of W_m: resident in registers for the whole layer pass
  float b[NT][CPW][8];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const float* Wr = L.Wm + ((size_t)dir * 4 * H + (size_t)u0 * 4 + n * 16 + li) * H;
#pragma unroll
    for (int c = 0; c < CPW; ++c) ld8_plain(Wr, (wave + c * NW) * 32 + kq * 8, H, true, b[n][c]);
  }
  const int es = tid / UB, eu = tid % UB;
  const int s_e = s0 + es;
  const bool e_ok = tid < ST * UB && s_e < s_end;
  float p_i = 0.f, p_f = 0.f, p_o = 0.f, cprev = 0.f;  // c_{t-1} of this thread's (sequence, unit) never leaves the register
  int len = 0;
  if (e_ok) {
    const float* pp = L.peep + (size_t)dir * 3 * H + u0 + eu;
    p_i = pp[0]; p_f = pp[H]; p_o = pp[2 * H];
    len = L.lens[s_e];
  }
  const size_t gcol = (size_t)dir * 4 * H + (u0 + eu) * 4;
  float4 gx = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e_ok) gx = *reinterpret_cast<const float4*>(L.G + (size_t)((dir == 0 ? 0 : T - 1) * S + s_e) * ldG + gcol);

  const __amdgpu_buffer_rsrc_t rY = make_rsrc(XCHG ? L.X : L.Y);  // loop-invariant (re-basing it every step costs ~0.25 us per step)
  // exchange layout (XCHG, MT == 1): block of (t, dir, sequence tile) = nch chunks x [2 halves][4 quads][16 sequences][4 floats]
  const int nch = (H + 31) / 32;
  const unsigned xblk = (unsigned)nch * 512u * 4u;                                   // bytes per block
  const int zt = (L.s_begin / ST) + bz;                                              // tile index within the whole batch
  const int nzall = (S + ST - 1) / ST;
  for (int step = 0; step < T; ++step) {
    const int t = dir == 0 ? step : T - 1 - step;
    const int tp = dir == 0 ? t - 1 : t + 1;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    EESEN_STAMP(0);
    if (step > 0) {  // m_{tp} complete? (step 0 reads the zero boundary: nothing to wait for, nothing to multiply)
      if (wave == EESEN_POLL_WAVE) {
        const bool go = wait_counters(my_cnt, nblk, (unsigned)step, err, spin_limit, lane, L.poll_delay);
        if (lane == 0) s_go = go ? 1 : 0;
      }
      __syncthreads();
      if (!s_go) return;
      EESEN_STAMP(1);
      if (L.milestone && step == L.milestone_step + 1 && bx == 0 && tid == 0)   // the whole group has published step milestone_step
        report_milestone(L.milestone, (unsigned)(R.ndir * R.nz));
      const unsigned ybase = (unsigned)(((size_t)(tp + 1) * S * ldY + dir * H) * 4);  // < 2 GB, checked on the host
      float a[MT][CPW][8];
      if constexpr (XCHG) {
        // one request per line: lanes (kq, li) of one load are 16 adjacent 16-byte pieces = 256 contiguous bytes per quad
        const unsigned xb = ((unsigned)(tp * L.ndir + dir) * (unsigned)nzall + (unsigned)zt) * xblk;
        constexpr unsigned kOob = 0x80000000u;
        const bool rok = s0 + li < s_end;
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
          const int ch = wave + c * NW;
          const unsigned o0 = xb + (unsigned)((ch * 2 * 4 + kq) * 64 + li * 4) * 4u;
          const bool ok = rok && ch < nch;
          const f32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(rY, ok ? o0 : kOob, 0, 0);
          const f32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(rY, ok ? o0 + 4u * 64u * 4u : kOob, 0, 0);
          a[0][c][0] = lo[0]; a[0][c][1] = lo[1]; a[0][c][2] = lo[2]; a[0][c][3] = lo[3];
          a[0][c][4] = hi[0]; a[0][c][5] = hi[1]; a[0][c][6] = hi[2]; a[0][c][7] = hi[3];
        }
      } else {
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        const int k = (wave + c * NW) * 32 + kq * 8;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int sa = s0 + m * 16 + li;
          ld8_sc1(rY, ybase + (unsigned)(((size_t)sa * ldY + k) * 4), k, H, sa < s_end, a[m][c]);
        }
      }
      }
      __builtin_amdgcn_sched_barrier(0);  // all loads in flight BEFORE the first MFMA (else they are issued lazily, 2 at a time)
#pragma unroll
      for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][c][j], b[n][c][j], acc[m][n], 0, 0, 0);
    }
    // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][m * 16 + 4 * kq + r][n * 16 + li] = acc[m][n][r];
    __syncthreads();
    EESEN_STAMP(2);
    if (e_ok) {
      float4 pre = gx;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float4 v = *reinterpret_cast<const float4*>(&red[w][es][eu * 4]);
        pre.x += v.x; pre.y += v.y; pre.z += v.z; pre.w += v.w;
      }
      float g = tanhf_(pre.x);
      float i = sigmoidf_(pre.y + p_i * cprev);
      float f = sigmoidf_(pre.z + p_f * cprev);
      float c = g * i + cprev * f;
      if (DROP) {  // recurrent dropout (a separate instantiation: the plain kernel carries no trace of it): see lstm_fwd_step_kernel
        const float mk = L.rmask[(size_t)((t + 1) * S + s_e) * ldY + dir * H + u0 + eu];
        c = L.drop_mode == 1 ? mk * (g * i) + cprev * f : mk * (g * i + cprev * f);
      }
      float h = tanhf_(c);
      float o = sigmoidf_(pre.w + p_o * c);
      float m = h * o;
      if (t >= len) { g = i = f = o = c = m = 0.f; }
      *reinterpret_cast<float4*>(L.G + (size_t)(t * S + s_e) * ldG + gcol) = make_float4(g, i, f, o);
      const size_t o1 = (size_t)((t + 1) * S + s_e) * ldY + dir * H + u0 + eu;
      L.C[o1] = c;
      __hip_atomic_store(L.Y + o1, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through: other XCDs read it next step
      if constexpr (XCHG) {  // the exchange copy, in the consumers' fetch order (see LstmLayerDev::X)
        const int k = u0 + eu;
        const size_t xo = ((size_t)(t * L.ndir + dir) * nzall + zt) * ((size_t)nch * 512) +
                          (size_t)(((k >> 5) * 2 + ((k & 7) >> 2)) * 4 + ((k & 31) >> 3)) * 64 + es * 4 + (k & 3);
        __hip_atomic_store(L.X + xo, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      cprev = c;
    }
    EESEN_STAMP(3);
    {  // published after EVERY step: the last one is what a gated GEMM of the next layer waits for
      if (tid < ST * UB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores (ST * UB is a multiple of 64)
      __syncthreads();                                                 // (also fences `red` for the next step)
      EESEN_STAMP(4);
      if (tid == 0) __hip_atomic_fetch_add(my_cnt + (bx & (kShards - 1)) * kShardStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e_ok && step + 1 < T)  // next step's gate pre-activations: issued AFTER the publish so the drain never waits for HBM
        gx = *reinterpret_cast<const float4*>(L.G + (size_t)((dir == 0 ? t + 1 : t - 1) * S + s_e) * ldG + gcol);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward on the bf16 matrix pipe: lstm_fwd_persistent_bf_kernel<CPW, NT, AP, WP>.  The recurrent product m_{t-1} W_m^T on
// v_mfma_f32_16x16x32_bf16 (fp32 accumulation; ~17 cycles per 16 x 16 x 32 block where the fp32-input form takes 8 x 32), with
// both operands held as bf16 PLANES: a value x is rounded to nearest-even bf16, the remainder (exact in fp32) is rounded again,
// and so on -- P planes carry 8 + 9 (P - 1) significant bits, three of them ALL 24 bits of an fp32 value.  W_m's WP planes are
// resident in registers for the whole layer pass; m_t is split into AP planes ONCE, by the thread that computes it, and travels
// as bf16 in the exchange buffer (LstmLayerDev::X: per (t, dir, 16-sequence tile) block AP planes of
// [32-unit chunk][k quad][16 sequences][8 bf16]; lane (sequence li, quad kq) of chunk ch reads 16 bytes at
// ((ch * 4 + kq) * 16 + li) * 16 of each plane -- 16 lanes read 256 contiguous bytes).  Of the AP x WP cross products those with
// plane indices i + j <= max(AP, WP) - 1 are formed, smallest first.  A and B fragments use the same lane -> k assignment, so the
// contraction is exact whatever the instruction's internal k order.  Two uses:
//   <AP = 3, WP = 3>  fp32-class arithmetic (the 3-way split of gemm.hip inside the recurrence): six products per fp32 product,
//        dropped terms <= 2^-23 |ab|, i.e. one fp32 rounding.  24 instructions of 17 cycles per wave and step at H = 512 instead
//        of 32 of 32 cycles: the fp32 MFMA chain was 0.9 of the 3.1 us step.  Round 2 costed this with the A operand split by
//        every CONSUMER (90-230 VALU instructions per step: not worth it) or stored pre-split at 1.5x the fetch (then thought
//        to be the bound; a 4 x 32 forward tile in round 4 -- a QUARTER of the fetch, no faster, since removed -- showed it is not).  The
//        narrow tile (16 sequences x 8 units) only: three planes of the wide tile's W_m do not fit the register file.
//        Not bit-identical to lstm_fwd_step_kernel any more (EESEN_FWD_SPLIT=0 restores the fp32-input kernel, which is).
//   <AP = 1, WP = 2>  BASELINE config 4's "bf16 forward" (eesen_net_set_forward_precision(1)): m_t rounded to ONE bf16 plane --
//        half the bytes every consumer fetches per step --, W_m as hi + lo (17 bits).  (With W_m as ONE plane as well -- an arm
//        of round 4, removed -- the gradients sat 2.7x further from the reference, 0.18 against 0.069 max-norm at full cfg4
//        size: a weight rounded to 8 bits is a systematic perturbation of the model that every one of the T steps sees, the
//        rounding of m_t is noise.)  Gate pre-activations, cell state, activations and everything stored for the backward
//        pass (G, C, Y) stay fp32; the backward pass is the fp32 one.
//   <AP = 2, WP = 2, F16>  (round 6) fp32-class arithmetic on TWO fp16 planes per operand and three products: with round-to-nearest
//        at both levels an fp32 value is hi + lo to within 2^-22 (gemm.hip, mode 2, has the argument and the measurements), the dropped
//        lo x lo' is <= 2^-22 |ab|.  fp16 has 5 exponent bits, so both operands carry an exact power of two: m_t times 2^14 (|m| = |o tanh c| < 1,
//        so its planes sit in fp16's top binades; what falls into the denormal range is multiplied exactly by the MFMA), W_m times
//        the power that brings max |W_m| of the layer (LstmLayerDev::wm_amax, measured by the host after every parameter change)
//        into [2^14, 2^15); the accumulators are multiplied by the inverse on their way to the cell.  12 MFMAs per wave and step at
//        H = 512 instead of 24, two planes to fetch instead of three -- and, unlike three bf16 planes, two fp16 planes of the WIDE
//        tile's 64 gate rows fit the register file (128 registers at H = 1024, what the fp32 rows take): wide layers leave the
//        fp32-input MFMA chain (3.6 of their 5.7-6.0 us step) as well.
// Tile: 16 sequences x 4 NT units; CPW 32-unit chunks of K = H per wave.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// (gemm.hip: half_scale) the power of two that brings an operand bounded by *amax into [2^14, 2^15), and its inverse
__device__ __forceinline__ void half_scale(const float* amax, float& scale, float& inv) {
  const int e = (int)((__float_as_uint(*amax) >> 23) & 0xffu);
  const int s = min(max(127 + 14 + 127 - e, 1), 253);
  scale = __uint_as_float((unsigned)s << 23);
  inv = __uint_as_float((unsigned)(254 - s) << 23);
}
__device__ __forceinline__ void half_scale(float amax, float& scale, float& inv) { half_scale(&amax, scale, inv); }
__device__ __forceinline__ unsigned rne_f16(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)x); }   // v_cvt_f16_f32: round to nearest even
__device__ __forceinline__ float f16_bits_to_f32(unsigned h) { return (float)__builtin_bit_cast(_Float16, (unsigned short)h); }
__device__ __forceinline__ unsigned rne_bf16(float x) {   // round-to-nearest-even bf16 (gemm.hip: rne_bf16_bits), in the low 16 bits
  const unsigned b = __float_as_uint(x);
  return (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;
}
// lane i of every quad receives lane i ^ 1 / i ^ 2 (DPP quad_perm: a VALU move, where __shfl_xor goes through the LDS crossbar)
__device__ __forceinline__ unsigned quad_xor1(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned quad_xor2(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true); }
template <int CPW, int NT, int AP, int WP, bool F16 = false>
__global__ __launch_bounds__(NW * 64) void lstm_fwd_persistent_bf_kernel(LstmLayerDev L, unsigned* cnt, unsigned* err,
                                                                         int spin_limit, unsigned long long* trace, Role R) {
  constexpr int ST = 16, UB = 4 * NT, RW = 16 * NT + 4, SMAX = (AP > WP ? AP : WP) - 1;
  constexpr float kMScale = 16384.f;   // F16: m_t travels times 2^14
  __shared__ __attribute__((aligned(16))) float red[NW][ST][RW];
  __shared__ int s_go;
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = L.H, S = L.S, T = L.T;
  if (T < 0) {   // residency census (bf_census): every workgroup checks in and waits -- at most 5 ms of wall clock -- until the whole grid has checked in
    if (tid == 0) {
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long t0 = wall_clock64();
      bool all = false;
      while (!(all = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) && wall_clock64() - t0 < 500000ull)   // 5 ms
        __builtin_amdgcn_s_sleep(8);
      if (!all) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  const int ldY = L.ndir * H, ldG = L.ndir * 4 * H;
  const int bx = R.unit_group(blockIdx.x), dir = R.dir(blockIdx.x), bz = R.seq_group(blockIdx.x);
  const int u0 = bx * UB, s0 = L.s_begin + bz * ST;
  const int s_end = L.s_begin + (L.s_count ? L.s_count : S);
  unsigned* my_cnt = cnt + (size_t)(dir * R.nz + bz) * kShards * kShardStride;
  const unsigned nblk = R.nblk;
  const int li = lane & 15, kq = lane >> 4;
  const int nch = H / 32;
  // this wave's part of the workgroup's 16 NT gate rows of W_m as WP 16-bit planes, resident for the whole layer pass
  float wscale = 1.f, unscale = 1.f;
  if constexpr (F16) { half_scale(L.wm_amax, wscale, unscale); unscale *= 1.f / kMScale; }
  f32x4 b[WP][NT][CPW];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const float* Wr = L.Wm + ((size_t)dir * 4 * H + (size_t)u0 * 4 + n * 16 + li) * H;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      float w[8];
      ld8_plain(Wr, (wave + c * NW) * 32 + kq * 8, H, true, w);
      if constexpr (F16) {
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] *= wscale;       // exact (a power of two)
      }
#pragma unroll
      for (int pl = 0; pl < WP; ++pl) {
        f32x4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (F16) {
            const unsigned h0 = rne_f16(w[2 * j]), h1 = rne_f16(w[2 * j + 1]);
            pk[j] = __uint_as_float(h0 | (h1 << 16));
            w[2 * j] -= f16_bits_to_f32(h0);              // exact: the next plane rounds the remainder
            w[2 * j + 1] -= f16_bits_to_f32(h1);
          } else {
            const unsigned h0 = rne_bf16(w[2 * j]), h1 = rne_bf16(w[2 * j + 1]);
            pk[j] = __uint_as_float(h0 | (h1 << 16));
            w[2 * j] -= __uint_as_float(h0 << 16);          // exact: the next plane rounds the remainder
            w[2 * j + 1] -= __uint_as_float(h1 << 16);
          }
        }
        b[pl][n][c] = pk;
      }
    }
  }
  const int es = tid / UB, eu = tid % UB;
  const int s_e = s0 + es;
  const bool e_act = tid < ST * UB;            // the cell waves (all of their lanes take part in the packing shuffles)
  const bool e_ok = e_act && s_e < s_end;
  float p_i = 0.f, p_f = 0.f, p_o = 0.f, cprev = 0.f;
  int len = 0;
  if (e_ok) {
    const float* pp = L.peep + (size_t)dir * 3 * H + u0 + eu;
    p_i = pp[0]; p_f = pp[H]; p_o = pp[2 * H];
    len = L.lens[s_e];
  }
  const size_t gcol = (size_t)dir * 4 * H + (u0 + eu) * 4;
  float4 gx = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e_ok) gx = *reinterpret_cast<const float4*>(L.G + (size_t)((dir == 0 ? 0 : T - 1) * S + s_e) * ldG + gcol);
  const __amdgpu_buffer_rsrc_t rX = make_rsrc(L.X);
  const unsigned xpl = (unsigned)nch * 1024u;                                        // bytes per plane of a block
  const unsigned xblk = xpl * AP;                                                    // bytes per block
  const int zt = (L.s_begin / ST) + bz;
  const int nzall = (S + ST - 1) / ST;
  unsigned char* const xbytes = reinterpret_cast<unsigned char*>(L.X);
  for (int step = 0; step < T; ++step) {
    const int t = dir == 0 ? step : T - 1 - step;
    const int tp = dir == 0 ? t - 1 : t + 1;
    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    EESEN_STAMP(0);
    if (step > 0) {
      if (wave == EESEN_POLL_WAVE) {
        const bool go = wait_counters(my_cnt, nblk, (unsigned)step, err, spin_limit, lane, L.poll_delay);
        if (lane == 0) s_go = go ? 1 : 0;
      }
      __syncthreads();
      if (!s_go) return;
      EESEN_STAMP(1);
      if (L.milestone && step == L.milestone_step + 1 && bx == 0 && tid == 0)
        report_milestone(L.milestone, (unsigned)(R.ndir * R.nz));
      const unsigned xb = ((unsigned)(tp * L.ndir + dir) * (unsigned)nzall + (unsigned)zt) * xblk;
      constexpr unsigned kOob = 0x80000000u;
      const bool rok = s0 + li < s_end;
      f32x4 a[AP][CPW];
#pragma unroll
      for (int pl = 0; pl < AP; ++pl)
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
          const int ch = wave + c * NW;
          a[pl][c] = __builtin_amdgcn_raw_buffer_load_b128(rX, (rok && ch < nch) ? xb + (unsigned)pl * xpl + (unsigned)(((ch * 4 + kq) * 16 + li) * 16) : kOob, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int sum = SMAX; sum >= 0; --sum)      // the smallest cross products first
#pragma unroll
        for (int i = 0; i < AP; ++i) {
          const int j = sum - i;
          if (j < 0 || j >= WP) continue;
#pragma unroll
          for (int c = 0; c < CPW; ++c)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              const f32x4& bb = b[j < WP && j >= 0 ? j : 0][n][c];
              if constexpr (F16) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a[i][c]), __builtin_bit_cast(f16x8_t, bb), acc[n], 0, 0, 0);
              else acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[i][c]), __builtin_bit_cast(bf16x8_t, bb), acc[n], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][4 * kq + r][n * 16 + li] = F16 ? acc[n][r] * unscale : acc[n][r];
    __syncthreads();
    EESEN_STAMP(2);
    if (e_act) {
      // the eight waves' partial sums: ALL eight LDS reads in flight, then the sum in the same order (hipcc kept two in flight and
      // waited for each in turn: eight ~100-cycle round trips on the step's critical path)
      float4 pv[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) pv[w] = *reinterpret_cast<const float4*>(&red[w][es][eu * 4]);
      __builtin_amdgcn_sched_barrier(0);
      float4 pre = gx;
#pragma unroll
      for (int w = 0; w < NW; ++w) { pre.x += pv[w].x; pre.y += pv[w].y; pre.z += pv[w].z; pre.w += pv[w].w; }
      float g = tanhf_(pre.x);
      float i = sigmoidf_(pre.y + p_i * cprev);
      float f = sigmoidf_(pre.z + p_f * cprev);
      float c = g * i + cprev * f;
      float h = tanhf_(c);
      float o = sigmoidf_(pre.w + p_o * c);
      float m = h * o;
      if (t >= len || !e_ok) { g = i = f = o = c = m = 0.f; }
      if (e_ok) {
        *reinterpret_cast<float4*>(L.G + (size_t)(t * S + s_e) * ldG + gcol) = make_float4(g, i, f, o);
        const size_t o1 = (size_t)((t + 1) * S + s_e) * ldY + dir * H + u0 + eu;
        L.C[o1] = c;
        __hip_atomic_store(L.Y + o1, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cprev = c;
      }
      // the AP planes of m: four adjacent units -> one 8-byte word per plane (a single 8-byte store is single-copy atomic; write-through)
      const int k = u0 + eu;
      const size_t xo = ((size_t)(t * L.ndir + dir) * nzall + zt) * (size_t)xblk +
                        (size_t)((((k >> 5) * 4 + ((k & 31) >> 3)) * 16 + es) * 16 + (k & 7) * 2);
      float rem = F16 ? m * kMScale : m;
#pragma unroll
      for (int pl = 0; pl < AP; ++pl) {
        const unsigned hb = F16 ? rne_f16(rem) : rne_bf16(rem);
        rem -= F16 ? f16_bits_to_f32(hb) : __uint_as_float(hb << 16);          // exact
        const unsigned pk = hb | (quad_xor1(hb) << 16);                        // valid in even lanes: (unit eu, eu + 1)
        const unsigned pk2 = quad_xor2(pk);                                     // lanes eu % 4 == 0: the pair of (eu + 2, eu + 3)
        if (e_ok && (eu & 3) == 0)
          __hip_atomic_store(reinterpret_cast<unsigned long long*>(xbytes + xo + (size_t)pl * xpl), (unsigned long long)pk | ((unsigned long long)pk2 << 32),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    EESEN_STAMP(3);
    {
      if (e_act) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      EESEN_STAMP(4);
      if (tid == 0) __hip_atomic_fetch_add(my_cnt + (bx & (kShards - 1)) * kShardStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e_ok && step + 1 < T)
        gx = *reinterpret_cast<const float4*>(L.G + (size_t)((dir == 0 ? t + 1 : t - 1) * S + s_e) * ldG + gcol);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward, TWO sequence tiles per workgroup (time-multiplexed): for batches whose tiles need more workgroups than can be
// co-resident (S = 64 at H = 1024: BASELINE config 5).  Instead of one cooperative launch per window of 32 sequences, run
// one after the other, every workgroup holds its 16 * NT gate rows of W_m once and steps TWO independent chains -- sequence
// tile 2g and 2g + 1 -- alternately: A_t, B_t, A_{t+1}, ...  While chain A's step is being published and acknowledged by the
// peers (the 1.1 us no workgroup can shorten), this workgroup computes chain B's step, so by the time it polls for A again
// the peers are through: the hand-off latency of one chain hides behind the arithmetic of the other.  Same arithmetic, same
// order per chain as lstm_fwd_persistent_kernel<CPW, 1, NT>: bit-identical results.
// ------------------------------------------------------------------------------------------------
template <int CPW, int NT, bool XCHG>
__global__ __launch_bounds__(NW * 64) void lstm_fwd_persistent_mux_kernel(LstmLayerDev L, unsigned* cnt, unsigned* err,
                                                                          int spin_limit, Role R) {
  constexpr int ST = 16, UB = 4 * NT, RW = 16 * NT + 4, Q = 2;
  __shared__ __attribute__((aligned(16))) float red[NW][ST][RW];
  __shared__ int s_go;
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = L.H, S = L.S, T = L.T;
  const int ldY = L.ndir * H, ldG = L.ndir * 4 * H;
  const int bx = R.unit_group(blockIdx.x), dir = R.dir(blockIdx.x), bg = R.seq_group(blockIdx.x);
  const int u0 = bx * UB;
  const int nzall = (S + ST - 1) / ST;
  const unsigned nblk = R.nblk;
  const int li = lane & 15, kq = lane >> 4;
  float b[NT][CPW][8];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const float* Wr = L.Wm + ((size_t)dir * 4 * H + (size_t)u0 * 4 + n * 16 + li) * H;
#pragma unroll
    for (int c = 0; c < CPW; ++c) ld8_plain(Wr, (wave + c * NW) * 32 + kq * 8, H, true, b[n][c]);
  }
  const int es = tid / UB, eu = tid % UB;
  float p_i = 0.f, p_f = 0.f, p_o = 0.f;
  if (tid < ST * UB) {
    const float* pp = L.peep + (size_t)dir * 3 * H + u0 + eu;
    p_i = pp[0]; p_f = pp[H]; p_o = pp[2 * H];
  }
  const size_t gcol = (size_t)dir * 4 * H + (u0 + eu) * 4;
  // per chain
  int zt[Q], s_e[Q], len[Q];
  bool e_ok[Q];
  float cprev[Q];
  float4 gx[Q];
  unsigned* my_cnt[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    zt[q] = bg * Q + q;
    s_e[q] = zt[q] * ST + es;
    e_ok[q] = tid < ST * UB && zt[q] < nzall && s_e[q] < S;
    len[q] = e_ok[q] ? L.lens[s_e[q]] : 0;
    cprev[q] = 0.f;
    gx[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e_ok[q]) gx[q] = *reinterpret_cast<const float4*>(L.G + (size_t)((dir == 0 ? 0 : T - 1) * S + s_e[q]) * ldG + gcol);
    my_cnt[q] = cnt + (size_t)(dir * nzall + zt[q]) * kShards * kShardStride;
  }
  const __amdgpu_buffer_rsrc_t rY = make_rsrc(XCHG ? L.X : L.Y);
  const int nch = (H + 31) / 32;
  const unsigned xblk = (unsigned)nch * 512u * 4u;
  bool polled = false;   // uniform: the slot about to start has already been found ready
  for (int step = 0; step < T; ++step) {
    const int t = dir == 0 ? step : T - 1 - step;
    const int tp = dir == 0 ? t - 1 : t + 1;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if (zt[q] >= nzall) continue;   // odd number of tiles: the last workgroup group steps one chain only (uniform per workgroup)
      f32x4 acc[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (step > 0) {
        // readiness of this slot was established by the poller wave during the PREVIOUS slot's cell phase (below); only the very
        // first polled slot (step 1, chain 0 when chain 1 does not exist) has to poll here
        if (!polled) {
          if (wave == EESEN_POLL_WAVE) {
            const bool go = wait_counters(my_cnt[q], nblk, (unsigned)step, err, spin_limit, lane, 0);
            if (lane == 0) s_go = go ? 1 : 0;
          }
          __syncthreads();
          if (!s_go) return;
        }
        if (L.milestone && step == L.milestone_step + 1 && bx == 0 && tid == 0)   // this chain's group has published step milestone_step
          report_milestone(L.milestone, (unsigned)(L.ndir * nzall));
        float a[CPW][8];
        const int s0 = zt[q] * ST;
        if constexpr (XCHG) {
          const unsigned xb = ((unsigned)(tp * L.ndir + dir) * (unsigned)nzall + (unsigned)zt[q]) * xblk;
          constexpr unsigned kOob = 0x80000000u;
          const bool rok = s0 + li < S;
#pragma unroll
          for (int c = 0; c < CPW; ++c) {
            const int ch = wave + c * NW;
            const unsigned o0 = xb + (unsigned)((ch * 2 * 4 + kq) * 64 + li * 4) * 4u;
            const bool ok = rok && ch < nch;
            const f32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(rY, ok ? o0 : kOob, 0, 0);
            const f32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(rY, ok ? o0 + 4u * 64u * 4u : kOob, 0, 0);
            a[c][0] = lo[0]; a[c][1] = lo[1]; a[c][2] = lo[2]; a[c][3] = lo[3];
            a[c][4] = hi[0]; a[c][5] = hi[1]; a[c][6] = hi[2]; a[c][7] = hi[3];
          }
        } else {
          const unsigned ybase = (unsigned)(((size_t)(tp + 1) * S * ldY + dir * H) * 4);
#pragma unroll
          for (int c = 0; c < CPW; ++c) {
            const int k = (wave + c * NW) * 32 + kq * 8;
            const int sa = s0 + li;
            ld8_sc1(rY, ybase + (unsigned)(((size_t)sa * ldY + k) * 4), k, H, sa < S, a[c]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < CPW; ++c)
#pragma unroll
          for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], b[n][c][j], acc[n], 0, 0, 0);
      }
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][4 * kq + r][n * 16 + li] = acc[n][r];
      __syncthreads();
      if (e_ok[q]) {
        float4 pre = gx[q];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          const float4 v = *reinterpret_cast<const float4*>(&red[w][es][eu * 4]);
          pre.x += v.x; pre.y += v.y; pre.z += v.z; pre.w += v.w;
        }
        float g = tanhf_(pre.x);
        float i = sigmoidf_(pre.y + p_i * cprev[q]);
        float f = sigmoidf_(pre.z + p_f * cprev[q]);
        float c = g * i + cprev[q] * f;
        float h = tanhf_(c);
        float o = sigmoidf_(pre.w + p_o * c);
        float m = h * o;
        if (t >= len[q]) { g = i = f = o = c = m = 0.f; }
        *reinterpret_cast<float4*>(L.G + (size_t)(t * S + s_e[q]) * ldG + gcol) = make_float4(g, i, f, o);
        const size_t o1 = (size_t)((t + 1) * S + s_e[q]) * ldY + dir * H + u0 + eu;
        L.C[o1] = c;
        __hip_atomic_store(L.Y + o1, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (XCHG) {
          const int k = u0 + eu;
          const size_t xo = ((size_t)(t * L.ndir + dir) * nzall + zt[q]) * ((size_t)nch * 512) +
                            (size_t)(((k >> 5) * 2 + ((k & 7) >> 2)) * 4 + ((k & 31) >> 3)) * 64 + es * 4 + (k & 3);
          __hip_atomic_store(L.X + xo, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        cprev[q] = c;
      }
      // While waves 0-1 finish the cell and drain their stores, the poller wave (idle otherwise) waits for the NEXT slot's
      // inputs -- the other chain, whose publish by the peers lies one whole slot back: that slot then starts without a poll.
      polled = false;
      {
        const int nq = q + 1 < Q && zt[Q - 1] < nzall ? q + 1 : 0;       // next slot: the other chain (or this one again)
        const int nstep = nq > q ? step : step + 1;
        if (nstep > 0 && nstep < T && !(nq == q)) {                        // same chain next: its inputs depend on OUR publish below
          if (wave == EESEN_POLL_WAVE) {
            const bool go = wait_counters(my_cnt[nq], nblk, (unsigned)nstep, err, spin_limit, lane, 0);
            if (lane == 0) s_go = go ? 1 : 0;
          }
          polled = true;
        }
      }
      if (tid < ST * UB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (polled && !s_go) return;
      if (tid == 0) __hip_atomic_fetch_add(my_cnt[q] + (bx & (kShards - 1)) * kShardStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e_ok[q] && step + 1 < T)
        gx[q] = *reinterpret_cast<const float4*>(L.G + (size_t)((dir == 0 ? t + 1 : t - 1) * S + s_e[q]) * ldG + gcol);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward: grid (ceil(H/16), ndir, ceil(S/16)), 512 threads -- the decomposition of lstm_bwd_step_kernel
// ------------------------------------------------------------------------------------------------
template <int CPW, int ST, bool DROP>  // ST = sequences per workgroup (16, or 8: half-filled MFMA rows but half the DG_next fetch per CU)
__global__ __launch_bounds__(NW * 64) void lstm_bwd_persistent_kernel(LstmLayerDev L, const float* __restrict__ dY,
                                                                      int lddy, float* __restrict__ DG, unsigned* cnt,
                                                                      unsigned* err, int spin_limit, unsigned long long* trace,
                                                                      Role R, int chunk) {
  constexpr bool X44 = ST == 8 && CPW <= 8;
  // X44 at CPW = 8: the B operand is 128 registers per lane (W_m^T is replicated over the two sequence halves), and with them
  // the kernel would need ~200 VGPRs -- two of its waves and one wave of a side-stream GEMM (84 + 64 accumulator registers) no
  // longer fit a SIMD's 512.  The last LDSB chunks' B values live in LDS instead (16 KB per chunk and workgroup) and are read
  // back every step, 4 ds_read_b128 per chunk, under the operand fetch.
  constexpr int LDSB = X44 && CPW == 8 ? 2 : 0, REGB = X44 ? CPW - LDSB : 1;
  __shared__ __attribute__((aligned(16))) float4 bl[LDSB ? LDSB : 1][4][LDSB ? NW * 64 : 1];   // [chunk][ABID][thread]
  __shared__ float red[NW][16][17];   // X44: [wave][k class (2) x sequence (8)][unit]
  __shared__ int s_go;
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = L.H, S = L.S, T = L.T;
  const int ldY = L.ndir * H, ldG = L.ndir * 4 * H, K4 = 4 * H;
  const int bx = R.unit_group(blockIdx.x), dir = R.dir(blockIdx.x), bz = R.seq_group(blockIdx.x);
  const int u0 = bx * 16, s0 = L.s_begin + bz * ST;
  const int s_end = L.s_begin + (L.s_count ? L.s_count : S);   // sequence window of this launch
  unsigned* my_cnt = cnt + (size_t)(dir * R.nz + bz) * kShards * kShardStride;
  const unsigned nblk = R.nblk;

  const int li = lane & 15, kq = lane >> 4;
  const int sa = s0 + li, ub = u0 + li;
  // X44: the 64 lanes are 16 blocks of 4 -- group g = lane >> 4 = (sequence half rb, k class ks), unit quad cb = (lane >> 2) & 3,
  // x = lane & 3.  One v_mfma_f32_4x4x1_16B_f32 multiplies, per block, a 4-vector of A (4 sequences, ONE k) with a 4-vector of B
  // (ONE k, 4 units) into a 4 x 4 tile: 16 blocks = 2 sequence halves x 4 unit quads x 2 k classes = the 8 x 16 outputs of the
  // workgroup for TWO k values, no padding rows, at the same 64 flops per cycle as the 16x16x4 form -- half the MFMA time.
  // The A vector of a block depends on (rb, ks) only, not on cb: CBSZ = 2 makes the four blocks of a group share the A lanes of
  // block ABID, so ONE operand register serves FOUR instructions (ABID = 0..3) and nothing is replicated: lane (rb, ks, cb', x)
  // loads 16 bytes of sequence rb*4 + x at k = chunk + (ks*4 + cb')*4 -- a whole 128-byte line per sequence and chunk, each
  // requested once -- and instruction (chunk, r, cb') takes component r with ABID = cb': it covers k = chunk + ks*16 + cb'*4 + r
  // in class ks.  B (resident): lane (rb, ks, cb, x) holds W_m^T[unit cb*4 + x][that k], 16 consecutive floats per chunk.
  const int g4 = lane >> 4, rb4 = g4 >> 1, ks4 = g4 & 1, cb4 = (lane >> 2) & 3, x4 = lane & 3;
  float bw[REGB][4][4];   // [chunk][component r][ABID cb']
  if constexpr (X44) {
    const int ub4 = u0 + cb4 * 4 + x4;
    const float* Br = L.WmT + ((size_t)dir * H + min(ub4, H - 1)) * K4;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      const int k0 = (wave + c * NW) * 32 + ks4 * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ub4 < H && k0 + q * 4 < K4) v = *reinterpret_cast<const float4*>(Br + k0 + q * 4);
        if (c < REGB) { bw[c][0][q] = v.x; bw[c][1][q] = v.y; bw[c][2][q] = v.z; bw[c][3][q] = v.w; }
        else bl[c - REGB][q][tid] = v;
      }
    }
  }
  float b[X44 ? 1 : CPW][8];  // this wave's part of the workgroup's 16 rows of W_m^T, resident for the whole layer pass
  if constexpr (!X44) {
    const float* Br = L.WmT + ((size_t)dir * H + ub) * K4;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      const int k0 = (wave + c * NW) * 32;
      if constexpr (ST == 8) {  // k order of the full-line operand fetch below: j < 4 -> k0 + 4kq + j, j >= 4 -> k0 + 16 + 4kq + (j - 4)
        float lo[8], hi[8];
        ld8_plain(Br, k0 + kq * 4, K4, ub < H, lo);       // only lo[0..3] / hi[0..3] are used
        ld8_plain(Br, k0 + 16 + kq * 4, K4, ub < H, hi);
#pragma unroll
        for (int j = 0; j < 4; ++j) { b[c][j] = lo[j]; b[c][4 + j] = hi[j]; }
      } else {
        ld8_plain(Br, k0 + kq * 8, K4, ub < H, b[c]);
      }
    }
  }
  const int es = tid >> 4, eu = tid & 15;
  const int s_e = s0 + es, u_e = u0 + eu;
  const bool e_ok = tid < ST * 16 && s_e < s_end && u_e < H;
  float p_i = 0.f, p_f = 0.f, p_o = 0.f;
  int len = 0;
  if (e_ok) {
    const float* pp = L.peep + (size_t)dir * 3 * H + u_e;
    p_i = pp[0]; p_f = pp[H]; p_o = pp[2 * H];
    len = L.lens[s_e];
  }
  // carried in registers between steps: d_c * f of the previous step and its d_i, d_f (the peephole terms, :483-484)
  float dcf = 0.f, dn_i = 0.f, dn_f = 0.f;
  const size_t gcol = (size_t)dir * K4 + u_e * 4;
  const size_t ycol = (size_t)dir * H + u_e;
  // this step's epilogue operands are loaded one step ahead
  float4 gt = make_float4(0.f, 0.f, 0.f, 0.f);
  float dy = 0.f, c_t = 0.f, c_p = 0.f;
  {
    const int t0 = dir == 0 ? T - 1 : 0, tp0 = dir == 0 ? t0 - 1 : t0 + 1;
    if (e_ok) {
      gt = *reinterpret_cast<const float4*>(L.G + (size_t)(t0 * S + s_e) * ldG + gcol);
      dy = dY[(size_t)(t0 * S + s_e) * lddy + ycol];
      c_t = L.C[(size_t)((t0 + 1) * S + s_e) * ldY + ycol];
      c_p = L.C[(size_t)((tp0 + 1) * S + s_e) * ldY + ycol];
    }
  }
  // 32-bit buffer offsets reach 2 GB, DG can be larger (config 5: T*S*8H*4 = 6.3 GB).  The buffer resource is re-based once
  // per CHUNK of `chunk` steps on the first row block the chunk touches (re-basing every step was measured: +0.27 us per
  // step); a chunk spans chunk + 1 row blocks, which the host sizes to stay below 2 GB.  Shapes below 2 GB: one chunk.
  __amdgpu_buffer_rsrc_t rDG = make_rsrc(DG);
  int tbS = 0;   // first row (t * S) the current resource is based on

  for (int step = 0; step < T; ++step) {
    const int t = dir == 0 ? T - 1 - step : step;
    const int tn = dir == 0 ? t + 1 : t - 1;
    if (chunk < T && step % chunk == 0) {   // uniform across the workgroup
      const int tb = dir == 0 ? max(0, T - step - chunk) : max(0, step - 1);
      tbS = tb * S;
      rDG = make_rsrc(DG + (size_t)tbS * ldG);
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    EESEN_STAMP(0);
    if (step > 0) {
      if (wave == EESEN_POLL_WAVE) {
        const bool go = wait_counters(my_cnt, nblk, (unsigned)step, err, spin_limit, lane, L.poll_delay);
        if (lane == 0) s_go = go ? 1 : 0;
      }
      __syncthreads();
      if (!s_go) return;
      EESEN_STAMP(1);
      const int tnb = tn * S - tbS;
      const size_t arow = ((size_t)(tnb + sa) * ldG + (size_t)dir * K4) * 4;  // byte offset of this lane's DG_next row in block tn
      if constexpr (X44) {
        const unsigned arow4 = (unsigned)(((size_t)(tnb + s0 + rb4 * 4 + x4) * ldG + (size_t)dir * K4) * 4);
        const bool rok = s0 + rb4 * 4 + x4 < s_end;
        // FOUR chunks of operands in flight (16 registers): chunk c + 4 is requested into chunk c's registers as soon as its 16
        // MFMAs have been issued -- with all eight in flight the kernel needs 222 VGPRs, and two of its waves plus one wave of a
        // side-stream GEMM (88) no longer fit a SIMD's 512
        constexpr int INF = CPW < 4 ? CPW : 4;
        auto fetch = [&](int c) {
          const int k = (wave + c * NW) * 32 + (ks4 * 4 + cb4) * 4;
          return __builtin_amdgcn_raw_buffer_load_b128(rDG, (rok && k < K4) ? arow4 + (unsigned)k * 4u : 0x80000000u, 0,
                                                       0);
        };
        f32x4 a4[INF];
#pragma unroll
        for (int c = 0; c < INF; ++c) a4[c] = fetch(c);
        __builtin_amdgcn_sched_barrier(0);  // these loads are in flight BEFORE the first MFMA
        f32x4 ac[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // two accumulators: dependent MFMAs two issues apart
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
          const f32x4 cur = a4[c % INF];
#pragma unroll
          for (int h = 0; h < 2; ++h) {   // ABID 2h and 2h + 1
            float w0[4], w1[4];
            if (c < REGB) {
#pragma unroll
              for (int r = 0; r < 4; ++r) { w0[r] = bw[c < REGB ? c : 0][r][2 * h]; w1[r] = bw[c < REGB ? c : 0][r][2 * h + 1]; }
            } else {   // this chunk's B from LDS, two ABID vectors at a time
              int lt = tid;
              asm volatile("" : "+v"(lt));   // opaque per step: the reads are loop-invariant, and hoisted they would be registers again
              const float4 v0 = bl[c < REGB ? 0 : c - REGB][2 * h][lt], v1 = bl[c < REGB ? 0 : c - REGB][2 * h + 1][lt];
              w0[0] = v0.x; w0[1] = v0.y; w0[2] = v0.z; w0[3] = v0.w;
              w1[0] = v1.x; w1[1] = v1.y; w1[2] = v1.z; w1[3] = v1.w;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (h == 0) {
                ac[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(cur[r], w0[r], ac[0], 2, 0, 0);
                ac[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(cur[r], w1[r], ac[1], 2, 1, 0);
              } else {
                ac[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(cur[r], w0[r], ac[0], 2, 2, 0);
                ac[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(cur[r], w1[r], ac[1], 2, 3, 0);
              }
            }
          }
          if (c + INF < CPW) {
            __builtin_amdgcn_sched_barrier(0);
            a4[c % INF] = fetch(c + INF);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // D of block (rb, ks, cb): vgpr i, lane x -> out[sequence rb*4 + i][unit cb*4 + x], partial sum of k class ks
#pragma unroll
        for (int i = 0; i < 4; ++i) acc0[i] = ac[0][i] + ac[1][i];
      } else if constexpr (ST == 8) {
        // Full-line fetch.  Only MFMA rows 0-7 carry sequences, so the lanes of rows 8-15 would idle.  Instead all 64 lanes
        // load: lane (li, kq) reads 16 bytes of sequence li & 7 at segment (li >> 3) * 4 + kq of the 128-byte chunk -- one
        // request per line and ONE load instruction per chunk instead of two half-empty ones that each touch every line (the
        // second request of a line occupies the L1 miss queue like the first; measured -740 ticks per step).  The upper
        // half of the chunk reaches rows 0-7 through a rotate-by-8 DPP move inside each 16-lane row; rows 8-15 of the
        // product are garbage that nobody reads.
        const unsigned arow8 = (unsigned)(((size_t)(tnb + s0 + (li & 7)) * ldG + (size_t)dir * K4) * 4);
        const bool rok = s0 + (li & 7) < s_end;
        const int seg = (li >> 3) * 4 + kq;
        f32x4 a4[CPW];
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
          const int k = (wave + c * NW) * 32 + seg * 4;
          a4[c] = __builtin_amdgcn_raw_buffer_load_b128(rDG, (rok && k < K4) ? arow8 + (unsigned)k * 4u : 0x80000000u, 0,
                                                        0);
        }
        __builtin_amdgcn_sched_barrier(0);  // all loads in flight BEFORE the first MFMA
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
          // row_ror:8 = lane i of a 16-lane row receives lane (i + 8) % 16
          const float hi[4] = {ror8(a4[c][0]), ror8(a4[c][1]), ror8(a4[c][2]), ror8(a4[c][3])};
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[c][0], b[c][0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[c][1], b[c][1], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[c][2], b[c][2], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[c][3], b[c][3], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(hi[0], b[c][4], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(hi[1], b[c][5], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(hi[2], b[c][6], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(hi[3], b[c][7], acc1, 0, 0, 0);
        }
      } else {
        // at most 8 chunks (64 registers) of operands in flight: K = 4H = 4096 (H = 1024) takes two rounds
        constexpr int CH = CPW > 8 ? 8 : CPW;
#pragma unroll
        for (int h = 0; h < CPW; h += CH) {
          float a[CH][8];
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const int k = (wave + (h + c) * NW) * 32 + kq * 8;
            ld8_sc1(rDG, (unsigned)(arow + (size_t)k * 4), k, K4, li < ST && sa < s_end, a[c]);
          }
          __builtin_amdgcn_sched_barrier(0);  // all loads of the round in flight BEFORE its first MFMA
#pragma unroll
          for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
              acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], b[h + c][j], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j + 1], b[h + c][j + 1], acc1, 0, 0, 0);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if constexpr (X44) {
#pragma unroll
      for (int i = 0; i < 4; ++i) red[wave][ks4 * 8 + rb4 * 4 + i][cb4 * 4 + x4] = acc0[i];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][4 * kq + r][li] = acc0[r] + acc1[r];
    }
    __syncthreads();
    EESEN_STAMP(2);
    if (e_ok) {
      float dm = dy;
#pragma unroll
      for (int w = 0; w < NW; ++w) dm += X44 ? red[w][es][eu] + red[w][8 + es][eu] : red[w][es][eu];
      const float g = gt.x, i = gt.y, f = gt.z, o = gt.w;
      const float h = tanhf_(c_t);
      const float dh = (1.f - h * h) * (dm * o);
      float dob = o * (1.f - o) * (dm * h);
      float dc = dh + dcf + dn_i * p_i + dn_f * p_f + dob * p_o;
      float dcm = dc, dcx = dc;  // recurrent dropout: see lstm_bwd_step_kernel
      if (DROP) {
        dcm = dc * L.rmask[(size_t)((t + 1) * S + s_e) * ldY + ycol];
        if (L.drop_mode == 2) dcx = dcm;
      }
      float df = f * (1.f - f) * (dcx * c_p);
      float di = i * (1.f - i) * (dcm * g);
      float dg = (1.f - g * g) * (dcm * i);
      float carry = dcx * f;
      if (t >= len) { dg = di = df = dob = 0.f; carry = 0.f; }
      const f32x4 out = {dg, di, df, dob};
      const unsigned ooff = (unsigned)(((size_t)(t * S - tbS + s_e) * ldG + gcol) * 4);
      __builtin_amdgcn_raw_buffer_store_b128(out, rDG, ooff, 0, kSc1);
      dcf = carry; dn_i = di; dn_f = df;
    }
    EESEN_STAMP(3);
    if (step + 1 < T) {
      if (tid < ST * 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      EESEN_STAMP(4);
      if (tid == 0) {
        __hip_atomic_fetch_add(my_cnt + (bx & (kShards - 1)) * kShardStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (e_ok) {  // next step's operands, issued after the publish
        const int t2 = dir == 0 ? t - 1 : t + 1, tp2 = dir == 0 ? t2 - 1 : t2 + 1;
        gt = *reinterpret_cast<const float4*>(L.G + (size_t)(t2 * S + s_e) * ldG + gcol);
        dy = dY[(size_t)(t2 * S + s_e) * lddy + ycol];
        c_t = c_p;  // the next step's cell is this step's recurrence source
        c_p = L.C[(size_t)((tp2 + 1) * S + s_e) * ldY + ycol];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward, 4 sequences x 32 units per workgroup (EESEN_BWD_Q4, default): the 4x4x1 MFMA with CBSZ = 3.
// With the padding rows gone (see X44 in lstm_bwd_persistent_kernel) the step is bound by the operand FETCH: 64 KB of DG_next per
// workgroup for 8 sequences.  The 16-block MFMA does not care how the 128 outputs of a workgroup are shaped, so here they are
// 4 sequences x 32 units: blocks = 8 unit quads x 2 k classes, ALL EIGHT blocks of a k class share the A lanes of block ABID
// (CBSZ = 3, one operand register serves eight instructions), the fetch is 32 KB per workgroup and step -- half --, W_m^T is
// not replicated at all (32 units x 256 k per wave = 128 registers per lane, the last quarter in LDS as in X44), and the grid is
// still one workgroup per CU (H/32 x ndir x S/4 = 256 at cfg2).  Lane l: k class ks = l >> 5, block ab = (l >> 2) & 7, x = l & 3.
//   A: lane (ks, ab, x) loads 16 bytes of sequence x at k = pair*64 + (ks*8 + ab)*4: per sequence and load two whole lines;
//   instruction (pair, r, ABID) takes component r: class ks covers k = pair*64 + ks*32 + ABID*4 + r;
//   B: lane (ks, cb = ab, x) holds W_m^T[unit cb*4 + x][that k]; D: vgpr i, lane (ks, cb, x) -> out[sequence i][unit cb*4 + x].
// Shapes: H % 32 == 0 up to 512 cells (K = 4H split over the 8 waves in CPW / 2 pairs of 64-float chunks, CPW = 2 ceil(H / 128); chunks
// beyond 4H read as zero -- the recipes' 320 cells run as CPW = 6 since round 5), no dropout, gate gradients below 2 GB.
// ST = 8 (round 5): TWO 4-sequence tiles per workgroup against the same resident W_m^T -- for batches whose 4-sequence grid needs
// more workgroups than there are CUs (S = 64 at H = 512: 512), where the alternative was the 16 x 16 tile and its 128 KB of
// operands per workgroup and step.  The MFMA chain doubles (1.8 us), the operand fetch is 64 KB, the hand-off, the drain and the
// cell phase (now on four waves) are paid once for eight sequences.
// ------------------------------------------------------------------------------------------------
// (Round 4 measured an arm that requests the cell operands of a step at the TOP of that step instead of at the end of the step
// before, and bumps the counter from a wave nobody waits on: bit-identical, 7140 ticks per step either way -- the stall it removed
// sat under the peers' own increment flight.  DESIGN.md section 4 "The cell operands at the top of the step" keeps the timeline;
// the arm is gone.)
template <int CPW, int ST>   // ST = 4 or 8 sequences per workgroup (8: two 4-sequence A tiles against the same resident weights)
__global__ __launch_bounds__(NW * 64) void lstm_bwd_persistent_q4_kernel(LstmLayerDev L, const float* __restrict__ dY, int lddy,
                                                                         float* __restrict__ DG, unsigned* cnt, unsigned* err,
                                                                         int spin_limit, unsigned long long* trace, Role R) {
  static_assert(ST == 4 || ST == 8, "one or two 4-sequence tiles");
  constexpr int HS = ST / 4, UW = 32, P = CPW / 2;    // 4-sequence tiles, units per workgroup; (load, 64-float) pairs per wave
  constexpr int LDSP = CPW == 8 ? 1 : 0, REGP = P - LDSP;   // pairs whose B values live in LDS / registers
  __shared__ __attribute__((aligned(16))) float4 bl[LDSP ? LDSP : 1][8][LDSP ? NW * 64 : 1];   // [pair][ABID][thread]
  __shared__ float red[NW][HS * 8][UW + 1];          // [wave][4-sequence tile x k class (2) x sequence (4)][unit]
  __shared__ int s_go;
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = L.H, S = L.S, T = L.T;
  const int ldY = L.ndir * H, ldG = L.ndir * 4 * H, K4 = 4 * H;
  const int bx = R.unit_group(blockIdx.x), dir = R.dir(blockIdx.x), bz = R.seq_group(blockIdx.x);
  const int u0 = bx * UW, s0 = bz * ST;
  unsigned* my_cnt = cnt + (size_t)(dir * R.nz + bz) * kShards * kShardStride;
  const unsigned nblk = R.nblk;
  const int ks = lane >> 5, ab = (lane >> 2) & 7, x = lane & 3;
  float bw[REGP ? REGP : 1][4][8];   // [pair][component r][ABID]
  {
    const int ub = u0 + ab * 4 + x;
    const float* Br = L.WmT + ((size_t)dir * H + min(ub, H - 1)) * K4;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int k0 = (wave + p * NW) * 64 + ks * 32;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ub < H && k0 + q * 4 < K4) v = *reinterpret_cast<const float4*>(Br + k0 + q * 4);
        if (p < REGP) { bw[p < REGP ? p : 0][0][q] = v.x; bw[p < REGP ? p : 0][1][q] = v.y; bw[p < REGP ? p : 0][2][q] = v.z; bw[p < REGP ? p : 0][3][q] = v.w; }
        else bl[p < REGP ? 0 : p - REGP][q][tid] = v;
      }
    }
  }
  const int es = tid >> 5, eu = tid & 31;
  const int s_e = s0 + es, u_e = u0 + eu;
  const bool e_ok = tid < ST * UW && s_e < S && u_e < H;
  float p_i = 0.f, p_f = 0.f, p_o = 0.f;
  int len = 0;
  if (e_ok) {
    const float* pp = L.peep + (size_t)dir * 3 * H + u_e;
    p_i = pp[0]; p_f = pp[H]; p_o = pp[2 * H];
    len = L.lens[s_e];
  }
  float dcf = 0.f, dn_i = 0.f, dn_f = 0.f;
  const size_t gcol = (size_t)dir * K4 + u_e * 4;
  const size_t ycol = (size_t)dir * H + u_e;
  float4 gt = make_float4(0.f, 0.f, 0.f, 0.f);
  float dy = 0.f, c_t = 0.f, c_p = 0.f;
  {
    const int t0 = dir == 0 ? T - 1 : 0, tp0 = dir == 0 ? t0 - 1 : t0 + 1;
    if (e_ok) {
      gt = *reinterpret_cast<const float4*>(L.G + (size_t)(t0 * S + s_e) * ldG + gcol);
      dy = dY[(size_t)(t0 * S + s_e) * lddy + ycol];
      c_t = L.C[(size_t)((t0 + 1) * S + s_e) * ldY + ycol];
      c_p = L.C[(size_t)((tp0 + 1) * S + s_e) * ldY + ycol];
    }
  }
  const __amdgpu_buffer_rsrc_t rDG = make_rsrc(DG);
  __syncthreads();   // bl is complete
  for (int step = 0; step < T; ++step) {
    const int t = dir == 0 ? T - 1 - step : step;
    const int tn = dir == 0 ? t + 1 : t - 1;
    f32x4 ac[HS][2];
#pragma unroll
    for (int hs = 0; hs < HS; ++hs) { ac[hs][0] = f32x4{0.f, 0.f, 0.f, 0.f}; ac[hs][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    f32x4 a4[HS][P];
    // the cell operands of THIS step: requested at the end of the step before, behind the publish
    const float4 gtc = gt;
    const float dyc = dy, ctc = c_t, cpc = c_p;
    EESEN_STAMP(0);
    if (step > 0) {
      if (wave == EESEN_POLL_WAVE) {
        const bool go = wait_counters(my_cnt, nblk, (unsigned)step, err, spin_limit, lane, L.poll_delay);
        if (lane == 0) s_go = go ? 1 : 0;
      }
      __syncthreads();
      if (!s_go) return;
      EESEN_STAMP(1);
#pragma unroll
      for (int hs = 0; hs < HS; ++hs) {
        const unsigned arow = (unsigned)(((size_t)(tn * S + s0 + hs * 4 + x) * ldG + (size_t)dir * K4) * 4);
        const bool rok = s0 + hs * 4 + x < S;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int k = (wave + p * NW) * 64 + (ks * 8 + ab) * 4;
          a4[hs][p] = __builtin_amdgcn_raw_buffer_load_b128(rDG, (rok && k < K4) ? arow + (unsigned)k * 4u : 0x80000000u, 0, 0);
        }
      }
    }
    if (step > 0) {
      __builtin_amdgcn_sched_barrier(0);  // all loads in flight BEFORE the first MFMA
#pragma unroll
      for (int p = 0; p < P; ++p) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {   // ABID 2h and 2h + 1
          float w0[4], w1[4];
          if (p < REGP) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { w0[r] = bw[p < REGP ? p : 0][r][2 * h]; w1[r] = bw[p < REGP ? p : 0][r][2 * h + 1]; }
          } else {
            int lt = tid;
            asm volatile("" : "+v"(lt));   // opaque per step (see X44)
            const float4 v0 = bl[p < REGP ? 0 : p - REGP][2 * h][lt], v1 = bl[p < REGP ? 0 : p - REGP][2 * h + 1][lt];
            w0[0] = v0.x; w0[1] = v0.y; w0[2] = v0.z; w0[3] = v0.w;
            w1[0] = v1.x; w1[1] = v1.y; w1[2] = v1.z; w1[3] = v1.w;
          }
          // (ST = 8: the second 4-sequence tile's chain goes through the SAME weight registers -- per tile the same instructions in the
          // same order as ST = 4, so its gate gradients are bit-identical to the 4-sequence kernel's)
#pragma unroll
          for (int hs = 0; hs < HS; ++hs) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (h == 0)      { ac[hs][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[hs][p][r], w0[r], ac[hs][0], 3, 0, 0); ac[hs][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[hs][p][r], w1[r], ac[hs][1], 3, 1, 0); }
              else if (h == 1) { ac[hs][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[hs][p][r], w0[r], ac[hs][0], 3, 2, 0); ac[hs][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[hs][p][r], w1[r], ac[hs][1], 3, 3, 0); }
              else if (h == 2) { ac[hs][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[hs][p][r], w0[r], ac[hs][0], 3, 4, 0); ac[hs][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[hs][p][r], w1[r], ac[hs][1], 3, 5, 0); }
              else             { ac[hs][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[hs][p][r], w0[r], ac[hs][0], 3, 6, 0); ac[hs][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[hs][p][r], w1[r], ac[hs][1], 3, 7, 0); }
            }
          }
        }
      }
    }
#pragma unroll
    for (int hs = 0; hs < HS; ++hs)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[wave][hs * 8 + ks * 4 + i][ab * 4 + x] = ac[hs][0][i] + ac[hs][1][i];
    EESEN_STAMP(2);
    __syncthreads();
    if (e_ok) {
      // (all sixteen LDS reads in flight, then the sum in the same order: see the forward kernel)
      float ra[NW], rb[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) { ra[w] = red[w][(es >> 2) * 8 + (es & 3)][eu]; rb[w] = red[w][(es >> 2) * 8 + 4 + (es & 3)][eu]; }
      __builtin_amdgcn_sched_barrier(0);
      float dm = dyc;
#pragma unroll
      for (int w = 0; w < NW; ++w) dm += ra[w] + rb[w];
      const float g = gtc.x, i = gtc.y, f = gtc.z, o = gtc.w;
      const float h = tanhf_(ctc);
      const float dh = (1.f - h * h) * (dm * o);
      float dob = o * (1.f - o) * (dm * h);
      const float dc = dh + dcf + dn_i * p_i + dn_f * p_f + dob * p_o;
      float df = f * (1.f - f) * (dc * cpc);
      float di = i * (1.f - i) * (dc * g);
      float dg = (1.f - g * g) * (dc * i);
      float carry = dc * f;
      if (t >= len) { dg = di = df = dob = 0.f; carry = 0.f; }
      const f32x4 out = {dg, di, df, dob};
      __builtin_amdgcn_raw_buffer_store_b128(out, rDG, (unsigned)(((size_t)(t * S + s_e) * ldG + gcol) * 4), 0, kSc1);
      dcf = carry; dn_i = di; dn_f = df;
    }
    EESEN_STAMP(3);
    if (step + 1 < T) {
      if (tid < ST * UW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      EESEN_STAMP(4);
      if (tid == 0) __hip_atomic_fetch_add(my_cnt + (bx & (kShards - 1)) * kShardStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e_ok) {
        const int t2 = dir == 0 ? t - 1 : t + 1, tp2 = dir == 0 ? t2 - 1 : t2 + 1;
        gt = *reinterpret_cast<const float4*>(L.G + (size_t)(t2 * S + s_e) * ldG + gcol);
        dy = dY[(size_t)(t2 * S + s_e) * lddy + ycol];
        c_t = c_p;
        c_p = L.C[(size_t)((tp2 + 1) * S + s_e) * ldY + ycol];
      }
    }
  }
}

// The K-split kernels' partial-sum exchange: a partial sum and the step it belongs to travel in ONE 8-byte word (a single
// 8-byte store is single-copy atomic), so the consumer polls the DATA: no counter to increment, no drain of the stores before
// an increment, no second round trip for the data after the counter.  Two slots per word, by step parity: a sibling can be at
// most one exchange ahead (its step s + 1 exchange needs this workgroup's step s + 1 partials).  The launcher zeroes the space
// (tag 0 = no step).
__device__ __forceinline__ void px_put(unsigned long long* p, float v, unsigned tag) {
  __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the words of the three siblings (src != ku) for this thread's (sequence, unit): issued early, looked at late (px_take)
template <int KU>
__device__ __forceinline__ void px_load(const unsigned long long* px, int ku, int off, unsigned long long (&w)[KU]) {
#pragma unroll
  for (int src = 0; src < KU; ++src)
    w[src] = src == ku ? 0ull : __hip_atomic_load(px + (size_t)(ku * KU + src) * 256 + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// adds the three partial sums (in order of src) once all three words carry this step's tag, re-reading while they do not;
// false: a sibling never wrote (bounded spin) or another workgroup gave up
template <int KU>
__device__ __forceinline__ bool px_take(const unsigned long long* px, int ku, int off, unsigned tag, unsigned* err, int spin_limit,
                                        unsigned long long (&w)[KU], float& sum) {
  for (int spins = 0; spins < spin_limit; ++spins) {
    bool ok = true;
#pragma unroll
    for (int src = 0; src < KU; ++src) ok = ok && (src == ku || (unsigned)(w[src] >> 32) == tag);
    if (ok) {
#pragma unroll
      for (int src = 0; src < KU; ++src)
        if (src != ku) sum += __uint_as_float((unsigned)w[src]);
      return true;
    }
    if ((spins & 1023) == 1023 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    __builtin_amdgcn_s_sleep(1);
    px_load<KU>(px, ku, off, w);
  }
  return false;
}

// ------------------------------------------------------------------------------------------------
// backward for WIDE layers (H = 1024: BASELINE configs 4 and 5), K split four ways (EESEN_BWD_KSPLIT, default on).
// The 16 x 16 tile of lstm_bwd_persistent_kernel fetches ALL 4H gate gradients of its 16 sequences every step -- 256 KB per
// workgroup, the same 256 KB in each of the 64 unit groups of a (direction, sequence tile): 64 MB per step through the L1s of
// the chip at ~34 GB/s per CU is 7 of the step's 9.4 us, beside 3.4 us of MFMA.  Here a workgroup multiplies only ONE QUARTER
// of K (the gate gradients of 256 of the 1024 units) into partial sums for FOUR times as many units (64 instead of 16): the
// same 256 KB of W_m^T in its registers, the same MFMA count, a QUARTER of the fetch (64 KB, hidden under the MFMA chain).  The
// four workgroups that share a 64-unit block then exchange their partial sums -- each writes the three 16 x 16 blocks its
// siblings own and reads the three it needs -- and each finishes the cell update of its own 16 units as before.  The
// exchange carries its own flag: every partial sum is an 8-byte word (value, step), polled by the thread that needs it
// (px_put / px_take above), so there is no counter, no drain before an increment and no second round trip; the words are read
// speculatively half way through the own block's MFMA pass.  Measured (cfg4 / cfg5, same box): backward recurrences
// 46.5 -> 38.9 ms / 333 -> 280 ms with a counted exchange; own block last 37.5 ms; tagged words 36 ms.
//   roles: unit block uu = bx / 4 (64 units), K quarter ku = bx % 4; cell units = uu*64 + ku*16 .. +16
//   hand-off 1 (DG_t): a consumer of quarter ku needs only the 16 producers whose cell units lie in that quarter's 256 units
//   hand-off 2 (partials): tagged words, two slots by step parity (4 MB for S = 32; zeroed by the launcher)
// Shapes: H % 256 == 0, 16-sequence tiles, no dropout.  Same cell arithmetic; the d_m sum is formed in a different order
// (as every backward variant here: parity tests, not bit equality, hold it).
// ------------------------------------------------------------------------------------------------
template <int CPW>   // 32-float chunks of this workgroup's K quarter per wave: (4H / 4) / (32 * NW)
__global__ __launch_bounds__(NW * 64) void lstm_bwd_persistent_ksplit_kernel(LstmLayerDev L, const float* __restrict__ dY, int lddy,
                                                                             float* __restrict__ DG, unsigned long long* __restrict__ PX, unsigned* cnt,
                                                                             unsigned* err, int spin_limit, Role R, int chunk, unsigned long long* trace) {
  constexpr int KU = 4, ST = 16, UW = 64, NT = 4;
  __shared__ float red[NW][ST][48 + 1];    // partial sums of the three sibling blocks, per wave
  __shared__ float red2[NW][ST][16 + 1];   // ... of the own block
  __shared__ int s_go, s_fail;
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = L.H, S = L.S, T = L.T;
  const int ldY = L.ndir * H, ldG = L.ndir * 4 * H, K4 = 4 * H, KQ = K4 / KU;
  const int bx = R.unit_group(blockIdx.x), dir = R.dir(blockIdx.x), bz = R.seq_group(blockIdx.x);
  const int uu = bx / KU, ku = bx % KU;
  const int um0 = uu * UW, uc0 = um0 + ku * 16;            // units of the MFMA outputs / of the cell update
  const int s0 = L.s_begin + bz * ST;
  const int s_end = L.s_begin + (L.s_count ? L.s_count : S);
  const int g = dir * R.nz + bz, ngroups = R.ndir * R.nz, nub = H / UW;
  const unsigned nprod = (unsigned)(H / KU / 16);          // producers of one K quarter
  unsigned* wait_cnt = cnt + (size_t)(g * KU + ku) * kShards * kShardStride;
  unsigned* pub_cnt = cnt + (size_t)(g * KU + uc0 / (H / KU)) * kShards * kShardStride + (size_t)(((uc0 / 16) % (int)nprod) & (kShards - 1)) * kShardStride;

  const int li = lane & 15, kq = lane >> 4;
  const int sa = s0 + li;
  // this wave's part of W_m^T: 64 unit rows x its CPW chunks of the K quarter, resident for the whole layer pass.  Local tile
  // n = 0..2 are the three SIBLINGS' 16-unit blocks (in order of their ku), n = 3 is this workgroup's own block: the siblings'
  // partial sums are produced, sent and on their way while the own block's quarter of the MFMA chain still runs.
  float b[NT][CPW][8];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int ut = n == 3 ? ku : n + (n >= ku ? 1 : 0);
    const float* Br = L.WmT + ((size_t)dir * H + um0 + ut * 16 + li) * K4 + (size_t)ku * KQ;
#pragma unroll
    for (int c = 0; c < CPW; ++c) ld8_plain(Br, (wave + c * NW) * 32 + kq * 8, KQ, true, b[n][c]);
  }
  const int es = tid >> 4, eu = tid & 15;
  const int s_e = s0 + es, u_e = uc0 + eu;
  const bool e_ok = tid < ST * 16 && s_e < s_end;
  float p_i = 0.f, p_f = 0.f, p_o = 0.f;
  int len = 0;
  if (e_ok) {
    const float* pp = L.peep + (size_t)dir * 3 * H + u_e;
    p_i = pp[0]; p_f = pp[H]; p_o = pp[2 * H];
    len = L.lens[s_e];
  }
  float dcf = 0.f, dn_i = 0.f, dn_f = 0.f;
  const size_t gcol = (size_t)dir * K4 + u_e * 4;
  const size_t ycol = (size_t)dir * H + u_e;
  float4 gt = make_float4(0.f, 0.f, 0.f, 0.f);
  float dy = 0.f, c_t = 0.f, c_p = 0.f;
  {
    const int t0 = dir == 0 ? T - 1 : 0, tp0 = dir == 0 ? t0 - 1 : t0 + 1;
    if (e_ok) {
      gt = *reinterpret_cast<const float4*>(L.G + (size_t)(t0 * S + s_e) * ldG + gcol);
      dy = dY[(size_t)(t0 * S + s_e) * lddy + ycol];
      c_t = L.C[(size_t)((t0 + 1) * S + s_e) * ldY + ycol];
      c_p = L.C[(size_t)((tp0 + 1) * S + s_e) * ldY + ycol];
    }
  }
  if (tid == 0) s_fail = 0;   // (the first barrier of step 1 orders it)
  __amdgpu_buffer_rsrc_t rDG = make_rsrc(DG);   // re-based once per chunk of steps (gate gradients beyond 2 GB), see lstm_bwd_persistent_kernel
  int tbS = 0;

  for (int step = 0; step < T; ++step) {
    const int t = dir == 0 ? T - 1 - step : step;
    const int tn = dir == 0 ? t + 1 : t - 1;
    if (chunk < T && step % chunk == 0) {
      const int tb = dir == 0 ? max(0, T - step - chunk) : max(0, step - 1);
      tbS = tb * S;
      rDG = make_rsrc(DG + (size_t)tbS * ldG);
    }
    float dm_in = 0.f;
    EESEN_STAMP(0);
    if (step > 0) {
      if (wave == EESEN_POLL_WAVE) {
        const bool go = wait_counters(wait_cnt, nprod, (unsigned)step, err, spin_limit, lane, L.poll_delay);
        if (lane == 0) s_go = go ? 1 : 0;
      }
      __syncthreads();
      if (!s_go) return;
      EESEN_STAMP(1);
      f32x4 acc[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      const size_t arow = ((size_t)(tn * S - tbS + sa) * ldG + (size_t)dir * K4 + (size_t)ku * KQ) * 4;
      float a[CPW][8];
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        const int k = (wave + c * NW) * 32 + kq * 8;
        ld8_sc1(rDG, (unsigned)(arow + (size_t)k * 4), k, KQ, sa < s_end, a[c]);
      }
      __builtin_amdgcn_sched_barrier(0);  // all loads in flight BEFORE the first MFMA
      // Costing probes (scripts/build_variant.py -DEESEN_PROBE_KSPLIT=n; never in the product build; results are garbage, only the
      // step time is read -- DESIGN.md section 9 round 6, the bf16-plane ledger): bit 0 = the MFMA chain shortened to what six
      // v_mfma_f32_16x16x32_bf16 per 32-wide k block would take (96 x 16 cycles per wave instead of 128 x 32: every fifth MFMA of
      // the fp32 chain is kept... see EESEN_PROBE_MFMA below); bit 1 = the operand fetch grown by half (three bf16 planes = 6 bytes
      // per gate gradient instead of 4).
#ifdef EESEN_PROBE_KSPLIT
#define EESEN_PROBE_MFMA(c, j) ((EESEN_PROBE_KSPLIT & 1) == 0 || (((c) * 8 + (j)) % 8) < 3)   /* 3 of 8: 48 x 32 = 96 x 16 cycles */
      if (EESEN_PROBE_KSPLIT & 2) {
        float extra[CPW / 2 > 0 ? CPW / 2 : 1][8];
#pragma unroll
        for (int c = 0; c < CPW / 2; ++c) {
          const int k = (wave + c * NW) * 32 + kq * 8;
          ld8_sc1(rDG, (unsigned)(arow + (size_t)k * 4 + (size_t)KQ * 4 * ((ku + 1) % KU - ku)), k, KQ, sa < s_end, extra[c]);   // another quarter's rows: lines nobody else of this workgroup reads
        }
#pragma unroll
        for (int c = 0; c < CPW / 2; ++c)
#pragma unroll
          for (int j = 0; j < 8; ++j) a[c][j] += 1e-30f * extra[c][j];
      }
#else
#define EESEN_PROBE_MFMA(c, j) true
#endif
      // pass 1: the siblings' three blocks (three accumulators interleaved)
#pragma unroll
      for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int n = 0; n < 3; ++n) if (EESEN_PROBE_MFMA(c, j)) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], b[n][c][j], acc[n], 0, 0, 0);
      // C/D map of the 16x16 MFMA: col = lane & 15 (unit), row = 4 * (lane >> 4) + reg (sequence)
#pragma unroll
      for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][4 * kq + r][n * 16 + li] = acc[n][r];
      __syncthreads();
      unsigned long long* px = PX + ((size_t)((size_t)(step & 1) * ngroups + g) * nub + uu) * (KU * KU * 256);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int o = tid + h * (NW * 64);
        if (o < ST * 48) {
          const int sq = o / 48, uc = o % 48, n = uc >> 4, dst = n + (n >= ku ? 1 : 0);
          float v = 0.f;
#pragma unroll
          for (int w = 0; w < NW; ++w) v += red[w][sq][uc];
          px_put(px + (size_t)(dst * KU + ku) * 256 + sq * 16 + (uc & 15), v, (unsigned)step);
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // the stores leave BEFORE pass 2 (nothing waits for them)
      // pass 2: the own block -- its quarter of the MFMA chain runs while the siblings' partial sums are in flight
#pragma unroll
      for (int c = 0; c < CPW / 2; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) if (EESEN_PROBE_MFMA(c, j)) acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], b[3][c][j], acc[3], 0, 0, 0);
      // half way through, the siblings' words are read SPECULATIVELY: the siblings run in lockstep, their stores left when this
      // workgroup's did, and the answer is back by the end of the pass (px_take re-reads in the rare case it was too early)
      __builtin_amdgcn_sched_barrier(0);
      unsigned long long sw[KU] = {};
      if (e_ok) px_load<KU>(px, ku, es * 16 + eu, sw);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = CPW / 2; c < CPW; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) if (EESEN_PROBE_MFMA(c, j)) acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], b[3][c][j], acc[3], 0, 0, 0);
#undef EESEN_PROBE_MFMA
#pragma unroll
      for (int r = 0; r < 4; ++r) red2[wave][4 * kq + r][li] = acc[3][r];
      EESEN_STAMP(2);
      __syncthreads();
      if (e_ok) {   // the three siblings' partial sums: they run in lockstep with this workgroup, the words left them a pass ago
#pragma unroll
        for (int w = 0; w < NW; ++w) dm_in += red2[w][es][eu];
        if (!px_take<KU>(px, ku, es * 16 + eu, (unsigned)step, err, spin_limit, sw, dm_in)) {
          __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_fail = 1;
        }
      }
      EESEN_STAMP(3);
    }
    if (e_ok) {
      const float dm = dy + dm_in;
      const float g_ = gt.x, i = gt.y, f = gt.z, o = gt.w;
      const float h = tanhf_(c_t);
      const float dh = (1.f - h * h) * (dm * o);
      float dob = o * (1.f - o) * (dm * h);
      const float dc = dh + dcf + dn_i * p_i + dn_f * p_f + dob * p_o;
      float df = f * (1.f - f) * (dc * c_p);
      float di = i * (1.f - i) * (dc * g_);
      float dg = (1.f - g_ * g_) * (dc * i);
      float carry = dc * f;
      if (t >= len) { dg = di = df = dob = 0.f; carry = 0.f; }
      const f32x4 out = {dg, di, df, dob};
      __builtin_amdgcn_raw_buffer_store_b128(out, rDG, (unsigned)(((size_t)(t * S - tbS + s_e) * ldG + gcol) * 4), 0, kSc1);
      dcf = carry; dn_i = di; dn_f = df;
    }
    if (step + 1 < T) {
      if (tid < ST * 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (s_fail) return;
      EESEN_STAMP(4);
      if (tid == 0) __hip_atomic_fetch_add(pub_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e_ok) {  // next step's operands, issued after the publish
        const int t2 = dir == 0 ? t - 1 : t + 1, tp2 = dir == 0 ? t2 - 1 : t2 + 1;
        gt = *reinterpret_cast<const float4*>(L.G + (size_t)(t2 * S + s_e) * ldG + gcol);
        dy = dY[(size_t)(t2 * S + s_e) * lddy + ycol];
        c_t = c_p;
        c_p = L.C[(size_t)((tp2 + 1) * S + s_e) * ldY + ycol];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The K-split backward tile on TWO fp16 PLANES per operand (round 6; LstmLayerDev::bwd_f16, EESEN_BWD_F16).  The fp32-input MFMA
// chain above is 3.4 of the step's 7.1 us (128 v_mfma_f32_16x16x4_f32 per wave, two waves per SIMD).  Two fp16 planes of W_m^T take
// the fp32 rows' registers (128 per lane) and three v_mfma_f32_16x16x32_f16 products per 32-wide k block replace 8 x 1 fp32 ones:
// 48 MFMAs of 16 cycles per wave and step.  What stood in the way: the A operand is the gate
// gradient this very kernel produces step by step -- no bound is known before the launch, and fp16 has 5 exponent bits.  So the
// PRODUCER scales: a wave of the cell phase (four sequences x 16 cells: four DPP rows) takes the maximum of its 256 gate gradients,
// derives the power of two that brings it into [2^14, 2^15), publishes the gradients a second time as two fp16 planes of the scaled
// values (LstmLayerDev::DGH: where a cell's four fp32 gradients sit in DG, its 16 bytes hold [4 x hi][4 x lo]: one whole-word store per
// lane) and the inverse power in LstmLayerDev::EX (one 16-byte word per wave of the producer = per four sequences x 64 values:
// [t][dir][16-sequence tile][producer][4] -- a 128-byte line holds two producers of ONE tile and ONE K quarter, whose consumers have
// waited for both; a line that also held another quarter's words could be read, and cached by an XCD's L2, before those were
// written); the fp32 gradients still go to DG for the GEMMs and the bias / peephole passes, stored LAST and left in flight by the
// drain in front of the publish.  The CONSUMER's
// 32-wide k block lies inside one producer's 64 values, so its three products carry ONE power per output row: they go through a
// temporary accumulator that is folded into the running one with the row's inverse power (4 FMAs per block and 16-unit tile).
// Same roles, hand-offs and partial-sum exchange as lstm_bwd_persistent_ksplit_kernel.  Error per product <= 3 * 2^-22 |ab|
// (gemm.hip, "half" mode, has the argument); padding frames publish zeros (scale 2^126: 0 stays 0).
// Shapes: as the fp32 K-split tile with an even number of k blocks per wave (H = 512, 1024).
// (Costing probes, never in the product build: -DEESEN_PROBE_KH=1 no plane stores, 2 no exponent loads, 4 no exponent stores.)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dpp_row_max16(float v) {   // maximum over the 16 lanes of a DPP row, in every lane of it
  v = fmaxf(v, __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0xB1, 0xf, 0xf, true)));    // quad_perm [1,0,3,2]
  v = fmaxf(v, __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0x4E, 0xf, 0xf, true)));    // quad_perm [2,3,0,1]
  v = fmaxf(v, __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0x141, 0xf, 0xf, true)));   // row_half_mirror
  v = fmaxf(v, __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0x140, 0xf, 0xf, true)));   // row_mirror
  return v;
}
template <int CPW>
__global__ __launch_bounds__(NW * 64) void lstm_bwd_persistent_ksplit_h_kernel(LstmLayerDev L, const float* __restrict__ dY, int lddy,
                                                                               float* __restrict__ DG, unsigned long long* __restrict__ PX, unsigned* cnt,
                                                                               unsigned* err, int spin_limit, Role R, int chunk, unsigned long long* trace) {
  constexpr int KU = 4, ST = 16, UW = 64, NT = 4;
  __shared__ float red[NW][ST][48 + 1];    // partial sums of the three sibling blocks, per wave
  __shared__ float red2[NW][ST][16 + 1];   // ... of the own block
  __shared__ int s_go, s_fail;
  __shared__ float park[6][ST * 16];       // the cell threads' peephole weights (loop-invariant) and carries (d_c f, d_i, d_f): read once per step, and the register file is full
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = L.H, S = L.S, T = L.T;
  const int ldY = L.ndir * H, ldG = L.ndir * 4 * H, K4 = 4 * H, KQ = K4 / KU;
  const int bx = R.unit_group(blockIdx.x), dir = R.dir(blockIdx.x), bz = R.seq_group(blockIdx.x);
  const int uu = bx / KU, ku = bx % KU;
  const int um0 = uu * UW, uc0 = um0 + ku * 16;            // units of the MFMA outputs / of the cell update
  const int s0 = L.s_begin + bz * ST;
  const int s_end = L.s_begin + (L.s_count ? L.s_count : S);
  const int g = dir * R.nz + bz, ngroups = R.ndir * R.nz, nub = H / UW;
  const unsigned nprod = (unsigned)(H / KU / 16);          // producers of one K quarter
  const int NP = H / 16;                                   // producers (16-unit groups) per direction
  const int NZ = (S + ST - 1) / ST, zt = s0 / ST;          // 16-sequence tiles of the whole batch / this workgroup's
  static_assert(CPW % 2 == 0, "pass 2 takes the k blocks in pairs");
  unsigned* wait_cnt = cnt + (size_t)(g * KU + ku) * kShards * kShardStride;
  unsigned* pub_cnt = cnt + (size_t)(g * KU + uc0 / (H / KU)) * kShards * kShardStride + (size_t)(((uc0 / 16) % (int)nprod) & (kShards - 1)) * kShardStride;

  const int li = lane & 15, kq = lane >> 4;
  const int sa = s0 + li;
  // this wave's part of W_m^T as two fp16 planes of 2^s W (s from the layer's max |W_m|): 64 unit rows x its CPW blocks of the K quarter
  float wscale, winv;
  half_scale(L.wm_amax, wscale, winv);
  f32x4 bh[NT][CPW], bl[NT][CPW];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int ut = n == 3 ? ku : n + (n >= ku ? 1 : 0);
    const float* Br = L.WmT + ((size_t)dir * H + um0 + ut * 16 + li) * K4 + (size_t)ku * KQ;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      float w[8];
      ld8_plain(Br, (wave + c * NW) * 32 + kq * 8, KQ, true, w);
      f32x4 ph, pl;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float w0 = w[2 * j] * wscale, w1 = w[2 * j + 1] * wscale;     // exact (a power of two)
        const unsigned h0 = rne_f16(w0), h1 = rne_f16(w1);
        const unsigned l0 = rne_f16(w0 - f16_bits_to_f32(h0)), l1 = rne_f16(w1 - f16_bits_to_f32(h1));
        ph[j] = __uint_as_float(h0 | (h1 << 16));
        pl[j] = __uint_as_float(l0 | (l1 << 16));
      }
      bh[n][c] = ph; bl[n][c] = pl;
    }
  }
  const int es = tid >> 4, eu = tid & 15;
  const int s_e = s0 + es, u_e = uc0 + eu;
  const bool e_act = tid < ST * 16;            // the cell waves: whole DPP rows of 16 threads per sequence
  const bool e_ok = e_act && s_e < s_end;
  int len = 0;
  if (e_act) {
    float p_i = 0.f, p_f = 0.f, p_o = 0.f;
    if (e_ok) {
      const float* pp = L.peep + (size_t)dir * 3 * H + u_e;
      p_i = pp[0]; p_f = pp[H]; p_o = pp[2 * H];
      len = L.lens[s_e];
    }
    park[0][tid] = p_i; park[1][tid] = p_f; park[2][tid] = p_o;
    park[3][tid] = 0.f; park[4][tid] = 0.f; park[5][tid] = 0.f;
  }
  const size_t gcol = (size_t)dir * K4 + u_e * 4;
  const size_t ycol = (size_t)dir * H + u_e;
  float4 gt = make_float4(0.f, 0.f, 0.f, 0.f);
  float dy = 0.f, c_t = 0.f, c_p = 0.f;
  {
    const int t0 = dir == 0 ? T - 1 : 0, tp0 = dir == 0 ? t0 - 1 : t0 + 1;
    if (e_ok) {
      gt = *reinterpret_cast<const float4*>(L.G + (size_t)(t0 * S + s_e) * ldG + gcol);
      dy = dY[(size_t)(t0 * S + s_e) * lddy + ycol];
      c_t = L.C[(size_t)((t0 + 1) * S + s_e) * ldY + ycol];
      c_p = L.C[(size_t)((tp0 + 1) * S + s_e) * ldY + ycol];
    }
  }
  if (tid == 0) s_fail = 0;   // (the first barrier of step 1 orders it)
  __amdgpu_buffer_rsrc_t rDG = make_rsrc(DG);   // re-based once per chunk of steps (gate gradients beyond 2 GB), see lstm_bwd_persistent_kernel
  __amdgpu_buffer_rsrc_t rDH = make_rsrc(reinterpret_cast<const float*>(L.DGH));   // the planes: the fp32 rows' bytes, the same offsets
  const __amdgpu_buffer_rsrc_t rEX = make_rsrc(L.EX);     // bytes
  int tbS = 0;

  for (int step = 0; step < T; ++step) {
    const int t = dir == 0 ? T - 1 - step : step;
    const int tn = dir == 0 ? t + 1 : t - 1;
    if (chunk < T && step % chunk == 0) {
      const int tb = dir == 0 ? max(0, T - step - chunk) : max(0, step - 1);
      tbS = tb * S;
      rDG = make_rsrc(DG + (size_t)tbS * ldG);
      rDH = make_rsrc(reinterpret_cast<const float*>(L.DGH) + (size_t)tbS * ldG);
    }
    float dm_in = 0.f;
    EESEN_STAMP(0);
    if (step > 0) {
      if (wave == EESEN_POLL_WAVE) {
        const bool go = wait_counters(wait_cnt, nprod, (unsigned)step, err, spin_limit, lane, L.poll_delay);
        if (lane == 0) s_go = go ? 1 : 0;
      }
      __syncthreads();
      if (!s_go) return;
      EESEN_STAMP(1);
      f32x4 acc[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      const size_t arow = ((size_t)(tn * S - tbS + sa) * ldG + (size_t)dir * K4 + (size_t)ku * KQ) * 4;
      constexpr unsigned kOob = 0x80000000u;
      f32x4 u0[CPW], u1[CPW];   // per k block two units' words of [4 x hi][4 x lo]; regrouped into the A fragments at their use (below)
      unsigned iv[CPW];   // the inverse powers' exponent bytes of the producer's four 4-sequence groups (this lane's accumulator registers hold group kq: C/D map row = 4 * kq + reg)
      // the inverse powers first (four bytes per lane and k block: back long before the planes; the first fold needs them)
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        const int blk = wave + c * NW;                     // 32-wide k block of the quarter: inside producer ku * nprod + blk / 2
        const int prod = ku * (int)nprod + blk / 2;
#if defined(EESEN_PROBE_KH) && (EESEN_PROBE_KH & 2)
        const bool iok = false;
#else
        const bool iok = blk * 32 < KQ;
#endif
        iv[c] = __builtin_amdgcn_raw_buffer_load_b32(rEX, iok ? (unsigned)((((size_t)(tn * L.ndir + dir) * NZ + zt) * NP + prod) * 64 + kq * 16) : kOob, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        const int blk = wave + c * NW;
        const bool ok = sa < s_end && blk * 32 < KQ;
        const unsigned off = (unsigned)(arow + (size_t)(blk * 32 + kq * 8) * 4);
        // (PLAIN loads, like ld8_sc1 above: every line is read for the first time in this launch, and the 16 workgroups of an XCD that
        // read the same quarter share one fabric fetch through its L2; with L1-bypassing sc1 loads each pulled its own copy: 25 GB/s per CU)
        u0[c] = __builtin_amdgcn_raw_buffer_load_b128(rDH, ok ? off : kOob, 0, 0);
        u1[c] = __builtin_amdgcn_raw_buffer_load_b128(rDH, ok ? off + 16u : kOob, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);  // all loads in flight BEFORE the first MFMA -- and nothing that reads a loaded register (the
                                          // regrouping moves: they would wait for every load) on this side of the fence
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      // the hi halves of both units make the A fragment's eight hi values, the lo halves its eight lo (same lane -> k assignment as W's planes)
      auto frag_hi = [&](int c) { return f32x4{u0[c][0], u0[c][1], u1[c][0], u1[c][1]}; };
      auto frag_lo = [&](int c) { return f32x4{u0[c][2], u0[c][3], u1[c][2], u1[c][3]}; };
      // pass 1: the siblings' three blocks (three chains interleaved); per k block lo x hi', hi x lo', hi x hi' into a temporary, folded
      // into the accumulator with the producer's inverse power
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        f32x4 tmp[3];
        const f32x4 ahc = frag_hi(c), alc = frag_lo(c);
#pragma unroll
        for (int n = 0; n < 3; ++n) tmp[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, alc), __builtin_bit_cast(f16x8_t, bh[n][c]), zero4, 0, 0, 0);
#pragma unroll
        for (int n = 0; n < 3; ++n) tmp[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, ahc), __builtin_bit_cast(f16x8_t, bl[n][c]), tmp[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < 3; ++n) tmp[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, ahc), __builtin_bit_cast(f16x8_t, bh[n][c]), tmp[n], 0, 0, 0);
        const float ivc = __uint_as_float(iv[c]);
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[n][r] = fmaf(tmp[n][r], ivc, acc[n][r]);
      }
      // C/D map of the 16x16 MFMA: col = lane & 15 (unit), row = 4 * (lane >> 4) + reg (sequence)
#pragma unroll
      for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][4 * kq + r][n * 16 + li] = acc[n][r] * winv;
      __syncthreads();
      unsigned long long* px = PX + ((size_t)((size_t)(step & 1) * ngroups + g) * nub + uu) * (KU * KU * 256);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int o = tid + h * (NW * 64);
        if (o < ST * 48) {
          const int sq = o / 48, uc = o % 48, n = uc >> 4, dst = n + (n >= ku ? 1 : 0);
          float rv[NW];   // (all eight LDS reads in flight, then the sum in the same order)
#pragma unroll
          for (int w = 0; w < NW; ++w) rv[w] = red[w][sq][uc];
          float v = 0.f;
#pragma unroll
          for (int w = 0; w < NW; ++w) v += rv[w];
          px_put(px + (size_t)(dst * KU + ku) * 256 + sq * 16 + (uc & 15), v, (unsigned)step);
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // the stores leave BEFORE pass 2 (nothing waits for them)
      // pass 2: the own block, two k blocks at a time (two independent chains); half way through, the siblings' words are read
      // SPECULATIVELY (see the fp32 kernel)
      unsigned long long sw[KU] = {};
#pragma unroll
      for (int c = 0; c < CPW; c += 2) {
        f32x4 tmp[2];
        const f32x4 ah2[2] = {frag_hi(c), frag_hi(c + 1)}, al2[2] = {frag_lo(c), frag_lo(c + 1)};
#pragma unroll
        for (int d = 0; d < 2; ++d) tmp[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, al2[d]), __builtin_bit_cast(f16x8_t, bh[3][c + d]), zero4, 0, 0, 0);
#pragma unroll
        for (int d = 0; d < 2; ++d) tmp[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, ah2[d]), __builtin_bit_cast(f16x8_t, bl[3][c + d]), tmp[d], 0, 0, 0);
#pragma unroll
        for (int d = 0; d < 2; ++d) tmp[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, ah2[d]), __builtin_bit_cast(f16x8_t, bh[3][c + d]), tmp[d], 0, 0, 0);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const float ivc = __uint_as_float(iv[c + d]);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[3][r] = fmaf(tmp[d][r], ivc, acc[3][r]);
        }
        if (c == 0) {
          __builtin_amdgcn_sched_barrier(0);
          if (e_ok) px_load<KU>(px, ku, es * 16 + eu, sw);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) red2[wave][4 * kq + r][li] = acc[3][r] * winv;
      EESEN_STAMP(2);
      __syncthreads();
      if (e_ok) {   // the three siblings' partial sums: they run in lockstep with this workgroup, the words left them a pass ago
        float rv[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) rv[w] = red2[w][es][eu];
#pragma unroll
        for (int w = 0; w < NW; ++w) dm_in += rv[w];
        if (!px_take<KU>(px, ku, es * 16 + eu, (unsigned)step, err, spin_limit, sw, dm_in)) {
          __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_fail = 1;
        }
      }
      EESEN_STAMP(3);
    }
    if (e_act) {
      const float dm = dy + dm_in;
      const float g_ = gt.x, i = gt.y, f = gt.z, o = gt.w;
      const float h = tanhf_(c_t);
      const float dh = (1.f - h * h) * (dm * o);
      float dob = o * (1.f - o) * (dm * h);
      const float dc = dh + park[3][tid] + park[4][tid] * park[0][tid] + park[5][tid] * park[1][tid] + dob * park[2][tid];
      float df = f * (1.f - f) * (dc * c_p);
      float di = i * (1.f - i) * (dc * g_);
      float dg = (1.f - g_ * g_) * (dc * i);
      float carry = dc * f;
      if (t >= len || !e_ok) { dg = di = df = dob = 0.f; carry = 0.f; }
      // the power of two of this wave's four sequences x 64 gate gradients (four DPP rows of 16 threads), then the planes
      float sc, inv;
      {
        const float rm = dpp_row_max16(fmaxf(fmaxf(fabsf(dg), fabsf(di)), fmaxf(fabsf(df), fabsf(dob))));
        const int ri = (int)__float_as_uint(rm);
        const float wm = fmaxf(fmaxf(__uint_as_float((unsigned)__builtin_amdgcn_readlane(ri, 0)), __uint_as_float((unsigned)__builtin_amdgcn_readlane(ri, 16))),
                               fmaxf(__uint_as_float((unsigned)__builtin_amdgcn_readlane(ri, 32)), __uint_as_float((unsigned)__builtin_amdgcn_readlane(ri, 48))));
        half_scale(wm, sc, inv);
      }
#if !defined(EESEN_PROBE_KH) || !(EESEN_PROBE_KH & 4)
      if (lane == 0 && s0 + 4 * wave < s_end) {   // one 16-byte word per wave (a whole-word write-through store; sub-dword ones cost 900 ticks of drain): [t][dir][tile][producer][4 groups]
        const f32x4 iv4 = {inv, inv, inv, inv};
        __builtin_amdgcn_raw_buffer_store_b128(iv4, rEX, (unsigned)((((size_t)(t * L.ndir + dir) * NZ + zt) * NP + uc0 / 16) * 64 + wave * 16), 0, kSc1);
      }
#endif
      if (e_ok) {
        const f32x4 out = {dg, di, df, dob};
        const unsigned o32 = (unsigned)(((size_t)(t * S - tbS + s_e) * ldG + gcol) * 4);
        const f32x2_t x01 = {dg * sc, di * sc}, x23 = {df * sc, dob * sc};         // exact
        const f16x2_t h01 = __builtin_convertvector(x01, f16x2_t), h23 = __builtin_convertvector(x23, f16x2_t);   // v_cvt_pk_f16_f32: round to nearest even
        const f32x2_t r01 = {x01[0] - (float)h01[0], x01[1] - (float)h01[1]}, r23 = {x23[0] - (float)h23[0], x23[1] - (float)h23[1]};   // exact
        const f16x2_t l01 = __builtin_convertvector(r01, f16x2_t), l23 = __builtin_convertvector(r23, f16x2_t);
        // this unit's 16 bytes (where its fp32 gradients sit in DG): [4 x hi][4 x lo] -- ONE whole-word store per lane, the 16 lanes of a
        // row cover 256 contiguous bytes (two 8-byte stores with holes between the lanes' pieces cost 1400 ticks of drain)
        const f32x4 pp = {__builtin_bit_cast(float, h01), __builtin_bit_cast(float, h23), __builtin_bit_cast(float, l01), __builtin_bit_cast(float, l23)};
#if !defined(EESEN_PROBE_KH) || !(EESEN_PROBE_KH & 1)
        __builtin_amdgcn_raw_buffer_store_b128(pp, rDH, o32, 0, kSc1);
#endif
        // the fp32 gradients LAST: nobody reads them before the kernel ends (the GEMMs and the bias / peephole passes do), so the
        // drain in front of the publish leaves this one store in flight (s_waitcnt vmcnt(1) below)
        __builtin_amdgcn_raw_buffer_store_b128(out, rDG, o32, 0, kSc1);
        park[3][tid] = carry; park[4][tid] = di; park[5][tid] = df;
      }
    }
    if (step + 1 < T) {
      if (tid < ST * 16) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");   // everything but the newest store: the fp32 gradients (see above)
      __syncthreads();
      if (s_fail) return;
      EESEN_STAMP(4);
      if (tid == 0) __hip_atomic_fetch_add(pub_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e_ok) {  // next step's operands, issued after the publish
        const int t2 = dir == 0 ? t - 1 : t + 1, tp2 = dir == 0 ? t2 - 1 : t2 + 1;
        gt = *reinterpret_cast<const float4*>(L.G + (size_t)(t2 * S + s_e) * ldG + gcol);
        dy = dY[(size_t)(t2 * S + s_e) * lddy + ycol];
        c_t = c_p;
        c_p = L.C[(size_t)((tp2 + 1) * S + s_e) * ldY + ycol];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K-split backward, TWO sequence tiles per workgroup (time-multiplexed): BASELINE config 5 (S = 64 at H = 1024).
// 64 sequences need 512 K-split workgroups -- two windows of 32, one launch after the other, 2 x 7.8 us per time step.  Here one
// launch covers all four tiles: a workgroup keeps its W_m^T slice ONCE and steps two independent chains, tile 2g and 2g + 1 of
// its direction, alternately and in two phases per time step:
//     MFMA(A) + partials(A) -> MFMA(B) + partials(B) -> cell(A) + publish(A) -> cell(B) + publish(B)
// so every wait of a chain falls behind the other chain's work: A's sibling partials were sent a whole MFMA chain (3.4 us) before
// cell(A) asks for them, A's gate gradients were published a cell phase before the next MFMA(A), B's an MFMA chain before
// MFMA(B).  Same arithmetic per chain as lstm_bwd_persistent_ksplit_kernel (bit-identical results).
// ------------------------------------------------------------------------------------------------
template <int CPW>
__global__ __launch_bounds__(NW * 64) void lstm_bwd_persistent_ksplit_mux_kernel(LstmLayerDev L, const float* __restrict__ dY, int lddy,
                                                                                 float* __restrict__ DG, unsigned long long* __restrict__ PX, unsigned* cnt,
                                                                                 unsigned* err, int spin_limit, Role R, int chunk) {
  constexpr int KU = 4, ST = 16, UW = 64, NT = 4, Q = 2;
  __shared__ float red[NW][ST][UW + 1];
  __shared__ float own[Q][ST][17];
  __shared__ int s_go, s_fail;
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = L.H, S = L.S, T = L.T;
  const int ldY = L.ndir * H, ldG = L.ndir * 4 * H, K4 = 4 * H, KQ = K4 / KU;
  const int bx = R.unit_group(blockIdx.x), dir = R.dir(blockIdx.x), bg = R.seq_group(blockIdx.x);
  const int uu = bx / KU, ku = bx % KU;
  const int um0 = uu * UW, uc0 = um0 + ku * 16;
  const int nzall = (S + ST - 1) / ST, ngroups = L.ndir * nzall, nub = H / UW;
  const unsigned nprod = (unsigned)(H / KU / 16);
  const int li = lane & 15, kq = lane >> 4;
  float b[NT][CPW][8];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const float* Br = L.WmT + ((size_t)dir * H + um0 + n * 16 + li) * K4 + (size_t)ku * KQ;
#pragma unroll
    for (int c = 0; c < CPW; ++c) ld8_plain(Br, (wave + c * NW) * 32 + kq * 8, KQ, true, b[n][c]);
  }
  const int es = tid >> 4, eu = tid & 15;
  const int u_e = uc0 + eu;
  float p_i = 0.f, p_f = 0.f, p_o = 0.f;
  if (tid < ST * 16) {
    const float* pp = L.peep + (size_t)dir * 3 * H + u_e;
    p_i = pp[0]; p_f = pp[H]; p_o = pp[2 * H];
  }
  const size_t gcol = (size_t)dir * K4 + u_e * 4;
  const size_t ycol = (size_t)dir * H + u_e;
  // per chain
  int zt[Q], s0[Q], s_e[Q], len[Q], g[Q];
  bool live[Q], e_ok[Q];
  float dcf[Q], dn_i[Q], dn_f[Q], dy[Q], c_t[Q], c_p[Q], dm_in[Q];
  float4 gt[Q];
  unsigned *wait_cnt[Q], *pub_cnt[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    zt[q] = bg * Q + q;
    live[q] = zt[q] < nzall;
    s0[q] = zt[q] * ST;
    s_e[q] = s0[q] + es;
    e_ok[q] = live[q] && tid < ST * 16 && s_e[q] < S;
    g[q] = dir * nzall + (live[q] ? zt[q] : 0);
    wait_cnt[q] = cnt + (size_t)(g[q] * KU + ku) * kShards * kShardStride;
    pub_cnt[q] = cnt + (size_t)(g[q] * KU + uc0 / (H / KU)) * kShards * kShardStride + (size_t)(((uc0 / 16) % (int)nprod) & (kShards - 1)) * kShardStride;
    len[q] = e_ok[q] ? L.lens[s_e[q]] : 0;
    dcf[q] = dn_i[q] = dn_f[q] = dy[q] = c_t[q] = c_p[q] = dm_in[q] = 0.f;
    gt[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int t0 = dir == 0 ? T - 1 : 0, tp0 = dir == 0 ? t0 - 1 : t0 + 1;
    if (e_ok[q]) {
      gt[q] = *reinterpret_cast<const float4*>(L.G + (size_t)(t0 * S + s_e[q]) * ldG + gcol);
      dy[q] = dY[(size_t)(t0 * S + s_e[q]) * lddy + ycol];
      c_t[q] = L.C[(size_t)((t0 + 1) * S + s_e[q]) * ldY + ycol];
      c_p[q] = L.C[(size_t)((tp0 + 1) * S + s_e[q]) * ldY + ycol];
    }
  }
  if (tid == 0) s_fail = 0;
  __amdgpu_buffer_rsrc_t rDG = make_rsrc(DG);
  int tbS = 0;

  for (int step = 0; step < T; ++step) {
    const int t = dir == 0 ? T - 1 - step : step;
    const int tn = dir == 0 ? t + 1 : t - 1;
    if (chunk < T && step % chunk == 0) {
      const int tb = dir == 0 ? max(0, T - step - chunk) : max(0, step - 1);
      tbS = tb * S;
      rDG = make_rsrc(DG + (size_t)tbS * ldG);
    }
    // ---- phase 1, both chains: partial sums of this K quarter from the chain's DG_next
    unsigned long long sw[Q][KU] = {};
    if (step > 0) {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        if (!live[q]) continue;   // odd number of tiles: the last workgroup group steps one chain only (uniform per workgroup)
        if (wave == EESEN_POLL_WAVE) {
          const bool go = wait_counters(wait_cnt[q], nprod, (unsigned)step, err, spin_limit, lane, 0);
          if (lane == 0) s_go = go ? 1 : 0;
        }
        __syncthreads();
        if (!s_go) return;
        f32x4 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int sa = s0[q] + li;
        const size_t arow = ((size_t)(tn * S - tbS + sa) * ldG + (size_t)dir * K4 + (size_t)ku * KQ) * 4;
        float a[CPW][8];
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
          const int k = (wave + c * NW) * 32 + kq * 8;
          ld8_sc1(rDG, (unsigned)(arow + (size_t)k * 4), k, KQ, sa < S, a[c]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < CPW / 2; ++c)
#pragma unroll
          for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], b[n][c][j], acc[n], 0, 0, 0);
        // chain A's sibling words, sent before this chain's fetch began, are asked for half way through chain B's MFMA chain:
        // back long before phase 2 looks at them
        __builtin_amdgcn_sched_barrier(0);
        if (q == 1 && e_ok[0])
          px_load<KU>(PX + ((size_t)((size_t)(step & 1) * ngroups + g[0]) * nub + uu) * (KU * KU * 256), ku, es * 16 + eu, sw[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = CPW / 2; c < CPW; ++c)
#pragma unroll
          for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], b[n][c][j], acc[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[wave][4 * kq + r][n * 16 + li] = acc[n][r];
        __syncthreads();
        unsigned long long* px = PX + ((size_t)((size_t)(step & 1) * ngroups + g[q]) * nub + uu) * (KU * KU * 256);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int o = tid + h * (NW * 64), sq = o >> 6, uc = o & 63, dst = uc >> 4;
          float v = 0.f;
#pragma unroll
          for (int w = 0; w < NW; ++w) v += red[w][sq][uc];
          if (dst == ku) own[q][sq][uc & 15] = v;
          else px_put(px + (size_t)(dst * KU + ku) * 256 + sq * 16 + (uc & 15), v, (unsigned)step);
        }
        __syncthreads();   // fences `red` for the other chain (the stores are not waited for)
      }
    }
    // ---- phase 2, both chains: the siblings' partials, the cell update, publish
    if (step > 0) {   // chain B's words (and chain A's where chain B is idle): under chain A's cell update
      if (live[1] && e_ok[1]) px_load<KU>(PX + ((size_t)((size_t)(step & 1) * ngroups + g[1]) * nub + uu) * (KU * KU * 256), ku, es * 16 + eu, sw[1]);
      if (!live[1] && e_ok[0]) px_load<KU>(PX + ((size_t)((size_t)(step & 1) * ngroups + g[0]) * nub + uu) * (KU * KU * 256), ku, es * 16 + eu, sw[0]);
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if (!live[q]) continue;
      float dm = dy[q];
      if (step > 0) {
        if (e_ok[q]) {
          const unsigned long long* px = PX + ((size_t)((size_t)(step & 1) * ngroups + g[q]) * nub + uu) * (KU * KU * 256);
          dm += own[q][es][eu];
          if (!px_take<KU>(px, ku, es * 16 + eu, (unsigned)step, err, spin_limit, sw[q], dm)) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_fail = 1;
          }
        }
      }
      if (e_ok[q]) {
        const float g_ = gt[q].x, i = gt[q].y, f = gt[q].z, o = gt[q].w;
        const float h = tanhf_(c_t[q]);
        const float dh = (1.f - h * h) * (dm * o);
        float dob = o * (1.f - o) * (dm * h);
        const float dc = dh + dcf[q] + dn_i[q] * p_i + dn_f[q] * p_f + dob * p_o;
        float df = f * (1.f - f) * (dc * c_p[q]);
        float di = i * (1.f - i) * (dc * g_);
        float dg = (1.f - g_ * g_) * (dc * i);
        float carry = dc * f;
        if (t >= len[q]) { dg = di = df = dob = 0.f; carry = 0.f; }
        const f32x4 out = {dg, di, df, dob};
        __builtin_amdgcn_raw_buffer_store_b128(out, rDG, (unsigned)(((size_t)(t * S - tbS + s_e[q]) * ldG + gcol) * 4), 0, kSc1);
        dcf[q] = carry; dn_i[q] = di; dn_f[q] = df;
      }
      if (step + 1 < T) {
        if (tid < ST * 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s_fail) return;
        if (tid == 0) __hip_atomic_fetch_add(pub_cnt[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (e_ok[q]) {
          const int t2 = dir == 0 ? t - 1 : t + 1, tp2 = dir == 0 ? t2 - 1 : t2 + 1;
          gt[q] = *reinterpret_cast<const float4*>(L.G + (size_t)(t2 * S + s_e[q]) * ldG + gcol);
          dy[q] = dY[(size_t)(t2 * S + s_e[q]) * lddy + ycol];
          c_t[q] = c_p[q];
          c_p[q] = L.C[(size_t)((tp2 + 1) * S + s_e[q]) * ldY + ycol];
        }
      }
    }
  }
}

// EESEN_GPU_SHARE=n (tuning.h): this process may count on 1/n of the device's CUs -- n trainer processes on one GPU (the reference's
// own two-jobs-one-GPU test mode; tests/test_gpu_multirank.py) each size their persistent grids against their share, so that all
// of them are co-resident TOGETHER instead of relying on the spin time-outs to find out that they are not.
// (round 6: a communicator that finds several of its ranks on ONE device sets the share itself -- set_gpu_share, comm.cpp -- so that
// co-located trainer jobs need no environment variable; an explicit EESEN_GPU_SHARE wins)
static std::atomic<int> g_share_override{0};
static int gpu_share() {
  static const int env = [] { const char* e = getenv("EESEN_GPU_SHARE"); const int v = e && *e ? atoi(e) : 0; return v < 1 ? 0 : v; }();
  if (env) return env;
  const int o = g_share_override.load(std::memory_order_relaxed);
  return o > 0 ? o : 1;
}
// the CUs this process sizes its tiles and grids against: the device's, divided by EESEN_GPU_SHARE
static int share_of_cus() {
  int ncu = 256, dev = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  return std::max(1, ncu / gpu_share());
}
// census: optional -- `bool(int wgs)`: have `wgs` workgroups of this very kernel been SEEN co-resident on this device (run once, cached)?
template <class K, class C = bool (*)(int)>
bool fits(K kernel, dim3 grid, int threads, C census = nullptr) {  // grid: the workgroups of ONE launch (one sequence window)
  int dev = 0, ncu = 0, nb = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  {  // the occupancy of an instantiation does not change: asked once per (device, kernel) -- this runs several times per layer pass
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, int> cache;   // (instantiations of one template share a function TYPE: keyed by address)
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(dev, reinterpret_cast<const void*>(kernel));
    auto it = cache.find(key);
    if (it == cache.end()) {
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, 0) != hipSuccess) return false;
      it = cache.emplace(key, nb).first;
    }
    nb = it->second;
  }
  // the occupancy query can over-report by one workgroup per CU (MI355X guide: SGPR-heavy kernels): keep a margin of one -- unless the
  // caller has a residency CENSUS for this instantiation (below), which settles whether the query's own number is real
  const long wgs = (long)grid.x * grid.y * grid.z;
  if (wgs <= (long)ncu * std::max(1, nb - 1) / gpu_share()) return true;
  return census && wgs <= (long)ncu * nb / gpu_share() && census((int)wgs);
}

// The persistent grids are launched as ORDINARY kernels: hipLaunchCooperativeKernel costs ~40 us more per launch (its dedicated
// queue; 0.3 ms per cfg2 step over eight launches, measured) and guarantees nothing that fits() has not checked already -- the
// runtime does not gang-schedule a cooperative grid either (the side-stream GEMMs co-run with it), it only refuses grids above the
// occupancy limit, which is the check fits() makes with a workgroup per CU of margin.  Workgroups that are not resident at once
// are waited for by the others' bounded spins, and a spin that gives up is recovered from (net.cpp).  (plan_launch below.)
}  // namespace

void set_gpu_share(int n) { g_share_override.store(n < 1 ? 1 : n, std::memory_order_relaxed); }
int gpu_share_value() { return gpu_share(); }

// One-way flight of an agent-scope increment between two CUs of the current device, in nanoseconds (measured once per device and
// process, ~3 ms: 2000 round trips, the median of five runs).  The first-poll delays of the recurrence kernels are multiples of it.
float handoff_flight_ns() {
  static float cache[64] = {0};
  static std::mutex mu;                       // Nets may be created from several host threads
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0.f;
  if (cache[dev] > 0.f) return cache[dev];
  unsigned* flags = nullptr;
  unsigned long long* out = nullptr;
  EESEN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&flags), 64 * sizeof(unsigned)));
  EESEN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&out), sizeof(unsigned long long)));
  const int rounds = 2000;
  double ns[5];
  for (int rep = 0; rep < 5; ++rep) {
    EESEN_HIP_CHECK(hipMemset(flags, 0, 64 * sizeof(unsigned)));
    hipLaunchKernelGGL(handoff_pingpong_kernel, dim3(2), dim3(64), 0, nullptr, flags, out, rounds);
    unsigned long long ticks = 0;
    EESEN_HIP_CHECK(hipMemcpy(&ticks, out, sizeof(ticks), hipMemcpyDeviceToHost));
    // 0 ticks: the two workgroups never saw each other (not co-resident on a busy device): no measurement, the default flight
    ns[rep] = ticks ? 10.0 * (double)ticks / (2.0 * rounds) : 500.0;
  }
  std::sort(ns, ns + 5);
  (void)hipFree(flags);
  (void)hipFree(out);
  // (the median of five.  The measurement has two modes on an MI355X -- 370-420 ns and 550-620 ns, by where the dispatcher puts the two
  // workgroups; some boxes read the far one every time -- and a step's time does not depend on which one its process measured: net.cpp
  // uses the delays tuned on this part whenever the reading is in that range, and the reading itself only as a plausibility check)
  cache[dev] = (float)std::min(2000.0, std::max(100.0, ns[2]));
  return cache[dev];
}

// ctl: [0 .. 2*ndir*nz) arrival counters (fwd then bwd use disjoint halves via `ctl_off`), last word = error flag
// Forward tile (sequences x hidden units per workgroup), chosen so that every CU gets ONE workgroup where the shape allows:
//   32 x 4  <MT=2, NT=1>: the decomposition of lstm_fwd_step_kernel (any H)
//   16 x 8  <MT=1, NT=2>: half the m_{t-1} fetch per CU; 256 workgroups at H = 512, S = 32
//   16 x 16 <MT=1, NT=4>: for wide layers (H = 1024: 16 x 8 would need 512 co-resident workgroups); 64 gate rows of W_m
//                         (256 KB at H = 1024) live in the registers of one workgroup
struct FwdTile { int mt, nt; };
static bool narrow2_ok(const LstmLayerDev& L, int need);
static FwdTile fwd_tile(const LstmLayerDev& L) {
  const int need = ((L.H + 31) / 32 + NW - 1) / NW;
  const int ncu = share_of_cus();
  // (wide layers take the 16-sequence tile at ANY batch size: their 32 x 4 tile would need H/4 x ndir workgroups -- 512 at H = 1024 --
  // and a batch of <= 16 sequences fell back to the per-step kernels: seen at S = 16, T = 3000 with the six-layer cfg5 stack, round 4)
  // (S <= 16 on narrow layers: the 32 x 4 tile has twice the workgroups of the 16 x 8 tile -- but only the latter has the bf16-pipe
  // kernel; LstmLayerDev::fwd_t16_small takes it there too where that kernel applies)
  const bool t16_ok = L.H % 8 == 0 && (L.S > 16 || need > 2 || (L.fwd_t16_small && L.fwd_split && !L.drop_mode && L.X && L.H % 32 == 0));
  // The narrow 16 x 8 tile needs H/8 x ndir x S/16 workgroups: one per CU up to S = 32 at H = 512.  Beyond that, up to TWO per CU
  // (round 5; S = 64 at H = 512: 512 workgroups of the bf16-pipe kernel, 108 registers) where a residency census has SEEN that many
  // co-resident (narrow2_ok) -- two chains per CU interleaved by the hardware: 3.5 us per step for 64 sequences against 4.2 on the
  // wide tile; otherwise the wide 16 x 16 tile.
  const long narrow_wgs = (long)(L.H / 8) * L.ndir * cdiv(L.S, 16);
  const bool narrow_full = narrow_wgs > ncu && !(narrow_wgs <= 2L * ncu && need <= 2 && narrow2_ok(L, need));
  if (t16_ok && L.H % 16 == 0 && need <= 4 && (need > 2 || narrow_full)) return {1, 4};
  if (t16_ok && need <= 2) return {1, 2};
  return {2, 1};
}

// Sequence windows: the smallest number of equal windows (each a multiple of the sequence tile) whose workgroups can all be
// co-resident.  1 for every configuration but the largest (S = 64 at H = 1024 needs 512 workgroups of the wide tiles: two
// windows of 32 sequences, run one after the other -- the sequences are independent chains).
template <class F>
static int pick_windows(int S, int seq_tile, F fits_with) {
  for (int nwin = 1; nwin <= 8; nwin *= 2) {
    if (S % nwin != 0 || (S / nwin) % seq_tile != 0) { if (nwin == 1 && fits_with(S)) return 1; continue; }
    if (fits_with(S / nwin)) return nwin;
  }
  return 0;
}

// Which instantiation of lstm_fwd_persistent_bf_kernel, if any, this layer's forward recurrence takes:
//   * BASELINE config 4's bf16 forward (L.fwd_bf16: W_m as hi + lo planes, m_t one plane): the wide tile's
//     geometry (16 sequences x 16 units), whole 256-unit multiples (each of the 8 waves owns CPW = H / 256 chunks of 32 units);
//   * the fp32-class 3-way split (L.fwd_split; three planes each): wherever the narrow 16 x 8 fp32 tile would be taken.
// Both: exchange buffer present, no recurrent dropout, block offsets within 32 bits -- and the instantiation's workgroups co-resident in
// SOME number of sequence windows: a shape or device on which only the bf16-pipe tile does not fit falls through to the fp32 tiles
// (which ran in this slot before round 4) instead of dropping to the one-launch-per-step kernels.
//   * round 6, fp32-class on two fp16 planes per operand (L.fwd_f16 and the layer's W_m bound on hand): the narrow tile like the
//     3-way split, AND the wide 16 x 16 tile (whole 256-unit multiples) that had only the fp32-input kernel.
struct BfPlan { bool on; int cpw, nt, ap, wp; bool f16; };
static BfPlan bf_plan_shape(const LstmLayerDev& L) {
  const BfPlan off{false, 0, 0, 0, 0, false};
  if (L.X == nullptr || L.drop_mode || L.T < 2 || L.H % 32 != 0) return off;
  auto small = [&](int ap) { return (size_t)L.T * L.ndir * cdiv(L.S, 16) * (size_t)(L.H / 32) * 1024 * ap < ((size_t)1 << 31); };
  if (L.fwd_bf16) {
    if (L.H % 256 != 0 || L.H / 256 > 4 || !small(1)) return off;
    return {true, L.H / 256, 4, 1, 2, false};
  }
  const bool f16 = L.fwd_f16 && L.wm_amax != nullptr;
  if (L.fwd_split || f16) {
    const FwdTile ft = fwd_tile(L);
    const int need = (L.H / 32 + NW - 1) / NW;
    if (f16 && ft.mt == 1 && ft.nt == 4 && L.H % 256 == 0 && L.H / 256 <= 4 && small(2)) return {true, L.H / 256, 4, 2, 2, true};
    if (ft.mt != 1 || ft.nt != 2 || need > 2 || L.H % 8 != 0 || !small(3)) return off;
    if (f16) return {true, need, 2, 2, 2, true};
    if (L.fwd_split) return {true, need, 2, 3, 3, false};
  }
  return off;
}
// calls F<CPW, NT, AP, WP, F16>() for the instantiation of the plan (the combinations bf_plan can return)
#define EESEN_BF_DISPATCH(P, F)                                                                           \
  do {                                                                                                    \
    if ((P).f16 && (P).nt == 2) { if ((P).cpw <= 1) F(1, 2, 2, 2, true); else F(2, 2, 2, 2, true); }       \
    else if ((P).f16) { switch ((P).cpw) { case 1: F(1, 4, 2, 2, true); break; case 2: F(2, 4, 2, 2, true); break; case 3: F(3, 4, 2, 2, true); break; default: F(4, 4, 2, 2, true); } } \
    else if ((P).nt == 2) { if ((P).cpw <= 1) F(1, 2, 3, 3, false); else F(2, 2, 3, 3, false); }           \
    else { switch ((P).cpw) { case 1: F(1, 4, 1, 2, false); break; case 2: F(2, 4, 1, 2, false); break; case 3: F(3, 4, 1, 2, false); break; default: F(4, 4, 1, 2, false); } } \
  } while (0)
// Residency census of one instantiation of the bf16-pipe kernel: `wgs` workgroups launched in its census mode (T < 0) on the idle
// device must all check in within 5 ms.  Once per (device, workgroup count) and process, ~1 ms; the occupancy query's own number is
// only trusted where this has seen it (the MI355X guide: the query over-reports by one block per CU for SGPR-heavy kernels).
template <int C, int N, int A, int W, bool F>
static bool bf_census(int wgs) {
  static std::mutex mu;
  static std::map<std::pair<int, int>, bool> seen;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  // A device this process SHARES (EESEN_GPU_SHARE > 1: peers' grids come and go) has no idle moment to take a census in, and its
  // answer would differ from rank to rank and run to run -- and with it the forward tile, whose two candidates are not bit-identical
  // (ADVICE r5): there the occupancy query's margin decides alone, the same way in every process.
  if (gpu_share() > 1) return false;
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_pair(dev, wgs);
  if (auto it = seen.find(key); it != seen.end()) return it->second;
  bool ok = false;
  unsigned* w = nullptr;
  if (hipDeviceSynchronize() == hipSuccess && hipMalloc(reinterpret_cast<void**>(&w), 2 * sizeof(unsigned)) == hipSuccess) {
    // up to three takes: a grid that fits an idle device is seen at once; one transient occupant (another stream's kernel retiring)
    // must not decide the tile for the life of the process
    for (int take = 0; take < 3 && !ok; ++take) {
      if (hipMemset(w, 0, 2 * sizeof(unsigned)) != hipSuccess) break;
      LstmLayerDev L{};
      L.T = -1;
      hipLaunchKernelGGL((lstm_fwd_persistent_bf_kernel<C, N, A, W, F>), dim3(wgs), dim3(NW * 64), 0, nullptr, L, w, w + 1, 0,
                         static_cast<unsigned long long*>(nullptr), Role{1, 1, 1, 0});
      unsigned e = 1;
      ok = hipMemcpy(&e, w + 1, sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess && e == 0;
    }
    (void)hipFree(w);
  }
  seen[key] = ok;
  if (getenv("EESEN_PRINT_PLAN")) fprintf(stderr, "LOG (eesen_hip) residency census: %d workgroups of lstm_fwd_persistent_bf_kernel<%d,%d,%d,%d,%s> on device %d: %s\n", wgs, C, N, A, W, F ? "true" : "false", dev, ok ? "seen co-resident" : "NOT seen");
  return ok;
}
static bool bf_fits(const LstmLayerDev& L, const BfPlan& P, int Sw) {
  dim3 grid(L.H / (4 * P.nt), L.ndir, cdiv(Sw, 16));
#define EESEN_BF_FITS(C, N, A, W, F) return fits(lstm_fwd_persistent_bf_kernel<C, N, A, W, F>, grid, NW * 64, &bf_census<C, N, A, W, F>)
  EESEN_BF_DISPATCH(P, EESEN_BF_FITS);
#undef EESEN_BF_FITS
  return false;
}
// the narrow bf16-pipe tile as up to two workgroups per CU (fwd_tile): the plan's own shape conditions, and the grid seen co-resident
static bool narrow2_ok(const LstmLayerDev& L, int need) {
  const bool f16 = L.fwd_f16 && L.wm_amax != nullptr;
  if (!L.fwd_narrow2 || !(L.fwd_split || f16) || L.fwd_bf16 || L.X == nullptr || L.drop_mode || L.T < 2 || L.H % 32 != 0) return false;
  if ((size_t)L.T * L.ndir * cdiv(L.S, 16) * (size_t)(L.H / 32) * 1024 * 3 >= ((size_t)1 << 31)) return false;
  return bf_fits(L, f16 ? BfPlan{true, need, 2, 2, 2, true} : BfPlan{true, need, 2, 3, 3, false}, L.S);
}
static BfPlan bf_plan(const LstmLayerDev& L) {
  BfPlan P = bf_plan_shape(L);
  if (P.on && pick_windows(L.S, 16, [&](int Sw) { return bf_fits(L, P, Sw); }) == 0) P.on = false;
  return P;
}

void lstm_fwd_persistent_geometry(const LstmLayerDev& L, int* nblk, int* nz, int* units_per_wg) {
  if (const BfPlan P = bf_plan(L); P.on) {
    *nblk = L.H / (4 * P.nt); *nz = cdiv(L.S, 16);
    if (units_per_wg) *units_per_wg = 4 * P.nt;
    return;
  }
  const FwdTile ft = fwd_tile(L);
  *nblk = L.H / (4 * ft.nt);
  *nz = cdiv(L.S, 16 * ft.mt);
  if (units_per_wg) *units_per_wg = 4 * ft.nt;
}

// true when the forward tile this layer takes leaves register and LDS room for a 128 x 128 GEMM workgroup on the same CU (the
// early middle part of the next layer's input GEMM, net.cpp): the narrow tiles (<= 8 units, <= 131 VGPRs); the wide fp32 tile
// (206 VGPRs) does not
bool lstm_fwd_persistent_leaves_room(const LstmLayerDev& L) {
  // (the bf16 forward's successor in BASELINE config 4 is a projection, not an LSTM layer; two narrow workgroups per CU leave no room)
  if (const BfPlan P = bf_plan(L); P.on) return P.nt <= 2 && (long)(L.H / (4 * P.nt)) * L.ndir * cdiv(L.S, 16) <= share_of_cus();
  const FwdTile ft = fwd_tile(L);
  return 4 * ft.nt <= 8;
}

// ---- the instantiations, by address: a plan carries the host stub of the kernel it chose; occupancy query, resource query and
// launch all go through that one pointer (hipOccupancyMaxActiveBlocksPerMultiprocessor / hipFuncGetAttributes / hipLaunchKernel)
template <int CPW, int MT, int NT>
static const void* fwd_f32_fn(bool drop, bool xchg) {
  if (drop) return reinterpret_cast<const void*>(&lstm_fwd_persistent_kernel<CPW, MT, NT, true>);
  if constexpr (MT == 1) {
    if (xchg) return reinterpret_cast<const void*>(&lstm_fwd_persistent_kernel<CPW, 1, NT, false, true>);
  }
  return reinterpret_cast<const void*>(&lstm_fwd_persistent_kernel<CPW, MT, NT, false>);
}
static const void* fwd_f32_fn(const FwdTile& ft, int need, bool drop, bool xchg) {
  if (ft.nt == 4) return need <= 1 ? fwd_f32_fn<1, 1, 4>(drop, xchg) : need <= 2 ? fwd_f32_fn<2, 1, 4>(drop, xchg) : fwd_f32_fn<4, 1, 4>(drop, xchg);
  if (ft.nt == 2) return need <= 1 ? fwd_f32_fn<1, 1, 2>(drop, xchg) : fwd_f32_fn<2, 1, 2>(drop, xchg);
  return need <= 1 ? fwd_f32_fn<1, 2, 1>(drop, false) : need <= 2 ? fwd_f32_fn<2, 2, 1>(drop, false) : fwd_f32_fn<4, 2, 1>(drop, false);
}
static int fwd_f32_cpw(const FwdTile& ft, int need) { return ft.nt == 2 ? (need <= 1 ? 1 : 2) : (need <= 1 ? 1 : need <= 2 ? 2 : 4); }
static const void* fwd_bf_fn(const BfPlan& B) {
#define EESEN_BF_FN(C, N, A, W, F) return reinterpret_cast<const void*>(&lstm_fwd_persistent_bf_kernel<C, N, A, W, F>)
  EESEN_BF_DISPATCH(B, EESEN_BF_FN);
#undef EESEN_BF_FN
  return nullptr;
}
static const void* bwd_q4_fn(int cpw, int stq) {
#define EESEN_Q4_FN(CPW) (stq == 8 ? reinterpret_cast<const void*>(&lstm_bwd_persistent_q4_kernel<CPW, 8>) : reinterpret_cast<const void*>(&lstm_bwd_persistent_q4_kernel<CPW, 4>))
  switch (cpw) { case 8: return EESEN_Q4_FN(8); case 6: return EESEN_Q4_FN(6); case 4: return EESEN_Q4_FN(4); default: return EESEN_Q4_FN(2); }
#undef EESEN_Q4_FN
}
static const void* bwd_ksplit_h_fn(int cpw) {
  switch (cpw) {
    case 4: return reinterpret_cast<const void*>(&lstm_bwd_persistent_ksplit_h_kernel<4>);
    case 2: return reinterpret_cast<const void*>(&lstm_bwd_persistent_ksplit_h_kernel<2>);
    default: return nullptr;
  }
}
static const void* bwd_ksplit_fn(int cpw, bool mux) {
  switch (cpw) {
    case 4: return mux ? reinterpret_cast<const void*>(&lstm_bwd_persistent_ksplit_mux_kernel<4>) : reinterpret_cast<const void*>(&lstm_bwd_persistent_ksplit_kernel<4>);
    case 3: return mux ? reinterpret_cast<const void*>(&lstm_bwd_persistent_ksplit_mux_kernel<3>) : reinterpret_cast<const void*>(&lstm_bwd_persistent_ksplit_kernel<3>);
    case 2: return mux ? reinterpret_cast<const void*>(&lstm_bwd_persistent_ksplit_mux_kernel<2>) : reinterpret_cast<const void*>(&lstm_bwd_persistent_ksplit_kernel<2>);
    default: return nullptr;
  }
}
template <int CPW>
static const void* bwd_generic_fn(int stile, bool drop) {
  if (stile == 8) return drop ? reinterpret_cast<const void*>(&lstm_bwd_persistent_kernel<CPW, 8, true>) : reinterpret_cast<const void*>(&lstm_bwd_persistent_kernel<CPW, 8, false>);
  return drop ? reinterpret_cast<const void*>(&lstm_bwd_persistent_kernel<CPW, 16, true>) : reinterpret_cast<const void*>(&lstm_bwd_persistent_kernel<CPW, 16, false>);
}
static int bwd_generic_cpw(int need) { return need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : 16; }
static const void* bwd_generic_fn(int need, int stile, bool drop) {
  switch (bwd_generic_cpw(need)) {
    case 1: return bwd_generic_fn<1>(stile, drop);
    case 2: return bwd_generic_fn<2>(stile, drop);
    case 4: return bwd_generic_fn<4>(stile, drop);
    case 8: return bwd_generic_fn<8>(stile, drop);
    default: return bwd_generic_fn<16>(stile, drop);
  }
}

// registers and LDS of the chosen instantiation, and what its grid leaves free on a CU (RecPlan::free_vgprs): a SIMD has 512
// registers per lane, allocated in blocks of 8; a 512-thread workgroup is two waves per SIMD
static void plan_resources(RecPlan& P, const dim3& grid) {
  P.grid[0] = (int)grid.x; P.grid[1] = (int)grid.y; P.grid[2] = (int)grid.z;
  P.wgs = (int)(grid.x * grid.y * grid.z);
  P.wgs_per_cu = std::max(1, cdiv(P.wgs, share_of_cus()));
  if (!P.fn) return;
  static std::mutex mu;
  static std::map<const void*, std::pair<int, int>> cache;
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(P.fn);
  if (it == cache.end()) {
    hipFuncAttributes a{};
    if (hipFuncGetAttributes(&a, P.fn) != hipSuccess) { (void)hipGetLastError(); return; }
    it = cache.emplace(P.fn, std::make_pair((int)a.numRegs, (int)a.sharedSizeBytes)).first;
  }
  P.vgprs = it->second.first;
  P.lds = it->second.second;
  if (P.vgprs > 0) P.free_vgprs = 512 - P.wgs_per_cu * (NW / 4) * ((P.vgprs + 7) & ~7);
}

// (see "The persistent grids are launched as ORDINARY kernels" above)
template <class... Args>
static void plan_launch(hipStream_t st, const RecPlan& P, const dim3& grid, Args... args) {
  void* argv[] = {static_cast<void*>(&args)...};
  EESEN_HIP_CHECK(hipLaunchKernel(P.fn, dim3(grid.x * grid.y * grid.z), dim3(NW * 64), argv, 0, st));
}

// number of sequence windows the forward pass of this layer takes (0: no persistent tile fits; 1: the whole batch at once)
int lstm_fwd_persistent_windows(const LstmLayerDev& L) {
  if (const BfPlan P = bf_plan(L); P.on) return pick_windows(L.S, 16, [&](int Sw) { return bf_fits(L, P, Sw); });
  const int nch = (L.H + 31) / 32;
  const int need = (nch + NW - 1) / NW;
  const FwdTile ft = fwd_tile(L);
  auto fits_with = [&](int Sw) {
    dim3 grid(L.H / (4 * ft.nt), L.ndir, cdiv(Sw, 16 * ft.mt));
    return fits(fwd_f32_fn(ft, need, L.drop_mode != 0, false), grid, NW * 64);
  };
  return pick_windows(L.S, 16 * ft.mt, fits_with);
}

// ---- forward ----------------------------------------------------------------------------------------------------------------
RecPlan lstm_fwd_plan(const LstmLayerDev& L0) {
  RecPlan P;
  const int nch = (L0.H + 31) / 32;
  const int need = (nch + NW - 1) / NW;
  const FwdTile ft = fwd_tile(L0);
  if (need > 4 || (ft.nt == 2 && need > 2) || L0.H % (4 * ft.nt) != 0 || L0.T < 2) return P;
  // The hand-off relies on every step reading cache lines nobody has touched before in this launch.  That holds only if
  // a time step's row block [S x ndir*H] of Y starts on a 128-byte line: otherwise the last line of block t also carries
  // the first bytes of block t+1, gets cached (L1 and the XCD's non-coherent L2) while block t+1 is still unwritten,
  // and is read back stale one step later (seen at S = 17, H = 20).  Such shapes use the per-step kernels.
  if (((size_t)L0.S * L0.ndir * L0.H * sizeof(float)) % 128 != 0) return P;
  if ((size_t)(L0.T + 2) * L0.S * L0.ndir * L0.H * sizeof(float) >= ((size_t)1 << 31)) return P;  // 32-bit buffer offsets over all of Y
  const int nwin = lstm_fwd_persistent_windows(L0);
  if (nwin == 0) return P;
  if (nwin > 1 && ((size_t)(L0.S / nwin) * L0.ndir * L0.H * sizeof(float)) % 128 != 0) return P;  // a window's rows start on a line too
  if (const BfPlan B = bf_plan(L0); B.on) {   // on the bf16 matrix pipe: config 4's bf16 forward, or the fp32-class 3-way split (lstm_fwd_persistent_bf_kernel)
    const dim3 grid(L0.H / (4 * B.nt), L0.ndir, cdiv(L0.S / nwin, 16));
    if ((size_t)grid.y * grid.z * kShards * kShardStride > (size_t)kCtlHalf) return P;
    P.kind = kRecFwdBf; P.fn = fwd_bf_fn(B); P.cpw = B.cpw; P.seq_tile = 16; P.units = 4 * B.nt; P.windows = nwin;
    snprintf(P.kernel, sizeof(P.kernel), "lstm_fwd_persistent_bf_kernel<%d,%d,%d,%d,%s>", B.nt == 2 ? (B.cpw <= 1 ? 1 : 2) : std::min(4, std::max(1, B.cpw)), B.nt, B.ap, B.wp, B.f16 ? "true" : "false");
    plan_resources(P, grid);
    return P;
  }
  // Two windows of the wide tile: one launch that time-multiplexes the two sequence tiles of every workgroup instead
  // (lstm_fwd_persistent_mux_kernel).  LstmLayerDev::fwd_mux = 0 (EESEN_FWD_MUX=0): the two launches, one after the other.
  if (L0.fwd_mux && nwin == 2 && ft.mt == 1 && ft.nt == 4 && need > 2 && need <= 4 && !L0.drop_mode && L0.H % 32 == 0) {
    const int nz = cdiv(L0.S, 16), ng = cdiv(nz, 2);
    const bool xchg = L0.X != nullptr && (size_t)L0.T * L0.ndir * nz * (size_t)(L0.H / 32) * 2048 < ((size_t)1 << 31);
    const dim3 grid(L0.H / 16, L0.ndir, ng);
    const void* fn = xchg ? reinterpret_cast<const void*>(&lstm_fwd_persistent_mux_kernel<4, 4, true>) : reinterpret_cast<const void*>(&lstm_fwd_persistent_mux_kernel<4, 4, false>);
    if (fits(fn, grid, NW * 64) && (size_t)L0.ndir * nz * kShards * kShardStride <= (size_t)kCtlHalf) {
      P.kind = kRecFwdMux; P.fn = fn; P.cpw = 4; P.seq_tile = 32; P.units = 16; P.windows = 1; P.xchg = xchg;
      snprintf(P.kernel, sizeof(P.kernel), "lstm_fwd_persistent_mux_kernel<4,4,%s>", xchg ? "true" : "false");
      plan_resources(P, grid);
      return P;
    }
  }
  const dim3 grid(L0.H / (4 * ft.nt), L0.ndir, cdiv(L0.S / nwin, 16 * ft.mt));
  if ((size_t)grid.y * grid.z * kShards * kShardStride > (size_t)kCtlHalf) return P;
  // exchange-layout operand fetch: 16-sequence tiles, whole 32-unit chunks, block offsets within 32 bits
  const bool xchg = L0.X != nullptr && ft.mt == 1 && L0.H % 32 == 0 && !L0.drop_mode &&
                    (size_t)L0.T * L0.ndir * cdiv(L0.S, 16) * (size_t)(L0.H / 32) * 2048 < ((size_t)1 << 31);
  P.kind = kRecFwdF32; P.fn = fwd_f32_fn(ft, need, L0.drop_mode != 0, xchg); P.cpw = fwd_f32_cpw(ft, need);
  P.seq_tile = 16 * ft.mt; P.units = 4 * ft.nt; P.windows = nwin; P.xchg = xchg;
  snprintf(P.kernel, sizeof(P.kernel), "lstm_fwd_persistent_kernel<%d,%d,%d,%s,%s>", P.cpw, ft.mt, ft.nt, L0.drop_mode ? "true" : "false", xchg && !L0.drop_mode ? "true" : "false");
  plan_resources(P, grid);
  return P;
}

bool lstm_fwd_persistent_is_bf16(const LstmLayerDev& L) { return L.fwd_bf16 && lstm_fwd_plan(L).kind == kRecFwdBf; }

bool lstm_fwd_persistent(hipStream_t st, const LstmLayerDev& L0, unsigned* cnt, unsigned* err, int spin_limit,
                         unsigned long long* trace, hipEvent_t after_reset) {
  const RecPlan P = lstm_fwd_plan(L0);
  if (P.kind == kRecNone) return false;
  const dim3 grid(P.grid[0], P.grid[1], P.grid[2]);
  const Role role{P.grid[0], P.grid[1], P.grid[2], L0.xcd_map};
  if (P.kind == kRecFwdMux) {
    LstmLayerDev L = L0;
    L.s_begin = 0; L.s_count = 0;
    EESEN_HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(unsigned) * L0.ndir * cdiv(L0.S, 16) * kShards * kShardStride, st));
    plan_launch(st, P, grid, L, cnt, err, spin_limit, role);
    return true;
  }
  for (int w = 0; w < P.windows; ++w) {
    LstmLayerDev L = L0;
    L.s_count = L0.S / P.windows;
    L.s_begin = w * L.s_count;
    // Two workgroups per CU (the narrow tile at --num-sequence 64): a step takes a quarter longer and its increments land later -- the
    // first poll 700 ns after the publish instead of 400.  Swept on the final kernels (round 6, cfg2 at S = 64, ms per step at a first
    // poll after 300 / 400 / 500 / 600 / 700 / 850 ns: 52.8 / 52.0 / 51.0-51.7 / 51.4 / 51.0 / 51.4; profiles/r06_poll_sweep.log).
    if (P.wgs_per_cu >= 2 && !L0.poll_raw) L.poll_delay = L0.poll_delay * 7 / 4;
    EESEN_HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(unsigned) * grid.y * grid.z * kShards * kShardStride, st));
    if (after_reset && P.windows == 1) EESEN_HIP_CHECK(hipEventRecord(after_reset, st));  // a gated consumer may start polling from here on
    plan_launch(st, P, grid, L, cnt, err, spin_limit, trace, role);
  }
  return true;
}

void wait_for_word(hipStream_t st, const unsigned* word, unsigned target, unsigned* err, double limit_s) {
  const unsigned long long ticks = (unsigned long long)(std::max(0.05, limit_s) * 1e8);   // wall_clock64: 100 MHz
  hipLaunchKernelGGL(wait_for_word_kernel, dim3(1), dim3(64), 0, st, word, target, err, ticks);
  check_launch("wait_for_word");
}

// ---- backward ---------------------------------------------------------------------------------------------------------------
// Floats of partial-sum exchange space the K-split backward kernels need for this layer shape: per (direction, 16-sequence tile)
// group and 64-unit block 16 blocks of 16 x 16 words of 8 bytes (value, step), two slots by step parity (px_put / px_take);
// 0 = the kernel does not apply (narrow layers take the 4 x 32
// tile, dropout layers and odd shapes the generic one).  LstmLayerDev::bwd_ksplit = 0 (EESEN_BWD_KSPLIT=0) switches it off.
size_t lstm_bwd_ksplit_px_floats(const LstmLayerDev& L) {
  if (!L.bwd_ksplit) return 0;
  if (L.drop_mode || L.H % 256 != 0 || L.H < 768 || L.H > 1024 || L.T < 2) return 0;
  const int ncu = share_of_cus();
  const long blocks16 = (long)cdiv(L.H, 16) * L.ndir * cdiv(L.S, 16);
  if (2 * blocks16 <= ncu && L.S > 8) return 0;   // the 8-sequence / 4 x 32 tiles are taken there
  // the largest window the launcher may pick is the whole batch
  return (size_t)2 * L.ndir * cdiv(L.S, 16) * (size_t)(L.H / 64) * 4096 * 2;
}

// the fp16-plane K-split tile applies (shape and switches; the buffers are the caller's to hand over)
static bool bwd_planes_shape(const LstmLayerDev& L) {
  if (!L.bwd_f16 || !L.bwd_ksplit || L.wm_amax == nullptr) return false;
  const int cpw = (4 * L.H / 4) / (32 * NW);
  return lstm_bwd_ksplit_px_floats(L) != 0 && (cpw == 2 || cpw == 4) && (4 * L.H / 4) % (32 * NW) == 0;
}
size_t lstm_bwd_planes_ex_floats(const LstmLayerDev& L) {
  return bwd_planes_shape(L) ? (size_t)L.T * L.ndir * cdiv(L.S, 16) * (size_t)(L.H / 16) * 16 : 0;   // 64 bytes per (t, dir, tile, producer)
}

RecPlan lstm_bwd_plan(const LstmLayerDev& L0, bool assume_px) {
  RecPlan P;
  const int nch = (4 * L0.H + 31) / 32;
  const int need = (nch + NW - 1) / NW;
  // Sequences per workgroup: 16 fills the MFMA rows; 8 wastes half of them but halves the 128 KB of DG_next each workgroup
  // must fetch per step, which is what bounds the step (measured: 3.75 us of fetch at ~34 GB/s per CU vs 1.8 us of MFMA).
  // Take 8 whenever 16 would leave half of the chip's CUs without a workgroup.
  const int ncu = share_of_cus();
  const long blocks16 = (long)cdiv(L0.H, 16) * L0.ndir * cdiv(L0.S, 16);
  const int stile = 2 * blocks16 <= ncu && L0.S > 8 ? 8 : 16;
  P.light = stile == 8;   // (a property of the SHAPE: it also holds for the per-step kernels a shape without a persistent tile takes)
  if (need > 16 || L0.T < 2) return P;
  if (((size_t)L0.S * L0.ndir * 4 * L0.H * sizeof(float)) % 128 != 0) return P;      // line-aligned DG row blocks (see forward)
  // 32-bit buffer offsets: the kernel re-bases its DG resource every `chunk` steps; a chunk touches chunk + 1 row blocks
  const size_t blk_bytes = (size_t)L0.S * L0.ndir * 4 * L0.H * sizeof(float);
  const long max_blocks = (long)((((size_t)1 << 31) - 1) / blk_bytes);
  if (max_blocks < 3) return P;
  int chunk = (int)std::min<long>(L0.T, max_blocks - 1);
  if (chunk < L0.T) { int p2 = 1; while (p2 * 2 <= chunk) p2 *= 2; chunk = p2; }   // a power of two keeps `step % chunk` cheap
  P.chunk = chunk;
  // The 4-sequence x 32-unit tile (lstm_bwd_persistent_q4_kernel<., 4>): wherever the 8-sequence tile would be taken and the shape
  // allows; with TWO 4-sequence tiles per workgroup (<., 8>, round 5) where that grid does not fit but half as many workgroups do
  // (S = 64 at H = 512) -- LstmLayerDev::bwd_q4_st8: 0 = never (the 16 x 16 tile there, as before round 5), 2 = wherever it applies,
  // before the one-tile form (the tests' A/B arm: same gate gradients bit for bit)
  for (int pass = 0; pass < 2; ++pass) {
    const int stq = (L0.bwd_q4_st8 == 2) == (pass == 0) ? 8 : 4;
    // (round 5: any whole number of 32-unit workgroups up to 512 cells -- the recipes' 320 among them: the waves' k chunks beyond 4H read
    // as zero and their weights ARE zero, so K = 4H need not fill the 8 waves x CPW / 2 pairs exactly)
    if (!L0.bwd_q4 || L0.drop_mode || L0.H % 32 != 0 || L0.H > 512 || chunk < L0.T) break;
    // (a ragged last tile is masked in the kernel -- operand rows and cells beyond S -- so S need not be a multiple of the tile: the
    // recipes' default --num-sequence 10 takes this tile too)
    if (stq == 4 && stile != 8) continue;
    if (stq == 8 && !(L0.bwd_q4_st8 && L0.S > 8)) continue;
    const int cpw = 2 * ((L0.H + 127) / 128);   // 2, 4, 6 or 8 chunks of 32 floats per wave: K = 4H in 8 waves x (cpw / 2) pairs of 64
    const dim3 grid(L0.H / 32, L0.ndir, cdiv(L0.S, stq));
    const size_t cwords = (size_t)grid.y * grid.z * kShards * kShardStride;
    const void* fn = bwd_q4_fn(cpw, stq);
    if (fits(fn, grid, NW * 64) && cwords <= (size_t)kCtlHalf) {
      P.kind = kRecBwdQ4; P.fn = fn; P.cpw = cpw; P.stq = stq; P.seq_tile = stq; P.units = 32; P.windows = 1;
      snprintf(P.kernel, sizeof(P.kernel), "lstm_bwd_persistent_q4_kernel<%d,%d>", cpw, stq);
      plan_resources(P, grid);
      return P;
    }
  }
  // Wide layers: K split four ways (lstm_bwd_persistent_ksplit_kernel) wherever the 16-sequence tile would be taken and the caller
  // handed over the partial-sum exchange buffer (lstm_bwd_ksplit_px_floats)
  const size_t px_need = stile == 16 && (L0.PX || assume_px) ? lstm_bwd_ksplit_px_floats(L0) : 0;
  if (px_need && (L0.PX ? L0.px_floats >= px_need : assume_px)) {
    const int cpw = (4 * L0.H / 4) / (32 * NW);
    auto kfits = [&](int Sw) {
      dim3 grid(L0.H / 64 * 4, L0.ndir, cdiv(Sw, 16));
      const size_t c1 = (size_t)grid.y * grid.z * 4 * kShards * kShardStride;
      if (c1 > (size_t)kCtlHalf) return false;
      const void* fn = bwd_ksplit_fn(cpw, false);
      return fn != nullptr && fits(fn, grid, NW * 64);
    };
    // Round 6: the same tile on two fp16 planes per operand (lstm_bwd_persistent_ksplit_h_kernel) where the caller handed over the
    // plane and exponent buffers; batches that need two windows take two launches of it (faster than one multiplexed fp32 launch)
    if (bwd_planes_shape(L0) && (assume_px || (L0.DGH && L0.EX))) {
      const void* fnh = bwd_ksplit_h_fn(cpw);
      auto hfits = [&](int Sw) {
        dim3 grid(L0.H / 64 * 4, L0.ndir, cdiv(Sw, 16));
        const size_t c1 = (size_t)grid.y * grid.z * 4 * kShards * kShardStride;
        return c1 <= (size_t)kCtlHalf && fnh != nullptr && fits(fnh, grid, NW * 64);
      };
      const int nwh = pick_windows(L0.S, 16, hfits);
      if (nwh > 0 && (nwh == 1 || ((size_t)(L0.S / nwh) * L0.ndir * 4 * L0.H * sizeof(float)) % 128 == 0)) {
        const dim3 grid(L0.H / 64 * 4, L0.ndir, cdiv(L0.S / nwh, 16));
        P.kind = kRecBwdKsplit; P.fn = fnh; P.cpw = cpw; P.seq_tile = 16; P.units = 64; P.windows = nwh;
        snprintf(P.kernel, sizeof(P.kernel), "lstm_bwd_persistent_ksplit_h_kernel<%d>", cpw);
        plan_resources(P, grid);
        return P;
      }
    }
    const int nwin = pick_windows(L0.S, 16, kfits);
    // Two windows: one launch that time-multiplexes two sequence tiles per workgroup instead (lstm_bwd_persistent_ksplit_mux_kernel;
    // LstmLayerDev::bwd_mux = 0 / EESEN_BWD_MUX=0: the two launches, one after the other)
    if (nwin == 2 && L0.bwd_mux && cpw >= 2 && cpw <= 4) {
      const int nz = cdiv(L0.S, 16), ng = cdiv(nz, 2);
      const dim3 grid(L0.H / 64 * 4, L0.ndir, ng);
      const size_t c1 = (size_t)L0.ndir * nz * 4 * kShards * kShardStride;
      const void* fn = bwd_ksplit_fn(cpw, true);
      if (c1 <= (size_t)kCtlHalf && fits(fn, grid, NW * 64)) {
        P.kind = kRecBwdKsplitMux; P.fn = fn; P.cpw = cpw; P.seq_tile = 32; P.units = 64; P.windows = 1;
        snprintf(P.kernel, sizeof(P.kernel), "lstm_bwd_persistent_ksplit_mux_kernel<%d>", cpw);
        plan_resources(P, grid);
        return P;
      }
    }
    if (nwin > 0 && (nwin == 1 || ((size_t)(L0.S / nwin) * L0.ndir * 4 * L0.H * sizeof(float)) % 128 == 0)) {
      const dim3 grid(L0.H / 64 * 4, L0.ndir, cdiv(L0.S / nwin, 16));
      P.kind = kRecBwdKsplit; P.fn = bwd_ksplit_fn(cpw, false); P.cpw = cpw; P.seq_tile = 16; P.units = 64; P.windows = nwin;
      snprintf(P.kernel, sizeof(P.kernel), "lstm_bwd_persistent_ksplit_kernel<%d>", cpw);
      plan_resources(P, grid);
      return P;
    }
  }
  const void* fn = bwd_generic_fn(need, stile, L0.drop_mode != 0);
  auto gfits = [&](int Sw) {
    dim3 grid(cdiv(L0.H, 16), L0.ndir, cdiv(Sw, stile));
    if ((size_t)grid.y * grid.z * kShards * kShardStride > (size_t)kCtlHalf) return false;
    return fits(fn, grid, NW * 64);
  };
  const int nwin = pick_windows(L0.S, stile, gfits);
  if (nwin == 0) return P;
  if (nwin > 1 && ((size_t)(L0.S / nwin) * L0.ndir * 4 * L0.H * sizeof(float)) % 128 != 0) return P;
  const dim3 grid(cdiv(L0.H, 16), L0.ndir, cdiv(L0.S / nwin, stile));
  P.kind = kRecBwdGeneric; P.fn = fn; P.cpw = bwd_generic_cpw(need); P.seq_tile = stile; P.units = 16; P.windows = nwin;
  snprintf(P.kernel, sizeof(P.kernel), "lstm_bwd_persistent_kernel<%d,%d,%s>", P.cpw, stile, L0.drop_mode ? "true" : "false");
  plan_resources(P, grid);
  return P;
}

bool lstm_bwd_persistent(hipStream_t st, const LstmLayerDev& L0, const float* dY, int lddy, float* DG, unsigned* cnt,
                         unsigned* err, int spin_limit, unsigned long long* trace) {
  const RecPlan P = lstm_bwd_plan(L0, false);
  if (P.kind == kRecNone) return false;
  const dim3 grid(P.grid[0], P.grid[1], P.grid[2]);
  const Role role{P.grid[0], P.grid[1], P.grid[2], L0.xcd_map};
  int chunk = P.chunk;
  if (P.kind == kRecBwdQ4) {
    LstmLayerDev L = L0;
    L.s_begin = 0; L.s_count = 0;
    EESEN_HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(unsigned) * grid.y * grid.z * kShards * kShardStride, st));
    plan_launch(st, P, grid, L, dY, lddy, DG, cnt, err, spin_limit, trace, role);
    return true;
  }
  if (P.kind == kRecBwdKsplitMux || P.kind == kRecBwdKsplit) {
    const size_t px_need = lstm_bwd_ksplit_px_floats(L0);
    unsigned long long* px = reinterpret_cast<unsigned long long*>(L0.PX);
    for (int w = 0; w < P.windows; ++w) {
      LstmLayerDev L = L0;
      if (P.kind == kRecBwdKsplitMux) { L.s_begin = 0; L.s_count = 0; }
      else { L.s_count = L0.S / P.windows; L.s_begin = w * L.s_count; }
      // counters: one set per (direction, 16-sequence tile) and K quarter -- the multiplexed launch covers every tile of the batch
      const size_t c1 = (size_t)L0.ndir * (P.kind == kRecBwdKsplitMux ? cdiv(L0.S, 16) : (int)grid.z) * 4 * kShards * kShardStride;
      EESEN_HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(unsigned) * c1, st));
      EESEN_HIP_CHECK(hipMemsetAsync(L.PX, 0, sizeof(float) * px_need, st));
      if (P.kind == kRecBwdKsplitMux) plan_launch(st, P, grid, L, dY, lddy, DG, px, cnt, err, spin_limit, role, chunk);
      else plan_launch(st, P, grid, L, dY, lddy, DG, px, cnt, err, spin_limit, role, chunk, trace);
    }
    return true;
  }
  for (int w = 0; w < P.windows; ++w) {
    LstmLayerDev L = L0;
    L.s_count = L0.S / P.windows;
    L.s_begin = w * L.s_count;
    EESEN_HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(unsigned) * (grid.y * grid.z * kShards * kShardStride), st));
    plan_launch(st, P, grid, L, dY, lddy, DG, cnt, err, spin_limit, trace, role, chunk);
  }
  return true;
}

}  // namespace eesen
