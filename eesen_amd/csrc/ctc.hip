// ctc.hip -- softmax and the CTC forward-backward over a padded utterance batch, for gfx950.
//
// Reference arithmetic: Ctc::EvalParallel /root/reference/src/net/ctc-loss.cc:101-194 and its three
// CUDA kernels src/gpucompute/cuda-kernels.cu:1367-1408 (alpha), :1482-1544 (beta, live branch
// :1527-1543), :1603-1627 (error), with the log-domain helpers of src/gpucompute/ctc-utils.h:53-96
// (-1e30 sentinel compared with ==, ExpA clamps).  The reference launches one kernel per time step per
// sweep (2T launches, each preceded by 2-3 blocking H2D copies of the label arrays).
//
// Here the lattice sweep of one utterance is ONE wavefront that walks all T_s steps inside a single
// launch: lane l owns the PL consecutive lattice positions [l*PL, (l+1)*PL), so the j-1 / j-2 (alpha) and
// j+1 / j+2 (beta) neighbours are in-register except at the chunk edge, where two wave shuffles
// (__shfl_up / __shfl_down) fetch them -- no LDS, no barrier on the 2T-step dependency chain.  The alpha
// and beta sweeps of all S utterances run concurrently (2S workgroups).  Lattices of more than 256 positions
// -- up to 4096; the reference takes any label length, ctc-loss.cc:116-129: character targets on a 35 s
// utterance -- are walked by NW wavefronts of one workgroup: wave w owns positions [w*64*PL, (w+1)*64*PL), and
// the two values that cross a wave boundary per step travel through a double-buffered LDS slot behind ONE
// workgroup barrier per step (round 5; measured faster than more positions per lane on one wave, see
// ctc_sweep_waves).  The arithmetic per position is the same whatever (PL, NW) covers it: bit-identical.  The next step's log-probability
// gather is issued one step ahead so its latency sits under the current step's log-add-exp.  alpha/beta
// rows are written utterance-major [S][T][64*PL] so every store is one coalesced line-aligned row.
// The per-frame gradient is then a bulk pass (one wavefront per frame) that stages alpha+beta in LDS, reduces the
// blank's ~L'/2 positions across the wave and lets lane k >= 1 fold class k's few lattice positions (precomputed per
// utterance), emitting d(-ln p)/d(logits) directly.
#include "kernels.h"

namespace eesen {
namespace {

// ---- src/gpucompute/ctc-utils.h:33-96, fp32 instantiation ------------------------------------------
constexpr float kLogZero = -1e30f, kLogInf = 1e30f, kExpLimit = 88.722839f, kFltMax = 3.4028235e+038f;
__device__ __forceinline__ float AddAB(float a, float b) { return (a == kLogZero || b == kLogZero) ? kLogZero : a + b; }
__device__ __forceinline__ float SubAB(float a, float b) {
  if (a == kLogZero) return kLogZero;
  if (b == kLogZero) return kLogInf;
  return a - b;
}
__device__ __forceinline__ float ExpA(float a) {
  if (a <= kLogZero) return 0.f;
  if (a >= kExpLimit) return kFltMax;
  return expf(a);
}
[[maybe_unused]] __device__ __forceinline__ float LogAPlusB(float a, float b) {   // (the reference helper the fast form below is held against)
  if (b < a) return AddAB(a, logf(1.f + ExpA(SubAB(b, a))));
  return AddAB(b, logf(1.f + ExpA(SubAB(a, b))));
}

// The same helper on the hardware transcendental units (v_exp_f32 / v_log_f32, ~1 ulp) for the lattice sweep, whose T-step
// dependency chain is made of exactly these two operations (8 exp + 8 log per lane and step at PL = 4).  Their error
// (< 4e-7 absolute on log(1 + e^d) in (0, ln 2]) is below the fp32 rounding of the alpha values it is added to.
// Branch-free form: with the FINITE sentinel -1e30 plain fp32 arithmetic already reproduces every special case of
// AddAB / SubAB / ExpA above -- x + (-1e30) == -1e30 exactly for |x| < 1e22 (ulp(1e30) = 7.6e22), exp(-1e30 - m) == 0,
// and logadd(-1e30, -1e30) = -1e30 + ln 2 == -1e30 -- so max + log(1 + exp(min - max)) equals LogAPlusB on every input
// the lattice can produce (log-probabilities are <= 0, so the exp never overflows), in 6 instructions instead of ~40.
__device__ __forceinline__ float LogAPlusB_fast(float a, float b) {
  const float m = fmaxf(a, b), n = fminf(a, b);
  return m + __logf(1.f + __expf(n - m));
}
__device__ __forceinline__ float AddAB_fast(float a, float b) { return a + b; }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- Softmax::PropagateFnc (softmax-layer.h:44-47): one wavefront per row ---------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                                           int ldy, int rows, int K) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + (size_t)r * ldx;
  float* yr = y + (size_t)r * ldy;
  float mx = -3.4e38f;
  for (int k = lane; k < K; k += 64) mx = fmaxf(mx, xr[k]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int k = lane; k < K; k += 64) sum += expf(xr[k] - mx);
  sum = wave_sum(sum);
  for (int k = lane; k < K; k += 64) yr[k] = expf(xr[k] - mx) / sum;
}

__global__ __launch_bounds__(256) void log_rows_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out,
                                                       int ldo, int rows, int K) {
  const size_t total = (size_t)rows * K;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / K, k = i % K;
    out[r * ldo + k] = logf(in[r * ldi + k]);
  }
}

template <int PL>
__device__ __forceinline__ void store_row(float* __restrict__ dst, const float (&v)[PL]) {
  if constexpr (PL % 4 == 0) {
#pragma unroll
    for (int i = 0; i < PL; i += 4) *reinterpret_cast<float4*>(dst + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
  } else if constexpr (PL == 2) {
    *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
  } else {
#pragma unroll
    for (int i = 0; i < PL; ++i) dst[i] = v[i];
  }
}

// ---- alpha / beta sweeps: grid (S, 2), one wavefront each -------------------------------------------
template <int PL, int NW, bool is_beta>
__device__ __forceinline__ void ctc_sweep(const float* __restrict__ logp, int ld, int T, int S, const int* __restrict__ labx,
                                          const int* __restrict__ lens, const int* __restrict__ lablens,
                                          float* __restrict__ alpha, float* __restrict__ beta, float* __restrict__ pzx,
                                          float* last, float (*edge)[NW][2]) {
  static_assert(NW == 1 || PL >= 2, "a wave boundary hands over TWO positions of the neighbouring lane");
  constexpr int Lpad = 64 * PL * NW;
  const int s = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int len = lens[s], ll = lablens[s];
  const int* lab = labx + (size_t)s * Lpad;
  const int j0 = (int)threadIdx.x * PL;
  // the lane whose chunk ends at a wave boundary publishes its two edge positions after every step; the lane on the other side
  // of the boundary takes them instead of the (wrapped) shuffle result
  const bool pub = NW > 1 && (is_beta ? (lane == 0 && wave > 0) : (lane == 63 && wave < NW - 1));
  const bool take = NW > 1 && (is_beta ? (lane == 63 && wave < NW - 1) : (lane == 0 && wave > 0));
  const int src = is_beta ? wave + 1 : wave - 1;
  auto publish = [&](int k, const float (&v)[PL]) {   // after step k (row 0: k = 0); ONE barrier per step, two slots by step parity
    if constexpr (NW > 1) {
      if (pub) { edge[k & 1][wave][0] = is_beta ? v[0] : v[PL - 1]; edge[k & 1][wave][1] = is_beta ? v[PL >= 2 ? 1 : 0] : v[PL >= 2 ? PL - 2 : 0]; }
      __syncthreads();
    }
  };

  // Everything per-position that does not change over time is decided once: the class to gather (0 for the -1 padding: a
  // valid address whose value is never used), and which neighbours take part.  A neighbour that does not take part enters
  // the log-add as the sentinel -1e30, for which LogAPlusB_fast is the exact identity (see above), so ONE straight-line
  // expression covers the reference's four cases (:1380-1405 / :1495-1541) -- no divergent branches on the T-step chain and
  // no branches around the gathers (which would also stop hipcc from counting them in flight).
  int gi[PL];
  bool valid[PL], use1[PL], use2[PL];
#pragma unroll
  for (int i = 0; i < PL; ++i) {
    const int j = j0 + i;
    const int c = lab[j];
    valid[i] = c >= 0;
    gi[i] = c >= 0 ? c : 0;
    if (!is_beta) {
      use1[i] = valid[i] && j >= 1;                                              // :1397-1403: alpha[j-1]
      use2[i] = valid[i] && j > 1 && (j & 1) && lab[j - 2] != c;                // :1396
    } else {
      use1[i] = valid[i] && j <= ll - 2;                                         // :1533-1539: beta[j+1]
      use2[i] = valid[i] && j < ll - 2 && (j & 1) && lab[j + 2] != c;           // :1532
    }
  }
  if (len <= 0) {   // (uniform over the workgroup: len belongs to the lattice)
    if (!is_beta && threadIdx.x == 0) pzx[s] = kLogZero;
    return;
  }
  float* out = (is_beta ? beta : alpha) + (size_t)s * T * Lpad + j0;
  float cur[PL];
  // The gathers of the log-probabilities run DEPTH steps ahead of their use (a ring of DEPTH register sets): one step
  // of look-ahead leaves the chain bound by the HBM latency of the gather (measured 1.66 us per step).
  constexpr int DEPTH = 4;
  const int t_first = is_beta ? len - 1 : 0, dt = is_beta ? -1 : 1;   // sweep direction
  auto row = [&](int k) {  // log-prob row of the k-th step of this sweep (clamped to a valid frame past the end)
    const int t = t_first + dt * k;
    return logp + (size_t)((k < len ? t : t_first) * S + s) * ld;
  };
  {  // first row (:1391-1393 / :1527-1529)
    const float* lp = row(0);
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      const int j = j0 + i;
      const bool on = valid[i] && (is_beta ? j > ll - 3 : j < 2);
      const float v = lp[gi[i]];
      cur[i] = on ? v : kLogZero;
    }
    store_row<PL>(out + (size_t)t_first * Lpad, cur);
    publish(0, cur);
  }
  float P[DEPTH][PL];
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) {
    const float* lr = row(1 + u);
#pragma unroll
    for (int i = 0; i < PL; ++i) P[u][i] = lr[gi[i]];
  }
  for (int base = 1; base < len; base += DEPTH) {
    float Q[DEPTH][PL];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {  // gathers for steps base+DEPTH .. base+2*DEPTH-1
      const float* lr = row(base + DEPTH + u);
#pragma unroll
      for (int i = 0; i < PL; ++i) Q[u][i] = lr[gi[i]];
    }
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      const int k = base + u;
      if (k < len) {  // wave-uniform
        // the two neighbours beyond this lane's chunk: alpha looks down (j-1, j-2), beta up (j+1, j+2)
        float e1 = is_beta ? __shfl_down(cur[0], 1) : __shfl_up(cur[PL - 1], 1);
        float e2 = is_beta ? (PL >= 2 ? __shfl_down(cur[PL >= 2 ? 1 : 0], 1) : __shfl_down(cur[0], 2))
                           : (PL >= 2 ? __shfl_up(cur[PL >= 2 ? PL - 2 : 0], 1) : __shfl_up(cur[0], 2));
        if constexpr (NW > 1) {
          if (take) { e1 = edge[(k - 1) & 1][src][0]; e2 = edge[(k - 1) & 1][src][1]; }
        }
        float nxt[PL];
#pragma unroll
        for (int i = 0; i < PL; ++i) {
          const int i1 = is_beta ? i + 1 : i - 1, i2 = is_beta ? i + 2 : i - 2;
          const bool in1 = i1 >= 0 && i1 < PL, in2 = i2 >= 0 && i2 < PL;
          float n1 = in1 ? cur[in1 ? i1 : 0] : e1;
          // the second neighbour is either in this lane, or the neighbour lane's edge (e1) / next-to-edge (e2) element
          float n2 = in2 ? cur[in2 ? i2 : 0] : ((is_beta ? i2 == PL : i2 == -1) ? e1 : e2);
          n1 = use1[i] ? n1 : kLogZero;
          n2 = use2[i] ? n2 : kLogZero;
          float acc = LogAPlusB_fast(n1, cur[i]);                               // :1397,:1399,:1403 / :1533,:1535,:1539
          if (PL % 2 != 0 || (i & 1)) acc = LogAPlusB_fast(n2, acc);           // :1400 / :1536 (odd positions only: labels)
          const float v = AddAB_fast(P[u][i], acc);                             // :1397-1405 / :1533-1541
          nxt[i] = valid[i] ? v : kLogZero;                                      // :1380-1383 / :1495-1498
        }
#pragma unroll
        for (int i = 0; i < PL; ++i) cur[i] = nxt[i];
        store_row<PL>(out + (size_t)(t_first + dt * k) * Lpad, cur);
        publish(k, cur);
      }
    }
#pragma unroll
    for (int u = 0; u < DEPTH; ++u)
#pragma unroll
      for (int i = 0; i < PL; ++i) P[u][i] = Q[u][i];
  }
  if (!is_beta) {
    // ln p(z|x) = logadd(alpha[T_s-1][L'_s-1], alpha[T_s-1][L'_s-2])  (ctc-loss.cc:147-153)
#pragma unroll
    for (int i = 0; i < PL; ++i) last[j0 + i] = cur[i];
    __syncthreads();
    if (threadIdx.x == 0) {
      const float tmp1 = last[ll - 1], tmp2 = ll >= 2 ? last[ll - 2] : kLogZero;
      pzx[s] = tmp1 + logf(1.f + ExpA(tmp2 - tmp1));
    }
  }
}

template <int PL, int NW>
__global__ __launch_bounds__(64 * NW) void ctc_alpha_beta_kernel(const float* __restrict__ logp, int ld, int T, int S,
                                                                 const int* __restrict__ labx, const int* __restrict__ lens,
                                                                 const int* __restrict__ lablens, float* __restrict__ alpha,
                                                                 float* __restrict__ beta, float* __restrict__ pzx) {
  __shared__ float last[64 * PL * NW];
  __shared__ float edge[2][NW][2];
  // the sweep direction is a template parameter so that every register-array index in the step is a compile-time constant
  if (blockIdx.y == 1) ctc_sweep<PL, NW, true>(logp, ld, T, S, labx, lens, lablens, alpha, beta, pzx, last, edge);
  else ctc_sweep<PL, NW, false>(logp, ld, T, S, labx, lens, lablens, alpha, beta, pzx, last, edge);
}

// ---- error kernel (:1603-1627) + softmax Jacobian (ctc-loss.cc:160-168): one wavefront per 8 frames of an utterance -----
// gamma_k = log sum_{j: l'_j = k} exp(alpha_j + beta_j).  The reference folds every class serially (thread (frame, k) loops
// over ALL L' positions).  The blank owns every second lattice position (~(L'+1)/2 = 101 at cfg2), every other class two or
// three: a class-per-lane fold is one lane grinding through 101 dependent log-add-exps while 63 wait (round 1: 0.19 ms for
// 32 000 frames, 5 % of the HBM roofline).  Here the work is POSITIONS-major: lane l owns positions l, l + 64, ...; the blank
// (even positions) is reduced across the wave -- max, then sum of exp(v - max), two shuffle trees --, the labels (odd
// positions) through two LDS atomics per position into their class's slot: an unsigned min on the bit pattern (all values are
// <= 0, so the smallest pattern is the largest value) and a float add of exp(v - max_k).  One wave owns a frame and its LDS
// slots, and same-address lanes of one LDS atomic are served in lane order, so the sums are reproducible run to run.
// Only the L'_s positions the utterance has are read (the lattice rows are padded to 64 * PL for the sweep's stores).
// A wave owns `frames` CONSECUTIVE frames of ONE utterance (round 3; it used to own one frame): the utterance's label table,
// lengths and ln p are read once and stay in registers, the lattice rows of the next frame are in flight while this one is
// folded (rows of one utterance are adjacent in alpha / beta), and the per-wave start-up -- three dependent scalar loads before
// the first row could even be addressed -- is paid once per `frames` frames.
// A lane owns PAIRS of lattice positions (2p, 2p + 1) = (a blank, a label): with positions dealt out one per lane the even
// lanes only ever saw blanks and the odd lanes only labels, so every branch of the fold ran with half the wave masked off.
// The per-position exponentials exp(v - max) (arguments <= 0) are v_exp_f32 (the sweep's argument: ~2 ulp, far below the fp32
// rounding of the sum they enter); the per-class logarithms and the final exponential keep the library functions.
// Measured (MI355X): the pass was instruction-bound, not latency-bound -- ~6500 issue cycles per frame at L' = 601.
template <int MAXP>   // lattice positions per lane: L' <= 64 * MAXP (MAXP even)
__global__ __launch_bounds__(256) void ctc_error_diff_kernel(const float* __restrict__ probs, int ld, int T, int S, int K,
                                                             int Lpad, const int* __restrict__ lens,
                                                             const int* __restrict__ lablens,
                                                             const int* __restrict__ labx,
                                                             const float* __restrict__ alpha,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ pzx, float* __restrict__ diff,
                                                             int ldd, int frames) {
  static_assert(MAXP % 2 == 0, "pairs of positions");
  constexpr int NP = MAXP / 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nchunk = (T + frames - 1) / frames;
  const int wid = blockIdx.x * (blockDim.x >> 6) + w;       // wave -> (utterance, chunk of frames), utterance fastest; 4, 2 or 1 waves per workgroup (the launcher: by K)
  if (wid >= nchunk * S) return;
  const int s = wid % S, t0 = (wid / S) * frames;
  const int t1 = min(T, t0 + frames);
  const int len = lens[s], Ls = lablens[s];
  const float pz = pzx[s];
  unsigned* mxb = reinterpret_cast<unsigned*>(smem) + (size_t)w * 2 * K;   // per class: bit pattern of the maximum ...
  float* sm = smem + (size_t)w * 2 * K + K;                             // ... and sum of exp(v - max); later e_k
  int cls[NP];   // class of the pair's label position (odd position 2p + 1); -1: beyond the utterance's lattice
  {
    const int* lx = labx + (size_t)s * Lpad;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int j1 = 2 * (lane + 64 * i) + 1;
      cls[i] = j1 < Ls ? lx[j1] : -1;
    }
  }
  float2 an[NP] = {}, bn[NP] = {};   // the next frame's lattice rows, in flight
  // straight-line loads (a predicated load is a branch per load): positions beyond the lattice read the row's last pair and are
  // masked afterwards; frames beyond the utterance (uniform per wave) are not read at all
  const int half = Lpad / 2;
  auto fetch = [&](int t) {
    if (t >= len) return;
    const float2* ar = reinterpret_cast<const float2*>(alpha + ((size_t)s * T + t) * Lpad);
    const float2* br = reinterpret_cast<const float2*>(beta + ((size_t)s * T + t) * Lpad);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = min(lane + 64 * i, half - 1);
      an[i] = ar[p];
      bn[i] = br[p];
    }
  };
  fetch(t0);
  for (int t = t0; t < t1; ++t) {
    const int r = t * S + s;
    float* drow = diff + (size_t)r * ldd;
    // alpha + beta of the pair's blank / label position.  Plain sums: x + (-1e30) == -1e30 exactly (ulp(1e30) = 7.6e22) and
    // (-1e30) + (-1e30) < -1e30, and everything below only asks "> -1e30" -- AddAB's cases without its branches.
    float vb[NP], vl[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      vb[i] = 2 * (lane + 64 * i) < Ls ? an[i].x + bn[i].x : kLogZero;
      vl[i] = cls[i] >= 0 ? an[i].y + bn[i].y : kLogZero;
    }
    if (t + 1 < t1) fetch(t + 1);
    if (t >= len) {  // ctc_err_ stays zero there (:1613), and so does diff
      for (int k = lane; k < K; k += 64) drow[k] = 0.f;
      continue;
    }
    const float* yr = probs + (size_t)r * ld;
    for (int k = lane; k < K; k += 64) { mxb[k] = 0xffffffffu; sm[k] = 0.f; }
    float bmax = kLogZero;
    __builtin_amdgcn_wave_barrier();
    // (branches on purpose: beyond the lattice whole instructions are skipped, and a branch-free fold through dump slots
    // measured 1.5x slower -- LDS atomics of all 64 lanes instead of the live ones)
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      bmax = fmaxf(bmax, vb[i]);
      // all values are <= 0, so the smallest bit pattern is the largest value.  (A target sequence that names class 0 itself
      // puts a "blank" on an odd position: it folds with the blanks, as the reference's per-class loop would have it.)
      if (cls[i] == 0) bmax = fmaxf(bmax, vl[i]);
      else if (vl[i] > kLogZero) atomicMin(&mxb[cls[i]], __builtin_bit_cast(unsigned, vl[i]));
    }
    __builtin_amdgcn_wave_barrier();  // same-wave LDS ops execute in order; this only pins the compiler's schedule
    bmax = wave_max(bmax);
    float bsum = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (vb[i] > kLogZero) bsum += __expf(vb[i] - bmax);
      if (vl[i] > kLogZero) {
        if (cls[i] == 0) bsum += __expf(vl[i] - bmax);
        else atomicAdd(&sm[cls[i]], __expf(vl[i] - __builtin_bit_cast(float, mxb[cls[i]])));
      }
    }
    bsum = wave_sum(bsum);
    __builtin_amdgcn_wave_barrier();
    float rsum = 0.f;
    for (int k = lane; k < K; k += 64) {
      float err;
      // (hardware log / exp, ~1 ulp of the log2 / exp2 units: 2e-6 absolute on ln y at y = 1e-10, against the 1e-4 bar)
      if (k == 0) err = bmax > kLogZero ? bmax + __logf(bsum) : kLogZero;
      else err = mxb[k] != 0xffffffffu ? __builtin_bit_cast(float, mxb[k]) + __logf(sm[k]) : kLogZero;                                 // :1617-1624
      const float y = yr[k];
      const float x = SubAB(err, AddAB(pz, y == 0.f ? kLogZero : 2.f * __logf(y)));
      const float val = x <= kLogZero ? 0.f : x >= kExpLimit ? kFltMax : __expf(x);          // ExpA, :1625
      const float e_k = (-1.0f * val) * y;                                                  // :1626, ctc-loss.cc:160
      sm[k] = e_k;
      rsum += e_k;
    }
    rsum = wave_sum(rsum);                                                                   // ctc-loss.cc:162
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < K; k += 64) drow[k] = sm[k] - yr[k] * rsum;                       // :164-168
    __builtin_amdgcn_wave_barrier();
  }
}

// m = (apply_log ? log(m) : m) - prior_scale * log_prior[col]   (net-output-extract.cc:103-112: ApplyLog, then
// ClassPrior::SubtractOnLogpost = AddVecToRows(-prior_scale, log_priors), class-prior.cc:80-91)
__global__ __launch_bounds__(256) void log_sub_prior_kernel(float* __restrict__ m, int ld, int rows, int K, int apply_log,
                                                            const float* __restrict__ log_prior, float prior_scale) {
  const size_t total = (size_t)rows * K;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / K, k = i % K;
    float v = m[r * ld + k];
    if (apply_log) v = logf(v);
    if (log_prior) v += -prior_scale * log_prior[k];
    m[r * ld + k] = v;
  }
}

__global__ __launch_bounds__(256) void row_argmax_kernel(const float* __restrict__ m, int ld, int rows, int K,
                                                         int* __restrict__ ids) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* mr = m + (size_t)r * ld;
  float best = -1e21f;  // cuda-matrix.cc:1045
  int bi = -1;
  for (int k = lane; k < K; k += 64) {
    const float v = mr[k];
    if (best < v) { best = v; bi = k; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > best || (ov == best && oi >= 0 && (bi < 0 || oi < bi))) { best = ov; bi = oi; }
  }
  if (lane == 0) ids[r] = bi;
}

}  // namespace

void softmax_rows(hipStream_t st, const float* x, int ldx, float* y, int ldy, int rows, int K) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, ldx, y, ldy, rows, K);
  check_launch("softmax_rows");
}

void log_rows(hipStream_t st, const float* in, int ldi, float* out, int ldo, int rows, int K) {
  if (rows <= 0) return;
  const size_t total = (size_t)rows * K;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(log_rows_kernel, dim3(blocks), dim3(256), 0, st, in, ldi, out, ldo, rows, K);
  check_launch("log_rows");
}

// Waves per lattice for a padded row of Lpad positions (default: see the measurements in ctc_sweep_waves).  `waves` > 0 asks for that
// many instead where such an instantiation exists
// (EESEN_CTC_WAVES, read when a Ctc is created: tests hold the multi-wave kernels bit for bit against the one-wave kernel where
// both exist; tuning.h).
static bool ctc_sweep_has(int pl, int nw) {   // the instantiations of ctc_alpha_beta below
  if (nw == 1) return pl == 1 || pl == 2 || pl == 4 || pl == 8 || pl == 16;
  const int lpad = 64 * pl * nw;
  if (pl == 2 && nw == 2) return true;   // 256 positions as 2 x 2 (an A/B arm: measured against the one-wave default)
  return pl >= 2 && pl <= 16 && nw <= 16 && (pl & (pl - 1)) == 0 && (nw & (nw - 1)) == 0 && lpad >= 512 && lpad <= 4096 && (pl >= 4 || lpad <= 2048);
}
int ctc_sweep_waves(int Lpad, int waves) {
  if (waves > 0 && Lpad % (64 * waves) == 0 && ctc_sweep_has(Lpad / (64 * waves), waves)) return waves;
  // Measured (MI355X, profiles/r05_ctc_waves.json; us per lattice step): a lane's positions are worked through one after the other,
  // so above 256 positions FEWER positions per lane on MORE waves win although every step then crosses a workgroup barrier --
  // 512 positions: 0.62 (8 x 1 wave), 0.45 (4 x 2), 0.35 (2 x 4); 1024: 1.25 (16 x 1), 0.75, 0.50, 0.45 (2 x 8); 2048: 1.36 (16 x 2),
  // 0.83 (8 x 4), 0.72 (4 x 8).  Up to 256 positions (4 per lane: 0.33) one wave, no barrier.
  // (second collection, profiles/r05b_ctc_waves.json: 2048 as 2 x 16 0.67 against 0.72 as 4 x 8; 4096 as 4 x 16 1.18, 8 x 8 1.28, 16 x 4 1.55;
  // 256 as 2 x 2 0.323 against 0.336 on one wave: not worth a barrier on the headline's chain)
  return Lpad <= 256 ? 1 : Lpad == 512 ? 4 : Lpad == 1024 ? 8 : 16;
}

void ctc_alpha_beta(hipStream_t st, const float* logp, int ld, int T, int S, int Lpad, const int* labx, const int* lens,
                    const int* lablens, float* alpha, float* beta, float* pzx, int waves) {
  const int nw = ctc_sweep_waves(Lpad, waves);
  const int pl = Lpad / (64 * nw);
  dim3 grid(S, 2), block(64 * nw);
  bool launched = false;
#define EESEN_AB(PL, NW)                                                                                                 \
  if (!launched && pl == PL && nw == NW) {                                                                               \
    hipLaunchKernelGGL((ctc_alpha_beta_kernel<PL, NW>), grid, block, 0, st, logp, ld, T, S, labx, lens, lablens, alpha, \
                       beta, pzx);                                                                                       \
    launched = true;                                                                                                     \
  }
  EESEN_AB(1, 1) EESEN_AB(2, 1) EESEN_AB(4, 1) EESEN_AB(8, 1) EESEN_AB(16, 1)          // <= 1024 positions, one wave
  EESEN_AB(2, 2)                                                                       // 256 positions as 2 waves
  EESEN_AB(4, 2) EESEN_AB(2, 4)                                                        // 512 positions as 2 / 4 waves
  EESEN_AB(8, 2) EESEN_AB(4, 4) EESEN_AB(2, 8)                                         // 1024 positions as 2 / 4 / 8 waves
  EESEN_AB(16, 2) EESEN_AB(8, 4) EESEN_AB(4, 8) EESEN_AB(2, 16)                        // 2048 positions
  EESEN_AB(16, 4) EESEN_AB(8, 8) EESEN_AB(4, 16)                                       // 4096 positions
#undef EESEN_AB
  if (!launched) throw Error(EESEN_ERR_INVALID, "ctc: no lattice kernel for this padded label length / wave count (at most 4096 positions)");
  check_launch("ctc_alpha_beta");
}

void ctc_error_diff(hipStream_t st, const float* probs, int ld, int T, int S, int K, int Lpad, int Lmax, const int* lens,
                    const int* lablens, const int* labx, const float* alpha, const float* beta,
                    const float* pzx, float* diff, int ldd) {
  const int rows = T * S;
  if (rows <= 0) return;
  // Every wave stages 2 K floats (per class: the running maximum and the sum of exp(v - max)) in LDS and the waves of a workgroup do not
  // talk to each other: four waves per workgroup up to 5120 classes (160 KB), two up to 10240, one up to 20480 -- word-piece inventories;
  // the reference itself takes any K (ctc-loss.cc:116-129: a [T x K] matrix), beyond 20480 this pass would need a second sweep per frame.
  const int nwb = K <= 5120 ? 4 : K <= 10240 ? 2 : 1;
  const size_t smem = (size_t)nwb * 2 * K * sizeof(float);
  // frames per wave: 8 where that still leaves >= 16384 waves (64 per CU: measured, cfg5 0.50 -> 0.37 ms), one for small
  // minibatches (cfg2's 32 000 frames: 3 per wave measured slower than 1)
  const int frames = std::max(1, std::min(8, rows / 16384));
  if (smem > 64 * 1024) {  // word / BPE targets (K in the thousands): ask for more than the default 64 KB of dynamic LDS (160 KB per CU)
    EESEN_REQUIRE(smem <= 160 * 1024, EESEN_ERR_INVALID, "ctc: more than 20480 classes: the gradient pass keeps 2 K floats per wave in LDS (160 KB per workgroup)");
  }
  auto launch = [&](auto kern) {
    // (asked for on every such launch: the grant is per device and per kernel, a process-wide "already granted" table was neither
    // -- a second device skipped the call and its launch failed; the call is a host-side table update, microseconds)
    if (smem > 64 * 1024)
      EESEN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kern, dim3(cdiv(cdiv(T, frames) * S, nwb)), dim3(64 * nwb), smem, st, probs, ld, T, S, K, Lpad, lens, lablens, labx, alpha,
                       beta, pzx, diff, ldd, frames);
  };
  EESEN_REQUIRE(Lpad <= 4096, EESEN_ERR_INVALID, "ctc: expanded label length above 4096");
  // positions per lane: the smallest even count that covers the longest lattice of the minibatch (Lmax = max 2 U_s + 1 <= Lpad)
  EESEN_REQUIRE(Lmax >= 1 && Lmax <= Lpad, EESEN_ERR_INVALID, "ctc: lattice length outside the padded row");
  if (Lmax <= 128) launch(ctc_error_diff_kernel<2>);
  else if (Lmax <= 256) launch(ctc_error_diff_kernel<4>);
  else if (Lmax <= 384) launch(ctc_error_diff_kernel<6>);
  else if (Lmax <= 512) launch(ctc_error_diff_kernel<8>);
  else if (Lmax <= 768) launch(ctc_error_diff_kernel<12>);
  else if (Lmax <= 1024) launch(ctc_error_diff_kernel<16>);
  else if (Lmax <= 1536) launch(ctc_error_diff_kernel<24>);
  else if (Lmax <= 2048) launch(ctc_error_diff_kernel<32>);
  else if (Lmax <= 3072) launch(ctc_error_diff_kernel<48>);
  else launch(ctc_error_diff_kernel<64>);
  check_launch("ctc_error_diff");
}

void log_sub_prior(hipStream_t st, float* m, int ld, int rows, int K, bool apply_log, const float* log_prior, float prior_scale) {
  if (rows <= 0) return;
  const size_t total = (size_t)rows * K;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(log_sub_prior_kernel, dim3(blocks), dim3(256), 0, st, m, ld, rows, K, apply_log ? 1 : 0, log_prior, prior_scale);
  check_launch("log_sub_prior");
}

void row_argmax(hipStream_t st, const float* m, int ld, int rows, int K, int* ids) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(row_argmax_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, m, ld, rows, K, ids);
  check_launch("row_argmax");
}

}  // namespace eesen
