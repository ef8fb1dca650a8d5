// gemm.hip -- fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), LDS-tiled, for gfx950.
//
// Serves every dense contraction of the hot path that is NOT inside the time recurrence
// (/root/reference/src/net/bilstm-parallel-layer.h:109,163 input->gates; :502,593 input gradient;
// :505-506,596-597 weight gradients; affine-trans-layer.h:165,171,182), i.e. what the reference sends
// to cublasSgemm through CuMatrixBase::AddMatMat (src/gpucompute/cuda-matrix.cc:604-639).
//
// Design (MI355X-first, not a cuBLAS call pattern):
//   * 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles
//     of 32x32, 64 accumulator registers), BK = 16, two LDS stages with register prefetch of the next
//     k-tile so global latency sits under 32 MFMAs (2048 matrix-pipe cycles per wave per k-tile).
//   * LDS holds both operands k-major ([k][m] / [k][n], row padded by 4 floats): every MFMA operand read
//     is a conflict-free ds_read_b32 of 32 consecutive floats per half-wave, whatever the storage order
//     of the operand in HBM (transposition happens on the LDS write, 2-way conflicts only).
//   * exact fp32: v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain (guide section 3), so results are
//     ordinary fp32 GEMM results -- no TF32-like truncation anywhere.
//   * split-K (deterministic two-pass: partial slabs + reduce kernel) for the weight-gradient shapes
//     whose M x N tile count cannot fill 256 CUs (e.g. 2048 x 512 with K = T*S = 32000).
#include <cstring>
#include <mutex>
#include <type_traits>

#include "kernels.h"

namespace eesen {
namespace {

// BK = 16: measured on MI355X, BK = 32 is -8 % on the k-contiguous shapes (97 vs 107 TF), +3 % on the tall-K transposed ones.
// (Raising the wave priority around the MFMA block, s_setprio 1-3: within noise.)
constexpr int BM = 128, BN = 128, BK = 16, LDP = 4;
constexpr int NLD = BM * BK / 4 / 256;        // float4 loads per thread per operand tile
constexpr int KQ_BITS = BK == 16 ? 2 : 3;     // log2(BK / 4): float4 per k-contiguous row
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmParams {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K;
  int lda, ldb, ldc;
  float alpha, beta;
  int splits;   // grid.y
  int k_chunk;  // K elements per split (multiple of BK)
  int tiles_n;
  int tiles_m;
  int gm;       // > 0: XCD-aware tile map with row groups of gm tiles (see tile_of); 0: row-major
  int bm, bn;   // output tile of the kernel flavour being launched (128 x 128, or 256 x 256 for the big split kernel)
  int nprod;    // split kernel: 6 = fp32-class (hi/mid/lo cross products), 1 = bf16 x bf16 only (operands rounded to bf16)
  // two-plane fp16 kernel: device words holding upper bounds of the operands' magnitudes (they set the power-of-two scales):
  // a_vec = 1: one word per M index (row of op(A)), else one word for the whole operand; b_vec: the same per N index
  const float* amax_a;
  const float* amax_b;
  int a_vec, b_vec;
  GemmGate gate;  // gate.cnt == nullptr: ordinary GEMM
};

// blockIdx.x -> output tile.  The dispatcher places block b on XCD b % 8 (observed; a speed hint only) and every XCD has
// its own L2, so with a plain row-major map the tiles_n blocks that share one A row-slab are spread over all eight L2s and
// each of them pulls the slab through the fabric: measured 4.8x the algorithmic bytes on the input->gates GEMM
// (profiles/r01e_pmc_fetch_write.md).  Here the tile grid is cut into row groups of `gm` tile-rows; group g belongs to XCD
// g % 8 (so all XCDs work on early rows first -- the gated GEMM visits rows in completion order), and inside a group the
// blocks of one XCD walk column-major (gm rows down, then the next tile column): the ~64 tiles resident on an XCD at any
// time form a gm x (64/gm) patch whose A slabs are shared by 64/gm and whose B slabs by gm concurrent tiles.
// Blocks beyond the ragged edge of the last group (or of the last round of groups) exit at once.
__device__ __forceinline__ bool tile_of(const GemmParams& p, int& tm, int& tn) {
  if (p.gm <= 0) {
    tm = blockIdx.x / p.tiles_n;
    tn = blockIdx.x % p.tiles_n;
    return true;
  }
  const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int gt = p.gm * p.tiles_n;
  const int g = (j / gt) * 8 + x, r = j % gt;
  const int first = g * p.gm;
  if (first >= p.tiles_m) return false;
  const int gs = min(p.gm, p.tiles_m - first);
  if (r >= gs * p.tiles_n) return false;
  tm = first + r % gs;
  tn = r / gs;
  return true;
}

// Load one [rows x BK] operand tile (rows = BM or BN) from HBM into registers: NLD float4 per thread.
// KC = true : operand stored [R x K], k contiguous.  float4 f -> (r = f / (BK/4), kq = f % (BK/4))
// KC = false: operand stored [K x R], r contiguous.  float4 f -> (k = f >> 5, rq = f & 31)
// GUARD = false: the tile lies completely inside the operand -- plain unconditional float4 loads, no branches (hipcc
// otherwise branches around every guarded load and waits for each one separately).
template <bool KC, bool GUARD>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int R, int r0, int k0, int kend,
                                          int tid, float4 (&v)[NLD]) {
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int f = tid + i * 256;
    if (!GUARD) {
      const size_t off = KC ? (size_t)(r0 + (f >> KQ_BITS)) * ld + k0 + ((f & ((1 << KQ_BITS) - 1)) << 2)
                            : (size_t)(k0 + (f >> 5)) * ld + r0 + ((f & 31) << 2);
      v[i] = *reinterpret_cast<const float4*>(P + off);
      continue;
    }
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KC) {
      const int r = r0 + (f >> KQ_BITS), k = k0 + ((f & ((1 << KQ_BITS) - 1)) << 2);
      if (r < R) {
        const float* src = P + (size_t)r * ld + k;
        if (k + 3 < kend) {
          x = *reinterpret_cast<const float4*>(src);
        } else {
          if (k + 0 < kend) x.x = src[0];
          if (k + 1 < kend) x.y = src[1];
          if (k + 2 < kend) x.z = src[2];
        }
      }
    } else {
      const int k = k0 + (f >> 5), r = r0 + ((f & 31) << 2);
      if (k < kend) {
        const float* src = P + (size_t)k * ld + r;
        if (r + 3 < R) {
          x = *reinterpret_cast<const float4*>(src);
        } else {
          if (r + 0 < R) x.x = src[0];
          if (r + 1 < R) x.y = src[1];
          if (r + 2 < R) x.z = src[2];
        }
      }
    }
    v[i] = x;
  }
}

template <bool KC>
__device__ __forceinline__ void store_tile(float (*T)[BM + LDP], int tid, const float4 (&v)[NLD]) {
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int f = tid + i * 256;
    if (KC) {
      const int r = f >> KQ_BITS, k = (f & ((1 << KQ_BITS) - 1)) << 2;
      T[k + 0][r] = v[i].x;
      T[k + 1][r] = v[i].y;
      T[k + 2][r] = v[i].z;
      T[k + 3][r] = v[i].w;
    } else {
      const int k = f >> 5, r = (f & 31) << 2;
      *reinterpret_cast<float4*>(&T[k][r]) = v[i];
    }
  }
}

// C/D map of the 32x32 MFMA (all input types): col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// row_w / col_w: offset of this wave's BMW x BNW blocks inside the workgroup's tile.
// SCALED (two-plane fp16 kernel): accumulator (row, col) holds 2^(sa[row] + sb[col]) times the products; inv[r] = 2^-sa of the tile's
// row r, inv[TM + c] = 2^-sb of its column c (exact powers of two, in LDS).
template <int BMW, int BNW, bool SCALED = false>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, const f32x16 (&acc)[BMW][BNW], int m0, int n0, int split, int row_w,
                                              int col_w, int lr, int lk, const float* inv = nullptr, int tm = 0) {
  float* C = p.C;
  size_t ldc = p.ldc;
  const bool partial = p.splits > 1;
  if (partial) {  // raw partial sums into slab `split` of the workspace, dense [M x N]
    C = p.C + (size_t)split * p.M * p.N;
    ldc = p.N;
  }
#pragma unroll
  for (int mi = 0; mi < BMW; ++mi)
#pragma unroll
    for (int ni = 0; ni < BNW; ++ni) {
      const int col = n0 + col_w + ni * 32 + lr;
      if (col >= p.N) continue;
      const float bv = (!partial && p.bias) ? p.bias[col] : 0.f;
      const float ub = SCALED ? inv[tm + col_w + ni * 32 + lr] : 1.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = row_w + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int row = m0 + rl;
        if (row >= p.M) continue;
        float* dst = C + (size_t)row * ldc + col;
        const float a = SCALED ? acc[mi][ni][r] * inv[rl] * ub : acc[mi][ni][r];
        if (partial) {
          *dst = a;
        } else {
          float v = p.alpha * a + bv;
          if (p.beta != 0.f) v += p.beta * *dst;
          *dst = v;
        }
      }
    }
}

// Which output tile this workgroup computes, and -- for the gated variant -- the wait until the producing recurrence has
// passed the tile's frames.  Returns false when the block has nothing to do (ragged edge of the XCD map, or the bounded
// spin gave up and raised the error word).
template <bool GATED>
__device__ __forceinline__ bool gemm_prologue(const GemmParams& p, int& m0, int& n0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int tm, tn;
  if (!tile_of(p, tm, tn)) return false;
  if (GATED) {  // middle-out over time: row tiles in the order in which a bidirectional layer completes their frames
    const int tiles_m = p.tiles_m, mid = tiles_m / 2;
    // i = 0, 1, 2, 3, ... -> mid, mid-1, mid+1, mid-2, ...; when one side runs out the other side continues
    const int i = tm, lo_cnt = mid, hi_cnt = tiles_m - mid;   // tiles below mid / at-or-above mid
    const int pairs = min(lo_cnt, hi_cnt);
    if (i < 2 * pairs) tm = (i & 1) ? mid - (i + 1) / 2 : mid + i / 2;
    else tm = hi_cnt > lo_cnt ? mid + (i - pairs) : mid - 1 - (i - pairs);
  }
  m0 = tm * p.bm; n0 = tn * p.bn;
  if (GATED) {
    __shared__ int s_go;
    const GemmGate& g = p.gate;
    const int t_lo = m0 / g.S, t_hi = min(p.M - 1, m0 + p.bm - 1) / g.S;
    if (wave == 0) {
      const int groups = g.ndir * g.nz;
      const bool mine = lane < groups * kShards;
      const int grp = lane / kShards, shard = lane % kShards, dir = grp / g.nz;
      const unsigned need = mine ? (unsigned)((g.nblk - shard + kShards - 1) / kShards) * (unsigned)(dir == 0 ? t_hi + 1 : g.T - t_lo) : 0u;
      const unsigned* c = g.cnt + (size_t)grp * kShards * kShardStride + shard * kShardStride;
      bool go = false;
      for (int spins = 0; spins < g.spin_limit; ++spins) {
        bool ok = true;
        if (mine) ok = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need;
        if (__all(ok)) { go = true; break; }
        if ((spins & 1023) == 1023 && __hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        __builtin_amdgcn_s_sleep(64);  // readiness changes once per recurrence step (~5 us): poll sparsely
        __builtin_amdgcn_s_sleep(64);
      }
      if (!go && lane == 0) __hip_atomic_store(g.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (lane == 0) s_go = go ? 1 : 0;
    }
    __syncthreads();
    if (!s_go) return false;
  }
  return true;
}

template <bool A_KC, bool B_KC, bool GUARD, bool GATED = false>
__device__ __forceinline__ void gemm_body(const GemmParams& p, float (*As)[BK][BM + LDP], float (*Bs)[BK][BN + LDP]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int m0, n0;
  if (!gemm_prologue<GATED>(p, m0, n0)) return;
  const int split = blockIdx.y;
  const int kbeg = split * p.k_chunk;
  const int kend = min(p.K, kbeg + p.k_chunk);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[NLD], rb[NLD];
  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk > 0) {
    load_tile<A_KC, GUARD>(p.A, p.lda, p.M, m0, kbeg, kend, tid, ra);
    load_tile<B_KC, GUARD>(p.B, p.ldb, p.N, n0, kbeg, kend, tid, rb);
    store_tile<A_KC>(As[0], tid, ra);
    store_tile<B_KC>(Bs[0], tid, rb);
  }
  __syncthreads();

  const int lr = lane & 31, lk = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      load_tile<A_KC, GUARD>(p.A, p.lda, p.M, m0, kbeg + (kt + 1) * BK, kend, tid, ra);
      load_tile<B_KC, GUARD>(p.B, p.ldb, p.N, n0, kbeg + (kt + 1) * BK, kend, tid, rb);
    }
    // keep the prefetch ABOVE the MFMA block: without the guards' branches hipcc sinks these loads to just before the
    // LDS stores, which serialises HBM latency with the matrix pipe (measured: 93 -> 66 TF on the tall-K shapes)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int kr = 2 * kk + lk;
      const float a0 = As[cur][kr][wm * 64 + lr];
      const float a1 = As[cur][kr][wm * 64 + 32 + lr];
      const float b0 = Bs[cur][kr][wn * 64 + lr];
      const float b1 = Bs[cur][kr][wn * 64 + 32 + lr];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      store_tile<A_KC>(As[cur ^ 1], tid, ra);
      store_tile<B_KC>(Bs[cur ^ 1], tid, rb);
    }
    __syncthreads();
  }

  gemm_epilogue<2, 2>(p, acc, m0, n0, split, wm * 64, wn * 64, lr, lk);
}

// GUARD = false is launched only when EVERY tile of the grid is interior and every split holds whole k-tiles (decided on
// the host): one body per kernel keeps the SGPR budget (two inlined bodies spill 48-80 SGPRs into the main loop).
template <bool A_KC, bool B_KC, bool GUARD>
__global__ __launch_bounds__(256, 2) void gemm_f32_mfma_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) float As[2][BK][BM + LDP];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN + LDP];
  gemm_body<A_KC, B_KC, GUARD>(p, As, Bs);
}

// The gated variant (see kernels.h): k-contiguous operands, unguarded tiles only (host checks the shape).
__global__ __launch_bounds__(256, 2) void gemm_f32_mfma_gated_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) float As[2][BK][BM + LDP];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN + LDP];
  gemm_body<true, true, false, true>(p, As, Bs);
}

// ------------------------------------------------------------------------------------------------------------------
// The same GEMM on the bf16 matrix pipe with fp32-class accuracy ("3-way split").
//
// gfx950 runs f32-input MFMA at the f32 VECTOR rate -- 1/16 of the bf16 MFMA rate (CDNA4 guide, section 3) -- and has no
// TF32-like form.  An fp32 value is EXACTLY the sum of three bf16 values: hi = top 8 significant bits (truncation),
// mid = top 8 bits of x - hi, lo = top 8 bits of x - hi - mid (each subtraction is exact in fp32; 8 + 8 + 8 = 24 bits).
// So a*b = sum of nine bf16 x bf16 products, each exact in fp32; the three smallest (mid*lo, lo*mid, lo*lo <= 2^-23 |ab|
// together) are dropped and the other six go through v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
//     a*b ~ hi*hi' + (hi*mid' + mid*hi') + (hi*lo' + lo*hi' + mid*mid'),   |error| <= 2^-23 |a*b|  (one fp32 rounding: 2^-24)
// Six MFMAs of K = 16 at 32 cycles against eight f32 MFMAs of K = 2 at 64 cycles for the same 32x32x16 block: 2.67x the
// f32 matrix rate at the accuracy of an fp32 GEMM (tests/test_gpu_gemm.py measures both against fp64).  The split costs
// ~5 VALU operations per element, once per element per workgroup, on the way into LDS; VALU and matrix pipe are separate.
//
// LDS: per operand and stage 3 planes (hi, mid, lo) x 2 k-halves x 128 rows x 8 bf16 (16 B): lane l of an MFMA reads row
// l & 31, k-half l >> 5 with ONE conflict-free ds_read_b128 (consecutive rows are consecutive 16-byte slots); the k-halves
// are 64 B apart modulo the bank window so that the ds_write_b64 of a k-contiguous loader do not collide either.
//
// Round 6: the same body on TWO fp16 planes ("half" mode, PL = 2).  With round-to-nearest at both levels an fp32 value a is
// hi + lo to within 2^-22 |a| (hi = fp16(a): 11 bits, |a - hi| <= 2^-11 |a|; lo = fp16(a - hi): |a - hi - lo| <= 2^-22 |a| -- the
// worst case; the residuals of two roundings to nearest rarely both sit at half an ulp: 2^-23 is the largest seen over 2 x 10^5
// random values, tests/test_half_planes_math.py), so
//     a*b ~ hi*hi' + (hi*lo' + lo*hi'),   |error| <= 3 * 2^-22 |a*b|   (dropped lo*lo' <= 2^-22 |ab|; two representation errors)
// -- the arithmetic of "3xTF32" (an 11-bit big part and an 11-bit small part, three products): a worst-case bound 12 times the
// three-bf16-plane split's 2^-23, random in sign from element to element, and MEASURED against fp64 over K-long dot products (where
// the fp32 accumulation's own rounding dominates either way) equal to the fp32 chain and to the six-product kernel on every shape
// of tests/test_gpu_gemm.py (profiles/r06_gemm_accuracy.json).  THREE products on v_mfma_f32_32x32x16_f16 instead of six on the
// bf16 form of the same rate, and two planes to split instead of three.  What fp16 lacks is exponent range (5 bits): every row of
// op(A) / column of op(B) is multiplied by the power of two that brings ITS largest magnitude into [2^14, 2^15) (half_scale: from
// device words holding those maxima -- amax_rows_cols below -- or a bound the caller knows, e.g. 1 for an LSTM output), and the
// epilogue multiplies the two powers back out -- both exact.  Where the residual a 2^s - hi falls below fp16's smallest normal
// number (2^-14) lo is an fp16 DENORMAL, which the gfx950 MFMA multiplies exactly (no flush; asserted on the device by
// tests/test_gpu_gemm.py): |a - (hi + lo) 2^-s| <= max(2^-22 |a|, 2^-39 x the row's largest) -- full precision for every element
// within 2^-15 of its row's largest, an absolute floor 2^-39 of it below that.
constexpr int kSplitMinW = 3;   // workgroups per CU of the 128 x 128 split kernel; measured: 171 -> 178 TF with two tiles of prefetch, 189 with three workgroups per CU
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned frag_t __attribute__((ext_vector_type(4)));   // eight 16-bit operand values of one lane, of either type

// Geometry of one split-kernel flavour.  TM x TN output tile, WGM x WGN waves, each wave (TM/WGM) x (TN/WGN) = BMW x BNW MFMA
// blocks of 32 x 32.  Two flavours are built:
//   128 x 128, 2 x 2 waves (256 threads, 24 MFMAs per wave and k-tile): <= 168 VGPRs, three workgroups per CU or ONE beside a
//     512-thread recurrence workgroup -- the side-stream (overlapped) GEMMs and everything small;
//   256 x 256, 2 x 4 waves (512 threads, 48 MFMAs per wave and k-tile): the same 16 floats to split per thread for twice the
//     MFMAs -- the kernel is bound by the SIMD issue port (split instructions), not by the matrix pipe -- one workgroup per
//     CU: the main-stream GEMMs of large shapes.
template <int TM_, int TN_, int WGM_, int WGN_, int PL_ = 3>
struct SplitGeo {
  static constexpr int TM = TM_, TN = TN_, WGM = WGM_, WGN = WGN_;
  static constexpr int PL = PL_;                                       // planes per operand: 3 bf16 (six products) or 2 fp16 (three)
  static constexpr int NPROD = PL == 3 ? 6 : 3;
  static constexpr int THREADS = WGM * WGN * 64;
  static constexpr int BMW = TM / WGM / 32, BNW = TN / WGN / 32;
  static constexpr int UA = TM * 4 / THREADS, UB = TN * 4 / THREADS;   // (row, k-quad) units per thread and k-tile, per operand
  static constexpr int HS_A = TM * 16 + 64, HS_B = TN * 16 + 64;       // bytes per k-half (rows x 16 B, + 16 banks)
  static constexpr int PS_A = 2 * HS_A, PS_B = 2 * HS_B;               // bytes per plane
  static constexpr int OP_A = PL * PS_A, OP_B = PL * PS_B;             // bytes per operand and stage
  static constexpr int STAGE = OP_A + OP_B;
  static constexpr int NMFMA = NPROD * BMW * BNW;                      // per wave and k-tile
  static_assert(UA == 2 && UB == 2, "the k loop below is written for two units per thread and operand");
};
using GeoSmall = SplitGeo<128, 128, 2, 2>;
using GeoBig = SplitGeo<256, 256, 2, 4>;
using GeoSmallH = SplitGeo<128, 128, 2, 2, 2>;
using GeoBigH = SplitGeo<256, 256, 2, 4, 2>;

// The power of two that brings an operand whose largest magnitude is *amax into [2^14, 2^15), and its inverse (both exact; an
// all-zero operand gets 2^126 and 2^-126).  Uniform: scalar loads.
__device__ __forceinline__ void half_scale(float amax, float& scale, float& inv) {
  const int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu);   // biased exponent of the bound
  const int s = min(max(127 + 14 + 127 - e, 1), 253);   // (253: the inverse stays a normal number)
  scale = __builtin_bit_cast(float, (unsigned)s << 23);
  inv = __builtin_bit_cast(float, (unsigned)(254 - s) << 23);
}

// (row, k-quad) units of an operand tile of ROWS rows x 16 k: each thread brings 2 units of 4 consecutive-k floats per k-tile.
//   KC (k contiguous in HBM):  unit f = tid + THREADS i -> row f >> 2, quad f & 3: one float4
//   !KC (row contiguous):      row tid % ROWS, quad tid / ROWS + (THREADS / ROWS) i: four dword loads, coalesced along the rows
template <bool KC, int THREADS, int ROWS>
__device__ __forceinline__ void unit_of(int tid, int i, int& row, int& q) {
  if (KC) { const int f = tid + i * THREADS; row = f >> 2; q = f & 3; }
  else { row = tid % ROWS; q = tid / ROWS + (THREADS / ROWS) * i; }
}

// Branch-free on purpose (see load_tile): a guarded load makes hipcc wait for every load separately, which serialises the
// HBM latency of the four loads of a k-tile.  Out-of-range rows / k are clamped to a valid address here; the store side zeroes
// them -- AFTER the MFMA block, so that nothing touches the loaded registers (and waits for them) before it.
template <bool KC, int THREADS, int ROWS>
__device__ __forceinline__ void split_load(const float* __restrict__ P, int ld, int R, int r0, int k0, int kend, int tid,
                                           float4 (&v)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row, q;
    unit_of<KC, THREADS, ROWS>(tid, i, row, q);
    const int r = r0 + row, k = k0 + (q << 2);
    if (KC) {
      // rows are padded to a multiple of 4 floats (ld % 4 == 0, checked on the host), so k + 3 < ld whenever k < K
      const int kc = min(k, max(kend - 1, 0) & ~3);
      v[i] = *reinterpret_cast<const float4*>(P + (size_t)min(r, R - 1) * ld + kc);
    } else {
      const float* col = P + min(r, R - 1);
      const int kl = max(kend - 1, 0);
      v[i].x = col[(size_t)min(k + 0, kl) * ld];
      v[i].y = col[(size_t)min(k + 1, kl) * ld];
      v[i].z = col[(size_t)min(k + 2, kl) * ld];
      v[i].w = col[(size_t)min(k + 3, kl) * ld];
    }
  }
}

// top 16 bits of two floats packed as two bf16 (low half = first): v_perm_b32
__device__ __forceinline__ unsigned pack_hi16(float a, float b) {
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float x) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u); }

// The split in three stages per unit, so that the k loop can place each stage in the shadow of MFMAs: A: guard, hi plane, first
// residual; B: mid plane, second residual; C: lo plane + the three LDS writes.
struct SplitUnit {
  float r[4];
  unsigned ph[2], pm[2];
};
// "these values exist HERE": pure arithmetic has no place of its own in the instruction stream -- the compiler emits it next
// to its first use, i.e. behind the MFMAs it is meant to hide under.  An empty volatile asm that takes the values as
// read-write operands is a use at this point, and it keeps its place among the scheduling fences.
#define EESEN_PIN6(a, b, c, d, e, f) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f))
// two floats -> two fp16, round to nearest even, packed (low half = first): v_cvt_pk_f16_f32
__device__ __forceinline__ unsigned pack_f16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ float f16_lo(unsigned p) { return (float)__builtin_bit_cast(f16x2, p)[0]; }
__device__ __forceinline__ float f16_hi(unsigned p) { return (float)__builtin_bit_cast(f16x2, p)[1]; }
// PL = 3: bf16 planes by truncation (scale unused).  PL = 2: fp16 planes by rounding, of the operand times `scale` (half_scale).
template <int PL, bool KC, bool GUARD, int THREADS, int ROWS>
__device__ __forceinline__ void split_stage_a(SplitUnit& u, const float4& v, int i, int tid, int R, int r0, int k0, int kend, float scale) {
  float x[4] = {v.x, v.y, v.z, v.w};
  if (GUARD) {
    int row, q;
    unit_of<KC, THREADS, ROWS>(tid, i, row, q);
    const bool ok = r0 + row < R;
    const int k = k0 + q * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = (ok && k + j < kend) ? x[j] : 0.f;
  }
  if (PL == 3) {
    u.ph[0] = pack_hi16(x[0], x[1]);
    u.ph[1] = pack_hi16(x[2], x[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) u.r[j] = x[j] - trunc_bf16(x[j]);   // exact
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] *= scale;                      // exact (a power of two)
    u.ph[0] = pack_f16(x[0], x[1]);
    u.ph[1] = pack_f16(x[2], x[3]);
    u.r[0] = x[0] - f16_lo(u.ph[0]);                                // exact: the residual of a rounding to 11 bits has <= 13
    u.r[1] = x[1] - f16_hi(u.ph[0]);
    u.r[2] = x[2] - f16_lo(u.ph[1]);
    u.r[3] = x[3] - f16_hi(u.ph[1]);
  }
}
template <int PL>
__device__ __forceinline__ void split_stage_b(SplitUnit& u) {
  if (PL == 3) {
    u.pm[0] = pack_hi16(u.r[0], u.r[1]);
    u.pm[1] = pack_hi16(u.r[2], u.r[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) u.r[j] = u.r[j] - trunc_bf16(u.r[j]);   // exact; the lo plane takes its top 16 bits
  } else {
    u.pm[0] = pack_f16(u.r[0], u.r[1]);
    u.pm[1] = pack_f16(u.r[2], u.r[3]);
  }
}
template <int PL, bool KC, int THREADS, int ROWS>
__device__ __forceinline__ void split_stage_c(const SplitUnit& u, unsigned char* base, int i, int tid) {
  constexpr int HS = ROWS * 16 + 64, PS = 2 * HS;
  int row, q;
  unit_of<KC, THREADS, ROWS>(tid, i, row, q);
  unsigned char* dst = base + (q >> 1) * HS + row * 16 + (q & 1) * 8;
  *reinterpret_cast<uint2*>(dst) = make_uint2(u.ph[0], u.ph[1]);
  *reinterpret_cast<uint2*>(dst + PS) = make_uint2(u.pm[0], u.pm[1]);
  if (PL == 3) *reinterpret_cast<uint2*>(dst + 2 * PS) = make_uint2(pack_hi16(u.r[0], u.r[1]), pack_hi16(u.r[2], u.r[3]));
}
template <int PL, bool KC, bool GUARD, int THREADS, int ROWS>
__device__ __forceinline__ void split_store(unsigned char* base, int tid, const float4 (&v)[2], int R, int r0, int k0, int kend, const float (&scale)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    SplitUnit u;
    split_stage_a<PL, KC, GUARD, THREADS, ROWS>(u, v[i], i, tid, R, r0, k0, kend, scale[i]);
    split_stage_b<PL>(u);
    split_stage_c<PL, KC, THREADS, ROWS>(u, base, i, tid);
  }
}
// one product of two planes on the matrix pipe: 32 x 32 x 16, fp32 accumulate
template <int PL>
__device__ __forceinline__ f32x16 mfma_planes(const frag_t& a, const frag_t& b, const f32x16& c) {
  if constexpr (PL == 3) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// nprod == 1 ("bf16 forward", BASELINE config 4): both operands rounded to nearest-even bf16, ONE MFMA product, fp32 accumulation.
__device__ __forceinline__ unsigned rne_bf16_bits(float x) {
  const unsigned b = __builtin_bit_cast(unsigned, x);
  return b + 0x7fffu + ((b >> 16) & 1u);   // top 16 bits = round-to-nearest-even bf16
}
template <bool KC, bool GUARD, int THREADS, int ROWS>
__device__ __forceinline__ void bf16_store(unsigned char* base, int tid, const float4 (&v)[2], int R, int r0, int k0, int kend) {
  constexpr int HS = ROWS * 16 + 64;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row, q;
    unit_of<KC, THREADS, ROWS>(tid, i, row, q);
    float x[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
    if (GUARD) {
      const bool ok = r0 + row < R;
      const int k = k0 + q * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = (ok && k + j < kend) ? x[j] : 0.f;
    }
    const unsigned b0 = rne_bf16_bits(x[0]), b1 = rne_bf16_bits(x[1]), b2 = rne_bf16_bits(x[2]), b3 = rne_bf16_bits(x[3]);
    *reinterpret_cast<uint2*>(base + (q >> 1) * HS + row * 16 + (q & 1) * 8) =
        make_uint2(__builtin_amdgcn_perm(b1, b0, 0x07060302u), __builtin_amdgcn_perm(b3, b2, 0x07060302u));
  }
}

template <class G, bool A_KC, bool B_KC, bool GUARD, bool GATED>
__device__ __forceinline__ void gemm_split_body(const GemmParams& p) {
  constexpr int TM = G::TM, TN = G::TN, TH = G::THREADS, BMW = G::BMW, BNW = G::BNW, PL = G::PL;
  // stage s: A planes at s * STAGE, B planes at s * STAGE + OP_A (indexed as an array, so the accesses stay ds_* ones)
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * G::STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / G::WGN, wn = wave % G::WGN;
  int m0, n0;
  if (!gemm_prologue<GATED>(p, m0, n0)) return;
  const int split = blockIdx.y;
  const int kbeg = split * p.k_chunk;
  const int kend = min(p.K, kbeg + p.k_chunk);
  constexpr int SBK = 16;

  f32x16 acc[BMW][BNW];
#pragma unroll
  for (int i = 0; i < BMW; ++i)
#pragma unroll
    for (int j = 0; j < BNW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (kend - kbeg + SBK - 1) / SBK;
  const int lr = lane & 31, lk = lane >> 5;
  const int a_off = lk * G::HS_A + (wm * (TM / G::WGM) + lr) * 16, b_off = G::OP_A + lk * G::HS_B + (wn * (TN / G::WGN) + lr) * 16;

  // tile t of the k loop: HBM -> registers / registers -> split -> LDS stage / LDS stage -> NMFMA MFMAs
  auto load = [&](int t, float4 (&ra)[2], float4 (&rb)[2]) {
    split_load<A_KC, TH, TM>(p.A, p.lda, p.M, m0, kbeg + t * SBK, kend, tid, ra);
    split_load<B_KC, TH, TN>(p.B, p.ldb, p.N, n0, kbeg + t * SBK, kend, tid, rb);
  };
  // two fp16 planes: the power-of-two scales of the rows this thread brings in (its two units per operand and k-tile sit in fixed
  // rows), from the operands' bounds -- one word per row of op(A) / column of op(B), or one for the whole operand
  float sa[2] = {1.f, 1.f}, sb[2] = {1.f, 1.f};
  if constexpr (PL == 2) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int row, q;
      float inv;
      unit_of<A_KC, TH, TM>(tid, i, row, q);
      half_scale(p.amax_a[p.a_vec ? min(m0 + row, p.M - 1) : 0], sa[i], inv);
      unit_of<B_KC, TH, TN>(tid, i, row, q);
      half_scale(p.amax_b[p.b_vec ? min(n0 + row, p.N - 1) : 0], sb[i], inv);
    }
  }
  auto store = [&](int t, const float4 (&ra)[2], const float4 (&rb)[2]) {
    const int st = (t & 1) * G::STAGE;
    split_store<PL, A_KC, GUARD, TH, TM>(&lds[st], tid, ra, p.M, m0, kbeg + t * SBK, kend, sa);
    split_store<PL, B_KC, GUARD, TH, TN>(&lds[st + G::OP_A], tid, rb, p.N, n0, kbeg + t * SBK, kend, sb);
  };
  // smallest terms first: hi*lo' + lo*hi' + mid*mid', then hi*mid' + mid*hi', then hi*hi' (two planes: hi*lo' + lo*hi', then hi*hi');
  // the output blocks in turn, so that dependent MFMAs are BMW * BNW issues apart
  constexpr int PA[6] = {0, PL == 3 ? 2 : 1, PL == 3 ? 1 : 0, 0, 1, 0}, PB[6] = {PL == 3 ? 2 : 1, 0, PL == 3 ? 1 : 0, 1, 0, 0};

  if (PL == 3 && p.nprod == 1) {   // bf16 x bf16 only: the hi plane, one product
    float4 ra[2], rb[2];
    if (nk > 0) {
      load(0, ra, rb);
      bf16_store<A_KC, GUARD, TH, TM>(&lds[0], tid, ra, p.M, m0, kbeg, kend);
      bf16_store<B_KC, GUARD, TH, TN>(&lds[G::OP_A], tid, rb, p.N, n0, kbeg, kend);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = (kt & 1) * G::STAGE, nxt = G::STAGE - cur;
      if (kt + 1 < nk) load(kt + 1, ra, rb);
      bf16x8 a[BMW], b[BNW];
#pragma unroll
      for (int i = 0; i < BMW; ++i) a[i] = *reinterpret_cast<const bf16x8*>(&lds[cur + a_off + i * 32 * 16]);
#pragma unroll
      for (int i = 0; i < BNW; ++i) b[i] = *reinterpret_cast<const bf16x8*>(&lds[cur + b_off + i * 32 * 16]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < BMW; ++mi)
#pragma unroll
        for (int ni = 0; ni < BNW; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nk) {
        bf16_store<A_KC, GUARD, TH, TM>(&lds[nxt], tid, ra, p.M, m0, kbeg + (kt + 1) * SBK, kend);
        bf16_store<B_KC, GUARD, TH, TN>(&lds[nxt + G::OP_A], tid, rb, p.N, n0, kbeg + (kt + 1) * SBK, kend);
      }
      __syncthreads();
    }
    gemm_epilogue<BMW, BNW>(p, acc, m0, n0, split, wm * (TM / G::WGM), wn * (TN / G::WGN), lr, lk);
    return;
  }

  // Two k-tiles of HBM prefetch in registers: tile t+2 is requested right after tile t+1 has left its registers for LDS, and is
  // consumed two MFMA blocks later.  In the steady state the split of tile t+1 -- ~90 VALU instructions and 12 ds_write_b64 per
  // thread -- is INTERLEAVED with the MFMAs of tile t inside the same wave: MFMA and VALU share the SIMD's issue port (one
  // 4-cycle slot each; a 32-cycle MFMA leaves room for ~5-6 others), and with the split after the MFMA block the port, not the
  // matrix pipe, was the limit (PMC: matrix pipe 55 % busy, VALU 40 %, summing to ~100 %).
  float4 ra0[2], rb0[2], ra1[2], rb1[2];
  // fragments of tile t: every plane of every block row / column of this wave
  auto frags = [&](int t, frag_t (&a)[BMW][PL], frag_t (&b)[BNW][PL]) {
    const int cur = (t & 1) * G::STAGE;
#pragma unroll
    for (int pl = 0; pl < PL; ++pl) {
#pragma unroll
      for (int i = 0; i < BMW; ++i) a[i][pl] = *reinterpret_cast<const frag_t*>(&lds[cur + pl * G::PS_A + a_off + i * 32 * 16]);
#pragma unroll
      for (int i = 0; i < BNW; ++i) b[i][pl] = *reinterpret_cast<const frag_t*>(&lds[cur + pl * G::PS_B + b_off + i * 32 * 16]);
    }
  };
  // steady state: MFMAs of tile t with the split of tile t + 1 in their shadow -- per unit  M..M [A]  M..M [B]  M..M [C], every
  // bracket ~10 VALU (40 issue cycles) behind NMFMA / 12 32-cycle MFMAs; scheduling fences pin the order
  auto fused = [&](int t, float4 (&ra)[2], float4 (&rb)[2]) {
    const int nxt = G::STAGE - (t & 1) * G::STAGE;
    const int k1 = kbeg + (t + 1) * SBK;
    frag_t a[BMW][PL], b[BNW][PL];
    frags(t, a, b);
    auto mm = [&](int idx) {   // MFMA number idx of the tile: product idx / (BMW * BNW), output block idx % (BMW * BNW)
      const int t6 = idx / (BMW * BNW), blk = idx % (BMW * BNW), mi = blk / BNW, ni = blk % BNW;
      acc[mi][ni] = mfma_planes<PL>(a[mi][PA[t6]], b[ni][PB[t6]], acc[mi][ni]);
    };
    constexpr int PER = G::NMFMA / 12;   // MFMAs next to every split stage
    SplitUnit su;
    // (Running every bracket BEFORE its MFMA group in the upper half of the waves, so that the two waves of a SIMD alternate
    // between multiplying and splitting, was measured on the eight-wave flavour: 204 vs 203 TF -- not kept.)
    auto group = [&](int g) {
#pragma unroll
      for (int j = 0; j < PER; ++j) mm(g * PER + j);
    };
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      __builtin_amdgcn_sched_barrier(0);
      group(3 * u);
      __builtin_amdgcn_sched_barrier(0);
      if (u < 2) split_stage_a<PL, A_KC, GUARD, TH, TM>(su, ra[u], u, tid, p.M, m0, k1, kend, sa[u]);
      else split_stage_a<PL, B_KC, GUARD, TH, TN>(su, rb[u - 2], u - 2, tid, p.N, n0, k1, kend, sb[u - 2]);
      EESEN_PIN6(su.r[0], su.r[1], su.r[2], su.r[3], su.ph[0], su.ph[1]);
      __builtin_amdgcn_sched_barrier(0);
      group(3 * u + 1);
      __builtin_amdgcn_sched_barrier(0);
      split_stage_b<PL>(su);
      EESEN_PIN6(su.r[0], su.r[1], su.r[2], su.r[3], su.pm[0], su.pm[1]);
      __builtin_amdgcn_sched_barrier(0);
      group(3 * u + 2);
      __builtin_amdgcn_sched_barrier(0);
      if (u < 2) split_stage_c<PL, A_KC, TH, TM>(su, &lds[nxt], u, tid);
      else split_stage_c<PL, B_KC, TH, TN>(su, &lds[nxt + G::OP_A], u - 2, tid);
    }
    __builtin_amdgcn_sched_barrier(0);
    load(t + 3, ra, rb);
    __syncthreads();
  };
  auto step = [&](int t, float4 (&ra)[2], float4 (&rb)[2], bool do_store, bool do_load) {
    {
      frag_t a[BMW][PL], b[BNW][PL];
      frags(t, a, b);
      __builtin_amdgcn_sched_barrier(0);  // global prefetches and fragment reads are issued before the MFMA block
#pragma unroll
      for (int t6 = 0; t6 < G::NPROD; ++t6)
#pragma unroll
        for (int mi = 0; mi < BMW; ++mi)
#pragma unroll
          for (int ni = 0; ni < BNW; ++ni)
            acc[mi][ni] = mfma_planes<PL>(a[mi][PA[t6]], b[ni][PB[t6]], acc[mi][ni]);
      __builtin_amdgcn_sched_barrier(0);  // ... and nothing that reads prefetched registers (a vmcnt wait) moves above them
    }
    if (do_store) store(t + 1, ra, rb);
    if (do_load) load(t + 3, ra, rb);
    __syncthreads();
  };
  if (nk > 0) { load(0, ra0, rb0); store(0, ra0, rb0); }
  if (nk > 1) load(1, ra0, rb0);
  if (nk > 2) load(2, ra1, rb1);
  __syncthreads();
  int kt = 0;
  for (; kt + 4 < nk; kt += 2) {   // steady state: tiles t + 1 and t + 3 exist
    fused(kt, ra0, rb0);
    fused(kt + 1, ra1, rb1);
  }
  for (; kt < nk; kt += 2) {
    step(kt, ra0, rb0, kt + 1 < nk, kt + 3 < nk);
    if (kt + 1 < nk) step(kt + 1, ra1, rb1, kt + 2 < nk, kt + 4 < nk);
  }
  if constexpr (PL == 2) {   // the inverse scales of the tile's rows and columns, through LDS (the k loop's last barrier has passed)
    static_assert(TH == TM + TN, "one thread per tile row / column");
    float* inv = reinterpret_cast<float*>(lds);
    float sc, iv;
    if (tid < TM) half_scale(p.amax_a[p.a_vec ? min(m0 + tid, p.M - 1) : 0], sc, iv);
    else half_scale(p.amax_b[p.b_vec ? min(n0 + tid - TM, p.N - 1) : 0], sc, iv);
    inv[tid] = iv;
    __syncthreads();
    gemm_epilogue<BMW, BNW, true>(p, acc, m0, n0, split, wm * (TM / G::WGM), wn * (TN / G::WGN), lr, lk, inv, TM);
  } else {
    gemm_epilogue<BMW, BNW>(p, acc, m0, n0, split, wm * (TM / G::WGM), wn * (TN / G::WGN), lr, lk);
  }
}

template <bool A_KC, bool B_KC, bool GUARD>
__global__ __launch_bounds__(256, kSplitMinW) void gemm_f32_split_bf16_kernel(GemmParams p) {
  gemm_split_body<GeoSmall, A_KC, B_KC, GUARD, false>(p);
}
__global__ __launch_bounds__(256, kSplitMinW) void gemm_f32_split_bf16_gated_kernel(GemmParams p) {
  gemm_split_body<GeoSmall, true, true, false, true>(p);
}
// 256 x 256 tiles, eight waves: unguarded shapes only (every tile interior, whole k-tiles; the host checks)
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(512, 2) void gemm_f32_split_bf16_big_kernel(GemmParams p) {
  gemm_split_body<GeoBig, A_KC, B_KC, false, false>(p);
}

// the same three kernels on two fp16 planes (three products)
template <bool A_KC, bool B_KC, bool GUARD>
__global__ __launch_bounds__(256, kSplitMinW) void gemm_f32_split_f16_kernel(GemmParams p) {
  gemm_split_body<GeoSmallH, A_KC, B_KC, GUARD, false>(p);
}
__global__ __launch_bounds__(256, kSplitMinW) void gemm_f32_split_f16_gated_kernel(GemmParams p) {
  gemm_split_body<GeoSmallH, true, true, false, true>(p);
}
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(512, 2) void gemm_f32_split_f16_big_kernel(GemmParams p) {
  gemm_split_body<GeoBigH, A_KC, B_KC, false, false>(p);
}

// Operand bounds of the two-plane kernels.  One pass over a [R x C] matrix (row stride ld): ROWS: out_rows[r] = max_c |P[r][c]|;
// COLS: part_cols[block][c] = max over the block's rows of |P[r][c]|, folded over the blocks by amax_fold_kernel; ALL: *out_all = max of
// everything (one atomic per block; the caller zeroes the word).  Non-negative floats order like their bit patterns, so every maximum
// is an unsigned integer maximum.  NaN / Inf bit patterns win, and the GEMM then produces what an fp32 GEMM would.
// VEC (rows of whole float4s): the matrix is cut into column slabs of 1024; a wave stays on ONE slab (global wave id % slabs) and walks
// rows, two at a time, so a lane owns the same 16 columns in every row it sees: the column maxima live in 16 registers, eight 16-byte
// loads are in flight per lane, and 50-odd registers let 32 waves share a CU.  With more than one slab the row maxima of the slabs
// meet through one atomic per (wave, row) -- out_rows is zeroed by the launcher then.  !VEC: any shape, one wave per row, column maxima
// through LDS atomics (small matrices only).
constexpr int kAmaxWaves = 16;   // waves per block (8 waves per CU streamed 3.6 TB/s)
template <bool ROWS, bool COLS, bool ALL, bool VEC>
__global__ __launch_bounds__(kAmaxWaves * 64) void amax_kernel(const float* __restrict__ P, long R, int C, int ld, unsigned* __restrict__ out_rows,
                                                               unsigned* __restrict__ part_cols, unsigned* __restrict__ out_all, int row_acc) {
  extern __shared__ unsigned smax[];   // COLS: Cp words; ALL: + kAmaxWaves
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Cp = (C + 3) & ~3;   // the partial rows are padded to whole 16-byte words (amax_fold_kernel)
  if (COLS) {
    for (int c = tid; c < Cp; c += kAmaxWaves * 64) smax[c] = 0u;
    __syncthreads();
  }
  unsigned all = 0u;
  if (VEC) {
    const int nslab = (C + 1023) / 1024;
    const long gw = (long)blockIdx.x * kAmaxWaves + wave, nw = (long)gridDim.x * kAmaxWaves;
    const int slab = (int)(gw % nslab);
    const int c0 = slab * 1024 + lane * 4;
    uint4 cm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cm[j] = make_uint4(0u, 0u, 0u, 0u);
    const long rstep = nw / nslab;          // (the launcher makes the grid's waves a multiple of the slabs)
    for (long r = gw / nslab; r < R; r += 2 * rstep) {
      const bool two = r + rstep < R;
      const float* row0 = P + r * ld;
      const float* row1 = P + (two ? r + rstep : r) * ld;
      uint4 v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = c0 + 256 * j;
        v[j] = c < C ? *reinterpret_cast<const uint4*>(row0 + c) : make_uint4(0u, 0u, 0u, 0u);
        v[4 + j] = c < C ? *reinterpret_cast<const uint4*>(row1 + c) : make_uint4(0u, 0u, 0u, 0u);
      }
      unsigned m0 = 0u, m1 = 0u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j].x &= 0x7fffffffu; v[j].y &= 0x7fffffffu; v[j].z &= 0x7fffffffu; v[j].w &= 0x7fffffffu;
        const unsigned mj = max(max(v[j].x, v[j].y), max(v[j].z, v[j].w));
        if (j < 4) m0 = max(m0, mj); else m1 = max(m1, mj);
        if (COLS) { uint4& q = cm[j & 3]; q.x = max(q.x, v[j].x); q.y = max(q.y, v[j].y); q.z = max(q.z, v[j].z); q.w = max(q.w, v[j].w); }
      }
      if (ROWS || ALL) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { m0 = max(m0, (unsigned)__shfl_xor((int)m0, o)); m1 = max(m1, (unsigned)__shfl_xor((int)m1, o)); }
        if (ROWS && lane == 0) {
          if (nslab > 1 || row_acc) { atomicMax(&out_rows[r], m0); if (two) atomicMax(&out_rows[r + rstep], m1); }
          else { out_rows[r] = m0; if (two) out_rows[r + rstep] = m1; }
        }
        all = max(all, max(m0, m1));
      }
    }
    if (COLS) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = c0 + 256 * j;
        if (c < C) { atomicMax(&smax[c], cm[j].x); atomicMax(&smax[c + 1], cm[j].y); atomicMax(&smax[c + 2], cm[j].z); atomicMax(&smax[c + 3], cm[j].w); }
      }
    }
  } else {
    for (long r = (long)blockIdx.x * kAmaxWaves + wave; r < R; r += (long)gridDim.x * kAmaxWaves) {
      const float* row = P + r * ld;
      unsigned m = 0u;
      for (int c = lane; c < C; c += 64) {
        const unsigned v = __builtin_bit_cast(unsigned, row[c]) & 0x7fffffffu;
        m = max(m, v);
        if (COLS) atomicMax(&smax[c], v);
      }
      if (ROWS || ALL) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        if (ROWS && lane == 0) { if (row_acc) atomicMax(&out_rows[r], m); else out_rows[r] = m; }
        all = max(all, m);
      }
    }
  }
  if (COLS) {
    __syncthreads();
    for (int c = tid; c < Cp; c += kAmaxWaves * 64) part_cols[(size_t)blockIdx.x * Cp + c] = smax[c];
  }
  if (ALL) {   // one atomic per block (the launcher gives kAmaxWaves words of LDS)
    unsigned* red = smax + (COLS ? Cp : 0);
    __syncthreads();
    if (lane == 0) red[wave] = all;
    __syncthreads();
    if (tid == 0) {
      unsigned m = 0u;
      for (int w = 0; w < kAmaxWaves; ++w) m = max(m, red[w]);
      if (m) atomicMax(out_all, m);
    }
  }
}
// out[c] = max_b part[b][c]: a block folds 64 columns as 16 column quads x 16 groups of partial rows (16-byte loads, eight in flight per
// lane), then across the groups through LDS.  C is a multiple of 4 here (the launcher pads the partial rows).
__global__ __launch_bounds__(256) void amax_fold_kernel(const unsigned* __restrict__ part, int nb, int Cp, int C, unsigned* __restrict__ out) {
  __shared__ uint4 red[16][16];
  const int cq = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cq * 4;
  uint4 m = make_uint4(0u, 0u, 0u, 0u);
  if (c < Cp)
    for (int b = g; b < nb; b += 16 * 8) {
      uint4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = b + 16 * k < nb ? *reinterpret_cast<const uint4*>(part + (size_t)(b + 16 * k) * Cp + c) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int k = 0; k < 8; ++k) { m.x = max(m.x, v[k].x); m.y = max(m.y, v[k].y); m.z = max(m.z, v[k].z); m.w = max(m.w, v[k].w); }
    }
  red[g][cq] = m;
  __syncthreads();
  if (g == 0 && c < Cp) {
#pragma unroll
    for (int k = 1; k < 16; ++k) { const uint4 o = red[k][cq]; m.x = max(m.x, o.x); m.y = max(m.y, o.y); m.z = max(m.z, o.z); m.w = max(m.w, o.w); }
    if (c + 0 < C) out[c + 0] = m.x;
    if (c + 1 < C) out[c + 1] = m.y;
    if (c + 2 < C) out[c + 2] = m.z;
    if (c + 3 < C) out[c + 3] = m.w;
  }
}

// C = alpha * sum_s ws[s] + beta * C + bias
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int M, int N,
                                                            float alpha, float beta, float* __restrict__ C, int ldc,
                                                            const float* __restrict__ bias) {
  const size_t total = (size_t)M * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += ws[(size_t)k * total + i];
    const int row = (int)(i / N), col = (int)(i % N);
    float v = alpha * s + (bias ? bias[col] : 0.f);
    float* dst = C + (size_t)row * ldc + col;
    if (beta != 0.f) v += beta * *dst;
    *dst = v;
  }
}

}  // namespace

// 0: v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain); 1: 3-way bf16 split on v_mfma_f32_32x32x16_bf16 (six products, fp32-class
// accuracy, 2.67x the matrix rate); 2: two fp16 planes on v_mfma_f32_32x32x16_f16 (three products, the same accuracy class, see the
// comment at kSplitMinW).  EESEN_GEMM_MODE=f32|split|half; read per call so that tests can flip it inside one process.
static int g_gemm_mode = -1;
int gemm_mode() {
  if (g_gemm_mode >= 0) return g_gemm_mode;
  const char* e = getenv("EESEN_GEMM_MODE");
  if (e && (!strcmp(e, "f32") || !strcmp(e, "0"))) return 0;
  if (e && (!strcmp(e, "split") || !strcmp(e, "1"))) return 1;
  if (e && (!strcmp(e, "half") || !strcmp(e, "2"))) return 2;
  return kDefaultGemmMode;
}
void set_gemm_mode(int mode) { g_gemm_mode = mode; }

void amax_abs(hipStream_t st, const float* P, long rows, int cols, int ld, float* out) {
  EESEN_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(float), st));
  amax_abs_accumulate(st, P, rows, cols, ld, out);
}
// blocks of amax_kernel for [rows x cols]: about two (row, 1024-column slab) units per wave, at most kAmaxBlocks, and a whole number of
// slabs' worth of waves (a wave stays on one slab: kAmaxWaves * blocks must be a multiple of the slabs -- 16 is, up to 16 slabs)
static int amax_blocks(long rows, int slabs) { return (int)std::max<long>(1, std::min<long>((rows * slabs + 2 * kAmaxWaves - 1) / (2 * kAmaxWaves), kAmaxBlocks)); }
#define EESEN_AMAX_LAUNCH(R_, C_, A_, LDS)                                                                                                        \
  do {                                                                                                                                            \
    if (vec) hipLaunchKernelGGL((amax_kernel<R_, C_, A_, true>), dim3(nb), dim3(kAmaxWaves * 64), LDS, st, Pc, rows, cc, ld, r, w, a, racc);     \
    else hipLaunchKernelGGL((amax_kernel<R_, C_, A_, false>), dim3(nb), dim3(kAmaxWaves * 64), LDS, st, Pc, rows, cc, ld, r, w, a, racc);        \
  } while (0)
constexpr int kAmaxSlabCols = 16384;   // columns one launch takes (16 slabs of 1024; 64 KB of LDS for the column maxima)
void amax_abs_accumulate(hipStream_t st, const float* P, long rows, int cols, int ld, float* out) {
  if (rows <= 0 || cols <= 0) return;
  if (cols == ld && cols < 1024 && (rows * cols) % 1024 == 0) { rows = rows * cols / 1024; cols = ld = 1024; }   // short contiguous rows: as one flat array
  unsigned *r = nullptr, *w = nullptr, *a = reinterpret_cast<unsigned*>(out);
  const int racc = 0;
  const bool vec = (cols & 3) == 0 && (ld & 3) == 0;
  for (int c0 = 0; c0 < cols; c0 += kAmaxSlabCols) {
    const float* Pc = P + c0;
    const int cc = std::min(kAmaxSlabCols, cols - c0), nb = amax_blocks(rows, vec ? (cc + 1023) / 1024 : 1);
    EESEN_AMAX_LAUNCH(false, false, true, kAmaxWaves * sizeof(unsigned));
  }
  check_launch("amax_abs");
}
void amax_rows_cols(hipStream_t st, const float* P, long rows, int cols, int ld, float* out_rows, float* out_cols, float* ws) {
  if (rows <= 0 || cols <= 0) return;
  EESEN_REQUIRE(!out_cols || ws, EESEN_ERR_INVALID, "amax: column bounds need a workspace");
  unsigned* r = reinterpret_cast<unsigned*>(out_rows);
  unsigned* w = reinterpret_cast<unsigned*>(ws);
  unsigned* a = nullptr;
  const bool vec = (cols & 3) == 0 && (ld & 3) == 0;
  // the row maxima of several column slabs meet through atomics: start from zero
  const bool multi = out_rows && ((vec && cols > 1024) || cols > kAmaxSlabCols);
  if (multi) EESEN_HIP_CHECK(hipMemsetAsync(out_rows, 0, (size_t)rows * sizeof(float), st));
  for (int c0 = 0; c0 < cols; c0 += kAmaxSlabCols) {
    const float* Pc = P + c0;
    const int cc = std::min(kAmaxSlabCols, cols - c0), racc = multi ? 1 : 0;
    const int nb = amax_blocks(rows, vec ? (cc + 1023) / 1024 : 1);
    const int cp = (cc + 3) & ~3;
    if (out_rows && out_cols) EESEN_AMAX_LAUNCH(true, true, false, (size_t)cp * sizeof(unsigned));
    else if (out_cols) EESEN_AMAX_LAUNCH(false, true, false, (size_t)cp * sizeof(unsigned));
    else if (out_rows) EESEN_AMAX_LAUNCH(true, false, false, 0);
    check_launch("amax_rows_cols");
    if (out_cols) {
      hipLaunchKernelGGL(amax_fold_kernel, dim3((cp + 63) / 64), dim3(256), 0, st, w, nb, cp, cc, reinterpret_cast<unsigned*>(out_cols) + c0);
      check_launch("amax_fold");
    }
  }
}
#undef EESEN_AMAX_LAUNCH

// Operand bounds for a two-plane call whose caller passed none: measured here, one word per row of op(A) / column of op(B), in an
// arena of device words handed out round-robin.  A slot is reused after kArena floats' worth of bounds -- at most an eighth of it per
// operand, so never within one call, and far beyond what any stream of this library has in flight; the Net passes its own buffers and
// never comes here.
static GemmBound arena_bound(hipStream_t st, const float* P, bool kc, int R, int K, int ld) {   // op(X) is [R x K]; kc: stored [R x K], else [K x R]
  constexpr size_t kArena = (size_t)4 << 20, kDevs = 16;
  static std::mutex mu;
  static float* arena[kDevs] = {nullptr};   // never freed: lives as long as the process's HIP context
  static size_t next[kDevs] = {0};
  int dev = 0;
  EESEN_HIP_CHECK(hipGetDevice(&dev));
  EESEN_REQUIRE(dev >= 0 && dev < (int)kDevs, EESEN_ERR_INVALID, "gemm: device index beyond the bounds arena table");
  const size_t need = ((size_t)R + 3) & ~(size_t)3;
  EESEN_REQUIRE(need <= kArena / 8, EESEN_ERR_INVALID, "gemm: operand too large for the bounds arena");
  float* slot;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!arena[dev]) EESEN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&arena[dev]), kArena * sizeof(float)));
    if (next[dev] + need > kArena) next[dev] = 0;
    slot = arena[dev] + next[dev];
    next[dev] += need;
  }
  if (kc) amax_rows_cols(st, P, R, K, ld, slot, nullptr, nullptr);
  else {   // the column pass's partial rows: stream-ordered scratch of this call
    float* ws = nullptr;
    EESEN_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&ws), (size_t)kAmaxBlocks * std::min<size_t>(need, kAmaxSlabCols) * sizeof(float), st));
    amax_rows_cols(st, P, K, R, ld, nullptr, slot, ws);
    EESEN_HIP_CHECK(hipFreeAsync(ws, st));
  }
  return GemmBound{slot, 1};
}

// Row-group height of the XCD-aware tile map (0 = plain row-major) and the grid it needs.  Groups of 8 tile-rows once
// every XCD gets at least two of them; fewer rows per group for short grids so that all eight XCDs still get work.
static int xcd_group_rows(int tiles_m, int tiles_n, unsigned* grid_x) {
  if (tiles_m < 8) return 0;
  int gm = 8;
  while (gm > 1 && cdiv(tiles_m, gm) < 16) gm >>= 1;
  const long groups = cdiv(tiles_m, gm), rounds = cdivl(groups, 8);
  const long gx = rounds * 8 * gm * tiles_n;
  if (gx > 0x7fffffffL) return 0;
  *grid_x = (unsigned)gx;
  return gm;
}

void gemm_f32(hipStream_t st, bool a_kc, bool b_kc, int M, int N, int K, float alpha, const float* A, int lda,
              const float* B, int ldb, float beta, float* C, int ldc, const float* bias, float* ws, size_t ws_floats,
              int extra_lds_bytes, bool bf16_operands, GemmBound bound_a, GemmBound bound_b) {
  if (M <= 0 || N <= 0) return;
  EESEN_REQUIRE((lda % 4) == 0 && (ldb % 4) == 0, EESEN_ERR_INVALID, "gemm: leading dimensions must be multiples of 4");
  EESEN_REQUIRE((((uintptr_t)A | (uintptr_t)B) & 15) == 0, EESEN_ERR_INVALID, "gemm: operands must be 16-byte aligned");
  const bool use_split = gemm_mode() >= 1 || bf16_operands;
  const bool half = gemm_mode() == 2 && !bf16_operands;   // two fp16 planes: needs a bound of each operand's largest magnitude
  GemmParams p;
  p.amax_a = p.amax_b = nullptr;
  p.a_vec = p.b_vec = 0;
  if (half) {
    if (!bound_a.p) bound_a = arena_bound(st, A, a_kc, M, K, lda);
    if (!bound_b.p) bound_b = arena_bound(st, B, b_kc, N, K, ldb);
    p.amax_a = bound_a.p; p.a_vec = bound_a.per_index;
    p.amax_b = bound_b.p; p.b_vec = bound_b.per_index;
  }
  p.A = A; p.B = B; p.C = C; p.bias = bias;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.alpha = alpha; p.beta = beta;
  p.gate = GemmGate{nullptr, nullptr, 0, 0, 0, 0, 0, 0};
  p.nprod = bf16_operands ? 1 : 6;
  // The 256 x 256 flavour of the split kernel: main-stream GEMMs (no occupancy cap requested) of shapes made of whole big tiles.
  // A side-stream GEMM has to fit beside a 512-thread recurrence workgroup (<= 168 VGPRs x one wave per SIMD): small flavour.
  const bool big = use_split && !bf16_operands && extra_lds_bytes == 0 && M % 256 == 0 && N % 256 == 0 && K % 16 == 0 &&
                   (long)(M / 256) * (N / 256) >= 16;
  const int TB = big ? 256 : BM;
  p.bm = p.bn = TB;
  const int tiles_m = cdiv(M, TB), tiles_n = cdiv(N, TB);
  p.tiles_n = tiles_n;
  p.tiles_m = tiles_m;
  const long tiles = (long)tiles_m * tiles_n;
  // split-K only when the tile grid cannot fill the chip and K is long enough to amortise the reduce pass
  int splits = 1;
  const int target = big ? 256 : 1024;   // >= 4 workgroups per CU (measured: tall-K W_x gradient 92 -> 110 TF); the big flavour runs one per CU
  if (ws && tiles < target && K >= 2048) {
    splits = (int)std::min<long>((target + tiles - 1) / tiles, K / 1024);
    splits = std::max(1, std::min(splits, 64));
    while (splits > 1 && (size_t)splits * M * N > ws_floats) --splits;
  }
  int k_chunk = cdiv(cdiv(K, splits), 16) * 16;   // whole k-tiles of either kernel (BK = 16 both)
  if (k_chunk < BK) k_chunk = BK;
  splits = std::max(1, cdiv(K, k_chunk));
  p.splits = splits;
  p.k_chunk = k_chunk;
  if (splits > 1) p.C = ws;
  unsigned gx = 0;
  p.gm = xcd_group_rows(tiles_m, tiles_n, &gx);
  dim3 grid(p.gm ? gx : (unsigned)tiles, (unsigned)splits), block(big ? 512 : 256);
  // measured on MI355X: the branch-free loads win only when both operands are k-contiguous (111 vs 107 TF); with an
  // m/n-contiguous operand the guarded code is faster (NN 107 vs 100 TF, tall-K TN 93 vs 67 TF), so it stays guarded
  const bool guard = (M % BM) != 0 || (N % BN) != 0 || (K % k_chunk) != 0 || (k_chunk % BK) != 0 || !(a_kc && b_kc);
#define EESEN_GEMM_LAUNCH(AK, BKC)                                                                        \
  do {                                                                                                    \
    if (guard) hipLaunchKernelGGL((gemm_f32_mfma_kernel<AK, BKC, true>), grid, block, extra_lds_bytes, st, p); \
    else hipLaunchKernelGGL((gemm_f32_mfma_kernel<AK, BKC, false>), grid, block, extra_lds_bytes, st, p);     \
  } while (0)
  if (big && half) {
    if (a_kc && b_kc) hipLaunchKernelGGL((gemm_f32_split_f16_big_kernel<true, true>), grid, block, 0, st, p);
    else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_f32_split_f16_big_kernel<true, false>), grid, block, 0, st, p);
    else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_f32_split_f16_big_kernel<false, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_f32_split_f16_big_kernel<false, false>), grid, block, 0, st, p);
  } else if (big) {   // K and k_chunk are multiples of 16: every split is made of whole k-tiles, every tile is interior
    if (a_kc && b_kc) hipLaunchKernelGGL((gemm_f32_split_bf16_big_kernel<true, true>), grid, block, 0, st, p);
    else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_f32_split_bf16_big_kernel<true, false>), grid, block, 0, st, p);
    else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_f32_split_bf16_big_kernel<false, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_f32_split_bf16_big_kernel<false, false>), grid, block, 0, st, p);
  } else if (use_split) {
    // 49.5 KB (three bf16 planes) / 33 KB (two fp16 planes) of LDS per workgroup: the occupancy caps of the callers (unused dynamic
    // LDS) are sized for the 33 KB of the f32 kernel; keep the same workgroups-per-CU they ask for
    const int extra = extra_lds_bytes > 0 ? std::max(0, extra_lds_bytes + 33792 - 2 * (half ? GeoSmallH::STAGE : GeoSmall::STAGE)) : 0;
    // every tile interior and every split made of whole k-tiles: no clamps, no selects
    const bool sg = (M % BM) != 0 || (N % BN) != 0 || (K % k_chunk) != 0 || (k_chunk % 16) != 0;
#define EESEN_SPLIT_LAUNCH(AK, BKC)                                                                                   \
  do {                                                                                                                \
    if (half && sg) hipLaunchKernelGGL((gemm_f32_split_f16_kernel<AK, BKC, true>), grid, block, extra, st, p);        \
    else if (half) hipLaunchKernelGGL((gemm_f32_split_f16_kernel<AK, BKC, false>), grid, block, extra, st, p);        \
    else if (sg) hipLaunchKernelGGL((gemm_f32_split_bf16_kernel<AK, BKC, true>), grid, block, extra, st, p);          \
    else hipLaunchKernelGGL((gemm_f32_split_bf16_kernel<AK, BKC, false>), grid, block, extra, st, p);                 \
  } while (0)
    if (a_kc && b_kc) EESEN_SPLIT_LAUNCH(true, true);
    else if (a_kc && !b_kc) EESEN_SPLIT_LAUNCH(true, false);
    else if (!a_kc && b_kc) EESEN_SPLIT_LAUNCH(false, true);
    else EESEN_SPLIT_LAUNCH(false, false);
#undef EESEN_SPLIT_LAUNCH
  } else if (a_kc && b_kc) EESEN_GEMM_LAUNCH(true, true);
  else if (a_kc && !b_kc) EESEN_GEMM_LAUNCH(true, false);
  else if (!a_kc && b_kc) EESEN_GEMM_LAUNCH(false, true);
  else EESEN_GEMM_LAUNCH(false, false);
#undef EESEN_GEMM_LAUNCH
  check_launch("gemm_f32_mfma");
  if (splits > 1) {
    const size_t total = (size_t)M * N;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)ws, splits, M, N, alpha, beta,
                       C, ldc, bias);
    check_launch("splitk_reduce");
  }
}

void gemm_f32_nt_gated(hipStream_t st, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                       int ldc, const float* bias, const GemmGate& gate, GemmBound bound_a, GemmBound bound_b) {
  EESEN_REQUIRE((M % BM) == 0 && (N % BN) == 0 && (K % BK) == 0, EESEN_ERR_INVALID, "gated GEMM needs whole tiles");
  EESEN_REQUIRE(gate.ndir * gate.nz * kShards <= 64, EESEN_ERR_INVALID, "gated GEMM: too many counter groups");
  GemmParams p;
  p.A = A; p.B = B; p.C = C; p.bias = bias;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.alpha = 1.f; p.beta = 0.f;
  p.splits = 1; p.k_chunk = K; p.tiles_n = N / BN; p.tiles_m = M / BM;
  p.gate = gate;
  p.nprod = 6;
  p.amax_a = bound_a.p; p.amax_b = bound_b.p; p.a_vec = bound_a.per_index; p.b_vec = bound_b.per_index;
  EESEN_REQUIRE(gemm_mode() != 2 || (bound_a.p && bound_b.p), EESEN_ERR_INVALID, "gated GEMM on fp16 planes needs both operand bounds (A is still being written)");
  p.bm = p.bn = BM;
  unsigned gx = (unsigned)(p.tiles_m * p.tiles_n);
  p.gm = xcd_group_rows(p.tiles_m, p.tiles_n, &gx);
  // Occupancy cap: waiting tiles SPIN, so they must never keep the producing (cooperative) kernel's workgroups from
  // becoming resident.  26 KB of unused dynamic LDS on top of the 33 KB static makes at most two of these workgroups
  // fit a CU (2 x 60 KB), which always leaves room for one 512-thread recurrence workgroup (20 KB LDS, 192 VGPRs/SIMD
  // next to 2 x 112) whatever the dispatch order.
  constexpr int gate_lds = 26 * 1024;
  if (gemm_mode() == 2) hipLaunchKernelGGL(gemm_f32_split_f16_gated_kernel, dim3(gx), dim3(256), std::max(0, gate_lds + 33792 - 2 * GeoSmallH::STAGE), st, p);
  else if (gemm_mode() == 1) hipLaunchKernelGGL(gemm_f32_split_bf16_gated_kernel, dim3(gx), dim3(256), std::max(0, gate_lds + 33792 - 2 * GeoSmall::STAGE), st, p);
  else hipLaunchKernelGGL(gemm_f32_mfma_gated_kernel, dim3(gx), dim3(256), gate_lds, st, p);
  check_launch("gemm_f32_mfma_gated");
}

}  // namespace eesen
