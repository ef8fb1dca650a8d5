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
#include "kernels.h"

namespace eesen {
namespace {

#ifndef EESEN_GEMM_PRIO
#define EESEN_GEMM_PRIO 0
#endif
#ifndef EESEN_GEMM_BK
#define EESEN_GEMM_BK 16  // measured on MI355X: BK=32 is -8 % on the k-contiguous shapes (97 vs 107 TF), +3 % on the tall-K transposed ones
#endif
constexpr int BM = 128, BN = 128, BK = EESEN_GEMM_BK, LDP = 4;
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

template <bool A_KC, bool B_KC, bool GUARD, bool GATED = false>
__device__ __forceinline__ void gemm_body(const GemmParams& p, float (*As)[BK][BM + LDP], float (*Bs)[BK][BN + LDP]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int tm, tn;
  if (!tile_of(p, tm, tn)) return;
  if (GATED) {  // middle-out over time: row tiles in the order in which a bidirectional layer completes their frames
    const int tiles_m = p.tiles_m, mid = tiles_m / 2;
    // i = 0, 1, 2, 3, ... -> mid, mid-1, mid+1, mid-2, ...; when one side runs out the other side continues
    const int i = tm, lo_cnt = mid, hi_cnt = tiles_m - mid;   // tiles below mid / at-or-above mid
    const int pairs = min(lo_cnt, hi_cnt);
    if (i < 2 * pairs) tm = (i & 1) ? mid - (i + 1) / 2 : mid + i / 2;
    else tm = hi_cnt > lo_cnt ? mid + (i - pairs) : mid - 1 - (i - pairs);
  }
  const int m0 = tm * BM, n0 = tn * BN;
  if (GATED) {
    __shared__ int s_go;
    const GemmGate& g = p.gate;
    const int t_lo = m0 / g.S, t_hi = min(p.M - 1, m0 + BM - 1) / g.S;
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
    if (!s_go) return;
  }
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
#if EESEN_GEMM_PRIO
    __builtin_amdgcn_s_setprio(EESEN_GEMM_PRIO);
#endif
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
#if EESEN_GEMM_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    if (kt + 1 < nk) {
      store_tile<A_KC>(As[cur ^ 1], tid, ra);
      store_tile<B_KC>(Bs[cur ^ 1], tid, rb);
    }
    __syncthreads();
  }

  // epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float* C = p.C;
  size_t ldc = p.ldc;
  const bool partial = p.splits > 1;
  if (partial) {  // raw partial sums into slab `split` of the workspace, dense [M x N]
    C = p.C + (size_t)split * p.M * p.N;
    ldc = p.N;
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wn * 64 + ni * 32 + lr;
      if (col >= p.N) continue;
      const float bv = (!partial && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row >= p.M) continue;
        float* dst = C + (size_t)row * ldc + col;
        if (partial) {
          *dst = acc[mi][ni][r];
        } else {
          float v = p.alpha * acc[mi][ni][r] + bv;
          if (p.beta != 0.f) v += p.beta * *dst;
          *dst = v;
        }
      }
    }
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

// Interference probes (EESEN_GEMM_SYNTH=1|2|3, side-stream launches only; results are garbage, timing experiments only):
// a stand-in with the GEMM's grid, occupancy and pacing that exercises ONE resource -- 1: the matrix pipe (32 MFMAs per
// k-tile on registers, no memory), 2: the global-load path (the GEMM's tile loads, then sleeps for the MFMA time),
// 3: LDS (the GEMM's LDS stores + reads, then sleeps).  Used to find what a co-running GEMM takes from the recurrence.
template <int MODE>
__global__ __launch_bounds__(256, 2) void gemm_synth_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) float As[2][BK][BM + LDP];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN + LDP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int tile = blockIdx.x, m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
  const int kbeg = blockIdx.y * p.k_chunk, kend = min(p.K, kbeg + p.k_chunk), nk = (kend - kbeg + BK - 1) / BK;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float keep = 0.f;
  float4 ra[NLD], rb[NLD];
  for (int i = 0; i < NLD; ++i) ra[i] = rb[i] = make_float4(1.f, 2.f, 3.f, 4.f);
  for (int kt = 0; kt < nk; ++kt) {
    if (MODE == 1) {
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        const float a0 = (float)lane, b0 = (float)kk;
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[1][1], 0, 0, 0);
      }
    } else if (MODE == 2) {
      load_tile<false, true>(p.A, p.lda, p.M, m0, kbeg + kt * BK, kend, tid, ra);
      load_tile<false, true>(p.B, p.ldb, p.N, n0, kbeg + kt * BK, kend, tid, rb);
      for (int i = 0; i < NLD; ++i) keep += ra[i].x + rb[i].w;
      __builtin_amdgcn_s_sleep(32);  // ~2048 cycles: the k-tile's MFMA time
    } else {
      store_tile<false>(As[kt & 1], tid, ra);
      store_tile<false>(Bs[kt & 1], tid, rb);
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < BK; ++kk) keep += As[kt & 1][kk][lane] + As[kt & 1][kk][64 + lane] + Bs[kt & 1][kk][lane] + Bs[kt & 1][kk][64 + lane];
      __builtin_amdgcn_s_sleep(32);
    }
  }
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) keep += acc[i][j][r];
  if (keep == 12345.678f) p.C[0] = keep;  // never true: keeps the work alive
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

// Row-group height of the XCD-aware tile map (0 = plain row-major) and the grid it needs.  Groups of 8 tile-rows once
// every XCD gets at least two of them; fewer rows per group for short grids so that all eight XCDs still get work.
static int xcd_group_rows(int tiles_m, int tiles_n, unsigned* grid_x) {
  static const int mode = getenv("EESEN_GEMM_XCD") ? atoi(getenv("EESEN_GEMM_XCD")) : 8;  // 0 disables, else the group height
  if (mode <= 0 || tiles_m < 8) return 0;
  int gm = mode;
  while (gm > 1 && cdiv(tiles_m, gm) < 16) gm >>= 1;
  const long groups = cdiv(tiles_m, gm), rounds = cdivl(groups, 8);
  const long gx = rounds * 8 * gm * tiles_n;
  if (gx > 0x7fffffffL) return 0;
  *grid_x = (unsigned)gx;
  return gm;
}

void gemm_f32(hipStream_t st, bool a_kc, bool b_kc, int M, int N, int K, float alpha, const float* A, int lda,
              const float* B, int ldb, float beta, float* C, int ldc, const float* bias, float* ws, size_t ws_floats,
              int extra_lds_bytes) {
  if (M <= 0 || N <= 0) return;
  EESEN_REQUIRE((lda % 4) == 0 && (ldb % 4) == 0, EESEN_ERR_INVALID, "gemm: leading dimensions must be multiples of 4");
  EESEN_REQUIRE((((uintptr_t)A | (uintptr_t)B) & 15) == 0, EESEN_ERR_INVALID, "gemm: operands must be 16-byte aligned");
  static const int synth = getenv("EESEN_GEMM_SYNTH") ? atoi(getenv("EESEN_GEMM_SYNTH")) : 0;
  GemmParams p;
  p.A = A; p.B = B; p.C = C; p.bias = bias;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.alpha = alpha; p.beta = beta;
  p.gate = GemmGate{nullptr, nullptr, 0, 0, 0, 0, 0, 0};
  const int tiles_m = cdiv(M, BM), tiles_n = cdiv(N, BN);
  p.tiles_n = tiles_n;
  p.tiles_m = tiles_m;
  const long tiles = (long)tiles_m * tiles_n;
  // split-K only when the tile grid cannot fill the chip and K is long enough to amortise the reduce pass
  int splits = 1;
  static const int target = getenv("EESEN_GEMM_TARGET_BLOCKS") ? atoi(getenv("EESEN_GEMM_TARGET_BLOCKS")) : 1024;  // >= 4 workgroups per CU (measured: tall-K W_x gradient 92 -> 110 TF)
  if (ws && tiles < target && K >= 2048) {
    splits = (int)std::min<long>((target + tiles - 1) / tiles, K / 1024);
    splits = std::max(1, std::min(splits, 64));
    while (splits > 1 && (size_t)splits * M * N > ws_floats) --splits;
  }
  int k_chunk = cdiv(cdiv(K, splits), BK) * BK;
  if (k_chunk < BK) k_chunk = BK;
  splits = std::max(1, cdiv(K, k_chunk));
  p.splits = splits;
  p.k_chunk = k_chunk;
  if (splits > 1) p.C = ws;
  unsigned gx = 0;
  p.gm = xcd_group_rows(tiles_m, tiles_n, &gx);
  if (synth) p.gm = 0;
  dim3 grid(p.gm ? gx : (unsigned)tiles, (unsigned)splits), block(256);
  // measured on MI355X: the branch-free loads win only when both operands are k-contiguous (111 vs 107 TF); with an
  // m/n-contiguous operand the guarded code is faster (NN 107 vs 100 TF, tall-K TN 93 vs 67 TF), so it stays guarded
  const bool guard = (M % BM) != 0 || (N % BN) != 0 || (K % k_chunk) != 0 || (k_chunk % BK) != 0 || !(a_kc && b_kc);
#define EESEN_GEMM_LAUNCH(AK, BKC)                                                                        \
  do {                                                                                                    \
    if (guard) hipLaunchKernelGGL((gemm_f32_mfma_kernel<AK, BKC, true>), grid, block, extra_lds_bytes, st, p); \
    else hipLaunchKernelGGL((gemm_f32_mfma_kernel<AK, BKC, false>), grid, block, extra_lds_bytes, st, p);     \
  } while (0)
  if (synth && extra_lds_bytes > 0 && !a_kc && !b_kc) {  // interference probe instead of the side-stream weight-gradient GEMM
    static bool warned = false;
    if (!warned) { fprintf(stderr, "eesen_hip: EESEN_GEMM_SYNTH=%d -- weight gradients are NOT computed (timing probe)\n", synth); warned = true; }
    if (synth == 1) hipLaunchKernelGGL(gemm_synth_kernel<1>, grid, block, extra_lds_bytes, st, p);
    else if (synth == 2) hipLaunchKernelGGL(gemm_synth_kernel<2>, grid, block, extra_lds_bytes, st, p);
    else hipLaunchKernelGGL(gemm_synth_kernel<3>, grid, block, extra_lds_bytes, st, p);
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
                       int ldc, const float* bias, const GemmGate& gate) {
  EESEN_REQUIRE((M % BM) == 0 && (N % BN) == 0 && (K % BK) == 0, EESEN_ERR_INVALID, "gated GEMM needs whole tiles");
  EESEN_REQUIRE(gate.ndir * gate.nz * kShards <= 64, EESEN_ERR_INVALID, "gated GEMM: too many counter groups");
  GemmParams p;
  p.A = A; p.B = B; p.C = C; p.bias = bias;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.alpha = 1.f; p.beta = 0.f;
  p.splits = 1; p.k_chunk = K; p.tiles_n = N / BN; p.tiles_m = M / BM;
  p.gate = gate;
  unsigned gx = (unsigned)(p.tiles_m * p.tiles_n);
  p.gm = xcd_group_rows(p.tiles_m, p.tiles_n, &gx);
  // Occupancy cap: waiting tiles SPIN, so they must never keep the producing (cooperative) kernel's workgroups from
  // becoming resident.  26 KB of unused dynamic LDS on top of the 33 KB static makes at most two of these workgroups
  // fit a CU (2 x 60 KB), which always leaves room for one 512-thread recurrence workgroup (20 KB LDS, 192 VGPRs/SIMD
  // next to 2 x 112) whatever the dispatch order.
  static const int gate_lds = (getenv("EESEN_GATE_LDS_KB") ? atoi(getenv("EESEN_GATE_LDS_KB")) : 26) * 1024;
  hipLaunchKernelGGL(gemm_f32_mfma_gated_kernel, dim3(gx), dim3(256), gate_lds, st, p);
  check_launch("gemm_f32_mfma_gated");
}

}  // namespace eesen
