// lstm.hip -- the time recurrence of the peephole (Bi-)LSTM, forward and backward, for gfx950.
//
// Reference arithmetic: /root/reference/src/net/bilstm-parallel-layer.h:97-206 (forward, both
// directions), :422-602 (backward), and the uni-directional twin src/net/lstm-parallel-layer.h:47-213.
// The reference spends ~13 elementwise launches + 1 cuBLAS SGEMM + 14 device syncs per step and
// direction; here ONE launch per step covers both directions, the recurrent GEMM (MFMA), the peephole /
// gate / cell math and the padding mask.
//
// Work decomposition (MI355X-first):
//   forward : one workgroup per (4 hidden units, direction, 32 sequences).  The 16 gate rows of W_m that
//             feed those 4 units (gate-interleaved layout) form the B operand of
//             v_mfma_f32_16x16x4_f32; the A operand is m_{t-1} [S x H].  The K = H reduction is split
//             over the 8 waves of the workgroup and combined through LDS; 128 threads then apply the
//             cell equations for their (sequence, unit) and write g,i,f,o / c / m.  At H = 512 this is
//             256 workgroups = one per CU, each streaming its 32 KB slice of W_m from its XCD's L2.
//   backward: one workgroup per (16 hidden units, direction, 16 sequences): d_m = dY_t + DG_next * W_m
//             has K = 4H, again split over 8 waves; 256 threads finish the cell gradient.
// Both operands are fetched as two float4 per lane along k (lane (i, kq) owns k = k0 + 8 kq .. +7, MFMA
// number c consumes component c), so every row access is a full 128-byte line and no transposition or
// LDS staging of operands is needed.
#include "kernels.h"

namespace eesen {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NW = 8;  // waves per workgroup in the step kernels

// v_exp_f32 / v_rcp_f32 (1 ulp each) instead of the ~40-instruction libm expf and IEEE division: the cell update is on the
// per-step critical path; the error (~2e-7 relative) is three orders below the parity bar
// __frcp_rn compiles to the IEEE division sequence (v_div_scale / v_rcp / 3 fma / v_div_fmas / v_div_fixup, ~10 dependent
// instructions); v_rcp_f32 alone is 1 ulp -- three orders below the parity bar, and the cell update is on the per-step critical path
#ifndef EESEN_RCP
#define EESEN_RCP(x) __builtin_amdgcn_rcpf(x)
#endif
__device__ __forceinline__ float sigmoidf_(float x) { return EESEN_RCP(1.f + __expf(-x)); }
// tanh through exp(2x): exact limits at +-inf (exp -> inf => 1, exp -> 0 => -1)
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f * EESEN_RCP(1.f + __expf(2.f * x)); }

__device__ __forceinline__ void ld8(const float* __restrict__ row, int k, int kmax, bool ok, float (&v)[8]) {
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (ok && k < kmax) a = *reinterpret_cast<const float4*>(row + k);
  if (ok && k + 4 < kmax) b = *reinterpret_cast<const float4*>(row + k + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// ------------------------------------------------------------------------------------------------
// forward step
// ------------------------------------------------------------------------------------------------
template <int CPW>  // CPW k-chunks (of 32) per wave: every operand load of the step is in flight before the first MFMA
__global__ __launch_bounds__(NW * 64) void lstm_fwd_step_kernel(LstmLayerDev L, int step) {
  __shared__ __attribute__((aligned(16))) float red[NW][32][20];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = L.H, S = L.S, T = L.T;
  const int ldY = L.ndir * H, ldG = L.ndir * 4 * H;
  const int u0 = blockIdx.x * 4, dir = blockIdx.y, s0 = blockIdx.z * 32;
  const int t = dir == 0 ? step : T - 1 - step;
  const int tp = dir == 0 ? t - 1 : t + 1;  // the step the recurrence reads (row block tp + 1; boundaries are zero)

  // epilogue operands first, so their latency hides under the MFMA loop
  const int es = tid >> 2, eu = tid & 3;
  const int s_e = s0 + es;
  const bool e_ok = tid < 128 && s_e < S;
  float4 gx = make_float4(0.f, 0.f, 0.f, 0.f);
  float cprev = 0.f, p_i = 0.f, p_f = 0.f, p_o = 0.f;
  int len = 0;
  const float* gptr = L.G + (size_t)(t * S + s_e) * ldG + (size_t)dir * 4 * H + (u0 + eu) * 4;
  if (e_ok) {
    gx = *reinterpret_cast<const float4*>(gptr);
    cprev = L.C[(size_t)((tp + 1) * S + s_e) * ldY + dir * H + u0 + eu];
    const float* pp = L.peep + (size_t)dir * 3 * H + u0 + eu;
    p_i = pp[0]; p_f = pp[H]; p_o = pp[2 * H];
    len = L.lens[s_e];
  }

  const int li = lane & 15, kq = lane >> 4;
  const float* Yp = L.Y + (size_t)(tp + 1) * S * ldY + dir * H;
  const float* Wr = L.Wm + ((size_t)dir * 4 * H + (size_t)u0 * 4 + li) * H;
  const int sa0 = s0 + li, sa1 = s0 + 16 + li;
  const float* A0 = Yp + (size_t)sa0 * ldY;
  const float* A1 = Yp + (size_t)sa1 * ldY;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int nch = (H + 31) >> 5;
  for (int cb = wave; cb < nch; cb += NW * CPW) {
    float b[CPW][8], a0[CPW][8], a1[CPW][8];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      const int ch = cb + c * NW;
      const int k = ch * 32 + kq * 8;  // k >= H for ch >= nch: ld8 then returns zeros
      ld8(A0, k, H, sa0 < S, a0[c]);
      ld8(A1, k, H, sa1 < S, a1[c]);
      ld8(Wr, k, H, true, b[c]);
    }
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c][j], b[c][j], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c][j], b[c][j], acc1, 0, 0, 0);
      }
  }
  // pull the NEXT step's gate pre-activations toward this XCD's L2 (same workgroup index = same XCD next launch);
  // issued after every real load so no wait of this step covers it, consumed by nothing but the final keep-alive
  float4 pf = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool pf_ok = e_ok && step + 1 < T;
  if (pf_ok) pf = *reinterpret_cast<const float4*>(dir == 0 ? gptr + (size_t)S * ldG : gptr - (size_t)S * ldG);
  // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    red[wave][4 * kq + r][li] = acc0[r];
    red[wave][16 + 4 * kq + r][li] = acc1[r];
  }
  __syncthreads();
  if (!e_ok) return;
  float4 pre = gx;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const float4 v = *reinterpret_cast<const float4*>(&red[w][es][eu * 4]);
    pre.x += v.x; pre.y += v.y; pre.z += v.z; pre.w += v.w;
  }
  // bilstm-parallel-layer.h:127-147: i,f see c_{prev} through the peepholes, o sees the new c
  float g = tanhf_(pre.x);
  float i = sigmoidf_(pre.y + p_i * cprev);
  float f = sigmoidf_(pre.z + p_f * cprev);
  float c = g * i + cprev * f;
  if (L.drop_mode) {  // recurrent dropout (:266-272): the mask multiplies g*i (no-memory-loss) or the whole new cell (RNNDrop)
    const float mk = L.rmask[(size_t)((t + 1) * S + s_e) * ldY + dir * H + u0 + eu];
    c = L.drop_mode == 1 ? mk * (g * i) + cprev * f : mk * (g * i + cprev * f);
  }
  float h = tanhf_(c);
  float o = sigmoidf_(pre.w + p_o * c);
  float m = h * o;
  if (t >= len) { g = i = f = o = c = m = 0.f; }  // padded frame (the reference masks the bw direction, :201-204)
  *reinterpret_cast<float4*>(L.G + (size_t)(t * S + s_e) * ldG + (size_t)dir * 4 * H + (u0 + eu) * 4) =
      make_float4(g, i, f, o);
  const size_t o1 = (size_t)((t + 1) * S + s_e) * ldY + dir * H + u0 + eu;
  L.C[o1] = c;
  L.Y[o1] = m;
  asm volatile("" ::"v"(pf.x), "v"(pf.y), "v"(pf.z), "v"(pf.w));
}

// ------------------------------------------------------------------------------------------------
// backward step
// ------------------------------------------------------------------------------------------------
template <int CPW>
__global__ __launch_bounds__(NW * 64) void lstm_bwd_step_kernel(LstmLayerDev L, int step, const float* __restrict__ dY,
                                                                int lddy, float* __restrict__ DG,
                                                                float* __restrict__ DCF) {
  __shared__ float red[NW][16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = L.H, S = L.S, T = L.T;
  const int ldY = L.ndir * H, ldG = L.ndir * 4 * H, K4 = 4 * H;
  const int u0 = blockIdx.x * 16, dir = blockIdx.y, s0 = blockIdx.z * 16;
  const int t = dir == 0 ? T - 1 - step : step;
  const int tn = dir == 0 ? t + 1 : t - 1;  // the step whose gradient flows in (processed just before)
  const int tp = dir == 0 ? t - 1 : t + 1;  // source of the forward recurrence (c_prev)
  const bool has_next = step > 0;

  const int es = tid >> 4, eu = tid & 15;
  const int s_e = s0 + es, u_e = u0 + eu;
  const bool e_ok = tid < 256 && s_e < S && u_e < H;
  float4 gt = make_float4(0.f, 0.f, 0.f, 0.f), dgn = gt;
  float dy = 0.f, c_t = 0.f, c_p = 0.f, dcf = 0.f, p_i = 0.f, p_f = 0.f, p_o = 0.f;
  int len = 0;
  const size_t gofs = (size_t)(t * S + s_e) * ldG + (size_t)dir * K4 + u_e * 4;
  const size_t yofs = (size_t)(t * S + s_e) * lddy + dir * H + u_e;
  const size_t cofs = (size_t)((t + 1) * S + s_e) * ldY + dir * H + u_e;
  if (e_ok) {
    gt = *reinterpret_cast<const float4*>(L.G + gofs);
    dy = dY[yofs];
    c_t = L.C[cofs];
    c_p = L.C[(size_t)((tp + 1) * S + s_e) * ldY + dir * H + u_e];
    if (has_next) {
      dgn = *reinterpret_cast<const float4*>(DG + (size_t)(tn * S + s_e) * ldG + (size_t)dir * K4 + u_e * 4);
      dcf = DCF[(size_t)s_e * ldY + dir * H + u_e];
    }
    const float* pp = L.peep + (size_t)dir * 3 * H + u_e;
    p_i = pp[0]; p_f = pp[H]; p_o = pp[2 * H];
    len = L.lens[s_e];
  }

  const int li = lane & 15, kq = lane >> 4;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  if (has_next) {
    // d_m += DG_next[S x 4H] * W_m[4H x H]  (:470 / :561); B operand rows come from the transposed copy
    const int sa = s0 + li, ub = u0 + li;
    const float* Ar = DG + (size_t)(tn * S + sa) * ldG + (size_t)dir * K4;
    const float* Br = L.WmT + ((size_t)dir * H + ub) * K4;
    const bool a_ok = sa < S, b_ok = ub < H;
    const int nch = (K4 + 31) >> 5;
    for (int cb = wave; cb < nch; cb += NW * CPW) {
      float a[CPW][8], b[CPW][8];
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        const int k = (cb + c * NW) * 32 + kq * 8;  // beyond K4: zeros
        ld8(Ar, k, K4, a_ok, a[c]);
        ld8(Br, k, K4, b_ok, b[c]);
      }
#pragma unroll
      for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], b[c][j], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j + 1], b[c][j + 1], acc1, 0, 0, 0);
        }
    }
  }
  // next step's epilogue operands -> this XCD's L2 (see the forward kernel)
  float4 pf = make_float4(0.f, 0.f, 0.f, 0.f);
  float pf1 = 0.f, pf2 = 0.f;
  if (e_ok && step + 1 < T) {
    const long d = dir == 0 ? -(long)S : (long)S;  // row offset of the next processed step
    pf = *reinterpret_cast<const float4*>(L.G + gofs + d * ldG);
    pf1 = dY[yofs + d * lddy];
    pf2 = L.C[cofs + 2 * d * ldY];  // its c_prev; its c_t is this step's c_prev, already here
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][4 * kq + r][li] = acc0[r] + acc1[r];
  __syncthreads();
  if (!e_ok) return;
  float dm = dy;
#pragma unroll
  for (int w = 0; w < NW; ++w) dm += red[w][es][eu];
  const float g = gt.x, i = gt.y, f = gt.z, o = gt.w;
  const float h = tanhf_(c_t);
  // bilstm-parallel-layer.h:473-497 (fw) / :564-588 (bw)
  const float dh = (1.f - h * h) * (dm * o);
  float dob = o * (1.f - o) * (dm * h);
  float dc = dh + dcf + dgn.y * p_i + dgn.z * p_f + dob * p_o;
  // recurrent dropout (:700-725): d_i, d_g see d_c times the mask; with RNNDrop so do d_f and the carry to the next step
  float dcm = dc, dcx = dc;
  if (L.drop_mode) {
    dcm = dc * L.rmask[cofs];
    if (L.drop_mode == 2) dcx = dcm;
  }
  float df = f * (1.f - f) * (dcx * c_p);
  float di = i * (1.f - i) * (dcm * g);
  float dg = (1.f - g * g) * (dcm * i);
  float carry = dcx * f;  // what the next step adds as d_c,next * f_next (:482 / :701,:706)
  if (t >= len) { dg = di = df = dob = 0.f; carry = 0.f; }
  *reinterpret_cast<float4*>(DG + gofs) = make_float4(dg, di, df, dob);
  DCF[(size_t)s_e * ldY + dir * H + u_e] = carry;
  asm volatile("" ::"v"(pf.x), "v"(pf.y), "v"(pf.z), "v"(pf.w), "v"(pf1), "v"(pf2));
}

// ------------------------------------------------------------------------------------------------
// bias / peephole gradients: column sums of DG and diag(DG^T C) products, two deterministic passes
// ------------------------------------------------------------------------------------------------
constexpr int RED_RB = 64;  // row blocks of pass 1

__global__ __launch_bounds__(256) void lstm_bias_peep_pass1(const float* __restrict__ DG, const float* __restrict__ C,
                                                            int R, int S, int H, int ndir, float* __restrict__ ws) {
  __shared__ float sm[4][64][7];
  const int ul = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int NU = ndir * H;
  const int gu = blockIdx.x * 64 + ul;
  const int rows_per = (R + RED_RB - 1) / RED_RB;
  const int rbeg = blockIdx.y * rows_per, rend = min(R, rbeg + rows_per);
  float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (gu < NU) {
    const int dir = gu / H;
    const size_t cs_shift = dir == 0 ? 0 : 2 * (size_t)S;  // c_{t-1} for fw (:508), c_{t+1} for bw (:599)
    for (int r = rbeg + rl; r < rend; r += 4) {
      const float4 d = *reinterpret_cast<const float4*>(DG + (size_t)r * NU * 4 + (size_t)gu * 4);
      const float cs = C[((size_t)r + cs_shift) * NU + gu];
      const float c1 = C[((size_t)r + S) * NU + gu];
      acc[0] += d.x; acc[1] += d.y; acc[2] += d.z; acc[3] += d.w;
      acc[4] += d.y * cs; acc[5] += d.z * cs; acc[6] += d.w * c1;
    }
  }
#pragma unroll
  for (int q = 0; q < 7; ++q) sm[rl][ul][q] = acc[q];
  __syncthreads();
  if (rl == 0 && gu < NU) {
#pragma unroll
    for (int q = 0; q < 7; ++q)
      ws[((size_t)blockIdx.y * NU + gu) * 7 + q] = sm[0][ul][q] + sm[1][ul][q] + sm[2][ul][q] + sm[3][ul][q];
  }
}

__global__ __launch_bounds__(256) void lstm_bias_peep_pass2(const float* __restrict__ ws, int H, int ndir,
                                                            float* __restrict__ bias_grad, float* __restrict__ peep_grad) {
  const int NU = ndir * H;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NU * 7) return;
  const int gu = idx / 7, q = idx % 7;
  float s = 0.f;
  for (int rb = 0; rb < RED_RB; ++rb) s += ws[((size_t)rb * NU + gu) * 7 + q];
  if (q < 4) {
    bias_grad[(size_t)gu * 4 + q] = s;
  } else {
    const int dir = gu / H, u = gu % H;
    peep_grad[((size_t)dir * 3 + (q - 4)) * H + u] = s;
  }
}

__global__ __launch_bounds__(256) void col_sums_pass1(const float* __restrict__ M, int rows, int cols, int ld,
                                                      float* __restrict__ ws) {
  __shared__ float sm[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int rows_per = (rows + RED_RB - 1) / RED_RB;
  const int rbeg = blockIdx.y * rows_per, rend = min(rows, rbeg + rows_per);
  float acc = 0.f;
  if (c < cols)
    for (int r = rbeg + rl; r < rend; r += 4) acc += M[(size_t)r * ld + c];
  sm[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && c < cols) ws[(size_t)blockIdx.y * cols + c] = sm[0][cl] + sm[1][cl] + sm[2][cl] + sm[3][cl];
}

__global__ __launch_bounds__(256) void col_sums_pass2(const float* __restrict__ ws, int cols, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (int rb = 0; rb < RED_RB; ++rb) s += ws[(size_t)rb * cols + c];
  out[c] = s;
}

}  // namespace

static int pick_cpw(int nch) {  // chunks per wave so that one pass covers the K range where possible
  const int need = (nch + NW - 1) / NW;
  return need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : 8;
}

void lstm_fwd_step(hipStream_t st, const LstmLayerDev& L, int step) {
  dim3 grid(L.H / 4, L.ndir, cdiv(L.S, 32)), block(NW * 64);
  switch (std::min(4, pick_cpw((L.H + 31) / 32))) {
    case 1: hipLaunchKernelGGL(lstm_fwd_step_kernel<1>, grid, block, 0, st, L, step); break;
    case 2: hipLaunchKernelGGL(lstm_fwd_step_kernel<2>, grid, block, 0, st, L, step); break;
    default: hipLaunchKernelGGL(lstm_fwd_step_kernel<4>, grid, block, 0, st, L, step); break;
  }
}

void lstm_bwd_step(hipStream_t st, const LstmLayerDev& L, int step, const float* dY, int lddy, float* DG, float* DCF) {
  dim3 grid(cdiv(L.H, 16), L.ndir, cdiv(L.S, 16)), block(NW * 64);
  // measured on MI355X at H = 512: 2 chunks in flight per wave (77 VGPRs) beat 8 (190 VGPRs): 10.0 vs 10.9 us per step
  if (pick_cpw((4 * L.H + 31) / 32) <= 1) hipLaunchKernelGGL(lstm_bwd_step_kernel<1>, grid, block, 0, st, L, step, dY, lddy, DG, DCF);
  else hipLaunchKernelGGL(lstm_bwd_step_kernel<2>, grid, block, 0, st, L, step, dY, lddy, DG, DCF);
}

size_t lstm_bias_peep_ws_floats(int T, int S, int H, int ndir) { return (size_t)RED_RB * ndir * H * 7; }

void lstm_bias_peep_grads(hipStream_t st, const LstmLayerDev& L, const float* DG, float* bias_grad, float* peep_grad,
                          float* ws, size_t ws_floats) {
  const int NU = L.ndir * L.H, R = L.T * L.S;
  EESEN_REQUIRE(ws_floats >= lstm_bias_peep_ws_floats(L.T, L.S, L.H, L.ndir), EESEN_ERR_INVALID, "reduction workspace too small");
  hipLaunchKernelGGL(lstm_bias_peep_pass1, dim3(cdiv(NU, 64), RED_RB), dim3(256), 0, st, DG, (const float*)L.C, R, L.S,
                     L.H, L.ndir, ws);
  check_launch("lstm_bias_peep_pass1");
  hipLaunchKernelGGL(lstm_bias_peep_pass2, dim3(cdiv(NU * 7, 256)), dim3(256), 0, st, (const float*)ws, L.H, L.ndir,
                     bias_grad, peep_grad);
  check_launch("lstm_bias_peep_pass2");
}

size_t col_sums_ws_floats(int rows, int cols) { return (size_t)RED_RB * cols; }

void col_sums(hipStream_t st, const float* M, int rows, int cols, int ld, float* out, float* ws, size_t ws_floats) {
  EESEN_REQUIRE(ws_floats >= col_sums_ws_floats(rows, cols), EESEN_ERR_INVALID, "reduction workspace too small");
  hipLaunchKernelGGL(col_sums_pass1, dim3(cdiv(cols, 64), RED_RB), dim3(256), 0, st, M, rows, cols, ld, ws);
  check_launch("col_sums_pass1");
  hipLaunchKernelGGL(col_sums_pass2, dim3(cdiv(cols, 256)), dim3(256), 0, st, (const float*)ws, cols, out);
  check_launch("col_sums_pass2");
}

}  // namespace eesen
