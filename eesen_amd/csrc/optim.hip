// optim.hip -- SGD update with momentum + clipping, and small layout helpers, for gfx950.
//
// Reference arithmetic: BiLstm::Update /root/reference/src/net/bilstm-layer.h:846-883 (clip each *_corr_
// to +-max_grad_ when max_grad_ > 0, then param += -lr*learn_rate_coef*corr), Lstm::Update
// src/net/lstm-layer.h:345-369, AffineTransform::Update src/net/affine-trans-layer.h:174-195.  In the
// reference the momentum fold (corr = mmt*corr + gradient) happens inside the gradient GEMMs
// (bilstm-parallel-layer.h:504-510); here the fresh gradient is kept separate until this kernel so a
// data-parallel all-reduce can sum it across ranks first.  One launch per layer
// over its whole flat parameter block (the reference: 24 clip + 12 axpy launches per BiLSTM layer).
#include "kernels.h"

namespace eesen {
namespace {

__global__ __launch_bounds__(256) void sgd_update_kernel(float* __restrict__ param, float* __restrict__ corr,
                                                         const float* __restrict__ fresh, long n, float mmt,
                                                         float lr_coef, float max_grad, const unsigned* __restrict__ skip,
                                                         const float* __restrict__ live) {
  // `skip`: the error word of the persistent recurrence kernels.  If one of them gave up waiting for a peer in THIS step, the
  // gradients are garbage: leave parameters and momentum untouched (the host then drops to the per-step kernels, net.cpp).
  // `live`: the data-parallel liveness word (summed over the ranks with the top layer's gradient bucket, comm.cpp): 0 means
  // NO rank had a minibatch this step -- the closing round of the zero-gradient protocol, which must not move the model.
  if (skip && *skip) return;
  if (live && *live == 0.f) return;
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 p = reinterpret_cast<float4*>(param)[i];
    float4 c = reinterpret_cast<float4*>(corr)[i];
    const float4 g = reinterpret_cast<const float4*>(fresh)[i];
    c.x = mmt * c.x + g.x; c.y = mmt * c.y + g.y; c.z = mmt * c.z + g.z; c.w = mmt * c.w + g.w;
    if (max_grad > 0.f) {
      c.x = fminf(fmaxf(c.x, -max_grad), max_grad); c.y = fminf(fmaxf(c.y, -max_grad), max_grad);
      c.z = fminf(fmaxf(c.z, -max_grad), max_grad); c.w = fminf(fmaxf(c.w, -max_grad), max_grad);
    }
    p.x -= lr_coef * c.x; p.y -= lr_coef * c.y; p.z -= lr_coef * c.z; p.w -= lr_coef * c.w;
    reinterpret_cast<float4*>(corr)[i] = c;
    reinterpret_cast<float4*>(param)[i] = p;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float c = mmt * corr[i] + fresh[i];
    if (max_grad > 0.f) c = fminf(fmaxf(c, -max_grad), max_grad);
    corr[i] = c;
    param[i] -= lr_coef * c;
  }
}

// Adagrad / RMSProp (TrainableLayer::AdagradAccuUpdate / RMSPropAccuUpdate / AdagradScaleCompute,
// /root/reference/src/net/trainable-layer.h:65-114, applied as in bilstm-layer.h:885-955, affine-trans-layer.h:196-218):
//   corr = mmt*corr + fresh; clip;  accu = accu + corr^2 (Adagrad)  |  rho*accu + (1-rho)*corr^2 (RMSProp);
//   param -= lr * corr / sqrt(accu + eps)      (_sqrt_elements cuda-kernels.cu:607-615, _invert_elements :597)
// The reference runs 5 elementwise launches per tensor (60 per BiLSTM layer); here one launch per layer.
__global__ __launch_bounds__(256) void adaptive_update_kernel(float* __restrict__ param, float* __restrict__ corr,
                                                              const float* __restrict__ fresh, float* __restrict__ accu,
                                                              long n, float mmt, float lr, float max_grad, float eps,
                                                              float rho, float one_minus_rho, int rmsprop,
                                                              const unsigned* __restrict__ skip, const float* __restrict__ live) {
  if (skip && *skip) return;   // see sgd_update_kernel
  if (live && *live == 0.f) return;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float c = mmt * corr[i] + fresh[i];
    if (max_grad > 0.f) c = fminf(fmaxf(c, -max_grad), max_grad);
    const float g2 = c * c;
    float a = accu[i];
    a = rmsprop ? (rho * a + one_minus_rho * g2) : (a + g2);
    const float scale = 1.0f / sqrtf(a + eps);
    corr[i] = c;
    accu[i] = a;
    param[i] += -lr * scale * c;
  }
}

// p[0..n) = 0 if *word != 0 (decided on the device, at execution time): a rank whose recurrence kernels raised the error word in
// this step must not put its garbage gradient into the all-reduce -- it contributes zero, like a rank without a minibatch (comm.cpp)
__global__ __launch_bounds__(256) void zero_if_set_kernel(float* __restrict__ p, long n, const unsigned* __restrict__ word) {
  if (*word == 0u) return;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0.f;
}

// 32x32 LDS-tiled transpose, coalesced on both sides
__global__ __launch_bounds__(256) void transpose2d_kernel(const float* __restrict__ src, int rows, int cols,
                                                          float* __restrict__ dst) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int r = r0 + ty + i, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + i][tx] = src[(size_t)r * cols + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int c = c0 + ty + i, r = r0 + tx;
    if (r < rows && c < cols) dst[(size_t)c * rows + r] = tile[tx][ty + i];
  }
}

__global__ __launch_bounds__(256) void copy2d_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst,
                                                     int ldd, int rows, int cols) {
  const size_t total = (size_t)rows * cols;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / cols, c = i % cols;
    dst[r * ldd + c] = src[r * lds + c];
  }
}

}  // namespace

void sgd_update(hipStream_t st, float* param, float* corr, const float* fresh, long n, float mmt, float lr_coef,
                float max_grad, const unsigned* skip, const float* live) {
  if (n <= 0) return;
  const int blocks = (int)std::min<long>(cdivl(n / 4 + 1, 256), 2048);
  hipLaunchKernelGGL(sgd_update_kernel, dim3(blocks), dim3(256), 0, st, param, corr, fresh, n, mmt, lr_coef, max_grad, skip, live);
  check_launch("sgd_update");
}

void zero_if_set(hipStream_t st, float* p, long n, const unsigned* word) {
  if (n <= 0 || !word) return;
  hipLaunchKernelGGL(zero_if_set_kernel, dim3(256), dim3(256), 0, st, p, n, word);
  check_launch("zero_if_set");
}

void adaptive_update(hipStream_t st, float* param, float* corr, const float* fresh, float* accu, long n, float mmt, float lr,
                     float max_grad, float eps, float rho, float one_minus_rho, bool rmsprop, const unsigned* skip, const float* live) {
  if (n <= 0) return;
  const int blocks = (int)std::min<long>(cdivl(n, 256), 4096);
  hipLaunchKernelGGL(adaptive_update_kernel, dim3(blocks), dim3(256), 0, st, param, corr, fresh, accu, n, mmt, lr, max_grad, eps,
                     rho, one_minus_rho, rmsprop ? 1 : 0, skip, live);
  check_launch("adaptive_update");
}

void transpose2d(hipStream_t st, const float* src, int rows, int cols, float* dst) {
  hipLaunchKernelGGL(transpose2d_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, st, src, rows, cols, dst);
  check_launch("transpose2d");
}

void copy2d(hipStream_t st, const float* src, int lds, float* dst, int ldd, int rows, int cols) {
  if (rows <= 0 || cols <= 0) return;
  const size_t total = (size_t)rows * cols;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(copy2d_kernel, dim3(blocks), dim3(256), 0, st, src, lds, dst, ldd, rows, cols);
  check_launch("copy2d");
}


namespace {
__device__ __forceinline__ float hash_uniform(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed + (idx + 1ull) * 0x9E3779B97F4A7C15ull;  // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);  // 24 random bits -> [0, 1)
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ out, long rows, int cols, int ld, float p, float scale,
                                                           unsigned long long seed, int per_column) {
  const long n = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols;
    const int c = (int)(i % cols);
    const float u = hash_uniform(seed, per_column ? (unsigned long long)c : (unsigned long long)i);
    out[r * ld + c] = (u - p > 0.f) ? scale : 0.f;
  }
}
__global__ __launch_bounds__(256) void mul_elements_kernel(const float* __restrict__ a, int lda, const float* __restrict__ m, int ldm,
                                                           float* __restrict__ out, int ldo, long rows, int cols) {
  const long n = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols;
    const int c = (int)(i % cols);
    out[r * ldo + c] = a[r * lda + c] * m[r * ldm + c];
  }
}
// y = 1 / (1 + exp(-x))  |  (exp(2x) - 1) / (exp(2x) + 1), 1 where exp(2x) overflows: the reference's device formulas
template <bool TANH>
__global__ __launch_bounds__(256) void activation_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, long rows, int cols) {
  const long n = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols;
    const int c = (int)(i % cols);
    const float v = x[r * ldx + c];
    float o;
    if (TANH) {
      const float e = expf(2.0f * v);
      o = isinf(e) ? 1.0f : (e - 1.0f) / (e + 1.0f);
    } else {
      o = 1.0f / (1.0f + expf(-v));
    }
    y[r * ldy + c] = o;
  }
}
template <bool TANH>
__global__ __launch_bounds__(256) void activation_diff_kernel(const float* __restrict__ y, int ldy, float* __restrict__ d, int ldd, long rows, int cols) {
  const long n = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols;
    const int c = (int)(i % cols);
    const float o = y[r * ldy + c];
    d[r * ldd + c] *= TANH ? (1.0f - o * o) : o * (1.0f - o);
  }
}
}  // namespace

void activation_rows(hipStream_t st, bool tanh_, const float* x, int ldx, float* y, int ldy, long rows, int cols) {
  if (rows <= 0 || cols <= 0) return;
  const int blocks = (int)std::min<long>(cdivl(rows * cols, 256), 256L * 16);
  if (tanh_) hipLaunchKernelGGL(activation_kernel<true>, dim3(blocks), dim3(256), 0, st, x, ldx, y, ldy, rows, cols);
  else hipLaunchKernelGGL(activation_kernel<false>, dim3(blocks), dim3(256), 0, st, x, ldx, y, ldy, rows, cols);
  check_launch("activation_rows");
}

void activation_diff_rows(hipStream_t st, bool tanh_, const float* y, int ldy, float* d, int ldd, long rows, int cols) {
  if (rows <= 0 || cols <= 0) return;
  const int blocks = (int)std::min<long>(cdivl(rows * cols, 256), 256L * 16);
  if (tanh_) hipLaunchKernelGGL(activation_diff_kernel<true>, dim3(blocks), dim3(256), 0, st, y, ldy, d, ldd, rows, cols);
  else hipLaunchKernelGGL(activation_diff_kernel<false>, dim3(blocks), dim3(256), 0, st, y, ldy, d, ldd, rows, cols);
  check_launch("activation_diff_rows");
}

void dropout_mask(hipStream_t st, float* out, long rows, int cols, int ld, float p, unsigned long long seed, bool per_column) {
  if (rows <= 0 || cols <= 0) return;
  const int blocks = (int)std::min<long>(cdivl(rows * cols, 256), 256L * 16);
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks), dim3(256), 0, st, out, rows, cols, ld, p, 1.0f / (1.0f - p), seed, per_column ? 1 : 0);
  check_launch("dropout_mask");
}

void mul_elements(hipStream_t st, const float* a, int lda, const float* m, int ldm, float* out, int ldo, long rows, int cols) {
  if (rows <= 0 || cols <= 0) return;
  const int blocks = (int)std::min<long>(cdivl(rows * cols, 256), 256L * 16);
  hipLaunchKernelGGL(mul_elements_kernel, dim3(blocks), dim3(256), 0, st, a, lda, m, ldm, out, ldo, rows, cols);
  check_launch("mul_elements");
}

// ---- Net::Info / Net::InfoGradient (net.cc:336-385): MomentStatistics of one tensor (utils-functions.h:50-82) -------------
namespace {
// One 1024-thread workgroup per tensor and pass (the statistics are printed once, when training ends): pass 0 leaves
// {min, max, sum} in ws[0..2], pass 1 the central power sums and the six printed figures in out6.
template <int PASS>
__global__ __launch_bounds__(1024) void tensor_moments_kernel(const float* __restrict__ base, long rows, int cols, long ld,
                                                              double* __restrict__ out6, double* __restrict__ ws, int nb, int hf, int hi) {
  __shared__ double red[4][16];
  const long n = rows * cols;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double a = PASS == 0 ? 1e300 : 0.0, b = PASS == 0 ? -1e300 : 0.0, c = 0.0, d = 0.0;
  const double mean = PASS == 1 ? ws[2] / (double)n : 0.0;
  for (long i = tid; i < n; i += 1024) {
    const int cf = (int)(i % cols);   // the FILE's column; (nb, hf, hi): where it lives in the row (Layer::in_col, net.h)
    const double v = (double)base[(i / cols) * ld + (nb ? (cf / hf) * hi + cf % hf : cf)];
    if (PASS == 0) { a = fmin(a, v); b = fmax(b, v); c += v; }
    else { const double e = v - mean, e2 = e * e; a += e2; b += e2 * e; c += e2 * e2; }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double a2 = __shfl_down(a, off), b2 = __shfl_down(b, off), c2 = __shfl_down(c, off);
    if (PASS == 0) { a = fmin(a, a2); b = fmax(b, b2); c += c2; }
    else { a += a2; b += b2; c += c2; }
  }
  if (lane == 0) { red[0][wave] = a; red[1][wave] = b; red[2][wave] = c; red[3][wave] = d; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w) {
      if (PASS == 0) { a = fmin(a, red[0][w]); b = fmax(b, red[1][w]); c += red[2][w]; }
      else { a += red[0][w]; b += red[1][w]; c += red[2][w]; }
    }
    if (PASS == 0) { ws[0] = a; ws[1] = b; ws[2] = c; }
    else {
      const double var = a / (double)n;
      out6[0] = ws[0]; out6[1] = ws[1]; out6[2] = mean; out6[3] = var;
      out6[4] = b / pow(var, 1.5) / (double)n;          // skewness
      out6[5] = c / (var * var) / (double)n - 3.0;      // kurtosis
    }
  }
}
}  // namespace

void tensor_moments(hipStream_t st, const float* base, long rows, int cols, long ld, double* out6, double* ws, int nb, int hf, int hi) {
  hipLaunchKernelGGL(tensor_moments_kernel<0>, dim3(1), dim3(1024), 0, st, base, rows, cols, ld, out6, ws, nb, hf, hi);
  hipLaunchKernelGGL(tensor_moments_kernel<1>, dim3(1), dim3(1024), 0, st, base, rows, cols, ld, out6, ws, nb, hf, hi);
  check_launch("tensor_moments");
}

}  // namespace eesen
