// oracle/ref_build/shim/cuda_cpu_shim.h -- TEST INFRASTRUCTURE (oracle), not product code.
//
// A minimal "CUDA on one CPU thread" vocabulary so that the reference's own kernel source
// (/root/reference/src/gpucompute/cuda-kernels.cu) can be compiled by g++ and its
// __global__ functions called directly, one emulated CUDA thread at a time.  Only kernels whose
// threads are independent (no __syncthreads / shared-memory exchange) are meaningful under this
// emulation; the CTC alpha/beta/error kernels (cuda-kernels.cu:1367-1408,1482-1544,1603-1627)
// and the elementwise LSTM helpers are of that kind.
#ifndef EESEN_ORACLE_CUDA_CPU_SHIM_H_
#define EESEN_ORACLE_CUDA_CPU_SHIM_H_
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#define HAVE_CUDA 1
#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct uint3_shim { unsigned x, y, z; };
static uint3_shim blockIdx, threadIdx;
static dim3 blockDim, gridDim;
static inline void __syncthreads() {}
// nvcc puts float overloads of the libm names into the global namespace (log(float) is logf);
// g++'s global ::log is the C double version, so pull the <cmath> overload sets in.
using std::isinf; using std::isnan; using std::min; using std::max;
using std::log; using std::exp; using std::sqrt; using std::pow; using std::fabs; using std::tanh; using std::abs;
#endif
