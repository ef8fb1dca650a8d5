// l2_probe.hip -- microbenchmark: what does a workgroup pay to fetch a private read-only slice (a) on every launch of a
// dependent kernel chain, (b) again inside the same launch.  Decides whether the recurrence's weight slices survive in
// the XCD's L2 across kernel boundaries.  Build: hipcc --offload-arch=gfx950 -O3 tools/l2_probe.hip -o /tmp/l2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(512) void fetch(const float4* __restrict__ w, size_t slice_f4, int reps, float* __restrict__ out) {
  const float4* p = w + (size_t)blockIdx.x * slice_f4;
  float acc = 0.f;
  for (int r = 0; r < reps; ++r) {
    for (size_t i = threadIdx.x; i < slice_f4; i += 512 * 8) {
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (i + j * 512 < slice_f4) ? p[i + j * 512] : make_float4(0, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
    }
    asm volatile("" ::: "memory");
  }
  if (acc == 12345.678f) out[blockIdx.x] = acc;
}

int main() {
  const int sizes_kb[] = {16, 32, 64, 128, 256};
  float* out; CK(hipMalloc(&out, 4096));
  for (int blocks : {128, 256}) {
    for (int kb : sizes_kb) {
      const size_t slice_f4 = (size_t)kb * 1024 / 16;
      float4* w; CK(hipMalloc(&w, slice_f4 * 16 * blocks));
      CK(hipMemset(w, 0, slice_f4 * 16 * blocks));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float ms_chain, ms_in;
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(fetch, dim3(blocks), dim3(512), 0, 0, w, slice_f4, 1, out);
      CK(hipEventRecord(e0, 0));
      const int N = 200;
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(fetch, dim3(blocks), dim3(512), 0, 0, w, slice_f4, 1, out);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_chain, e0, e1));
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(fetch, dim3(blocks), dim3(512), 0, 0, w, slice_f4, N, out);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_in, e0, e1));
      // an empty-ish chain for the launch floor
      float ms_floor;
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(fetch, dim3(blocks), dim3(512), 0, 0, w, (size_t)0, 1, out);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_floor, e0, e1));
      printf("blocks %3d slice %3d KB: per-launch %.2f us (floor %.2f us) -> %.1f GB/s/CU over the floor | inside one launch %.2f us/pass -> %.1f GB/s/CU\n",
             blocks, kb, 1e3 * ms_chain / N, 1e3 * ms_floor / N, kb * 1024.0 / (1e3 * (ms_chain - ms_floor) / N) / 1e3,
             1e3 * ms_in / N, kb * 1024.0 / (1e3 * ms_in / N) / 1e3);
      CK(hipFree(w));
    }
  }
  return 0;
}
