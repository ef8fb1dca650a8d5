// Probe of v_mfma_f32_4x4x1_16B_f32 on gfx950: operand / result lane layout and the CBSZ / ABID broadcast of the A operand.
// build: hipcc --offload-arch=gfx950 -O2 tools/mfma4x4_probe.hip -o tools/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID>
__global__ void probe(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, CBSZ, ABID, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

int main() {
  std::vector<float> a(64), b(64), d(256);
  // A lane l: 1000 * block + 10 * i  (block = l / 4, i = l % 4) + 1 ; B lane l: value 1 if j == 0 else 100^j marker
  for (int l = 0; l < 64; ++l) { a[l] = 100.f * (l / 4) + (l % 4) + 1.f; b[l] = (float)(1 << (4 * (l % 4))) ; }
  float *da, *db, *dd;
  hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
  hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
  auto show = [&](const char* name) {
    hipDeviceSynchronize();
    hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    printf("%s\n", name);
    for (int l = 0; l < 64; ++l) {
      // D = A_i * B_j: report which A value (value / B[lane]) every (lane, vgpr) holds
      printf("lane %2d (blk %2d, x %d):", l, l / 4, l % 4);
      for (int r = 0; r < 4; ++r) printf(" %8.1f", d[l * 4 + r] / b[l]);
      printf("\n");
    }
  };
  hipLaunchKernelGGL((probe<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dd); show("cbsz=0 abid=0: D[lane][vgpr] / B[lane] = the A value used");
  hipLaunchKernelGGL((probe<2, 0>), dim3(1), dim3(64), 0, 0, da, db, dd); show("cbsz=2 abid=0");
  hipLaunchKernelGGL((probe<2, 3>), dim3(1), dim3(64), 0, 0, da, db, dd); show("cbsz=2 abid=3");
  hipLaunchKernelGGL((probe<3, 5>), dim3(1), dim3(64), 0, 0, da, db, dd); show("cbsz=3 abid=5");
  return 0;
}
