// Calibration microbenchmarks (not part of the product): sustained v_mfma_f32_32x32x2_f32 rate on
// the whole chip under DVFS with random operands, as a function of accumulator-chain shape and
// waves per SIMD; plus empty-kernel launch cost.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, int iters) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float av = in[threadIdx.x], bv = in[threadIdx.x + 256];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
    av += 1e-9f;
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void empty_kernel(float* p) { if (p == nullptr) p[0] = 1.f; }

template <typename F> float time_ms(F f, int reps) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  f(); hipDeviceSynchronize();
  hipEventRecord(s); for (int i = 0; i < reps; ++i) f(); hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e); return ms / reps;
}
int main() {
  float *in, *out; hipMalloc(&in, 4096); hipMalloc(&out, 4 * 256 * 256 * 16);
  std::vector<float> h(1024); for (int i = 0; i < 1024; ++i) h[i] = (float)rand() / RAND_MAX * 2 - 1;
  hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
  printf("empty kernel: %.2f us per launch (back-to-back)\n", 1e3 * time_ms([&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, out); }, 200));
  for (int iters : {64, 256, 1024, 4096}) {
    for (int wgs_per_cu : {1, 2, 4}) {
      const int grid = 256 * wgs_per_cu;
      const double flop = (double)grid * 4 /*waves*/ * iters * 16 * 4096.0;
      float t1 = time_ms([&] { hipLaunchKernelGGL((mfma_loop<1>), dim3(grid), dim3(256), 0, 0, in, out, iters); }, 10);
      float t2 = time_ms([&] { hipLaunchKernelGGL((mfma_loop<2>), dim3(grid), dim3(256), 0, 0, in, out, iters); }, 10);
      float t4 = time_ms([&] { hipLaunchKernelGGL((mfma_loop<4>), dim3(grid), dim3(256), 0, 0, in, out, iters); }, 10);
      printf("iters %5d wg/cu %d : dep-chain %8.1f us %6.1f TF | 2 acc %8.1f us %6.1f TF | 4 acc %8.1f us %6.1f TF\n", iters,
             wgs_per_cu, t1 * 1e3, flop / t1 / 1e9, t2 * 1e3, flop / t2 / 1e9, t4 * 1e3, flop / t4 / 1e9);
    }
  }
  // partial occupancy: 196 and 392 workgroups (the VGG mid-layer grids)
  for (int grid : {196, 392, 784, 1176}) {
    const int iters = 36 * 2;  // 36 steps x 32 MFMA (128x64 tile) = 1152 MFMAs = 72 iters of 16
    const double flop = (double)grid * 4 * iters * 16 * 4096.0;
    float t = time_ms([&] { hipLaunchKernelGGL((mfma_loop<2>), dim3(grid), dim3(256), 0, 0, in, out, iters); }, 10);
    printf("grid %4d x 1152 MFMA/wave: %8.1f us %6.1f TF\n", grid, t * 1e3, flop / t / 1e9);
  }
  return 0;
}
