// Sustained v_mfma_f32_32x32x16_f16 rate of the whole chip under DVFS (not part of the product): every CU, 1 or 2 waves
// per SIMD, back-to-back MFMAs on 4 independent accumulators with RANDOM f16 operands (rotating among 4 register
// fragments: operand toggling is what the power goes into) -- pure MFMA, and with the conv kernels' operand traffic
// beside it (two ds_read_b128 per MFMA triple).  Prints TFLOP/s against the 2.5 PFLOP/s dense peak of the data sheet
// (2.4 GHz boost): what a kernel that issued NOTHING but MFMAs would reach on this part.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f16_rate mfma_f16_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LDSR>
__global__ __launch_bounds__(256) void mfma_loop(const _Float16* __restrict__ in, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[8 * 1024];
  for (int i = threadIdx.x; i < 8 * 1024; i += 256) lds[i] = in[i];
  __syncthreads();
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  h8 fa[4], fb[4];
  for (int k = 0; k < 4; ++k) {
    fa[k] = *reinterpret_cast<const h8*>(in + ((threadIdx.x * 4 + k) * 8) % 8192);
    fb[k] = *reinterpret_cast<const h8*>(in + 8192 + ((threadIdx.x * 4 + k) * 8) % 8192);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      if (LDSR && u % 3 == 0) {   // the conv kernels' A-fragment traffic: two 16-byte LDS reads per three MFMAs
        fa[(u / 3) & 3] = *reinterpret_cast<const h8*>(&lds[((threadIdx.x + 64 * u + it) * 8) & 8191]);
        fb[(u / 3 + 1) & 3] = *reinterpret_cast<const h8*>(&lds[((threadIdx.x + 64 * u + 32 + it) * 8) & 8191]);
      }
      acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[u & 3], fb[(u + (u >> 2)) & 3], acc[u & 3], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F> float time_ms(F f, int reps) {
  hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(s); for (int i = 0; i < reps; ++i) f(); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
  float ms; (void)hipEventElapsedTime(&ms, s, e); return ms / reps;
}
int main() {
  _Float16* in; float* out;
  (void)hipMalloc(&in, 16384 * 2); (void)hipMalloc(&out, 4 * 256 * 1024);
  std::vector<_Float16> h(16384);
  for (auto& x : h) x = (_Float16)(((float)rand() / (float)RAND_MAX) * 2.f - 1.f);
  (void)hipMemcpy(in, h.data(), 16384 * 2, hipMemcpyHostToDevice);
  for (int ldsr = 0; ldsr < 2; ++ldsr)
    for (int wgs_per_cu : {1, 2})
      for (int iters : {512, 4096, 32768}) {
        const int grid = 256 * wgs_per_cu;
        const double flop = (double)grid * 4 * iters * 12 * 32768.0;
        const int reps = iters >= 32768 ? 3 : 10;
        float t = ldsr ? time_ms([&] { hipLaunchKernelGGL((mfma_loop<1>), dim3(grid), dim3(256), 0, 0, in, out, iters); }, reps)
                       : time_ms([&] { hipLaunchKernelGGL((mfma_loop<0>), dim3(grid), dim3(256), 0, 0, in, out, iters); }, reps);
        printf("%s, %d wave(s) per SIMD, %6d x 12 MFMAs per wave: %9.1f us  %7.1f TFLOP/s = %.3f of 2500\n",
               ldsr ? "MFMA + 2 ds_read_b128 per 3" : "pure MFMA                  ", wgs_per_cu, iters, t * 1e3, flop / t / 1e9, flop / t / 1e9 / 2500.0);
      }
  return 0;
}
