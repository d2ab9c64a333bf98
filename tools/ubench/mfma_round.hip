// How v_mfma_f32_32x32x16_f16 rounds its fp32 accumulation (not part of the product): a chain of L MFMAs on one
// accumulator against the exact sum in double; prints the SIGNED mean error in units of the result's ulp (a value
// near 0 = round to nearest, unbiased; near -0.5 L ... = truncation) and the rms, for positive and for signed products;
// and the same sums with the chain restarted every S MFMAs and the segments added in fp32 VALU (conv_h2w's SEG form).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_round mfma_round.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A: [32][16 L] f16 row-major, B: [16 L][32] given as Bt [32][16 L]; lane (j, g): a = A[j][16 k + 8 g ..], b = Bt[j][16 k + 8 g ..]
__global__ void chain(const _Float16* A, const _Float16* Bt, float* out, int L, int S) {
  const int lane = threadIdx.x, j = lane & 31, g = lane >> 5;
  f32x16 acc, tot;
  for (int r = 0; r < 16; ++r) acc[r] = tot[r] = 0.f;
  for (int k = 0; k < L; ++k) {
    const h8 a = *reinterpret_cast<const h8*>(A + (size_t)j * 16 * L + 16 * k + 8 * g);
    const h8 b = *reinterpret_cast<const h8*>(Bt + (size_t)j * 16 * L + 16 * k + 8 * g);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (S > 0 && (k + 1) % S == 0) {
      for (int r = 0; r < 16; ++r) { tot[r] += acc[r]; acc[r] = 0.f; }
    }
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * g;   // C layout: lane (j, g) register r -> row, column j
    out[row * 32 + j] = S > 0 ? tot[r] + acc[r] : acc[r];
  }
}

int main() {
  for (int sign = 0; sign < 2; ++sign)
    for (int L : {27, 54, 108, 216, 432}) {
      std::vector<_Float16> A((size_t)32 * 16 * L), Bt((size_t)32 * 16 * L);
      srand(7 + L);
      auto rnd = [&]() { float u = (float)rand() / RAND_MAX; return sign ? 2.f * u - 1.f : u; };
      for (auto& x : A) x = (_Float16)rnd();
      for (auto& x : Bt) x = (_Float16)rnd();
      _Float16 *dA, *dB; float* dO;
      hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, Bt.size() * 2); hipMalloc(&dO, 4096);
      hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
      hipMemcpy(dB, Bt.data(), Bt.size() * 2, hipMemcpyHostToDevice);
      std::vector<double> ex(1024);
      for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
          double s = 0;
          for (int k = 0; k < 16 * L; ++k) s += (double)(float)A[(size_t)m * 16 * L + k] * (double)(float)Bt[(size_t)n * 16 * L + k];
          ex[m * 32 + n] = s;
        }
      for (int S : {0, 1, 9, 27, 54}) {
        if (S > L) continue;
        hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, dA, dB, dO, L, S);
        std::vector<float> o(1024);
        hipMemcpy(o.data(), dO, 4096, hipMemcpyDeviceToHost);
        double mean = 0, rms = 0, rel = 0;
        for (int i = 0; i < 1024; ++i) {
          const double ulp = std::ldexp(1.0, std::ilogb(std::fabs(ex[i]) + 1e-300) - 23);
          const double e = ((double)o[i] - ex[i]) / ulp;
          mean += e; rms += e * e; rel += std::fabs((double)o[i] - ex[i]);
        }
        printf("%s products, chain %3d, restart every %2d: signed mean error %+8.3f ulp, rms %7.3f ulp\n", sign ? "signed  " : "positive", L, S,
               mean / 1024, std::sqrt(rms / 1024));
      }
      hipFree(dA); hipFree(dB); hipFree(dO);
    }
  return 0;
}
