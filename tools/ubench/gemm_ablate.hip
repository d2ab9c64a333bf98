// Ablation of the product GEMM kernel (includes the product source; instantiates ABL != 0
// variants that the library never builds).  Prints time per variant on VGG conv shapes.
#include "../../disn_amd/csrc/gemm_mfma.hip"
#include <cstdio>
#include <vector>
using namespace disn;

template <typename F> float time_us(F f, int reps) {
  hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
  f(); f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(s); for (int i = 0; i < reps; ++i) f(); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
  float ms; (void)hipEventElapsedTime(&ms, s, e); return ms * 1e3f / reps;
}

template <int BM, int BN, int ABL>
float run(const GemmParams& p, int S, float* ws) {   // S > 0: W = tiles*S (split-K); S < 0: W = -S (stream-K)
  GemmDev d; d.p = p; d.ws = ws; d.ksteps = p.K / 32; d.mtiles = (p.M + BM - 1) / BM; d.ntiles = p.N / BN;
  d.W = S > 0 ? d.mtiles * d.ntiles * S : -S; d.units = (long)d.mtiles * d.ntiles * d.ksteps; d.dbg = nullptr;
  dim3 grid(d.W);
  return time_us([&] { hipLaunchKernelGGL((gemm_f32_mfma<BM, BN, GEMM_CONV3, ABL>), grid, dim3(256), 0, 0, d); }, 10);
}

template <int BM, int BN>
void sweep(const char* name, int H, int Cin, int Cout, int S, float* in, float* bp, float* bias, float* out, float* ws) {
  GemmParams p{}; p.a1 = in; p.H = H; p.W = H; p.Cin = Cin; p.M = H * H; p.N = Cout; p.K = 9 * Cin;
  p.bp = bp; p.bias = bias; p.out = out; p.ldc = Cout; p.relu = 1;
  const double gf = 2.0 * H * H * Cout * 9.0 * Cin / 1e9;
  const int wgs = S > 0 ? ((p.M + BM - 1) / BM) * (Cout / BN) * S : -S;
  printf("%-18s tile %3dx%-3d S=%-2d wgs %5d | full %6.1f | -A %6.1f | -B %6.1f | -A-B %6.1f | -sync %6.1f | -dsread %6.1f | -A-B-sync %6.1f | mfma-only %6.1f us  (ideal@150TF %.1f)\n",
         name, BM, BN, S, wgs, run<BM, BN, 0>(p, S, ws), run<BM, BN, 1>(p, S, ws), run<BM, BN, 2>(p, S, ws),
         run<BM, BN, 3>(p, S, ws), run<BM, BN, 4>(p, S, ws), run<BM, BN, 8>(p, S, ws), run<BM, BN, 7>(p, S, ws),
         run<BM, BN, 15>(p, S, ws), gf / 150.0 * 1e3);
}

// phase timeline of a few workgroups (cycles, relative to the earliest start in the launch)
template <int BM, int BN>
void phases(const char* name, int H, int Cin, int Cout, int S, float* in, float* bp, float* bias, float* out, float* ws) {
  GemmParams p{}; p.a1 = in; p.H = H; p.W = H; p.Cin = Cin; p.M = H * H; p.N = Cout; p.K = 9 * Cin;
  p.bp = bp; p.bias = bias; p.out = out; p.ldc = Cout; p.relu = 1;
  GemmDev d; d.p = p; d.ws = ws; d.ksteps = p.K / 32; d.mtiles = (p.M + BM - 1) / BM; d.ntiles = p.N / BN;
  d.W = S > 0 ? d.mtiles * d.ntiles * S : -S; d.units = (long)d.mtiles * d.ntiles * d.ksteps;
  long long* dbg; (void)hipMalloc(&dbg, (size_t)d.W * 16 * 8); (void)hipMemset(dbg, 0, (size_t)d.W * 16 * 8);
  d.dbg = dbg;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((gemm_f32_mfma<BM, BN, GEMM_CONV3, 16>), dim3(d.W), dim3(256), 0, 0, d);
  (void)hipDeviceSynchronize();
  std::vector<long long> h((size_t)d.W * 16);
  (void)hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
  long long t0 = h[0], tend = 0;
  for (int w = 0; w < d.W; ++w) { if (h[w * 16] < t0) t0 = h[w * 16]; int n = (int)h[w * 16 + 15]; if (n > 0 && h[w * 16 + n - 1] > tend) tend = h[w * 16 + n - 1]; }
  printf("%s tile %dx%d W=%d: span %lld cycles; per-WG [start | (prologue_end, kloop_end, epilogue_end) per segment]\n", name, BM, BN, d.W, tend - t0);
  for (int w : {0, 1, 7, 100, d.W / 2, d.W - 1}) {
    if (w >= d.W) continue;
    int n = (int)h[w * 16 + 15];
    printf("  wg %4d:", w);
    for (int k = 0; k < n; ++k) printf(" %lld", h[w * 16 + k] - t0);
    printf("\n");
  }
  (void)hipFree(dbg);
}

int main() {
  float *in, *bp, *bias, *out, *ws;
  (void)hipMalloc(&in, 64u << 20); (void)hipMalloc(&bp, 64u << 20); (void)hipMalloc(&bias, 1 << 16);
  (void)hipMalloc(&out, 64u << 20); (void)hipMalloc(&ws, 512u << 20);
  std::vector<float> h(16u << 20);
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  (void)hipMemcpy(in, h.data(), 64u << 20, hipMemcpyHostToDevice);
  (void)hipMemcpy(bp, h.data(), 64u << 20, hipMemcpyHostToDevice);
  (void)hipMemset(bias, 0, 1 << 16);
  phases<128, 128>("conv2_2", 112, 128, 128, -256, in, bp, bias, out, ws);
  phases<64, 128>("conv2_2", 112, 128, 128, -512, in, bp, bias, out, ws);
  phases<64, 64>("conv2_2", 112, 128, 128, 1, in, bp, bias, out, ws);
  sweep<128, 128>("conv2_2 112 128>128", 112, 128, 128, -256, in, bp, bias, out, ws);
  sweep<128, 128>("conv2_2 112 128>128", 112, 128, 128, -512, in, bp, bias, out, ws);
  sweep<64, 128>("conv2_2 112 128>128", 112, 128, 128, -512, in, bp, bias, out, ws);
  sweep<64, 128>("conv2_2 112 128>128", 112, 128, 128, 1, in, bp, bias, out, ws);
  sweep<64, 64>("conv2_2 112 128>128", 112, 128, 128, 1, in, bp, bias, out, ws);
  sweep<128, 128>("conv2_2 112 128>128", 112, 128, 128, 4, in, bp, bias, out, ws);
  sweep<128, 64>("conv1_2 224 64>64", 224, 64, 64, 1, in, bp, bias, out, ws);
  sweep<64, 64>("conv1_2 224 64>64", 224, 64, 64, 1, in, bp, bias, out, ws);
  sweep<64, 64>("conv3_2 56 256>256", 56, 256, 256, 6, in, bp, bias, out, ws);
  sweep<128, 128>("conv3_2 56 256>256", 56, 256, 256, 4, in, bp, bias, out, ws);
  sweep<128, 128>("conv4_2 28 512>512", 28, 512, 512, 9, in, bp, bias, out, ws);
  sweep<64, 64>("conv5_2 14 512>512", 14, 512, 512, 8, in, bp, bias, out, ws);
  return 0;
}
