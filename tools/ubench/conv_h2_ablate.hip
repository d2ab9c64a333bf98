// Ablation of the conv_h2 kernel (includes the product source, instantiates ABL != 0 variants the library never
// builds).  Times per variant on VGG shapes.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCH2_UBENCH
//   -I disn_amd/csrc tools/ubench/conv_h2_ablate.hip -o tools/ubench/conv_h2_ablate
#include "../../disn_amd/csrc/conv_h2.hip"
#include <cstdio>
#include <vector>
using namespace disn;

template <typename F> float time_us(F f, int reps) {
  hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
  f(); f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(s); for (int i = 0; i < reps; ++i) f(); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
  float ms; (void)hipEventElapsedTime(&ms, s, e); return ms * 1e3f / reps;
}

template <int MB, int NW, int SEG, int TW, int D, int WK, int ABL>
float run(ConvH2Dev d) {
  constexpr int TH = MB * (32 / SEG);
  d.tiles_x = (d.W + TW - 1) / TW; d.tiles_y = (d.H + TH - 1) / TH;
  const int grid = d.B * d.tiles_x * d.tiles_y * (d.Cout / (32 * NW));
  return time_us([&] { hipLaunchKernelGGL((conv_h2_kernel<MB, NW, SEG, TW, D, WK, ABL>), dim3(grid), dim3(64 * WK * NW), 0, 0, d); }, 20);
}

template <int MB, int NW, int SEG, int TW, int D, int WK>
void sweep(const char* name, ConvH2Dev d) {
  printf("%-10s <%d,%d,%d,%d,%d,%d> full %6.1f | -B %6.1f | -halo %6.1f | -split %6.1f | -B-halo-split %6.1f | -Aread %6.1f | -all %6.1f us\n", name,
         MB, NW, SEG, TW, D, WK, run<MB, NW, SEG, TW, D, WK, 0>(d), run<MB, NW, SEG, TW, D, WK, 1>(d), run<MB, NW, SEG, TW, D, WK, 2>(d),
         run<MB, NW, SEG, TW, D, WK, 4>(d), run<MB, NW, SEG, TW, D, WK, 7>(d), run<MB, NW, SEG, TW, D, WK, 8>(d),
         run<MB, NW, SEG, TW, D, WK, 15>(d));
  fflush(stdout);
}

int main() {
  const size_t nin = 224 * 224 * 64, nw = (size_t)512 * 9 * 512 * 4 + 256;
  float *in, *out, *pool, *bias, *amax; unsigned char* w;
  (void)hipMalloc(&in, nin * 4); (void)hipMalloc(&out, nin * 4); (void)hipMalloc(&pool, nin); (void)hipMalloc(&bias, 2048);
  (void)hipMalloc(&amax, 1024); (void)hipMalloc(&w, nw);
  std::vector<float> h(nin, 0.5f);
  (void)hipMemcpy(in, h.data(), nin * 4, hipMemcpyHostToDevice);
  (void)hipMemset(w, 0x3c, nw); (void)hipMemset(bias, 0, 2048);
  std::vector<float> am(256, 1.0f);
  (void)hipMemcpy(amax, am.data(), 1024, hipMemcpyHostToDevice);
  float one[2] = {1.f, 1.f};
  auto dev = [&](int hw, int cin, int cout) {
    ConvH2Dev d{}; d.in = in; d.wimg = w; d.bias = bias; d.in_amax = amax; d.out = out; d.pool_out = nullptr; d.out_amax = amax + 64;
    d.B = 1; d.H = hw; d.W = hw; d.Cin = cin; d.Cout = cout; d.relu = 1;
    (void)hipMemcpy(w + (size_t)cin * 9 * cout * 4, one, 8, hipMemcpyHostToDevice);
    return d;
  };
  sweep<2, 1, 32, 28, 9, 8>("conv4_2", dev(28, 512, 512));
  sweep<2, 1, 32, 28, 9, 4>("conv4_2", dev(28, 512, 512));
  sweep<1, 1, 16, 14, 9, 8>("conv5_x", dev(14, 512, 512));
  sweep<1, 1, 16, 14, 9, 4>("conv5_x", dev(14, 512, 512));
  sweep<2, 2, 32, 28, 9, 4>("conv3_2", dev(56, 256, 256));
  sweep<4, 2, 16, 16, 3, 4>("conv2_2", dev(112, 128, 128));
  sweep<4, 2, 16, 16, 3, 4>("conv1_2", dev(224, 64, 64));
  return 0;
}
