// Small-batch fully-connected layers: out[b][n] = act(sum_k x[b][k] W[k][n] + bias[n]).
// VGG fc6/fc7/fc8 at B <= 8 (models/CNN/vgg.py:198-214) and the per-image fold of the
// global-feature block of sdfprediction/fold2/conv1 (models/sdfnet.py:78-84).
//
// Bound: HBM (weight read): fc6 alone is 411 MB per forward.  Each wave streams whole
// 1-KiB row segments of W (64 lanes x float4, coalesced), four waves of a block take
// interleaved rows, K is split across blockIdx.y; partials are combined by
// splitk_reduce_kernel (deterministic, no atomics).  x[b][k] is wave-uniform -> scalar loads.
#include "kernels.hpp"

namespace disn {

template <int NB>
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ x, int K,
                                                   const float* __restrict__ w, int N,
                                                   float* __restrict__ partial, int Btot, int b0) {
  __shared__ float red[4][NB][256];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int S = gridDim.y, z = blockIdx.y;
  const int kbeg = (int)(((long)K * z) / S), kend = (int)(((long)K * (z + 1)) / S);
  const int col = blockIdx.x * 256 + lane * 4;
  float4 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* wp = w + col;
#pragma unroll 8
  for (int k = kbeg + wave; k < kend; k += 4) {
    // streamed once: non-temporal, so the 411 MB of fc6 do not evict the activations from L2 / MALL
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f wt = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(wp + (size_t)k * N));
    const float4 wv = make_float4(wt[0], wt[1], wt[2], wt[3]);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float xv = x[(size_t)(b0 + b) * K + k];
      acc[b].x += xv * wv.x; acc[b].y += xv * wv.y; acc[b].z += xv * wv.z; acc[b].w += xv * wv.w;
    }
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    red[wave][b][lane * 4 + 0] = acc[b].x;
    red[wave][b][lane * 4 + 1] = acc[b].y;
    red[wave][b][lane * 4 + 2] = acc[b].z;
    red[wave][b][lane * 4 + 3] = acc[b].w;
  }
  __syncthreads();
  const int t = threadIdx.x;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const float v = (red[0][b][t] + red[1][b][t]) + (red[2][b][t] + red[3][b][t]);
    partial[((size_t)z * Btot + b0 + b) * N + blockIdx.x * 256 + t] = v;
  }
}

int gemv_splits(int K, int N) {
  const int colblocks = N / 256;
  int s = (2048 + colblocks - 1) / colblocks;
  const int smax = K / 32 > 0 ? K / 32 : 1;
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  return s;
}

size_t gemv_ws_bytes(int B, int K, int N) {
  return (size_t)gemv_splits(K, N) * B * N * sizeof(float);
}

hipError_t gemv_launch(const float* x, int B, int K, const float* w_kn, const float* bias, int N,
                       int relu, float* out, float* ws, hipStream_t st) {
  const int S = gemv_splits(K, N);
  dim3 grid(N / 256, S);
  for (int b0 = 0; b0 < B; b0 += 8) {
    const int nb = (B - b0) < 8 ? (B - b0) : 8;
    switch (nb) {
#define DISN_GEMV_CASE(NB)                                                                  \
  case NB:                                                                                  \
    hipLaunchKernelGGL((gemv_kernel<NB>), grid, dim3(256), 0, st, x, K, w_kn, N, ws, B, b0); \
    break;
      DISN_GEMV_CASE(1) DISN_GEMV_CASE(2) DISN_GEMV_CASE(3) DISN_GEMV_CASE(4)
      DISN_GEMV_CASE(5) DISN_GEMV_CASE(6) DISN_GEMV_CASE(7) DISN_GEMV_CASE(8)
#undef DISN_GEMV_CASE
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return splitk_reduce_launch(ws, S, B, N, bias, 0, relu, out, N, st);
}

}  // namespace disn
