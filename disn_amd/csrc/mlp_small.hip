// The two thin ends of the point MLPs (models/sdfnet.py:71-72,173-174 and :88,:186):
//   fold1/conv1 : 3 -> 64 (+ReLU), K=3 is too thin for MFMA -> VALU, both streams at once
//   fold2/conv5 : 256 -> 1 linear, N=1 -> a wave-level dot product, both streams, + the sum
//                 pred_sdf = global + local (models/model_normalization.py:204)
#include "kernels.hpp"

namespace disn {

__global__ __launch_bounds__(256) void pt_embed_kernel(const float* __restrict__ pts, int64_t M,
                                                       const float* __restrict__ g_w1,
                                                       const float* __restrict__ g_b1,
                                                       const float* __restrict__ l_w1,
                                                       const float* __restrict__ l_b1,
                                                       float* __restrict__ out_g,
                                                       float* __restrict__ out_l,
                                                       float* __restrict__ amax_gl,
                                                       float* __restrict__ zero, int nzero, int images) {
  // side job of the first launch of a point-MLP chain (dense_h2.hip layers): clear the activation-maximum slots
  // of the layers behind it (slots 128..1023 of every image's set) and the all-zero bias row
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nzero; i += gridDim.x * blockDim.x) zero[i] = 0.f;
  // with maxima: gridDim.x = images * G workgroups, workgroup (img, iw) walks the rows of image img only
  const int G = amax_gl ? gridDim.x / images : gridDim.x;
  const int img = amax_gl ? blockIdx.x / G : 0, iw = amax_gl ? blockIdx.x - img * G : blockIdx.x;
  const int64_t rows = amax_gl ? M / images : M;
  if (amax_gl)
    for (int i = iw * blockDim.x + threadIdx.x; i < 896; i += G * blockDim.x) amax_gl[(size_t)img * 1024 + 128 + i] = 0.f;
  // thread -> (point, 4 channels of one stream): 32 threads per point (16 per stream)
  const int64_t total = rows * 32;
  float mx = 0.f;
  for (int64_t i = (int64_t)iw * blockDim.x + threadIdx.x; i < total; i += (int64_t)G * blockDim.x) {
    const int64_t m = (int64_t)img * rows + (i >> 5);
    const int q = (int)(i & 31);
    const bool local = q >= 16;
    const int c = (q & 15) * 4;
    const float* w = local ? l_w1 : g_w1;
    const float* bb = local ? l_b1 : g_b1;
    const float x = pts[m * 3], y = pts[m * 3 + 1], z = pts[m * 3 + 2];
    const float4 w0 = *reinterpret_cast<const float4*>(w + c);
    const float4 w1 = *reinterpret_cast<const float4*>(w + 64 + c);
    const float4 w2 = *reinterpret_cast<const float4*>(w + 128 + c);
    const float4 b4 = *reinterpret_cast<const float4*>(bb + c);
    float4 o;
    o.x = fmaxf(x * w0.x + y * w1.x + z * w2.x + b4.x, 0.f);
    o.y = fmaxf(x * w0.y + y * w1.y + z * w2.y + b4.y, 0.f);
    o.z = fmaxf(x * w0.z + y * w1.z + z * w2.z + b4.z, 0.f);
    o.w = fmaxf(x * w0.w + y * w1.w + z * w2.w + b4.w, 0.f);
    *reinterpret_cast<float4*>((local ? out_l : out_g) + m * 64 + c) = o;
    mx = fmaxf(fmaxf(fmaxf(mx, o.x), o.y), fmaxf(o.z, o.w));
  }
  if (amax_gl) {
    // the maximum of each stream's output, one slot per workgroup (gridDim.x <= 64; the launcher clears nothing:
    // slots of absent workgroups are written 0 by workgroup 0): amax_gl[0..63] global, [64..127] local.
    // blockDim.x * gridDim.x is a multiple of 32, so a thread serves ONE stream for all its points.
    __shared__ float red[2][4];
    const bool local = (threadIdx.x & 31) >= 16;
    float mg = local ? 0.f : mx, ml = local ? mx : 0.f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      mg = fmaxf(mg, __shfl_xor(mg, off));
      ml = fmaxf(ml, __shfl_xor(ml, off));
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = mg; red[1][threadIdx.x >> 6] = ml; }
    __syncthreads();
    if (threadIdx.x < 2) {
      const int st = threadIdx.x;
      float* a = amax_gl + (size_t)img * 1024 + 64 * st;
      a[iw] = fmaxf(fmaxf(red[st][0], red[st][1]), fmaxf(red[st][2], red[st][3]));
      if (iw == 0)
        for (int i = G; i < 64; ++i) a[i] = 0.f;
    }
  }
}

hipError_t pt_embed_launch(const float* pts, int64_t M, const float* g_w1, const float* g_b1,
                           const float* l_w1, const float* l_b1, float* out_g, float* out_l,
                           hipStream_t st, float* amax_gl, float* zero, int nzero, int images) {
  if (images < 1 || !amax_gl) images = 1;
  int64_t blocks = (M / images * 32 + 255) / 256;  // per image
  if (blocks > 16384) blocks = 16384;
  if (amax_gl && blocks > 64) blocks = 64;  // one maximum slot per workgroup
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(pt_embed_kernel, dim3((unsigned)(blocks * images)), dim3(256), 0, st, pts, M, g_w1, g_b1,
                     l_w1, l_b1, out_g, out_l, amax_gl, zero, nzero, images);
  return hipGetLastError();
}

// one wave per point row: lane holds float4 of the 256-wide activations of each stream
__global__ __launch_bounds__(256) void final_dot_kernel(const float* __restrict__ g5,
                                                        const float* __restrict__ l5, int64_t M,
                                                        const float* __restrict__ g_w6,
                                                        const float* __restrict__ g_b6,
                                                        const float* __restrict__ l_w6,
                                                        const float* __restrict__ l_b6,
                                                        float* __restrict__ sdf,
                                                        float* __restrict__ sdf_g,
                                                        float* __restrict__ sdf_l, float out_div) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const float4 wg = *reinterpret_cast<const float4*>(g_w6 + lane * 4);
  const float4 wl = *reinterpret_cast<const float4*>(l_w6 + lane * 4);
  const float bg = g_b6[0], bl = l_b6[0];
  for (int64_t m = wave; m < M; m += nwaves) {
    const float4 a = *reinterpret_cast<const float4*>(g5 + m * 256 + lane * 4);
    const float4 b = *reinterpret_cast<const float4*>(l5 + m * 256 + lane * 4);
    float vg = (a.x * wg.x + a.y * wg.y) + (a.z * wg.z + a.w * wg.w);
    float vl = (b.x * wl.x + b.y * wl.y) + (b.z * wl.z + b.w * wl.w);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      vg += __shfl_xor(vg, off);
      vl += __shfl_xor(vl, off);
    }
    if (lane == 0) {
      vg += bg;
      vl += bl;
      if (sdf_g) sdf_g[m] = vg;
      if (sdf_l) sdf_l[m] = vl;
      sdf[m] = (vg + vl) / out_div;
    }
  }
}

hipError_t final_dot_launch(const float* g5, const float* l5, int64_t M, const float* g_w6,
                            const float* g_b6, const float* l_w6, const float* l_b6, float* sdf,
                            float* sdf_g, float* sdf_l, float out_div, hipStream_t st) {
  int64_t blocks = (M + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(final_dot_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g5, l5, M, g_w6,
                     g_b6, l_w6, l_b6, sdf, sdf_g, sdf_l, out_div);
  return hipGetLastError();
}

}  // namespace disn
