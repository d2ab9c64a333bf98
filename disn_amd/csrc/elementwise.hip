// Element-wise / gather rows of the path.  THIS FILE IS COMPILED WITH
// -ffp-contract=off: every multiply and add is rounded separately, in the
// operation order of the oracle (oracle/disn_oracle.py), so these kernels are
// bit-exact with it.  All of them are HBM/L2-bandwidth bound; the rules that
// matter are coalescing (16 B per lane, channel-contiguous NHWC) and enough
// workgroups to fill 256 CUs.
//
//   resize_bilinear  -- tf.image.resize_bilinear legacy  (models/model_normalization.py:72,171-183)
//   project          -- get_img_points                   (models/model_normalization.py:241-251)
//   gather           -- 5 x contrib.resampler + concat   (models/model_normalization.py:172-190)
//   maxpool2x2       -- slim.max_pool2d                  (models/CNN/vgg.py:188-196)
//   grid_points      -- linspace/meshgrid grid           (test/create_sdf.py:246-256)
#include "kernels.hpp"
#include "tuning.hpp"

namespace disn {

#define DISN_FEAT 1472
#define DISN_FEAT4 368
#define DISN_IMG 137

// ---------------------------------------------------------------------------
// legacy bilinear resize
// ---------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void resize_kernel(const float* __restrict__ in, int B, int Hin,
                                                     int Win, int C, float* __restrict__ out,
                                                     int Hout, int Wout, int out_cstride,
                                                     int out_coff, float sy, float sx,
                                                     float* __restrict__ zero, int nzero) {
  // side job of the encoder's first launch: clear the activation-maximum slots of the convolution stack
  // (api.hip vgg_features) -- saves a memset launch and the stream bubble around it
  // (spread over the launch: one workgroup clearing the 14 x 64 slots of SIXTEEN images alone was most of the launch's 14 us)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)nzero; i += (size_t)gridDim.x * blockDim.x) zero[i] = 0.f;
  const int cv = C / VEC;
  const size_t total = (size_t)B * Hout * Wout * cv;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VEC;
    size_t pidx = i / cv;
    const int ox = (int)(pidx % Wout);
    pidx /= Wout;
    const int oy = (int)(pidx % Hout);
    const int b = (int)(pidx / Hout);
    const float fy = (float)oy * sy, fx = (float)ox * sx;
    const int ylo = (int)floorf(fy), xlo = (int)floorf(fx);
    const int yhi = min(ylo + 1, Hin - 1), xhi = min(xlo + 1, Win - 1);
    const float yl = fy - (float)ylo, xl = fx - (float)xlo;
    const float* base = in + (size_t)b * Hin * Win * C + c;
    const float* ptl = base + ((size_t)ylo * Win + xlo) * C;
    const float* ptr = base + ((size_t)ylo * Win + xhi) * C;
    const float* pbl = base + ((size_t)yhi * Win + xlo) * C;
    const float* pbr = base + ((size_t)yhi * Win + xhi) * C;
    float* po = out + ((size_t)(b * Hout + oy) * Wout + ox) * out_cstride + out_coff + c;
    if (VEC == 4) {
      const float4 tl = *reinterpret_cast<const float4*>(ptl);
      const float4 tr = *reinterpret_cast<const float4*>(ptr);
      const float4 bl = *reinterpret_cast<const float4*>(pbl);
      const float4 br = *reinterpret_cast<const float4*>(pbr);
      float4 o;
#define DISN_LERP(f)                              \
  {                                               \
    const float top = tl.f + (tr.f - tl.f) * xl;  \
    const float bot = bl.f + (br.f - bl.f) * xl;  \
    o.f = top + (bot - top) * yl;                 \
  }
      DISN_LERP(x) DISN_LERP(y) DISN_LERP(z) DISN_LERP(w)
#undef DISN_LERP
      *reinterpret_cast<float4*>(po) = o;
    } else {
      const float top = *ptl + (*ptr - *ptl) * xl;
      const float bot = *pbl + (*pbr - *pbl) * xl;
      *po = top + (bot - top) * yl;
    }
  }
}

static inline int grid_for(size_t total, int cap = 8192) {
  size_t b = (total + 255) / 256;
  if (b > (size_t)cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

hipError_t resize_bilinear_launch(const float* in, int B, int Hin, int Win, int C, float* out,
                                  int Hout, int Wout, int out_cstride, int out_coff,
                                  hipStream_t st, int max_blocks, float* zero, int nzero) {
  const int cap = max_blocks > 0 ? max_blocks : 8192;
  const float sy = (float)Hin / (float)Hout;  // CalculateResizeScale, float32 division
  const float sx = (float)Win / (float)Wout;
  const bool vec = (C % 4 == 0) && (out_cstride % 4 == 0) && (out_coff % 4 == 0);
  if (vec) {
    const size_t total = (size_t)B * Hout * Wout * (C / 4);
    hipLaunchKernelGGL((resize_kernel<4>), dim3(grid_for(total, cap)), dim3(256), 0, st, in, B, Hin, Win,
                       C, out, Hout, Wout, out_cstride, out_coff, sy, sx, zero, nzero);
  } else {
    const size_t total = (size_t)B * Hout * Wout * C;
    hipLaunchKernelGGL((resize_kernel<1>), dim3(grid_for(total, cap)), dim3(256), 0, st, in, B, Hin, Win,
                       C, out, Hout, Wout, out_cstride, out_coff, sy, sx, zero, nzero);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// 2x2 stride-2 VALID max pool, NHWC, C % 4 == 0
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ in, int B, int H,
                                                      int W, int C, float* __restrict__ out) {
  const int Ho = H / 2, Wo = W / 2, c4n = C / 4;
  const size_t total = (size_t)B * Ho * Wo * c4n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    size_t pidx = i / c4n;
    const int ox = (int)(pidx % Wo);
    pidx /= Wo;
    const int oy = (int)(pidx % Ho);
    const int b = (int)(pidx / Ho);
    const float* p = in + (((size_t)b * H + 2 * oy) * W + 2 * ox) * C + c;
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b2 = *reinterpret_cast<const float4*>(p + C);
    const float4 c2 = *reinterpret_cast<const float4*>(p + (size_t)W * C);
    const float4 d = *reinterpret_cast<const float4*>(p + (size_t)W * C + C);
    float4 o;
    o.x = fmaxf(fmaxf(a.x, b2.x), fmaxf(c2.x, d.x));
    o.y = fmaxf(fmaxf(a.y, b2.y), fmaxf(c2.y, d.y));
    o.z = fmaxf(fmaxf(a.z, b2.z), fmaxf(c2.z, d.z));
    o.w = fmaxf(fmaxf(a.w, b2.w), fmaxf(c2.w, d.w));
    *reinterpret_cast<float4*>(out + (((size_t)b * Ho + oy) * Wo + ox) * C + c) = o;
  }
}

hipError_t maxpool2x2_launch(const float* in, int B, int H, int W, int C, float* out,
                             hipStream_t st) {
  const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total)), dim3(256), 0, st, in, B, H, W, C, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// projection  ((x*T0 + y*T1) + z*T2) + T3 ; xy / z ; clamp [0,136], NaN propagates
// ---------------------------------------------------------------------------
__device__ __forceinline__ float clamp_px(float v) {
  return (v != v) ? v : fminf(136.0f, fmaxf(0.0f, v));
}

__device__ __forceinline__ void project_point(const float* __restrict__ T, float x, float y,
                                              float z, float& px, float& py) {
  float p[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float a = x * T[0 * 3 + j] + y * T[1 * 3 + j];
    a = a + z * T[2 * 3 + j];
    p[j] = a + T[3 * 3 + j];
  }
  px = clamp_px(p[0] / p[2]);
  py = clamp_px(p[1] / p[2]);
}

__global__ __launch_bounds__(256) void project_kernel(const float* __restrict__ pts,
                                                      const float* __restrict__ trans_mat, int B,
                                                      int N, float* __restrict__ xy) {
  const size_t total = (size_t)B * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / N);
    float px, py;
    project_point(trans_mat + (size_t)b * 12, pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2], px, py);
    xy[i * 2] = px;
    xy[i * 2 + 1] = py;
  }
}

hipError_t project_launch(const float* pts, const float* trans_mat, int B, int N, float* xy,
                          hipStream_t st) {
  hipLaunchKernelGGL(project_kernel, dim3(grid_for((size_t)B * N)), dim3(256), 0, st, pts, trans_mat,
                     B, N, xy);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// gather: one thread per (point, float4 of the 1472 channels).  Consecutive lanes read
// consecutive 16-byte pieces of the same pixel (5888 contiguous bytes per tap pixel) and
// write consecutive 16-byte pieces of the output row: every access is a full-line stream.
// The two x-taps (fx, fx+1) of a row are adjacent pixels = 11776 contiguous bytes.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float4 sample4(const float* __restrict__ map, float x, float y, int c) {
  // TF-1.10 resampler functor: zero outside, weights from the ceil corner.
  const bool ok = x > -1.0f && y > -1.0f && x < (float)DISN_IMG && y < (float)DISN_IMG;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!ok) return o;
  const float fx = floorf(x), fy = floorf(y);
  const float cx = fx + 1.0f, cy = fy + 1.0f;
  const float dx = cx - x, dy = cy - y;
  const int ifx = (int)fx, ify = (int)fy, icx = (int)cx, icy = (int)cy;
  const float w_ff = dx * dy;
  const float w_cc = (1.0f - dx) * (1.0f - dy);
  const float w_fc = dx * (1.0f - dy);
  const float w_cf = (1.0f - dx) * dy;
  const bool xf = ifx >= 0 && ifx < DISN_IMG, xc = icx >= 0 && icx < DISN_IMG;
  const bool yf = ify >= 0 && ify < DISN_IMG, yc = icy >= 0 && icy < DISN_IMG;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 v_ff = (xf && yf) ? *reinterpret_cast<const float4*>(
                                       map + ((size_t)ify * DISN_IMG + ifx) * DISN_FEAT + c)
                                 : z4;
  const float4 v_cc = (xc && yc) ? *reinterpret_cast<const float4*>(
                                       map + ((size_t)icy * DISN_IMG + icx) * DISN_FEAT + c)
                                 : z4;
  const float4 v_fc = (xf && yc) ? *reinterpret_cast<const float4*>(
                                       map + ((size_t)icy * DISN_IMG + ifx) * DISN_FEAT + c)
                                 : z4;
  const float4 v_cf = (xc && yf) ? *reinterpret_cast<const float4*>(
                                       map + ((size_t)ify * DISN_IMG + icx) * DISN_FEAT + c)
                                 : z4;
#define DISN_ACC(f)             \
  {                             \
    float v = w_ff * v_ff.f;    \
    v = v + w_cc * v_cc.f;      \
    v = v + w_fc * v_fc.f;      \
    v = v + w_cf * v_cf.f;      \
    o.f = v;                    \
  }
  DISN_ACC(x) DISN_ACC(y) DISN_ACC(z) DISN_ACC(w)
#undef DISN_ACC
  return o;
}

__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ featmap,
                                                     const float* __restrict__ xy, int B, int N,
                                                     float* __restrict__ feat) {
  const size_t total = (size_t)B * N * DISN_FEAT4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t pt = i / DISN_FEAT4;
    const int c = (int)(i - pt * DISN_FEAT4) * 4;
    const int b = (int)(pt / N);
    const float x = xy[pt * 2], y = xy[pt * 2 + 1];
    const float4 o = sample4(featmap + (size_t)b * DISN_IMG * DISN_IMG * DISN_FEAT, x, y, c);
    *reinterpret_cast<float4*>(feat + pt * DISN_FEAT + c) = o;
  }
}

hipError_t gather_launch(const float* featmap, const float* xy, int B, int N, float* feat,
                         hipStream_t st) {
  const size_t total = (size_t)B * N * DISN_FEAT4;
  hipLaunchKernelGGL(gather_kernel, dim3(grid_for(total, 16384)), dim3(256), 0, st, featmap, xy, B,
                     N, feat);
  return hipGetLastError();
}

// project + gather fused (the xy of a point is recomputed by each of its 368 threads:
// 12 multiplies, cheaper than a round trip of xy through memory)
__global__ __launch_bounds__(256) void project_gather_kernel(const float* __restrict__ featmap_b,
                                                             const float* __restrict__ trans_mat_b,
                                                             const float* __restrict__ pts, int n,
                                                             float* __restrict__ feat) {
  const size_t total = (size_t)n * DISN_FEAT4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t pt = i / DISN_FEAT4;
    const int c = (int)(i - pt * DISN_FEAT4) * 4;
    float px, py;
    project_point(trans_mat_b, pts[pt * 3], pts[pt * 3 + 1], pts[pt * 3 + 2], px, py);
    const float4 o = sample4(featmap_b, px, py, c);
    *reinterpret_cast<float4*>(feat + pt * DISN_FEAT + c) = o;
  }
}

hipError_t project_gather_launch(const float* featmap_b, const float* trans_mat_b, const float* pts,
                                 int n, float* feat, hipStream_t st) {
  const size_t total = (size_t)n * DISN_FEAT4;
  hipLaunchKernelGGL(project_gather_kernel, dim3(grid_for(total, 16384)), dim3(256), 0, st,
                     featmap_b, trans_mat_b, pts, n, feat);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// project + gather straight from the five taps (no 110 MB feature map): every feature-map pixel
// the resampler touches is recomputed from its four tap pixels with resize_kernel's exact
// expression, so the result is bit-identical to resize -> gather.  16 tap reads per output
// float4 instead of 4 map reads, but the taps (24 MB per image) stay in L2 / MALL and nothing
// is written but the [n,1472] rows: the right trade below ~10^4 points per image, where the
// map would be written (110 MB) to be read once (29 KB per point).
// ---------------------------------------------------------------------------
struct TapSet {
  const float* p[5];
  float s[5];      // (float)Hin / (float)137, as resize_bilinear_launch computes it
  size_t stride[5];  // floats per image of each tap
};

__device__ __forceinline__ float4 tap_pixel(const float* __restrict__ tap, int hw, int ch, float s,
                                            int oy, int ox, int cl) {
  const float fy = (float)oy * s, fx = (float)ox * s;
  const int ylo = (int)floorf(fy), xlo = (int)floorf(fx);
  const int yhi = min(ylo + 1, hw - 1), xhi = min(xlo + 1, hw - 1);
  const float yl = fy - (float)ylo, xl = fx - (float)xlo;
  const float* base = tap + cl;
  const float4 tl = *reinterpret_cast<const float4*>(base + ((size_t)ylo * hw + xlo) * ch);
  const float4 tr = *reinterpret_cast<const float4*>(base + ((size_t)ylo * hw + xhi) * ch);
  const float4 bl = *reinterpret_cast<const float4*>(base + ((size_t)yhi * hw + xlo) * ch);
  const float4 br = *reinterpret_cast<const float4*>(base + ((size_t)yhi * hw + xhi) * ch);
  float4 o;
#define DISN_LERP(f)                             \
  {                                              \
    const float top = tl.f + (tr.f - tl.f) * xl; \
    const float bot = bl.f + (br.f - bl.f) * xl; \
    o.f = top + (bot - top) * yl;                \
  }
  DISN_LERP(x) DISN_LERP(y) DISN_LERP(z) DISN_LERP(w)
#undef DISN_LERP
  return o;
}

// power of two s with amax * s in [2^14, 2^15) (amax > 0 after feat_split_amax's floor); as pow2_scale_for of
// mlp_fused.hip: the two sides of the split form must pick the same scale
__device__ __forceinline__ float split_pow2_scale(float amax) {
  const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127;
  if (!(amax > 0.f) || e > 100 || e < -100) return 1.0f;
  return __uint_as_float((unsigned)(127 + 14 - e) << 23);
}

template <bool L16>
__global__ __launch_bounds__(1024) void project_gather_taps_kernel(TapSet t,
                                                                   const float* __restrict__ trans_mat,
                                                                   const float* __restrict__ pts, int B,
                                                                   int n, int c4_begin, int c4_count,
                                                                   float* __restrict__ feat, int feat_ld,
                                                                   float* __restrict__ amax, size_t amax_stride,
                                                                   const float* __restrict__ split_amax) {
  // B images x n points each (rows image-major); tap k of image b at t.p[k] + b * t.stride[k].  feat_ld > 1472
  // (all five taps only): rows of feat_ld floats, columns 1472 .. feat_ld - 1 written as zeros (c4_count covers
  // them) -- a zero-padded K for a GEMM that wants 256-column chunks (dense_h2.hip).
  // amax != nullptr: gridDim.x = B * G, workgroup (b, iw) walks image b only and stores the maximum |feat| it wrote
  // at amax[b * amax_stride + iw] (plain store, every (b, iw) writes): the dense_h2 layer behind takes the maximum
  // over the G entries as its operand scale -- no extra pass over feat and no atomics (same-line atomics cost ~4 ns
  // EACH on this part: one per wave made the 14 us kernel a 60 us one).
  // split_amax != nullptr (round 4, the operand of mlp_fused_kernel<local, FEAT>): the row is written in SPLIT form --
  // the same fp32 value v, then x = v * s with the image's power-of-two scale s = 2^14 / 2^e(max(split_amax[b], 2^-20))
  // (split_amax[b] >= max |tap| of image b bounds every feature), h = f16(x), l = f16(x - h); every 8 channels take
  // their 32 bytes as [h8 | l8].  A thread's four channels are 8 bytes of each plane.
  const size_t per_img = (size_t)n * c4_count;
  size_t i, end, step;
  int iw = 0, bimg = 0;
  if (amax) {
    const int G = gridDim.x / B;
    bimg = blockIdx.x / G;
    iw = blockIdx.x - bimg * G;
    i = (size_t)bimg * per_img + (size_t)iw * blockDim.x + threadIdx.x;
    end = (size_t)(bimg + 1) * per_img;
    step = (size_t)G * blockDim.x;
  } else {
    i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    end = (size_t)B * per_img;
    step = (size_t)gridDim.x * blockDim.x;
  }
  float vmax = 0.f;
  for (; i < end; i += step) {
    const size_t pt = i / c4_count;
    const int c = (c4_begin + (int)(i - pt * c4_count)) * 4;
    if (c >= DISN_FEAT) {  // padding columns
      *reinterpret_cast<float4*>(feat + pt * feat_ld + c) = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    const int b = (int)(pt / n);
    float x, y;
    project_point(trans_mat + (size_t)b * 12, pts[pt * 3], pts[pt * 3 + 1], pts[pt * 3 + 2], x, y);
    const int k = c < 64 ? 0 : (c < 192 ? 1 : (c < 448 ? 2 : (c < 960 ? 3 : 4)));
    const int hw = 224 >> k;
    const int ch = k == 0 ? 64 : (k == 1 ? 128 : (k == 2 ? 256 : 512));
    const int cl = c - (k == 0 ? 0 : (k == 1 ? 64 : (k == 2 ? 192 : (k == 3 ? 448 : 960))));
    const float* tap = t.p[k] + (size_t)b * t.stride[k];
    const float s = t.s[k];
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!L16) {
    // the resampler of sample4, its four map reads replaced by tap_pixel
    const bool ok = x > -1.0f && y > -1.0f && x < (float)DISN_IMG && y < (float)DISN_IMG;
    if (ok) {
      const float fx = floorf(x), fy = floorf(y);
      const float cx = fx + 1.0f, cy = fy + 1.0f;
      const float dx = cx - x, dy = cy - y;
      const int ifx = (int)fx, ify = (int)fy, icx = (int)cx, icy = (int)cy;
      const float w_ff = dx * dy;
      const float w_cc = (1.0f - dx) * (1.0f - dy);
      const float w_fc = dx * (1.0f - dy);
      const float w_cf = (1.0f - dx) * dy;
      const bool xf = ifx >= 0 && ifx < DISN_IMG, xc = icx >= 0 && icx < DISN_IMG;
      const bool yf = ify >= 0 && ify < DISN_IMG, yc = icy >= 0 && icy < DISN_IMG;
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 v_ff = (xf && yf) ? tap_pixel(tap, hw, ch, s, ify, ifx, cl) : z4;
      const float4 v_cc = (xc && yc) ? tap_pixel(tap, hw, ch, s, icy, icx, cl) : z4;
      const float4 v_fc = (xf && yc) ? tap_pixel(tap, hw, ch, s, icy, ifx, cl) : z4;
      const float4 v_cf = (xc && yf) ? tap_pixel(tap, hw, ch, s, ify, icx, cl) : z4;
#define DISN_ACC(f)          \
  {                          \
    float v = w_ff * v_ff.f; \
    v = v + w_cc * v_cc.f;   \
    v = v + w_fc * v_fc.f;   \
    v = v + w_cf * v_cf.f;   \
    o.f = v;                 \
  }
      DISN_ACC(x) DISN_ACC(y) DISN_ACC(z) DISN_ACC(w)
#undef DISN_ACC
    }
    } else {
    // the resampler of sample4, its four map reads replaced by the up-sampled tap pixels (tap_pixel's expression).
    // The four map pixels are {ify, icy} x {ifx, icx}, so their 16 tap pixels are a 4 x 4 grid {ylo, yhi of both map
    // rows} x {xlo, xhi of both map columns}: all 16 loads are issued before any is used (one memory round trip per
    // output instead of four; addresses of out-of-map pixels are clamped, their values replaced by the resampler's
    // zeros afterwards -- no branch around a load).
    const bool ok = x > -1.0f && y > -1.0f && x < (float)DISN_IMG && y < (float)DISN_IMG;
    if (ok) {
      const float fx = floorf(x), fy = floorf(y);
      const float cx = fx + 1.0f, cy = fy + 1.0f;
      const float dx = cx - x, dy = cy - y;
      const int ifx = (int)fx, ify = (int)fy, icx = (int)cx, icy = (int)cy;
      const float w_ff = dx * dy;
      const float w_cc = (1.0f - dx) * (1.0f - dy);
      const float w_fc = dx * (1.0f - dy);
      const float w_cf = (1.0f - dx) * dy;
      const bool xf = ifx >= 0 && ifx < DISN_IMG, xc = icx >= 0 && icx < DISN_IMG;
      const bool yf = ify >= 0 && ify < DISN_IMG, yc = icy >= 0 && icy < DISN_IMG;
      int roff[4], coff[4];
      float yl[2], xl[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int oy = min(max(h ? icy : ify, 0), DISN_IMG - 1), ox = min(max(h ? icx : ifx, 0), DISN_IMG - 1);
        const float ty = (float)oy * s, tx = (float)ox * s;
        const int ylo = (int)floorf(ty), xlo = (int)floorf(tx);
        const int yhi = min(ylo + 1, hw - 1), xhi = min(xlo + 1, hw - 1);
        yl[h] = ty - (float)ylo;
        xl[h] = tx - (float)xlo;
        roff[2 * h] = ylo * hw * ch;
        roff[2 * h + 1] = yhi * hw * ch;
        coff[2 * h] = xlo * ch;
        coff[2 * h + 1] = xhi * ch;
      }
      const float* base = tap + cl;
      float4 T[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) T[r][q] = *reinterpret_cast<const float4*>(base + roff[r] + coff[q]);
      // map pixel (row half hy, column half hx): tl, tr, bl, br = T[2 hy][2 hx], T[2 hy][2 hx + 1], T[2 hy + 1][..]
#define DISN_LERP(hy, hx, f) \
  ((T[2 * hy][2 * hx].f + (T[2 * hy][2 * hx + 1].f - T[2 * hy][2 * hx].f) * xl[hx]) + \
   ((T[2 * hy + 1][2 * hx].f + (T[2 * hy + 1][2 * hx + 1].f - T[2 * hy + 1][2 * hx].f) * xl[hx]) - \
    (T[2 * hy][2 * hx].f + (T[2 * hy][2 * hx + 1].f - T[2 * hy][2 * hx].f) * xl[hx])) * yl[hy])
#define DISN_ACC(f)                                         \
  {                                                         \
    const float p_ff = (xf && yf) ? DISN_LERP(0, 0, f) : 0.f; \
    const float p_cc = (xc && yc) ? DISN_LERP(1, 1, f) : 0.f; \
    const float p_fc = (xf && yc) ? DISN_LERP(1, 0, f) : 0.f; \
    const float p_cf = (xc && yf) ? DISN_LERP(0, 1, f) : 0.f; \
    float v = w_ff * p_ff;                                  \
    v = v + w_cc * p_cc;                                    \
    v = v + w_fc * p_fc;                                    \
    v = v + w_cf * p_cf;                                    \
    o.f = v;                                                \
  }
      DISN_ACC(x) DISN_ACC(y) DISN_ACC(z) DISN_ACC(w)
#undef DISN_ACC
#undef DISN_LERP
    }
    }
    if (split_amax) {
      const float sc = split_pow2_scale(feat_split_amax(split_amax[b]));
      const float xs[4] = {o.x * sc, o.y * sc, o.z * sc, o.w * sc};
      _Float16 hh[4], ll[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hh[e] = (_Float16)xs[e];
        ll[e] = (_Float16)(xs[e] - (float)hh[e]);
      }
      // Lanes 2i, 2i + 1 hold the two halves (channels c .. c + 3, c + 4 .. c + 7) of ONE 8-channel group (c4_begin and
      // c4_count are even, threads leave the loop in pairs): the even lane sends its l half and takes the partner's h
      // half, the odd lane the other way round -- one 16-byte store per lane (h8 | l8), a wave's store instruction 1 KiB
      // of contiguous bytes, instead of two 8-byte stores at half density
      const uint2 hv = *reinterpret_cast<const uint2*>(hh), lv = *reinterpret_cast<const uint2*>(ll);
      const bool odd = (c & 4) != 0;
      const uint2 send = odd ? hv : lv;
      uint2 recv;
      recv.x = (unsigned)__shfl_xor((int)send.x, 1);
      recv.y = (unsigned)__shfl_xor((int)send.y, 1);
      const uint4 out = odd ? make_uint4(recv.x, recv.y, lv.x, lv.y) : make_uint4(hv.x, hv.y, recv.x, recv.y);
      unsigned char* row = reinterpret_cast<unsigned char*>(feat) + (pt * (size_t)feat_ld + (size_t)(c & ~7)) * 4 + (odd ? 16 : 0);
      *reinterpret_cast<uint4*>(row) = out;
    } else {
      *reinterpret_cast<float4*>(feat + pt * feat_ld + c) = o;
    }
    vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
  }
  if (amax) {
    __shared__ float red[16];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = 0.f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, red[w]);
      amax[(size_t)bimg * amax_stride + iw] = m;
    }
  }
}

// ---------------------------------------------------------------------------
// Round 6: the same rows, ONE WAVE PER POINT (all five taps, models/model_normalization.py:171-190).  The kernel above
// gives every (point, 4 channels) its own thread: 368 threads re-derive the point's projection, resampler weights and
// tap coordinates, and each issues 16 tap loads -- 100 KB per point requested through L1 for 31 KB of distinct bytes
// (r05i PMC), load-issue bound at 0.45-0.55 of the HBM roofline.  Here the projection and the five taps' geometry are
// wave-uniform (computed once per point, the row / column offsets passed through readfirstlane so that the branches
// below are scalar), a lane owns 4 channels, and a point is six passes of the wave:
//     tap 4 channels 0..255 | 256..511, tap 3 the same, tap 2, then [tap 1 (lanes 0..31) | tap 0 (32..47) | the
//     zero padding columns 1472 .. feat_ld - 1 (48..63)].
// The resampler's four map pixels {ify, icy} x {ifx, icx} are up-sampled from the tap rows {ylo, yhi}(ify), {ylo,
// yhi}(icy) and the same four columns; for the up-sampled taps (scale < 1: taps 1..4) the rows of icy are those of ify
// (case a: 1 - s of the points), or start at ify's second row (case b) -- wave-uniform facts: duplicate rows and
// columns are neither loaded nor interpolated twice.  4.4-5.8 loads per pass instead of 16 for taps 2..4 (expected,
// s = 0.41 / 0.20 / 0.10), ~40 wave loads and 36 KB requested per point instead of 92 and 94 KB.  The arithmetic is the
// kernel above's expression by expression (horizontal lerp of a row pair, vertical lerp, the resampler's weighted sum in
// its order; this file is compiled with -ffp-contract=off): the SAME BITS (tests: both against the oracle bit for bit).
// ---------------------------------------------------------------------------
struct GatherSlots { const float* p[5]; size_t stride; float* out; };   // see project_gather_taps_wave_kernel
typedef float gf2 __attribute__((ext_vector_type(2)));
struct GF4 { gf2 lo, hi; };   // a float4 as two packed pairs: v_pk_add_f32 / v_pk_mul_f32 (never fused: -ffp-contract=off)
__device__ __forceinline__ GF4 gf4_load(const float* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  return GF4{gf2{v.x, v.y}, gf2{v.z, v.w}};
}
// a + (b - a) * t
__device__ __forceinline__ GF4 gf4_lerp(const GF4& a, const GF4& b, float t) {
  const gf2 tt{t, t};
  return GF4{a.lo + (b.lo - a.lo) * tt, a.hi + (b.hi - a.hi) * tt};
}
__device__ __forceinline__ GF4 gf4_sel(bool c, const GF4& a, const GF4& b) { return c ? a : b; }

struct TapGeom {
  int roff[4], coff[4];   // floats: rows {ylo, yhi}(ify), {ylo, yhi}(icy); columns the same for ifx, icx
  float xl[2], yl[2];
};
__device__ __forceinline__ TapGeom tap_geom(int hw, int ch, float s, int ify, int icy, int ifx, int icx) {
  TapGeom g;
#pragma unroll
  for (int h = 0; h < 2; ++h) {   // (project_gather_taps_kernel<true>'s lines: clamped map pixel -> tap_pixel's coordinates)
    const int oy = min(max(h ? icy : ify, 0), DISN_IMG - 1), ox = min(max(h ? icx : ifx, 0), DISN_IMG - 1);
    const float ty = (float)oy * s, tx = (float)ox * s;
    const int ylo = (int)floorf(ty), xlo = (int)floorf(tx);
    const int yhi = min(ylo + 1, hw - 1), xhi = min(xlo + 1, hw - 1);
    g.yl[h] = ty - (float)ylo;
    g.xl[h] = tx - (float)xlo;
    g.roff[2 * h] = ylo * hw * ch;
    g.roff[2 * h + 1] = yhi * hw * ch;
    g.coff[2 * h] = xlo * ch;
    g.coff[2 * h + 1] = xhi * ch;
  }
  return g;
}

struct PointGeom {
  float w_ff, w_cc, w_fc, w_cf;
  bool ok, ff, cc, fc, cf;
  int ify, icy, ifx, icx;
};

// the four up-sampled map pixels of this lane's 4 channels, then the resampler's sum.  RC / CC: how the tap rows (columns)
// of the map's second row icy (column icx) relate to those of the first -- 0: the same two (case a), 1: they start at the
// first pair's second row (case b: one new row), 2: two new rows (always right: duplicates are then loaded twice).
template <int RC, int CC>
__device__ __forceinline__ float4 tap_resample(const float* __restrict__ base, const TapGeom& g, const PointGeom& pg) {
  constexpr bool rneed[4] = {true, true, RC == 2, RC != 0};
  constexpr bool cneed[4] = {true, true, CC == 2, CC != 0};
  GF4 T[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (rneed[r] && cneed[q]) T[r][q] = gf4_load(base + g.roff[r] + g.coff[q]);   // (every load before any use)
  GF4 H[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (rneed[r]) {
      H[r][0] = gf4_lerp(T[r][0], T[r][1], g.xl[0]);
      H[r][1] = gf4_lerp(CC == 0 ? T[r][0] : (CC == 1 ? T[r][1] : T[r][2]), CC == 0 ? T[r][1] : T[r][3], g.xl[1]);
    }
  GF4 P[2][2];
#pragma unroll
  for (int hx = 0; hx < 2; ++hx) {
    P[0][hx] = gf4_lerp(H[0][hx], H[1][hx], g.yl[0]);
    P[1][hx] = gf4_lerp(RC == 0 ? H[0][hx] : (RC == 1 ? H[1][hx] : H[2][hx]), RC == 0 ? H[1][hx] : H[3][hx], g.yl[1]);
  }
  const gf2 z{0.f, 0.f};
  // v = w_ff p_ff; v = v + w_cc p_cc; v = v + w_fc p_fc; v = v + w_cf p_cf   (sample4's order)
  const gf2 wff{pg.w_ff, pg.w_ff}, wcc{pg.w_cc, pg.w_cc}, wfc{pg.w_fc, pg.w_fc}, wcf{pg.w_cf, pg.w_cf};
  gf2 lo = wff * (pg.ff ? P[0][0].lo : z), hi = wff * (pg.ff ? P[0][0].hi : z);
  lo = lo + wcc * (pg.cc ? P[1][1].lo : z);
  hi = hi + wcc * (pg.cc ? P[1][1].hi : z);
  lo = lo + wfc * (pg.fc ? P[1][0].lo : z);
  hi = hi + wfc * (pg.fc ? P[1][0].hi : z);
  lo = lo + wcf * (pg.cf ? P[0][1].lo : z);
  hi = hi + wcf * (pg.cf ? P[0][1].hi : z);
  return make_float4(lo[0], lo[1], hi[0], hi[1]);
}
// wave-uniform offsets (scalar registers): the case by comparison, a scalar branch to one of nine straight-line bodies
__device__ __forceinline__ float4 tap_resample_uniform(const float* __restrict__ base, const TapGeom& g, const PointGeom& pg) {
  const int rc = g.roff[2] == g.roff[0] ? 0 : (g.roff[2] == g.roff[1] ? 1 : 2);
  const int cc = g.coff[2] == g.coff[0] ? 0 : (g.coff[2] == g.coff[1] ? 1 : 2);
  switch (rc * 3 + cc) {
    case 0: return tap_resample<0, 0>(base, g, pg);
    case 1: return tap_resample<0, 1>(base, g, pg);
    case 2: return tap_resample<0, 2>(base, g, pg);
    case 3: return tap_resample<1, 0>(base, g, pg);
    case 4: return tap_resample<1, 1>(base, g, pg);
    case 5: return tap_resample<1, 2>(base, g, pg);
    case 6: return tap_resample<2, 0>(base, g, pg);
    case 7: return tap_resample<2, 1>(base, g, pg);
    default: return tap_resample<2, 2>(base, g, pg);
  }
}

template <bool SPLIT>
__global__ __launch_bounds__(256) void project_gather_taps_wave_kernel(TapSet t, const float* __restrict__ trans_mat,
                                                                       const float* __restrict__ pts, int B, int n,
                                                                       float* __restrict__ feat, int feat_ld,
                                                                       float* __restrict__ amax, size_t amax_stride,
                                                                       const float* __restrict__ split_amax, GatherSlots slots) {
  // amax != nullptr: gridDim.x = B * G, workgroup (b, iw) walks image b and stores the maximum |feat| it wrote at
  // amax[b * amax_stride + iw] (every (b, iw) writes), as the kernel above.
  // slots.p[0] != nullptr (SPLIT): the image's tap maximum is taken HERE from its 5 x 64 activation-maximum slots (five
  // loads and a wave maximum per point: what tap_amax_kernel did in a launch of its own on the call's critical path) and
  // published by the image's first point in slots.out[b] for the fused kernel behind
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  size_t pt, end, step;
  int iw = 0, bimg = 0;
  if (amax) {
    const int G = gridDim.x / B;
    bimg = blockIdx.x / G;
    iw = blockIdx.x - bimg * G;
    pt = (size_t)bimg * n + (size_t)iw * nw + wave;
    end = (size_t)(bimg + 1) * n;
    step = (size_t)G * nw;
  } else {
    pt = (size_t)blockIdx.x * nw + wave;
    end = (size_t)B * n;
    step = (size_t)gridDim.x * nw;
  }
  float vmax = 0.f;
  float img_amax = 0.f;
  auto emit = [&](size_t p, int b, int c, const float4& o) __attribute__((always_inline)) {
    if (SPLIT) {   // (the kernel above's split store: lanes 2i, 2i + 1 are the halves of one 8-channel group)
      const float sc = split_pow2_scale(feat_split_amax(img_amax));
      const float xs[4] = {o.x * sc, o.y * sc, o.z * sc, o.w * sc};
      _Float16 hh[4], ll[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hh[e] = (_Float16)xs[e];
        ll[e] = (_Float16)(xs[e] - (float)hh[e]);
      }
      const uint2 hv = *reinterpret_cast<const uint2*>(hh), lv = *reinterpret_cast<const uint2*>(ll);
      const bool odd = (c & 4) != 0;
      const uint2 send = odd ? hv : lv;
      uint2 recv;
      recv.x = (unsigned)__shfl_xor((int)send.x, 1);
      recv.y = (unsigned)__shfl_xor((int)send.y, 1);
      const uint4 out = odd ? make_uint4(recv.x, recv.y, lv.x, lv.y) : make_uint4(hv.x, hv.y, recv.x, recv.y);
      unsigned char* row = reinterpret_cast<unsigned char*>(feat) + (p * (size_t)feat_ld + (size_t)(c & ~7)) * 4 + (odd ? 16 : 0);
      *reinterpret_cast<uint4*>(row) = out;
    } else {
      *reinterpret_cast<float4*>(feat + p * feat_ld + c) = o;
    }
    vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
  };
  for (; pt < end; pt += step) {
    const int b = (int)(pt / n);
    if (SPLIT) {
      if (slots.p[0]) {
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) m = fmaxf(m, slots.p[k][(size_t)b * slots.stride + lane]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        img_amax = m;
        if (lane == 0 && pt == (size_t)b * n) slots.out[b] = m;
      } else {
        img_amax = split_amax[b];
      }
    }
    float x, y;
    project_point(trans_mat + (size_t)b * 12, pts[pt * 3], pts[pt * 3 + 1], pts[pt * 3 + 2], x, y);
    PointGeom pg;
    pg.ok = x > -1.0f && y > -1.0f && x < (float)DISN_IMG && y < (float)DISN_IMG;
    {
      const float fx = floorf(x), fy = floorf(y);
      const float cx = fx + 1.0f, cy = fy + 1.0f;
      const float dx = cx - x, dy = cy - y;
      pg.ifx = (int)fx; pg.ify = (int)fy; pg.icx = (int)cx; pg.icy = (int)cy;
      pg.w_ff = dx * dy;
      pg.w_cc = (1.0f - dx) * (1.0f - dy);
      pg.w_fc = dx * (1.0f - dy);
      pg.w_cf = (1.0f - dx) * dy;
      const bool xf = pg.ifx >= 0 && pg.ifx < DISN_IMG, xc = pg.icx >= 0 && pg.icx < DISN_IMG;
      const bool yf = pg.ify >= 0 && pg.ify < DISN_IMG, yc = pg.icy >= 0 && pg.icy < DISN_IMG;
      pg.ff = xf && yf; pg.cc = xc && yc; pg.fc = xf && yc; pg.cf = xc && yf;
    }
    // a NaN projection (degenerate camera) fails `ok`: zeros, as sample4.  The int conversions above are then unused.
    const bool okw = __builtin_amdgcn_readfirstlane((int)pg.ok) != 0;
    // ---- taps 4, 3 (two passes each), tap 2: wave-uniform geometry
#pragma unroll 1
    for (int k = 4; k >= 2; --k) {
      const int hw = 224 >> k, ch = k == 2 ? 256 : 512, coff0 = k == 2 ? 192 : (k == 3 ? 448 : 960);
      const float* tap = t.p[k] + (size_t)b * t.stride[k];
      TapGeom g;
      if (okw) {
        g = tap_geom(hw, ch, t.s[k], pg.ify, pg.icy, pg.ifx, pg.icx);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          g.roff[i] = __builtin_amdgcn_readfirstlane(g.roff[i]);
          g.coff[i] = __builtin_amdgcn_readfirstlane(g.coff[i]);
        }
      }
#pragma unroll 1
      for (int half = 0; half < (k == 2 ? 1 : 2); ++half) {
        const int cl = 4 * lane + 256 * half;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (okw) o = tap_resample_uniform(tap + cl, g, pg);
        emit(pt, b, coff0 + cl, o);
      }
    }
    // ---- tap 1 (lanes 0..31), tap 0 (32..47), padding columns (48..63): per-lane geometry, no skipping
    {
      const bool is1 = lane < 32, is0 = lane >= 32 && lane < 48;
      const int cl = is1 ? 4 * lane : 4 * (lane - 32);
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (okw && (is1 || is0)) {
        const TapGeom g1 = tap_geom(112, 128, t.s[1], pg.ify, pg.icy, pg.ifx, pg.icx);
        const TapGeom g0 = tap_geom(224, 64, t.s[0], pg.ify, pg.icy, pg.ifx, pg.icx);
        TapGeom g;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          g.roff[i] = is1 ? g1.roff[i] : g0.roff[i];
          g.coff[i] = is1 ? g1.coff[i] : g0.coff[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          g.xl[i] = is1 ? g1.xl[i] : g0.xl[i];
          g.yl[i] = is1 ? g1.yl[i] : g0.yl[i];
        }
        const float* tap = (is1 ? t.p[1] + (size_t)b * t.stride[1] : t.p[0] + (size_t)b * t.stride[0]) + cl;
        o = tap_resample<2, 2>(tap, g, pg);
      }
      const int c = is1 ? 64 + cl : (is0 ? cl : DISN_FEAT + 4 * (lane - 48));
      if (is1 || is0) emit(pt, b, c, o);
      else if (c < feat_ld) {
        *reinterpret_cast<float4*>(feat + pt * feat_ld + c) = make_float4(0.f, 0.f, 0.f, 0.f);   // (zeros in either form)
      }
    }
  }
  if (amax) {
    __shared__ float red[4];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off));
    if (lane == 0) red[wave] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = 0.f;
      for (int w = 0; w < nw; ++w) m = fmaxf(m, red[w]);
      amax[(size_t)bimg * amax_stride + iw] = m;
    }
  }
}

bool project_gather_taps_takes_slots(int B, int n, int feat_ld) {
  if (feat_ld <= 0) feat_ld = DISN_FEAT;
  return feat_ld >= DISN_FEAT && feat_ld <= DISN_FEAT + 64 && feat_ld % 4 == 0 && tune::gather_l16 == 0 && (size_t)B * n >= 10240;
}

// workgroups per image of the launch with maxima (1024 threads each) -- the count of entries the consumer reads.
// All five taps: up to 448 entries (the free tail of an image's slot set, api.hip); a tap RANGE (the two gathers of a
// batched call: taps 0..3 behind conv4_3, tap 4 behind conv5_3): up to 224, the two launches' entries side by side.
static int gather_c4_count(int tap_begin, int tap_end, int feat_ld) {
  static const int c4_off[6] = {0, 16, 48, 112, 240, DISN_FEAT4};
  return (tap_end == 5 && feat_ld > DISN_FEAT ? feat_ld / 4 : c4_off[tap_end]) - c4_off[tap_begin];
}
int project_gather_taps_amax_blocks(int n, int feat_ld, int tap_begin, int tap_end) {
  const size_t per_img = (size_t)n * gather_c4_count(tap_begin, tap_end, feat_ld > 0 ? feat_ld : DISN_FEAT);
  const size_t g = (per_img + 1023) / 1024;
  const size_t cap = tap_begin == 0 && tap_end == 5 ? 448 : 224;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

hipError_t project_gather_taps_launch(const float* const taps[5], const float* trans_mat,
                                      const float* pts, int B, int n, int tap_begin, int tap_end,
                                      float* feat, hipStream_t st, int feat_ld, float* amax,
                                      size_t amax_stride, int amax_cap, const float* split_amax,
                                      const float* const* tap_slots, size_t slot_stride) {
  static const int c4_off[6] = {0, 16, 48, 112, 240, DISN_FEAT4};
  static const int ch[5] = {64, 128, 256, 512, 512};
  TapSet t;
  for (int k = 0; k < 5; ++k) {
    t.p[k] = taps[k];
    t.s[k] = (float)(224 >> k) / (float)DISN_IMG;
    t.stride[k] = (size_t)(224 >> k) * (224 >> k) * ch[k];
  }
  if (feat_ld <= 0) feat_ld = DISN_FEAT;
  // all five taps into rows of 1472 .. 1536 floats, from 10 240 points on: one wave per point (round 6; the SAME BITS, so
  // the choice is free).  Its six passes per point are six dependent memory round trips: below ~5 x 2048 points, where
  // the chip is not filled with waves anyway, the thread-per-float4 kernel's 368 independent threads per point win
  // (profiles/r06j_gather_ab.txt: 1 x 2048 points 14 against 35 us, 4 x 2048 44 / 49, 8 x 2048 85 / 70, 16 x 2048 174 / 123).
  // tune::gather_l16 != 0 (tuning builds) forces the thread-per-float4 kernel: 1 its all-loads-first schedule, 2 the default one
  if (tap_begin == 0 && tap_end == 5 && project_gather_taps_takes_slots(B, n, feat_ld)) {
    GatherSlots gs{};
    if (tap_slots && split_amax) {
      for (int k = 0; k < 5; ++k) gs.p[k] = tap_slots[k];
      gs.stride = slot_stride;
      gs.out = const_cast<float*>(split_amax);
    }
    if (amax) {
      int G = project_gather_taps_amax_blocks(n, feat_ld, tap_begin, tap_end);
      if (amax_cap > 0 && G > amax_cap) G = amax_cap;
      if (split_amax) hipLaunchKernelGGL(project_gather_taps_wave_kernel<true>, dim3((unsigned)(B * G)), dim3(256), 0, st, t, trans_mat, pts, B, n, feat, feat_ld, amax, amax_stride, split_amax, gs);
      else hipLaunchKernelGGL(project_gather_taps_wave_kernel<false>, dim3((unsigned)(B * G)), dim3(256), 0, st, t, trans_mat, pts, B, n, feat, feat_ld, amax, amax_stride, split_amax, gs);
    } else {
      const size_t npt = (size_t)B * n;
      const unsigned grid = (unsigned)((npt + 3) / 4 < 32768 ? (npt + 3) / 4 : 32768);
      if (split_amax) hipLaunchKernelGGL(project_gather_taps_wave_kernel<true>, dim3(grid), dim3(256), 0, st, t, trans_mat, pts, B, n, feat, feat_ld, amax, amax_stride, split_amax, gs);
      else hipLaunchKernelGGL(project_gather_taps_wave_kernel<false>, dim3(grid), dim3(256), 0, st, t, trans_mat, pts, B, n, feat, feat_ld, amax, amax_stride, split_amax, gs);
    }
    return hipGetLastError();
  }
  const int c4_begin = c4_off[tap_begin];
  const int c4_count = (tap_end == 5 && feat_ld > DISN_FEAT ? feat_ld / 4 : c4_off[tap_end]) - c4_begin;
  const size_t total = (size_t)B * n * c4_count;
  if (amax) {
    int G = project_gather_taps_amax_blocks(n, feat_ld, tap_begin, tap_end);
    if (amax_cap > 0 && G > amax_cap) G = amax_cap;   // entries the caller has room for
    if (tune::gather_l16 == 1) hipLaunchKernelGGL(project_gather_taps_kernel<true>, dim3((unsigned)(B * G)), dim3(1024), 0, st, t, trans_mat, pts, B, n,
                       c4_begin, c4_count, feat, feat_ld, amax, amax_stride, split_amax);
    else hipLaunchKernelGGL(project_gather_taps_kernel<false>, dim3((unsigned)(B * G)), dim3(1024), 0, st, t, trans_mat, pts, B, n,
                       c4_begin, c4_count, feat, feat_ld, amax, amax_stride, split_amax);
    return hipGetLastError();
  }
  if (tune::gather_l16 == 1) hipLaunchKernelGGL(project_gather_taps_kernel<true>, dim3(grid_for(total, 16384)), dim3(256), 0, st, t,
                     trans_mat, pts, B, n, c4_begin, c4_count, feat, feat_ld, amax, amax_stride, split_amax);
  else hipLaunchKernelGGL(project_gather_taps_kernel<false>, dim3(grid_for(total, 16384)), dim3(256), 0, st, t,
                     trans_mat, pts, B, n, c4_begin, c4_count, feat, feat_ld, amax, amax_stride, split_amax);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// folded local fold2/conv1: h[pt][:] = relu(pre[pt][:] + sum_c w_c * pmap[pixel_c][:] + bias), where
// pmap = featmap . W_feat ([137*137][512], disn_fold_local) and pre = point512 . W_point.  The
// resampler weights / validity are sample4's; one thread per (point, float4 of the 512 outputs), so a
// point's four 2-KiB pmap rows are read by 128 consecutive lanes.  In place (h may be pre).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_fold_kernel(const float* __restrict__ pmap_b,
                                                          const float* __restrict__ trans_mat_b,
                                                          const float* __restrict__ pts, int n,
                                                          const float* pre,
                                                          const float* __restrict__ bias, float* h) {
  const size_t total = (size_t)n * 128;
  // XCD-aware order: hardware workgroup w runs on XCD w % 8; give each XCD a CONTIGUOUS eighth of the
  // blocks, so the pmap rows its points touch (neighbouring points project to neighbouring pixels)
  // fit its 4 MB L2 instead of all eight L2s streaming the whole footprint from the Infinity Cache
  unsigned lb = blockIdx.x;
  {
    const unsigned W = gridDim.x, q = W >> 3, r = W & 7, xcd = lb & 7, idx = lb >> 3;
    lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  for (size_t i = (size_t)lb * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t pt = i >> 7;
    const int c = (int)(i & 127) * 4;
    float x, y;
    project_point(trans_mat_b, pts[pt * 3], pts[pt * 3 + 1], pts[pt * 3 + 2], x, y);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool ok = x > -1.0f && y > -1.0f && x < (float)DISN_IMG && y < (float)DISN_IMG;
    if (ok) {
      const float fx = floorf(x), fy = floorf(y);
      const float cx = fx + 1.0f, cy = fy + 1.0f;
      const float dx = cx - x, dy = cy - y;
      const int ifx = (int)fx, ify = (int)fy, icx = (int)cx, icy = (int)cy;
      const float w_ff = dx * dy;
      const float w_cc = (1.0f - dx) * (1.0f - dy);
      const float w_fc = dx * (1.0f - dy);
      const float w_cf = (1.0f - dx) * dy;
      const bool xf = ifx >= 0 && ifx < DISN_IMG, xc = icx >= 0 && icx < DISN_IMG;
      const bool yf = ify >= 0 && ify < DISN_IMG, yc = icy >= 0 && icy < DISN_IMG;
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* m = pmap_b + c;
      const float4 v_ff = (xf && yf) ? *reinterpret_cast<const float4*>(m + ((size_t)ify * DISN_IMG + ifx) * 512) : z4;
      const float4 v_cc = (xc && yc) ? *reinterpret_cast<const float4*>(m + ((size_t)icy * DISN_IMG + icx) * 512) : z4;
      const float4 v_fc = (xf && yc) ? *reinterpret_cast<const float4*>(m + ((size_t)icy * DISN_IMG + ifx) * 512) : z4;
      const float4 v_cf = (xc && yf) ? *reinterpret_cast<const float4*>(m + ((size_t)ify * DISN_IMG + icx) * 512) : z4;
#define DISN_ACC(f)           \
  {                           \
    float t = w_ff * v_ff.f;  \
    t = t + w_cc * v_cc.f;    \
    t = t + w_fc * v_fc.f;    \
    t = t + w_cf * v_cf.f;    \
    v.f = t;                  \
  }
      DISN_ACC(x) DISN_ACC(y) DISN_ACC(z) DISN_ACC(w)
#undef DISN_ACC
    }
    const float4 p4 = *reinterpret_cast<const float4*>(pre + pt * 512 + c);
    const float4 b4 = *reinterpret_cast<const float4*>(bias + c);
    float4 o;
    o.x = fmaxf((p4.x + v.x) + b4.x, 0.f);
    o.y = fmaxf((p4.y + v.y) + b4.y, 0.f);
    o.z = fmaxf((p4.z + v.z) + b4.z, 0.f);
    o.w = fmaxf((p4.w + v.w) + b4.w, 0.f);
    *reinterpret_cast<float4*>(h + pt * 512 + c) = o;
  }
}

hipError_t gather_fold_launch(const float* pmap_b, const float* trans_mat_b, const float* pts, int n,
                              const float* pre, const float* bias, float* h, hipStream_t st) {
  const size_t total = (size_t)n * 128;
  // (block count 1024 .. 65536: no effect on the chunk time, tools/fold_time.py)
  hipLaunchKernelGGL(gather_fold_kernel, dim3(grid_for(total, 16384)), dim3(256), 0, st, pmap_b,
                     trans_mat_b, pts, n, pre, bias, h);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// dense grid points: numpy.linspace in float64 (i*step + start, last = stop), cast float32
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grid_points_kernel(GridSpec g, int64_t k0, int64_t k1,
                                                          float* __restrict__ pts) {
  const int64_t n = k1 - k0;
  const int64_t res = g.res;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = k0 + i;
    const int ix = (int)(k % res);
    const int iy = (int)((k / res) % res);
    const int iz = (int)(k / (res * res));
    const int idx[3] = {ix, iy, iz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      double v = (double)idx[a] * g.step[a];
      v = v + g.start[a];
      if (idx[a] == g.res - 1 && g.res > 1) v = g.stop[a];
      pts[i * 3 + a] = (float)v;
    }
  }
}

hipError_t grid_points_launch(const GridSpec& g, int64_t k0, int64_t k1, float* pts,
                              hipStream_t st) {
  hipLaunchKernelGGL(grid_points_kernel, dim3(grid_for((size_t)(k1 - k0))), dim3(256), 0, st, g, k0,
                     k1, pts);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void scale_div_kernel(const float* __restrict__ in, float divisor,
                                                        int64_t n, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = in[i] / divisor;
}

hipError_t scale_div_launch(const float* in, float divisor, int64_t n, float* out, hipStream_t st) {
  hipLaunchKernelGGL(scale_div_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, st, in, divisor, n,
                     out);
  return hipGetLastError();
}

// rows of B images re-strided: dst[b][r][0..w) = src[b][r][0..w) for r < min(rs, rd), zeros for the rows rs .. rd - 1
// (pad points (0, 0, 0) -- what test/create_sdf.py:241,256 appends -- in; the first N results out)
__global__ __launch_bounds__(256) void restride_rows_kernel(const float* __restrict__ src, int rs, float* __restrict__ dst,
                                                            int rd, int w, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t per = (int64_t)rd * w;
    const int64_t b = i / per, e = i - b * per;
    const int r = (int)(e / w);
    dst[i] = r < rs ? src[(b * rs + r) * w + (e - (int64_t)r * w)] : 0.0f;
  }
}

hipError_t restride_rows_launch(const float* src, int B, int rs, float* dst, int rd, int w, hipStream_t st) {
  const int64_t n = (int64_t)B * rd * w;
  hipLaunchKernelGGL(restride_rows_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, st, src, rs, dst, rd, w, n);
  return hipGetLastError();
}

// equalised <-> true units of taps / gathered features (disn_equalise_weights): out[r][c] = in[r][c] * scale[c] or
// in[r][c] / scale[c]; the factors are powers of two, both forms exact.  C % 4 == 0 (64 .. 1472): float4 per thread
__global__ __launch_bounds__(256) void scale_channels_kernel(const float* __restrict__ in, int64_t n4, int C4,
                                                             const float* __restrict__ scale, int invert,
                                                             float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const float4 v = reinterpret_cast<const float4*>(in)[i];
    const float4 s = reinterpret_cast<const float4*>(scale)[c4];
    float4 r;
    if (invert) { r.x = v.x / s.x; r.y = v.y / s.y; r.z = v.z / s.z; r.w = v.w / s.w; }
    else { r.x = v.x * s.x; r.y = v.y * s.y; r.z = v.z * s.z; r.w = v.w * s.w; }
    reinterpret_cast<float4*>(out)[i] = r;
  }
}

hipError_t scale_channels_launch(const float* in, int64_t rows, int C, const float* scale, int invert, float* out,
                                 hipStream_t st) {
  const int64_t n4 = rows * (C / 4);
  hipLaunchKernelGGL(scale_channels_kernel, dim3(grid_for((size_t)n4)), dim3(256), 0, st, in, n4, C / 4, scale, invert,
                     out);
  return hipGetLastError();
}

}  // namespace disn
