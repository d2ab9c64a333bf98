// Backward of the image-side element-wise rows (training step):
//   gather_bwd    -- ResamplerGrad w.r.t. data       (models/model_normalization.py:172-190)
//   resize_bwd    -- ResizeBilinearGrad (legacy)     (models/model_normalization.py:171-183)
//   maxpool_bwd   -- MaxPoolGrad 2x2/2               (models/CNN/vgg.py:188-196)
//   im2col_c3     -- patches of the 3-channel input for the conv1_1 weight gradient
// All HBM/L2-bound.  resize_bwd and maxpool_bwd are written as GATHERS over the output gradient
// (fixed summation order, no atomics); gather_bwd scatters 4 x 1472 floats per point with hardware
// fp32 atomics into a zeroed map (a point set hits pixels in data-dependent order, as TF's GPU
// ResamplerGrad does).
#include "kernels.hpp"

namespace disn {

#define DISN_FEAT 1472
#define DISN_FEAT4 368
#define DISN_IMG 137

// dfeat [B*N][1472], xy [B*N][2] -> dmap [B,137,137,1472] += w * dfeat   (dmap zeroed by the caller)
// One wave per point; lane l takes channels l, l+64, ...: every atomic instruction covers 256
// contiguous bytes (two full cache lines) -- 4x fewer L2 line operations than a float4-per-lane
// mapping, whose four component instructions each touch a quarter of eight lines.
__global__ __launch_bounds__(256) void gather_bwd_kernel(const float* __restrict__ dfeat,
                                                         const float* __restrict__ xy, int B, int N,
                                                         float* __restrict__ dmap) {
  const int lane = threadIdx.x & 63;
  const long npts = (long)B * N;
  for (long pt = (long)blockIdx.x * 4 + (threadIdx.x >> 6); pt < npts; pt += (long)gridDim.x * 4) {
    const int b = (int)(pt / N);
    const float x = xy[pt * 2], y = xy[pt * 2 + 1];
    const bool ok = x > -1.0f && y > -1.0f && x < (float)DISN_IMG && y < (float)DISN_IMG;
    if (!ok) continue;
    const float fx = floorf(x), fy = floorf(y);
    const float cx = fx + 1.0f, cy = fy + 1.0f;
    const float dx = cx - x, dy = cy - y;
    const int ifx = (int)fx, ify = (int)fy, icx = (int)cx, icy = (int)cy;
    const bool xf = ifx >= 0 && ifx < DISN_IMG, xc = icx >= 0 && icx < DISN_IMG;
    const bool yf = ify >= 0 && ify < DISN_IMG, yc = icy >= 0 && icy < DISN_IMG;
    const float w_ff = dx * dy, w_cc = (1.0f - dx) * (1.0f - dy);
    const float w_fc = dx * (1.0f - dy), w_cf = (1.0f - dx) * dy;
    float* mb = dmap + (size_t)b * DISN_IMG * DISN_IMG * DISN_FEAT;
    float* p_ff = mb + ((size_t)ify * DISN_IMG + ifx) * DISN_FEAT;
    float* p_cc = mb + ((size_t)icy * DISN_IMG + icx) * DISN_FEAT;
    float* p_fc = mb + ((size_t)icy * DISN_IMG + ifx) * DISN_FEAT;
    float* p_cf = mb + ((size_t)ify * DISN_IMG + icx) * DISN_FEAT;
    const float* g = dfeat + pt * DISN_FEAT;
    for (int c = lane; c < DISN_FEAT; c += 64) {
      const float v = g[c];
      if (xf && yf) atomicAdd(p_ff + c, w_ff * v);
      if (xc && yc) atomicAdd(p_cc + c, w_cc * v);
      if (xf && yc) atomicAdd(p_fc + c, w_fc * v);
      if (xc && yf) atomicAdd(p_cf + c, w_cf * v);
    }
  }
}

hipError_t gather_bwd_launch(const float* dfeat, const float* xy, int B, int N, float* dmap,
                             hipStream_t st) {
  const long npts = (long)B * N;
  long blocks = (npts + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(gather_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dfeat, xy, B, N, dmap);
  return hipGetLastError();
}

// din[b][y][x][c] (+)= sum over the output pixels (oy,ox) whose 2x2 footprint contains (y,x) of
// weight * dout[b][oy][ox][coff + c].  Output rows are monotone in oy, so the candidates of input
// row y are the contiguous range found from the scale; each candidate is then tested with exactly
// the forward's float arithmetic.
__device__ __forceinline__ void axis_range(int y, float s, int nout, int& lo, int& hi) {
  // outputs with floor(o*s) in {y-1, y}:  o in [ (y-1)/s, (y+1)/s ), widened by one each side
  lo = (int)floorf((float)(y - 1) / s) - 1;
  hi = (int)ceilf((float)(y + 1) / s) + 1;
  if (lo < 0) lo = 0;
  if (hi > nout) hi = nout;
}

__device__ __forceinline__ float axis_weight(int o, int y, float s, int nin) {
  const float f = (float)o * s;
  const int lo = (int)floorf(f);
  const int hi = min(lo + 1, nin - 1);
  const float l = f - (float)lo;
  float w = 0.f;
  if (lo == y) w += 1.0f - l;
  if (hi == y) w += l;
  return w;
}

__global__ __launch_bounds__(256) void resize_bwd_kernel(const float* __restrict__ dout, int B, int Hin,
                                                         int Win, int C, int Hout, int Wout,
                                                         int out_cstride, int out_coff, float sy,
                                                         float sx, float* __restrict__ din,
                                                         int accumulate) {
  const int c4n = C / 4;
  const size_t total = (size_t)B * Hin * Win * c4n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    size_t pidx = i / c4n;
    const int x = (int)(pidx % Win);
    pidx /= Win;
    const int y = (int)(pidx % Hin);
    const int b = (int)(pidx / Hin);
    int oy0, oy1, ox0, ox1;
    axis_range(y, sy, Hout, oy0, oy1);
    axis_range(x, sx, Wout, ox0, ox1);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oy = oy0; oy < oy1; ++oy) {
      const float wy = axis_weight(oy, y, sy, Hin);
      if (wy == 0.f) continue;
      const float* row = dout + ((size_t)(b * Hout + oy) * Wout) * out_cstride + out_coff + c;
      for (int ox = ox0; ox < ox1; ++ox) {
        const float wx = axis_weight(ox, x, sx, Win);
        if (wx == 0.f) continue;
        const float4 g = *reinterpret_cast<const float4*>(row + (size_t)ox * out_cstride);
        const float w = wy * wx;
        acc.x += w * g.x; acc.y += w * g.y; acc.z += w * g.z; acc.w += w * g.w;
      }
    }
    float* o = din + ((size_t)(b * Hin + y) * Win + x) * C + c;
    if (accumulate) {
      const float4 p = *reinterpret_cast<const float4*>(o);
      acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
    }
    *reinterpret_cast<float4*>(o) = acc;
  }
}

// Separable form for strong up-sampling (an input pixel of the 14x14 tap is referenced by ~20x20
// outputs): rows first into tmp [B,Hin,Wout,C], then columns -- 2x20 loads per thread instead of 400.
__global__ __launch_bounds__(256) void resize_bwd_rows_kernel(const float* __restrict__ dout, int B,
                                                              int Hin, int C, int Hout, int Wout,
                                                              int out_cstride, int out_coff, float sy,
                                                              float* __restrict__ tmp) {
  const int c4n = C / 4;
  const size_t total = (size_t)B * Hin * Wout * c4n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    size_t pidx = i / c4n;
    const int ox = (int)(pidx % Wout);
    pidx /= Wout;
    const int y = (int)(pidx % Hin);
    const int b = (int)(pidx / Hin);
    int oy0, oy1;
    axis_range(y, sy, Hout, oy0, oy1);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oy = oy0; oy < oy1; ++oy) {
      const float wy = axis_weight(oy, y, sy, Hin);
      if (wy == 0.f) continue;
      const float4 g = *reinterpret_cast<const float4*>(
          dout + ((size_t)(b * Hout + oy) * Wout + ox) * out_cstride + out_coff + c);
      acc.x += wy * g.x; acc.y += wy * g.y; acc.z += wy * g.z; acc.w += wy * g.w;
    }
    *reinterpret_cast<float4*>(tmp + i * 4) = acc;
  }
}

__global__ __launch_bounds__(256) void resize_bwd_cols_kernel(const float* __restrict__ tmp, int B,
                                                              int Hin, int Win, int C, int Wout,
                                                              float sx, float* __restrict__ din,
                                                              int accumulate) {
  const int c4n = C / 4;
  const size_t total = (size_t)B * Hin * Win * c4n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    size_t pidx = i / c4n;
    const int x = (int)(pidx % Win);
    const size_t by = pidx / Win;  // b * Hin + y
    int ox0, ox1;
    axis_range(x, sx, Wout, ox0, ox1);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* row = tmp + (by * Wout) * C + c;
    for (int ox = ox0; ox < ox1; ++ox) {
      const float wx = axis_weight(ox, x, sx, Win);
      if (wx == 0.f) continue;
      const float4 g = *reinterpret_cast<const float4*>(row + (size_t)ox * C);
      acc.x += wx * g.x; acc.y += wx * g.y; acc.z += wx * g.z; acc.w += wx * g.w;
    }
    float* o = din + i * 4;
    if (accumulate) {
      const float4 p = *reinterpret_cast<const float4*>(o);
      acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
    }
    *reinterpret_cast<float4*>(o) = acc;
  }
}

size_t resize_bwd_ws_bytes(int B, int Hin, int Win, int C, int Hout, int Wout) {
  // separable when an input pixel is referenced by >= ~3x3 outputs
  if (Hout < 2 * Hin || Wout < 2 * Win) return 0;
  return (size_t)B * Hin * Wout * C * sizeof(float);
}

hipError_t resize_bwd_launch(const float* dout, int B, int Hin, int Win, int C, int Hout, int Wout,
                             int out_cstride, int out_coff, float* din, int accumulate, float* tmp,
                             hipStream_t st) {
  const float sy = (float)Hin / (float)Hout, sx = (float)Win / (float)Wout;
  if (tmp && resize_bwd_ws_bytes(B, Hin, Win, C, Hout, Wout) > 0) {
    size_t t1 = (size_t)B * Hin * Wout * (C / 4), b1 = (t1 + 255) / 256;
    if (b1 > 16384) b1 = 16384;
    hipLaunchKernelGGL(resize_bwd_rows_kernel, dim3((unsigned)b1), dim3(256), 0, st, dout, B, Hin, C, Hout,
                       Wout, out_cstride, out_coff, sy, tmp);
    size_t t2 = (size_t)B * Hin * Win * (C / 4), b2 = (t2 + 255) / 256;
    if (b2 > 16384) b2 = 16384;
    hipLaunchKernelGGL(resize_bwd_cols_kernel, dim3((unsigned)b2), dim3(256), 0, st, tmp, B, Hin, Win, C,
                       Wout, sx, din, accumulate);
    return hipGetLastError();
  }
  const size_t total = (size_t)B * Hin * Win * (C / 4);
  size_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(resize_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dout, B, Hin, Win, C,
                     Hout, Wout, out_cstride, out_coff, sy, sx, din, accumulate);
  return hipGetLastError();
}

// dx[b][2oy+i][2ox+j][c] = dy[b][oy][ox][c] at the FIRST maximum of the window (scan order
// (0,0),(0,1),(1,0),(1,1), as the forward's fmaxf chain), else 0.  H, W even.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ dy, int B, int H,
                                                          int W, int C, float* __restrict__ dx) {
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
    const size_t base = ((size_t)(b * H + 2 * oy) * W + 2 * ox) * C + c;
    const size_t off[4] = {0, (size_t)C, (size_t)W * C, (size_t)W * C + C};
    float v[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 t = *reinterpret_cast<const float4*>(x + base + off[k]);
      v[k][0] = t.x; v[k][1] = t.y; v[k][2] = t.z; v[k][3] = t.w;
    }
    const float4 g4 = *reinterpret_cast<const float4*>(dy + i * 4);
    const float g[4] = {g4.x, g4.y, g4.z, g4.w};
    float o[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int am = 0;
      float m = v[0][j];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k][j] > m) { m = v[k][j]; am = k; }
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k][j] = (k == am) ? g[j] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      *reinterpret_cast<float4*>(dx + base + off[k]) = make_float4(o[k][0], o[k][1], o[k][2], o[k][3]);
  }
}

hipError_t maxpool_bwd_launch(const float* x, const float* dy, int B, int H, int W, int C, float* dx,
                              hipStream_t st) {
  const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 4);
  size_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, dy, B, H, W, C, dx);
  return hipGetLastError();
}

// col[m][k] = img[b][y+ky-1][x+kx-1][ci] for k = (ky*3+kx)*3+ci < 27, zero beyond / outside; 64 columns
__global__ __launch_bounds__(256) void im2col_c3_kernel(const float* __restrict__ img, int B, int H,
                                                        int W, float* __restrict__ col) {
  const size_t total = (size_t)B * H * W * 16;  // float4 groups of the 64 columns
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i >> 4;
    const int k0 = (int)(i & 15) * 4;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (k0 < 27) {
      const int x = (int)(m % W);
      const int y = (int)((m / W) % H);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + j;
        if (k < 27) {
          const int tap = k / 3, ci = k - tap * 3;
          const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
          if (yy >= 0 && yy < H && xx >= 0 && xx < W)
            o[j] = img[(m + (long)(tap / 3 - 1) * W + (tap % 3 - 1)) * 3 + ci];
        }
      }
    }
    *reinterpret_cast<float4*>(col + i * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

hipError_t im2col_c3_launch(const float* img, int B, int H, int W, float* col, hipStream_t st) {
  const size_t total = (size_t)B * H * W * 16;
  size_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(im2col_c3_kernel, dim3((unsigned)blocks), dim3(256), 0, st, img, B, H, W, col);
  return hipGetLastError();
}

}  // namespace disn
