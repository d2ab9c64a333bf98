// Weight-gradient GEMM ("TN"): C[P][Q] = sum_m A[m][p] * B[m][q], reduction over the ROWS of two
// row-major activations -- dW = X^T dZ of a 1x1 conv (models/sdfnet.py layers) or, with implicit
// im2col, of a 3x3 SAME conv (models/CNN/vgg.py:187-196): p = tap*Cin + ci, A[m][p] = X at pixel m
// shifted by the tap (zero outside the image).  Training step of SURVEY 8f #3.
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32); 64x64 output tile per workgroup, 4 waves 2x2, one 32x32
// accumulator each; the reduction advances 32 rows per step.  Both operands are staged through LDS
// as [32 rows][64 cols] row-major copies of global memory (coalesced float4 loads); the MFMA
// fragments want 32 consecutive COLUMNS of one row per half-wave, i.e. consecutive LDS words:
// conflict-free ds_read_b32.  k-permutation as in gemm_mfma.hip: MFMA k-index h of step t inside an
// 8-row block is row 4h + t on both operands.
// The output is small (<= 4608 x 512) and the reduction long (2k .. 400k rows), so the work is
// ALWAYS stream-K: (tile, row-step) units split evenly over W workgroups, every segment goes to a
// slab, tn_fixup sums the slabs of a tile in workgroup order (deterministic, no atomics).
#include "kernels.hpp"

namespace disn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct TnDev {
  TnParams p;
  float* ws;  // slabs [2*W][64*64]
  int msteps, ptiles, qtiles, W;
  long units;
};

__device__ __forceinline__ long tn_unit_begin(long U, int W, int w) { return (U * w) / W; }

template <bool CONV>
__global__ __launch_bounds__(256) void gemm_tn_f32_mfma(const TnDev d) {
  constexpr int BP = 64, BQ = 64;
  __shared__ __attribute__((aligned(16))) float lds[2 * 32 * (BP + BQ)];
  const TnParams& p = d.p;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int w = blockIdx.x;
  {
    const int q = d.W >> 3, r = d.W & 7, xcd = w & 7, idx = w >> 3;
    w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int MS = d.msteps;
  const long u0 = tn_unit_begin(d.units, d.W, w), u1 = tn_unit_begin(d.units, d.W, w + 1);

  for (long u = u0; u < u1;) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // keep lane-dependent addressing inside the segment loop
    const int lane = tid & 63;
    const int tile = (int)(u / MS);
    const int s_begin = (int)(u - (long)tile * MS);
    const long rest = u1 - (long)tile * MS;
    const int s_end = rest < MS ? (int)rest : MS;
    const int slot = (u == u0) ? 2 * w : 2 * w + 1;
    u += s_end - s_begin;
    const int pt = tile / d.qtiles, qt = tile - pt * d.qtiles;
    const int p0 = pt * BP, q0 = qt * BQ;
    // conv: this P tile lies inside ONE tap (Cin % 64 == 0)
    int dy = 0, dx = 0, ci0 = p0;
    if (CONV) {
      const int tap = p0 / p.Cin;
      ci0 = p0 - tap * p.Cin;
      dy = tap / 3 - 1;
      dx = tap - (tap / 3) * 3 - 1;
    }
    // loader: thread -> rows (tid>>4) and (tid>>4)+16 of the 32-row step, float4 column tid&15
    const int lrow = tid >> 4, lc4 = (tid & 15) * 4;
    float4 ra[2], rb[2];
    auto load = [&](int s) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const long m = (long)s * 32 + lrow + 16 * i;
        bool oka = m < p.M;
        const float* src = p.a;
        if (CONV) {
          const int hw = p.H * p.W;
          const long mm = oka ? m : 0;
          const int rem = (int)(mm % hw);
          const int y = rem / p.W, x = rem - y * p.W;
          const int yy = y + dy, xx = x + dx;
          oka = oka && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
          src = p.a + (size_t)(oka ? mm + dy * p.W + dx : 0) * p.Cin + ci0 + lc4;
        } else {
          src = p.a + (size_t)(oka ? m : 0) * p.lda + p0 + lc4;
        }
        const float4 va = *reinterpret_cast<const float4*>(src);
        ra[i] = oka ? va : make_float4(0.f, 0.f, 0.f, 0.f);
        const bool okb = m < p.M;
        const float4 vb = *reinterpret_cast<const float4*>(p.b + (size_t)(okb ? m : 0) * p.ldb + q0 + lc4);
        rb[i] = okb ? vb : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    auto store = [&](int buf) {
      float* la = &lds[buf * 32 * (BP + BQ)];
      float* lb = la + 32 * BP;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        *reinterpret_cast<float4*>(&la[(lrow + 16 * i) * BP + lc4]) = ra[i];
        *reinterpret_cast<float4*>(&lb[(lrow + 16 * i) * BQ + lc4]) = rb[i];
      }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    load(s_begin);
    store(0);
    __syncthreads();
    int cur = 0;
    for (int s = s_begin; s < s_end; ++s) {
      const int sn = (s + 1 < s_end) ? s + 1 : s;
      load(sn);
      const float* la = &lds[cur * 32 * (BP + BQ)] + wm * 32 + (lane & 31);
      const float* lb = &lds[cur * 32 * (BP + BQ)] + 32 * BP + wn * 32 + (lane & 31);
      const int hrow = (lane >> 5) * 4;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float a = la[(kb * 8 + hrow + t) * BP];
          const float b = lb[(kb * 8 + hrow + t) * BQ];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
      }
      store(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
    if (s_begin == 0 && s_end == MS) {
      // the whole reduction of this tile: finish in place (tn_fixup skips such tiles)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const size_t o = (size_t)(p0 + row) * p.ldc + q0 + wn * 32 + (lane & 31);
        float v = acc[r];
        if (p.l2 != 0.f) v += p.l2 * p.wcur[o];
        p.c[o] = v;
      }
    } else {
      // slab: tile-local row-major [64][64]
      float* slab = d.ws + (size_t)slot * BP * BQ;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        slab[row * BQ + wn * 32 + (lane & 31)] = acc[r];
      }
    }
    if (u < u1) __syncthreads();
  }
}

// C[p][q] (+)= sum of the slabs of tile (pt,qt) in workgroup order, + l2 * Wcur[p][q]
__global__ __launch_bounds__(256) void tn_fixup(const TnDev d) {
  constexpr int BP = 64, BQ = 64;
  const TnParams& p = d.p;
  const int tile = blockIdx.x;
  const long MS = d.msteps, tb = (long)tile * MS, te = tb + MS;
  int w = (int)((tb * d.W) / d.units);
  while (w + 1 < d.W && tn_unit_begin(d.units, d.W, w + 1) <= tb) ++w;
  while (w > 0 && tn_unit_begin(d.units, d.W, w) > tb) --w;
  int w_last = (int)(((te - 1) * d.W) / d.units);
  while (w_last + 1 < d.W && tn_unit_begin(d.units, d.W, w_last + 1) <= te - 1) ++w_last;
  while (w_last > 0 && tn_unit_begin(d.units, d.W, w_last) > te - 1) --w_last;
  if (w == w_last) return;  // one workgroup reduced the whole tile and wrote it in place
  const int idx4 = blockIdx.y * 256 + threadIdx.x;
  const int lrow = idx4 / (BQ / 4), lcol = (idx4 - lrow * (BQ / 4)) * 4;
  const float* base = d.ws + (size_t)lrow * BQ + lcol;
  const size_t slab = (size_t)BP * BQ;
  const int first_slot = (tn_unit_begin(d.units, d.W, w) >= tb) ? 2 * w : 2 * w + 1;
  float4 v = *reinterpret_cast<const float4*>(base + (size_t)first_slot * slab);
#pragma unroll 4
  for (int k = w + 1; k <= w_last; ++k) {
    const float4 u = *reinterpret_cast<const float4*>(base + (size_t)(2 * k) * slab);
    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
  }
  const int pt = tile / d.qtiles, qt = tile - pt * d.qtiles;
  float* o = p.c + (size_t)(pt * BP + lrow) * p.ldc + qt * BQ + lcol;
  if (p.l2 != 0.f) {  // d(wd*|w|^2/2)/dw = wd * w
    const float4 wv = *reinterpret_cast<const float4*>(p.wcur + (size_t)(pt * BP + lrow) * p.ldc + qt * BQ + lcol);
    v.x += p.l2 * wv.x; v.y += p.l2 * wv.y; v.z += p.l2 * wv.z; v.w += p.l2 * wv.w;
  }
  *reinterpret_cast<float4*>(o) = v;
}

size_t gemm_tn_ws_bytes(long M, int P, int Q) {
  const long units = (long)(P / 64) * (Q / 64) * ((M + 31) / 32);
  const long W = units < 512 ? units : 512;
  return (size_t)2 * W * 64 * 64 * sizeof(float);
}

hipError_t gemm_tn_launch(const TnParams& p, float* ws, hipStream_t st) {
  TnDev d;
  d.p = p;
  d.ws = ws;
  d.msteps = (int)((p.M + 31) / 32);
  d.ptiles = p.P / 64;
  d.qtiles = p.Q / 64;
  d.units = (long)d.ptiles * d.qtiles * d.msteps;
  d.W = (int)(d.units < 512 ? d.units : 512);
  if (p.Cin > 0)
    hipLaunchKernelGGL((gemm_tn_f32_mfma<true>), dim3(d.W), dim3(256), 0, st, d);
  else
    hipLaunchKernelGGL((gemm_tn_f32_mfma<false>), dim3(d.W), dim3(256), 0, st, d);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(tn_fixup, dim3(d.ptiles * d.qtiles, 64 * 64 / 1024), dim3(256), 0, st, d);
  return hipGetLastError();
}

}  // namespace disn
