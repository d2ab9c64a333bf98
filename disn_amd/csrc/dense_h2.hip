// The point-MLP layers at a few thousand rows (models/sdfnet.py:71-88,173-186 through utils/tf_util.py:119-184,
// [1,1] convolutions): out[M][N] = act(f(A)[M][K] . W[K][N] + bias), fp32 in, fp32 out, fp32-accurate products on
// the f16 matrix pipes -- the 1x1 sibling of conv_h2.hip, same building blocks:
//   * two-term f16 split of both operands (weights at pack time -- the conv_h2 image with one tap; activations
//     when a 64 / 256-channel chunk of BM rows is staged in LDS as h / l planes, scale from the producer's
//     activation maximum), three v_mfma_f32_32x32x16_f16 per 32x32x16 block;
//   * four k-waves per n-block: wave wk owns k16 blocks 4 t + wk of a chunk (t < KPW) for the whole BM x 32 tile,
//     streams its weight fragments straight from L2 (one chunk ahead), and the four partial tiles are summed
//     through LDS in the fixed order (w0 + w2) + (w1 + w3); every wave then finishes a quarter of the rows;
//   * A-fragment rows via sigma (h2_common.hpp): 16 consecutive rows per ds_read_b128 lane group, row stride
//     CK * 4 + 16 bytes (an odd multiple of 16): conflict-free.
// Why not the f32-input GEMM of gemm_mfma.hip here: at 2048 rows a layer is 128-256 tiles, i.e. launch- and
// latency-bound (7-15 us for 0.07-1.1 GFLOP); with 1/8 of the MFMA cycles and no split-K / stream-K fix-up pass
// it is one short launch, and TWO independent problems of one shape (the same layer of the global and the local
// stream) share a launch (blockIdx.y).
// On load, A can be [a (k1 columns) | a2 (K - k1 columns)] (the tf.concat of models/sdfnet.py:180 read in place)
// and can be transformed as relu(A + in_bias[k]) -- the deferred bias + ReLU of a layer whose product was
// computed before its per-image bias existed (the split global fold2/conv1 of disn_encode_query).
#include "kernels.hpp"
#include "h2_common.hpp"
#include "tuning.hpp"

#include <type_traits>

namespace disn {

template <int MB, int NW, int KPW>
__global__ __launch_bounds__(256 * NW, 1) void dense_h2_kernel(const DenseH2Dev D) {
  constexpr int WK = 4;
  constexpr int CK = 16 * WK * KPW;   // input columns per chunk
  constexpr int KPIX = CK * 4 + 16;   // bytes per row in LDS: h plane, l plane, pad
  constexpr int UPP = CK / 4;         // float4 units per row
  constexpr int BM = 32 * MB;
  constexpr int BUF = BM * KPIX;
  constexpr int NT = 256 * NW;
  constexpr int LP = (BM * UPP + NT - 1) / NT;
  constexpr int XCH = NW * WK * MB * 4096;
  constexpr int LDS_BYTES = 2 * BUF > XCH ? 2 * BUF : XCH;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static_assert((BM * UPP) % NT == 0, "loader units");
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const DenseH2Prob& P = D.p[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave % WK, wn = wave / WK;
  const int j = lane & 31, g = lane >> 5;

  // n-tile major, every XCD a contiguous eighth of the tiles (as conv_h2_kernel)
  int l;
  {
    const int T = gridDim.x, L = blockIdx.x, q = T >> 3, r = T & 7, xcd = L & 7, idx = L >> 3;
    l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int nt = l / D.mtiles, mt = l - nt * D.mtiles;
  const int m0 = mt * BM;
  const int n0 = (nt * NW + wn) * 32;
  const int M = P.M, N = P.N, K = P.K;
  const int NC = K / CK;

  // ---- row loader: unit u = (row, float4 of the chunk's columns); every load unconditional (a row beyond M reads
  // row 0 and is zeroed by its scale), see conv_h2_kernel ----------------------------------------------------------
  int grow[LP], woff[LP], gcol[LP], gbrow[LP];
  float gscale[LP];
#pragma unroll
  for (int k = 0; k < LP; ++k) {
    const int u = tid + k * NT;
    const int r = u / UPP, c4 = u % UPP;
    const bool ok = m0 + r < M;
    grow[k] = ok ? m0 + r : 0;
    gbrow[k] = P.in_bias_rows > 0 ? (grow[k] / P.in_bias_rows) * K : 0;  // in_bias row of this row's image
    gcol[k] = 4 * c4;
    gscale[k] = ok ? 1.0f : 0.0f;
    woff[k] = r * KPIX + 8 * c4;
  }
  auto load_chunk = [&](int c, float4 (&ra)[LP]) {
    const int k0 = c * CK;
    const bool first = k0 < P.k1;
    const float* src = first ? P.a + k0 : P.a2 + (k0 - P.k1);
    const int ld = first ? P.lda : P.lda2;
#pragma unroll
    for (int k = 0; k < LP; ++k) ra[k] = *reinterpret_cast<const float4*>(src + (size_t)grow[k] * ld + gcol[k]);
  };
  float sa = 1.0f;
  auto store_unit = [&](int buf, int c, const float4 (&ra)[LP], int k) {
    float x[4] = {ra[k].x, ra[k].y, ra[k].z, ra[k].w};
    if (P.in_bias) {
      const float4 bb = *reinterpret_cast<const float4*>(P.in_bias + gbrow[k] + c * CK + gcol[k]);
      x[0] = fmaxf(x[0] + bb.x, 0.f); x[1] = fmaxf(x[1] + bb.y, 0.f);
      x[2] = fmaxf(x[2] + bb.z, 0.f); x[3] = fmaxf(x[3] + bb.w, 0.f);
    }
    const float s = sa * gscale[k];
    ch_h4 hh, ll;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = x[e] * s;
      const _Float16 h = (_Float16)v;
      hh[e] = h;
      ll[e] = (_Float16)(v - (float)h);
    }
    *reinterpret_cast<ch_h4*>(&lds[buf * BUF + woff[k]]) = hh;
    *reinterpret_cast<ch_h4*>(&lds[buf * BUF + woff[k] + CK * 2]) = ll;
  };

  float4 ra0[LP];
  load_chunk(0, ra0);

  // ---- scales (requested here, used after the weight queue is in flight) -------------------------------------
  const int Kimg = P.Kimg > 0 ? P.Kimg : K;
  const float* meta = reinterpret_cast<const float*>(P.wimg + (size_t)Kimg * N * 4);
  const size_t aoff = P.amax_rows > 0 ? (size_t)(m0 / P.amax_rows) * P.amax_stride : 0;  // this tile's image
  float amax_lane = P.in_amax[aoff + lane];
  if (P.in_amax2) {
    const int n2 = P.in_amax2_n > 0 ? P.in_amax2_n : 64;
    for (int i = lane; i < n2; i += 64) amax_lane = fmaxf(amax_lane, P.in_amax2[aoff + i]);
  }
  float bmax_lane = 0.f;
  if (P.in_bias) {
    // the bound max|a| + max|in_bias|: over this tile's image when the maxima are per image (tiles do not straddle
    // images then: the scale must not depend on the batch), else over every bias row of the call
    const float* ib = P.in_bias;
    int nb = P.in_bias_rows > 0 ? ((M + P.in_bias_rows - 1) / P.in_bias_rows) * K : K;
    if (P.amax_rows > 0 && P.in_bias_rows > 0) { ib += (size_t)(m0 / P.in_bias_rows) * K; nb = K; }
    for (int i = lane; i < nb; i += 64) bmax_lane = fmaxf(bmax_lane, fabsf(ib[i]));
  }
  const float inv_sw = meta[n0 + (lane & 31)];  // per output channel (column): the pack scales every column to [2^13, 2^14)

  int arow[MB];
  {
    const int Lr = ch2::sigma(lane & 31);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) arow[mb] = (mb * 32 + Lr) * KPIX + (16 * wk + 8 * g) * 2;
  }

  // weight fragments of this wave: k16 block (c KPW + t) 4 + wk, t < KPW: 2 KiB each, 8 KiB apart
  const unsigned char* wp = P.wimg + ((size_t)(n0 >> 5) * (Kimg >> 4) + wk) * 2048 + lane * 16;
  ch_h8 qh[KPW], ql[KPW];
#pragma unroll
  for (int t = 0; t < KPW; ++t) {
    qh[t] = *reinterpret_cast<const ch_h8*>(wp + (size_t)t * WK * 2048);
    ql[t] = *reinterpret_cast<const ch_h8*>(wp + (size_t)t * WK * 2048 + 1024);
  }

  ch_f16v acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    amax_lane = fmaxf(amax_lane, __shfl_xor(amax_lane, off));
    bmax_lane = fmaxf(bmax_lane, __shfl_xor(bmax_lane, off));
  }
  sa = ch2::pow2_scale(amax_lane + bmax_lane, 14);   // |relu(a + b)| <= max|a| + max|b|
  const float descale = (1.0f / sa) * inv_sw;

#pragma unroll
  for (int k = 0; k < LP; ++k) store_unit(0, 0, ra0, k);
  __syncthreads();

  auto chunk = [&](int c, auto more_c) {
    constexpr bool MORE = decltype(more_c)::value;
    float4 ra[LP];
    if (MORE) load_chunk(c + 1, ra);
    const unsigned char* A = &lds[(c & 1) * BUF];
    const unsigned char* wnext = wp + (size_t)(c + 1) * KPW * WK * 2048;
#pragma unroll
    for (int t = 0; t < KPW; ++t) {
      const ch_h8 bh = qh[t], bl = ql[t];
      if (MORE) {
        qh[t] = *reinterpret_cast<const ch_h8*>(wnext + (size_t)t * WK * 2048);
        ql[t] = *reinterpret_cast<const ch_h8*>(wnext + (size_t)t * WK * 2048 + 1024);
      }
      ch_h8 ah[MB], al[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        ah[mb] = *reinterpret_cast<const ch_h8*>(A + arow[mb] + t * WK * 32);
        al[mb] = *reinterpret_cast<const ch_h8*>(A + arow[mb] + t * WK * 32 + CK * 2);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mb], bh, acc[mb], 0, 0, 0);
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], bl, acc[mb], 0, 0, 0);
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], bh, acc[mb], 0, 0, 0);
      }
      if (MORE) {  // the next chunk's rows are split between the MFMAs, LP / KPW units per step
#pragma unroll
        for (int k = 0; k < LP; ++k)
          if ((k * KPW) / LP == t) store_unit((c + 1) & 1, c + 1, ra, k);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };
#pragma unroll 1
  for (int c = 0; c + 1 < NC; ++c) chunk(c, std::true_type{});
  chunk(NC - 1, std::false_type{});

  // ---- (w0 + w2) + (w1 + w3) through LDS; wave wk finishes register quad wk of every block -----------------------
  float* xch = reinterpret_cast<float*>(lds);
  auto xaddr = [&](int slot, int mb, int r) { return (((wn * WK + slot) * MB + mb) * 16 + r) * 64 + lane; };
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) xch[xaddr(wk, mb, r)] = acc[mb][r];
  __syncthreads();

  const float bias_j = P.bias[n0 + j];
  const int L0 = ch2::sigma(8 * wk + 4 * g);
  float vmax = 0.f;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * wk + e, m = m0 + mb * 32 + L0 + e;
      float v = (xch[xaddr(0, mb, r)] + xch[xaddr(2, mb, r)]) + (xch[xaddr(1, mb, r)] + xch[xaddr(3, mb, r)]);
      v = fmaf(v, descale, bias_j);
      if (P.relu) v = fmaxf(v, 0.f);
      if (m < M) {
        P.out[(size_t)m * P.ldc + n0 + j] = v;
        vmax = fmaxf(vmax, fabsf(v));
      }
    }
  if (P.out_amax) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off));
    if (lane == 0)
      atomicMax(reinterpret_cast<unsigned*>(P.out_amax) + aoff + ((blockIdx.x * WK * NW + wave) & 63), __float_as_uint(vmax));
  }
}

bool dense_h2_supported(int M, int K, int N, int k1) {
  if (M <= 0 || N <= 0 || N % 64 || K <= 0 || K % 64) return false;
  const int ck = K % 256 == 0 && (k1 == K || k1 % 256 == 0) ? 256 : 64;
  return k1 > 0 && k1 <= K && k1 % ck == 0;
}

// 1 or 2 problems of ONE shape (M, N, K, k1) in one launch
hipError_t dense_h2_launch(const DenseH2Prob* probs, int nprob, hipStream_t st) {
  DenseH2Dev d{};
  if (nprob < 1 || nprob > 2) return hipErrorInvalidValue;
  for (int i = 0; i < nprob; ++i) {
    d.p[i] = probs[i];
    if (!d.p[i].a2) { d.p[i].a2 = d.p[i].a; d.p[i].lda2 = d.p[i].lda; d.p[i].k1 = d.p[i].K; }
    if (probs[i].M != probs[0].M || probs[i].N != probs[0].N || probs[i].K != probs[0].K || d.p[i].k1 != d.p[0].k1)
      return hipErrorInvalidValue;
    if (d.p[i].Kimg != 0 && d.p[i].Kimg < d.p[i].K) return hipErrorInvalidValue;
  }
  d.nprob = nprob;
  const DenseH2Prob& p = d.p[0];
  // Rows of >= kConvWideMinImages images (a batched call): the batched form of dense_h2w.hip (another K summation
  // order: fp32 rounding apart from the tiles below, not bit for bit; the rule looks at the call's image count and
  // the rows per image only, so an image's bits never depend on its companions)
  if (!p.no_wide && p.amax_rows > 0 && p.amax_rows % 128 == 0 && p.M / p.amax_rows >= tune::conv_wide_min) {
    bool ok = true;
    for (int i = 0; i < nprob; ++i) ok = ok && dense_h2w_supported(d.p[i]) && d.p[i].amax_rows == p.amax_rows;
    if (ok) return dense_h2w_go(d, st);
  }
  for (int i = 0; i < nprob; ++i)
    if (d.p[i].k_begin || d.p[i].add_in || d.p[i].in_amax_n) return hipErrorInvalidValue;  // the batched form's extras
  // Tile rows BM = 32 MB.  Neither the tile shape nor the chunk width changes a result bit (a k-wave adds its k16
  // blocks in ascending order and the four partial tiles are summed in one fixed order whatever the tiling):
  //   64 rows when that gives ~200 workgroups, else 32 (a single image's 2048 rows: latency-bound); 128 rows
  //   (instantiated, tuning builds) measured no faster than 64 at 8192 rows.
  const long t64 = (long)((p.M + 63) / 64) * (p.N / 64) * nprob;
  int mb = t64 >= 192 ? 2 : 1;
  if (tune::dense_mb > 0) mb = tune::dense_mb;
  for (int i = 0; i < nprob; ++i)
    if (d.p[i].amax_rows > 0 && d.p[i].amax_rows % (32 * mb)) return hipErrorInvalidValue;  // a tile lies in one image
  d.mtiles = (p.M + 32 * mb - 1) / (32 * mb);
  // 128 columns (16 waves) when 64 x 128 tiles still give a full round: the batched calls' thousands of rows.  Every
  // workgroup splits its rows' operand chunk itself, so at 64 columns the split is done N / 64 times per element --
  // measured at 8192 rows (tools/dense_h2_mb.py): 2048 -> 512: 123 -> 86 us, 512 -> 512: 50 -> 34 us.
  const long t128 = (long)((p.M + 63) / 64) * (p.N / 128) * nprob;
  int nw = mb == 2 && p.N % 128 == 0 && t128 >= 256 ? 4 : 2;
  if (tune::dense_nw > 0) nw = tune::dense_nw == 4 && mb == 2 && p.N % 128 == 0 ? 4 : 2;
  const dim3 grid(d.mtiles * (p.N / (32 * nw)), nprob);
  const bool c256 = p.K % 256 == 0 && p.k1 % 256 == 0;  // 256-column chunks (four k16 blocks per wave and chunk)
  const bool c128 = p.K % 128 == 0 && p.k1 % 128 == 0;
  if (nw == 4) {
    if (c256 && tune::dense_kpw == 4) hipLaunchKernelGGL((dense_h2_kernel<2, 4, 4>), grid, dim3(1024), 0, st, d);
    else if (c128) hipLaunchKernelGGL((dense_h2_kernel<2, 4, 2>), grid, dim3(1024), 0, st, d);
    else hipLaunchKernelGGL((dense_h2_kernel<2, 4, 1>), grid, dim3(1024), 0, st, d);
  } else if (mb == 4) {
    if (c128) hipLaunchKernelGGL((dense_h2_kernel<4, 2, 2>), grid, dim3(512), 0, st, d);
    else hipLaunchKernelGGL((dense_h2_kernel<4, 2, 1>), grid, dim3(512), 0, st, d);
  } else if (c256) {
    if (mb == 2) hipLaunchKernelGGL((dense_h2_kernel<2, 2, 4>), grid, dim3(512), 0, st, d);
    else hipLaunchKernelGGL((dense_h2_kernel<1, 2, 4>), grid, dim3(512), 0, st, d);
  } else {
    if (mb == 2) hipLaunchKernelGGL((dense_h2_kernel<2, 2, 1>), grid, dim3(512), 0, st, d);
    else hipLaunchKernelGGL((dense_h2_kernel<1, 2, 1>), grid, dim3(512), 0, st, d);
  }
  return hipGetLastError();
}

}  // namespace disn

// ---- C ABI: one layer as a unit (tests, composition) ---------------------------------------------------------------
#include "../../include/disn_amd.h"

extern "C" {

size_t disn_pack_dense_h2_bytes(int K, int N) {
  if (K <= 0 || N <= 0 || K % 64 || N % 64) return 0;
  return disn::h2_image_bytes(K, N, 1);
}

int disn_pack_dense_h2(const float* w_kn, int K, int N, void* image, void* stream) {
  if (!w_kn || !image || K <= 0 || N <= 0) return DISN_E_ARG;
  if (K % 64 || N % 64) return DISN_E_SHAPE;
  const hipError_t e = disn::conv_h2_pack_launch(w_kn, K, N, image, nullptr, (hipStream_t)stream, 1);
  return e == hipSuccess ? 0 : (int)e;
}

size_t disn_dense_h2_workspace_bytes(int images) { return (size_t)(images > 0 ? images : 1) * 1024; }

int disn_dense_h2(const float* a1, int lda1, int k1, const float* a2, int lda2, int k2, const float* in_bias, int M,
                  int rows_per_image, const void* image, int image_k, int k_begin, const float* add_in, const float* bias,
                  int N, int relu, float* out, float* out_amax, void* ws, size_t ws_bytes, void* stream) {
  if (!a1 || !image || !bias || !out || !ws || M <= 0 || k1 <= 0 || k2 < 0 || (k2 > 0 && !a2) || rows_per_image < 0)
    return DISN_E_ARG;
  const int K = k1 + k2;
  if (!disn::dense_h2_supported(M, K, N, k1) || lda1 != k1 || (k2 > 0 && lda2 != k2)) return DISN_E_SHAPE;
  if (image_k < 0 || k_begin < 0 || k_begin % 16 || (image_k > 0 && k_begin + K > image_k) || (image_k == 0 && k_begin)) return DISN_E_SHAPE;
  if (rows_per_image > 0 && (M % rows_per_image || rows_per_image % 64)) return DISN_E_SHAPE;
  const int imgs = rows_per_image > 0 ? M / rows_per_image : 1;
  const size_t rows = rows_per_image > 0 ? rows_per_image : M;
  if (ws_bytes < (size_t)imgs * 1024) return DISN_E_WS;
  hipStream_t st = (hipStream_t)stream;
  // per image (256 floats apart): 64 slots max |a1|, 64 slots max |a2|, 64 slots max |out|
  float* s1 = static_cast<float*>(ws);
  hipError_t e = hipMemsetAsync(s1, 0, (size_t)imgs * 1024, st);
  if (e == hipSuccess) e = disn::amax64_accumulate_launch(a1, rows * k1, s1, st, imgs, 256);
  if (e == hipSuccess && k2 > 0) e = disn::amax64_accumulate_launch(a2, rows * k2, s1 + 64, st, imgs, 256);
  if (e != hipSuccess) return (int)e;
  disn::DenseH2Prob p{};
  p.a = a1; p.lda = lda1; p.k1 = k1; p.a2 = k2 > 0 ? a2 : nullptr; p.lda2 = lda2; p.in_bias = in_bias;
  p.wimg = static_cast<const unsigned char*>(image); p.bias = bias; p.in_amax = s1; p.in_amax2 = k2 > 0 ? s1 + 64 : nullptr;
  p.out = out; p.ldc = N; p.out_amax = out_amax ? s1 + 128 : nullptr; p.M = M; p.N = N; p.K = K; p.relu = relu;
  p.Kimg = image_k; p.k_begin = k_begin; p.add_in = add_in;
  if (rows_per_image > 0) {
    p.amax_rows = rows_per_image; p.amax_stride = 256;
    if (in_bias) p.in_bias_rows = rows_per_image;
  }
  e = disn::dense_h2_launch(&p, 1, st);
  if (e == hipErrorInvalidValue) return DISN_E_SHAPE;   // k_begin / add_in outside the batched form
  if (e == hipSuccess && out_amax) e = disn::amax_fold_launch(s1 + 128, out_amax, st, 64, imgs, 256);
  return e == hipSuccess ? 0 : (int)e;
}

}  // extern "C"
