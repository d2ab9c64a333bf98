// The point-MLP layers (models/sdfnet.py:71-88,173-186 through utils/tf_util.py:119-184, [1,1] convolutions) of a
// BATCHED call -- B x 2048 rows, B >= 4: out[M][N] = act(f(A)[M][K] . W[K][N] + bias).  The throughput sibling of
// dense_h2.hip, built like conv_h2w.hip (same arithmetic: two-term f16 split of both operands, l_a h_b + h_a l_b +
// h_a h_b on v_mfma_f32_32x32x16_f16, fp32 accumulate; same weight image; same activation-maximum slots):
//
//  * dense_h2.hip is cut for one image's 2048 rows: 64 x 64 / 64 x 128 tiles, four k-waves per n-block, and every
//    workgroup splits its rows' operand chunk into h / l itself -- N / 64 .. N / 128 times per element, 150 - 260 VALU
//    instructions per 24 MFMAs (profiles/r02x_isa_mix.txt): at thousands of rows the split, not the matrix pipe, sets
//    the pace (0.16 - 0.24 of the f16 peak).
//  * Here a workgroup is 128 rows x 256 columns: eight n-waves, each owning one 32-column n-block for all four row
//    blocks and walking K sequentially.  The operand chunk (128 rows x 128 columns) is split once per 256 output
//    columns by 512 threads (8 float4 units each per chunk: ~230 VALU per 96 MFMAs and wave), each wave streams its own
//    weight fragments from L2 (2 KiB per 12 MFMAs), MFMAs are issued as pairs of row blocks (no back-to-back dependence).
//  * Everything dense_h2.hip does on load is kept: A = [a | a2] read in place (models/sdfnet.py:180's concat),
//    relu(A + in_bias[image]) (the deferred bias of the split global fold2/conv1), per-image maxima / scales.
//
// Summation order of an output element: k16 blocks in ascending order in ONE fp32 accumulator -- not dense_h2.hip's
// four-k-wave tree: results agree to fp32 rounding, not bit for bit.  The launcher's rule (dense_h2_launch): calls
// of >= 4 images whose rows per image are a multiple of 128 take this form, whatever the other images are.
#include "kernels.hpp"
#include "h2_common.hpp"
#include "tuning.hpp"

#include <type_traits>

namespace disn {

// MB row blocks x NWV n-waves; KC k16 blocks per chunk
template <int MB, int NWV, int KC>
__global__ __launch_bounds__(64 * NWV, NWV / 4) void dense_h2w_kernel(const DenseH2Dev D) {
  constexpr int CK = 16 * KC;          // input columns per chunk
  constexpr int KPIX = CK * 4 + 16;    // bytes per row in LDS: h plane, l plane, pad (an odd multiple of 16)
  constexpr int UPP = CK / 4;          // float4 units per row
  constexpr int BM = 32 * MB;
  constexpr int BUF = BM * KPIX;
  constexpr int NT = 64 * NWV;
  constexpr int LP = BM * UPP / NT;
  static_assert((BM * UPP) % NT == 0, "loader units");
  static_assert((KPIX / 16) % 2 == 1, "row stride");
  static_assert(2 * BUF <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];

  const DenseH2Prob& P = D.p[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, g = lane >> 5;

  // n-tile major, every XCD a contiguous eighth of the tiles (as conv_h2w_kernel)
  int l;
  {
    const int T = gridDim.x, L = blockIdx.x, q = T >> 3, r = T & 7, xcd = L & 7, idx = L >> 3;
    l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int nt = l / D.mtiles, mt = l - nt * D.mtiles;
  const int m0 = mt * BM;
  const int n0 = (nt * NWV + wn) * 32;
  const int M = P.M, N = P.N, K = P.K;
  const int NC = K / CK;

  // ---- row loader: unit u = (row, float4 of the chunk's columns); every load unconditional (a row beyond M reads
  // row 0 and is zeroed by its scale) -----------------------------------------------------------------------------
  int grow[LP], gbrow[LP];
  float gscale[LP];
  const int c4 = tid % UPP;            // NT % UPP == 0: a thread keeps its column quad
  static_assert(NT % UPP == 0, "column quad per thread");
#pragma unroll
  for (int k = 0; k < LP; ++k) {
    const int r = (tid + k * NT) / UPP;
    const bool ok = m0 + r < M;
    grow[k] = ok ? m0 + r : 0;
    gbrow[k] = P.in_bias_rows > 0 ? (grow[k] / P.in_bias_rows) * K : 0;  // in_bias row of this row's image
    gscale[k] = ok ? 1.0f : 0.0f;
  }
  const int woff0 = (tid / UPP) * KPIX + 8 * c4;       // unit k: + k (NT / UPP) KPIX
  auto load_chunk = [&](int c, float4 (&ra)[LP]) {
    const int k0 = c * CK;
    const bool first = k0 < P.k1;
    const float* src = (first ? P.a + k0 : P.a2 + (k0 - P.k1)) + 4 * c4;
    const int ld = first ? P.lda : P.lda2;
#pragma unroll
    for (int k = 0; k < LP; ++k) ra[k] = *reinterpret_cast<const float4*>(src + (size_t)grow[k] * ld);
  };
  float sa = 1.0f;
  auto store_unit = [&](int buf, int c, const float4 (&ra)[LP], int k) {
    float x[4] = {ra[k].x, ra[k].y, ra[k].z, ra[k].w};
    if (P.in_bias) {
      const float4 bb = *reinterpret_cast<const float4*>(P.in_bias + gbrow[k] + c * CK + 4 * c4);
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
    const int wo = buf * BUF + woff0 + k * (NT / UPP) * KPIX;
    *reinterpret_cast<ch_h4*>(&lds[wo]) = hh;
    *reinterpret_cast<ch_h4*>(&lds[wo + CK * 2]) = ll;
  };

  float4 ra0[LP];
  load_chunk(0, ra0);

  // ---- scales (requested here, used after the weight queue is in flight) -------------------------------------
  const int Kimg = P.Kimg > 0 ? P.Kimg : K;
  const float* meta = reinterpret_cast<const float*>(P.wimg + (size_t)Kimg * N * 4);
  const size_t aoff = P.amax_rows > 0 ? (size_t)(m0 / P.amax_rows) * P.amax_stride : 0;  // this tile's image
  float amax_lane = 0.f;
  {
    const int n1 = P.in_amax_n > 0 ? P.in_amax_n : 64;
    for (int i = lane; i < n1; i += 64) amax_lane = fmaxf(amax_lane, P.in_amax[aoff + i]);
  }
  if (P.in_amax2) {
    const int n2 = P.in_amax2_n > 0 ? P.in_amax2_n : 64;
    for (int i = lane; i < n2; i += 64) amax_lane = fmaxf(amax_lane, P.in_amax2[aoff + i]);
  }
  float bmax_lane = 0.f;
  if (P.in_bias) {  // the bound max|a| + max|in_bias| over this tile's image (dense_h2_kernel's rule)
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
    for (int mb = 0; mb < MB; ++mb) arow[mb] = (mb * 32 + Lr) * KPIX + (8 * g) * 2;
  }

  // this wave's weight stream: k16 blocks 0, 1, 2 ... of its n-block, 2 KiB each, contiguous; DQ pairs in flight
  // (slot kb % DQ): a pair is replaced by the one DQ blocks ahead as soon as its last MFMA is issued
  constexpr int DQ = 4;
  static_assert(KC % DQ == 0, "queue depth");
  const unsigned char* wp = P.wimg + ((size_t)(n0 >> 5) * (Kimg >> 4) + (P.k_begin >> 4)) * 2048 + lane * 16;
  ch_h8 qh[DQ], ql[DQ];
#pragma unroll
  for (int t = 0; t < DQ; ++t) {
    qh[t] = *reinterpret_cast<const ch_h8*>(wp + (size_t)t * 2048);
    ql[t] = *reinterpret_cast<const ch_h8*>(wp + (size_t)t * 2048 + 1024);
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

  // ---- one chunk: KC k16 blocks x MB row blocks = KC MB sub-steps (block kb = s / MB, rows s % MB), issued as pairs
  // l0 l1 | h0 h1 | h0 h1; the A fragments of the next pair are read while this pair's MFMAs run; MORE: the next
  // chunk's rows are requested at the top and split into the other buffer between the MFMAs of the later pairs --------
  static_assert(MB % 2 == 0, "pairs of row blocks");
  constexpr int S = KC * MB, NP = S / 2, NB = 4;
  constexpr int P0 = NP - LP > NP / 2 ? NP / 2 : NP - LP;   // first pair that carries a unit of the next chunk
  static_assert(LP <= NP - P0, "loader units per thread");
  auto chunk = [&](int c, auto more_c) {
    constexpr bool MORE = decltype(more_c)::value;
    float4 ra[LP];
    if (MORE) load_chunk(c + 1, ra);
    int ab[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) ab[mb] = arow[mb] + (c & 1) * BUF;
    const unsigned char* wcur = wp + (size_t)c * KC * 2048;
    ch_h8 ah[NB], al[NB];
    auto rd = [&](int s) {
      const int kb = s / MB, mb = s % MB;
      ah[s % NB] = *reinterpret_cast<const ch_h8*>(&lds[ab[mb] + kb * 32]);
      al[s % NB] = *reinterpret_cast<const ch_h8*>(&lds[ab[mb] + kb * 32 + CK * 2]);
    };
    rd(0);
    rd(1);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int s0 = 2 * p, s1 = 2 * p + 1;
      const int kb = s0 / MB, m0_ = s0 % MB, m1_ = s1 % MB;   // MB even: a pair lies in one k16 block
      if (s0 + 2 < S) rd(s0 + 2);
      if (s1 + 2 < S) rd(s1 + 2);
      __builtin_amdgcn_sched_barrier(0);
      acc[m0_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s0 % NB], qh[kb % DQ], acc[m0_], 0, 0, 0);
      acc[m1_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s1 % NB], qh[kb % DQ], acc[m1_], 0, 0, 0);
      acc[m0_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s0 % NB], ql[kb % DQ], acc[m0_], 0, 0, 0);
      acc[m1_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s1 % NB], ql[kb % DQ], acc[m1_], 0, 0, 0);
      acc[m0_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s0 % NB], qh[kb % DQ], acc[m0_], 0, 0, 0);
      acc[m1_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s1 % NB], qh[kb % DQ], acc[m1_], 0, 0, 0);
      bool any = false;
      if (MORE && p >= P0) {
#pragma unroll
        for (int k = 0; k < LP; ++k)
          if ((k * (NP - P0)) / LP == p - P0) { store_unit((c + 1) & 1, c + 1, ra, k); any = true; }
      }
      if (any) {
#pragma unroll
        for (int m6 = 0; m6 < 6; ++m6) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s1 == (kb + 1) * MB - 1 && (MORE || kb + DQ < KC)) {   // block kb is done: its slot takes the block DQ ahead
        qh[kb % DQ] = *reinterpret_cast<const ch_h8*>(wcur + (size_t)(kb + DQ) * 2048);
        ql[kb % DQ] = *reinterpret_cast<const ch_h8*>(wcur + (size_t)(kb + DQ) * 2048 + 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };
#pragma unroll 1
  for (int c = 0; c + 1 < NC; ++c) chunk(c, std::true_type{});
  chunk(NC - 1, std::false_type{});

  // ---- epilogue: bias, ReLU, row-major store (32 lanes = 128 contiguous bytes of a row), maximum of |out| ----------
  const float bias_j = P.bias[n0 + j];
  float* outb = P.out + (size_t)m0 * P.ldc + n0 + j;
  const float* addb = P.add_in ? P.add_in + (size_t)m0 * P.ldc + n0 + j : nullptr;
  float vmax = 0.f;
  const bool full = m0 + BM <= M;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r0 = mb * 32 + ch2::quad_row(q, g);   // four consecutive rows
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool live = full || m0 + r0 + e < M;
        float v = fmaf(acc[mb][4 * q + e], descale, bias_j);
        if (addb && live) v += addb[(r0 + e) * P.ldc];
        if (P.relu) v = fmaxf(v, 0.f);
        if (live) {
          outb[(r0 + e) * P.ldc] = v;
          vmax = fmaxf(vmax, fabsf(v));
        }
      }
    }
  if (P.out_amax) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off));
    if (lane == 0)
      atomicMax(reinterpret_cast<unsigned*>(P.out_amax) + aoff + ((blockIdx.x * NWV + wn) & 63), __float_as_uint(vmax));
  }
}

bool dense_h2w_supported(const DenseH2Prob& p) {
  return p.M >= 128 && p.N % 256 == 0 && p.K % 64 == 0 && (p.k1 == p.K || p.k1 % 64 == 0) && p.k_begin % 16 == 0 &&
         (p.amax_rows == 0 || p.amax_rows % 128 == 0) && (size_t)p.M * (p.ldc > p.lda ? p.ldc : p.lda) < ((size_t)1 << 31);
}

// 1 or 2 problems of ONE shape (validated by dense_h2_launch) through the batched form.  Tile rows and chunk width are
// speed knobs only (K is summed in ascending order in one accumulator whatever they are): 128-column chunks where K and
// the source boundary allow, else 64 (fold1/conv2: K = 64); 64-row tiles where 128-row ones would leave half the chip
// idle (the 256-column layers of an eight-step call: 128 -> 256 workgroups).
hipError_t dense_h2w_go(DenseH2Dev d, hipStream_t st) {
  const DenseH2Prob& p = d.p[0];
  // (128-column chunks from K = 1024 on: at K <= 512 the 64-column ones measured 3-5 % faster -- twice the workgroups per
  //  CU by LDS, a shorter prologue; profiles/r03t_dense_h2w_knobs.txt)
  bool c128 = p.K >= 1024 && p.K % 128 == 0 && (p.k1 == p.K || p.k1 % 128 == 0);
  const long wg128 = (long)((p.M + 127) / 128) * (p.N / 256) * d.nprob;
  bool m64 = wg128 < 200;
  if (tune::densew_m64 >= 0) m64 = tune::densew_m64 != 0;          // tuning builds (tools/dense_h2w_time.py)
  if (tune::densew_c128 >= 0) c128 = c128 && tune::densew_c128 != 0;
  d.mtiles = (p.M + (m64 ? 63 : 127)) / (m64 ? 64 : 128);
  const dim3 grid(d.mtiles * (p.N / 256), d.nprob);
  if (m64) {
    if (c128) hipLaunchKernelGGL((dense_h2w_kernel<2, 8, 8>), grid, dim3(512), 0, st, d);
    else hipLaunchKernelGGL((dense_h2w_kernel<2, 8, 4>), grid, dim3(512), 0, st, d);
  } else {
    if (c128) hipLaunchKernelGGL((dense_h2w_kernel<4, 8, 8>), grid, dim3(512), 0, st, d);
    else hipLaunchKernelGGL((dense_h2w_kernel<4, 8, 4>), grid, dim3(512), 0, st, d);
  }
  return hipGetLastError();
}

}  // namespace disn
