// Small-batch fully-connected layers: out[b][n] = act(sum_k x[b][k] W[k][n] + bias[n]).
// VGG fc6/fc7/fc8 (models/CNN/vgg.py:198-214) and the per-image fold of the
// global-feature block of sdfprediction/fold2/conv1 (models/sdfnet.py:78-84).
// One to three rows (a step at a time): gemv_kernel<NB> (fc6) / gemv_rows_kernel (fc7, fc8, the fold); four rows and
// more (a batched call): gemv_mfma_kernel, sixteen rows per pass on the fp32 matrix pipe.
//
// Bound: HBM (weight read): fc6 alone is 411 MB per forward.  Each wave streams whole
// 1-KiB row segments of W (64 lanes x float4, coalesced), four waves of a block take
// interleaved rows, K is split across blockIdx.y; partials are combined by
// splitk_reduce_kernel (deterministic, no atomics).  x[b][k] is wave-uniform -> scalar loads.
#include "kernels.hpp"
#include "tuning.hpp"

namespace disn {

template <int NB>
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ x, int K,
                                                   const float* __restrict__ w, int N,
                                                   float* __restrict__ partial, int Btot, int b0) {
  constexpr int RB = NB;
  __shared__ float red[4][RB][256];
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
  const int t = threadIdx.x;
#pragma unroll
  for (int r0 = 0; r0 < NB; r0 += RB) {
    if (r0) __syncthreads();
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      red[wave][b][lane * 4 + 0] = acc[r0 + b].x;
      red[wave][b][lane * 4 + 1] = acc[r0 + b].y;
      red[wave][b][lane * 4 + 2] = acc[r0 + b].z;
      red[wave][b][lane * 4 + 3] = acc[r0 + b].w;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const float v = (red[0][b][t] + red[1][b][t]) + (red[2][b][t] + red[3][b][t]);
      partial[((size_t)z * Btot + b0 + r0 + b) * N + blockIdx.x * 256 + t] = v;
    }
  }
}

// Four batch rows and more (a batched call): the same split-K stream with the products on the fp32 matrix pipe.
// Round 6: v_mfma_f32_16x16x4_f32 -- D[i][j] += sum_{k < 4} A[i][k] B[k][j], lane l holding A[i = l % 16][k = l / 16] and
// B[k = l / 16][j = l % 16].  A wave takes FOUR consecutive k rows per step: lane (j, k) loads the float4 W[k0 + k][col0 +
// 64 cb + 4 j ..] of each of the four 64-column blocks cb (16 lanes = 256 contiguous bytes of a row, four rows per
// instruction) -- component c of that float4 is the B operand of the MFMA for columns 64 cb + 4 j + c -- and the ONE x
// value x[row j][k0 + k] as the A operand: sixteen MFMAs per step as before (round 4: 16x16x1 in four blocks, one product
// per MFMA), but the four products of a step are summed INSIDE the MFMA before they touch the accumulator -- a wave's chain
// on one accumulator is a quarter as long (fc6: 25 instead of 98: the VALU kernels' length; profiles/r06o_sweep_diag2_set35.txt
// had the one-product form at 2 x the VALU head's error).  acc[c][4 cb + r] of lane l: batch row 4 (l / 16) + r, column 64 cb
// + 4 (l % 16) + c of the workgroup's 256 (the layout the epilogue always had).  The workgroup's four waves take
// consecutive steps; partial slabs and reduce pass as before.  Summation order of one output: k quads ascending inside a wave,
// (w0 + w1) + (w2 + w3), then splitk_reduce_kernel: fixed by (K, N, S) -- never by the batch.
__global__ __launch_bounds__(256) void gemv_mfma_kernel(const float* __restrict__ x, int K,
                                                        const float* __restrict__ w, int N,
                                                        float* __restrict__ partial, int Btot, int b0) {
  typedef float v16f __attribute__((ext_vector_type(16)));
  typedef float v4f __attribute__((ext_vector_type(4)));
  __shared__ float red[4][8][4][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int S = gridDim.y, z = blockIdx.y;
  const int K4 = K >> 2;
  const int kbeg = 4 * (int)(((long)K4 * z) / S), kend = 4 * (int)(((long)K4 * (z + 1)) / S);
  const int col0 = blockIdx.x * 256;
  const int row = lane & 15, kq = lane >> 4;
  const bool live = b0 + row < Btot;
  const float* xr = x + (size_t)(live ? b0 + row : b0) * K + kq;
  const float* wp = w + col0 + 4 * row + (size_t)kq * N;
  v4f acc4[4][4];   // [c][cb]
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc4[c][cb] = v4f{0.f, 0.f, 0.f, 0.f};
  // two register sets: the four row segments (and the x value) of step i + 1 are requested before the sixteen MFMAs of
  // step i are issued (left to itself the compiler waits for every load right behind its issue: one KiB in flight per wave)
  float4 wv[2][4];
  float xv[2];
  auto load = [&](int k, float4 (&wq)[4], float& xq) {
    const int kk = k < kend ? k : kbeg;            // past the end: a valid address, the values are not used
    xq = xr[kk];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) wq[cb] = nt_load4(wp + (size_t)kk * N + 64 * cb);
  };
  auto mac = [&](const float4 (&wq)[4], float xq) {
    if (!live) xq = 0.f;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      acc4[0][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq, wq[cb].x, acc4[0][cb], 0, 0, 0);
      acc4[1][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq, wq[cb].y, acc4[1][cb], 0, 0, 0);
      acc4[2][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq, wq[cb].z, acc4[2][cb], 0, 0, 0);
      acc4[3][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq, wq[cb].w, acc4[3][cb], 0, 0, 0);
    }
  };
  const int k0 = kbeg + 4 * wave;
  const int n = k0 < kend ? (kend - k0 + 15) >> 4 : 0;   // steps of this wave
  load(k0, wv[0], xv[0]);
  for (int i = 0; i + 1 < n; i += 2) {                   // the same loads in flight at the top from both entries
    load(k0 + 16 * (i + 1), wv[1], xv[1]);
    __builtin_amdgcn_sched_barrier(0);
    mac(wv[0], xv[0]);
    __builtin_amdgcn_sched_barrier(0);
    load(k0 + 16 * (i + 2), wv[0], xv[0]);
    __builtin_amdgcn_sched_barrier(0);
    mac(wv[1], xv[1]);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (n & 1) mac(wv[0], xv[0]);
  v16f acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[c][4 * cb + r] = acc4[c][cb][r];
  // (w0 + w1) + (w2 + w3) through LDS, eight accumulator registers (two column blocks of 64) per round
  const int t = threadIdx.x;
#pragma unroll
  for (int r0 = 0; r0 < 16; r0 += 8) {
    if (r0) __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 8; ++rr)
#pragma unroll
      for (int c = 0; c < 4; ++c) red[wave][rr][c][lane] = acc[c][r0 + rr];
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int u = h * 256 + t, rr = u >> 6, l = u & 63;
      float o[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = (red[0][rr][c][l] + red[1][rr][c][l]) + (red[2][rr][c][l] + red[3][rr][c][l]);
      const int r = r0 + rr;
      const int brow = b0 + 4 * (l >> 4) + (r & 3);
      if (brow < Btot)
        *reinterpret_cast<float4*>(partial + ((size_t)z * Btot + brow) * N + col0 + 64 * (r >> 2) + 4 * (l & 15)) =
            make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// The same layer from the TRANSPOSED weight matrix wt [N][K] (one contiguous K-long row per output): a wave owns R
// consecutive outputs and streams their rows (non-temporal, 64 lanes x 16 B = 1 KiB per instruction, U row pieces
// in flight per row) against x; no K split, so no partial slabs and no reduce launch -- one launch per layer
// instead of two, and at fc7 / fc8 sizes (67 / 17 MB, launch-latency bound) each of them is shorter.  x (16-100 KB)
// is re-read by every wave from L1 / L2.  Fixed summation order: per lane over k = 4 lane + 256 i, then the
// xor-shuffle tree.
template <int NB, int R, int U>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const float* __restrict__ x, int K,
                                                        const float* __restrict__ wt, int N,
                                                        const float* __restrict__ bias, int relu,
                                                        float* __restrict__ out, int b0) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
  const int n0 = wave * R;
  if (n0 >= N) return;
  float acc[R][NB];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
  const float* wr = wt + (size_t)n0 * K + lane * 4;
  const float* xr = x + (size_t)b0 * K + lane * 4;
  for (int k = 0; k < K; k += 256 * U) {
    float4 wv[R][U];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const bool ok = k + 256 * u + lane * 4 < K && n0 + r < N;
        wv[r][u] = ok ? nt_load4(wr + (size_t)r * K + k + 256 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = k + 256 * u + lane * 4 < K;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float4 xv = ok ? *reinterpret_cast<const float4*>(xr + (size_t)b * K + k + 256 * u)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < R; ++r)
          acc[r][b] += (wv[r][u].x * xv.x + wv[r][u].y * xv.y) + (wv[r][u].z * xv.z + wv[r][u].w * xv.w);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float v = acc[r][b];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0 && n0 + r < N) {
        v += bias[n0 + r];
        if (relu) v = fmaxf(v, 0.f);
        out[(size_t)(b0 + b) * N + n0 + r] = v;
      }
    }
}

// wt_nk: [N][K] (the TF [K][N] matrix transposed); K % 4 == 0
hipError_t gemv_rows_launch(const float* x, int B, int K, const float* wt_nk, const float* bias, int N, int relu,
                            float* out, hipStream_t st) {
#ifdef DISN_TUNING
  if (tune::gemv_rows_cfg > 0 && B == 1) {   // (R, U) experiments on the one-row form: the bits do not depend on them
    const int c = tune::gemv_rows_cfg;
    const int R = c == 2 || c == 5 ? 2 : (c == 4 ? 4 : 1);
    const dim3 g(((N + R - 1) / R + 3) / 4);
    switch (c) {
      case 1: hipLaunchKernelGGL((gemv_rows_kernel<1, 1, 8>), g, dim3(256), 0, st, x, K, wt_nk, N, bias, relu, out, 0); break;
      case 2: hipLaunchKernelGGL((gemv_rows_kernel<1, 2, 8>), g, dim3(256), 0, st, x, K, wt_nk, N, bias, relu, out, 0); break;
      case 3: hipLaunchKernelGGL((gemv_rows_kernel<1, 1, 16>), g, dim3(256), 0, st, x, K, wt_nk, N, bias, relu, out, 0); break;
      case 4: hipLaunchKernelGGL((gemv_rows_kernel<1, 4, 4>), g, dim3(256), 0, st, x, K, wt_nk, N, bias, relu, out, 0); break;
      case 5: hipLaunchKernelGGL((gemv_rows_kernel<1, 2, 4>), g, dim3(256), 0, st, x, K, wt_nk, N, bias, relu, out, 0); break;
      default: hipLaunchKernelGGL((gemv_rows_kernel<1, 1, 4>), g, dim3(256), 0, st, x, K, wt_nk, N, bias, relu, out, 0); break;
    }
    return hipGetLastError();
  }
#endif
  // two rows per wave when there are enough outputs to fill the chip that way (halves the x re-reads)
  const bool two = N >= 4096;
  const int waves = two ? (N + 1) / 2 : N;
  const dim3 grid((waves + 3) / 4);
  // Round 6 (tools/fc_rows_time.py, cold weights, one row): row pieces in flight per lane by K for the one-row-per-wave
  // layers -- sixteen for the 4096-long rows of fc8 (16.8 -> 12.8 us), four for the 1000-long rows of the bias fold (18.3
  // -> 11.9 us: the eight-deep form spent its time in masked tail pieces).  U does not change a lane's k order: the same
  // bits.  (fc7 with ONE row per wave is 6 us faster too, but the compiler contracts that body differently: other bits
  // for the single-image forms -- on the 48-set sweep their worst request moved from 8.8e-6 to 9.9e-6, r06t; not taken.)
  const int u1 = K >= 4096 ? 16 : 4;
  // up to eight batch rows per launch: the matrix is read once per launch, and a batch row's sum is the same
  // instruction sequence whatever NB (an eight-step call must not read fc7's 67 MB twice)
  for (int b0 = 0; b0 < B; b0 += 8) {
    const int nb = (B - b0) < 8 ? (B - b0) : 8;
    switch (nb) {
#define DISN_GR_CASE(NB)                                                                                          \
  case NB:                                                                                                        \
    if (two) hipLaunchKernelGGL((gemv_rows_kernel<NB, 2, 4>), grid, dim3(256), 0, st, x, K, wt_nk, N, bias, relu, out, b0); \
    else if (u1 == 16) hipLaunchKernelGGL((gemv_rows_kernel<NB, 1, 16>), grid, dim3(256), 0, st, x, K, wt_nk, N, bias, relu, out, b0); \
    else hipLaunchKernelGGL((gemv_rows_kernel<NB, 1, 4>), grid, dim3(256), 0, st, x, K, wt_nk, N, bias, relu, out, b0);    \
    break;
      DISN_GR_CASE(1) DISN_GR_CASE(2) DISN_GR_CASE(3) DISN_GR_CASE(4)
      DISN_GR_CASE(5) DISN_GR_CASE(6) DISN_GR_CASE(7) DISN_GR_CASE(8)
#undef DISN_GR_CASE
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// K splits.  One to three batch rows (a step at a time): 2048 workgroups = every wave slot of the chip, what a lone
// latency-bound stream wants (r01).  Four rows and more (a batched call): 1024 -- each workgroup already carries B
// accumulator sets, the partial slabs (S x B x N floats written and re-read) shrink 2x and the reduce pass with them
// (512: fc6 beside the point-MLP layers of the call's tail 84 -> 191 us, too few waves for its share of HBM; r03e)
// (fc6 of an eight-step call: 16.8 MB of partials, a 45 us reduce).  By the batch size only: the summation order of a
// row never depends on the other rows' data.
int gemv_splits(int K, int N, int B) {
  const int colblocks = N / 256;
  const int wgs = tune::gemv_wgs > 0 ? tune::gemv_wgs : (B >= tune::conv_wide_min ? 1024 : 2048);
  int s = (wgs + colblocks - 1) / colblocks;
  // at least 32 k rows per split (a one-row call: 2048 workgroups want them short); a batched call: at least 128 -- its
  // small layers otherwise write more partial bytes than they read weights (round 5: fc8 128 splits, the bias fold 31:
  // 8 / 16 MB of slabs for 16 / 2 MB of weights, a 42 us reduce pass; round 6: 32 / 7)
  const int per = B >= tune::conv_wide_min ? 128 : 32;
  const int smax = K / per > 0 ? K / per : 1;
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  return s;
}

size_t gemv_ws_bytes(int B, int K, int N) {
  // (the single-row split count: what gemv_launch(single_form) uses for any B -- twice the batched form's)
  return (size_t)gemv_splits(K, N, 1) * B * N * sizeof(float);
}

// single_form ("strict", disn_vgg_weights_t.strict_forms = 1): the split count, the VALU kernels and the reduce lanes of a
// call of ONE row for any B (eight rows per pass over the matrix) -- every row bit for bit what it gets alone
hipError_t gemv_launch(const float* x, int B, int K, const float* w_kn, const float* bias, int N,
                       int relu, float* out, float* ws, hipStream_t st, bool single_form) {
  const int S = gemv_splits(K, N, single_form ? 1 : B);
  dim3 grid(N / 256, S);
  if (!single_form && B >= tune::conv_wide_min && K % 4 == 0) {   // a batched call: sixteen rows per pass on the matrix pipe
    for (int b0 = 0; b0 < B; b0 += 16) {
      hipLaunchKernelGGL(gemv_mfma_kernel, grid, dim3(256), 0, st, x, K, w_kn, N, ws, B, b0);
      const hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
    }
    // the lane count a call of four rows gets, for every B (sixteen rows of N = 4096 would otherwise cross the
    // launcher's threshold and sum the slabs in another order than eight)
    return splitk_reduce_launch(ws, S, B, N, bias, 0, relu, out, N, st, S < 4 ? 1 : (S < 16 ? 4 : 16));
  }
  for (int b0 = 0; b0 < B;) {
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
    b0 += nb;
  }
  // (single_form: the lane count the reduce picks for ONE row, whatever B)
  const size_t total1 = (size_t)(N / 4);
  const int sl1 = (total1 >= 131072 || S < 4) ? 1 : ((total1 >= 16384 || S < 16) ? 4 : 16);
  return splitk_reduce_launch(ws, S, B, N, bias, 0, relu, out, N, st, single_form ? sl1 : 0);
}

}  // namespace disn
