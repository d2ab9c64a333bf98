// The small kernels of the training step (reference: train/train_sdf.py:221-268,
// models/model_normalization.py:254-300).  All of them are HBM-bound streaming passes or tiny
// reductions; every reduction is a fixed-order two-stage sum (partials per row chunk, then a finish
// kernel), so gradients are reproducible run to run.
#include "kernels.hpp"

namespace disn {

// ---------------------------------------------------------------------------
// weight re-packing for the data-gradient GEMMs (weights change every step)
// ---------------------------------------------------------------------------
// packed W^T ([N][K] matrix, disn_pack_kn order): value (r, c) = W[c][r], r < N, c < K
__global__ __launch_bounds__(256) void pack_kn_T_kernel(const float* __restrict__ w, int K, int N,
                                                        float* __restrict__ packed) {
  const size_t total = (size_t)K * N;
  const int nb32 = K >> 5;  // columns of W^T in blocks of 32
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(i & 3), lane = (int)((i >> 2) & 63);
    const size_t blk = i >> 8;
    const int r8 = (int)(blk / nb32), cb = (int)(blk - (size_t)r8 * nb32);
    const int r = r8 * 8 + 4 * (lane >> 5) + t;  // row of W^T = column n of W
    const int c = cb * 32 + (lane & 31);         // col of W^T = row k of W
    packed[i] = w[(size_t)c * N + r];
  }
}

hipError_t pack_kn_T_launch(const float* w, int K, int N, float* packed, hipStream_t st) {
  const size_t total = (size_t)K * N;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pack_kn_T_kernel, dim3(blocks), dim3(256), 0, st, w, K, N, packed);
  return hipGetLastError();
}

// conv backward-data: dX = conv3x3_same(dZ, Wb), Wb[t'][co][ci] = W[8 - t'][ci][co]
// -> [9*Cout][Cin] matrix in disn_pack_kn order
__global__ __launch_bounds__(256) void pack_conv_bwd_kernel(const float* __restrict__ w, int Cin,
                                                            int Cout, float* __restrict__ packed) {
  const size_t total = (size_t)9 * Cout * Cin;
  const int nb32 = Cin >> 5;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(i & 3), lane = (int)((i >> 2) & 63);
    const size_t blk = i >> 8;
    const int r8 = (int)(blk / nb32), cb = (int)(blk - (size_t)r8 * nb32);
    const int r = r8 * 8 + 4 * (lane >> 5) + t;  // (t', co)
    const int ci = cb * 32 + (lane & 31);
    const int tp = r / Cout, co = r - tp * Cout;
    packed[i] = w[((size_t)(8 - tp) * Cin + ci) * Cout + co];
  }
}

hipError_t pack_conv_bwd_launch(const float* w, int Cin, int Cout, float* packed, hipStream_t st) {
  const size_t total = (size_t)9 * Cout * Cin;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pack_conv_bwd_kernel, dim3(blocks), dim3(256), 0, st, w, Cin, Cout, packed);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// column sums (bias gradients), optionally with the ReLU mask applied in place
// ---------------------------------------------------------------------------
constexpr int kRowsPerChunk = 256;

// block: 256 threads = CG float4 column groups x RL row lanes; chunk of 256 rows
// amax != nullptr: also max |masked dy| per image of rpi rows, spread over the image's 64 slots (atomic max of
// non-negative floats as unsigned: exact, order-free) -- the operand scales of the data gradient that follows
// (train.hip: saves the separate pass over dy).  A chunk of <= rpc rows touches at most three images (rpi >= rpc / 2).
__global__ __launch_bounds__(256) void colsum_partial_kernel(float* __restrict__ dy,
                                                             const float* __restrict__ y, long M, int N,
                                                             int relu, int rpc,
                                                             float* __restrict__ partial, float* __restrict__ amax,
                                                             long rpi) {
  __shared__ float4 red[256];
  const int cgn = N >> 2;                       // float4 column groups of the matrix
  const int cg_per_block = cgn < 256 ? cgn : 256;
  const int rl_n = 256 / cg_per_block;
  const int cg = threadIdx.x % cg_per_block, rl = threadIdx.x / cg_per_block;
  const int col = (blockIdx.x * cg_per_block + cg) * 4;
  const long r0 = (long)blockIdx.y * rpc;
  const long r1 = r0 + rpc < M ? r0 + rpc : M;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const long img0 = amax ? r0 / rpi : 0;
  const long b1 = (img0 + 1) * rpi, b2 = b1 + rpi;   // rows of image img0 end at b1, of img0 + 1 at b2
  float m0 = 0.f, m1 = 0.f, m2 = 0.f;
  if (col < N && rl < rl_n) {
    for (long r = r0 + rl; r < r1; r += rl_n) {
      float4 d = *reinterpret_cast<const float4*>(dy + (size_t)r * N + col);
      if (relu) {
        const float4 v = *reinterpret_cast<const float4*>(y + (size_t)r * N + col);
        d.x = v.x > 0.f ? d.x : 0.f; d.y = v.y > 0.f ? d.y : 0.f;
        d.z = v.z > 0.f ? d.z : 0.f; d.w = v.w > 0.f ? d.w : 0.f;
        *reinterpret_cast<float4*>(dy + (size_t)r * N + col) = d;
      }
      acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
      if (amax) {
        const float a = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)));
        if (r < b1) m0 = fmaxf(m0, a);
        else if (r < b2) m1 = fmaxf(m1, a);
        else m2 = fmaxf(m2, a);
      }
    }
  }
  if (amax) {   // wave maxima -> one atomic per wave and image touched
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      m0 = fmaxf(m0, __shfl_xor(m0, off));
      m1 = fmaxf(m1, __shfl_xor(m1, off));
      m2 = fmaxf(m2, __shfl_xor(m2, off));
    }
    if ((threadIdx.x & 63) == 0) {
      const int slot = (int)((blockIdx.y * 4 + (threadIdx.x >> 6) + blockIdx.x * 7) & 63);
      unsigned* a = reinterpret_cast<unsigned*>(amax);
      if (m0 > 0.f) atomicMax(a + img0 * 64 + slot, __float_as_uint(m0));
      if (m1 > 0.f) atomicMax(a + (img0 + 1) * 64 + slot, __float_as_uint(m1));
      if (m2 > 0.f) atomicMax(a + (img0 + 2) * 64 + slot, __float_as_uint(m2));
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (rl == 0 && col < N) {
    for (int k = 1; k < rl_n; ++k) {
      const float4 u = red[k * cg_per_block + cg];
      acc.x += u.x; acc.y += u.y; acc.z += u.z; acc.w += u.w;
    }
    *reinterpret_cast<float4*>(partial + (size_t)blockIdx.y * N + col) = acc;
  }
}

// out[n] = sum_chunks partial[c][n] + l2 * wcur[n]
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ partial, int chunks,
                                                            int N, float* __restrict__ out,
                                                            const float* __restrict__ wcur, float l2) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += partial[(size_t)c * N + n];
  if (l2 != 0.f) s += l2 * wcur[n];
  out[n] = s;
}

// the same sum with 8 lanes per column (lane j takes chunks j, j+8, ...; fixed-order LDS tree):
// for the long reductions (up to 512 chunks) of the convolution bias gradients
// blockIdx.y = group of `chunks` consecutive partial rows -> out[group][N]
__global__ __launch_bounds__(256) void colsum_finish8_kernel(const float* __restrict__ partial, int chunks,
                                                             int N, float* __restrict__ out) {
  __shared__ float red[32][9];
  const int c = threadIdx.x & 7, j = threadIdx.x >> 3;  // 8 columns x 32 lanes
  const int n = blockIdx.x * 8 + c;
  const float* pg = partial + (size_t)blockIdx.y * chunks * N;
  float s = 0.f;
  if (n < N)
    for (int k = j; k < chunks; k += 32) s += pg[(size_t)k * N + n];
  red[j][c] = s;
  __syncthreads();
  for (int w = 16; w >= 1; w >>= 1) {  // fixed tree over the 32 lanes
    if (j < w) red[j][c] += red[j + w][c];
    __syncthreads();
  }
  if (j == 0 && n < N) out[(size_t)blockIdx.y * N + n] = red[0][c];
}

// rows per chunk: about M/1024 (one workgroup per chunk and column block: ~1000 workgroups keep
// 256 CUs streaming), at least 32, a multiple of 16 (the row lanes of a workgroup)
static int colsum_rpc(long M) {
  long r = (M + 1023) / 1024;
  r = (r + 15) & ~15L;
  return (int)(r < 32 ? 32 : r);
}

size_t colsum_ws_bytes(long M, int N) {
  const int rpc = colsum_rpc(M);
  return (size_t)((M + rpc - 1) / rpc) * N * sizeof(float);
}

// amax (optional): [M / rows_per_image][64] slots, ZEROED by the caller, receive max |masked dy| per image; needs
// rows_per_image >= colsum_rpc(M) / 2 (a chunk then touches at most three images), else ignored -> returns through
// *amax_done (optional) whether the maxima were written
hipError_t relu_bwd_colsum_launch(float* dy, const float* y, long M, int N, int relu, float* db,
                                  float* ws, hipStream_t st, float* amax, long rows_per_image, bool* amax_done) {
  const int rpc = colsum_rpc(M);
  const int chunks = (int)((M + rpc - 1) / rpc);
  const int cgn = N / 4, cgb = cgn < 256 ? cgn : 256;
  const bool with_max = amax && rows_per_image > 0 && 2 * rows_per_image >= rpc;
  if (amax_done) *amax_done = with_max;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((cgn + cgb - 1) / cgb, chunks), dim3(256), 0, st, dy, y,
                     M, N, relu, rpc, ws, with_max ? amax : nullptr, with_max ? rows_per_image : 1L);
  hipLaunchKernelGGL(colsum_finish8_kernel, dim3((N + 7) / 8), dim3(256), 0, st, ws, chunks, N, db);
  return hipGetLastError();
}

hipError_t image_colsum_launch(const float* x, int B, long N, int C, float* out, float* ws,
                               hipStream_t st) {
  const int rpc = colsum_rpc((long)B * N);
  if (N % rpc == 0) {  // chunks never straddle two images: one pass + a grouped finish
    const int cpg = (int)(N / rpc), cgn = C / 4, cgb = cgn < 256 ? cgn : 256;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((cgn + cgb - 1) / cgb, B * cpg), dim3(256), 0, st,
                       const_cast<float*>(x), (const float*)nullptr, (long)B * N, C, 0, rpc, ws, (float*)nullptr, 1L);
    hipLaunchKernelGGL(colsum_finish8_kernel, dim3((C + 7) / 8, B), dim3(256), 0, st, ws, cpg, C, out);
    return hipGetLastError();
  }
  for (int b = 0; b < B; ++b) {
    hipError_t e = relu_bwd_colsum_launch(const_cast<float*>(x) + (size_t)b * N * C, nullptr, N, C, 0,
                                          out + (size_t)b * C, ws, st);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// ---------------------------------------------------------------------------
// loss gradient  (models/model_normalization.py:283-289)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ pred,
                                                        const float* __restrict__ gt, long M,
                                                        float sdf_weight, float mask_weight,
                                                        float* __restrict__ dpred) {
  const float scale = 1000.0f / (float)M;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long)gridDim.x * blockDim.x) {
    const float g = gt[i];
    const float w = g <= 0.01f ? mask_weight : 1.0f;
    const float e = g * sdf_weight - pred[i];
    const float s = e > 0.f ? -1.f : (e < 0.f ? 1.f : 0.f);  // d|e|/dpred
    dpred[i] = s * w * scale;
  }
}

hipError_t loss_grad_launch(const float* pred, const float* gt, long M, float sdf_weight,
                            float mask_weight, float* dpred, hipStream_t st) {
  int blocks = (int)((M + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(loss_grad_kernel, dim3(blocks), dim3(256), 0, st, pred, gt, M, sdf_weight,
                     mask_weight, dpred);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// fold2/conv5 (256 -> 1, linear) backward fused with the ReLU mask of fold2/conv2.
// workgroup = chunk of 32 rows: 64 float4 column groups x 4 row lanes
// ---------------------------------------------------------------------------
constexpr int kFinalRows = 32;

__global__ __launch_bounds__(256) void final_bwd_kernel(const float* __restrict__ h5,
                                                        const float* __restrict__ dpred, long M,
                                                        const float* __restrict__ w6,
                                                        float* __restrict__ dz5,
                                                        float* __restrict__ partial) {
  __shared__ float4 red[2][4][64];
  __shared__ float redd[4];
  const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * kFinalRows;
  const long r1 = r0 + kFinalRows < M ? r0 + kFinalRows : M;
  const float4 wk = *reinterpret_cast<const float4*>(w6 + cg * 4);
  float4 aw = make_float4(0.f, 0.f, 0.f, 0.f), ab = aw;
  float ad = 0.f;
  for (long r = r0 + rl; r < r1; r += 4) {
    const float d = dpred[r];
    const float4 h = *reinterpret_cast<const float4*>(h5 + (size_t)r * 256 + cg * 4);
    float4 dz;
    dz.x = h.x > 0.f ? d * wk.x : 0.f; dz.y = h.y > 0.f ? d * wk.y : 0.f;
    dz.z = h.z > 0.f ? d * wk.z : 0.f; dz.w = h.w > 0.f ? d * wk.w : 0.f;
    *reinterpret_cast<float4*>(dz5 + (size_t)r * 256 + cg * 4) = dz;
    aw.x += h.x * d; aw.y += h.y * d; aw.z += h.z * d; aw.w += h.w * d;
    ab.x += dz.x; ab.y += dz.y; ab.z += dz.z; ab.w += dz.w;
    ad += d;
  }
  red[0][rl][cg] = aw;
  red[1][rl][cg] = ab;
  if (cg == 0) redd[rl] = ad;
  __syncthreads();
  if (rl < 2) {  // rl selects dw6 / db5
    const float4 a0 = red[rl][0][cg], a1 = red[rl][1][cg], a2 = red[rl][2][cg], a3 = red[rl][3][cg];
    float4 o;
    o.x = (a0.x + a1.x) + (a2.x + a3.x); o.y = (a0.y + a1.y) + (a2.y + a3.y);
    o.z = (a0.z + a1.z) + (a2.z + a3.z); o.w = (a0.w + a1.w) + (a2.w + a3.w);
    *reinterpret_cast<float4*>(partial + (size_t)blockIdx.x * 768 + rl * 256 + cg * 4) = o;
  } else if (rl == 2) {  // db6 partial, replicated over the 256 columns (column 0 is used)
    const float o = (redd[0] + redd[1]) + (redd[2] + redd[3]);
    *reinterpret_cast<float4*>(partial + (size_t)blockIdx.x * 768 + 512 + cg * 4) = make_float4(o, o, o, o);
  }
}

__global__ __launch_bounds__(256) void final_bwd_emit_kernel(const float* __restrict__ sums,
                                                             const float* __restrict__ w6, float l2,
                                                             float* __restrict__ dw6,
                                                             float* __restrict__ db6,
                                                             float* __restrict__ db5) {
  const int k = threadIdx.x;
  dw6[k] = sums[k] + l2 * w6[k];
  db5[k] = sums[256 + k];
  if (k == 0) db6[0] = sums[512];
}

size_t final_bwd_ws_bytes(long M) {
  return (size_t)((M + kFinalRows - 1) / kFinalRows + 1) * 768 * sizeof(float);
}

hipError_t final_bwd_launch(const float* h5, const float* dpred, long M, const float* w6, float* dz5,
                            float* dw6, float* db6, float* db5, float l2, float* ws, hipStream_t st) {
  const int chunks = (int)((M + kFinalRows - 1) / kFinalRows);
  float* sums = ws + (size_t)chunks * 768;
  hipLaunchKernelGGL(final_bwd_kernel, dim3(chunks), dim3(256), 0, st, h5, dpred, M, w6, dz5, ws);
  hipLaunchKernelGGL(colsum_finish8_kernel, dim3(768 / 8, 1), dim3(256), 0, st, ws, chunks, 768, sums);
  hipLaunchKernelGGL(final_bwd_emit_kernel, dim3(1), dim3(256), 0, st, sums, w6, l2, dw6, db6, db5);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// fold1/conv1 (3 -> 64): dw1[c][n] = sum_m p[m][c] dz1[m][n]   (dz1 already ReLU-masked)
// thread = (n, row lane): 64 columns x 4 row lanes, chunk of 256 rows
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ pts,
                                                        const float* __restrict__ dz1, long M,
                                                        float* __restrict__ partial) {
  __shared__ float red[4][3][64];
  const int n = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * kRowsPerChunk;
  const long r1 = r0 + kRowsPerChunk < M ? r0 + kRowsPerChunk : M;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (long r = r0 + rl; r < r1; r += 4) {
    const float d = dz1[(size_t)r * 64 + n];
    a0 += pts[r * 3] * d; a1 += pts[r * 3 + 1] * d; a2 += pts[r * 3 + 2] * d;
  }
  red[rl][0][n] = a0; red[rl][1][n] = a1; red[rl][2][n] = a2;
  __syncthreads();
  if (rl < 3) {  // rl now indexes the component c
    const float s = (red[0][rl][n] + red[1][rl][n]) + (red[2][rl][n] + red[3][rl][n]);
    partial[(size_t)blockIdx.x * 192 + rl * 64 + n] = s;
  }
}

hipError_t embed_bwd_launch(const float* pts, const float* dz1, long M, float* dw1, const float* w1,
                            float l2, float* ws, hipStream_t st) {
  const int chunks = (int)((M + kRowsPerChunk - 1) / kRowsPerChunk);
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(chunks), dim3(256), 0, st, pts, dz1, M, ws);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3(1), dim3(256), 0, st, ws, chunks, 192, dw1, w1, l2);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// small-batch fully-connected backward (fc6/fc7/fc8 and the folded global block)
// ---------------------------------------------------------------------------
// c[k][n] = sum_b x[b][k] dy[b][n] + l2 w[k][n]; thread = (k, float4 of n); writes K*N floats once
__global__ __launch_bounds__(256) void outer_kernel(const float* __restrict__ x,
                                                    const float* __restrict__ dy, int B, int K, int N,
                                                    float* __restrict__ c, const float* __restrict__ wcur,
                                                    float l2) {
  const size_t n4 = (size_t)N >> 2, total = (size_t)K * n4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t k = i / n4;
    const int n = (int)(i - k * n4) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = 0; b < B; ++b) {
      const float xv = x[(size_t)b * K + k];
      const float4 d = *reinterpret_cast<const float4*>(dy + (size_t)b * N + n);
      acc.x += xv * d.x; acc.y += xv * d.y; acc.z += xv * d.z; acc.w += xv * d.w;
    }
    if (l2 != 0.f) {
      const float4 wv = nt_load4(wcur + k * N + n);
      acc.x += l2 * wv.x; acc.y += l2 * wv.y; acc.z += l2 * wv.z; acc.w += l2 * wv.w;
    }
    nt_store4(c + k * N + n, acc);
  }
}

hipError_t outer_launch(const float* x, const float* dy, int B, int K, int N, float* c,
                        const float* wcur, float l2, hipStream_t st) {
  const size_t total = (size_t)K * (N / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(outer_kernel, dim3(blocks), dim3(256), 0, st, x, dy, B, K, N, c, wcur, l2);
  return hipGetLastError();
}

// dx[b][k] = (sum_n W[k][n] dy[b][n]) * (xact[b][k] > 0); one wave per row k, B <= 8 per pass
template <int NB>
__global__ __launch_bounds__(256) void gemv_t_kernel(const float* __restrict__ w,
                                                     const float* __restrict__ dy, int Btot, int b0,
                                                     int K, int N, const float* __restrict__ xact,
                                                     float* __restrict__ dx) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int k = wave; k < K; k += nwaves) {
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.f;
    const float* wr = w + (size_t)k * N;
    for (int n = lane * 4; n < N; n += 256) {
      const float4 wv = nt_load4(wr + n);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float4 d = *reinterpret_cast<const float4*>(dy + (size_t)(b0 + b) * N + n);
        acc[b] += (wv.x * d.x + wv.y * d.y) + (wv.z * d.z + wv.w * d.w);
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float v = acc[b];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) {
        if (xact && !(xact[(size_t)(b0 + b) * K + k] > 0.f)) v = 0.f;
        dx[(size_t)(b0 + b) * K + k] = v;
      }
    }
  }
}

hipError_t gemv_t_launch(const float* w_kn, const float* dy, int B, int K, int N, const float* xact,
                         float* dx, hipStream_t st) {
  int blocks = (K + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  for (int b0 = 0; b0 < B; b0 += 8) {
    const int nb = (B - b0) < 8 ? (B - b0) : 8;
    switch (nb) {
#define DISN_GT_CASE(NB)                                                                         \
  case NB:                                                                                       \
    hipLaunchKernelGGL((gemv_t_kernel<NB>), dim3(blocks), dim3(256), 0, st, w_kn, dy, B, b0, K, N, \
                       xact, dx);                                                                \
    break;
      DISN_GT_CASE(1) DISN_GT_CASE(2) DISN_GT_CASE(3) DISN_GT_CASE(4)
      DISN_GT_CASE(5) DISN_GT_CASE(6) DISN_GT_CASE(7) DISN_GT_CASE(8)
#undef DISN_GT_CASE
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// Round 4: both halves of a small-batch fc backward in ONE pass over W (B <= 8):
//     dW[k][n] = sum_b x[b][k] dy[b][n] + l2 W[k][n]        (outer_kernel's expression)
//     dx[b][k] = (sum_n W[k][n] dy[b][n]) * (xact[b][k] > 0)   (gemv_t_kernel's expression and summation order)
// outer_kernel + gemv_t_kernel read W twice and re-load the eight dy rows from L1 / L2 for every output float4 (18 / 9
// memory instructions per 16 bytes of output: issue-bound, 0.9 / 1.6 TB/s on fc6's 411 MB).  Here dy [B][N] sits in LDS
// (128 KB at N = 4096), a wave owns a row k: its x values are wave-uniform (scalar loads), each lane streams the row's
// float4s once (nt), forms the dx partial dots and the dW float4 from the same registers and stores dW (nt): two memory
// instructions per float4 of W, one read + one write of W's bytes in all.
template <int NB>
__global__ __launch_bounds__(512) void fc_bwd_fused_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           int K, int N, const float* __restrict__ w, float l2,
                                                           float* __restrict__ dw, const float* __restrict__ xact,
                                                           float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) float dys[];   // [NB][N]
  for (int i = threadIdx.x * 4; i < NB * N; i += blockDim.x * 4)
    *reinterpret_cast<float4*>(&dys[i]) = *reinterpret_cast<const float4*>(dy + i);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int k = wave; k < K; k += nwaves) {
    float xv[NB], acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      xv[b] = x[(size_t)b * K + k];      // wave-uniform
      acc[b] = 0.f;
    }
    const float* wr = w + (size_t)k * N;
    float* dwr = dw + (size_t)k * N;
    for (int n0 = lane * 4; n0 < N; n0 += 1024) {   // four float4s of the row in flight per lane
      float4 wv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (n0 + 256 * u < N) wv[u] = nt_load4(wr + n0 + 256 * u);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int n = n0 + 256 * u;
        if (n >= N) break;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float4 d = *reinterpret_cast<const float4*>(&dys[b * N + n]);
          acc[b] += (wv[u].x * d.x + wv[u].y * d.y) + (wv[u].z * d.z + wv[u].w * d.w);
          o.x += xv[b] * d.x; o.y += xv[b] * d.y; o.z += xv[b] * d.z; o.w += xv[b] * d.w;
        }
        if (l2 != 0.f) { o.x += l2 * wv[u].x; o.y += l2 * wv[u].y; o.z += l2 * wv[u].z; o.w += l2 * wv[u].w; }
        nt_store4(dwr + n, o);
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float v = acc[b];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) {
        if (xact && !(xact[(size_t)b * K + k] > 0.f)) v = 0.f;
        dx[(size_t)b * K + k] = v;
      }
    }
  }
}

// dW and dx of one fc layer; B <= 8 and B * N * 4 <= 128 KB take the fused pass, otherwise outer + gemv_t as before
hipError_t fc_bwd_launch(const float* x, const float* dy, int B, int K, int N, const float* w, float l2, float* dw,
                         const float* xact, float* dx, hipStream_t st) {
  const size_t lds = (size_t)B * N * sizeof(float);
  if (B > 8 || lds > 128 * 1024 || N % 4) {
    hipError_t e = outer_launch(x, dy, B, K, N, dw, w, l2, st);
    if (e != hipSuccess) return e;
    return gemv_t_launch(w, dy, B, K, N, xact, dx, st);
  }
  int blocks = (K + 7) / 8;
  if (blocks > 512) blocks = 512;        // two workgroups of 8 waves per CU at N <= 2048, one at N = 4096 (LDS)
  switch (B) {
#define DISN_FB_CASE(NB)                                                                                       \
  case NB:                                                                                                     \
    hipLaunchKernelGGL((fc_bwd_fused_kernel<NB>), dim3(blocks), dim3(512), lds, st, x, dy, K, N, w, l2, dw, xact, dx); \
    break;
    DISN_FB_CASE(1) DISN_FB_CASE(2) DISN_FB_CASE(3) DISN_FB_CASE(4)
    DISN_FB_CASE(5) DISN_FB_CASE(6) DISN_FB_CASE(7) DISN_FB_CASE(8)
#undef DISN_FB_CASE
  }
  return hipGetLastError();
}

// out[i] = src[i] + l2 * w[i]
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ src,
                                                    const float* __restrict__ w, float l2, size_t n,
                                                    float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    out[i] = src[i] + l2 * w[i];
}

hipError_t axpby_launch(const float* src, const float* w, float l2, size_t n, float* out,
                        hipStream_t st) {
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, w, l2, n, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// losses  (models/model_normalization.py:273-299) -- one workgroup, fixed-order tree
// out[0] accuracy, [1] sdf_loss_realvalue, [2] sdf_loss, [3] regularization (left alone), [4] overall
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void loss_reduce_kernel(const float* __restrict__ pred,
                                                           const float* __restrict__ gt, long M,
                                                           float sdf_weight, float mask_weight,
                                                           float* __restrict__ out) {
  __shared__ float red[3][1024];
  float a = 0.f, r = 0.f, s = 0.f;
  for (long i = threadIdx.x; i < M; i += 1024) {
    const float g = gt[i], p = pred[i];
    a += ((g > 0.f) == (p > 0.f)) ? 1.f : 0.f;
    r += fabsf(g - p / sdf_weight);
    s += fabsf(g * sdf_weight - p) * (g <= 0.01f ? mask_weight : 1.0f);
  }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = r; red[2][threadIdx.x] = s;
  __syncthreads();
  for (int w = 512; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) {
      red[0][threadIdx.x] += red[0][threadIdx.x + w];
      red[1][threadIdx.x] += red[1][threadIdx.x + w];
      red[2][threadIdx.x] += red[2][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = red[0][0] / (float)M;
    out[1] = red[1][0] / (float)M;
    out[2] = red[2][0] / (float)M * 1000.0f;
    out[4] = out[2] + out[3];
  }
}

hipError_t loss_reduce_launch(const float* pred, const float* gt, long M, float sdf_weight,
                              float mask_weight, float* out5, hipStream_t st) {
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(1024), 0, st, pred, gt, M, sdf_weight,
                     mask_weight, out5);
  return hipGetLastError();
}

// regularization = wd/2 * sum over the weight segments of sum(w^2): 256 partial sums per segment
// (segments start on 64-float boundaries and hold a multiple of 4 floats: float4 loads)
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ params,
                                                            const SumsqSegs segs,
                                                            float* __restrict__ partial) {
  __shared__ float red[256];
  const int seg = blockIdx.y;
  const float4* p = reinterpret_cast<const float4*>(params + segs.off[seg]);
  const long n4 = segs.cnt[seg] >> 2;
  // (round 4: eight loads in flight per thread -- with one, the 65 536 threads of a segment waited out a memory round
  // trip per 16 bytes: 0.36 ms for the 563 MB of weights)
  float a = 0.f, b = 0.f;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 7 * 65536L < n4; i += 8 * 65536L) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = nt_load4(reinterpret_cast<const float*>(p + i + u * 65536L));
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a += v[u].x * v[u].x + v[u].y * v[u].y;
      b += v[u].z * v[u].z + v[u].w * v[u].w;
    }
  }
  for (; i < n4; i += 65536L) {
    const float4 v = nt_load4(reinterpret_cast<const float*>(p + i));
    a += v.x * v.x + v.y * v.y;
    b += v.z * v.z + v.w * v.w;
  }
  red[threadIdx.x] = a + b;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[seg * 256 + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void sumsq_finish_kernel(const float* __restrict__ partial, int nseg,
                                                           float half_wd, float* __restrict__ out) {
  __shared__ float red[256];
  float a = 0.f;
  for (int s = 0; s < nseg; ++s) a += partial[s * 256 + threadIdx.x];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = half_wd * red[0];
}

hipError_t sumsq_launch(const float* params, const SumsqSegs& segs, float half_wd, float* out, float* ws,
                        hipStream_t st) {
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(256, segs.n), dim3(256), 0, st, params, segs, ws);
  hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, st, ws, segs.n, half_wd, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// tf.train.AdamOptimizer update (train/train_sdf.py:251):  g' = g * gscale;
// m = b1 m + (1-b1) g'; v = b2 v + (1-b2) g'^2; w -= lr_t * m / (sqrt(v) + eps)
// lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) is computed by the host (double) and passed in
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   size_t n4, float lr_t, float b1, float b2, float eps,
                                                   float gscale) {
  const float c1 = 1.0f - b1, c2 = 1.0f - b2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (size_t)gridDim.x * blockDim.x) {
    float4 wv = nt_load4(w + 4 * i);
    const float4 gv = nt_load4(g + 4 * i);
    float4 mv = nt_load4(m + 4 * i), vv = nt_load4(v + 4 * i);
#define DISN_ADAM(f)                                   \
  {                                                    \
    const float gg = gv.f * gscale;                    \
    mv.f = b1 * mv.f + c1 * gg;                        \
    vv.f = b2 * vv.f + c2 * (gg * gg);                 \
    wv.f = wv.f - lr_t * mv.f / (sqrtf(vv.f) + eps);   \
  }
    DISN_ADAM(x) DISN_ADAM(y) DISN_ADAM(z) DISN_ADAM(w)
#undef DISN_ADAM
    nt_store4(w + 4 * i, wv);
    nt_store4(m + 4 * i, mv);
    nt_store4(v + 4 * i, vv);
  }
}

hipError_t adam_launch(float* w, const float* g, float* m, float* v, size_t n, float lr_t, float b1,
                       float b2, float eps, float gscale, hipStream_t st) {
  const size_t n4 = n / 4;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w, g, m, v, n4, lr_t, b1, b2,
                     eps, gscale);
  return hipGetLastError();
}

}  // namespace disn
