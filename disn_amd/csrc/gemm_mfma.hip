// fp32 MFMA GEMM for gfx950: one kernel template serves
//   * the VGG-16 3x3 SAME convolutions as an implicit GEMM over NHWC input
//     (rows = pixels, cols = Cout, K = (ky,kx,ci))      -- models/CNN/vgg.py:187-196
//   * the point-MLP 1x1 convolutions (rows = points)    -- models/sdfnet.py:71-88,173-186
// with bias + ReLU fused in the epilogue and optional split-K.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate, exact f32 fma
// chain) because the path's tolerance is 1e-5 absolute; bf16/fp16 MFMA cannot
// meet it.  Peak for this instruction is 157.3 TFLOP/s.
//
// Tiling: 256 threads = 4 waves in a 2x2 grid; block tile BM x BN in
// {64,128}^2, each wave owns (BM/2)x(BN/2) = TM x TN MFMA tiles of 32x32.
// K advances in steps of 32.  A (activations) is staged through LDS in full
// 128-byte rows (8 lanes x 16 B per row), double buffered, rows padded to 36
// floats so that the ds_read_b128 fragment reads are bank-conflict free
// (36*row mod 64 is a bijection over 16 consecutive rows).  B (weights) is
// pre-packed in MFMA fragment order (disn_pack_kn) so each lane fetches its
// four k-values of a 32-column block with one coalesced 16-byte load straight
// from L2 -- 1 KiB per wave instruction, no LDS round trip.
// k-permutation: MFMA k-index h of step t inside an 8-k block is k = 4h + t on
// both operands, so one float4 per operand feeds four MFMAs.
#include "kernels.hpp"

#include <cstdlib>

namespace disn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// zero a loaded vector with a bit mask (not a select: a select lets the compiler sink the
// load into a branch, which splits the K loop into basic blocks and makes its s_waitcnt
// bookkeeping pessimistic)
__device__ __forceinline__ float4 mask4(float4 v, bool keep) {
  const unsigned m = keep ? 0xffffffffu : 0u;
  return make_float4(__uint_as_float(__float_as_uint(v.x) & m), __uint_as_float(__float_as_uint(v.y) & m),
                     __uint_as_float(__float_as_uint(v.z) & m), __uint_as_float(__float_as_uint(v.w) & m));
}

struct GemmDev {
  GemmParams p;
  float* ws;
  int ksteps, mtiles, ntiles;
};

template <int BM, int BN, int MODE>
__global__ __launch_bounds__(256) void gemm_f32_mfma(const GemmDev d) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int LDA = 36;
  constexpr int APASS = BM / 32;
  __shared__ __attribute__((aligned(16))) float lds[2 * BM * LDA];

  const GemmParams& p = d.p;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: block b runs on XCD b%8; give each XCD a contiguous run of
  // tile ids (neighbouring tiles share A rows / B columns in that XCD's L2).
  int bid = blockIdx.x;
  {
    const int nwg = d.mtiles * d.ntiles;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mt = bid / d.ntiles, nt = bid - mt * d.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int S = gridDim.y, z = blockIdx.y;
  const int s_begin = (int)(((long)d.ksteps * z) / S);
  const int s_end = (int)(((long)d.ksteps * (z + 1)) / S);

  // ---- per-thread A-loader state: rows (tid>>3)+32*pass, 16-byte column c4 -------
  const int c4 = tid & 7;
  const int arow = tid >> 3;
  int pix[APASS];  // CONV: flat pixel index (b*H+y)*W+x, -1 when the row is past M
  int yx[APASS];   // CONV: (y<<16)|x ; DENSE: unused
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int m = m0 + arow + 32 * i;
    if (MODE == GEMM_DENSE) {
      pix[i] = (m < p.M) ? m : -1;
      yx[i] = 0;
    } else {
      if (m < p.M) {
        const int hw = p.H * p.W;
        const int b = m / hw, rem = m - b * hw;
        const int y = rem / p.W, x = rem - y * p.W;
        pix[i] = m;
        yx[i] = (y << 16) | x;
      } else {
        pix[i] = -1;
        yx[i] = 0;
      }
    }
  }

  float4 areg[APASS];
  unsigned aok = 0;  // bit i: pass i of the in-flight A tile is inside the image / matrix
  auto load_a = [&](int s) {
    if (MODE == GEMM_DENSE) {
      const int k0 = s * 32;
      const float* base;
      int ld;
      if (k0 < p.k1) {
        base = p.a1 + k0;
        ld = p.lda1;
      } else {
        base = p.a2 + (k0 - p.k1);
        ld = p.lda2;
      }
#pragma unroll
      for (int i = 0; i < APASS; ++i) {
        const bool ok = pix[i] >= 0;
        areg[i] = *reinterpret_cast<const float4*>(base + (size_t)(ok ? pix[i] : 0) * ld + c4 * 4);
        aok = ok ? (aok | (1u << i)) : (aok & ~(1u << i));
      }
    } else if (MODE == GEMM_CONV3) {
      const int cblocks = p.Cin >> 5;
      const int kyx = s / cblocks, ci0 = (s - kyx * cblocks) << 5;
      const int dy = kyx / 3 - 1, dx = kyx - (kyx / 3) * 3 - 1;
#pragma unroll
      for (int i = 0; i < APASS; ++i) {
        const int yy = (yx[i] >> 16) + dy, xx = (yx[i] & 0xffff) + dx;
        const bool ok = pix[i] >= 0 && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
        // branch-free: out-of-image taps read pixel 0 and are zeroed by the select, so the
        // K loop stays one basic block and the compiler's vmcnt bookkeeping stays exact
        areg[i] = *reinterpret_cast<const float4*>(
            p.a1 + (size_t)(ok ? pix[i] + dy * p.W + dx : 0) * p.Cin + ci0 + c4 * 4);
        aok = ok ? (aok | (1u << i)) : (aok & ~(1u << i));
      }
    } else {  // GEMM_CONV3_C3: Cin == 3, K = 27 padded to 32, a single k-step
#pragma unroll
      for (int i = 0; i < APASS; ++i) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = c4 * 4 + j;
          const int kyx = k / 3, ci = k - kyx * 3;
          const int dy = kyx / 3 - 1, dx = kyx - (kyx / 3) * 3 - 1;
          const int yy = (yx[i] >> 16) + dy, xx = (yx[i] & 0xffff) + dx;
          const bool ok = k < 27 && pix[i] >= 0 && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
          v[j] = ok ? p.a1[(size_t)(pix[i] + dy * p.W + dx) * 3 + ci] : 0.f;
        }
        areg[i] = make_float4(v[0], v[1], v[2], v[3]);
        aok |= 1u << i;
      }
    }
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < APASS; ++i)
      *reinterpret_cast<float4*>(&lds[buf * BM * LDA + (arow + 32 * i) * LDA + c4 * 4]) =
          mask4(areg[i], (aok >> i) & 1u);  // the zeroing is applied here, off the load's path
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (s_begin < s_end) {
    load_a(s_begin);
    store_a(0);
  }
  __syncthreads();

  const int nb32 = p.N >> 5;
  const int frag_row = wm * (BM / 2) + (lane & 31);
  const int frag_k = (lane >> 5) * 4;
  const int colblk0 = (n0 >> 5) + wn * TN;
  // B fragments are software-pipelined one 8-k block ahead (carried across the step
  // boundary) so their L2 latency hides under the 16*TM*TN/4 MFMAs of the current block.
  auto load_b = [&](int s, int kb, float4 (&b)[TN]) {
    const float* bp = p.bp + (((size_t)s * 4 + kb) * nb32 + colblk0) * 256 + lane * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(bp + (size_t)j * 256);
  };
  float4 bcur[TN], bnext[TN];
  if (s_begin < s_end) load_b(s_begin, 0, bcur);
  int cur = 0;
  for (int s = s_begin; s < s_end; ++s) {
    // the last iteration re-fetches its own tile / B block (clamped index) instead of
    // branching: harmless, and the loop body stays branch-free
    const int sn = (s + 1 < s_end) ? s + 1 : s;
    const float* la = &lds[cur * BM * LDA + frag_row * LDA + frag_k];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      if (kb < 3) load_b(s, kb + 1, bnext);
      else load_b(sn, 0, bnext);
      float4 a[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = *reinterpret_cast<const float4*>(la + i * 32 * LDA + kb * 8);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, bcur[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, bcur[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, bcur[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, bcur[j].w, acc[i][j], 0, 0, 0);
        }
#pragma unroll
      for (int j = 0; j < TN; ++j) bcur[j] = bnext[j];
      if (kb == 0) {
        // next A tile: issued behind the first MFMA block so that block's operands are not
        // queued behind it; it lands under the remaining three blocks
        __builtin_amdgcn_sched_barrier(0);
        load_a(sn);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    store_a(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31,
  //      row = (r&3) + 8*(r>>2) + 4*(lane>>5)  (cdna_hip_programming.md §3) ----------
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
    const float bv = (S == 1) ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.M) {
          float v = acc[i][j][r];
          if (S == 1) {
            v += bv;
            if (p.relu) v = fmaxf(v, 0.f);
            p.out[(size_t)row * p.ldc + col] = v;
          } else {
            d.ws[((size_t)z * p.M + row) * p.N + col] = v;
          }
        }
      }
    }
  }
}

// out[m][n] = act(sum_s ws[s][m][n] + bias[row(m)][n]);  float4 per thread
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int S,
                                                            int M, int N,
                                                            const float* __restrict__ bias,
                                                            int rows_per_bias, int relu,
                                                            float* __restrict__ out, int ldc) {
  const size_t n4 = (size_t)N >> 2;
  const size_t total = (size_t)M * n4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i / n4;
    const int c = (int)(i - m * n4) * 4;
    float4 v = *reinterpret_cast<const float4*>(ws + m * N + c);
    for (int s = 1; s < S; ++s) {
      const float4 u = *reinterpret_cast<const float4*>(ws + ((size_t)s * M + m) * N + c);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const size_t brow = rows_per_bias ? m / rows_per_bias : 0;
    const float4 bv = *reinterpret_cast<const float4*>(bias + brow * N + c);
    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    if (relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    *reinterpret_cast<float4*>(out + m * ldc + c) = v;
  }
}

// packed[((k/8)*(N/32) + n/32)*256 + lane*4 + t] = W[8*(k/8) + 4*(lane>>5) + t][32*(n/32)+(lane&31)]
__global__ __launch_bounds__(256) void pack_kn_kernel(const float* __restrict__ w, int K, int N,
                                                      int Kpad, float* __restrict__ packed) {
  const size_t total = (size_t)Kpad * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(i & 3), lane = (int)((i >> 2) & 63);
    const size_t blk = i >> 8;
    const int nb32 = N >> 5;
    const int k8 = (int)(blk / nb32), nb = (int)(blk - (size_t)k8 * nb32);
    const int k = k8 * 8 + 4 * (lane >> 5) + t;
    const int n = nb * 32 + (lane & 31);
    packed[i] = (k < K) ? w[(size_t)k * N + n] : 0.f;
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static bool parse_force(int* bm, int* bn, int* s) {
  const char* e = std::getenv("DISN_GEMM_FORCE");  // "BM,BN,S" -- tuning/debug only
  if (!e) return false;
  return std::sscanf(e, "%d,%d,%d", bm, bn, s) == 3;
}

GemmPlan gemm_plan(int M, int N, int K, size_t max_ws) {
  const int ksteps = K / 32;
  const int cand[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
  const int svals[] = {1, 2, 3, 4, 6, 8, 9, 12, 16, 18, 24, 36};
  double best = 1e300;
  GemmPlan plan{64, 64, 1, 0};
  for (auto& c : cand) {
    const int bm = c[0], bn = c[1];
    if (N % bn) continue;
    const long wgs = (long)((M + bm - 1) / bm) * (N / bn);
    const int tiles = (bm / 64) * (bn / 64);
    for (int s : svals) {
      if (s > 1 && (ksteps % s || ksteps / s < 4)) continue;
      // split-K needs S partial slabs: never plan beyond the caller's workspace
      if (s > 1 && (size_t)s * M * N * sizeof(float) > max_ws) continue;
      // cost model (cycles): CU rounds x per-WG time (+ split-K reduce pass).
      // Co-residency: small tiles fit 3 WGs per CU, 128x128 fits 2.
      const int per_cu = tiles >= 4 ? 2 : 3;
      const double rounds = (double)((wgs * s + 256L * per_cu - 1) / (256L * per_cu));
      const double step = tiles * 16.0 * 64.0 + 500.0;  // MFMA issue + per-step overhead
      const double wg = (ksteps / s) * step * per_cu + 4000.0;
      double cost = rounds * wg;
      if (s > 1) cost += 6000.0 + (double)s * M * N * 8.0 / 2000.0;  // ~4.8 TB/s at 2.4 GHz
      if (cost < best) {
        best = cost;
        plan.bm = bm; plan.bn = bn; plan.splitk = s;
      }
    }
  }
  int fbm, fbn, fs;
  if (parse_force(&fbm, &fbn, &fs) && (fbm == 64 || fbm == 128) && (fbn == 64 || fbn == 128) &&
      N % fbn == 0 && fs >= 1 && fs <= ksteps &&
      (fs == 1 || (size_t)fs * M * N * sizeof(float) <= max_ws)) {
    plan.bm = fbm; plan.bn = fbn; plan.splitk = fs;
  }
  plan.ws_bytes = plan.splitk > 1 ? (size_t)plan.splitk * M * N * sizeof(float) : 0;
  return plan;
}

template <int BM, int BN>
static hipError_t launch_mode(const GemmDev& d, GemmMode mode, dim3 grid, hipStream_t st) {
  switch (mode) {
    case GEMM_DENSE:
      hipLaunchKernelGGL((gemm_f32_mfma<BM, BN, GEMM_DENSE>), grid, dim3(256), 0, st, d);
      break;
    case GEMM_CONV3:
      hipLaunchKernelGGL((gemm_f32_mfma<BM, BN, GEMM_CONV3>), grid, dim3(256), 0, st, d);
      break;
    default:
      hipLaunchKernelGGL((gemm_f32_mfma<BM, BN, GEMM_CONV3_C3>), grid, dim3(256), 0, st, d);
      break;
  }
  return hipGetLastError();
}

hipError_t gemm_launch(const GemmParams& p, GemmMode mode, const GemmPlan& plan, float* ws,
                       hipStream_t st) {
  GemmDev d;
  d.p = p;
  d.ws = ws;
  d.ksteps = p.K / 32;
  d.mtiles = (p.M + plan.bm - 1) / plan.bm;
  d.ntiles = p.N / plan.bn;
  dim3 grid(d.mtiles * d.ntiles, plan.splitk);
  hipError_t e;
  if (plan.bm == 128 && plan.bn == 128) e = launch_mode<128, 128>(d, mode, grid, st);
  else if (plan.bm == 128) e = launch_mode<128, 64>(d, mode, grid, st);
  else if (plan.bn == 128) e = launch_mode<64, 128>(d, mode, grid, st);
  else e = launch_mode<64, 64>(d, mode, grid, st);
  if (e != hipSuccess) return e;
  if (plan.splitk > 1)
    return splitk_reduce_launch(ws, plan.splitk, p.M, p.N, p.bias, p.rows_per_bias, p.relu, p.out,
                                p.ldc, st);
  return hipSuccess;
}

hipError_t splitk_reduce_launch(const float* ws, int S, int M, int N, const float* bias,
                                int rows_per_bias, int relu, float* out, int ldc, hipStream_t st) {
  const size_t total = (size_t)M * (N / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, ws, S, M, N, bias,
                     rows_per_bias, relu, out, ldc);
  return hipGetLastError();
}

hipError_t pack_kn_launch(const float* w, int K, int N, int Kpad, float* packed, hipStream_t st) {
  const size_t total = (size_t)Kpad * N;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pack_kn_kernel, dim3(blocks), dim3(256), 0, st, w, K, N, Kpad, packed);
  return hipGetLastError();
}

}  // namespace disn
