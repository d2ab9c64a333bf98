// GEMM on the bf16 matrix pipes, two uses (template parameter NS):
//   NS = 3  the INFERENCE path's fp32-accurate kernel: every fp32 operand is split into three bf16 terms
//           and the product is accumulated from the six largest cross terms (see below) -- 12 of the 13
//           VGG convolutions (models/CNN/vgg.py:187-196) and the deep / large-batch point-MLP layers;
//   NS = 1  plain bf16 multiply for the mixed-precision TRAINING step (BASELINE config 5 names bf16):
//           operands rounded to nearest-even when they are staged, "bf16 compute, fp32 master".
// Same two shapes as gemm_mfma.hip -- the 3x3 SAME convolution as an implicit GEMM over NHWC input and
// the point-MLP 1x1 convolutions -- with fp32 activations in HBM, fp32 accumulation and fp32 output.
//
//   v_mfma_f32_32x32x16_bf16: 16x the rate of the f32-input MFMA (2.5 PFLOP/s dense peak); per lane
//   8 consecutive k of one row (A) / one column (B): A-fragment lane (i = l&31, g = l>>5) holds
//   A[i][8g..8g+7], B-fragment lane holds B[8g..8g+7][l&31]; C/D layout as the f32 form.
//
// Tiling as gemm_mfma.hip: 256 threads = 4 waves 2x2, block tile BM x BN, K step 32 (two MFMAs per
// 32x32 accumulator).  A: coalesced float4 loads (8 lanes per 128-byte row), converted with
// v_cvt_pk_bf16_f32 and staged in LDS as bf16 rows of 40 (80 bytes: 16 consecutive rows hit 16
// distinct 4-bank groups, the ds_read_b128 fragment reads are conflict free), double buffered with
// register prefetch.  B: weights pre-packed in bf16 fragment order (pack_bf16_launch), one 16-byte
// load per lane per MFMA straight from L2.  One workgroup per tile (no stream-K: at the training
// batch every layer has >= 200 tiles and the kernel is L2-bandwidth bound, not MFMA bound:
// 128x128x32 moves 32 KB per 256 MFMA cycles).
#include "kernels.hpp"
#include "tuning.hpp"

namespace disn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------
// weight packing: an [R][C] matrix (R = reduction, padded to Rpad % 32 == 0; C % 32 == 0) taken from
// a TF-layout tensor in one of three views, to bf16 B-fragment order:
//   packed[((r/16)*(C/32) + c/32)*512 + lane*8 + t] = M[16*(r/16) + 8*(lane>>5) + t][32*(c/32) + (lane&31)]
// view 0: M = W [K][N]                         (forward)
// view 1: M = W^T, W [K][N] -> M [N][K]        (dA = dZ W^T)
// view 2: M[(t',co)][ci] = W[8-t'][ci][co]     (conv backward-data, W [3,3,Cin,Cout])
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ w, int view, int K,
                                                        int N, int R, int Rpad, int C, int nsplit,
                                                        __bf16* __restrict__ packed) {
  const size_t total = (size_t)Rpad * C;
  const int cb32 = C >> 5;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const size_t blk = i >> 9;
    const int rb = (int)(blk / cb32), cb = (int)(blk - (size_t)rb * cb32);
    const int r = rb * 16 + 8 * (lane >> 5) + t;
    const int c = cb * 32 + (lane & 31);
    float v = 0.f;
    if (r < R) {
      if (view == 0) {
        v = w[(size_t)r * N + c];
      } else if (view == 1) {
        v = w[(size_t)c * N + r];
      } else {  // K = Cin, N = Cout here
        const int tp = r / N, co = r - tp * N;
        v = w[((size_t)(8 - tp) * K + c) * N + co];
      }
    }
    for (int pl = 0; pl < nsplit; ++pl) {  // nsplit = 3: planes h, m, l with h + m + l == v exactly
      const __bf16 h = (__bf16)v;
      packed[(size_t)pl * total + i] = h;
      v -= (float)h;
    }
  }
}

hipError_t pack_bf16_launch(const float* w, int view, int K, int N, void* packed, hipStream_t st,
                            int nsplit) {
  int R, C;
  if (view == 0) { R = K; C = N; }
  else if (view == 1) { R = N; C = K; }
  else { R = 9 * N; C = K; }
  const int Rpad = (R + 31) & ~31;
  const size_t total = (size_t)Rpad * C;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pack_bf16_kernel, dim3(blocks), dim3(256), 0, st, w, view, K, N, R, Rpad, C,
                     nsplit == 3 ? 3 : 1, reinterpret_cast<__bf16*>(packed));
  return hipGetLastError();
}

// All the weight re-packs of one training step in ONE launch (41 of them, 5-7 us each as separate
// launches): a job list in the kernel arguments, one grid-stride loop over the concatenated element
// space.  T = 8: bf16 fragment order (above); T = 4: the fp32 order of disn_pack_kn.
// A "slot" = one lane of one fragment block = T consecutive output elements (16 bytes); slots of all
// jobs are concatenated (begin counts slots); a workgroup covers 1024 consecutive slots (4 per
// thread), finds its first job once, and a thread only steps forward over job borders.
__global__ __launch_bounds__(256) void pack_multi_kernel(const PackJobs jobs) {
  __shared__ int s_job;
  const long base = (long)blockIdx.x * 1024;
  if (threadIdx.x == 0) {
    int lo = 0, hi = jobs.n - 1;  // last job with begin <= base
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs.j[mid].begin <= base) lo = mid; else hi = mid - 1;
    }
    s_job = lo;
  }
  __syncthreads();
  int ji = s_job;
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    const long g = base + k * 256 + threadIdx.x;
    if (g >= jobs.total) return;
    while (ji + 1 < jobs.n && jobs.j[ji + 1].begin <= g) ++ji;
    const PackJob& J = jobs.j[ji];
    const long slot = g - J.begin;
    const int T = J.T, lane = (int)(slot & 63);
    const long blk = slot >> 6;
    const int cb32 = J.C >> 5;
    const int rb = (int)(blk / cb32), cb = (int)(blk - (long)rb * cb32);
    const int r0 = rb * 2 * T + T * (lane >> 5);
    const int c = cb * 32 + (lane & 31);
    float v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int r = r0 + t;
      v[t] = 0.f;
      if (t < T && r < J.R) {
        if (J.view == 0) {
          v[t] = J.src[(size_t)r * J.N + c];
        } else if (J.view == 1) {
          v[t] = J.src[(size_t)c * J.N + r];
        } else {
          const int tp = r / J.N, co = r - tp * J.N;
          v[t] = J.src[((size_t)(8 - tp) * J.K + c) * J.N + co];
        }
      }
    }
    if (T == 8) {
      for (int pl = 0; pl < J.ns; ++pl) {  // ns = 3: planes h, m, l with h + m + l == v
        bf16x8 o;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          o[t] = (__bf16)v[t];
          v[t] -= (float)o[t];
        }
        reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(J.dst) + (size_t)pl * J.plane)[slot] = o;
      }
    } else {
      reinterpret_cast<float4*>(J.dst)[slot] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

void pack_job_add(PackJobs& jobs, const float* src, void* dst, int view, int K, int N, int ns) {
  PackJob& J = jobs.j[jobs.n++];
  J.src = src; J.dst = dst; J.view = view; J.K = K; J.N = N; J.T = ns ? 8 : 4; J.ns = ns ? ns : 1;
  if (view == 0) { J.R = K; J.C = N; }
  else if (view == 1) { J.R = N; J.C = K; }
  else { J.R = 9 * N; J.C = K; }
  J.plane = (long)((J.R + 31) & ~31) * J.C;
  J.begin = jobs.total;  // in slots of T elements
  jobs.total += J.plane / J.T;
}

hipError_t pack_multi_launch(const PackJobs& jobs, hipStream_t st) {
  if (jobs.n == 0) return hipSuccess;
  const long blocks = (jobs.total + 1023) / 1024;
  hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, jobs);
  return hipGetLastError();
}

// Ablation mask of tools/ablate_x3.py (WRONG results, timing only; honoured by tuning builds alone):
// 1: no residual chain in the A split, 2: B fragments loaded once, 4: A tile loaded once,
// 8: one MFMA instead of six, 16: no LDS staging of A after the first tile
#if !defined(DISN_TUNING) || !defined(DISN_ABL)
#undef DISN_ABL
#define DISN_ABL 0
#endif

// ---------------------------------------------------------------------------
struct BfDev {
  GemmParams p;  // p.bp unused; p.K = padded reduction length (multiple of 32)
  const __bf16* bpk;
  int mtiles, ntiles;
  int S;      // split-K factor = gridDim.y; S > 1: raw partials to ws [S][M][N], splitk_reduce finishes
  float* ws;
  int nsplit;  // 1: bf16 product; 3: fp32-accurate product from three bf16 terms per operand
  float* pool_out;  // CONV3 with S > 1: the split-K reduce also writes the 2x2 max pool here
  int nmajor;       // workgroup order n-tile major (weight-stationary per XCD) instead of m-tile major
};

// NS = 1: plain bf16 multiply.  NS = 3: fp32-accurate product on the bf16 pipes ("3xBF16"): every
// fp32 operand is split into three bf16 terms x = h + m + l (8 + 8 + 8 mantissa bits: exact), and
// a*b is accumulated from the six largest cross terms hh, hm, mh, hl, lh, mm (the dropped ones are
// below 2^-24 relative, the size of an fp32 rounding): 6 MFMAs of 32 cycles instead of 8 f32-input
// MFMAs of 64 cycles for the same 32x32x16 block.
template <int BM, int BN, int MODE, int NS, int PF>
__global__ __launch_bounds__(256, (BM * BN >= 128 * 128) ? 2 : 3) void gemm_bf16_mfma(const BfDev d) {
  constexpr int TM = BM / 64, TN = BN / 64;
  // A planes in LDS: rows of 32 bf16 padded to 40 (80 B).  ds_read_b128 (MI355X_MICROARCH.md, LDS: four
  // non-contiguous 16-lane groups, bank = (a/4) mod 64): the 16 rows of a group land on 16 different
  // 16-byte slots ((5 row + c) mod 16) -> conflict-free.  ds_write_b64 (contiguous 16-lane groups, bank =
  // (a/4) mod 32): the two rows a group writes must be 4 apart (80 dwords = 16 mod 32), not adjacent
  // (20 dwords: four banks shared, every write 2 cycles, SQ_LDS_BANK_CONFLICT 33 % of the LDS cycles) --
  // hence the row order of the A loader below.
  constexpr int LDA = 40;
  constexpr int APASS = BM / 32;
  constexpr int PLANE = BM * LDA;
  // (the 128-row plain-bf16 variant keeps enough LDS for the staged epilogue: 4 waves x 32x36 floats)
  constexpr int LDS_ELEMS = (BM >= 128 && 2 * NS * PLANE < 4 * 32 * 36 * 2) ? 4 * 32 * 36 * 2 : 2 * NS * PLANE;
  __shared__ __attribute__((aligned(16))) __bf16 lds[LDS_ELEMS];
  const GemmParams& p = d.p;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware placement (hardware workgroup L runs on XCD L % 8; each XCD has its own 4 MB L2).
  int mt, nt, ks;
  if (d.nmajor) {
    // weight-stationary: the operand worth keeping in one L2 is B (the 14x14 / 28x28 convolutions: 14 MB of
    // three-plane weights against <= 1.6 MB of input).  Every XCD gets a CONTIGUOUS eighth of the order
    // (n-tile, k-split, m-tile): all m-tiles of one (n-tile, k-split) -- the workgroups that read the same
    // weight slice -- sit on one XCD, and with N/BN == 8 an XCD reads one n-tile's columns only: B crosses
    // the fabric once instead of eight times (measured r01: 117 MB fetched per conv4_x layer for 14.2 MB).
    const int T = gridDim.x * gridDim.y, L = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = T >> 3, r = T & 7, xcd = L & 7, idx = L >> 3;
    const int l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int per = d.S * d.mtiles;
    nt = l / per;
    const int rem = l - nt * per;
    ks = rem / d.mtiles;
    mt = rem - ks * d.mtiles;
  } else {  // consecutive tiles (sharing A rows) on the same XCD
    int w = blockIdx.x;
    const int W = gridDim.x, q = W >> 3, r = W & 7, xcd = w & 7, idx = w >> 3;
    w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    mt = w / d.ntiles;
    nt = w - mt * d.ntiles;
    ks = blockIdx.y;
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const int KS_all = p.K >> 5;
  const int s0 = (int)(((long)KS_all * ks) / d.S), s1 = (int)(((long)KS_all * (ks + 1)) / d.S);

  // ---- A loader: one row per 8-lane octet (+ 32 rows per pass), float4 column (tid&7) -------------
  // each 8-lane octet takes one row; the octets of a wave take rows 0,4,1,5,2,6,3,7 of its 8-row band, so
  // that the two rows of a 16-lane ds_write_b64 group are 4 apart (see LDA above)
  const int oct = (tid >> 3) & 7;
  const int arow = (tid >> 6) * 8 + (oct >> 1) + 4 * (oct & 1), c4 = (tid & 7) * 4;
  const float* pa1[APASS];
  const float* pa2[APASS];
  unsigned vmask[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int m = m0 + arow + 32 * i;
    const bool valid = m < p.M;
    const size_t mm = valid ? (size_t)m : 0;
    if (MODE == GEMM_DENSE) {
      pa1[i] = p.a1 + mm * p.lda1 + c4;
      pa2[i] = p.a2 ? p.a2 + mm * p.lda2 + c4 : p.a1;
      vmask[i] = valid ? 0x1ffu : 0u;
    } else {
      const int hw = p.H * p.W;
      const int rem = (int)(mm % hw);
      const int y = rem / p.W, x = rem - y * p.W;
      pa1[i] = p.a1 + mm * p.Cin + c4;
      pa2[i] = p.a1;
      unsigned vm = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        if (valid && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) vm |= 1u << t;
      }
      vmask[i] = vm;
    }
  }
  auto load_a = [&](int s, float4 (&ra)[APASS]) {
    if (MODE == GEMM_DENSE) {
      const int k0 = s * 32;
      const bool first = k0 < p.k1;
      const long off = first ? k0 : k0 - p.k1;
#pragma unroll
      for (int i = 0; i < APASS; ++i) {
        const float4 v = *reinterpret_cast<const float4*>((first ? pa1[i] : pa2[i]) + off);
        ra[i] = (vmask[i] & 1u) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      const int cblocks = p.Cin >> 5;
      const int kyx = s / cblocks, ci0 = (s - kyx * cblocks) << 5;
      const int dy = kyx / 3 - 1, dx = kyx - (kyx / 3) * 3 - 1;
      const long delta = (long)(dy * p.W + dx) * p.Cin + ci0;
#pragma unroll
      for (int i = 0; i < APASS; ++i) {
        const bool ok = (vmask[i] >> kyx) & 1u;
        const float4 v = *reinterpret_cast<const float4*>(pa1[i] + (ok ? delta : 0));
        ra[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto store_a = [&](int buf, const float4 (&ra)[APASS]) {
    __bf16* la = &lds[buf * NS * PLANE];
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      float x[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
#pragma unroll
      for (int pl = 0; pl < NS; ++pl) {
        bf16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = (__bf16)x[e];
          if (!(DISN_ABL & 1)) x[e] -= (float)v[e];  // exact: the residual of a nearest-even bf16 rounding fits in fp32
        }
        *reinterpret_cast<bf16x4*>(&la[pl * PLANE + (arow + 32 * i) * LDA + c4]) = v;
      }
    }
  };
  // ---- B loader: fragment-ordered bf16, [k16 block][n32 block][lane][8] ---------------------------
  const int nb0 = (n0 >> 5) + wn * (BN / 64);
  const int nblocks = p.N >> 5;
  const size_t bplane = (size_t)p.K * p.N;  // bf16 elements per split plane of the packed weights
  auto load_b = [&](int s, bf16x8 (&rb)[NS][2][TN]) {
#pragma unroll
    for (int pl = 0; pl < NS; ++pl)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          rb[pl][kk][j] = *reinterpret_cast<const bf16x8*>(
              d.bpk + pl * bplane + (((size_t)(2 * s + kk) * nblocks + nb0 + j) * 64 + lane) * 8);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // software pipeline, prefetch distance 2: while step s computes, the A tile / B fragments of step
  // s+2 are in flight and those of step s+1 (requested one iteration ago) are converted and staged --
  // a full iteration of MFMAs (384 cycles at NS = 3) plus the other resident waves cover the L2 latency
  // (the 128x128 three-term variant has no registers for a third B set: distance 1 there)
  // PF = 2 costs ~40 VGPRs: chosen by the launcher for big grids only (small ones need the occupancy)
  constexpr bool PF2 = PF == 2 && !(NS == 3 && BM * BN >= 128 * 128);
  float4 a1[APASS], a2[PF2 ? APASS : 1];
  bf16x8 b0[NS][2][TN], b1[NS][2][TN], b2[PF2 ? NS : 1][2][TN];
  load_a(s0, a1);
  load_b(s0, b0);
  store_a(0, a1);
  if constexpr (PF2) {
    const int sa = s0 + 1 < s1 ? s0 + 1 : s0;
    load_a(sa, a1);
    load_b(sa, b1);
  }
  __syncthreads();
  int cur = 0;
  for (int s = s0; s < s1; ++s) {
    if constexpr (PF2) {
      const int sn = s + 2 < s1 ? s + 2 : s1 - 1;
      if (!(DISN_ABL & 4)) load_a(sn, a2);
      if (!(DISN_ABL & 2)) load_b(sn, b2);
    } else {
      const int sn = s + 1 < s1 ? s + 1 : s;
      if (!(DISN_ABL & 4)) load_a(sn, a1);
      if (!(DISN_ABL & 2)) load_b(sn, b1);
    }
    const __bf16* la = &lds[cur * NS * PLANE] + (wm * (BM / 2) + (lane & 31)) * LDA + 8 * (lane >> 5);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[NS][TM];
#pragma unroll
      for (int pl = 0; pl < NS; ++pl)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[pl][i] = *reinterpret_cast<const bf16x8*>(la + pl * PLANE + i * 32 * LDA + kk * 16);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (NS == 3 && !(DISN_ABL & 8)) {  // small terms first
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], b0[1][kk][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], b0[2][kk][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], b0[0][kk][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], b0[1][kk][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], b0[0][kk][j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], b0[0][kk][j], acc[i][j], 0, 0, 0);
        }
    }
    if (!(DISN_ABL & 16)) store_a(cur ^ 1, a1);
    __syncthreads();
    cur ^= 1;
    if constexpr (PF2) {
#pragma unroll
      for (int i = 0; i < APASS; ++i)
        if (!(DISN_ABL & 4)) a1[i] = a2[i];
    }
#pragma unroll
    for (int pl = 0; pl < NS; ++pl)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (DISN_ABL & 2) continue;
          b0[pl][kk][j] = b1[pl][kk][j];
          if constexpr (PF2) b1[pl][kk][j] = b2[pl][kk][j];
        }
  }

  // ---- epilogue: bias, ReLU, fp32 store.  Each wave turns its 32x32 accumulator tiles into row-major
  // order through a private 32x36-float slice of the (now idle) LDS and stores 16 bytes per lane, 8
  // lanes per 128-byte row (4 store instructions per tile instead of 16; DS operations of one wave
  // execute in order, so the exchange needs no barrier).  NS = 1 with the 64-row tile has too little
  // LDS for that and stores element-wise.
  constexpr bool STAGE = (size_t)LDS_ELEMS * sizeof(__bf16) >= (size_t)4 * 32 * 36 * sizeof(float);
  if (STAGE) {
    float* stg = reinterpret_cast<float*>(lds) + wave * 32 * 36;
    const int srow = lane >> 3, scol = (lane & 7) * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * (BN / 2) + j * 32 + scol;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (d.S == 1 && !p.rows_per_bias) bv = *reinterpret_cast<const float4*>(p.bias + col);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          stg[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 36 + (lane & 31)] = acc[i][j][r];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int rr = srow + 8 * k;
          float4 v = *reinterpret_cast<const float4*>(&stg[rr * 36 + scol]);
          const int row = m0 + wm * (BM / 2) + i * 32 + rr;
          if (row < p.M) {
            if (d.S > 1) {
              *reinterpret_cast<float4*>(d.ws + ((size_t)ks * p.M + row) * p.N + col) = v;
            } else {
              // rows_per_bias: bias row m / rows_per_bias (the folded per-image term of the global stream's fold2/conv1)
              if (p.rows_per_bias) bv = *reinterpret_cast<const float4*>(p.bias + (size_t)(row / p.rows_per_bias) * p.N + col);
              v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
              if (p.relu) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
              }
              *reinterpret_cast<float4*>(p.out + (size_t)row * p.ldc + col) = v;
            }
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
    const float bv = d.S > 1 ? 0.f : p.bias[col];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.M) {
          if (d.S > 1) {
            d.ws[((size_t)ks * p.M + row) * p.N + col] = acc[i][j][r];
          } else {
            float v = acc[i][j][r] + (p.rows_per_bias ? p.bias[(size_t)(row / p.rows_per_bias) * p.N + col] : bv);
            if (p.relu) v = fmaxf(v, 0.f);
            p.out[(size_t)row * p.ldc + col] = v;
          }
        }
      }
  }
}

template <int BM, int BN, int NS, int PF>
static hipError_t bf_launch_kernel(const BfDev& d, GemmMode mode, hipStream_t st) {
  const dim3 grid(d.mtiles * d.ntiles, d.S);
  if (mode == GEMM_DENSE)
    hipLaunchKernelGGL((gemm_bf16_mfma<BM, BN, GEMM_DENSE, NS, PF>), grid, dim3(256), 0, st, d);
  else
    hipLaunchKernelGGL((gemm_bf16_mfma<BM, BN, GEMM_CONV3, NS, PF>), grid, dim3(256), 0, st, d);
  return hipGetLastError();
}

template <int BM, int BN>
static hipError_t bf_launch_mode(const BfDev& d, GemmMode mode, hipStream_t st) {
  // prefetch distance 2 (more registers, fewer resident waves) pays once the grid is several waves
  // deep: measured -7 % on the training / dense-grid shapes, +15 % on the 200..1200-workgroup layers
  // of a single image
  const bool deep = (long)d.mtiles * d.ntiles * d.S >= 2048;
  hipError_t e;
  if (d.nsplit == 3)
    e = deep ? bf_launch_kernel<BM, BN, 3, 2>(d, mode, st) : bf_launch_kernel<BM, BN, 3, 1>(d, mode, st);
  else
    e = deep ? bf_launch_kernel<BM, BN, 1, 2>(d, mode, st) : bf_launch_kernel<BM, BN, 1, 1>(d, mode, st);
  if (e != hipSuccess || d.S == 1) return e;
  if (d.pool_out)
    return splitk_reduce_pool_launch(d.ws, d.S, d.p.M / (d.p.H * d.p.W), d.p.H, d.p.W, d.p.N, d.p.bias,
                                     d.p.relu, d.p.out, d.pool_out, st);
  return splitk_reduce_launch(d.ws, d.S, d.p.M, d.p.N, d.p.bias, d.p.rows_per_bias, d.p.relu, d.p.out, d.p.ldc, st);
}

// split-K factor for a layer with few 64x64 tiles (the 14x14 / 28x28 convolutions, small point
// sets): enough workgroups for ~3 per CU, at least 4 k-steps each, partials within ws_bytes
static int bf_splits(long tiles, int ksteps, int M, int N, size_t ws_bytes) {
  const int forced = tune::bf_splits;  // 0 in the product build
  if (tiles >= 512 && !forced) return 1;
  // measured per layer at B = 1 (split-factor sweep of a tuning build): ~1200 workgroups in flight and at least
  // 12 k-steps per workgroup (a shorter loop does not amortise its prologue and the reduce pass)
  int s = forced ? forced : (int)((1176 + tiles / 2) / tiles);
  if (s < 1) s = 1;
  if (s > 16) s = 16;
  const int min_steps = forced ? 1 : 12;
  while (s > 1 && (ksteps / s < min_steps || (size_t)s * M * N * sizeof(float) > ws_bytes)) --s;
  return s;
}

size_t gemm_bf16_ws_bytes(int M, int N, int K) {
  const long tiles = (long)((M + 63) / 64) * (N / 64);
  return (size_t)bf_splits(tiles, K / 32, M, N, ~size_t(0)) * M * N * sizeof(float);
}

// p.bp is ignored; bpk = pack_bf16_launch output for the [p.K][p.N] operand; mode DENSE or CONV3;
// ws: gemm_bf16_ws_bytes (less is allowed: fewer splits)
hipError_t gemm_bf16_launch(const GemmParams& p, GemmMode mode, const void* bpk, float* ws,
                            size_t ws_bytes, hipStream_t st, int nsplit, float* pool_out, bool* pooled) {
  BfDev d;
  d.pool_out = nullptr;
  d.nmajor = 0;
  if (pooled) *pooled = false;
  d.p = p;
  d.nsplit = nsplit == 3 ? 3 : 1;
  d.bpk = reinterpret_cast<const __bf16*>(bpk);
  d.S = 1;
  d.ws = ws;
  // 128x128 when that still gives every CU two tiles; otherwise the small tile (+ split-K).
  // Measured at the B = 8 training shapes: 128x128 with 256..511 tiles and a 128x64 tile for the
  // 64-channel layers are both SLOWER than 64x64 (2.33 vs 2.13 ms over the 24 conv launches).
  const long t128 = (long)((p.M + 127) / 128) * (p.N / 128);
  if (p.N % 128 == 0 && t128 >= 512) {
    d.mtiles = (p.M + 127) / 128; d.ntiles = p.N / 128;
    return bf_launch_mode<128, 128>(d, mode, st);
  }
  d.mtiles = (p.M + 63) / 64; d.ntiles = p.N / 64;
  d.S = ws ? bf_splits((long)d.mtiles * d.ntiles, p.K / 32, p.M, p.N, ws_bytes) : 1;
  {  // keep the LARGER operand stationary in the XCDs' L2s: bytes of the weight image vs bytes of the input
    const size_t b_bytes = (size_t)p.K * p.N * 2 * (nsplit == 3 ? 3 : 1);
    const size_t a_bytes = (size_t)p.M * (mode == GEMM_CONV3 ? p.Cin : p.K) * sizeof(float);
    d.nmajor = b_bytes > a_bytes;
  }
  if (pool_out && pooled && d.S > 1 && mode == GEMM_CONV3 && p.ldc == p.N && !(p.H & 1) && !(p.W & 1)) {
    d.pool_out = pool_out;
    *pooled = true;
  }
  return bf_launch_mode<64, 64>(d, mode, st);
}

}  // namespace disn

// ---- C ABI: the two layer types in bf16 compute (unit-test / composition surface) -----------------
#include "../../include/disn_amd.h"

extern "C" {

static size_t bf_packed_bytes(int K, int N) { return (((size_t)3 * K * N * 2) + 255) & ~size_t(255); }

size_t disn_dense_bf16_workspace_bytes(int M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0 || K % 32 || N % 64) return 0;
  return bf_packed_bytes(K, N) + disn::gemm_bf16_ws_bytes(M, N, K);
}

int disn_dense_bf16(const float* a1, int lda1, int k1, const float* a2, int lda2, int k2, int M,
                    const float* w_kn, const float* bias, int N, int relu, int nsplit, float* out, void* ws,
                    size_t ws_bytes, void* stream) {
  if (!a1 || !w_kn || !bias || !out || !ws || M <= 0 || k1 <= 0 || k2 < 0 || (k2 > 0 && !a2)) return DISN_E_ARG;
  if (k1 % 32 || k2 % 32 || N <= 0 || N % 64 || lda1 < k1 || (k2 > 0 && lda2 < k2) || lda1 % 4 ||
      (k2 > 0 && lda2 % 4))
    return DISN_E_SHAPE;
  const int K = k1 + k2;
  if (ws_bytes < disn_dense_bf16_workspace_bytes(M, K, N)) return DISN_E_WS;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = disn::pack_bf16_launch(w_kn, 0, K, N, ws, st, nsplit);
  if (e != hipSuccess) return (int)e;
  disn::GemmParams p{};
  p.a1 = a1; p.lda1 = lda1; p.k1 = k1; p.a2 = a2; p.lda2 = lda2;
  p.M = M; p.N = N; p.K = K;
  p.bias = bias; p.out = out; p.ldc = N; p.relu = relu;
  const size_t pb = bf_packed_bytes(K, N);
  e = disn::gemm_bf16_launch(p, disn::GEMM_DENSE, ws, reinterpret_cast<float*>(static_cast<char*>(ws) + pb),
                             ws_bytes - pb, st, nsplit);
  return e == hipSuccess ? 0 : (int)e;
}

// the product form of the three-term path: weights already in disn_pack_kn_x3 order
size_t disn_conv3x3_x3_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 64) return 0;
  const size_t b = disn::gemm_bf16_ws_bytes(B * H * W, Cout, 9 * Cin);
  return b > 256 ? b : 256;
}

int disn_conv3x3_x3(const float* in, int B, int H, int W, int Cin, const void* w_x3, const float* bias,
                    int Cout, int relu, float* out, void* ws, size_t ws_bytes, void* stream) {
  if (!in || !w_x3 || !bias || !out || B <= 0 || H <= 0 || W <= 0) return DISN_E_ARG;
  if (Cin <= 0 || Cin % 32 || Cout <= 0 || Cout % 64) return DISN_E_SHAPE;
  disn::GemmParams p{};
  p.a1 = in; p.H = H; p.W = W; p.Cin = Cin;
  p.M = B * H * W; p.N = Cout; p.K = 9 * Cin;
  p.bias = bias; p.out = out; p.ldc = Cout; p.relu = relu;
  const hipError_t e = disn::gemm_bf16_launch(p, disn::GEMM_CONV3, w_x3, static_cast<float*>(ws),
                                              ws ? ws_bytes : 0, (hipStream_t)stream, 3);
  return e == hipSuccess ? 0 : (int)e;
}

size_t disn_conv3x3_bf16_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 64) return 0;
  return bf_packed_bytes(9 * Cin, Cout) + disn::gemm_bf16_ws_bytes(B * H * W, Cout, 9 * Cin);
}

int disn_conv3x3_bf16(const float* in, int B, int H, int W, int Cin, const float* w_hwio,
                      const float* bias, int Cout, int relu, int nsplit, float* out, void* ws,
                      size_t ws_bytes, void* stream) {
  if (!in || !w_hwio || !bias || !out || !ws || B <= 0 || H <= 0 || W <= 0) return DISN_E_ARG;
  if (Cin <= 0 || Cin % 32 || Cout <= 0 || Cout % 64) return DISN_E_SHAPE;
  if (ws_bytes < disn_conv3x3_bf16_workspace_bytes(B, H, W, Cin, Cout)) return DISN_E_WS;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = disn::tune::skip_pack ? hipSuccess : disn::pack_bf16_launch(w_hwio, 0, 9 * Cin, Cout, ws, st, nsplit);
  if (e != hipSuccess) return (int)e;
  disn::GemmParams p{};
  p.a1 = in; p.H = H; p.W = W; p.Cin = Cin;
  p.M = B * H * W; p.N = Cout; p.K = 9 * Cin;
  p.bias = bias; p.out = out; p.ldc = Cout; p.relu = relu;
  const size_t pb = bf_packed_bytes(9 * Cin, Cout);
  e = disn::gemm_bf16_launch(p, disn::GEMM_CONV3, ws, reinterpret_cast<float*>(static_cast<char*>(ws) + pb),
                             ws_bytes - pb, st, nsplit);
  return e == hipSuccess ? 0 : (int)e;
}

}  // extern "C"
