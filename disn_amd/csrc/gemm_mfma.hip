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
#include "tuning.hpp"


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
  float* ws;   // stream-K partial slabs [2*W][BM*BN] (tile-local row-major)
  int ksteps, mtiles, ntiles;
  int W;       // workgroups = gridDim.x
  long units;  // mtiles * ntiles * ksteps
  long long* dbg;  // profiling only (ABL & 16): per-workgroup phase timestamps; null in the product
};

// Stream-K decomposition.  The work is the list of (tile, k-step) units, tile-major; workgroup w
// owns the contiguous range [U*w/W, U*(w+1)/W).  A range is cut into SEGMENTS at tile borders:
// a segment that covers a whole tile [0,KS) is finished in place (bias + ReLU fused); a partial
// one goes to slab 2w (the workgroup's first segment) or 2w+1 (its last) and streamk_fixup sums
// the slabs of every cut tile in workgroup order (deterministic).  With W = tiles*S this is
// classic split-K, with W = tiles plain data-parallel; W = 256*G balances any shape over the 256
// CUs, which matters because one VGG layer at batch 1 is only 50-400 tiles (measured: 196 / 392
// / 784 equal workgroups take 1 / 2 / 4 rounds, tools/ubench/mfma_ubench.hip).
__device__ __forceinline__ long unit_begin(long U, int W, int w) { return (U * w) / W; }

// ABL: profiling ablation mask, 0 in every product instantiation (tools/ubench/gemm_ablate.hip
// instantiates the others): 1 = no A-tile global loads in the loop, 2 = no B loads in the loop,
// 4 = no LDS store + barrier, 8 = no A-fragment ds_reads.  Results are garbage when != 0.
template <int BM, int BN, int MODE, int ABL = 0>
__global__ __launch_bounds__(256, (BM * BN >= 128 * 128) ? 2 : 1)  // 128x128 must fit two waves per SIMD
void gemm_f32_mfma(const GemmDev d) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int LDA = 36;
  constexpr int APASS = BM / 32;
  __shared__ __attribute__((aligned(16))) float lds[2 * BM * LDA];

  const GemmParams& p = d.p;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware order: block b runs on XCD b%8; give each XCD a contiguous run of logical
  // workgroup ids, i.e. of tiles (neighbouring tiles share A rows / B columns in that XCD's L2).
  int w = blockIdx.x;
  {
    const int q = d.W >> 3, r = d.W & 7, xcd = w & 7, idx = w >> 3;
    w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int KS = d.ksteps;
  const long u0 = unit_begin(d.units, d.W, w), u1 = unit_begin(d.units, d.W, w + 1);

  int dbg_n = 0;
  if ((ABL & 16) && threadIdx.x == 0) d.dbg[(size_t)blockIdx.x * 16 + dbg_n++] = __builtin_readcyclecounter();
  for (long u = u0; u < u1;) {
  // The thread id is made opaque per segment: otherwise LICM hoists every lane-dependent address
  // (LDS offsets, row pointers, epilogue indices) out of this loop and keeps ~100 extra VGPRs live
  // across the K loop -- one wave per SIMD instead of two.
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int c4 = tid & 7;
  const int arow = tid >> 3;
  const int tile = (int)(u / KS);
  const int s_begin = (int)(u - (long)tile * KS);
  const long rest = u1 - (long)tile * KS;
  const int s_end = rest < KS ? (int)rest : KS;
  const bool complete = s_begin == 0 && s_end == KS;
  const int slot = (u == u0) ? 2 * w : 2 * w + 1;
  u += s_end - s_begin;
  const int mt = tile / d.ntiles, nt = tile - mt * d.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread A-loader state: rows (tid>>3)+32*pass, 16-byte column c4 ---------------
  // Everything per-row is precomputed ONCE: a base pointer and, for the convolution, a 9-bit
  // validity mask of the 3x3 taps.  Per k-step only a wave-uniform offset (SALU) is added.
  const float* pa1[APASS];
  const float* pa2[APASS];
  unsigned vmask[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int m = m0 + arow + 32 * i;
    const bool valid = m < p.M;
    const size_t mm = valid ? (size_t)m : 0;
    if (MODE == GEMM_DENSE) {
      pa1[i] = p.a1 + mm * p.lda1 + c4 * 4;
      pa2[i] = p.a2 ? p.a2 + mm * p.lda2 + c4 * 4 : p.a1;
      vmask[i] = valid ? 0x1ffu : 0u;
    } else {
      const int hw = p.H * p.W;
      const int rem = (int)(mm % hw);
      const int y = rem / p.W, x = rem - y * p.W;
      pa1[i] = p.a1 + mm * p.Cin + (MODE == GEMM_CONV3 ? c4 * 4 : 0);
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

  // A cursor: wave-uniform position of the NEXT A tile to request.  It advances by one k-step
  // per load_a() and sticks at the last step (the tail re-requests valid data; result unused),
  // all with selects -- the K loop stays one basic block.
  int a_step = s_begin;
  int a_kyx = 0, a_ci0 = 0;
  if (MODE == GEMM_CONV3) {
    const int cblocks = p.Cin >> 5;
    a_kyx = s_begin / cblocks;
    a_ci0 = (s_begin - a_kyx * cblocks) << 5;
  }
  auto load_a = [&](float4 (&areg)[APASS], unsigned& aok) {
    aok = 0;
    if (MODE == GEMM_DENSE) {
      const int k0 = a_step * 32;
      const bool first = k0 < p.k1;
      const long off = first ? k0 : k0 - p.k1;
#pragma unroll
      for (int i = 0; i < APASS; ++i) {
        areg[i] = *reinterpret_cast<const float4*>((first ? pa1[i] : pa2[i]) + off);
        aok |= (vmask[i] & 1u) << i;
      }
    } else if (MODE == GEMM_CONV3) {
      const int dy = a_kyx / 3 - 1, dx = a_kyx - (a_kyx / 3) * 3 - 1;
      const long delta = (long)(dy * p.W + dx) * p.Cin + a_ci0;
#pragma unroll
      for (int i = 0; i < APASS; ++i) {
        const unsigned ok = (vmask[i] >> a_kyx) & 1u;
        // out-of-image taps read the (always valid) tensor base and are zeroed at store time
        areg[i] = *reinterpret_cast<const float4*>(ok ? pa1[i] + delta : p.a1);
        aok |= ok << i;
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
          const bool ok = k < 27 && ((vmask[i] >> kyx) & 1u);
          v[j] = ok ? pa1[i][(dy * p.W + dx) * 3 + ci] : 0.f;
        }
        areg[i] = make_float4(v[0], v[1], v[2], v[3]);
        aok |= 1u << i;
      }
    }
    const int adv = (a_step + 1 < s_end) ? 1 : 0;
    a_step += adv;
    if (MODE == GEMM_CONV3) {
      a_ci0 += 32 * adv;
      const int wrap = (a_ci0 == p.Cin) ? 1 : 0;
      a_ci0 = wrap ? 0 : a_ci0;
      a_kyx += wrap;
    }
  };
  auto store_a = [&](int buf, const float4 (&areg)[APASS], unsigned aok) {
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

  const int nb32 = p.N >> 5;
  const int frag_row = wm * (BM / 2) + (lane & 31);
  const int frag_k = (lane >> 5) * 4;
  // B cursor: per-lane pointer to this wave's first fragment of the next step to request
  const size_t b_stride = (size_t)4 * nb32 * 256;
  const float* bptr = p.bp + ((size_t)s_begin * 4 * nb32 + (n0 >> 5) + wn * TN) * 256 + lane * 4;
  int b_step = s_begin;
  auto load_b = [&](float4 (&b)[4][TN]) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[kb][j] = *reinterpret_cast<const float4*>(bptr + ((size_t)kb * nb32 + j) * 256);
    const int adv = (b_step + 1 < s_end) ? 1 : 0;
    b_step += adv;
    bptr += adv ? b_stride : 0;
  };

  // Software pipeline, one step = 32 k = 16*TM*TN MFMAs per wave:
  //   * the B fragments of step s+1 and the A tile of step s+2 are requested DURING the MFMAs
  //     of step s, one VMEM instruction every Q MFMAs (sched_group_barrier): a 16-byte global
  //     load costs ~60 cycles of in-order issue, which hides under a 64-cycle fp32 MFMA but
  //     stalls the matrix pipe when ten of them are issued back to back (measured: 13-17 us of
  //     a 55 us layer, tools/ubench/gemm_ablate.hip);
  //   * the loop is unrolled by two with NAMED register sets (b0/b1, ra0/ra1): no register
  //     copies, which would force a wait on the prefetch;
  //   * end of step: A tile s+1 -> other LDS buffer, one barrier.
  constexpr int NMFMA = 16 * TM * TN;
  constexpr int NLOAD = 4 * TN + APASS;
  constexpr int Q = NMFMA / NLOAD > 0 ? NMFMA / NLOAD : 1;
  int cur = 0;
  float4 af[4][TM];
  auto step = [&](const float4 (&bc)[4][TN], float4 (&bn)[4][TN], const float4 (&a_st)[APASS],
                  unsigned ok_st, float4 (&a_ld)[APASS], unsigned& ok_ld) {
    const float* la = &lds[cur * BM * LDA + frag_row * LDA + frag_k];
    if (!(ABL & 8)) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[kb][i] = *reinterpret_cast<const float4*>(la + i * 32 * LDA + kb * 8);
    }
    if (!(ABL & 2)) load_b(bn);
    // (the K=27 first layer has a single k-step: nothing to prefetch, and its loader is 16 scalar
    //  loads per row)
    if (!(ABL & 1) && MODE != GEMM_CONV3_C3) load_a(a_ld, ok_ld);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kb][i].x, bc[kb][j].x, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kb][i].y, bc[kb][j].y, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kb][i].z, bc[kb][j].z, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kb][i].w, bc[kb][j].w, acc[i][j], 0, 0, 0);
    }
    // issue order: all A-fragment reads, then {Q MFMAs, 1 global load} x NLOAD, then the rest
    __builtin_amdgcn_sched_group_barrier(0x100, 4 * TM, 0);
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, Q, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - Q * NLOAD > 0 ? NMFMA - Q * NLOAD : 0, 0);
    if (!(ABL & 4)) {
      store_a(cur ^ 1, a_st, ok_st);     // tile of step s+1, requested one full step ago
      // the LDS stores stay behind the last MFMA: hoisted into the block they would need their
      // operands early and drain the in-flight prefetches (vmcnt(0)) mid-step
      __builtin_amdgcn_sched_group_barrier(0x200, APASS, 0);
      __syncthreads();
    }
    cur ^= 1;
  };
  float4 b0[4][TN], b1[4][TN];
  float4 ra0[APASS], ra1[APASS];
  unsigned ok0 = 0, ok1 = 0;
  if (s_begin < s_end) {
    load_a(ra0, ok0);          // tile s_begin
    load_b(b0);                // B of s_begin
    store_a(0, ra0, ok0);
    if (MODE != GEMM_CONV3_C3) load_a(ra0, ok0);   // tile s_begin+1 (in flight while step s_begin computes)
    if (ABL) {                 // give every register set a defined value for the ablated variants
      ra1[0] = ra0[0];
#pragma unroll
      for (int i = 0; i < APASS; ++i) ra1[i] = ra0[i];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int j = 0; j < TN; ++j) b1[kb][j] = b0[kb][j];
    }
  }
  __syncthreads();
  if ((ABL & 16) && threadIdx.x == 0 && dbg_n < 15) d.dbg[(size_t)blockIdx.x * 16 + dbg_n++] = __builtin_readcyclecounter();
  if (ABL & 8) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[kb][i] = *reinterpret_cast<const float4*>(&lds[frag_row * LDA + frag_k + i * 32 * LDA + kb * 8]);
  }
  int s = s_begin;
  for (; s + 1 < s_end; s += 2) {
    step(b0, b1, ra0, ok0, ra1, ok1);
    step(b1, b0, ra1, ok1, ra0, ok0);
  }
  if (s < s_end) step(b0, b1, ra0, ok0, ra1, ok1);

  // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31,
  //      row = (r&3) + 8*(r>>2) + 4*(lane>>5)  (cdna_hip_programming.md section 3) -------------
  if ((ABL & 16) && threadIdx.x == 0 && dbg_n < 15) d.dbg[(size_t)blockIdx.x * 16 + dbg_n++] = __builtin_readcyclecounter();
  // (the lane id is made opaque here: otherwise LICM hoists all of this lane-dependent address
  //  arithmetic out of the segment loop and keeps ~100 VGPRs live across the K loop)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  // Each wave turns its 32x32 accumulator tiles into row-major order through a private 32x36
  // slice of the (now idle) A-tile LDS and stores 16 bytes per lane, 8 lanes per 128-byte row:
  // 4 store instructions per tile instead of 16 scalar ones (the tail was store-ISSUE bound:
  // ~6.4k cycles per workgroup, tools/ubench/gemm_ablate.hip).  DS operations of one wave execute
  // in order, so the write -> read exchange needs no barrier.
  float* stage = &lds[wave * 32 * LDA];
  const int srow = lane_e >> 3, scol = (lane_e & 7) * 4;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int lcol = wn * (BN / 2) + j * 32 + scol;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (complete) bv = *reinterpret_cast<const float4*>(p.bias + n0 + lcol);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        stage[((r & 3) + 8 * (r >> 2) + 4 * (lane_e >> 5)) * LDA + (lane_e & 31)] = acc[i][j][r];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int rr = srow + 8 * k;
        float4 v = *reinterpret_cast<const float4*>(&stage[rr * LDA + scol]);
        const int lrow = wm * (BM / 2) + i * 32 + rr;
        if (complete) {
          if (m0 + lrow < p.M) {
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            if (p.relu) {
              v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            *reinterpret_cast<float4*>(p.out + (size_t)(m0 + lrow) * p.ldc + n0 + lcol) = v;
          }
        } else {
          *reinterpret_cast<float4*>(d.ws + ((size_t)slot * BM + lrow) * BN + lcol) = v;
        }
      }
    }
  }
  // the staging slices alias the A buffers the next segment's prologue writes
  if (u < u1) __syncthreads();
  if ((ABL & 16) && threadIdx.x == 0 && dbg_n < 15) d.dbg[(size_t)blockIdx.x * 16 + dbg_n++] = __builtin_readcyclecounter();
  }  // segment loop
  if ((ABL & 16) && threadIdx.x == 0) d.dbg[(size_t)blockIdx.x * 16 + 15] = dbg_n;
}

// Sum the partial slabs of every tile that stream-K cut, in workgroup order; + bias, ReLU.
// grid = (tiles, BM*BN/1024); a tile finished in place by one workgroup returns immediately.
template <int BM, int BN>
__global__ __launch_bounds__(256) void streamk_fixup(const GemmDev d) {
  const GemmParams& p = d.p;
  const int tile = blockIdx.x;
  const long KS = d.ksteps, tb = (long)tile * KS, te = tb + KS;
  int w = (int)((tb * d.W) / d.units);
  while (w + 1 < d.W && unit_begin(d.units, d.W, w + 1) <= tb) ++w;
  while (w > 0 && unit_begin(d.units, d.W, w) > tb) --w;
  if (unit_begin(d.units, d.W, w) <= tb && unit_begin(d.units, d.W, w + 1) >= te) return;
  const int idx4 = blockIdx.y * 256 + threadIdx.x;
  const int lrow = idx4 / (BN / 4), lcol = (idx4 - lrow * (BN / 4)) * 4;
  // contributing workgroups w .. w_last (the planner guarantees W <= units: no empty ranges).  Only
  // the first one can have started before this tile (its segment is then its LAST -> slab 2w+1);
  // every later one starts inside the tile (its FIRST segment -> slab 2w).  Both bounds are found
  // once (wave-uniform), so the summation loop is divide-free and its loads are independent.
  int w_last = (int)(((te - 1) * d.W) / d.units);
  while (w_last + 1 < d.W && unit_begin(d.units, d.W, w_last + 1) <= te - 1) ++w_last;
  while (w_last > 0 && unit_begin(d.units, d.W, w_last) > te - 1) --w_last;
  const float* base = d.ws + (size_t)lrow * BN + lcol;
  const size_t slab = (size_t)BM * BN;
  const int first_slot = (unit_begin(d.units, d.W, w) >= tb) ? 2 * w : 2 * w + 1;
  float4 v = *reinterpret_cast<const float4*>(base + (size_t)first_slot * slab);
#pragma unroll 4
  for (int k = w + 1; k <= w_last; ++k) {
    const float4 u = *reinterpret_cast<const float4*>(base + (size_t)(2 * k) * slab);
    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
  }
  const int mt = tile / d.ntiles, nt = tile - mt * d.ntiles;
  const int row = mt * BM + lrow, col = nt * BN + lcol;
  if (row >= p.M) return;
  const float4 bv = *reinterpret_cast<const float4*>(p.bias + col);
  v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
  if (p.relu) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  }
  *reinterpret_cast<float4*>(p.out + (size_t)row * p.ldc + col) = v;
}

// out[m][n] = act(sum_s ws[s][m][n] + bias[row(m)][n]).  One float4 column group per CG
// threads; the S partial slabs are split over SL "s-lanes" of the block (independent loads in
// flight instead of one latency-serial chain) and combined through LDS in a fixed order, so the
// result is deterministic.  SL is chosen by the launcher from the amount of parallelism M*N/4.
template <int SL>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int S,
                                                            int M, int N,
                                                            const float* __restrict__ bias,
                                                            int rows_per_bias, int relu,
                                                            float* __restrict__ out, int ldc) {
  constexpr int CG = 256 / SL;
  __shared__ float4 red[SL > 1 ? SL : 1][CG];
  const int sl = threadIdx.x / CG, cg = threadIdx.x - sl * CG;
  const size_t n4 = (size_t)N >> 2;
  const size_t total = (size_t)M * n4;
  const size_t i = (size_t)blockIdx.x * CG + cg;
  const bool live = i < total;
  const size_t m = live ? i / n4 : 0;
  const int c = live ? (int)(i - m * n4) * 4 : 0;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    const float* p = ws + m * N + c;
    const size_t slab = (size_t)M * N;
#pragma unroll 4
    for (int s = sl; s < S; s += SL) {
      const float4 u = *reinterpret_cast<const float4*>(p + (size_t)s * slab);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
  }
  if (SL > 1) {
    red[sl][cg] = v;
    __syncthreads();
    if (sl != 0) return;
#pragma unroll
    for (int k = 1; k < SL; ++k) {
      const float4 u = red[k][cg];
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
  }
  if (!live) return;
  const size_t brow = rows_per_bias ? m / rows_per_bias : 0;
  const float4 bv = *reinterpret_cast<const float4*>(bias + brow * N + c);
  v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
  if (relu) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  }
  *reinterpret_cast<float4*>(out + m * ldc + c) = v;
}

// splitk_reduce for a convolution that a 2x2 max pool follows (NHWC rows m = (b*H + y)*W + x): one
// column group per 2x2 quad, the same s-lane partition and summation order per pixel as
// splitk_reduce_kernel<SL> (bit-identical outputs), then the four finished pixels are stored (the
// layer's own output, a tap) and their maximum goes to pool_out -- the pool costs no launch and
// no re-read.
template <int SL>
__global__ __launch_bounds__(256) void splitk_reduce_pool_kernel(const float* __restrict__ ws, int S,
                                                                 int B, int H, int W, int N,
                                                                 const float* __restrict__ bias,
                                                                 int relu, float* __restrict__ out,
                                                                 float* __restrict__ pool_out) {
  constexpr int CG = 256 / SL;
  __shared__ float4 red[SL > 1 ? SL : 1][4][CG];
  const int sl = threadIdx.x / CG, cg = threadIdx.x - sl * CG;
  const int Ho = H >> 1, Wo = W >> 1;
  const size_t n4 = (size_t)N >> 2;
  const size_t total = (size_t)B * Ho * Wo * n4;
  const size_t i = (size_t)blockIdx.x * CG + cg;
  const bool live = i < total;
  size_t q = live ? i / n4 : 0;
  const int c = live ? (int)(i - q * n4) * 4 : 0;
  const int ox = (int)(q % Wo);
  q /= Wo;
  const int oy = (int)(q % Ho);
  const int b = (int)(q / Ho);
  const size_t m00 = ((size_t)b * H + 2 * oy) * W + 2 * ox;
  const size_t moff[4] = {0, 1, (size_t)W, (size_t)W + 1};
  float4 v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    const size_t slab = (size_t)B * H * W * N;
#pragma unroll 6
    for (int s = sl; s < S; s += SL) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 u = *reinterpret_cast<const float4*>(ws + (size_t)s * slab + (m00 + moff[j]) * N + c);
        v[j].x += u.x; v[j].y += u.y; v[j].z += u.z; v[j].w += u.w;
      }
    }
  }
  if (SL > 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) red[sl][j][cg] = v[j];
    __syncthreads();
    if (sl != 0) return;
#pragma unroll
    for (int k = 1; k < SL; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 u = red[k][j][cg];
        v[j].x += u.x; v[j].y += u.y; v[j].z += u.z; v[j].w += u.w;
      }
  }
  if (!live) return;
  const float4 bv = *reinterpret_cast<const float4*>(bias + c);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j].x += bv.x; v[j].y += bv.y; v[j].z += bv.z; v[j].w += bv.w;
    if (relu) {
      v[j].x = fmaxf(v[j].x, 0.f); v[j].y = fmaxf(v[j].y, 0.f);
      v[j].z = fmaxf(v[j].z, 0.f); v[j].w = fmaxf(v[j].w, 0.f);
    }
    *reinterpret_cast<float4*>(out + (m00 + moff[j]) * N + c) = v[j];
  }
  float4 o;
  o.x = fmaxf(fmaxf(v[0].x, v[1].x), fmaxf(v[2].x, v[3].x));
  o.y = fmaxf(fmaxf(v[0].y, v[1].y), fmaxf(v[2].y, v[3].y));
  o.z = fmaxf(fmaxf(v[0].z, v[1].z), fmaxf(v[2].z, v[3].z));
  o.w = fmaxf(fmaxf(v[0].w, v[1].w), fmaxf(v[2].w, v[3].w));
  *reinterpret_cast<float4*>(pool_out + (((size_t)b * Ho + oy) * Wo + ox) * N + c) = o;
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
static size_t slab_bytes(int bm, int bn, int W) { return (size_t)2 * W * bm * bn * sizeof(float); }

static bool needs_fixup(long tiles, int ksteps, int W) {
  const long U = tiles * ksteps;
  return !(U % W == 0 && (U / W) % ksteps == 0);
}

GemmPlan gemm_plan(int M, int N, int K, size_t max_ws, const int* force) {
  const int ksteps = K / 32;
  const int cand[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
  double best = 1e300;
  GemmPlan plan{64, 64, 1, 0};
  for (auto& c : cand) {
    const int bm = c[0], bn = c[1];
    if (N % bn) continue;
    const long tiles = (long)((M + bm - 1) / bm) * (N / bn);
    const long U = tiles * ksteps;
    const int tq = (bm / 64) * (bn / 64);           // 32x32 MFMA tiles per wave
    const double unit = tq * 16.0 * 64.0;            // MFMA cycles of one k-step per wave
    const int occ = tq >= 4 ? 2 : (tq == 2 ? 3 : 5); // co-resident workgroups per CU (VGPR bound)
    // candidate workgroup counts: data-parallel (one tile each) and stream-K over 256*G
    long wlist[4] = {tiles, 256, 512, 768};
    for (long W : wlist) {
      if (W > U) W = U;
      if (W < 1) continue;
      const bool fix = needs_fixup(tiles, ksteps, (int)W);
      if (fix && slab_bytes(bm, bn, (int)W) > max_ws) continue;
      const long per_wg = (U + W - 1) / W;          // k-steps of the busiest workgroup
      // workgroups the busiest CU executes: round-robin for a few, dynamically balanced (half a
      // workgroup of tail) once there are >= 4 per CU
      const double cu_wgs = W >= 1024 ? (double)W / 256.0 + (W % 256 ? 0.5 : 0.0) : (double)((W + 255) / 256);
      const long resident = (W + 255) / 256 < occ ? (W + 255) / 256 : occ;   // per CU at a time
      // Co-resident workgroups share the CU's MFMA pipes, so the busiest CU needs
      // cu_wgs * per_wg * unit MFMA cycles however they are scheduled.  MFMA-busy fractions fitted
      // to tools/sweep_gemm.py at B = 1 and B = 8: a wave alone on its SIMD loses ~14 % to the
      // per-step LDS/barrier bubble; bigger tiles re-use more of each staged operand, and the A
      // side (im2col gather) is the expensive one to widen.
      const double tile_eff = tq >= 4 ? 0.93 : (tq == 1 ? 0.84 : (bm == 128 ? 0.79 : 0.86));
      const double eff = (resident >= 2 ? 1.0 : 0.86) * tile_eff;
      double cost = cu_wgs * per_wg * unit / eff + 6000.0;
      // persistent stream-K workgroups finish together; the dispatcher balances one-tile
      // workgroups a little better
      if (W != tiles) cost *= 1.03;
      cost += 600.0 * cu_wgs;                        // prologue/epilogue per workgroup
      if (fix) {
        const double segs = (double)W + (double)(tiles < W ? tiles : W);
        cost += 16000.0 + segs * bm * bn * 8.0 / 2500.0;  // fix-up launch + slab write/read
      }
      if (cost < best) {
        best = cost;
        plan.bm = bm; plan.bn = bn; plan.wgs = (int)W;
      }
    }
  }
  // an explicit plan (disn_conv3x3_planned: the plan-invariance test surface): {BM, BN, workgroups}
  if (!force && tune::gemm_force[0]) force = tune::gemm_force;  // tuning builds only
  int fbm = force ? force[0] : 0, fbn = force ? force[1] : 0, fw = force ? force[2] : 0;
  if (force && (fbm == 64 || fbm == 128) && (fbn == 64 || fbn == 128) && N % fbn == 0 &&
      (fw >= 1 || fw == -1)) {  // W = -1: one workgroup per tile
    const long tiles = (long)((M + fbm - 1) / fbm) * (N / fbn);
    if (fw == -1) fw = (int)tiles;
    if (fw > tiles * ksteps) fw = (int)(tiles * ksteps);
    if (!needs_fixup(tiles, ksteps, fw) || slab_bytes(fbm, fbn, fw) <= max_ws) {
      plan.bm = fbm; plan.bn = fbn; plan.wgs = fw;
    }
  }
  {
    const long tiles = (long)((M + plan.bm - 1) / plan.bm) * (N / plan.bn);
    plan.ws_bytes = needs_fixup(tiles, ksteps, plan.wgs) ? slab_bytes(plan.bm, plan.bn, plan.wgs) : 0;
  }
  return plan;
}

template <int BM, int BN>
static hipError_t launch_mode(const GemmDev& d, GemmMode mode, bool fix, hipStream_t st) {
  const dim3 grid(d.W);
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
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || !fix) return e;
  hipLaunchKernelGGL((streamk_fixup<BM, BN>), dim3(d.mtiles * d.ntiles, BM * BN / 1024), dim3(256), 0,
                     st, d);
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
  d.W = plan.wgs;
  d.units = (long)d.mtiles * d.ntiles * d.ksteps;
  d.dbg = nullptr;
  const bool fix = plan.ws_bytes != 0;
  if (plan.bm == 128 && plan.bn == 128) return launch_mode<128, 128>(d, mode, fix, st);
  if (plan.bm == 128) return launch_mode<128, 64>(d, mode, fix, st);
  if (plan.bn == 128) return launch_mode<64, 128>(d, mode, fix, st);
  return launch_mode<64, 64>(d, mode, fix, st);
}

hipError_t splitk_reduce_launch(const float* ws, int S, int M, int N, const float* bias,
                                int rows_per_bias, int relu, float* out, int ldc, hipStream_t st, int force_sl) {
  const size_t total = (size_t)M * (N / 4);
  // enough column groups to fill the chip -> 1 s-lane; otherwise spread S over 4 or 16 lanes.  force_sl (1, 4, 16): the
  // caller fixes the lane count -- and with it the summation order -- whatever M (the fc layers of a batched call: a
  // row's bits must not depend on how many rows travel with it)
  const int sl = force_sl ? force_sl : (total >= 131072 || S < 4) ? 1 : ((total >= 16384 || S < 16) ? 4 : 16);
  const int cg = 256 / sl;
  const unsigned blocks = (unsigned)((total + cg - 1) / cg);
  switch (sl) {
    case 1:
      hipLaunchKernelGGL((splitk_reduce_kernel<1>), dim3(blocks), dim3(256), 0, st, ws, S, M, N, bias,
                         rows_per_bias, relu, out, ldc);
      break;
    case 4:
      hipLaunchKernelGGL((splitk_reduce_kernel<4>), dim3(blocks), dim3(256), 0, st, ws, S, M, N, bias,
                         rows_per_bias, relu, out, ldc);
      break;
    default:
      hipLaunchKernelGGL((splitk_reduce_kernel<16>), dim3(blocks), dim3(256), 0, st, ws, S, M, N,
                         bias, rows_per_bias, relu, out, ldc);
      break;
  }
  return hipGetLastError();
}

hipError_t splitk_reduce_pool_launch(const float* ws, int S, int B, int H, int W, int N,
                                     const float* bias, int relu, float* out, float* pool_out,
                                     hipStream_t st) {
  // the s-lane count splitk_reduce_launch would use for this layer: same summation order
  const size_t total = (size_t)B * H * W * (N / 4);
  const int sl = (total >= 131072 || S < 4) ? 1 : ((total >= 16384 || S < 16) ? 4 : 16);
  const int cg = 256 / sl;
  const size_t quads = total / 4;
  const unsigned blocks = (unsigned)((quads + cg - 1) / cg);
  switch (sl) {
    case 1:
      hipLaunchKernelGGL((splitk_reduce_pool_kernel<1>), dim3(blocks), dim3(256), 0, st, ws, S, B, H, W, N,
                         bias, relu, out, pool_out);
      break;
    case 4:
      hipLaunchKernelGGL((splitk_reduce_pool_kernel<4>), dim3(blocks), dim3(256), 0, st, ws, S, B, H, W, N,
                         bias, relu, out, pool_out);
      break;
    default:
      hipLaunchKernelGGL((splitk_reduce_pool_kernel<16>), dim3(blocks), dim3(256), 0, st, ws, S, B, H, W,
                         N, bias, relu, out, pool_out);
      break;
  }
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
