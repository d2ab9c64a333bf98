// 3x3 SAME convolution of the VGG-16 stack (models/CNN/vgg.py:187-196, 'conv1_2' .. 'conv4_3'; called from
// models/model_normalization.py:74-76) for a BATCH of images (the calls of several steps at once): the throughput
// sibling of conv_h2.hip.  Same arithmetic (two-term f16 split of both operands, l_a h_b + h_a l_b + h_a h_b on
// v_mfma_f32_32x32x16_f16, fp32 accumulate), same weight image (conv_h2_pack_kernel), same activation-maximum
// slots; what changes is who owns what:
//
//  * conv_h2.hip is cut for ONE image: all K parallelism inside the workgroup (4 / 8 k-waves per n-block), tiles
//    of 32 .. 128 pixels, so that 196 .. 50176 pixels still give a few hundred workgroups.  Per weight fragment
//    pair (2 KiB from L2) a wave issues 3 .. 12 MFMAs: with thousands of workgroups (a batch) the weight stream
//    per MFMA, not the matrix pipe, sets the pace (r02w: 0.26 of the f16 peak, 4.2x the algorithmic reads).
//  * Here a wave owns ONE 32-channel n-block for MB = 4 .. 7 row blocks (128 .. 224 pixels) and walks K
//    sequentially (optionally two k-waves where a layer's M x N is too small for 1024 SIMDs): 12 .. 21 MFMAs per
//    weight pair, the pair queue three taps deep, and a workgroup's NWV n-waves share one halo patch in LDS.
//    The patch is 8 rows x 28 (or 32) pixels = exactly 7 (8) blocks of 32: no padded MFMA rows where the image
//    size is a multiple of the patch (224, 112, 56: all of conv1_2 .. conv3_3; 28-pixel images waste the patch rows
//    28 .. 31).
//  * Row <-> pixel map: logical rows 4w .. 4w + 3 of the tile are the 2x2 WINDOW w (windows row-major over the
//    patch).  The C layout of the MFMA hands a lane four consecutive logical rows per register quad
//    (h2_common.hpp), i.e. one window: the fused 2x2 max pool is a maximum of four registers, no cross-lane
//    traffic, whatever the patch width.
//  * LDS: a halo pixel is CK f16 of h, CK of l, 16 B pad (80 B at CK = 16, 144 B at CK = 32: odd multiples of
//    16), and a halo ROW is padded to 128 (mod 256) bytes: the 16 lanes of a ds_read_b128 group read 2 rows x 8
//    consecutive pixels (four windows) and land on 16 different 16-byte slots (tests/conv_h2w_emulation.py
//    counts the conflicts of every configuration).  Tap shifts are immediate offsets of the read.
//
// Summation order of one output element: k16 blocks in ascending channel order within a k-wave, taps 0..8 inside a
// block, then (two k-waves) p0 + p1.  It depends on WK only -- fixed per LAYER by the launcher -- never on the patch,
// the number of n-waves or the batch: an image's bits do not depend on the images it travels with.  They DO differ
// (fp32 rounding, both within the 1e-5 bar of the float64 oracle) from conv_h2.hip's four / eight-k-wave tree, which
// serves calls of fewer than four images: see conv_h2_launch.
#include "kernels.hpp"
#include "tuning.hpp"
#include "h2_common.hpp"

#include <type_traits>

namespace disn {

namespace {
constexpr int w_row_bytes(int raw) {  // smallest size >= raw that is 128 (mod 256)
  int r = (raw / 256) * 256 + 128;
  return r >= raw ? r : r + 256;
}
}  // namespace

#ifdef DISN_TUNING
#define CH2W_STAMP(i) \
  if (P.stamps && threadIdx.x == 0) P.stamps[(size_t)blockIdx.x * 16 + (i)] = (i) == 0 ? (long long)wall_clock64() : (long long)clock64()
#else
#define CH2W_STAMP(i)
#endif

// MB row blocks per wave, MWV m-waves x NWV n-waves x WK k-waves per workgroup, patch TH x TW (= 32 MB MWV pixels),
// OCC workgroups per CU the register budget is cut for
// PARK (WK == 1, SEG == 2): the TWO-K-HALVES order of the two-k-wave segmented variant in ONE k-wave -- the lower half of the
// k16 blocks first (p0: what k-wave 0 of <.., WK = 2, .., SEG = 2> sums), that total parked in LDS, then the upper half (p1), p0 + p1:
// bit for bit the two-k-wave variant's result with four n-waves per halo instead of two (large calls), see conv_h2w_launch
template <int MB, int MWV, int NWV, int WK, int TH, int TW, int OCC, int SEG = 0, bool PARK = false>
__global__ __launch_bounds__(64 * MWV * NWV * WK, (OCC * MWV * NWV * WK + 3) / 4) void conv_h2w_kernel(const ConvH2Dev P) {
  constexpr int NWAVES = MWV * NWV * WK, NT = 64 * NWAVES;
  constexpr int CK = 16 * WK;              // input channels per chunk: one k16 block per k-wave
  constexpr int KPIX = CK * 4 + 16;        // bytes per halo pixel
  constexpr int UPP = CK / 4;              // float4 units per pixel
  constexpr int RP = TW + 2, HR = TH + 2;  // halo pixels per row, halo rows
  constexpr int ROWB = w_row_bytes(RP * KPIX);
  constexpr int BUF = HR * ROWB;
  constexpr int UNITS = HR * RP * UPP;
  constexpr int LP = (UNITS + NT - 1) / NT;
  constexpr int WPR = TW / 2;              // windows per patch row pair
  constexpr int D = 3;                     // weight pairs in flight per wave (taps ahead)
  constexpr int XCH = WK == 2 ? MWV * NWV * MB * 4096 : 0;
  constexpr int PARKB = PARK ? NWAVES * MB * 4096 : 0;   // the parked p0 of every wave, behind the halo buffers
  static_assert(!PARK || (WK == 1 && SEG == 2), "PARK: one k-wave in segments of two chunks");
  constexpr int LDS_BYTES = PARK ? 2 * BUF + PARKB : (2 * BUF > XCH ? 2 * BUF : XCH);
  static_assert(TH * TW == 32 * MB * MWV && TH % 2 == 0 && TW % 2 == 0, "patch = whole row blocks of 2x2 windows");
  static_assert(WK == 1 || WK == 2, "k-waves");
  static_assert(LDS_BYTES * OCC <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave % WK, wn = (wave / WK) % NWV, wm = wave / (WK * NWV);
  const int j = lane & 31, g = lane >> 5;
  CH2W_STAMP(0);
  CH2W_STAMP(1);

  // ---- tile: n-tile major, every XCD (hardware workgroup L runs on XCD L % 8) a contiguous eighth --------------
  int l;
  {
    const int T = gridDim.x, L = blockIdx.x, q = T >> 3, r = T & 7, xcd = L & 7, idx = L >> 3;
    l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int per_img = P.tiles_y * P.tiles_x;
  const int mtiles = P.B * per_img;
  const int nt = l / mtiles;
  int mt = l - nt * mtiles;
  const int b = mt / per_img;
  mt -= b * per_img;
  const int tyi = mt / P.tiles_x, txi = mt - tyi * P.tiles_x;
  const int y0 = tyi * TH, x0 = txi * TW;
  const int n0 = (nt * NWV + wn) * 32;
  const int H = P.H, W = P.W, Cin = P.Cin, Cout = P.Cout;
  const int NC = Cin / CK;
  const float* inb = P.in + (size_t)b * H * W * Cin;
  // HALVES (the two-k-wave SEGMENTED variant): k-wave 0 walks the LOWER half of the k16 blocks, k-wave 1 the upper half --
  // chunk i holds blocks i and Cin / 32 + i (two 64-byte pieces of two cache lines per halo pixel; a line's other half is
  // the next chunk's) -- so that the one-k-wave PARK variant, which walks all blocks in order and parks the lower half's
  // total at the midpoint, sums the same things in the same order.  (A first version split even / odd blocks: the PARK
  // walk then touched every 128-byte line of the input twice, a whole K half apart -- 2.6 x the input bytes from the
  // memory side, profiles/r06r_pmc_traffic_even_odd.json.)
  constexpr bool HALVES = WK == 2 && SEG > 0;

  // ---- halo loader: unit u = (halo pixel, float4 of the chunk's CK channels).  Every load is unconditional and
  // every loaded value is used (an out-of-image unit reads a valid address and is ANDed with 0): no exec-mask
  // branches around loads, so the compiler can count the loads in flight (s_waitcnt vmcnt(N)) ----------------------
  int goff[LP], woff[LP];
  unsigned vbits = 0;
#pragma unroll
  for (int k = 0; k < LP; ++k) {
    // (threads beyond the last unit repeat it: same address, same value -- no branch around a load or a store)
    const int u = tid + k * NT < UNITS ? tid + k * NT : UNITS - 1;
    const int hp = u / UPP, c4 = u % UPP;
    const int hy = hp / RP, hx = hp - hy * RP;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = y >= 0 && y < H && x >= 0 && x < W;
    const int coff_u = HALVES ? (c4 < 4 ? 4 * c4 : (Cin >> 1) + 4 * (c4 - 4)) : 4 * c4;
    goff[k] = ok ? (y * W + x) * Cin + coff_u : coff_u;
    vbits |= ok ? (1u << k) : 0u;
    woff[k] = hy * ROWB + hx * KPIX + 8 * c4;
  }
  auto load_chunk = [&](int c, float4 (&ra)[LP]) {
#pragma unroll
    for (int k = 0; k < LP; ++k) ra[k] = *reinterpret_cast<const float4*>(inb + goff[k] + (HALVES ? 16 : CK) * c);
  };
  float sa = 1.0f;
  auto store_unit = [&](int buf, const float4 (&ra)[LP], int k) {
    const unsigned msk = 0u - ((vbits >> k) & 1u);
    const float x[4] = {ra[k].x, ra[k].y, ra[k].z, ra[k].w};
    ch_h4 hh, ll;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = __uint_as_float(__float_as_uint(x[e]) & msk) * sa;
      const _Float16 h = (_Float16)v;
      hh[e] = h;
      ll[e] = (_Float16)(v - (float)h);
    }
    *reinterpret_cast<ch_h4*>(&lds[buf * BUF + woff[k]]) = hh;
    *reinterpret_cast<ch_h4*>(&lds[buf * BUF + woff[k] + CK * 2]) = ll;
  };

  // the first halo is requested before anything else (loads return in order: nothing may queue in front of it)
  float4 ra0[LP];
  load_chunk(0, ra0);

  const float* meta = reinterpret_cast<const float*>(P.wimg + (size_t)Cin * 9 * Cout * 4);
  float amax_lane = P.in_amax[(size_t)b * P.amax_stride + lane];
  const float inv_sw = meta[n0 + (lane & 31)];  // per output channel (column): the pack scales every column to [2^13, 2^14)

  // ---- A rows of this lane: hardware row i of block bg is logical row sigma(i) -> tile row 32 bg + sigma(i) ->
  // window (row-major over the patch) and position in it -> top-left tap of that pixel in the halo -----------------
  int arow[MB];
  {
    const int Lr = ch2::sigma(lane & 31);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int m = 32 * (wm * MB + mb) + Lr;
      const int w = m >> 2, e = m & 3;
      const int wy = w / WPR, wx = w - wy * WPR;
      arow[mb] = (2 * wy + (e >> 1)) * ROWB + (2 * wx + (e & 1)) * KPIX + (16 * wk + 8 * g) * 2;
    }
  }

  // ---- this wave's weight stream: k16 block WK c + wk in chunk c, nine contiguous 2-KiB pairs per chunk ---------
  const unsigned char* wp = P.wimg + ((size_t)((n0 >> 5) * (Cin >> 4) + (HALVES ? wk * (Cin >> 5) : wk)) * 9) * 2048 + lane * 16;
  constexpr size_t kChunkStride = (size_t)(HALVES ? 1 : WK) * 9 * 2048;
  ch_h8 qh[D], ql[D];
#pragma unroll
  for (int t = 0; t < D; ++t) {
    qh[t] = *reinterpret_cast<const ch_h8*>(wp + (size_t)t * 2048);
    ql[t] = *reinterpret_cast<const ch_h8*>(wp + (size_t)t * 2048 + 1024);
  }

  ch_f16v acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
  // SEG > 0: the accumulators restart every SEG chunks (chains of 27 SEG MFMAs) and the finished segment is added to
  // a second register set: out = ((s0 + s1) + s2) + ... in fp32 VALU adds
  // (tot as register PAIRS, never an MFMA operand: no 16-register tuples to keep aligned; one v_pk_add_f32 per pair)
  typedef float f2v __attribute__((ext_vector_type(2)));
  f2v tot[SEG > 0 ? MB : 1][8];
  ch_f16v zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
  if (SEG > 0) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 8; ++r) tot[mb][r] = f2v{0.f, 0.f};
  }
  // (the finished segment sits in accumulation registers -- the 256 arch VGPRs are taken by tot, the A ring and the loader --
  // and VALU cannot read those: two v_accvgpr_read + one v_pk_add_f32 per register pair, written out so that the compiler
  // keeps tot in arch VGPRs instead of shuttling it through the accumulation file around every add)
  bool park_now = false;   // PARK: wave-uniform, true in the first chunk of the second K half
  auto park_at = [&](int mb, int r) __attribute__((always_inline)) { return 2 * BUF + (((wave * MB + mb) * 8 + r) * 64 + lane) * 8; };
  auto flush = [&](int mb) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      f2v t;
      asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t[0]) : "a"(acc[mb][2 * r]));
      asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t[1]) : "a"(acc[mb][2 * r + 1]));
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(tot[mb][r]) : "v"(t));
    }
    if (PARK && park_now) {   // p0 = the first half's total: parked, the second half starts from zero
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        *reinterpret_cast<f2v*>(&lds[park_at(mb, r)]) = tot[mb][r];
        tot[mb][r] = f2v{0.f, 0.f};
      }
    }
  };

#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) amax_lane = fmaxf(amax_lane, __shfl_xor(amax_lane, off));
  sa = ch2::pow2_scale(amax_lane, 14);
  const float descale = (1.0f / sa) * inv_sw;

  CH2W_STAMP(2);
#pragma unroll
  for (int k = 0; k < LP; ++k) store_unit(0, ra0, k);
  __syncthreads();
  CH2W_STAMP(3);

  // ---- one chunk: 9 taps x MB blocks = 9 MB sub-steps (tap t = s / MB, block s % MB) of three MFMAs from LDS buffer
  // c & 1, issued as PAIRS of sub-steps: l0 l1 | h0 h1 | h0 h1 -- consecutive MFMAs never wait for each other's
  // accumulator (what a wave alone on its SIMD cannot hide).  The A fragments of pair p + PD are read while the
  // MFMAs of pair p run (ring of 2 (PD + 1) register pairs); a tap's weight pair is replaced by the one D taps ahead
  // as soon as its last sub-step is issued; MORE: the next chunk's halo is requested at the top and split into the
  // other buffer one unit at a time between the MFMAs of the pairs from tap T0 on -----------------------------------
  constexpr int S = 9 * MB;
  constexpr int NP = (S + 1) / 2;                // pairs of sub-steps
  constexpr int PD = OCC * NWAVES >= 8 ? 1 : 2;  // pairs the A reads run ahead (two with one wave per SIMD: 512 registers)
  constexpr int NB = 2 * (PD + 1);
  constexpr int T0 = 4;
  constexpr int P0 = (T0 * MB + 1) / 2;          // first pair that may carry a unit of the next halo
  constexpr int SLOTS = NP - P0;
  auto chunk = [&](int c, auto more_c, auto first_c) __attribute__((always_inline)) {
    constexpr bool MORE = decltype(more_c)::value;
    constexpr bool FIRST = decltype(first_c)::value;   // first chunk of a segment (SEG > 0): see the flush below
    float4 ra[LP];
    if (MORE) load_chunk(c + 1, ra);
    int ab[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) ab[mb] = arow[mb] + (c & 1) * BUF;
    const unsigned char* wcur = wp + (size_t)c * kChunkStride;
    const unsigned char* wnxt = wcur + kChunkStride;   // (not dereferenced behind the last chunk)
    ch_h8 ah[NB], al[NB];
    auto rd = [&](int s) {
      const int t = s / MB, mb = s % MB;
      const int off = (t / 3) * ROWB + (t % 3) * KPIX;
      ah[s % NB] = *reinterpret_cast<const ch_h8*>(&lds[ab[mb] + off]);
      al[s % NB] = *reinterpret_cast<const ch_h8*>(&lds[ab[mb] + off + CK * 2]);
    };
#pragma unroll
    for (int s = 0; s < 2 * PD; ++s) rd(s);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int s0 = 2 * p, s1 = 2 * p + 1;
      const int t0 = s0 / MB, m0 = s0 % MB, t1 = s1 / MB, m1 = s1 % MB;
      if (s0 + 2 * PD < S) rd(s0 + 2 * PD);
      if (s1 + 2 * PD < S) rd(s1 + 2 * PD);
      __builtin_amdgcn_sched_barrier(0);  // the reads above are ISSUED here, not sunk next to their uses
      // segment start: block m's finished segment goes to tot[m] right before the block's first MFMA of the new
      // segment, which starts from C = 0 -- the adds of one block run under the MFMAs of the others
      const bool f0 = FIRST && t0 == 0, f1 = FIRST && t1 == 0 && s1 < S;
      if (f0) flush(m0);
      if (f1) flush(m1);
      if (s1 < S) {
        acc[m0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s0 % NB], qh[t0 % D], f0 ? zero16 : acc[m0], 0, 0, 0);
        acc[m1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s1 % NB], qh[t1 % D], f1 ? zero16 : acc[m1], 0, 0, 0);
        acc[m0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s0 % NB], ql[t0 % D], acc[m0], 0, 0, 0);
        acc[m1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s1 % NB], ql[t1 % D], acc[m1], 0, 0, 0);
        acc[m0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s0 % NB], qh[t0 % D], acc[m0], 0, 0, 0);
        acc[m1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s1 % NB], qh[t1 % D], acc[m1], 0, 0, 0);
      } else {
        acc[m0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s0 % NB], qh[t0 % D], f0 ? zero16 : acc[m0], 0, 0, 0);
        acc[m0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s0 % NB], ql[t0 % D], acc[m0], 0, 0, 0);
        acc[m0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s0 % NB], qh[t0 % D], acc[m0], 0, 0, 0);
      }
      bool any = false;
      if (MORE && p >= P0) {  // unit k rides on pair P0 + k SLOTS / LP (LP <= SLOTS)
#pragma unroll
        for (int k = 0; k < LP; ++k)
          if ((k * SLOTS) / LP == p - P0) { store_unit((c + 1) & 1, ra, k); any = true; }
      }
      if (any) {
#pragma unroll
        for (int m6 = 0; m6 < (s1 < S ? 6 : 3); ++m6) {  // one MFMA, then a share of the split's VALU instructions
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // the taps whose last sub-step was just issued hand their queue slot to the pair D taps ahead
#pragma unroll
      for (int t = t0; t <= (s1 < S ? t1 : t0); ++t) {
        const int last = (t + 1) * MB - 1;
        if ((last == s0 || (last == s1 && s1 < S)) && (MORE || t + D < 9)) {
          const unsigned char* wa = (t + D < 9 ? wcur : wnxt) + (size_t)((t + D) % 9) * 2048;
          qh[t % D] = *reinterpret_cast<const ch_h8*>(wa);
          ql[t % D] = *reinterpret_cast<const ch_h8*>(wa + 1024);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };
  static_assert(LP <= SLOTS, "halo units per thread");
  if constexpr (SEG == 0) {
#pragma unroll 1
    for (int c = 0; c + 1 < NC; ++c) {
      chunk(c, std::true_type{}, std::false_type{});
      if (c < 8) { CH2W_STAMP(4 + c); }
    }
    chunk(NC - 1, std::false_type{}, std::false_type{});
  } else {
    // segments of SEG chunks (SEG = 1, 2, 4; NC % SEG == 0: the launcher's business); the very first "flush" adds
    // zeros to zeros
    static_assert(SEG == 1 || SEG == 2 || SEG == 4, "chunks per segment");
    constexpr std::true_type T{};
    constexpr std::false_type F{};
#pragma unroll 1
    for (int c = 0; c + SEG < NC; c += SEG) {
      if (PARK) park_now = 2 * c == NC;
      chunk(c, T, T);
      if constexpr (SEG >= 2) chunk(c + 1, T, F);
      if constexpr (SEG >= 4) { chunk(c + 2, T, F); chunk(c + 3, T, F); }
      if (c / SEG < 8) { CH2W_STAMP(4 + c / SEG); }
    }
    if constexpr (SEG == 1) chunk(NC - 1, F, T);
    if (PARK) park_now = false;   // (NC >= 8: the second half starts inside the loop above)
    if constexpr (SEG == 2) { chunk(NC - 2, T, T); chunk(NC - 1, F, F); }
    if constexpr (SEG == 4) { chunk(NC - 4, T, T); chunk(NC - 3, T, F); chunk(NC - 2, T, F); chunk(NC - 1, F, F); }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        f2v t = tot[mb][r] + f2v{acc[mb][2 * r], acc[mb][2 * r + 1]};
        if (PARK) t = *reinterpret_cast<const f2v*>(&lds[park_at(mb, r)]) + t;   // p0 + p1
        acc[mb][2 * r] = t[0];
        acc[mb][2 * r + 1] = t[1];
      }
  }
  CH2W_STAMP(12);

  // ---- two k-waves: p0 + p1 through LDS.  k-wave 0 finishes blocks 0 .. MBH - 1, k-wave 1 the others: each hands
  // over the blocks it does not finish ----------------------------------------------------------------------------
  constexpr int MBH = WK == 2 ? (MB + 1) / 2 : MB;
  const int mb_lo = WK == 2 && wk == 1 ? MBH : 0;
  const int mb_hi = WK == 2 && wk == 0 ? MBH : MB;
  if (WK == 2) {
    float* xch = reinterpret_cast<float*>(lds);
    auto xaddr = [&](int mb, int r) { return ((((wm * NWV + wn) * MB + mb) * 16 + r) * 64 + lane); };
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const bool mine = mb >= mb_lo && mb < mb_hi;
      if (!mine) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[xaddr(mb, r)] = acc[mb][r];
      }
    }
    __syncthreads();
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const bool mine = mb >= mb_lo && mb < mb_hi;
      if (mine) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float o = xch[xaddr(mb, r)];
          acc[mb][r] = wk == 0 ? acc[mb][r] + o : o + acc[mb][r];   // p0 + p1
        }
      }
    }
  }
  CH2W_STAMP(13);

  // ---- epilogue: bias, ReLU, fp32 NHWC store, 2x2 max pool of the lane's window, maximum of |out| for the next
  // layer's scale.  Round 4: the C layout hands lane (j, g) ONE channel of 16 pixels -- sixteen 4-byte stores per block,
  // and a store instruction costs a wave alone ~60 cycles of in-order issue (the epilogue was ~9 k cycles of a 134 k-cycle
  // tile).  Each wave now turns its block through a private LDS slice (16 ds_write_b32, 4 ds_read_b128; rows of 36 floats:
  // conflict-free both ways) so that lane (j', g) holds FOUR consecutive channels 4 (j' & 7) .. of the four pixels of the
  // ONE window q' = j' >> 3 of its half: four 16-byte stores (8 lanes = the 128 contiguous bytes of a pixel, 8 pixels per
  // instruction) + one for the pooled window instead of twenty 4-byte ones.  The values are the same: bits unchanged.
  if (WK == 2) __syncthreads();   // every wave is done with the exchange area: the slices below may overlap it
  const float bias_j = P.bias[n0 + j];
  const int WC = W * Cout;
  float* o00 = P.out + ((size_t)b * H * W + (size_t)y0 * W + x0) * Cout + n0;
  float* o01 = o00 + Cout;
  float* o10 = o00 + WC;
  float* o11 = o10 + Cout;
  const int Wp = W >> 1;
  float* pb = P.pool_out ? P.pool_out + ((size_t)b * (H >> 1) * Wp + (size_t)(y0 >> 1) * Wp + (x0 >> 1)) * Cout + n0 : nullptr;
  const bool full = y0 + TH <= H && x0 + TW <= W;
  float vmax = 0.f;
  constexpr int TROW = 36;                                  // floats per transposition row (16-byte aligned, 4 row mod 32 banks)
  static_assert(NWAVES * 32 * TROW * 4 <= LDS_BYTES, "transposition slices");
  float* tr = reinterpret_cast<float*>(lds) + wave * (32 * TROW);
  const int qn = j >> 3, c4 = 4 * (j & 7);                  // this lane's window (quad) and channel group after the turn
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    if (mb < mb_lo || mb >= mb_hi) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float t = fmaf(acc[mb][r], descale, bias_j);
      t = P.relu ? fmaxf(t, 0.f) : t;
      tr[(16 * g + r) * TROW + j] = t;                      // pixel 4 q + e of half g, channel j
    }
    __builtin_amdgcn_wave_barrier();   // (a wave's LDS operations execute in order; this only pins the compiler's order)
    // window of (block mb, quad qn, half g): logical rows 32 (wm MB + mb) + quad_row(qn, g) .. + 3
    const int w = (32 * (wm * MB + mb) + ch2::sigma(8 * qn + 4 * g)) >> 2;
    const int wy = w / WPR, wx = w - wy * WPR;
    const int off = 2 * wy * WC + 2 * wx * Cout + c4;
    float4 v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = *reinterpret_cast<const float4*>(&tr[(16 * g + 4 * qn + e) * TROW + c4]);
    __builtin_amdgcn_wave_barrier();
    const float4 m4 = make_float4(fmaxf(fmaxf(v[0].x, v[1].x), fmaxf(v[2].x, v[3].x)), fmaxf(fmaxf(v[0].y, v[1].y), fmaxf(v[2].y, v[3].y)),
                                  fmaxf(fmaxf(v[0].z, v[1].z), fmaxf(v[2].z, v[3].z)), fmaxf(fmaxf(v[0].w, v[1].w), fmaxf(v[2].w, v[3].w)));
    auto amax4 = [](const float4& a) { return fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))); };
    if (full) {
      *reinterpret_cast<float4*>(o00 + off) = v[0];
      *reinterpret_cast<float4*>(o01 + off) = v[1];
      *reinterpret_cast<float4*>(o10 + off) = v[2];
      *reinterpret_cast<float4*>(o11 + off) = v[3];
      vmax = fmaxf(vmax, fmaxf(fmaxf(amax4(v[0]), amax4(v[1])), fmaxf(amax4(v[2]), amax4(v[3]))));
      if (pb) *reinterpret_cast<float4*>(pb + (wy * Wp + wx) * Cout + c4) = m4;
    } else {
      const int y = y0 + 2 * wy, x = x0 + 2 * wx;
      if (y < H && x < W) { *reinterpret_cast<float4*>(o00 + off) = v[0]; vmax = fmaxf(vmax, amax4(v[0])); }
      if (y < H && x + 1 < W) { *reinterpret_cast<float4*>(o01 + off) = v[1]; vmax = fmaxf(vmax, amax4(v[1])); }
      if (y + 1 < H && x < W) { *reinterpret_cast<float4*>(o10 + off) = v[2]; vmax = fmaxf(vmax, amax4(v[2])); }
      if (y + 1 < H && x + 1 < W) { *reinterpret_cast<float4*>(o11 + off) = v[3]; vmax = fmaxf(vmax, amax4(v[3])); }
      if (pb && y + 1 < H && x + 1 < W) *reinterpret_cast<float4*>(pb + (wy * Wp + wx) * Cout + c4) = m4;
    }
  }
  if (P.out_amax) {  // 64 slots: same-address atomics serialise in L2 (~10 ns each)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off));
    if (lane == 0)
      atomicMax(reinterpret_cast<unsigned*>(P.out_amax) + (size_t)b * P.amax_stride + ((blockIdx.x * NWAVES + wave) & 63),
                __float_as_uint(vmax));
  }
  CH2W_STAMP(14);
}

template <int MB, int MWV, int NWV, int WK, int TH, int TW, int OCC, int SEG = 0, bool PARK = false>
static hipError_t conv_h2w_go(ConvH2Dev d, hipStream_t st) {
  if (SEG > 0 && (d.Cin / (16 * WK)) % SEG != 0) return hipErrorInvalidValue;   // whole segments only
  if (PARK && (d.Cin % 64 != 0 || d.Cin < 128)) return hipErrorInvalidValue;    // two halves of whole segments, the second one starting inside the loop
  d.tiles_x = (d.W + TW - 1) / TW;
  d.tiles_y = (d.H + TH - 1) / TH;
  const int grid = d.B * d.tiles_x * d.tiles_y * (d.Cout / (32 * NWV));
  hipLaunchKernelGGL((conv_h2w_kernel<MB, MWV, NWV, WK, TH, TW, OCC, SEG, PARK>), dim3(grid), dim3(64 * MWV * NWV * WK), 0, st, d);
  return hipGetLastError();
}

// k-waves of the batched form, by LAYER SHAPE only (the summation order must not depend on the batch): one where
// M x N of a few images already fills the chip's 1024 SIMDs with 7-block waves, two from the 28-pixel layers on.
// (Until r04j the 56-pixel layers had two as well: at 12 and 16 images per call the one-k-wave form with two workgroups
// per CU is 6-8 % faster there -- 158 against 168 us for conv3_2 at 16 images, tools/conv_h2w_variants_time.py -- and
// equal at 4 and 8 with the 64-channel patch variant below.)
int conv_h2w_kwaves(int H, int W, int Cin, int Cout) {
  (void)Cin;
  return (long)H * W * Cout >= (long)56 * 56 * 256 ? 1 : 2;  // conv1_2, conv2_x, conv3_x: 1; conv4_x: 2
}

bool conv_h2w_supported(int H, int W, int Cin, int Cout) {
  return conv_h2_supported(H, W, Cin, Cout) && Cin % 32 == 0 && (long)H * W >= 28 * 28;
}

// variant: 0 = by shape (the inference path), -1 = round 3's selection by shape and batch (the training step); 1..5 force
// <4,2,2,1,8,32>, <7,1,4,1,8,28>, <7,1,2,1,8,28>, <7,1,4,2,8,28>, <7,1,2,2,8,28>, 6 the segmented <7,1,4,1,8,28,1,SEG=2>
// (tests: every shape through every variant; 1..3 have one k-wave, 4..5 two -- variants with the same number of
// k-waves give the same bits; 6 has its own: one k-wave in segments of two chunks)
hipError_t conv_h2w_launch(ConvH2Dev d, hipStream_t st, int variant) {
  // variant 0 (the inference path): layers whose one-k-wave chain would exceed 108 MFMAs per accumulator (Cin >= 128)
  // take the SEGMENTED form -- chains of 54, the segments summed in fp32 -- whatever the batch; conv1_2 / conv2_1
  // (Cin = 64: chains of 108 as they are) the round-3 variants below.  variant -1 (the training step): round 3's
  // selection for every layer (chains of up to 432: faster at 4 .. 11 images per call, 1.7 x the error)
  if (variant == 0 && d.Cin >= 128 && d.Cout % 128 == 0) {
    // two K halves (lower / upper k16 blocks), each in segments of two chunks, p0 + p1: as two k-waves over 64-channel
    // workgroups (variant 7) while that is what fills the chip, as ONE k-wave that parks p0 in LDS (variant 6: four n-waves
    // per halo, half the loader work per MFMA) from ~200 128-channel workgroups on -- the same bits
    const long wg128 = (long)d.B * ((d.H + 7) / 8) * ((d.W + 27) / 28) * (d.Cout / 128);
    variant = wg128 >= 200 ? 6 : 7;
  }
  if (variant == 0 || variant == -1) {
    const int wk = conv_h2w_kwaves(d.H, d.W, d.Cin, d.Cout);
    if (wk == 1) {
      // 128-channel workgroups (two per CU) while they give >= ~200 workgroups, else the 64-channel 8 x 32 patches
      const long wg128 = (long)d.B * ((d.H + 7) / 8) * ((d.W + 27) / 28) * (d.Cout / 128);
      if (d.Cout % 128 == 0 && wg128 >= 200) variant = 2;
      else variant = d.W % 32 == 0 || d.Cout % 128 == 0 ? 1 : 3;
    } else {
      // 128-channel workgroups of eight waves (two per SIMD) while they still give >= ~200 workgroups, else 64-channel ones
      const long wg128 = (long)d.B * ((d.H + 7) / 8) * ((d.W + 27) / 28) * (d.Cout / 128);
      variant = d.Cout % 128 == 0 && wg128 >= 200 ? 4 : 5;
    }
  }
  switch (variant) {
    case 1: return conv_h2w_go<4, 2, 2, 1, 8, 32, 2>(d, st);
    case 2: return conv_h2w_go<7, 1, 4, 1, 8, 28, 2>(d, st);
    case 3: return conv_h2w_go<7, 1, 2, 1, 8, 28, 2>(d, st);
    case 4: return conv_h2w_go<7, 1, 4, 2, 8, 28, 1>(d, st);
    case 5: return conv_h2w_go<7, 1, 2, 2, 8, 28, 1>(d, st);
    case 6: return conv_h2w_go<7, 1, 4, 1, 8, 28, 1, 2, true>(d, st);
    case 7: return conv_h2w_go<7, 1, 2, 2, 8, 28, 1, 2>(d, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace disn
