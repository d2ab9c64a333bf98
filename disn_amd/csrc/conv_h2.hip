// 3x3 SAME convolution of the VGG-16 stack (models/CNN/vgg.py:187-196, 'conv1_2' .. 'conv5_3'; called from
// models/model_normalization.py:74-76) for a SINGLE image (or a few): fp32 in, fp32 out, fp32-accurate
// products on the f16 matrix pipes, no split-K pass and no im2col re-reads.
//
// Why a second convolution kernel.  At B = 1 a VGG layer is a skinny GEMM (M = 196 .. 50176 pixels,
// N = 64 .. 512, K = 576 .. 4608).  The implicit-GEMM kernel of gemm_bf16_mfma.hip cuts it into 64x64
// tiles x 3..11 K-splits of short-lived workgroups plus a reduce launch, re-reads every input pixel nine
// times (im2col) and every weight once per WAVE ROW; measured 23-45 us per layer + 5-7 us per reduce,
// 0.47 of the f32 MFMA peak (profiles/r02b_*).  Here:
//
//  * Two-term f16 split (as mlp_fused.hip): x = h + l, h = f16(x s), l = f16(x s - h), s a power of two
//    that puts the tensor's largest magnitude at 2^14 (activations: from the producer's atomic max;
//    weights: at pack time, 2^13).  a b is accumulated in fp32 from l_a h_b + h_a l_b + h_a h_b: three
//    v_mfma_f32_32x32x16_f16 per 32x32x16 block instead of six bf16 ones, 4 instead of 6 bytes per weight.
//  * The M tile is a 2-D patch of the image (TH x TW pixels, TW a multiple of the lane-group size, see
//    below); its (TH+2) x (TW+2) halo of 64 input channels is split ONCE into h / l planes in LDS and all
//    nine taps read it there at shifted addresses: each input element is fetched ~1.3-2x instead of 9x and
//    split once instead of nine times.
//  * All K parallelism is INSIDE the workgroup: the four k16 blocks of a 64-channel chunk belong to four
//    waves (wk = 0..3), each of which owns the whole BM x 32 output tile of its n-block and streams ITS
//    weight fragments straight from L2 into registers -- one contiguous 2 KiB per (chunk, tap), nine
//    (chunk, tap) steps ahead.  No operand is fetched twice by a workgroup, nothing but the halo goes
//    through LDS, and the four partial tiles are summed through LDS in a fixed order at the end
//    ((w0 + w2) + (w1 + w3)): deterministic, no partials in HBM, no second launch.
//  * Tiles are small (32..128 pixels x 32..64 channels) so that a single image still gives 112..392
//    workgroups; n-tile-major XCD placement keeps the workgroups that stream the same weights on one L2.
//  * Epilogue: bias, ReLU, fp32 NHWC store, optional 2x2 max pool of the same tile (the five pooled
//    layers are the five taps: both are needed), and the atomic max |out| that gives the NEXT layer its
//    activation scale.
//
// LDS layout of a halo pixel: 128 B of h (64 channels), 128 B of l, 16 B pad -> 272 B = 17 x 16 B, so 16
// CONSECUTIVE pixels land on 16 different 16-byte slots: ds_read_b128 is conflict-free when each of its
// 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31} per half wave; MI355X_MICROARCH.md, LDS) reads 16
// consecutive pixels of one halo row.  The A-fragment row <-> pixel map is chosen for that: hardware row i
// of a 32-row block is LOGICAL row sigma(i) (group-1 lanes in ascending order are logical 0..15, group-2
// lanes 16..31); logical rows are `32 / SEG` runs of SEG consecutive pixels of one patch row (SEG = 16:
// two patch rows of <= 16 pixels; SEG = 32: one patch row of <= 32).  The C layout of the MFMA then gives
// lane (j, g) and register quad q the four consecutive logical rows L0(q, g) .. L0 + 3 of column j.
#include "kernels.hpp"
#include "tuning.hpp"
#include "h2_common.hpp"

#include <type_traits>

namespace disn {


// ---------------------------------------------------------------------------------------------------
// weight image: for n-block nb (32 output channels), k16 block kb (input channels 16 kb .. 16 kb + 15), tap t:
//   two 1-KiB planes (h, l), lane (j, g) holds W[t][16 kb + 8 g + e][32 nb + j] * s_w[32 nb + j], e = 0..7
// at byte ((((nb * (Cin / 16) + kb) * 9 + t) * 2 + plane) * 64 + lane) * 16: the nine taps of one k16 block are one
// contiguous 18 KiB -- what one k-wave reads per chunk, whatever the number of k-waves.
// Scales are PER OUTPUT CHANNEL (round 4): s_w[c] = the power of two that puts max |W[:, :, c]| into [2^13, 2^14).  A
// per-tensor scale leaves an entry 2^-19 below the tensor's maximum with five significant bits (f16's subnormal floor
// is 2^-24 absolute), which is where trained weights with outlier output channels put the columns of the ordinary
// channels (tools/split_model.py: 2e-3 on pred against 2e-6).  Behind the image (the tail): inv_sw[Cout] = 1 / s_w[c]
// -- what a lane multiplies its column's accumulators with, next to the bias -- and Cout floats of scratch for the
// column-maximum pass.
// ---------------------------------------------------------------------------------------------------
// flip_t: w is the FORWARD tensor [taps][Cout][Cin] of the layer whose data gradient this image serves -- the image is
// the one of w'[t][ci][co] = w[taps - 1 - t][co][ci] (taps mirrored, channels transposed: dx = conv(dz, w'))

// column maxima of one tensor, viewed as [outer][mid][inner] floats (TF HWIO: taps, Cin_fwd, Cout_fwd): max |w| per INNER
// index -> cmax_inner (the columns of the forward image) and per MID index -> cmax_mid (the columns of the flipped
// image); either may be null.  A workgroup reduces 4096 consecutive floats in LDS tables, then one global atomic per
// touched column (non-negative floats: integer maximum of the bit patterns; the tables must be zeroed first).
__device__ __forceinline__ void h2_colmax_chunk(const float* __restrict__ p, long base, int mid, int inner,
                                                float* __restrict__ cmax_inner, float* __restrict__ cmax_mid,
                                                unsigned* tab_i, unsigned* tab_m) {
  // more than 1024 columns (a 4096-wide dense layer): no LDS table for that index, one global atomic per
  // element instead -- pack time only
  const bool big_i = inner > 1024, big_m = mid > 1024;
  for (int i = threadIdx.x; i < 1024; i += 256) { tab_i[i] = 0u; tab_m[i] = 0u; }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long e = base + (long)(k * 256 + threadIdx.x) * 4;
    const float4 v = *reinterpret_cast<const float4*>(p + e);
    const float a[4] = {fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w)};
    if (cmax_inner) {
      const int c0 = (int)(e % inner);  // inner % 4 == 0: the four elements are four consecutive columns
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (big_i) atomicMax(reinterpret_cast<unsigned*>(cmax_inner) + c0 + q, __float_as_uint(a[q]));
        else atomicMax(&tab_i[c0 + q], __float_as_uint(a[q]));
      }
    }
    if (cmax_mid) {
      const unsigned m4 = __float_as_uint(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])));
      const int r = (int)((e / inner) % mid);
      if (big_m) atomicMax(reinterpret_cast<unsigned*>(cmax_mid) + r, m4);
      else atomicMax(&tab_m[r], m4);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 256) {
    if (cmax_inner && !big_i && i < inner && tab_i[i]) atomicMax(reinterpret_cast<unsigned*>(cmax_inner) + i, tab_i[i]);
    if (cmax_mid && !big_m && i < mid && tab_m[i]) atomicMax(reinterpret_cast<unsigned*>(cmax_mid) + i, tab_m[i]);
  }
}

__global__ __launch_bounds__(256) void h2_colmax_kernel(const float* __restrict__ w, int mid, int inner,
                                                        float* __restrict__ cmax_inner, float* __restrict__ cmax_mid) {
  __shared__ unsigned tab_i[1024], tab_m[1024];
  h2_colmax_chunk(w, (long)blockIdx.x * 4096, mid, inner, cmax_inner, cmax_mid, tab_i, tab_m);
}

__device__ __forceinline__ float* h2_tail(unsigned char* image, int Cin, int Cout, int taps) {
  return reinterpret_cast<float*>(image + (size_t)Cin * taps * Cout * 4);  // inv_sw[Cout], then cmax[Cout]
}

__global__ __launch_bounds__(256) void conv_h2_pack_kernel(const float* __restrict__ w, int Cin, int Cout, int taps,
                                                           unsigned char* __restrict__ image, int flip_t) {
  const int KB = Cin >> 4;
  const size_t frags = (size_t)(Cout >> 5) * KB * taps;
  float* tail = h2_tail(image, Cin, Cout, taps);
  const float* cmax = tail + Cout;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < frags * 64; idx += (size_t)gridDim.x * 256) {
    const int lane = (int)(idx & 63);
    size_t f = idx >> 6;
    int t = (int)(f % taps);
    int kb = (int)((f / taps) % KB);
    const int nb = (int)(f / ((size_t)taps * KB));
    if (flip_t) {  // walk the k16 blocks fastest: a lane's reads of consecutive fragments are consecutive 32-byte pieces of
                   // one row of the forward tensor (whole cache lines per workgroup pass instead of a quarter of each)
      kb = (int)(f % KB);
      t = (int)((f / KB) % taps);
      f = ((size_t)nb * KB + kb) * taps + t;
    }
    const int j = lane & 31, g = lane >> 5;
    const float s = ch2::pow2_scale(cmax[32 * nb + j], 13);
    if (kb == 0 && t == 0 && g == 0) tail[32 * nb + j] = 1.0f / s;
    ch_h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ci = 16 * kb + 8 * g + e;
      const float v = (flip_t ? w[((size_t)(taps - 1 - t) * Cout + 32 * nb + j) * Cin + ci]
                              : w[((size_t)t * Cin + ci) * Cout + 32 * nb + j]) * s;
      const _Float16 h = (_Float16)v;
      hi[e] = h;
      lo[e] = (_Float16)(v - (float)h);
    }
    ch_h8* out = reinterpret_cast<ch_h8*>(image);
    out[(f * 2) * 64 + lane] = hi;
    out[(f * 2 + 1) * 64 + lane] = lo;
  }
}

size_t h2_image_bytes(int K, int N, int taps) { return (size_t)K * taps * N * 4 + (size_t)N * 8 + 256; }
size_t conv_h2_image_bytes(int Cin, int Cout) { return h2_image_bytes(Cin, Cout, 9); }

// ---- the same for a list of 3x3 tensors: one clearing pass, one maximum pass, one pack pass (train.hip) -----------
__global__ __launch_bounds__(256) void conv_h2_tail_clear_multi_kernel(const ConvH2PackJobs jobs) {
  const ConvH2PackJob& J = jobs.j[blockIdx.x];
  float* cmax = h2_tail(J.image, J.Cin, J.Cout, 9) + J.Cout;
  for (int i = threadIdx.x; i < J.Cout; i += 256) cmax[i] = 0.f;
}

__global__ __launch_bounds__(256) void conv_h2_colmax_multi_kernel(const ConvH2PackJobs jobs) {
  // a workgroup reduces 4096 consecutive floats of ONE tensor (tensor sizes are multiples of 36864 = 9 * 4096) for the
  // forward image (columns = inner index) and the flipped image (columns = mid index) of its slot
  __shared__ unsigned tab_i[1024], tab_m[1024];
  const long base = (long)blockIdx.x * 4096;
  int s = 0;
  while (s + 1 < jobs.nslots && jobs.seg_begin[s + 1] <= base) ++s;
  h2_colmax_chunk(jobs.seg[s], base - jobs.seg_begin[s], jobs.seg_mid[s], jobs.seg_inner[s], jobs.cmax_fwd[s],
                  jobs.cmax_flip[s], tab_i, tab_m);
}

__global__ __launch_bounds__(256) void conv_h2_pack_multi_kernel(const ConvH2PackJobs jobs) {
  // four fragments per workgroup; fragment counts of every job are multiples of four (72 at 64 x 64 channels)
  const long f0 = (long)blockIdx.x * 4;
  int ji = 0;
  {
    int lo = 0, hi = jobs.n - 1;  // last job with frag_begin <= f0
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs.j[mid].frag_begin <= f0) lo = mid; else hi = mid - 1;
    }
    ji = lo;
  }
  const ConvH2PackJob& J = jobs.j[ji];
  const int Cin = J.Cin, Cout = J.Cout, KB = Cin >> 4, taps = 9;
  float* tail = h2_tail(J.image, Cin, Cout, taps);
  const float* cmax = tail + Cout;
  size_t f = (size_t)(f0 - J.frag_begin) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  int t = (int)(f % taps);
  int kb = (int)((f / taps) % KB);
  const int nb = (int)(f / ((size_t)taps * KB));
  if (J.flip_t) {  // as conv_h2_pack_kernel: k16 blocks fastest
    kb = (int)(f % KB);
    t = (int)((f / KB) % taps);
    f = ((size_t)nb * KB + kb) * taps + t;
  }
  const int j = lane & 31, g = lane >> 5;
  const float s = ch2::pow2_scale(cmax[32 * nb + j], 13);
  if (kb == 0 && t == 0 && g == 0) tail[32 * nb + j] = 1.0f / s;
  ch_h8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ci = 16 * kb + 8 * g + e;
    const float v = (J.flip_t ? J.w[((size_t)(taps - 1 - t) * Cout + 32 * nb + j) * Cin + ci]
                              : J.w[((size_t)t * Cin + ci) * Cout + 32 * nb + j]) * s;
    const _Float16 h = (_Float16)v;
    hi[e] = h;
    lo[e] = (_Float16)(v - (float)h);
  }
  ch_h8* out = reinterpret_cast<ch_h8*>(J.image);
  out[(f * 2) * 64 + lane] = hi;
  out[(f * 2 + 1) * 64 + lane] = lo;
}

void conv_h2_pack_job_add(ConvH2PackJobs& jobs, const float* w_fwd, int Cin_fwd, int Cout_fwd, void* image, int flip_t,
                          int slot) {
  ConvH2PackJob& J = jobs.j[jobs.n++];
  J.w = w_fwd; J.image = static_cast<unsigned char*>(image); J.flip_t = flip_t; J.slot = slot;
  J.Cin = flip_t ? Cout_fwd : Cin_fwd;
  J.Cout = flip_t ? Cin_fwd : Cout_fwd;
  J.frag_begin = jobs.total_frags;
  jobs.total_frags += (long)(J.Cout >> 5) * (J.Cin >> 4) * 9;
  if (slot >= jobs.nslots) {   // slots are added in order (0, 1, ...), one tensor each; seg_begin[0] == 0
    jobs.seg[slot] = w_fwd;
    jobs.seg_mid[slot] = Cin_fwd;
    jobs.seg_inner[slot] = Cout_fwd;
    jobs.cmax_fwd[slot] = nullptr;
    jobs.cmax_flip[slot] = nullptr;
    jobs.seg_begin[slot + 1] = jobs.seg_begin[slot] + (long)9 * Cin_fwd * Cout_fwd;
    jobs.nslots = slot + 1;
  }
  // the scratch half of the image's tail collects its column maxima (one image per slot and direction)
  float* cmax = reinterpret_cast<float*>(J.image + (size_t)J.Cin * 9 * J.Cout * 4) + J.Cout;
  if (flip_t) jobs.cmax_flip[slot] = cmax; else jobs.cmax_fwd[slot] = cmax;
}

#ifndef CH2_UBENCH
hipError_t conv_h2_pack_multi_launch(const ConvH2PackJobs& jobs, hipStream_t st) {
  if (jobs.n == 0) return hipSuccess;
  hipLaunchKernelGGL(conv_h2_tail_clear_multi_kernel, dim3((unsigned)jobs.n), dim3(256), 0, st, jobs);
  hipLaunchKernelGGL(conv_h2_colmax_multi_kernel, dim3((unsigned)(jobs.seg_begin[jobs.nslots] / 4096)), dim3(256), 0, st, jobs);
  hipLaunchKernelGGL(conv_h2_pack_multi_kernel, dim3((unsigned)(jobs.total_frags / 4)), dim3(256), 0, st, jobs);
  return hipGetLastError();
}
#endif

#ifndef CH2_UBENCH
// w: TF HWIO [taps][Cin][Cout] (flip_t: the forward tensor [taps][Cout][Cin] of the image's layer); the column maxima
// are collected in the image's own tail (`scratch` is no longer used)
hipError_t conv_h2_pack_launch(const float* w, int Cin, int Cout, void* image, float* scratch, hipStream_t st,
                               int taps, int flip_t) {
  (void)scratch;
  if (((size_t)taps * Cin * Cout) % 4096) return hipErrorInvalidValue;   // (Cin, Cout multiples of 64: always true)
  unsigned char* img = static_cast<unsigned char*>(image);
  float* cmax = reinterpret_cast<float*>(img + (size_t)Cin * taps * Cout * 4) + Cout;
  hipError_t e = hipMemsetAsync(cmax, 0, (size_t)Cout * sizeof(float), st);
  if (e != hipSuccess) return e;
  // forward: columns = the inner index of [taps][Cin][Cout]; flipped: the tensor is [taps][Cout_img][Cin_img] and the
  // image's columns are its MID index
  const unsigned grid = (unsigned)(((size_t)taps * Cin * Cout) / 4096);
  if (flip_t) {
    if (Cin % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(h2_colmax_kernel, dim3(grid), dim3(256), 0, st, w, Cout, Cin, (float*)nullptr, cmax);
  } else {
    hipLaunchKernelGGL(h2_colmax_kernel, dim3(grid), dim3(256), 0, st, w, Cin, Cout, cmax, (float*)nullptr);
  }
  hipLaunchKernelGGL(conv_h2_pack_kernel, dim3(1024), dim3(256), 0, st, w, Cin, Cout, taps,
                     static_cast<unsigned char*>(image), flip_t);
  return hipGetLastError();
}

#endif  // CH2_UBENCH

// ---------------------------------------------------------------------------------------------------
// struct ConvH2Dev: h2_common.hpp (shared with conv_h2w.hip)

#ifdef DISN_TUNING
#define CH2_STAMP(i) \
  if (P.stamps && threadIdx.x == 0) P.stamps[(size_t)blockIdx.x * 16 + (i)] = (i) == 0 ? (long long)wall_clock64() : (long long)clock64()
#else
#define CH2_STAMP(i)
#endif

// WK k-waves per n-block (4 or 8): a chunk is CK = 16 WK input channels.  Eight k-waves put two waves on every
// SIMD of the CU: a global_load_dwordx4 blocks its wave's in-order issue for ~50 cycles (measured: a chunk with
// loads costs 1100-1400 cycles more than the same chunk without, r02d stamps), which only ANOTHER wave's MFMAs can
// cover.
// ABL (tools/ubench/conv_h2_ablate.hip only; 0 in the library): 1 no weight loads in the loop, 2 no halo loads,
// 4 no split + LDS store, 8 no A-fragment reads in the loop (wrong results: timing only)
// OCC: workgroups per CU the register budget is cut for (2: the batched calls' multi-round grids, tuning builds)
// FL > 0 (the 14 x 14 layers of a batched call): a k-wave's accumulators restart every FL chunks (chains of 27 FL MFMAs)
// and the finished segment is added to a second register set in fp32 VALU adds, as in conv_h2w.hip's SEG form
template <int MB, int NW, int SEG, int TW, int D, int WK, int ABL = 0, int OCC = 1, int FL = 0>
__global__ __launch_bounds__(64 * WK * NW, OCC == 1 ? 1 : OCC * WK * NW / 4) void conv_h2_kernel(const ConvH2Dev P) {
  // (HIP's second launch-bound is waves per SIMD, not workgroups per CU)
  constexpr int CK = 16 * WK;         // input channels per chunk
  constexpr int KPIX = CK * 4 + 16;   // bytes per halo pixel: h plane, l plane, pad -- an odd multiple of 16
  constexpr int UPP = CK / 4;         // float4 units per pixel
  constexpr int RPS = 32 / SEG;       // patch rows per 32-row block
  constexpr int TH = MB * RPS;        // patch rows
  constexpr int RP = TW + 2;          // halo row pitch in pixels
  constexpr int HP = (TH + 2) * RP;   // halo pixels
  constexpr int BUF = HP * KPIX;
  constexpr int NT = 64 * WK * NW;
  constexpr int LP = (HP * UPP + NT - 1) / NT;  // float4 units per thread and chunk
  constexpr int XCH = NW * WK * MB * 4096;      // exchange area of the final K reduction
  // reads of the rows beyond TW (never stored) run up to 34 - RP pixels past a buffer
  constexpr int LDS_BYTES = (2 * BUF + 8 * KPIX) > XCH ? (2 * BUF + 8 * KPIX) : XCH;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave % WK, wn = wave / WK;
  const int j = lane & 31, g = lane >> 5;
  CH2_STAMP(0);
  CH2_STAMP(1);

  // ---- tile: n-tile major, every XCD (hardware workgroup L runs on XCD L % 8) a contiguous eighth --------
  int l;
  {
    const int T = gridDim.x, L = blockIdx.x, q = T >> 3, r = T & 7, xcd = L & 7, idx = L >> 3;
    l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int per_img = P.tiles_y * P.tiles_x;
  int nt, b, mt;
  if (P.img_major) {   // an XCD's eighth = whole images (or a share of one): its L2 keeps the image's input
    const int per_b = (P.Cout / (32 * NW)) * per_img;
    b = l / per_b;
    mt = l - b * per_b;
    nt = mt / per_img;
    mt -= nt * per_img;
  } else {             // an XCD's eighth = whole n-tiles: its L2 keeps their weights
    const int mtiles = P.B * per_img;
    nt = l / mtiles;
    mt = l - nt * mtiles;
    b = mt / per_img;
    mt -= b * per_img;
  }
  const int tyi = mt / P.tiles_x, txi = mt - tyi * P.tiles_x;
  const int y0 = tyi * TH, x0 = txi * TW;
  const int n0 = (nt * NW + wn) * 32;
  const int H = P.H, W = P.W, Cin = P.Cin, Cout = P.Cout;
  const int NC = Cin / CK;
  const float* inb = P.in + (size_t)b * H * W * Cin;

  // ---- halo loader: unit u = (pixel, float4 of the chunk's 64 channels); 16 lanes = one pixel ----------------
  // every load is unconditional and every loaded value is used (an out-of-image unit reads a valid pixel and is
  // ANDed with 0): no exec-mask branches, so the compiler can count the loads in flight (s_waitcnt vmcnt(N))
  // REG (the 14-pixel patches with 64-channel chunks: 16 halo pixels x 16 units = the 256 threads): unit k of a thread is
  // halo ROW k of its (column, channel group) -- offsets and masks are arithmetic in k, three registers instead of 3 LP
  constexpr bool REG = FL > 0 && RP * UPP == NT;
  int goff[REG ? 1 : LP], woff[REG ? 1 : LP];
  unsigned gmask[REG ? 1 : LP];
  const int r_hx = tid / UPP, r_c4 = tid % UPP;
  const bool r_xok = x0 - 1 + r_hx >= 0 && x0 - 1 + r_hx < W;
  const int r_g0 = ((y0 - 1) * W + x0 - 1 + r_hx) * Cin + 4 * r_c4;
  if constexpr (!REG) {
#pragma unroll
    for (int k = 0; k < LP; ++k) {
      const int u = tid + k * NT;
      const int hp = u / UPP, c4 = u % UPP;
      const int hy = hp / RP, hx = hp - hy * RP;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool in_halo = u < HP * UPP;
      const bool ok = in_halo && y >= 0 && y < H && x >= 0 && x < W;
      goff[k] = ok ? (y * W + x) * Cin + 4 * c4 : 4 * c4;
      gmask[k] = ok ? 0xffffffffu : 0u;
      woff[k] = in_halo ? hp * KPIX + 8 * c4 : -1;
    }
  }
  auto unit_ok = [&](int k) __attribute__((always_inline)) { return r_xok && y0 - 1 + k >= 0 && y0 - 1 + k < H; };
  auto load_chunk = [&](int c, float4 (&ra)[LP]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < LP; ++k) {
      if constexpr (REG) ra[k] = *reinterpret_cast<const float4*>(inb + (unit_ok(k) ? r_g0 + k * W * Cin : 4 * r_c4) + CK * c);
      else ra[k] = *reinterpret_cast<const float4*>(inb + goff[k] + CK * c);
    }
  };
  float sa = 1.0f;
  auto store_unit = [&](int buf, const float4 (&ra)[LP], int k) __attribute__((always_inline)) {
    int wo;
    unsigned gm;
    if constexpr (REG) {
      wo = (k * RP + r_hx) * KPIX + 8 * r_c4;
      gm = unit_ok(k) ? 0xffffffffu : 0u;
    } else {
      if (woff[k] < 0) return;
      wo = woff[k];
      gm = gmask[k];
    }
    const float x[4] = {ra[k].x, ra[k].y, ra[k].z, ra[k].w};
    ch_h4 hh, ll;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = __uint_as_float(__float_as_uint(x[e]) & gm) * sa;
      const _Float16 h = (_Float16)v;
      hh[e] = h;
      ll[e] = (_Float16)(v - (float)h);
    }
    *reinterpret_cast<ch_h4*>(&lds[buf * BUF + wo]) = hh;
    *reinterpret_cast<ch_h4*>(&lds[buf * BUF + wo + CK * 2]) = ll;
  };

  // the first halo is requested before anything else (loads return in order: nothing may queue in front of it)
  float4 ra0[LP];
  load_chunk(0, ra0);

  // ---- scales: the producer's maximum (64 slots, see the epilogue) and the weight image's; requested here, used
  // after the weight queue below is in flight (loads return in order: waiting for these must not wait for those) ----
  const float* meta = reinterpret_cast<const float*>(P.wimg + (size_t)Cin * 9 * Cout * 4);
  float amax_lane = P.in_amax[(size_t)b * P.amax_stride + lane];
  const float inv_sw = meta[n0 + (lane & 31)];  // per output channel (column): the pack scales every column to [2^13, 2^14)

  // ---- A rows of this lane: logical row sigma(i) of block mb -> centre pixel in the halo -----------------------
  int arow[MB];
  {
    const int Lr = ch2::sigma(lane & 31);
    const int seg = SEG == 16 ? (Lr >> 4) : 0, pos = SEG == 16 ? (Lr & 15) : Lr;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
      arow[mb] = ((mb * RPS + seg + 1) * RP + pos + 1) * KPIX + (16 * wk + 8 * g) * 2;
  }

  // ---- this wave's weight stream -----------------------------------------------------------------------------
  // k16 block WK c + wk in chunk c: nine contiguous pairs per chunk, WK * 18 KiB from one chunk to the next.
  // Queue of D pairs (D divides 9): tap t of a chunk sits in slot t % D; the pair D taps ahead is requested into
  // the slot a tap just freed.
  static_assert(9 % D == 0, "queue depth");
  const unsigned char* wp = P.wimg + ((size_t)((n0 >> 5) * (Cin >> 4) + wk) * 9) * 2048 + lane * 16;
  constexpr size_t kChunkStride = (size_t)WK * 9 * 2048;
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
  typedef float f2v __attribute__((ext_vector_type(2)));
  f2v tot[FL > 0 ? MB : 1][8];
  ch_f16v zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
  if (FL > 0) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 8; ++r) tot[mb][r] = f2v{0.f, 0.f};
  }

#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) amax_lane = fmaxf(amax_lane, __shfl_xor(amax_lane, off));
  sa = ch2::pow2_scale(amax_lane, 14);
  const float descale = (1.0f / sa) * inv_sw;

  CH2_STAMP(2);
#pragma unroll
  for (int k = 0; k < LP; ++k) store_unit(0, ra0, k);
  __syncthreads();
  CH2_STAMP(3);

  // one 64-channel chunk: nine taps from LDS buffer c & 1.  MORE: the next chunk's halo is requested at the top
  // and split into the other buffer one unit at a time BETWEEN the MFMAs of taps 1..8 (a wave issues in order:
  // the ~7 VALU slots between two 8-pass MFMAs are free), and each tap requests the fragment pair D steps
  // ahead into the queue slot it just freed.  The A fragments of tap t + 1 are read while the MFMAs of tap t run.
  auto read_a = [&](const unsigned char* A, int t, ch_h8 (&ah)[MB], ch_h8 (&al)[MB]) {
    const int shift = ((t / 3 - 1) * RP + (t % 3 - 1)) * KPIX;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      ah[mb] = *reinterpret_cast<const ch_h8*>(A + arow[mb] + shift);
      al[mb] = *reinterpret_cast<const ch_h8*>(A + arow[mb] + shift + CK * 2);
    }
  };
  constexpr int T0 = 5;
  auto chunk = [&](int c, auto more_c, auto first_c) __attribute__((always_inline)) {
    constexpr bool MORE = decltype(more_c)::value;
    constexpr bool FIRST = decltype(first_c)::value;   // first chunk of a segment (FL > 0)
    float4 ra[LP];
    if (MORE && !(ABL & 2)) load_chunk(c + 1, ra);
    if (ABL & 2) {
#pragma unroll
      for (int k = 0; k < LP; ++k) ra[k] = make_float4(1.f, 2.f, 3.f, 4.f);
    }
    const unsigned char* A = &lds[(c & 1) * BUF];
    const unsigned char* wcur = wp + (size_t)c * kChunkStride;
    if constexpr (FL > 0 && MB > 2) {
      // the segmented form of the whole-image tiling: the second accumulator set takes the registers of the double-buffered
      // A fragments (2 x 7 blocks x 8) -- here a ring of four (block, tap) sub-steps, read two sub-steps ahead
      constexpr int S = 9 * MB, NBR = 3, AH = 2;
      ch_h8 rh[NBR], rl[NBR];
      auto rd = [&](int s) __attribute__((always_inline)) {
        const int t = s / MB, mb = s % MB;
        const int shift = ((t / 3 - 1) * RP + (t % 3 - 1)) * KPIX;
        rh[s % NBR] = *reinterpret_cast<const ch_h8*>(A + arow[mb] + shift);
        rl[s % NBR] = *reinterpret_cast<const ch_h8*>(A + arow[mb] + shift + CK * 2);
      };
#pragma unroll
      for (int s = 0; s < AH; ++s) rd(s);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const ch_h8 bh = qh[t % D], bl = ql[t % D];
        if (MORE || t + D < 9) {
          const unsigned char* wa = wcur + ((t + D) / 9) * kChunkStride + (size_t)((t + D) % 9) * 2048;
          qh[t % D] = *reinterpret_cast<const ch_h8*>(wa);
          ql[t % D] = *reinterpret_cast<const ch_h8*>(wa + 1024);
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const int sidx = t * MB + mb;
          if (sidx + AH < S) rd(sidx + AH);
          const bool f = FIRST && t == 0;
          if (f) {   // (accumulation registers -> arch VGPR pair -> v_pk_add_f32, as conv_h2w.hip's flush)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              f2v tt;
              asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(tt[0]) : "a"(acc[mb][2 * r]));
              asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(tt[1]) : "a"(acc[mb][2 * r + 1]));
              asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(tot[mb][r]) : "v"(tt));
            }
          }
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rl[sidx % NBR], bh, f ? zero16 : acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh[sidx % NBR], bl, acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh[sidx % NBR], bh, acc[mb], 0, 0, 0);
        }
        if (MORE && t >= T0) {
#pragma unroll
          for (int k = 0; k < LP; ++k)
            if (T0 + ((9 - T0) * k) / LP == t) store_unit((c + 1) & 1, ra, k);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
    ch_h8 ah[2][MB], al[2][MB];
    read_a(A, 0, ah[0], al[0]);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const ch_h8 bh = qh[t % D], bl = ql[t % D];
      if ((MORE || t + D < 9) && !(ABL & 1)) {  // the pair D taps ahead: this chunk's or the next one's
        const unsigned char* wa = wcur + ((t + D) / 9) * kChunkStride + (size_t)((t + D) % 9) * 2048;
        qh[t % D] = *reinterpret_cast<const ch_h8*>(wa);
        ql[t % D] = *reinterpret_cast<const ch_h8*>(wa + 1024);
      }
      if (t < 8 && !(ABL & 8)) read_a(A, t + 1, ah[(t + 1) & 1], al[(t + 1) & 1]);
      if (t < 8 && (ABL & 8)) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) { ah[(t + 1) & 1][mb] = ah[0][mb]; al[(t + 1) & 1][mb] = al[0][mb]; }
      }
      __builtin_amdgcn_sched_barrier(0);  // the loads above are ISSUED here, not sunk next to their uses
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const bool f = FIRST && t == 0;   // the block's finished segment -> tot, the new one starts from C = 0
        if (f) {
#pragma unroll
          for (int r = 0; r < 8; ++r) tot[mb][r] += f2v{acc[mb][2 * r], acc[mb][2 * r + 1]};
        }
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t & 1][mb], bh, f ? zero16 : acc[mb], 0, 0, 0);
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t & 1][mb], bl, acc[mb], 0, 0, 0);
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t & 1][mb], bh, acc[mb], 0, 0, 0);
      }
      if (MORE && t >= T0 && !(ABL & 4)) {  // units k with T0 + (9 - T0) k / LP == t: not before the loads had time to return
#pragma unroll
        for (int k = 0; k < LP; ++k)
          if (T0 + ((9 - T0) * k) / LP == t) store_unit((c + 1) & 1, ra, k);
#pragma unroll
        for (int i = 0; i < 3 * MB; ++i) {  // one MFMA, then up to 7 VALU instructions of the split
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 2 * ((LP + 8 - T0) / (9 - T0)), 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    __syncthreads();
  };
  constexpr std::true_type T{};
  constexpr std::false_type F{};
  if constexpr (FL == 0) {
#pragma unroll 1
    for (int c = 0; c + 1 < NC; ++c) {
      chunk(c, T, F);
      if (c < 8) { CH2_STAMP(4 + c); }
    }
    chunk(NC - 1, F, F);
  } else {
    static_assert(FL == 2, "chunks per segment");   // (NC % 2 == 0: Cin % (32 WK) == 0, the launcher's business)
#pragma unroll 1
    for (int c = 0; c + 2 < NC; c += 2) {
      chunk(c, T, T);
      chunk(c + 1, T, F);
    }
    chunk(NC - 2, T, T);
    chunk(NC - 1, F, F);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const f2v t = tot[mb][r] + f2v{acc[mb][2 * r], acc[mb][2 * r + 1]};
        acc[mb][2 * r] = t[0];
        acc[mb][2 * r + 1] = t[1];
      }
  }
  CH2_STAMP(12);

  // ---- sum of the k-waves through LDS in a fixed order -- (w0 + w2) + (w1 + w3), with eight waves
  // ((w0 + w4) + (w2 + w6)) + ((w1 + w5) + (w3 + w7)).  Every wave finishes a share of the tile: register quad
  // q = wk & 3 of every block (with eight waves: rows 2 hh, 2 hh + 1 of the quad, hh = wk >> 2), so the stores,
  // the pool and the maximum are spread over all waves ----
  float* xch = reinterpret_cast<float*>(lds);
  auto xaddr = [&](int slot, int mb, int r) { return (((wn * WK + slot) * MB + mb) * 16 + r) * 64 + lane; };
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) xch[xaddr(wk, mb, r)] = acc[mb][r];
  __syncthreads();
  CH2_STAMP(13);

  constexpr int NE = WK == 8 ? 2 : 4;  // rows of the quad this wave finishes
  const int q4 = wk & 3, e0 = WK == 8 ? 2 * (wk >> 2) : 0;
  const float bias_j = P.bias[n0 + j];
  float* outb = P.out + (size_t)b * H * W * Cout + n0 + j;
  const int L0 = ch2::sigma(8 * q4 + 4 * g);  // first of the quad's four consecutive logical rows
  const int seg = SEG == 16 ? (L0 >> 4) : 0, pos0 = SEG == 16 ? (L0 & 15) : L0;
  float val[MB][NE];
  float vmax = 0.f;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int y = y0 + mb * RPS + seg;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int r = 4 * q4 + e0 + e, tx = pos0 + e0 + e, x = x0 + tx;
      float v;
      if (WK == 8)
        v = ((xch[xaddr(0, mb, r)] + xch[xaddr(4, mb, r)]) + (xch[xaddr(2, mb, r)] + xch[xaddr(6, mb, r)])) +
            ((xch[xaddr(1, mb, r)] + xch[xaddr(5, mb, r)]) + (xch[xaddr(3, mb, r)] + xch[xaddr(7, mb, r)]));
      else
        v = (xch[xaddr(0, mb, r)] + xch[xaddr(2, mb, r)]) + (xch[xaddr(1, mb, r)] + xch[xaddr(3, mb, r)]);
      v = fmaf(v, descale, bias_j);
      if (P.relu) v = fmaxf(v, 0.f);
      val[mb][e] = v;
      if (tx < TW && y < H && x < W) {
        outb[((size_t)y * W + x) * Cout] = v;
        vmax = fmaxf(vmax, fabsf(v));
      }
    }
  }
  if (P.out_amax) {  // 64 slots: same-address atomics serialise in L2 (~10 ns each)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off));
    if (lane == 0)
      atomicMax(reinterpret_cast<unsigned*>(P.out_amax) + (size_t)b * P.amax_stride + ((blockIdx.x * WK * NW + wave) & 63),
                __float_as_uint(vmax));
  }
  if (P.pool_out) {  // H, W even; y0, x0 even: a 2x2 window never leaves the tile, nor this wave's quad
    const int Hp = H >> 1, Wp = W >> 1;
    float* pb = P.pool_out + (size_t)b * Hp * Wp * Cout + n0 + j;
    if (SEG == 32) {  // patch row = block: the window's rows are blocks 2p, 2p + 1 of the same lane
#pragma unroll
      for (int p = 0; p < MB / 2; ++p)
#pragma unroll
        for (int e = 0; e < NE; e += 2) {
          const int tx = L0 + e0 + e, x = x0 + tx, y = y0 + 2 * p;
          const float m = fmaxf(fmaxf(val[2 * p][e], val[2 * p][e + 1]), fmaxf(val[2 * p + 1][e], val[2 * p + 1][e + 1]));
          if (tx < TW && y + 1 < H && x + 1 < W) pb[((size_t)(y >> 1) * Wp + (x >> 1)) * Cout] = m;
        }
    } else {  // two patch rows per block: the window's second row is held by the other half wave
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int e = 0; e < NE; e += 2) {
          float m = fmaxf(val[mb][e], val[mb][e + 1]);
          m = fmaxf(m, __shfl_xor(m, 32));
          const int tx = (L0 & 15) + e0 + e, x = x0 + tx, y = y0 + 2 * mb;
          if (L0 < 16 && tx < TW && y + 1 < H && x + 1 < W) pb[((size_t)(y >> 1) * Wp + (x >> 1)) * Cout] = m;
        }
    }
  }
  CH2_STAMP(14);
}

// ---------------------------------------------------------------------------------------------------
// conv1_1 (models/CNN/vgg.py:187, 3 -> 64 channels, K = 27): 0.17 GFLOP per image -- not matrix-pipe work.  Direct fp32
// FMA convolution: a workgroup walks tiles of 2 rows x 32 pixels (grid-stride), stages their 4 x 34 x 3 input window in
// LDS as nine [row][channel] planes of 36 pixels, thread (row, group of 4 pixels, channel quad) keeps its 27 x 4 weights
// in registers for the whole walk, reads its 6-pixel span of every plane with one 16-byte + one 8-byte LDS read
// (24 + 12 reads per 432 FMAs; round 2: one 4-byte read per 4 FMAs -- LDS-issue bound, 53 us for eight images) and
// stores 16-byte channel quads (16 lanes = one pixel = 256 contiguous bytes); per-workgroup maximum -> the 64 slots the
// first conv_h2 layer scales by.
// ---------------------------------------------------------------------------------------------------
// RT: row pairs per tile (tile = 2 RT rows x 32 pixels).  One row pair is ~0.4 us of FMAs behind a 1-2 us window load that one
// tile of look-ahead cannot hide (72 us per 16 images, where the 205 MB of stores need ~45): RT = 4 puts 1.6 us of work behind
// every window.
template <int RT>
__global__ __launch_bounds__(256) void conv1_1_direct_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                             const float* __restrict__ bias, int B, int H, int W,
                                                             int relu, float* __restrict__ out,
                                                             float* __restrict__ out_amax, int amax_stride) {
  constexpr int PW = 40;                                 // plane pitch in floats (36 used; 160 B keeps 16-byte alignment)
  constexpr int WR = 2 * RT + 2;                         // window rows
  constexpr int WE = WR * 34 * 3;                        // window elements
  constexpr int LQ = (WE + 255) / 256;                   // per thread
  __shared__ __attribute__((aligned(16))) float win[2][WR * 3 * PW];   // [buffer][row][channel][pixel -1 .. 34]
  __shared__ __attribute__((aligned(16))) float wl[27 * 64];
  __shared__ float red[4];
  const int tid = threadIdx.x, c4 = tid & 15, pg = (tid >> 4) & 7, ry = tid >> 7;   // pixels 4 pg .. 4 pg + 3 of row ry
  for (int i = tid; i < 27 * 16; i += 256)  // the 6.9 KB of weights once per workgroup, then per thread from LDS
    reinterpret_cast<float4*>(wl)[i] = reinterpret_cast<const float4*>(w)[i];
  __syncthreads();
  float4 wr[27];                                         // [r][dx][c] as in the TF tensor [3][3][3][64]
#pragma unroll
  for (int k = 0; k < 27; ++k) wr[k] = *reinterpret_cast<const float4*>(&wl[k * 64 + 4 * c4]);
  const float4 bv = *reinterpret_cast<const float4*>(bias + 4 * c4);
  const int segs_x = (W + 31) >> 5, rows2 = (H + 2 * RT - 1) / (2 * RT);   // tiles per row band, row bands
  const long ntile = (long)B * rows2 * segs_x;
  float vmax = 0.f;
  int buf = 0, bcur = -1;
  // the maximum is kept per IMAGE (amax_stride floats between the images' slot groups): a workgroup that walks
  // from one image into the next hands its maximum in first
  auto flush = [&](int bimg) {
    float m = vmax;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0)
      atomicMax(reinterpret_cast<unsigned*>(out_amax) + (size_t)bimg * amax_stride + (blockIdx.x & 63),
                __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
    __syncthreads();
    vmax = 0.f;
  };
  // Round 4: the window of tile t + 1 is REQUESTED (two 4-byte loads per thread) before tile t is computed and written
  // into the other LDS buffer after it -- the global-load latency (1-2 us: as long as a tile's 432 FMAs per thread)
  // no longer sits between two tiles.
  auto tile_of = [&](long tl, int& b, int& y0, int& x0) {
    const int sx = (int)(tl % segs_x);
    const long by = tl / segs_x;
    y0 = 2 * RT * (int)(by % rows2);
    b = (int)(by / rows2);
    x0 = sx * 32;
  };
  auto win_load = [&](long tl, float (&v)[LQ]) {   // window element (row r, pixel px, channel c), read in memory order
    int b, y0, x0;
    tile_of(tl, b, y0, x0);
    const float* img = in + (size_t)b * H * W * 3;
#pragma unroll
    for (int k = 0; k < LQ; ++k) {
      const int i = tid + 256 * k;
      const int r = i / 102, rem = i - r * 102, px = rem / 3, c = rem - px * 3;
      const int yy = y0 - 1 + r, xx = x0 - 1 + px;
      v[k] = (i < WE && yy >= 0 && yy < H && xx >= 0 && xx < W) ? img[((size_t)yy * W + xx) * 3 + c] : 0.f;
    }
  };
  auto win_store = [&](int bf, const float (&v)[LQ]) {
#pragma unroll
    for (int k = 0; k < LQ; ++k) {
      const int i = tid + 256 * k;
      const int r = i / 102, rem = i - r * 102, px = rem / 3, c = rem - px * 3;
      if (i < WE) win[bf][(r * 3 + c) * PW + px] = v[k];
    }
  };
  float nxt[LQ];
  if ((long)blockIdx.x < ntile) {
    win_load(blockIdx.x, nxt);
    win_store(0, nxt);
  }
  __syncthreads();
  for (long tl = blockIdx.x; tl < ntile; tl += gridDim.x, buf ^= 1) {
    int b, y0, x0;
    tile_of(tl, b, y0, x0);
    if (out_amax && amax_stride && bcur >= 0 && b != bcur) flush(bcur);
    bcur = b;
    const bool more = tl + gridDim.x < ntile;
    if (more) win_load(tl + gridDim.x, nxt);
#pragma unroll 1
    for (int q = 0; q < RT; ++q) {
    float4 a[4] = {bv, bv, bv, bv};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* pl = &win[buf][((2 * q + ry + r) * 3 + c) * PW + 4 * pg];
        const float4 v0 = *reinterpret_cast<const float4*>(pl);        // window pixels 4 pg .. 4 pg + 3 (image x0 - 1 + ..)
        const float2 v1 = *reinterpret_cast<const float2*>(pl + 4);    // 4 pg + 4, 4 pg + 5
        const float v[6] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y};
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float4 ww = wr[(r * 3 + dx) * 3 + c];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            a[p].x = fmaf(v[p + dx], ww.x, a[p].x); a[p].y = fmaf(v[p + dx], ww.y, a[p].y);
            a[p].z = fmaf(v[p + dx], ww.z, a[p].z); a[p].w = fmaf(v[p + dx], ww.w, a[p].w);
          }
        }
      }
    const int y = y0 + 2 * q + ry;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      float4 o = a[p];
      if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      const int x = x0 + 4 * pg + p;
      if (y < H && x < W) {
        *reinterpret_cast<float4*>(out + (((size_t)b * H + y) * W + x) * 64 + 4 * c4) = o;
        vmax = fmaxf(fmaxf(fmaxf(vmax, fabsf(o.x)), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w)));
      }
    }
    }
    if (more) win_store(buf ^ 1, nxt);
    __syncthreads();  // one barrier per tile: the next window is complete, this one is read by nobody any more
  }
  if (out_amax && bcur >= 0) flush(amax_stride ? bcur : 0);
}

// w: the TF tensor [3][3][3][64] as is; out_amax: 64 slots (per image when amax_stride > 0) zeroed by the caller, or nullptr
hipError_t conv1_1_direct_launch(const float* in, int B, int H, int W, const float* w_hwio, const float* bias, int relu,
                                 float* out, float* out_amax, hipStream_t st, int amax_stride) {
  // a step at a time: two-row tiles (784 per image fill the chip); a batched call: eight-row tiles
  const int rt = B >= tune::conv_wide_min ? (tune::conv11_rt > 0 ? tune::conv11_rt : 4) : 1;
  const long ntile = (long)B * ((H + 2 * rt - 1) / (2 * rt)) * ((W + 31) / 32);
  const long cap = tune::conv11_wgs > 0 ? tune::conv11_wgs : 512;
  const int grid = (int)(ntile < cap ? ntile : cap);  // two workgroups per CU
  if (rt == 4) hipLaunchKernelGGL(conv1_1_direct_kernel<4>, dim3(grid), dim3(256), 0, st, in, w_hwio, bias, B, H, W, relu, out, out_amax, amax_stride);
  else if (rt == 2) hipLaunchKernelGGL(conv1_1_direct_kernel<2>, dim3(grid), dim3(256), 0, st, in, w_hwio, bias, B, H, W, relu, out, out_amax, amax_stride);
  else hipLaunchKernelGGL(conv1_1_direct_kernel<1>, dim3(grid), dim3(256), 0, st, in, w_hwio, bias, B, H, W, relu, out, out_amax, amax_stride);
  return hipGetLastError();
}

template <int MB, int NW, int SEG, int TW, int D, int WK, int OCC = 1, int FL = 0>
static hipError_t conv_h2_go(ConvH2Dev d, hipStream_t st) {
  if (FL > 0 && (d.Cin / (16 * WK)) % FL != 0) return hipErrorInvalidValue;
  constexpr int TH = MB * (32 / SEG);
  d.tiles_x = (d.W + TW - 1) / TW;
  d.tiles_y = (d.H + TH - 1) / TH;
  const int grid = d.B * d.tiles_x * d.tiles_y * (d.Cout / (32 * NW));
  hipLaunchKernelGGL((conv_h2_kernel<MB, NW, SEG, TW, D, WK, 0, OCC, FL>), dim3(grid), dim3(64 * WK * NW), 0, st, d);
  return hipGetLastError();
}

bool conv_h2_supported(int H, int W, int Cin, int Cout) {
  return Cin > 0 && Cin % 64 == 0 && Cout > 0 && Cout % 64 == 0 && H > 0 && W > 0 && H < 32768 && W < 32768 &&
         (size_t)H * W * (Cin > Cout ? Cin : Cout) < (size_t)1 << 31;
}

// cfg: 0 = by shape and batch; 1..4 force <1,1,16,14>, <2,1,32,28>, <2,2,32,28>, <4,2,16,16> (tests: every shape through
// every tiling); 5..9 force variants 1..5 of the batched form (conv_h2w.hip)
hipError_t conv_h2_launch(const float* in, int B, int H, int W, int Cin, const void* wimg, const float* bias,
                          int Cout, int relu, const float* in_amax, float* out, float* pool_out, float* out_amax,
                          hipStream_t st, int cfg, int amax_stride) {
  ConvH2Dev d{};
  d.in = in; d.wimg = static_cast<const unsigned char*>(wimg); d.bias = bias; d.in_amax = in_amax;
  d.out = out; d.pool_out = pool_out; d.out_amax = out_amax;
  d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.Cout = Cout; d.relu = relu; d.amax_stride = amax_stride;
#ifdef DISN_TUNING
  d.stamps = tune::ch2_stamps;
#endif
  // Which operand an XCD's L2 keeps: with the n-tile-major order every XCD streams the whole input of the launch
  // (8 x B H W Cin floats over the fabric) and reads its own eighth of the weights once; image-major it is the other
  // way round.  Measured on the 13 layers of 2 / 4 / 8 images (tools/conv_stack_time.py, knob conv_img_major):
  // image-major where B H W > 9 Cout 388 / 645-658 / 1059-1078 us against 385 / 627 / 1089 n-tile-major, everywhere
  // 399 / 665 / 1112 -- no gain: the launches are not bound by fabric traffic.  n-tile-major stays.
  d.img_major = 0;
  if (tune::conv_img_major >= 0) d.img_major = tune::conv_img_major;
  // Calls of several images (the steps of a batched call): the batched form of conv_h2w.hip -- waves own n-blocks
  // and walk K sequentially, 12 .. 21 MFMAs per weight pair.  Its summation order differs from the k-wave tree
  // below (fp32 rounding; both inside the 1e-5 bar): an image's bits depend on WHICH FORM runs it (fewer than
  // kConvWideMinImages images per call or not), never on the other images of its call.  tiling 5..9: forced variants.
  // 14 x 14 layers (conv5_x) of a batched call (B >= 4, as for the batched form below): FOUR k-waves, whatever B -- so
  // their bits, like the other layers', depend on "four images or more per call" only.  Where that fills the chip
  // (B Cout / 32 >= 200 workgroups: from 13 images on at 512 channels) ONE workgroup per image and n-block takes the
  // whole image as one patch of seven 32-pixel blocks: the two-row patches give a weight pair three MFMAs (1.03 GB of
  // fragment reads for conv5_x of 16 images: L2-bound, 32 % MFMA busy, 65 us), the whole image 21 (52 us).  Smaller
  // calls keep the two-row patches (measured equal to the eight-k-wave tiling at 4 / 8 / 12 images: r03ad).
  if (cfg == 10) return conv_h2_go<7, 1, 16, 14, 3, 4>(d, st);
  if (cfg == 19) return conv_h2_go<7, 1, 16, 14, 3, 4, 1, 2>(d, st);   // the whole-image tiling in segments of two chunks
  // cfg 11 ("strict", disn_vgg_weights_t.strict_forms = 1): the single-image tilings below by shape, whatever B -- the same
  // k-waves and summation tree as a call of one image: the same bits
  const bool strict = cfg == 11;
  if (strict) cfg = 0;
  // cfg 18 (the training step): the batched forms of round 3 for every layer (conv_h2w_launch variant -1)
  const bool fast = cfg == 18;
  if (fast) cfg = 0;
  if (!strict && cfg == 0 && tune::conv5_whole && W <= 14 && H <= 14 && B >= tune::conv_wide_min) {
    if (fast || Cin % 128 != 0) {   // (the training step: round 3's chains of 216)
      if ((long)B * (Cout / 32) >= 200) return conv_h2_go<7, 1, 16, 14, 3, 4>(d, st);
      return conv_h2_go<1, 1, 16, 14, 3, 4, 2>(d, st);
    }
    // inference: the four k-waves in segments of two chunks (chains of 54), whatever the patch
    if ((long)B * (Cout / 32) >= 200) return conv_h2_go<7, 1, 16, 14, 3, 4, 1, 2>(d, st);
    return conv_h2_go<1, 1, 16, 14, 3, 4, 2, 2>(d, st);
  }
  if (cfg >= 5) return conv_h2w_supported(H, W, Cin, Cout) ? conv_h2w_launch(d, st, cfg >= 12 ? cfg - 6 : cfg - 4) : hipErrorInvalidValue;
  if (!strict && cfg == 0 && B >= tune::conv_wide_min && conv_h2w_supported(H, W, Cin, Cout)) return conv_h2w_launch(d, st, fast ? -1 : 0);
  if (cfg == 0) {
    // patch shape by image width; n-blocks per workgroup so that one image still gives >= ~200 workgroups
    if (W <= 14) cfg = 1;
    else if (W <= 28 || (W % 28 == 0 && W % 16 != 0)) {
      // by the workgroups ONE image gives, whatever the batch: the tiling fixes the number of k-waves, i.e. the
      // summation order -- an image's bits must not depend on the batch it travels in
      const long wgs1 = (long)((H + 1) / 2) * ((W + 27) / 28) * (Cout / 32);
      cfg = wgs1 <= 320 ? 2 : 3;
    } else cfg = 4;
  }
  // one n-block per workgroup: eight k-waves (two waves per SIMD) when the channel count allows 128-channel chunks
  const bool wk8 = Cin % 128 == 0;
  // Launches of more than one round of workgroups (the batched calls; the 224 / 112-pixel layers of a single image):
  // variants cut for TWO workgroups per CU -- 128 registers (weight queue one pair deep: the other workgroup's waves
  // hide the L2 latency instead), <= 80 KB LDS (half-height patches at 16 x 16) -- so that one workgroup's prologue,
  // chunk barriers and epilogue run under the other's MFMAs.  Same k-waves, same summation order: same bits.
  // Measured (tools/conv_stack_time.py 4, r02v): the 13 layers of four images 697 -> 631 us.
  if (tune::conv_occ != 1) {
    const long per_img_tiles = cfg == 1 ? (long)((H + 1) / 2) * ((W + 13) / 14)
                             : cfg == 3 ? (long)((H + 1) / 2) * ((W + 27) / 28)
                             : cfg == 4 ? (long)((H + 3) / 4) * ((W + 15) / 16) : 0;
    const long wgs = B * per_img_tiles * (Cout / (cfg == 1 ? 32 : 64));
    const int m = tune::conv_occ_mask;   // bit per tiling (tuning builds)
    // From 1.5 workgroups per CU on where the patch stays the same; where it is halved (more halo per output) from 6
    // per CU on -- a single image's 224 / 112-pixel layers are slower with it (the 13 layers 276 -> 308 us), two
    // images' the same, four images' faster.
    if (cfg == 1 && wk8 && (m & 1) && wgs >= tune::conv_occ_min) return conv_h2_go<1, 1, 16, 14, 3, 8, 2>(d, st);
    // 28-pixel layers with eight k-waves (conv4): their own tiling needs 131 KB of LDS; the 2 x 14 patch of conv5
    // has the same halo overhead (2.3x against 2.1x) and fits twice
    if (cfg == 2 && wk8 && (m & 8) && B * (long)((H + 1) / 2) * ((W + 13) / 14) * (Cout / 32) >= 4 * tune::conv_occ_min)
      return conv_h2_go<1, 1, 16, 14, 3, 8, 2>(d, st);
    if (cfg == 3 && (m & 2) && wgs >= tune::conv_occ_min) return conv_h2_go<2, 2, 32, 28, 1, 4, 2>(d, st);
    if (cfg == 4 && (m & 4) && wgs >= 4 * tune::conv_occ_min) return conv_h2_go<2, 2, 16, 16, 1, 4, 2>(d, st);
  }
  switch (cfg) {
    case 1: return wk8 ? conv_h2_go<1, 1, 16, 14, 9, 8>(d, st) : conv_h2_go<1, 1, 16, 14, 9, 4>(d, st);
    case 2: return wk8 ? conv_h2_go<2, 1, 32, 28, 9, 8>(d, st) : conv_h2_go<2, 1, 32, 28, 9, 4>(d, st);
    case 3: return conv_h2_go<2, 2, 32, 28, 9, 4>(d, st);
    default: return conv_h2_go<4, 2, 16, 16, 3, 4>(d, st);
  }
}

}  // namespace disn

#ifndef CH2_UBENCH
// ---- C ABI: the layer as a unit (tests, composition); disn_encode* use conv_h2_launch directly ----------
#include "../../include/disn_amd.h"

extern "C" {

size_t disn_pack_conv_h2_bytes(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 64) return 0;
  return disn::conv_h2_image_bytes(Cin, Cout);
}

int disn_pack_conv_h2(const float* w_hwio, int Cin, int Cout, void* image, void* stream) {
  if (!w_hwio || !image || Cin <= 0 || Cout <= 0) return DISN_E_ARG;
  if (Cin % 64 || Cout % 64) return DISN_E_SHAPE;
  const hipError_t e = disn::conv_h2_pack_launch(w_hwio, Cin, Cout, image, nullptr, (hipStream_t)stream);
  return e == hipSuccess ? 0 : (int)e;
}

int disn_conv_h2_gain_span(const void* image, int Cin, int Cout, float* span_log2, void* stream) {
  if (!image || !span_log2 || Cin <= 0 || Cout <= 0 || Cout > 4096) return DISN_E_ARG;
  if (Cin % 64 || Cout % 64) return DISN_E_SHAPE;
  // the image's tail: inv_sw[Cout], then the column maxima cmax[Cout] the pack measured (what the scales came from)
  float inv[4096];
  hipError_t e = hipMemcpyAsync(inv, static_cast<const unsigned char*>(image) + (size_t)Cin * 9 * Cout * 4 + (size_t)Cout * 4,
                                (size_t)Cout * 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  float lo = 0.f, hi = 0.f;
  for (int f = 0; f < Cout; ++f) {
    const float v = inv[f];
    if (!(v > 0.f) || !(v < 3.0e38f)) continue;   // an all-zero column has no gain
    if (lo == 0.f || v < lo) lo = v;
    if (v > hi) hi = v;
  }
  *span_log2 = lo > 0.f ? log2f(hi / lo) : 0.f;
  return *span_log2 > 12.0f ? DISN_W_GAIN_SPAN : 0;
}

size_t disn_conv3x3_h2_workspace_bytes(int B) { return B > 0 ? (size_t)B * 512 : 0; }

int disn_conv3x3_h2(const float* in, int B, int H, int W, int Cin, const void* image, const float* bias, int Cout,
                    int relu, float* out, float* pool_out, float* out_amax, int tiling, void* ws, size_t ws_bytes,
                    void* stream) {
  if (!in || !image || !bias || !out || !ws || B <= 0 || H <= 0 || W <= 0) return DISN_E_ARG;
  if (!disn::conv_h2_supported(H, W, Cin, Cout) || tiling < 0 || tiling > 19 || (tiling >= 14 && tiling <= 17) || (pool_out && ((H | W) & 1)))
    return DISN_E_SHAPE;
  if (((tiling >= 5 && tiling <= 9) || tiling == 12 || tiling == 13) && !disn::conv_h2w_supported(H, W, Cin, Cout)) return DISN_E_SHAPE;
  if ((tiling == 6 || tiling == 8 || tiling == 12) && Cout % 128) return DISN_E_SHAPE;   // four n-waves = 128 channels per workgroup
  if ((tiling == 12 || tiling == 13) && (Cin % 64 || Cin < 128)) return DISN_E_SHAPE;       // two K halves of whole segments
  if (tiling == 19 && (H > 14 || W > 14 || Cin % 128)) return DISN_E_SHAPE;
  if (tiling == 10 && (H > 14 || W > 14)) return DISN_E_SHAPE;
  if (ws_bytes < (size_t)B * 512) return DISN_E_WS;
  hipStream_t st = (hipStream_t)stream;
  // every image its own 64 slots in / 64 slots out, as inside disn_encode*: an image's scale, hence its bits, do
  // not depend on the other images of the call
  float* amax_in = static_cast<float*>(ws);
  float* amax_out = amax_in + (size_t)B * 64;
  hipError_t e = hipMemsetAsync(amax_in, 0, (size_t)B * 512, st);
  if (e != hipSuccess) return (int)e;
  e = disn::amax64_accumulate_launch(in, (size_t)H * W * Cin, amax_in, st, B, 64);
  if (e != hipSuccess) return (int)e;
  e = disn::conv_h2_launch(in, B, H, W, Cin, image, bias, Cout, relu, amax_in, out, pool_out,
                           out_amax ? amax_out : nullptr, st, tiling, 64);
  if (e == hipSuccess && out_amax) e = disn::amax_fold_launch(amax_out, out_amax, st, B * 64);
  return e == hipSuccess ? 0 : (int)e;
}

size_t disn_conv1_1_workspace_bytes(void) { return 256; }

int disn_conv1_1(const float* in, int B, int H, int W, const float* w_hwio, const float* bias, int relu, float* out,
                 float* out_amax, void* ws, size_t ws_bytes, void* stream) {
  if (!in || !w_hwio || !bias || !out || !ws || B <= 0 || H <= 0 || W <= 0) return DISN_E_ARG;
  if (ws_bytes < 256) return DISN_E_WS;
  hipStream_t st = (hipStream_t)stream;
  float* slots = static_cast<float*>(ws);
  hipError_t e = hipSuccess;
  if (out_amax) e = hipMemsetAsync(slots, 0, 256, st);
  if (e == hipSuccess) e = disn::conv1_1_direct_launch(in, B, H, W, w_hwio, bias, relu, out, out_amax ? slots : nullptr, st);
  if (e == hipSuccess && out_amax) e = disn::amax_fold_launch(slots, out_amax, st);
  return e == hipSuccess ? 0 : (int)e;
}

}  // extern "C"
#endif  // CH2_UBENCH
