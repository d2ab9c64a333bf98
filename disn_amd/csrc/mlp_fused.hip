// Fused point MLP: one persistent kernel per MLP stream runs
//     fold1/conv1 -> conv2 -> conv3 -> fold2/conv1 (+ per-image term) -> conv2 -> conv5
// (models/sdfnet.py:71-88 'sdfprediction', :173-186 'sdfprediction_imgfeat'; the sum of the two
// streams is models/model_normalization.py:204) for 32 points per wave with EVERY activation in
// registers; only the weights move (L2 -> LDS ring -> MFMA A operand) and, for the local stream, the
// four resampled rows of the folded feature map (disn_fold_local; models/model_normalization.py:172-190).
//
// Transposed formulation.  A layer is D[feature][point] = W^T[feature][k] . X[k][point]: the weights
// are the MFMA "A" operand, a wave's 32 points are the 32 columns of the "B" operand.  The C/D layout of
// v_mfma_f32_32x32x16_f16 gives lane (j = l&31, g = l>>5) the 16 features (r&3) + 8(r>>2) + 4g of point
// j; the B operand of the next layer wants from the same lane 8 reduction slots of point j.  Taking
// registers r = 8*half + t (t = 0..7) of output tile nt as the slots (g, t) of reduction block
// kb = 2*nt + half fixes the slot <-> feature map
//     phi(kb, g, t) = 16 kb + (t & 3) + 8 (t >> 2) + 4 g
// and the weights are packed once in that slot order, so an output tile turns into the next layer's
// operand with element-wise work only: no LDS round trip, no cross-lane traffic.
//
// fp32 results from the f16 matrix pipes -- the two-term split.  Every fp32 operand x (scaled by a
// power of two into fp16's range) is split as x = h + l, h = fp16(x), l = fp16(x - h): |x - h - l| <=
// 2^-23 |x| (11 + 11 significand bits and the sign of l), and a*b is accumulated in fp32 from
// l_a h_b + h_a l_b + h_a h_b (the dropped l_a l_b is below 2^-22 |a b|): three MFMAs per block
// instead of the six of the bf16 three-term split, 4 instead of 6 bytes per staged operand -- which is
// what lets 32 points x 512 features (x 2: input and output of fold2/conv1) live in one wave's 512
// registers.  Scales: weights per layer (pack time, max -> [2^13, 2^14)); activations per POINT, from
// the actual maximum of the point's features (conv1..conv3 outputs) or, for fold2/conv1 whose output
// tiles are consumed while later ones are still being computed, from the bound
// max|h3| * max_f ||W4[:,f]||_1 + max|additive term|.  All scales are powers of two (exact).
//
// Register plan (one wave per SIMD, 256 arch VGPRs + 256 accumulation VGPRs).  conv3 is chained into
// fold2/conv1 with the REDUCTION of fold2/conv1 as the outer loop: an output tile of conv3 (32
// features) becomes two reduction blocks of fold2/conv1 at once and is accumulated into all 16 of its
// output tiles, which are the 256 accumulation registers; the arch VGPRs hold conv2's output (128), the
// tile in flight and the weight fragments.  fold2/conv1's tiles are then drained one by one into the
// eight accumulators of fold2/conv2 the same way.  Only conv2's full output ever exists at once.
//
// Weight stream.  The packed image (2.16 MB per MLP stream) is ONE linear sequence of 1-KiB fragment
// planes in consumption order; a workgroup (4 waves = 128 points) pulls it through an LDS ring of
// 16-KiB slots with global_load_lds (each wave loads a quarter of a slot), D slots ahead, one
// s_barrier per slot, counted vmcnt (never 0 in steady state), across point-tile boundaries.
#include "kernels.hpp"
#include "tuning.hpp"

#include <type_traits>
#include <utility>

namespace disn {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace fm {
// stream order: conv2 (32 pairs); phase A, 16 x {conv3 tile it: 16 pairs; fold2/conv1 reduction blocks
// 2it, 2it+1 for its 16 output tiles: 32 pairs}; phase B, 16 x {fold2/conv2 reduction blocks 2it, 2it+1
// for its 8 output tiles: 16 pairs}
constexpr int kPairsL2 = 32, kPairsA = 48, kPairsB = 16;
constexpr int kPairs = kPairsL2 + 16 * kPairsA + 16 * kPairsB;  // 1056 (tile, k-block) pairs of (h, l) planes
constexpr int kSlotBytes = 16384;                             // 8 pairs
constexpr int kSlots = kPairs / 8;                            // 132 per pass over the image
constexpr size_t kImageBytes = (size_t)kPairs * 2048;         // 2 162 688
// FEAT form (round 4: the local stream of a SMALL point set, the gathered features as extra reduction blocks of
// fold2/conv1 -- models/sdfnet.py:180's concat [point 512 | feature 1472]): between phase A and phase B, phase A2 =
// 16 iterations x {six reduction blocks 32 + 6 it2 .. (feature columns 96 it2 .. 96 it2 + 95 of the zero-padded
// 1536) for the 16 output tiles: 96 pairs}.  1536 = 32 x 48 pairs: every later pair keeps its position mod 48, i.e.
// its fragment register set (mod 3) and ring-slot parity (mod 16)
constexpr int kFeatBlocks = 96, kFeatCols = 16 * kFeatBlocks, kFeatReal = 1472;   // 1536 columns, 1472 real
constexpr int kPairsA2 = 96;
constexpr int kPairsFeat = kPairs + 16 * kPairsA2;              // 2592
constexpr size_t kImageBytesFeat = (size_t)kPairsFeat * 2048;   // 5 308 416
// EQUALISED hidden features (round 4).  The kernel's hidden activations never leave it, so every hidden feature f of
// layer l may carry its own power-of-two factor c_l[f]: the image holds W~_l = diag(1 / c_{l-1}) W_l diag(c_l) (c_1 = 1:
// fold1/conv1 is not an MFMA layer), the kernel adds b_l c_l, scales the folded map's rows / the per-image bias row by
// c_4 and divides w6 by c_5 -- ReLU commutes with a positive factor, powers of two are exact, the stream's sum is
// unchanged.  c_l[f] = 2^-e(max_k |W_l[k, f]| / c_{l-1}[k]): every column of W~ has its largest entry in [1, 2), one
// weight scale 2^13 serves all, and -- the point -- the column 1-norms behind the activation-scale BOUNDS (conv3 and
// fold2/conv1 outputs are consumed tile by tile, their scales come from |h| <= max |h_prev| max_f ||W~[:, f]||_1 + ...)
// are those of a normalised network whatever gains training left in the channels.  With the raw weights a trained-like
// set (oracle trained_like_weights: log-normal channel gains, 1 % outliers x 1000) has column 1-norms 10^4 above the
// typical one; two such bounds compound to 2^-33 of the f16 range and the result is noise (max error 12 on |pred| 100
// in tests/fused_emulation.py; 1e-6 with the equalisation).
// meta (behind the image): floats [8] max_f ||W~4_point[:,f]||_1; [9] max_f ||W~3[:,f]||_1; [10] the same of the feature
// rows of fold2/conv1 (FEAT form); [11] max_f c_4[f]; from float 64 on c_2[256], c_3[512], c_4[512], c_5[256]
constexpr int mS2 = 64, mS3 = mS2 + 256, mS4 = mS3 + 512, mS5 = mS4 + 512, kMetaFloats = mS5 + 256;
constexpr float kInvSw = 1.0f / 8192.0f;   // every column of W~ is packed with the scale 2^13
constexpr float kColFloor = 1.0f / 1048576.0f;   // 2^-20: columns below that fraction of the layer's largest are not equalised further
// constants of one stream in LDS (floats)
constexpr int cW1 = 0, cB1 = 192, cB2 = 256, cB3 = 512, cB4 = 1024, cB5 = 1536, cW6 = 1792, cB6 = 2048;
constexpr int cS2 = 2052, cS3 = cS2 + 256, cS4 = cS3 + 512, cS5 = cS4 + 512;
constexpr int kConstFloats = cS5 + 256;
}  // namespace fm

// ---------------------------------------------------------------------------------------------------
// packing
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pow2_scale_for(float amax, int target_exp) {
  // power of two s with amax * s in [2^target_exp, 2^(target_exp+1)); 1 for amax == 0 / non-finite
  const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127;
  if (!(amax > 0.f) || e > 100 || e < -100) return 1.0f;
  return __uint_as_float((unsigned)(127 + target_exp - e) << 23);
}

// one workgroup per layer: per-feature (column) amax -> inverse scale; conv3 and fold2/conv1 (point rows) also the
// largest column 1-norm
// k4: rows of w4 -- 512 (the point rows) or 1984 (FEAT form: + the 1472 feature rows; their input is not a hidden
// feature: no row factor).  ONE workgroup: the layers in order, layer l's row factors are layer l-1's column factors.
__host__ __device__ inline int fm_meta_off(int layer) {
  return layer == 0 ? fm::mS2 : (layer == 1 ? fm::mS3 : (layer == 2 ? fm::mS4 : fm::mS5));
}
__global__ __launch_bounds__(256) void fm_meta_kernel(const float* __restrict__ w2, const float* __restrict__ w3,
                                                      const float* __restrict__ w4, const float* __restrict__ w5,
                                                      float* __restrict__ meta, int k4) {
  __shared__ float red[256];
  __shared__ float rrow[512];   // 1 / c_{l-1}[k] of the hidden input features (1 for layer 0)
  for (int i = threadIdx.x; i < 512; i += 256) rrow[i] = 1.0f;
  if (threadIdx.x < 16) meta[threadIdx.x] = 0.f;
  __syncthreads();
  for (int layer = 0; layer < 4; ++layer) {
    const float* w = layer == 0 ? w2 : (layer == 1 ? w3 : (layer == 2 ? w4 : w5));
    const int Kall = layer == 0 ? 64 : (layer == 1 ? 256 : (layer == 2 ? k4 : 512));
    const int K = layer == 0 ? 64 : (layer == 1 ? 256 : 512);     // hidden input features (rows with a factor)
    const int N = layer == 0 ? 256 : (layer == 3 ? 256 : 512);
    float* c = meta + fm_meta_off(layer);
    float l1p = 0.f, l1f = 0.f, cmax = 0.f;
    // column maxima first (parked in c[]), then the layer's largest: a nearly dead column (maximum 2^-40 of its
    // neighbours', with an ordinary bias) must not get c ~ 2^40 -- b c would dominate the activation bounds and flush
    // every other feature of the tile, and the next layer's rows W / c would fall below f16's floor.  The
    // factor of a column is that of max(its maximum, 2^-20 of the layer's largest): columns within 2^20 of the largest
    // are equalised exactly as before, the others keep |W~| < 1 (still >= 19 bits beside a 2^20-fold bias term)
    float mloc = 0.f;
    for (int f = threadIdx.x; f < N; f += 256) {
      float m = 0.f;
      for (int k = 0; k < Kall; ++k) m = fmaxf(m, fabsf(w[(size_t)k * N + f]) * (k < K ? rrow[k] : 1.0f));
      c[f] = m;
      mloc = fmaxf(mloc, m);
    }
    __syncthreads();
    red[threadIdx.x] = mloc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
      __syncthreads();
    }
    const float mfloor = red[0] * fm::kColFloor;
    __syncthreads();
    for (int f = threadIdx.x; f < N; f += 256) {
      const float cf = pow2_scale_for(fmaxf(c[f], mfloor), 0);   // m cf in [1, 2) (< 1 below the floor); 1 for an all-zero layer
      c[f] = cf;
      cmax = fmaxf(cmax, cf);
      float a = 0.f, af = 0.f;
      for (int k = 0; k < K; ++k) a += fabsf(w[(size_t)k * N + f]) * rrow[k];
      for (int k = K; k < Kall; ++k) af += fabsf(w[(size_t)k * N + f]);
      l1p = fmaxf(l1p, a * cf);
      l1f = fmaxf(l1f, af * cf);
    }
    for (int q = 0; q < 3; ++q) {   // workgroup maxima of the two 1-norms and of c
      __syncthreads();
      red[threadIdx.x] = q == 0 ? l1p : (q == 1 ? l1f : cmax);
      __syncthreads();
      for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
      }
      if (threadIdx.x == 0) {   // summation-order slack on the 1-norms: the bounds must hold
        if (q == 0 && layer == 1) meta[9] = red[0] * 1.0001f;
        if (q == 0 && layer == 2) meta[8] = red[0] * 1.0001f;
        if (q == 1 && layer == 2) meta[10] = red[0] * 1.0001f;
        if (q == 2 && layer == 2) meta[11] = red[0];
      }
    }
    __syncthreads();
    for (int f = threadIdx.x; f < N; f += 256) rrow[f] = 1.0f / c[f];   // the next layer's row factors
    __syncthreads();
  }
}

// pair index -> layer (0: conv2, 1: conv3, 2: fold2/conv1, 3: fold2/conv2), output tile, reduction block
__host__ __device__ inline void fm_pair_coords(int p, int& layer, int& nt, int& kb, bool feat = false) {
  if (p < fm::kPairsL2) { layer = 0; nt = p >> 2; kb = p & 3; return; }
  p -= fm::kPairsL2;
  if (p < 16 * fm::kPairsA) {  // phase A, iteration it = output tile of conv3
    const int it = p / fm::kPairsA, r = p - it * fm::kPairsA;
    if (r < 16) { layer = 1; nt = it; kb = r; return; }
    layer = 2; nt = (r - 16) & 15; kb = 2 * it + ((r - 16) >> 4);
    return;
  }
  p -= 16 * fm::kPairsA;
  if (feat) {  // phase A2, iteration it2: reduction blocks 32 + 6 it2 + {0..5}, two at a time for the 16 output tiles
    if (p < 16 * fm::kPairsA2) {
      const int it2 = p / fm::kPairsA2, r = p - it2 * fm::kPairsA2;
      layer = 2; nt = r & 15; kb = 32 + 6 * it2 + 2 * (r >> 5) + ((r >> 4) & 1);
      return;
    }
    p -= 16 * fm::kPairsA2;
  }
  // phase B, iteration it = output tile of fold2/conv1
  layer = 3; nt = p & 7; kb = 2 * (p >> 4) + ((p >> 3) & 1);
}

// k4 > 512: the FEAT image (w4 = the whole [k4][512] fold2/conv1 matrix; rows >= k4 of the padded 2048 are zeros).
// Slot order of the FEATURE blocks (kb >= 32): natural -- lane (i, g), slot t <-> row 16 kb + 8 g + t: the B operand
// comes from memory there (eight consecutive channels per lane), not from an MFMA's C layout
__global__ __launch_bounds__(256) void fm_pack_kernel(const float* __restrict__ w2, const float* __restrict__ w3,
                                                      const float* __restrict__ w4, const float* __restrict__ w5,
                                                      const float* __restrict__ meta, _Float16* __restrict__ image,
                                                      int k4) {
  const bool feat = k4 > 512;
  const int idx = blockIdx.x * 256 + threadIdx.x;  // (pair, lane)
  if (idx >= (feat ? fm::kPairsFeat : fm::kPairs) * 64) return;
  const int p = idx >> 6, lane = idx & 63;
  int layer, nt, kb;
  fm_pair_coords(p, layer, nt, kb, feat);
  const float* w = layer == 0 ? w2 : (layer == 1 ? w3 : (layer == 2 ? w4 : w5));
  const int N = layer == 0 ? 256 : (layer == 3 ? 256 : 512);
  const int i = lane & 31, g = lane >> 5;
  const float s = meta[fm_meta_off(layer) + 32 * nt + i] * 8192.0f;   // c_l[f] 2^13
  h8 hi, lo;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int k = (layer == 2 && kb >= 32) ? 16 * kb + 8 * g + t : 16 * kb + (t & 3) + 8 * (t >> 2) + 4 * g;  // phi(kb, g, t)
    // row factor 1 / c_{l-1}[k] of a hidden input feature (layer 0's input and the gathered features have none)
    const float rk = (layer == 0 || k >= 512) ? 1.0f : 1.0f / meta[fm_meta_off(layer - 1) + k];
    const float v = (layer == 2 && k >= k4) ? 0.f : w[(size_t)k * N + 32 * nt + i] * rk * s;
    const _Float16 h = (_Float16)v;
    hi[t] = h;
    lo[t] = (_Float16)(v - (float)h);
  }
  h8* out = reinterpret_cast<h8*>(image);
  out[((size_t)p * 2) * 64 + lane] = hi;
  out[((size_t)p * 2 + 1) * 64 + lane] = lo;
}

size_t mlp_fused_image_bytes() { return fm::kImageBytes + fm::kMetaFloats * sizeof(float) + 64; }
size_t mlp_fused_feat_image_bytes() { return fm::kImageBytesFeat + fm::kMetaFloats * sizeof(float) + 64; }

hipError_t mlp_fused_pack_launch(const float* w2, const float* w3, const float* w4_point, const float* w5,
                                 void* image, hipStream_t st) {
  float* meta = reinterpret_cast<float*>(static_cast<char*>(image) + fm::kImageBytes);
  hipLaunchKernelGGL(fm_meta_kernel, dim3(1), dim3(256), 0, st, w2, w3, w4_point, w5, meta, 512);
  hipLaunchKernelGGL(fm_pack_kernel, dim3((fm::kPairs * 64 + 255) / 256), dim3(256), 0, st, w2, w3, w4_point, w5,
                     meta, reinterpret_cast<_Float16*>(image), 512);
  return hipGetLastError();
}
// the FEAT image of the local stream: w4 = the whole fold2/conv1 matrix [512 + 1472][512]
hipError_t mlp_fused_feat_pack_launch(const float* w2, const float* w3, const float* w4, const float* w5, void* image,
                                      hipStream_t st) {
  float* meta = reinterpret_cast<float*>(static_cast<char*>(image) + fm::kImageBytesFeat);
  hipLaunchKernelGGL(fm_meta_kernel, dim3(1), dim3(256), 0, st, w2, w3, w4, w5, meta, 512 + fm::kFeatReal);
  hipLaunchKernelGGL(fm_pack_kernel, dim3((fm::kPairsFeat * 64 + 255) / 256), dim3(256), 0, st, w2, w3, w4, w5, meta,
                     reinterpret_cast<_Float16*>(image), 512 + fm::kFeatReal);
  return hipGetLastError();
}

// max |x| over n floats -> max over out[0 .. slots) (non-negative floats: integer atomicMax on the bit pattern, one
// per workgroup; same-address atomics serialise in L2 at ~10 ns each, hence few workgroups and optional slots).
// out[0 .. slots) must be zeroed first (the launchers do)
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, size_t n4, float* __restrict__ out,
                                                   int slots, size_t out_stride) {
  // blockIdx.y: image -- n4 float4 each, back to back; its slots at out + image * out_stride
  x += (size_t)blockIdx.y * n4 * 4;
  out += (size_t)blockIdx.y * out_stride;
  __shared__ float red[4];
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    m = fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    atomicMax(reinterpret_cast<unsigned*>(out) + (blockIdx.x & (slots - 1)), __float_as_uint(m));
  }
}

static hipError_t amax_go(const float* x, size_t n, float* out, int slots, hipStream_t st, bool clear = true,
                          int images = 1, size_t out_stride = 0) {
  if (clear) {
    const hipError_t e = hipMemsetAsync(out, 0, slots * sizeof(float), st);
    if (e != hipSuccess) return e;
  }
  const size_t n4 = n / 4;
  const int grid = (int)(n4 < 512 * 256 ? (n4 + 255) / 256 : 512);
  hipLaunchKernelGGL(amax_kernel, dim3(grid > 0 ? grid : 1, images), dim3(256), 0, st, x, n4, out, slots, out_stride);
  return hipGetLastError();
}
hipError_t amax_launch(const float* x, size_t n, float* out, hipStream_t st) { return amax_go(x, n, out, 1, st); }
// the 64-slot form the conv_h2 kernels read (max over the slots = max |x|)
hipError_t amax64_launch(const float* x, size_t n, float* out64, hipStream_t st) { return amax_go(x, n, out64, 64, st); }
hipError_t amax64_accumulate_launch(const float* x, size_t n, float* out64, hipStream_t st, int images,
                                    size_t out_stride) {
  return amax_go(x, n, out64, 64, st, false, images, out_stride);
}

__global__ void amax_fold_kernel(const float* __restrict__ slots, int n, int groups, int stride, float* __restrict__ out) {
  float m = 0.f;
  for (int gi = 0; gi < groups; ++gi)
    for (int i = threadIdx.x; i < n; i += 64) m = fmaxf(m, slots[(size_t)gi * stride + i]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if (threadIdx.x == 0) out[0] = m;
}
hipError_t amax_fold_launch(const float* slots, float* out, hipStream_t st, int n, int groups, int stride) {
  hipLaunchKernelGGL(amax_fold_kernel, dim3(1), dim3(64), 0, st, slots, n, groups, stride, out);
  return hipGetLastError();
}

struct TapSlots { const float* p[5]; size_t stride; };
__global__ void tap_amax_kernel(const TapSlots t, float* __restrict__ out) {
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < 5; ++k) m = fmaxf(m, t.p[k][(size_t)blockIdx.x * t.stride + threadIdx.x]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if (threadIdx.x == 0) out[blockIdx.x] = m;
}
hipError_t tap_amax_launch(const float* const slots[5], size_t image_stride, int B, float* out, hipStream_t st) {
  TapSlots t;
  for (int k = 0; k < 5; ++k) t.p[k] = slots[k];
  t.stride = image_stride;
  hipLaunchKernelGGL(tap_amax_kernel, dim3(B), dim3(64), 0, st, t, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------------
struct FusedDev {
  const unsigned char* image;  // packed stream image + meta
  const float *w1, *b1, *b2, *b3, *b4, *b5, *w6, *b6;  // b4: local stream: bias [512]; global: folded per-image bias row
  const float* pts;      // [n][3] projected (local stream only); unused in grid mode
  const float* pts_rot;  // [n][3] MLP input; nullptr: grid mode
  GridSpec grid;         // grid mode: point k0 + i of the (R+1)^3 grid (create_sdf.py:246-256)
  long long k0;
  long long n;
  const float* T;          // trans_mat [4][3] of this image, device memory (local)
  const float* pmap;       // [137*137][512] (local)
  const float* pmap_amax;  // max |pmap| (local)
  const float* add_in;     // per-point sums of the other stream (added before the division), or nullptr
  float* out;
  float out_div;
  // rows of several images in one launch (image-major, rows_per_image a multiple of 128; 0: one image): the global
  // stream takes image b's bias row b4 + 512 b, the FEAT form image b's feature maximum
  long long rows_per_image;
  // FEAT form: gathered features in SPLIT form (project_gather_taps_kernel with a scale): row p = feat_ld floats' worth
  // of bytes, every 8 channels as [h8 | l8] (f16 planes of feature * feat_split_scale(feat_amax[image]))
  const unsigned char* feat;
  const float* feat_amax;  // [images] max |tap| of the image: the bound of every gathered feature
  int feat_ld;             // 1536
};

#define FM_FENCE() asm volatile("" ::: "memory")

// The MFMAs are inline asm so that the register FILE of every accumulator is chosen here: the 16 output
// tiles of fold2/conv1 take all 256 accumulation VGPRs ("a"), every other accumulator is an arch VGPR
// ("v").  (hipcc's AGPR-form MFMA wants EVERY accumulator in the 256 AGPRs; 272 are live in phase A.)
// hipcc neither schedules nor pads around an asm statement (cdna_hip_programming.md 5.7): volatile keeps
// the statements in program order -- the weight prefetch below is therefore written out by hand -- and
// the MFMA -> VALU / VALU -> MFMA wait states are the FM_SETTLE* statements, tied to the registers by
// operands so that no reader can be scheduled above them.
#define FM_MFMA(CLS, ACC, A, B) \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+" CLS(ACC) : "v"(A), "v"(B))
// first MFMA of a chain: C = 0 (no zeroed accumulator to keep around)
#define FM_MFMA0(CLS, ACC, A, B) \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&" CLS(ACC) : "v"(A), "v"(B))
// an MFMA result about to be read by compiler-generated code: >= 19 wait states (16-pass XDL)
#define FM_SETTLE_ACC(CLS, ACC) asm volatile("s_nop 15\n\ts_nop 7" : "+" CLS(ACC))
// VALU-written registers about to be read by an MFMA
#define FM_SETTLE_IN4(A, B, C, D) asm volatile("s_nop 3" : "+v"(A), "+v"(B), "+v"(C), "+v"(D))
#define FM_SETTLE_IN1(CLS, A) asm volatile("s_nop 3" : "+" CLS(A))

// LDS-DMA as inline asm: hipcc would otherwise drain every LDS-DMA (s_waitcnt vmcnt(0)) in front of any
// ds_read it cannot prove disjoint -- here the bias reads of every tile.  M0 = LDS byte address of lane
// 0's 16 bytes; the instruction offset applies to the global AND the LDS address.
__device__ __forceinline__ unsigned fm_lds_addr(const void* p) {
  return (unsigned)(unsigned long)((const __attribute__((address_space(3))) void*)p);
}
// four consecutive 1-KiB pieces: global (sbase + voff) + 1024 q -> LDS dst + 1024 q (+ 16 lane)
__device__ __forceinline__ void fm_glds_4k(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}
// four pieces of one resampler tap of output tile NT: global (sbase + voff) + 128 NT + 32 q bytes -> LDS
// dst + 1024 q (+ 16 lane).  The instruction offset moves the LDS address too: M0 is set back by it.
template <int NT>
__device__ __forceinline__ void fm_glds_tap(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:%4\n\t"
      "s_add_u32 m0, m0, 992\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:%5\n\t"
      "s_add_u32 m0, m0, 992\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:%6\n\t"
      "s_add_u32 m0, m0, 992\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:%7\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst - 128u * NT), "n"(128 * NT), "n"(128 * NT + 32), "n"(128 * NT + 64),
        "n"(128 * NT + 96)
      : "memory", "scc");
}
// s_waitcnt vmcnt(m) for a wave-uniform run-time n, m = the largest of {40, 24, 8, 0} that is <= n: waiting
// for FEWER outstanding loads than necessary is always safe, and these are the counts the steady state
// produces (4 loads per ring request, 16 per gather request)
__device__ __forceinline__ void fm_wait_vm(int n) {
  if (n >= 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
  else if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// the FEAT form's counts (4 per ring request, 12 per feature request: every multiple of 4 up to 60 occurs)
__device__ __forceinline__ void fm_wait_vm4(int n) {
  switch (n >= 60 ? 15 : (n >> 2)) {
    case 15: asm volatile("s_waitcnt vmcnt(60)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(52)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
// the six reduction blocks of one phase-A2 iteration for this wave's 32 points: lane (j, g) takes, per block b, the 32
// bytes [h8 | l8] of channels 16 b + 8 g .. + 7 of its row -- global (sbase + voff) + 64 b (+ 16: the l plane) -> LDS
// dst + 1024 (2 b + plane) (+ 16 lane).  The instruction offset moves the LDS address too: M0 is stepped so that the
// twelve pieces land 1 KiB apart (piece 2b at M0 = dst + 2048 b - 64 b, piece 2b + 1 at M0 = dst + 2048 b + 1024 - 64 b - 16)
__device__ __forceinline__ void fm_glds_feat(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_add_u32 m0, m0, 1008\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:16\n\t"
      "s_add_u32 m0, m0, 976\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:64\n\t"
      "s_add_u32 m0, m0, 1008\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:80\n\t"
      "s_add_u32 m0, m0, 976\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:128\n\t"
      "s_add_u32 m0, m0, 1008\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:144\n\t"
      "s_add_u32 m0, m0, 976\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:192\n\t"
      "s_add_u32 m0, m0, 1008\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:208\n\t"
      "s_add_u32 m0, m0, 976\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:256\n\t"
      "s_add_u32 m0, m0, 1008\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:272\n\t"
      "s_add_u32 m0, m0, 976\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:320\n\t"
      "s_add_u32 m0, m0, 1008\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:336\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory", "scc");
}

// projection + resampler corner weights exactly as elementwise.hip (project_point / sample4): no
// contraction, so the pixels and weights are those of the unfused path
#pragma clang fp contract(off)
__device__ __forceinline__ void fm_project_taps(const float* T, float x, float y, float z, int (&pix)[4],
                                                float (&wt)[4]) {
  float p[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float a = x * T[0 * 3 + j] + y * T[1 * 3 + j];
    a = a + z * T[2 * 3 + j];
    p[j] = a + T[3 * 3 + j];
  }
  float px = p[0] / p[2], py = p[1] / p[2];
  px = (px != px) ? px : fminf(136.0f, fmaxf(0.0f, px));
  py = (py != py) ? py : fminf(136.0f, fmaxf(0.0f, py));
  const bool ok = px > -1.0f && py > -1.0f && px < 137.0f && py < 137.0f;
  const float fx = floorf(px), fy = floorf(py);
  const float cx = fx + 1.0f, cy = fy + 1.0f;
  const float dx = cx - px, dy = cy - py;
  const int ifx = ok ? (int)fx : 0, ify = ok ? (int)fy : 0, icx = ok ? (int)cx : 0, icy = ok ? (int)cy : 0;
  const bool xf = ifx >= 0 && ifx < 137, xc = icx >= 0 && icx < 137;
  const bool yf = ify >= 0 && ify < 137, yc = icy >= 0 && icy < 137;
  const int cfx = min(max(ifx, 0), 136), cfy = min(max(ify, 0), 136);
  const int ccx = min(max(icx, 0), 136), ccy = min(max(icy, 0), 136);
  // order ff, cc, fc, cf (sample4's summation order)
  pix[0] = (cfy * 137 + cfx) * 512; wt[0] = (ok && xf && yf) ? dx * dy : 0.f;
  pix[1] = (ccy * 137 + ccx) * 512; wt[1] = (ok && xc && yc) ? (1.0f - dx) * (1.0f - dy) : 0.f;
  pix[2] = (ccy * 137 + cfx) * 512; wt[2] = (ok && xf && yc) ? dx * (1.0f - dy) : 0.f;
  pix[3] = (cfy * 137 + ccx) * 512; wt[3] = (ok && xc && yf) ? (1.0f - dx) * dy : 0.f;
}
#pragma clang fp contract(fast)

__device__ __forceinline__ float fm_exp2i(int e) {  // 2^e, e in [-126, 127]
  return __uint_as_float((unsigned)(e + 127) << 23);
}
// exponent e of m (m < 2^(e+1)), clamped so that 2^(14-e) and 2^(e-14) are normal floats
__device__ __forceinline__ int fm_exp_of(float m) {
  const int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;
  return min(max(e, -100), 100);
}

// bias + ReLU + scale + two-term split of one output tile -> the two reduction blocks it becomes
template <bool GATHER>
__device__ __forceinline__ void fm_tile_to_frags(const f32x16& acc, const float* bias32 /* LDS, + 4g applied */,
                                                 const float* c32 /* LDS, + 4g applied: c_4 of the tile's features (GATHER) */,
                                                 const unsigned char* grows /* LDS: 16 pieces, + 16 lane applied */,
                                                 const float (&wt)[4], float inv, float s, h8 (&fh)[2],
                                                 h8 (&fl)[2]) {
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) {
    const float4 bb = *reinterpret_cast<const float4*>(bias32 + 8 * rq);
    float b4[4] = {bb.x, bb.y, bb.z, bb.w};
    if (GATHER) {  // + the four resampled pmap rows (sample4's summation order: ff, cc, fc, cf)
      float gs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tp = 0; tp < 4; ++tp) {
        const float4 v = *reinterpret_cast<const float4*>(grows + (tp * 4 + rq) * 1024);
        gs[0] = fmaf(wt[tp], v.x, gs[0]);
        gs[1] = fmaf(wt[tp], v.y, gs[1]);
        gs[2] = fmaf(wt[tp], v.z, gs[2]);
        gs[3] = fmaf(wt[tp], v.w, gs[3]);
      }
      const float4 cc = *reinterpret_cast<const float4*>(c32 + 8 * rq);   // the equalised feature: pmap row f times c_4[f]
      b4[0] = fmaf(gs[0], cc.x, b4[0]);
      b4[1] = fmaf(gs[1], cc.y, b4[1]);
      b4[2] = fmaf(gs[2], cc.z, b4[2]);
      b4[3] = fmaf(gs[3], cc.w, b4[3]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int r = 4 * rq + c;
      const float v = fmaxf(fmaf(acc[r], inv, b4[c]), 0.f) * s;
      const _Float16 h = (_Float16)v;
      fh[r >> 3][r & 7] = h;
      fl[r >> 3][r & 7] = (_Float16)(v - (float)h);
    }
  }
}

template <int N, class F, int... I>
__device__ __forceinline__ void fm_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void fm_static_for(F&& f) {
  fm_static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// FEAT (with LOCAL): the local stream of a small point set -- no folded map; the gathered features (split form, one
// scale per image) are 96 more reduction blocks of fold2/conv1 (phase A2), see fm::kPairsA2
template <bool LOCAL, bool SAFE, bool FEAT = false>
__global__ __launch_bounds__(256, 1) void mlp_fused_kernel(const FusedDev P) {
  static_assert(!FEAT || LOCAL, "the FEAT form is the local stream's");
  constexpr int R = FEAT ? 6 : (LOCAL ? 5 : 9);  // ring slots
  constexpr int DP = R - 2;         // a slot is requested DP syncs before the sync that waits for it
  constexpr int GBUF = FEAT ? 12288 : (LOCAL ? 16384 : 0);   // per wave: four pmap rows of a tile / six feature blocks
  constexpr int KP = FEAT ? fm::kPairsFeat : fm::kPairs;     // pairs per tile
  constexpr int KSLOTS = KP / 8;
  constexpr int QB = fm::kPairsL2 + 16 * fm::kPairsA + (FEAT ? 16 * fm::kPairsA2 : 0);   // first pair of phase B
  constexpr int RING_BYTES = R * fm::kSlotBytes;
  constexpr int CONST_OFF = RING_BYTES + 4 * GBUF;
  constexpr int LDS_BYTES = CONST_OFF + fm::kConstFloats * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, g = lane >> 5;
  float* const cst = reinterpret_cast<float*>(&lds[CONST_OFF]);
  const int gbuf = RING_BYTES + wave * GBUF;  // byte offset of this wave's gather buffer
  const unsigned lds0 = fm_lds_addr(lds);
  const float* meta = reinterpret_cast<const float*>(P.image + (FEAT ? fm::kImageBytesFeat : fm::kImageBytes));

  // ---- constants -> LDS -------------------------------------------------------------------------------
  for (int i = tid; i < 192; i += 256) cst[fm::cW1 + i] = P.w1[i];
  if (tid < 64) cst[fm::cB1 + tid] = P.b1[tid];
  // (the equalised form: bias_l c_l, w6 / c_5; the c_4 row stays in LDS for the folded map's rows)
  cst[fm::cB2 + tid] = P.b2[tid] * meta[fm::mS2 + tid];
  for (int i = tid; i < 512; i += 256) {
    cst[fm::cB3 + i] = P.b3[i] * meta[fm::mS3 + i];
    cst[fm::cB4 + i] = P.b4[i] * meta[fm::mS4 + i];
    cst[fm::cS4 + i] = meta[fm::mS4 + i];
  }
  cst[fm::cB5 + tid] = P.b5[tid] * meta[fm::mS5 + tid];
  cst[fm::cW6 + tid] = P.w6[tid] / meta[fm::mS5 + tid];
  if (tid == 0) cst[fm::cB6] = P.b6[0];
  const float cw4 = meta[8], cw3 = meta[9], cw4f = FEAT ? meta[10] : 0.f;
  // |pmap row f| c_4[f] <= max |pmap| max c_4
  float addmax4 = (LOCAL && !FEAT) ? P.pmap_amax[0] * meta[11] : 0.f, addmax3 = 0.f;
  float T[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) T[i] = (LOCAL && !FEAT) ? P.T[i] : 0.f;
  __syncthreads();
  {  // largest |bias| of conv3, largest |additive term| of fold2/conv1 (bias, or the folded per-image bias)
    float m3 = 0.f, m4 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      m3 = fmaxf(m3, fabsf(cst[fm::cB3 + lane + 64 * i]));
      m4 = fmaxf(m4, fabsf(cst[fm::cB4 + lane + 64 * i]));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      m3 = fmaxf(m3, __shfl_xor(m3, off));
      m4 = fmaxf(m4, __shfl_xor(m4, off));
    }
    addmax3 = m3;
    addmax4 += m4;
  }
  const float addmax4_bias = addmax4;   // FEAT: max |b4| (the feature term is added per image)

  // ---- weight ring + the ledger of this wave's LDS-DMA loads ------------------------------------------------
  // Loads retire in order, so "the loads of batch X have landed" == "at most (issued - issued_after_X)
  // loads are outstanding".  `issued` counts every LDS-DMA instruction of this wave; mark[] is the FIFO of
  // issued_after_X for the ring slots in flight (the next sync waits for mark[0]); gmark the same for the
  // gather rows.  All of it is wave-uniform scalar arithmetic; fm_wait_vm turns the difference into the
  // immediate of s_waitcnt.  (Never vmcnt(0) in steady state: the slots DP-1 syncs ahead stay in flight.)
  int issued = 0, gmark = 0;
  int mark[DP + 1];
  int issue_slot = 0, issue_pos = 0;  // slot within the image / ring position of the next slot to request
  const unsigned img_voff = wave * 4096 + lane * 16;
  auto ring_issue = [&](int mi) {  // this wave's pieces 4*wave .. 4*wave+3 of the next slot
    fm_glds_4k(P.image + (size_t)issue_slot * fm::kSlotBytes, img_voff, lds0 + issue_pos * fm::kSlotBytes + wave * 4096);
    issued += 4;
    mark[mi] = issued;
    issue_slot = issue_slot + 1 == KSLOTS ? 0 : issue_slot + 1;
    issue_pos = issue_pos + 1 == R ? 0 : issue_pos + 1;
  };
  int pix[4] = {0, 0, 0, 0};  // byte offsets of this lane's 16 bytes in the four pmap rows (tile 0)
  float wt[4] = {0.f, 0.f, 0.f, 0.f};
  // the four resampled pmap rows of output tile NT: 16 B per lane and piece (NT is a type: compile time)
  auto gather_issue = [&](auto nt_c) {
    if constexpr (LOCAL && !FEAT) {
      constexpr int NT = decltype(nt_c)::value;
#pragma unroll
      for (int tp = 0; tp < 4; ++tp) fm_glds_tap<NT>(P.pmap, (unsigned)pix[tp], lds0 + gbuf + tp * 4096);
      issued += 16;
      gmark = issued;
    }
  };
  auto gather_wait = [&]() {
    FM_FENCE();
    if (SAFE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (FEAT) fm_wait_vm4(issued - gmark);
    else fm_wait_vm(issued - gmark);
    FM_FENCE();
  };
  // FEAT: the six feature blocks of phase-A2 iteration it2 of the tile whose first row is row0 -> this wave's buffer
  unsigned feat_voff = 0;   // this lane's row within the tile (clamped to the last row of the launch) and 32-byte half
  auto feat_issue = [&](long long row0, int it2) {
    if constexpr (FEAT) {
      fm_glds_feat(P.feat + (size_t)row0 * (size_t)(P.feat_ld * 4) + (size_t)it2 * 384, feat_voff, lds0 + gbuf);
      issued += 12;
      gmark = issued;
    }
  };
  // The sync for a slot runs in the MIDDLE of the slot before it (after pair 3 of 8), so that the first
  // fragments of the next slot can be read ahead of the last MFMAs of the current one: wait for this
  // wave's quarter of the slot, barrier (publishes it; every wave is past the slot before the current
  // one, whose position is free), request the slot DP ahead into that position.
  int cons_pos = 0;                    // ring position of the slot the next sync publishes
  int sa[2] = {lane * 16, lane * 16};  // this lane's byte address in the slots of even / odd number
  auto sync_slot = [&](int parity, bool issue) {
    FM_FENCE();
    if (SAFE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (FEAT) fm_wait_vm4(issued - mark[0]);
    else fm_wait_vm(issued - mark[0]);
    __builtin_amdgcn_s_barrier();
    FM_FENCE();
    // the FIFO holds the DP slots in flight after a sync (DP + 1 only between the prologue and the first sync)
#pragma unroll
    for (int i = 0; i < DP; ++i) mark[i] = mark[i + 1];
    if (issue) ring_issue(DP - 1);
    sa[parity] = cons_pos * fm::kSlotBytes + lane * 16;
    cons_pos = cons_pos + 1 == R ? 0 : cons_pos + 1;
  };
#pragma unroll
  for (int i = 0; i <= DP; ++i) ring_issue(i);  // slots 0 .. DP
  sync_slot(0, false);                          // slot 0 (the only sync outside a slot)

  // weight fragments, three register sets: pair q uses set q % 3, the pair two ahead is read meanwhile
  h8 wh[3], wl[3];
#define FM_LDW(Q)                                                                                   \
  {                                                                                                 \
    wh[(Q) % 3] = *reinterpret_cast<const h8*>(&lds[sa[((Q) >> 3) & 1] + ((Q)&7) * 2048]);         \
    wl[(Q) % 3] = *reinterpret_cast<const h8*>(&lds[sa[((Q) >> 3) & 1] + ((Q)&7) * 2048 + 1024]);  \
  }
  // pair Q of the stream (Q: position within the tile's 1056 pairs; compile-time after unrolling):
  // read pair Q+2, three MFMAs (small terms first), and after pair 3 of a slot the sync for the next slot
#define FM_STEP_(Q, CLS, ACC, XH, XL, FIRST)                 \
  {                                                          \
    if ((Q) + 2 < KP) FM_LDW((Q) + 2);                       \
    if (FIRST) FM_MFMA0(CLS, ACC, wl[(Q) % 3], XH);          \
    else FM_MFMA(CLS, ACC, wl[(Q) % 3], XH);                 \
    FM_MFMA(CLS, ACC, wh[(Q) % 3], XL);                      \
    FM_MFMA(CLS, ACC, wh[(Q) % 3], XH);                      \
    if (((Q)&7) == 3) sync_slot((((Q) >> 3) + 1) & 1, true); \
  }
#define FM_STEP(Q, CLS, ACC, XH, XL) FM_STEP_(Q, CLS, ACC, XH, XL, false)
#define FM_STEP0(Q, CLS, ACC, XH, XL, FIRST) FM_STEP_(Q, CLS, ACC, XH, XL, FIRST)

  const long long ntiles = (P.n + 127) >> 7;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // ---- this lane's point -------------------------------------------------------------------------
    const long long pidx = tile * 128 + wave * 32 + j;
    const bool valid = pidx < P.n;
    const long long pc = valid ? pidx : P.n - 1;
    // ---- this tile's image (rows of several images in one launch) ---------------------------------------
    float sfeat = 1.0f, inv_sfeat = 1.0f;
    if (P.rows_per_image > 0 || FEAT) {
      const int img = P.rows_per_image > 0 ? (int)((tile * 128) / P.rows_per_image) : 0;
      if (!LOCAL) {  // the global stream's folded bias row of this image (and the bound's additive term)
        __syncthreads();
        for (int i = tid; i < 512; i += 256) cst[fm::cB4 + i] = P.b4[(size_t)img * 512 + i] * cst[fm::cS4 + i];
        __syncthreads();
        float m4 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) m4 = fmaxf(m4, fabsf(cst[fm::cB4 + lane + 64 * i]));
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m4 = fmaxf(m4, __shfl_xor(m4, off));
        addmax4 = m4;
      }
      if (FEAT) {  // every gathered feature is a convex combination of tap values: |feat| <= max |tap| of the image
        const float fmax = feat_split_amax(P.feat_amax[img]);
        sfeat = pow2_scale_for(fmax, 14);
        inv_sfeat = 1.0f / sfeat;
        addmax4 = fmaf(fmax, cw4f, addmax4_bias);
      }
    }
    if (FEAT) {
      feat_voff = (unsigned)(pc - tile * 128) * (unsigned)(P.feat_ld * 4) + (unsigned)g * 32u;
      feat_issue(tile * 128, 0);
    }
    float x, y, z, xp, yp, zp;
    if (P.pts_rot) {
      x = P.pts_rot[pc * 3]; y = P.pts_rot[pc * 3 + 1]; z = P.pts_rot[pc * 3 + 2];
      xp = x; yp = y; zp = z;
      if (LOCAL && !FEAT) { xp = P.pts[pc * 3]; yp = P.pts[pc * 3 + 1]; zp = P.pts[pc * 3 + 2]; }
    } else {  // grid_points_kernel's expression (numpy.linspace in float64, cast to float32)
      const long long k = P.k0 + pc, res = P.grid.res;
      const int idx[3] = {(int)(k % res), (int)((k / res) % res), (int)(k / (res * res))};
      float c[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        double v = (double)idx[a] * P.grid.step[a];
        v = v + P.grid.start[a];
        if (idx[a] == P.grid.res - 1 && P.grid.res > 1) v = P.grid.stop[a];
        c[a] = (float)v;
      }
      x = xp = c[0]; y = yp = c[1]; z = zp = c[2];
    }
    if (LOCAL && !FEAT) {
      fm_project_taps(T, xp, yp, zp, pix, wt);
#pragma unroll
      for (int tp = 0; tp < 4; ++tp) pix[tp] = (pix[tp] + 4 * g) * 4;
    }
    FM_LDW(0);
    FM_LDW(1);

    // ---- fold1/conv1 (3 -> 64, VALU): this lane's 32 of the 64 features in slot order ------------------
    h8 x1h[4], x1l[4];
    float inv2;
    {
      float e1[4][8];
      float m = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int tq = 0; tq < 2; ++tq) {
          const int f0 = 16 * kb + 8 * tq + 4 * g;
          const float4 wa = *reinterpret_cast<const float4*>(&cst[fm::cW1 + f0]);
          const float4 wb = *reinterpret_cast<const float4*>(&cst[fm::cW1 + 64 + f0]);
          const float4 wc = *reinterpret_cast<const float4*>(&cst[fm::cW1 + 128 + f0]);
          const float4 bb = *reinterpret_cast<const float4*>(&cst[fm::cB1 + f0]);
          e1[kb][4 * tq + 0] = fmaxf(x * wa.x + y * wb.x + z * wc.x + bb.x, 0.f);
          e1[kb][4 * tq + 1] = fmaxf(x * wa.y + y * wb.y + z * wc.y + bb.y, 0.f);
          e1[kb][4 * tq + 2] = fmaxf(x * wa.z + y * wb.z + z * wc.z + bb.z, 0.f);
          e1[kb][4 * tq + 3] = fmaxf(x * wa.w + y * wb.w + z * wc.w + bb.w, 0.f);
#pragma unroll
          for (int c = 0; c < 4; ++c) m = fmaxf(m, e1[kb][4 * tq + c]);
        }
      m = fmaxf(m, __shfl_xor(m, 32));
      const int e = fm_exp_of(m);
      const float s = fm_exp2i(14 - e);
      inv2 = fm_exp2i(e - 14) * fm::kInvSw;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float v = e1[kb][t] * s;
          const _Float16 h = (_Float16)v;
          x1h[kb][t] = h;
          x1l[kb][t] = (_Float16)(v - (float)h);
        }
      FM_SETTLE_IN4(x1h[0], x1h[1], x1h[2], x1h[3]);
      FM_SETTLE_IN4(x1l[0], x1l[1], x1l[2], x1l[3]);
    }

    // ---- fold1/conv2 (64 -> 256): pairs 0..31; scale from the actual maximum -----------------------------
    h8 x2h[16], x2l[16];
    float inv3, s3, inv4, s4, inv5;
    {
      f32x16 z2[8];
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) FM_STEP0(nt * 4 + kb, "v", z2[nt], x1h[kb], x1l[kb], kb == 0);
      }
      float m = 0.f;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        FM_SETTLE_ACC("v", z2[nt]);
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const float4 bb = *reinterpret_cast<const float4*>(&cst[fm::cB2 + 32 * nt + 8 * rq + 4 * g]);
          const float b4[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float v = fmaxf(fmaf(z2[nt][4 * rq + c], inv2, b4[c]), 0.f);
            z2[nt][4 * rq + c] = v;
            m = fmaxf(m, v);
          }
        }
      }
      m = fmaxf(m, __shfl_xor(m, 32));
      const int e2 = fm_exp_of(m);
      const float s2 = fm_exp2i(14 - e2);
      inv3 = fm_exp2i(e2 - 14) * fm::kInvSw;
      // conv3's and fold2/conv1's outputs are consumed tile by tile: scales from the bounds
      //   |h3| <= max|h2| * max_f ||W3[:,f]||_1 + max|b3|,   |h4| <= that * max_f ||W4[:,f]||_1 + max|additive term|
      const float bound3 = fmaf(m, cw3, addmax3);
      const int e3 = fm_exp_of(bound3);
      s3 = fm_exp2i(14 - e3);
      inv4 = fm_exp2i(e3 - 14) * fm::kInvSw;
      const int e4 = fm_exp_of(fmaf(bound3, cw4, addmax4));
      s4 = fm_exp2i(14 - e4);
      inv5 = fm_exp2i(e4 - 14) * fm::kInvSw;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const float v = z2[nt][8 * hf + t] * s2;
            const _Float16 h = (_Float16)v;
            x2h[2 * nt + hf][t] = h;
            x2l[2 * nt + hf][t] = (_Float16)(v - (float)h);
          }
#pragma unroll
      for (int i = 0; i < 16; i += 2) FM_SETTLE_IN4(x2h[i], x2l[i], x2h[i + 1], x2l[i + 1]);
    }

    // ---- phase A: conv3 (256 -> 512) chained into fold2/conv1 (512 -> 512): 16 iterations x 48 pairs ------
    f32x16 acc4[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc4[i] = f32x16{0.f};
      FM_SETTLE_IN1("a", acc4[i]);
    }
#pragma unroll 1
    for (int it = 0; it < 16; ++it) {
      f32x16 acc;
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) FM_STEP0(32 + kb, "v", acc, x2h[kb], x2l[kb], kb == 0);
      h8 fh[2], fl[2];
      FM_SETTLE_ACC("v", acc);
      fm_tile_to_frags<false>(acc, &cst[fm::cB3 + 32 * it + 4 * g], &cst[fm::cS3 + 32 * it + 4 * g], nullptr, wt, inv3, s3, fh, fl);
      FM_SETTLE_IN4(fh[0], fl[0], fh[1], fl[1]);
#pragma unroll
      for (int r = 0; r < 32; ++r)  // reduction block 2it (r < 16) / 2it+1 for output tile r & 15
        FM_STEP(32 + 16 + r, "a", acc4[r & 15], fh[r >> 4], fl[r >> 4]);
      if (LOCAL && it == 14) gather_issue(std::integral_constant<int, 0>{});  // pmap rows of fold2/conv1's tile 0: one iteration of cover
    }

    // ---- phase A2 (FEAT): the gathered features as 96 more reduction blocks of fold2/conv1 ------------------
    if constexpr (FEAT) {
      // the accumulators so far hold (h3 s3)(W s_w); the features come scaled by the image's s_feat: one exact
      // power-of-two rescale, s_feat / s3, makes the two parts one sum at scale s_feat s_w
      const float rs = sfeat * (1.0f / s3);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        FM_SETTLE_ACC("a", acc4[i]);
        acc4[i] = acc4[i] * rs;
        FM_SETTLE_IN1("a", acc4[i]);
      }
      inv4 = inv_sfeat * fm::kInvSw;
#pragma unroll 1
      for (int it2 = 0; it2 < 16; ++it2) {
        h8 fh6[6], fl6[6];
        gather_wait();   // this iteration's six blocks are in the wave's buffer
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          fh6[b] = *reinterpret_cast<const h8*>(&lds[gbuf + (2 * b) * 1024 + lane * 16]);
          fl6[b] = *reinterpret_cast<const h8*>(&lds[gbuf + (2 * b + 1) * 1024 + lane * 16]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the buffer is read: the next iteration's blocks may land
        if (it2 < 15) feat_issue(tile * 128, it2 + 1);
        fm_static_for<96>([&](auto r_c) {  // reduction block 6 it2 + 2 (r >> 5) + ((r >> 4) & 1) for output tile r & 15
          constexpr int r = decltype(r_c)::value;
          FM_STEP(32 + 768 + r, "a", acc4[r & 15], fh6[2 * (r >> 5) + ((r >> 4) & 1)], fl6[2 * (r >> 5) + ((r >> 4) & 1)]);
        });
      }
    }

    // ---- phase B: fold2/conv1's tiles drained into fold2/conv2 (512 -> 256): 16 iterations x 16 pairs ------
    f32x16 acc5[8];
    fm_static_for<16>([&](auto it_c) {
      constexpr int it = decltype(it_c)::value;
      h8 fh[2], fl[2];
      FM_SETTLE_ACC("a", acc4[it]);
      if (LOCAL && !FEAT) gather_wait();
      fm_tile_to_frags<(LOCAL && !FEAT)>(acc4[it], &cst[fm::cB4 + 32 * it + 4 * g], &cst[fm::cS4 + 32 * it + 4 * g], &lds[gbuf + lane * 16], wt, inv4, s4, fh, fl);
      if (LOCAL && !FEAT && it < 15) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of the gather buffer are done
        gather_issue(std::integral_constant<int, (it < 15 ? it + 1 : 15)>{});
      }
      FM_SETTLE_IN4(fh[0], fl[0], fh[1], fl[1]);
#pragma unroll
      for (int r = 0; r < 16; ++r)
        FM_STEP0(QB + 16 * it + r, "v", acc5[r & 7], fh[r >> 3], fl[r >> 3], it == 0 && r < 8);
    });

    // ---- fold2/conv2 epilogue + fold2/conv5 (256 -> 1): dot with w6 --------------------------------------
    {
      float dot = 0.f;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        FM_SETTLE_ACC("v", acc5[nt]);
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int f0 = 32 * nt + 8 * rq + 4 * g;
          const float4 bb = *reinterpret_cast<const float4*>(&cst[fm::cB5 + f0]);
          const float4 ww = *reinterpret_cast<const float4*>(&cst[fm::cW6 + f0]);
          dot = fmaf(fmaxf(fmaf(acc5[nt][4 * rq + 0], inv5, bb.x), 0.f), ww.x, dot);
          dot = fmaf(fmaxf(fmaf(acc5[nt][4 * rq + 1], inv5, bb.y), 0.f), ww.y, dot);
          dot = fmaf(fmaxf(fmaf(acc5[nt][4 * rq + 2], inv5, bb.z), 0.f), ww.z, dot);
          dot = fmaf(fmaxf(fmaf(acc5[nt][4 * rq + 3], inv5, bb.w), 0.f), ww.w, dot);
        }
      }
      dot += __shfl_xor(dot, 32);
      dot += cst[fm::cB6];
      if (valid && g == 0) {
        if (P.add_in) P.out[pidx] = (P.add_in[pidx] + dot) / P.out_div;
        else if (P.out_div != 1.0f) P.out[pidx] = dot / P.out_div;
        else P.out[pidx] = dot;
      }
    }
  }
  // drain the LDS-DMA still in flight before the workgroup's LDS is released
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// one stream for n points of one image.  local: gather + 'sdfprediction_imgfeat', out = (add_in + sum) / out_div;
// global: 'sdfprediction' with the folded bias row, out = sum
hipError_t mlp_fused_launch(bool local, const void* image, const float* w1, const float* b1, const float* b2,
                            const float* b3, const float* b4, const float* b5, const float* w6, const float* b6,
                            const float* pts, const float* pts_rot, const GridSpec* grid, long long k0,
                            long long n, const float* trans_mat_b, const float* pmap, const float* pmap_amax,
                            const float* add_in, float* out, float out_div, hipStream_t st) {
  FusedDev P{};
  P.image = static_cast<const unsigned char*>(image);
  P.w1 = w1; P.b1 = b1; P.b2 = b2; P.b3 = b3; P.b4 = b4; P.b5 = b5; P.w6 = w6; P.b6 = b6;
  P.pts = pts; P.pts_rot = pts_rot;
  if (grid) P.grid = *grid;
  P.k0 = k0; P.n = n;
  P.T = trans_mat_b;
  P.pmap = pmap; P.pmap_amax = pmap_amax; P.add_in = add_in; P.out = out; P.out_div = out_div;
  const long long tiles = (n + 127) / 128;
  const int grid_x = (int)(tiles < 256 ? tiles : 256);
  if (local) {
    if (tune::fused_safe) hipLaunchKernelGGL((mlp_fused_kernel<true, true>), dim3(grid_x), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((mlp_fused_kernel<true, false>), dim3(grid_x), dim3(256), 0, st, P);
  } else {
    if (tune::fused_safe) hipLaunchKernelGGL((mlp_fused_kernel<false, true>), dim3(grid_x), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((mlp_fused_kernel<false, false>), dim3(grid_x), dim3(256), 0, st, P);
  }
  return hipGetLastError();
}

// Both streams of a SMALL point set -- `images` images x rows_per_image points (image-major; rows_per_image a multiple
// of 128) -- in one launch each (round 4): the GLOBAL stream with image b's folded bias row gbias + 512 b, out = its
// sums (add_in = nullptr) or (add_in + sum) / out_div; the LOCAL stream in the FEAT form on the split-form gathered
// features (project_gather_taps_launch with split_amax) -- no folded map, activations never leave the registers.
hipError_t mlp_fused_small_launch(bool local, const void* image, const float* w1, const float* b1, const float* b2,
                                  const float* b3, const float* b4, const float* b5, const float* w6, const float* b6,
                                  const float* pts_rot, long long rows_per_image, int images, const void* feat_split,
                                  int feat_ld, const float* feat_amax, const float* add_in, float* out, float out_div,
                                  hipStream_t st) {
  if (rows_per_image <= 0 || rows_per_image % 128 || images <= 0) return hipErrorInvalidValue;
  if (local && (!feat_split || !feat_amax || feat_ld != fm::kFeatCols)) return hipErrorInvalidValue;
  FusedDev P{};
  P.image = static_cast<const unsigned char*>(image);
  P.w1 = w1; P.b1 = b1; P.b2 = b2; P.b3 = b3; P.b4 = b4; P.b5 = b5; P.w6 = w6; P.b6 = b6;
  P.pts_rot = pts_rot;
  P.n = rows_per_image * images;
  P.rows_per_image = rows_per_image;
  P.add_in = add_in; P.out = out; P.out_div = out_div;
  P.feat = static_cast<const unsigned char*>(feat_split); P.feat_amax = feat_amax; P.feat_ld = feat_ld;
  const long long tiles = P.n / 128;
  const int grid_x = (int)(tiles < 256 ? tiles : 256);
  if (local) {
    if (tune::fused_safe) hipLaunchKernelGGL((mlp_fused_kernel<true, true, true>), dim3(grid_x), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((mlp_fused_kernel<true, false, true>), dim3(grid_x), dim3(256), 0, st, P);
  } else {
    if (tune::fused_safe) hipLaunchKernelGGL((mlp_fused_kernel<false, true>), dim3(grid_x), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((mlp_fused_kernel<false, false>), dim3(grid_x), dim3(256), 0, st, P);
  }
  return hipGetLastError();
}

}  // namespace disn
