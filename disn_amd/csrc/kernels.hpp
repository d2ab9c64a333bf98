// Internal launcher declarations shared by the .hip translation units.
// Public C ABI: include/disn_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// disn_ctx_t of include/disn_amd.h: the auxiliary stream and the fork/join events of one caller stream
struct disn_ctx {
  hipStream_t aux;
  hipEvent_t ev[10];
  // software pipeline of consecutive steps (disn_ctx_pipeline): the convolution stack of this step starts behind
  // `pipe_wait` (the previous step's `pipe_record`) and records `pipe_record` when it is done
  hipEvent_t pipe_wait = nullptr, pipe_record = nullptr;
};

namespace disn {

// Non-temporal 16-byte access for data that is streamed exactly once (fc weights, optimizer state):
// measured 4.4 -> 5.7 TB/s on the 411 MB fc6 weight read, and it leaves L2 / MALL to the activations.
#ifdef __HIPCC__
typedef float nt_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float* p) {
  const nt_v4f v = __builtin_nontemporal_load(reinterpret_cast<const nt_v4f*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void nt_store4(float* p, const float4& v) {
  nt_v4f t;
  t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
  __builtin_nontemporal_store(t, reinterpret_cast<nt_v4f*>(p));
}
#endif

// ---- gemm_mfma.hip -------------------------------------------------------
enum GemmMode { GEMM_DENSE = 0, GEMM_CONV3 = 1, GEMM_CONV3_C3 = 2 };

struct GemmParams {
  // A operand.  DENSE: rows of [a1 (k1 cols) | a2 (K-k1 cols)].  CONV3*: NHWC image(s).
  const float* a1;
  int lda1;
  int k1;
  const float* a2;
  int lda2;
  int H, W, Cin;
  int M, N, K;      // K already padded to a multiple of 32
  const float* bp;  // weights in disn_pack_kn order
  const float* bias;
  int rows_per_bias;  // 0: one bias row; else bias row = m / rows_per_bias
  float* out;
  int ldc;
  int relu;
};

struct GemmPlan {
  int bm, bn, wgs;  // tile shape and number of (stream-K) workgroups
  size_t ws_bytes;  // stream-K partial slabs (0 when every workgroup owns whole tiles)
};

// max_ws bounds the stream-K partial slabs the plan may use (the plan is a pure speed choice);
// force = {BM, BN, workgroups} replaces the cost model's choice when it is a valid plan
GemmPlan gemm_plan(int M, int N, int K, size_t max_ws = ~size_t(0), const int* force = nullptr);
hipError_t gemm_launch(const GemmParams& p, GemmMode mode, const GemmPlan& plan, float* ws,
                       hipStream_t st);
hipError_t pack_kn_launch(const float* w, int K, int N, int Kpad, float* packed, hipStream_t st);
// out[m][n] = act(sum_s ws[s][m][n] + bias[row(m)][n])
hipError_t splitk_reduce_launch(const float* ws, int S, int M, int N, const float* bias,
                                int rows_per_bias, int relu, float* out, int ldc, hipStream_t st, int force_sl = 0);
// the same for an NHWC conv output [B,H,W,N] (H, W even, ldc = N) + its 2x2 max pool -> pool_out
hipError_t splitk_reduce_pool_launch(const float* ws, int S, int B, int H, int W, int N,
                                     const float* bias, int relu, float* out, float* pool_out,
                                     hipStream_t st);

// ---- gemm_bf16_mfma.hip: bf16-compute variant for the mixed-precision training step ---------
// view 0: W [K][N]; 1: W^T (reduction N, columns K); 2: flipped 3x3 kernel for conv backward-data
// (K = Cin, N = Cout: reduction 9*Cout, columns Cin).  packed: reduction padded to 32, 2 bytes each
// nsplit = 3: three planes (h, m, l: w == h + m + l) for the fp32-accurate 3xBF16 product
hipError_t pack_bf16_launch(const float* w, int view, int K, int N, void* packed, hipStream_t st,
                            int nsplit = 1);
// many re-packs in one launch: views as pack_bf16_launch; bf16 = false gives the fp32 disn_pack_kn
// order (view 1 == pack_kn_T_launch, view 2 == pack_conv_bwd_launch)
struct PackJob {
  const float* src;
  void* dst;
  int view, K, N, R, C, T, ns;
  long begin, plane;
};
struct PackJobs {
  int n;
  long total;
  PackJob j[48];
};
// ns: 0 fp32 order (disn_pack_kn), 1 bf16, 3 three bf16 planes
void pack_job_add(PackJobs& jobs, const float* src, void* dst, int view, int K, int N, int ns);
hipError_t pack_multi_launch(const PackJobs& jobs, hipStream_t st);
// as gemm_launch (DENSE or CONV3, fp32 in / fp32 out, bias + optional ReLU), multiply in bf16
size_t gemm_bf16_ws_bytes(int M, int N, int K);  // split-K partials for layers with few tiles
hipError_t gemm_bf16_launch(const GemmParams& p, GemmMode mode, const void* bpk, float* ws,
                            size_t ws_bytes, hipStream_t st, int nsplit = 1, float* pool_out = nullptr,
                            bool* pooled = nullptr);

// ---- gemv.hip ------------------------------------------------------------
int gemv_splits(int K, int N, int B = 1);
size_t gemv_ws_bytes(int B, int K, int N);
// out[b][n] = act(sum_k x[b][k] W[k][n] + bias[n]); N % 256 == 0
hipError_t gemv_launch(const float* x, int B, int K, const float* w_kn, const float* bias, int N,
                       int relu, float* out, float* ws, hipStream_t st, bool single_form = false);
// the same from the transposed matrix wt_nk [N][K]: one launch, no split-K partials (K % 4 == 0)
hipError_t gemv_rows_launch(const float* x, int B, int K, const float* wt_nk, const float* bias, int N, int relu,
                            float* out, hipStream_t st);


// ---- elementwise.hip (compiled with -ffp-contract=off) --------------------
// max_blocks > 0 caps the grid (grid-stride): a background launch that should trickle under
// MFMA-bound work instead of flooding the CUs; zero / nzero: floats the launch also clears (side job)
hipError_t resize_bilinear_launch(const float* in, int B, int Hin, int Win, int C, float* out,
                                  int Hout, int Wout, int out_cstride, int out_coff,
                                  hipStream_t st, int max_blocks = 0, float* zero = nullptr, int nzero = 0);
hipError_t restride_rows_launch(const float* src, int B, int rs, float* dst, int rd, int w, hipStream_t st);
hipError_t scale_channels_launch(const float* in, int64_t rows, int C, const float* scale, int invert, float* out,
                                 hipStream_t st);
hipError_t maxpool2x2_launch(const float* in, int B, int H, int W, int C, float* out,
                             hipStream_t st);
hipError_t project_launch(const float* pts, const float* trans_mat, int B, int N, float* xy,
                          hipStream_t st);
hipError_t gather_launch(const float* featmap, const float* xy, int B, int N, float* feat,
                         hipStream_t st);
// project + gather for a chunk of ONE image (points are a slice of image b's points)
hipError_t project_gather_launch(const float* featmap_b, const float* trans_mat_b, const float* pts,
                                 int n, float* feat, hipStream_t st);
// same result without a feature map, for B images x n points each: up-samples taps [tap_begin,
// tap_end) (taps[k] = [B,hw,hw,ch] NHWC) at the touched pixels, writes their channels of [B*n,1472]
hipError_t project_gather_taps_launch(const float* const taps[5], const float* trans_mat,
                                      const float* pts, int B, int n, int tap_begin, int tap_end,
                                      float* feat, hipStream_t st, int feat_ld = 0, float* amax = nullptr,
                                      size_t amax_stride = 0, int amax_cap = 0, const float* split_amax = nullptr,
                                      const float* const* tap_slots = nullptr, size_t slot_stride = 0);
// tap_slots != nullptr (with split_amax, from 10 240 points on -- the one-wave-per-point kernel): image b's 64 activation-
// maximum slots of tap k at tap_slots[k] + b * slot_stride.  The kernel then takes the image's maximum over the 5 x 64
// slots ITSELF and writes it to split_amax[b] (an OUTPUT then: what tap_amax_launch would have produced, for the fused
// kernel behind) -- no tap_amax launch in front of the gather.
bool project_gather_taps_takes_slots(int B, int n, int feat_ld);   // whether that form runs for this shape (else: tap_amax_launch first)
// split_amax != nullptr ([B] floats, >= max |tap| of each image): rows in SPLIT form (two f16 planes of feature * the
// image's power-of-two scale, [h8 | l8] per 8 channels) -- the operand of mlp_fused_small_launch(local)
// amax != nullptr (all five taps): max |feat| per workgroup at amax[b * amax_stride + (0 .. blocks - 1)]
int project_gather_taps_amax_blocks(int n, int feat_ld, int tap_begin = 0, int tap_end = 5);  // feat_ld > 1472: zero-padded rows
// folded local fold2/conv1 (disn_fold_local): h = relu(pre + resample(pmap_b)(pts) + bias), [n,512]
hipError_t gather_fold_launch(const float* pmap_b, const float* trans_mat_b, const float* pts, int n,
                              const float* pre, const float* bias, float* h, hipStream_t st);
struct GridSpec {
  double start[3], step[3], stop[3];
  int res;  // R+1
};
hipError_t grid_points_launch(const GridSpec& g, int64_t k0, int64_t k1, float* pts,
                              hipStream_t st);
hipError_t scale_div_launch(const float* in, float divisor, int64_t n, float* out, hipStream_t st);

// ---- gemm_tn_mfma.hip: weight gradients C[P][Q] = sum_m A[m][p] B[m][q] ----------------
struct TnParams {
  const float* a;  // dense: [M][lda]; conv (Cin > 0): NHWC activations, P = 9*Cin (implicit im2col)
  int lda;
  const float* b;  // [M][ldb]
  int ldb;
  long M;
  int P, Q;  // multiples of 64
  float* c;  // [P][ldc]
  int ldc;
  int H, W, Cin;
  float l2;           // the fix-up adds l2 * wcur (weight-decay gradient); 0: nothing
  const float* wcur;  // same layout as c
  int bf16;           // 1: multiply on the bf16 MFMA (operands rounded when staged), fp32 accumulate;
                      // 2: fp32-accurate two-term f16 split on the f16 MFMA -- needs the operands' maxima:
  const float* amax_a = nullptr;  // amax_a_n device floats whose maximum is >= max |a| (e.g. per-image slots)
  const float* amax_b = nullptr;
  int amax_a_n = 0, amax_b_n = 0;
};
size_t gemm_tn_ws_bytes(long M, int P, int Q);
hipError_t gemm_tn_launch(const TnParams& p, float* ws, hipStream_t st);

// ---- backward.hip: the small kernels of the training step --------------------------
// packed (disn_pack_kn order) form of W^T for dX = dZ W^T; W is [K][N], K % 32 == 0, N % 8 == 0
hipError_t pack_kn_T_launch(const float* w, int K, int N, float* packed, hipStream_t st);
// packed form of the flipped / transposed 3x3 kernel for conv backward-data:
// rows (8-t)*Cout + co, cols ci  <-  W[t][ci][co]
hipError_t pack_conv_bwd_launch(const float* w, int Cin, int Cout, float* packed, hipStream_t st);
// dZ = dY * (Y > 0) in place (relu != 0) and column sums of dZ -> db[N] (+= when accumulate)
size_t colsum_ws_bytes(long M, int N);
// amax / rows_per_image / amax_done: see backward.hip (per-image maxima of the masked gradient in the same pass)
hipError_t relu_bwd_colsum_launch(float* dy, const float* y, long M, int N, int relu, float* db,
                                  float* ws, hipStream_t st, float* amax = nullptr, long rows_per_image = 0,
                                  bool* amax_done = nullptr);
// d(sdf_loss)/d(pred): -sign(10*gt - pred) * w * 1000/M, w = 4 if gt <= 0.01 else 1
hipError_t loss_grad_launch(const float* pred, const float* gt, long M, float sdf_weight,
                            float mask_weight, float* dpred, hipStream_t st);
// last layer of both streams: dZ5 = (dpred w6) * (h5 > 0) -> dz5 [M][256]; dw6[k] = sum_m h5 dpred;
// db6 = sum dpred; db5 = colsum(dz5).  ws >= final_bwd_ws_bytes
size_t final_bwd_ws_bytes(long M);
hipError_t final_bwd_launch(const float* h5, const float* dpred, long M, const float* w6, float* dz5,
                            float* dw6, float* db6, float* db5, float l2, float* ws, hipStream_t st);
// first layer (K = 3): dw1[3][64] = sum_m p[m][c] dz1[m][n] (+ l2 w1)
hipError_t embed_bwd_launch(const float* pts, const float* dz1, long M, float* dw1, const float* w1,
                            float l2, float* ws, hipStream_t st);
// per-image sums: out[b][n] = sum over the N rows of image b of x[b*N + i][n]
hipError_t image_colsum_launch(const float* x, int B, long N, int C, float* out, float* ws,
                               hipStream_t st);
// rank-B outer product: c[k][n] (+ l2 w) = sum_b x[b][k] dy[b][n]
hipError_t outer_launch(const float* x, const float* dy, int B, int K, int N, float* c, const float* wcur,
                        float l2, hipStream_t st);
// dx[b][k] = sum_n W[k][n] dy[b][n] (* (xact[b][k] > 0) when xact != nullptr)
// both of the above in one pass over W (B <= 8: dy in LDS); same expressions and summation orders
hipError_t fc_bwd_launch(const float* x, const float* dy, int B, int K, int N, const float* w, float l2, float* dw,
                         const float* xact, float* dx, hipStream_t st);
hipError_t gemv_t_launch(const float* w_kn, const float* dy, int B, int K, int N, const float* xact,
                         float* dx, hipStream_t st);

// out[i] = src[i] + l2 * w[i]
hipError_t axpby_launch(const float* src, const float* w, float l2, size_t n, float* out,
                        hipStream_t st);
// out5 = {accuracy, sdf_loss_realvalue, sdf_loss, regularization (read), overall_loss}
hipError_t loss_reduce_launch(const float* pred, const float* gt, long M, float sdf_weight,
                              float mask_weight, float* out5, hipStream_t st);
struct SumsqSegs {
  int n;
  long off[32], cnt[32];
};
// *out = half_wd * sum over the segments of sum(params[off..off+cnt)^2); ws: 32*256 floats
hipError_t sumsq_launch(const float* params, const SumsqSegs& segs, float half_wd, float* out, float* ws,
                        hipStream_t st);
// TF Adam on n floats (n % 4 == 0); the gradient is scaled by gscale first (1/world for DDP)
hipError_t adam_launch(float* w, const float* g, float* m, float* v, size_t n, float lr_t, float b1,
                       float b2, float eps, float gscale, hipStream_t st);

// ---- backward_img.hip ------------------------------------------------------------------
// dmap [B,137,137,1472] (zeroed by the caller) += resampler-grad of dfeat [B*N][1472] at xy [B*N][2]
hipError_t gather_bwd_launch(const float* dfeat, const float* xy, int B, int N, float* dmap,
                             hipStream_t st);
// din [B,Hin,Win,C] (+)= ResizeBilinearGrad of channels [coff, coff+C) of dout [B,Hout,Wout,cstride]
// tmp: resize_bwd_ws_bytes of scratch for the separable (rows, then columns) form used for
// >= 2x up-sampling; nullptr (or 0 bytes needed): one direct pass
size_t resize_bwd_ws_bytes(int B, int Hin, int Win, int C, int Hout, int Wout);
hipError_t resize_bwd_launch(const float* dout, int B, int Hin, int Win, int C, int Hout, int Wout,
                             int out_cstride, int out_coff, float* din, int accumulate, float* tmp,
                             hipStream_t st);
// x [B,H,W,C] pre-pool activations, dy [B,H/2,W/2,C] -> dx [B,H,W,C] (every element written)
hipError_t maxpool_bwd_launch(const float* x, const float* dy, int B, int H, int W, int C, float* dx,
                              hipStream_t st);
// 3x3 SAME patches of a 3-channel image as [B*H*W][64] rows (27 used, rest zero)
hipError_t im2col_c3_launch(const float* img, int B, int H, int W, float* col, hipStream_t st);

// ---- marching_cubes.hip (compiled with -ffp-contract=off) ------------------------
size_t mc_ws_bytes(int R);
// counts[0] = vertices, counts[1] = triangles (device memory); fills ws for mc_emit_launch
hipError_t mc_count_launch(const float* vol, int R, float iso, unsigned long long* counts, void* ws,
                           hipStream_t st);
hipError_t mc_emit_launch(const float* vol, const GridSpec& g, float iso, float* verts, int* faces,
                          void* ws, hipStream_t st);

// ---- mlp_fused.hip: both point MLPs as one persistent kernel per stream, activations in registers ----
size_t mlp_fused_image_bytes();
size_t mlp_fused_feat_image_bytes();
// the floor under an image's feature maximum before it becomes the split scale (shared by the gather that writes the
// split form and the fused kernel that consumes it: the same bits); 2^-20 loses nothing (the scaled values stay < 2^15)
__host__ __device__ inline float feat_split_amax(float a) { return a > 9.5367431640625e-07f ? a : 9.5367431640625e-07f; }
// w4: the whole local fold2/conv1 matrix [512 + 1472][512]
hipError_t mlp_fused_feat_pack_launch(const float* w2, const float* w3, const float* w4, const float* w5, void* image,
                                      hipStream_t st);
hipError_t mlp_fused_small_launch(bool local, const void* image, const float* w1, const float* b1, const float* b2,
                                  const float* b3, const float* b4, const float* b5, const float* w6, const float* b6,
                                  const float* pts_rot, long long rows_per_image, int images, const void* feat_split,
                                  int feat_ld, const float* feat_amax, const float* add_in, float* out, float out_div,
                                  hipStream_t st);
// w2 [64][256], w3 [256][512], w4_point [512][512] (the point rows of fold2/conv1), w5 [512][256]: TF [K][N]
hipError_t mlp_fused_pack_launch(const float* w2, const float* w3, const float* w4_point, const float* w5,
                                 void* image, hipStream_t st);
hipError_t amax_launch(const float* x, size_t n, float* out, hipStream_t st);  // n % 4 == 0
// the same into 64 slots (their maximum is max |x|) / the maximum of 64 slots -> out[0]
hipError_t amax64_launch(const float* x, size_t n, float* out64, hipStream_t st);
// the same without clearing the slots first (the caller's chain cleared them): one launch
// images > 1: `images` arrays of n floats back to back, the slots of image i at out64 + i * out_stride
hipError_t amax64_accumulate_launch(const float* x, size_t n, float* out64, hipStream_t st, int images = 1,
                                    size_t out_stride = 0);
// maximum of n non-negative slots; groups > 1: `groups` runs of n slots, `stride` floats apart
hipError_t amax_fold_launch(const float* slots, float* out, hipStream_t st, int n = 64, int groups = 1, int stride = 0);
// out[b] = the maximum over five 64-slot groups (the five taps' activation maxima of image b: slots[k] + b * image_stride)
hipError_t tap_amax_launch(const float* const slots[5], size_t image_stride, int B, float* out, hipStream_t st);
// one stream for n points of one image; pts_rot == nullptr: points k0.. of `grid`.  local: gather from
// pmap + 'sdfprediction_imgfeat', out = (add_in + sum) / out_div; global: 'sdfprediction' with b4 = the
// folded per-image bias row, out = sum
hipError_t mlp_fused_launch(bool local, const void* image, const float* w1, const float* b1, const float* b2,
                            const float* b3, const float* b4, const float* b5, const float* w6, const float* b6,
                            const float* pts, const float* pts_rot, const GridSpec* grid, long long k0,
                            long long n, const float* trans_mat_b, const float* pmap, const float* pmap_amax,
                            const float* add_in, float* out, float out_div, hipStream_t st);

// ---- conv_h2.hip: the single-image 3x3 convolution (two-term f16 split, halo in LDS, K parallel inside the workgroup) ----
size_t conv_h2_image_bytes(int Cin, int Cout);
size_t h2_image_bytes(int K, int N, int taps);   // payload + inv_sw[N] + column-maximum scratch[N] + pad
// w: TF HWIO [3][3][Cin][Cout] (taps = 9) or a [K = Cin][N = Cout] matrix (taps = 1, dense_h2.hip); scratch: one
// device float; the image holds taps * Cin * Cout * 4 + 256 bytes.  flip_t: w is the forward tensor [taps][Cout][Cin]
// of a layer and the image the one of its data gradient, w'[t][ci][co] = w[taps - 1 - t][co][ci] (train.hip)
hipError_t conv_h2_pack_launch(const float* w, int Cin, int Cout, void* image, float* scratch, hipStream_t st,
                               int taps = 9, int flip_t = 0);
// Many 3x3 weight images in TWO launches (train.hip re-packs 12 forward + 12 data-gradient images per step; as 72
// separate memset / max / pack launches they were 0.46 ms of serial 5-us kernels): one max |w| pass over all tensors
// (job k's scale from wmax[slot]; images of one tensor share a slot), one pack pass over all fragments.
struct ConvH2PackJob {
  const float* w;          // forward tensor [9][Cin_fwd][Cout_fwd]
  unsigned char* image;
  int Cin, Cout;           // of the IMAGE (swapped against the tensor's when flip_t)
  int flip_t, slot;
  long frag_begin;         // first fragment (of 64 lanes) of this job in the concatenated space
};
struct ConvH2PackJobs {
  int n, nslots;
  long total_frags;
  float* wmax;             // unused since round 4 (column maxima live in the images' tails)
  const float* seg[16];    // tensor of slot s
  int seg_mid[16], seg_inner[16];          // its [9][mid][inner] shape (Cin_fwd, Cout_fwd)
  float *cmax_fwd[16], *cmax_flip[16];     // column-maximum scratch in the tail of the slot's forward / flipped image
  long seg_begin[17];      // in floats, multiples of 4096
  ConvH2PackJob j[32];
};
void conv_h2_pack_job_add(ConvH2PackJobs& jobs, const float* w_fwd, int Cin_fwd, int Cout_fwd, void* image, int flip_t,
                          int slot);
hipError_t conv_h2_pack_multi_launch(const ConvH2PackJobs& jobs, hipStream_t st);
bool conv_h2_supported(int H, int W, int Cin, int Cout);  // Cin, Cout multiples of 64
// in_amax: 64 floats whose maximum is max |in|; out_amax (optional, 64 floats zeroed by the caller): atomic
// max |out| spread over the slots; pool_out
// (optional, H and W even): the 2x2 max pool of out; tiling 0: by shape
hipError_t conv_h2_launch(const float* in, int B, int H, int W, int Cin, const void* wimg, const float* bias,
                          int Cout, int relu, const float* in_amax, float* out, float* pool_out, float* out_amax,
                          hipStream_t st, int tiling = 0, int amax_stride = 0);  // amax_stride > 0: slot groups PER IMAGE

// ---- conv_h2w.hip: the same convolution for a batch of images (waves own n-blocks, K sequential; see the file) ----
struct ConvH2Dev;
bool conv_h2w_supported(int H, int W, int Cin, int Cout);
int conv_h2w_kwaves(int H, int W, int Cin, int Cout);   // by layer shape only: fixes the summation order
hipError_t conv_h2w_launch(ConvH2Dev d, hipStream_t st, int variant);  // variant 0: by shape and batch; 1..5 forced
constexpr int kConvWideMinImages = 4;   // conv_h2_launch(tiling = 0) takes the batched form from this many images on

// conv1_1 (Cin = 3, Cout = 64) as a direct fp32 FMA convolution; w_hwio: the TF tensor [3][3][3][64] as is
hipError_t conv1_1_direct_launch(const float* in, int B, int H, int W, const float* w_hwio, const float* bias, int relu,
                                 float* out, float* out_amax, hipStream_t st, int amax_stride = 0);

// ---- dense_h2.hip: the point-MLP layers at a few thousand rows (two-term f16 split, 1x1 sibling of conv_h2) ----
struct DenseH2Prob {     // out[M][N] = act(f(A) . W + bias), A = [a (k1 columns) | a2 (K - k1)], f = relu(. + in_bias) or identity
  const float* a;
  int lda;
  const float* a2;       // nullptr: one source (k1 == K)
  int lda2, k1;
  const float* in_bias;  // [K] (or [images][K] with in_bias_rows) or nullptr
  int in_bias_rows;      // > 0: row m uses in_bias row m / in_bias_rows (a per-image bias, rows image-major)
  const unsigned char* wimg;  // conv_h2_pack_launch(w [Kimg][N], Kimg, N, ..., taps = 1)
  int Kimg;              // rows of the packed matrix (0: K); > K: an image zero-padded beyond the K used here
  const float* bias;     // [N]
  const float* in_amax;  // 64 slots: max |a|
  const float* in_amax2; // 64 slots: max |a2| (nullptr with one source)
  int in_amax2_n;        // > 0: in_amax2 has this many entries instead of 64 (per-workgroup maxima of a producer)
  float* out;
  int ldc;
  float* out_amax;       // 64 slots (zeroed by the caller) or nullptr
  int M, N, K, relu;
  int amax_rows;         // > 0: rows per image (a multiple of 64): the maxima of the tile's image are read / written at
  int amax_stride;       //      in_amax / in_amax2 / out_amax + image * amax_stride  // dense_h2w.hip only (dense_h2_launch rejects them for the four-k-wave tiles):
  int k_begin;           // first row of the packed matrix this product uses (a multiple of 16): W[k_begin .. k_begin + K)
  const float* add_in;   // [M][ldc] addend (a partial product formed earlier) or nullptr: out = act(A . W + bias + add_in); may alias out
  int in_amax_n;         // > 0: in_amax has this many entries instead of 64
  int no_wide;           // 1: the four-k-wave tiles of dense_h2.hip whatever the number of images ("strict": the bits of a call of one image)
};
struct DenseH2Dev {
  DenseH2Prob p[2];
  int nprob, mtiles;
};
bool dense_h2_supported(int M, int K, int N, int k1);  // K, N multiples of 64; k1 a multiple of the chunk (256 when K % 256 == 0, else 64)
// one or two problems of ONE shape in one launch (the same layer of the global and the local stream)
hipError_t dense_h2_launch(const DenseH2Prob* probs, int nprob, hipStream_t st);
// dense_h2w.hip: the batched form (128 x 256 tiles, n-waves, K sequential).  dense_h2_launch takes it when the rows
// belong to >= kConvWideMinImages images (amax_rows > 0, M / amax_rows images) of a multiple of 128 rows each and the
// shape allows (N % 256 == 0, K and k1 multiples of 128): by the call's image count, never by the other images
bool dense_h2w_supported(const DenseH2Prob& p);
hipError_t dense_h2w_go(DenseH2Dev d, hipStream_t st);

// ---- mlp_small.hip ---------------------------------------------------------
// relu(p . W1 + b1) for both streams: pts [M][3] -> out_g [M][64], out_l [M][64]
// amax_gl (optional): 128 floats, [0..63] slots of max |out_g|, [64..127] of max |out_l| (dense_h2.hip reads the
// maximum over the slots); zero / nzero (optional): floats this launch also clears
// images > 1 (with amax_gl): M = images * rows_per_image rows image-major; the slot set of image i (1024 floats:
// see api.hip MlpWs) at amax_gl + i * 1024: [0..63] / [64..127] its two maxima, [128..1023] cleared; zero / nzero
// additionally cleared once
hipError_t pt_embed_launch(const float* pts, int64_t M, const float* g_w1, const float* g_b1,
                           const float* l_w1, const float* l_b1, float* out_g, float* out_l,
                           hipStream_t st, float* amax_gl = nullptr, float* zero = nullptr, int nzero = 0,
                           int images = 1);
// sdf[m] = (g5[m].g_w6 + g_b6) + (l5[m].l_w6 + l_b6); optional separate outputs
hipError_t final_dot_launch(const float* g5, const float* l5, int64_t M, const float* g_w6,
                            const float* g_b6, const float* l_w6, const float* l_b6, float* sdf,
                            float* sdf_g, float* sdf_l, float out_div, hipStream_t st);

}  // namespace disn
