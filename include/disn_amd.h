/*
 * disn_amd.h -- C ABI of the MI355X (gfx950) DISN SDF-query hot path.
 *
 * The reference (laughtervv/DISN) has NO C/FFI boundary on this path: the
 * boundary is the Python module API of models/model_normalization.py /
 * models/sdfnet.py executed by TensorFlow.  This header is therefore the
 * boundary a binding for that module API would use; each entry point cites the
 * reference graph fragment (file:line under /root/reference) it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - all tensors are float32, row-major, NHWC; sizes are plain ints;
 *   - the caller owns every buffer (the library never allocates); scratch is
 *     sized by the *_workspace_bytes() queries and passed in;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - return value: 0 ok; <0 invalid argument (DISN_E_*); >0 a hipError_t.
 *     Nothing throws or aborts.  Entry points are re-entrant per stream and keep
 *     no global mutable state.
 */
#ifndef DISN_AMD_H
#define DISN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DISN_IMG_H 137
#define DISN_IMG_W 137
#define DISN_VGG_SIZE 224
#define DISN_FEAT_DIM 1472 /* 64+128+256+512+512, models/model_normalization.py:41 */
#define DISN_EMBED_DIM 1024

#define DISN_E_ARG (-1)   /* null pointer / non-positive size */
#define DISN_E_SHAPE (-2) /* unsupported shape (channel multiple, image size ...) */
#define DISN_E_WS (-3)    /* workspace too small */

/* ABI version of this header; disn_abi_version() returns the library's. */
#define DISN_ABI_VERSION 10
int disn_abi_version(void);

/* ---------------------------------------------------------------------- *
 * Weight packing.  GEMM-shaped layers read their [K][N] weight matrix      *
 * (TF HWIO flattened: K = kh*kw*Cin, N = Cout) in MFMA B-fragment order:   *
 * packed[((k/8)*(N/32) + n/32)*256 + lane*4 + t] =                         *
 *     W[8*(k/8) + 4*(lane>>5) + t][32*(n/32) + (lane&31)].                 *
 * K is zero-padded up to Kpad (multiple of 32); N must be a multiple of 32.*
 * `packed` holds Kpad*N floats.                                            *
 * ---------------------------------------------------------------------- */
int disn_pack_kn(const float* w_kn, int K, int N, int Kpad, float* packed, void* stream);
/* Three-term bf16 image of the same [K][N] matrix for the fp32-accurate path on the bf16 MFMA pipes:
 * planes h, m, l (w == h + m + l exactly), each in bf16 B-fragment order
 * plane[((k/16)*(N/32) + n/32)*512 + lane*8 + t] = W[16*(k/16) + 8*(lane>>5) + t][32*(n/32) + (lane&31)],
 * K zero-padded to a multiple of 32.  `packed` holds disn_pack_kn_x3_bytes(K, N) bytes. */
size_t disn_pack_kn_x3_bytes(int K, int N);
int disn_pack_kn_x3(const float* w_kn, int K, int N, void* packed, void* stream);
/* disn_conv3x3 on that image (Cin a multiple of 32): what disn_vgg16_forward / disn_encode* run for a
 * layer whose conv_w_x3 entry is set.  Same result as disn_conv3x3 to fp32 rounding. */
size_t disn_conv3x3_x3_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int disn_conv3x3_x3(const float* in, int B, int H, int W, int Cin, const void* w_x3, const float* bias,
                    int Cout, int relu, float* out, void* ws, size_t ws_bytes, void* stream);
/* The single-image convolution (conv_h2.hip; models/CNN/vgg.py:187-196): fp32-accurate products from a two-term
 * f16 split of both operands (x = h + l after a power-of-two scale; l_a h_b + h_a l_b + h_a h_b accumulated in
 * fp32), the 3x3 halo of a 2-D pixel patch staged once in LDS, all K parallelism inside the workgroup -- no
 * split-K pass.  disn_pack_conv_h2: TF HWIO [3][3][Cin][Cout] -> the weight image (Cin, Cout multiples of 64;
 * `image` holds disn_pack_conv_h2_bytes).  OPERAND SCALES (round 4): activations one power of two per IMAGE (largest
 * magnitude -> [2^14, 2^15)); weights one power of two per OUTPUT CHANNEL (largest |entry| of the column -> [2^13,
 * 2^14); the inverse scales are the image's tail, a lane multiplies its column's accumulators with its own next to
 * the bias).  The two-term split keeps ~22 bits of an operand down to 2^-17 of its scale's maximum (below, f16's
 * 2^-24 subnormal floor takes over): with per-column weight scales, trained-like weights with log-normal channel
 * gains and outlier channels x 1000 stay inside the 1e-5 bar (tests/test_gpu_stress.py; a per-TENSOR weight scale, rounds
 * 2-3, left the ordinary columns of such a tensor with a handful of bits: tools/split_model.py).  disn_conv3x3_h2: out = act(conv(in) + bias) [B,H,W,Cout]; optional
 * pool_out = its 2x2 max pool [B,H/2,W/2,Cout] (H, W even), optional out_amax = max |out| (the activation scale
 * of a following disn_conv3x3_h2 inside disn_encode*; here every image's own maximum is measured first: as inside
 * disn_encode* the scale of an image, hence its bits, never depend on the other images of the call).
 * tiling: 0 = by shape and batch, 1..4 = force one of the four single-image workgroup tilings (the K order of an
 * output element depends on the number of k-waves only: tilings with the same number give the same bits), 5..9 =
 * force variant 1..5 of the BATCHED form (conv_h2w.hip: waves own 32-channel n-blocks of 128..224-pixel patches and
 * walk K sequentially; variants 1..3 one k-wave, 4..5 two; H*W >= 784), 10 = the whole-image tiling of layers of at
 * most 14 x 14 pixels (one workgroup per image and 32-channel block, four k-waves), 11 = by shape, single-image form
 * whatever B (what disn_vgg_weights_t.strict_forms = 1 runs), 12 / 13 = the SEGMENTED batched variants (round 6: the k16
 * blocks in two halves -- lower, upper --, each half in segments of two 16-channel chunks -- chains of 54 MFMAs -- whose
 * finished sums are added in fp32 VALU adds, p0 + p1 at the end; 13: two k-waves over 64-channel workgroups, 12: ONE k-wave
 * that parks p0 in LDS, 128-channel workgroups, Cout % 128 == 0; the SAME bits; Cin % 64 == 0, Cin >= 128), 18 = round 3's
 * batched selection by shape and batch (one or two k-waves, chains of up to 432: what the training step runs), 19 = tiling
 * 10 in segments of two 64-channel chunks (Cin % 128 == 0).
 * SELECTION RULE of tiling 0 (also inside disn_vgg16_* / disn_encode*): calls of B >= 4 images take, by LAYER SHAPE only,
 *   - layers of 28 x 28 pixels and more with Cin >= 128 (conv2_2 .. conv4_3): the segmented batched variants (13 while
 *     that is what fills the chip, 12 from ~200 128-channel workgroups on: the same bits);
 *   - layers of 28 x 28 pixels and more with Cin = 64 (conv1_2, conv2_1): the one-k-wave batched form (chains of 108);
 *   - the 14 x 14 layers: four k-waves in segments of two chunks (19 where B * Cout / 32 >= 200, two-row patches below:
 *     same bits);
 * everything else the single-image form.  The forms sum K in different orders: results agree to fp32 rounding (all
 * within 1e-6 of the layer's scale of the float64 convolution; rms 3-4e-8), NOT bit for bit -- an image's bits depend on
 * which form ran it (B >= 4 or not), never on its companions, its position or the exact B.  Why segments: a chain of
 * L MFMAs on one fp32 accumulator carries ~0.3 sqrt(L) ulp of rounding noise (tools/ubench/mfma_round.hip: 6.2 ulp rms
 * at 432, 0.9-1.0 with a restart every 27-54); round 3's chains of 216-432 put 2.7 % of trained-like batched requests at
 * 1.0-1.46e-5 of the float64 oracle (profiles/r05k_sweep_full.json), the segmented forms none (profiles/r06w_sweep_full.json: worst 9.5e-6).
 * ws: disn_conv3x3_h2_workspace_bytes(B). */
size_t disn_pack_conv_h2_bytes(int Cin, int Cout);
int disn_pack_conv_h2(const float* w_hwio, int Cin, int Cout, void* image, void* stream);
/* Round 6 (ABI 10) -- the accuracy contract's guard: log2 of (largest / smallest non-zero) per-output-channel weight
 * scale of a packed image, i.e. the channel-gain span of the variable it was packed from, into *span_log2 (HOST float;
 * the call synchronises `stream`).  Returns 0, or DISN_W_GAIN_SPAN (= 1, a WARNING: the image is usable) when the span
 * exceeds 12 binades -- the activations the layer produces then span as much, and the NEXT layer's two-term split (one
 * power-of-two scale per image, ~22 bits) starves its small channels: pack the variables through
 * disn_equalise_weights first (an equalised checkpoint reports <= 2). */
#define DISN_W_GAIN_SPAN 1
int disn_conv_h2_gain_span(const void* image, int Cin, int Cout, float* span_log2, void* stream);
/* conv1_1 (3 -> 64 channels; models/CNN/vgg.py:187) as a direct fp32 FMA convolution: w_hwio is the TF tensor
 * [3][3][3][64] as is; optional out_amax = max |out|.  ws: disn_conv1_1_workspace_bytes(). */
size_t disn_conv1_1_workspace_bytes(void);
int disn_conv1_1(const float* in, int B, int H, int W, const float* w_hwio, const float* bias, int relu, float* out,
                 float* out_amax, void* ws, size_t ws_bytes, void* stream);
size_t disn_conv3x3_h2_workspace_bytes(int B);
int disn_conv3x3_h2(const float* in, int B, int H, int W, int Cin, const void* image, const float* bias, int Cout,
                    int relu, float* out, float* pool_out, float* out_amax, int tiling, void* ws, size_t ws_bytes,
                    void* stream);

/* ---------------------------------------------------------------------- *
 * Row A / E: tf.image.resize_bilinear, TF1 legacy (align_corners=False,    *
 * no half-pixel) -- models/model_normalization.py:72 and :171-183.         *
 * Writes channels [out_coff, out_coff+C) of an output whose pixel stride is*
 * out_cstride floats (out_cstride==C, out_coff==0 for a plain resize).     *
 * Bit-exact with the oracle (FMA contraction off).                         *
 * ---------------------------------------------------------------------- */
int disn_resize_bilinear(const float* in, int B, int Hin, int Win, int C, float* out, int Hout,
                         int Wout, int out_cstride, int out_coff, void* stream);

/* ---------------------------------------------------------------------- *
 * Rows B + C: slim vgg_16(num_classes, is_training=False,                  *
 * spatial_squeeze=False) -- call models/model_normalization.py:74-78,      *
 * architecture models/CNN/vgg.py:187-214.                                  *
 * conv_w[i]: disn_pack_kn of 'vgg_16/convX/convX_Y/weights' ([3,3,Cin,Cout]*
 *   -> K=9*Cin; conv1_1 K=27 padded to 32).  fc_w[i]: the TF tensor as is  *
 *   ([7,7,512,4096], [1,1,4096,4096], [1,1,4096,num_classes]) = [K][N].    *
 * ---------------------------------------------------------------------- */
typedef struct disn_vgg_weights {
  const float* conv_w[13]; /* packed, order conv1_1 .. conv5_3 */
  const float* conv_b[13];
  const float* fc_w[3]; /* fc6, fc7, fc8 : [K][N] row-major */
  const float* fc_b[3];
  int num_classes; /* 1024 on this path */
  /* optional (NULL = not used): disn_pack_kn_x3 of the same conv weights (index 0 is ignored).  With
   * it a layer runs as a three-term bf16 split on the bf16 MFMA pipes -- the same fp32 accuracy
   * (every fp32 operand is the exact sum of three bf16 terms; six cross products accumulated in
   * fp32), 1.1-1.5x the speed of the f32-input MFMA. */
  const void* conv_w_x3[13];
  /* optional (NULL = not used): disn_pack_conv_h2 of the HWIO tensor; index 0 (Cin = 3): the TF tensor [3][3][3][64]
   * as is.  With it a layer runs the single-image kernels of conv_h2.hip (two-term f16 split; conv1_1: direct fp32
   * FMA) -- takes precedence over conv_w_x3; all 13 entries must be set for the activation-scale chain. */
  const void* conv_w_h2[13];
  /* optional (NULL = not used): fc6, fc7, fc8 TRANSPOSED, [N][K] row-major (one contiguous K-long row per output).
   * With it a layer is one launch (a wave per output row pair, no split-K partials, no reduce pass). */
  const float* fc_w_t[3];
  /* 0 (default): the kernel forms are chosen by the call size (B < 4: conv_h2.hip / dense_h2.hip, B >= 4: conv_h2w.hip /
   * the fused small-set point MLP -- see disn_conv3x3_h2, disn_encode_query).  1 ("strict"): the single-image forms for
   * EVERY call size -- conv_h2.hip's and dense_h2.hip's four / eight k-wave trees sum K in chains of ~108 MFMAs per
   * accumulator where the batched forms have up to 432; the fc head with the split count, VALU kernels and reduce lanes
   * of a one-row call -- so the taps, the embedding AND pred_sdf of a request in a call of any size are BIT FOR BIT
   * those of the request alone (any N), i.e. it keeps the single-request form's distance from the float64
   * oracle (tests/test_gpu_sweep.py: EVERY request of the trained-like sweep <= 1e-5 -- median 2.2e-6 --
   * where the default's batched forms leave 2.7 % of the requests at 1.0-1.46e-5).  Costs ~39 % of a batched call's
   * throughput (bench.py --strict: 10.3 M against 17 M points/s).  N >= 8192 per request keeps the fused kernels (per-point scales) either
   * way -- there, too, bit for bit the request alone. */
  int strict_forms;
} disn_vgg_weights_t;

size_t disn_vgg16_workspace_bytes(int B);

/* img: [B,137,137,3] BGR in [0,1] (demo/demo.py:263-264).  Outputs:
 * resized224 [B,224,224,3] ('resized_ref_img'); taps[0..4] = conv1_2
 * [B,224,224,64], conv2_2 [B,112,112,128], conv3_3 [B,56,56,256], conv4_3
 * [B,28,28,512], conv5_3 [B,14,14,512]; embedding [B,num_classes]. */
int disn_vgg16_forward(const disn_vgg_weights_t* w, const float* img, int B, float* resized224,
                       float* const taps[5], float* embedding, void* ws, size_t ws_bytes,
                       void* stream);
/* Rows A + B only: the resize and the 13 convolutions (+ pools) of that forward, no fc head -- the launches
 * bench.py's `roofline` times.  pool5 [B,7,7,512] optional (NULL: not copied out).  Workspace as
 * disn_vgg16_forward. */
int disn_vgg16_conv_stack(const disn_vgg_weights_t* w, const float* img, int B, float* resized224,
                          float* const taps[5], float* pool5, void* ws, size_t ws_bytes, void* stream);

/* Single layers of the above (unit-test / composition surface).
 * disn_conv3x3: SAME 3x3 stride-1 conv + bias + optional ReLU, NHWC, Cin in {3} or a
 *   multiple of 32, Cout a multiple of 64; ws >= disn_conv3x3_workspace_bytes. */
size_t disn_conv3x3_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int disn_conv3x3(const float* in, int B, int H, int W, int Cin, const float* w_packed,
                 const float* bias, int Cout, int relu, float* out, void* ws, size_t ws_bytes,
                 void* stream);
/* disn_conv3x3 under an EXPLICIT GEMM plan -- tile bm x bn in {64,128}^2 and `wgs` stream-K workgroups
 * (-1: one per tile) -- instead of the cost model's choice.  The plan is a pure speed knob: every valid
 * plan gives the same layer (tests/test_gpu_kernels.py::test_conv3x3_every_tile_config_and_splitk). */
size_t disn_conv3x3_planned_workspace_bytes(int B, int H, int W, int Cin, int Cout, int bm, int bn, int wgs);
int disn_conv3x3_planned(const float* in, int B, int H, int W, int Cin, const float* w_packed,
                         const float* bias, int Cout, int relu, float* out, void* ws, size_t ws_bytes,
                         int bm, int bn, int wgs, void* stream);
int disn_maxpool2x2(const float* in, int B, int H, int W, int C, float* out, void* stream);
/* out[b][n] = act(sum_k x[b][k] W[k][n] + bias[n]); ws >= disn_fc_workspace_bytes */
size_t disn_fc_workspace_bytes(int B, int K, int N);
int disn_fc(const float* x, int B, int K, const float* w_kn, const float* bias, int N, int relu,
            float* out, void* ws, size_t ws_bytes, void* stream);
/* the same layer from the transposed matrix wt_nk [N][K] (K % 4 == 0): one launch, no workspace -- what
 * disn_vgg16_forward / disn_encode* run for a layer whose fc_w_t entry is set */
int disn_fc_t(const float* x, int B, int K, const float* wt_nk, const float* bias, int N, int relu, float* out,
              void* stream);

/* Row G3: tf_util.conv2d with a [1,1] kernel (utils/tf_util.py:119-184) == per-row
 * out[m][n] = act(sum_k A[m][k] W[k][n] + bias[n]) where A = [a1 (k1 cols) | a2 (k2 cols)]
 * (the tf.concat(axis=3) of models/sdfnet.py:82,180 is read in place, never materialised).
 * k1, k2 multiples of 32 (k2 may be 0 with a2 NULL), N a multiple of 64,
 * w_packed = disn_pack_kn of W [k1+k2][N].  fp32 MFMA. */
/* The same primitive for a few thousand rows (dense_h2.hip): fp32-accurate products from a two-term f16 split on
 * the f16 MFMA pipes, one short launch, no split-K pass.  disn_pack_dense_h2: W [K][N] row-major (K, N multiples of
 * 64) -> weight image (disn_pack_dense_h2_bytes).  disn_dense_h2: out = act(f([a1 | a2]) . W + bias) with
 * f = relu(. + in_bias[k]) when in_bias != NULL (the deferred bias + ReLU of a layer whose product was formed before
 * its bias existed), else the identity; lda1 == k1, lda2 == k2 (the rows' maxima are measured over the whole
 * buffers); k1 a multiple of 256 when K = k1 + k2 is, else of 64.  Optional out_amax = max |out|.
 * rows_per_image: 0 = the M rows are one set (one activation scale per source); > 0 (a multiple of 64 dividing M):
 * rows image-major, every image its own maxima / scales -- as inside disn_encode_query -- and in_bias is
 * [M / rows_per_image][K], one row per image.
 * SELECTION RULE (also inside disn_encode_query): rows of >= 4 images with rows_per_image % 128 == 0, N % 256 == 0,
 * K and k1 multiples of 128 take the BATCHED form (dense_h2w.hip: 128 x 256 tiles, eight n-waves, K summed in one
 * accumulator in ascending order), everything else the four-k-wave tiles of dense_h2.hip.  The two forms agree to
 * fp32 rounding (both within 2e-6 of the output scale of the float64 product), not bit for bit; which form runs
 * depends on the call's image count and rows per image only, never on the other images' data.
 * Batched form only (DISN_E_SHAPE otherwise): image_k > K = the image was packed from a matrix of image_k rows and
 * this product uses its rows k_begin .. k_begin + K (multiples of 16) -- a K range of a layer; add_in [M][N] != NULL =
 * a partial product formed earlier: out = act([a1 | a2] . W[k_begin..] + bias + add_in), may alias out.  (How
 * disn_encode_query cuts the 1984-deep local fold2/conv1 into the part whose inputs exist before conv5 and the rest.)
 * image_k = 0: the image has exactly K rows.
 * ws: disn_dense_h2_workspace_bytes(images). */
size_t disn_pack_dense_h2_bytes(int K, int N);
int disn_pack_dense_h2(const float* w_kn, int K, int N, void* image, void* stream);
size_t disn_dense_h2_workspace_bytes(int images);
int disn_dense_h2(const float* a1, int lda1, int k1, const float* a2, int lda2, int k2, const float* in_bias, int M,
                  int rows_per_image, const void* image, int image_k, int k_begin, const float* add_in,
                  const float* bias, int N, int relu, float* out, float* out_amax, void* ws, size_t ws_bytes,
                  void* stream);

/* Row K: the scalars of get_loss (models/model_normalization.py:273-299, regression branch) in one launch:
 * out5 = {accuracy, sdf_loss_realvalue, sdf_loss, regularization, overall_loss}; pred [M] = pred_sdf
 * (un-divided), gt [M] = ref_sdf; out5[3] is READ (the caller's wd/2 * sum w^2, 0 without regularization) and
 * overall_loss = sdf_loss + out5[3].  The kernel disn_train_step uses. */
int disn_get_loss(const float* pred, const float* gt, int64_t M, float sdf_weight, float mask_weight, float* out5,
                  void* stream);

size_t disn_dense_workspace_bytes(int M, int K, int N);
int disn_dense(const float* a1, int lda1, int k1, const float* a2, int lda2, int k2, int M,
               const float* w_packed, const float* bias, int N, int relu, float* out, void* ws,
               size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------- *
 * Row E: the five resize_bilinear(tap,(137,137)) of                        *
 * models/model_normalization.py:171-183, written channel-concatenated (in  *
 * the reference's concat order, :187-189) into one map                     *
 * featmap [B,137,137,1472] so that one pixel is 5888 contiguous bytes.     *
 * ---------------------------------------------------------------------- */
int disn_build_featmap(const float* const taps[5], int B, float* featmap, void* stream);

/* Row D: get_img_points -- models/model_normalization.py:241-251.
 * pts [B,N,3], trans_mat [B,4,3] -> xy [B,N,2], clamped to [0,136]. */
int disn_project(const float* pts, const float* trans_mat, int B, int N, float* xy, void* stream);

/* Row F: 5 x tf.contrib.resampler.resampler + concat --
 * models/model_normalization.py:172-190.  featmap [B,137,137,1472],
 * xy [B,N,2] -> feat [B,N,1472] ('point_img_feat').  Bit-exact with the oracle. */
int disn_gather(const float* featmap, const float* xy, int B, int N, float* feat, void* stream);

/* Rows D + E + F without the feature map (what disn_encode_query runs with featmap == NULL): project
 * pts [B,N,3] with trans_mat [B,4,3], up-sample the five taps (models/model_normalization.py:171-183) at
 * the <= 4 pixels each point touches and resample (:172-190) -> feat [B,N,1472].  Bit-identical to
 * disn_build_featmap + disn_project + disn_gather. */
int disn_gather_taps(const float* const taps[5], const float* trans_mat, const float* pts, int B, int N,
                     float* feat, void* stream);
/* The same rows in SPLIT form (round 4), the operand of the fused small-set local stream (disn_query_taps_fused):
 * feat_split holds B * N rows of 1536 * 4 bytes -- every 8 channels as [h8 | l8], the two f16 planes of
 * feature * 2^k with k from feat_amax[b] (>= max |tap| of image b; floored at 2^-20): h = f16(x), l = f16(x - h);
 * channels 1472..1535 zero.  The fp32 feature is disn_gather_taps's, bit for bit. */
int disn_gather_taps_split(const float* const taps[5], const float* trans_mat, const float* pts, int B, int N,
                           const float* feat_amax, void* feat_split, void* stream);
/* The gather of the folded local stream (see disn_fold_local): for N points of ONE image
 * h[n][:] = relu(pre[n][:] + resample(pmap_b)(project(pts[n])) + bias), all [N,512]; h may alias pre. */
int disn_gather_fold(const float* pmap_b, const float* trans_mat_b, const float* pts, int N, const float* pre,
                     const float* bias, float* h, void* stream);

/* ---------------------------------------------------------------------- *
 * Rows G + H: the two point MLPs and their sum --                          *
 * models/sdfnet.py:69-92 (scope 'sdfprediction'), :171-190 (scope          *
 * 'sdfprediction_imgfeat'), sum models/model_normalization.py:204.         *
 * g_* = global stream, l_* = local stream.  w2,w3,w5 and l_w4 are          *
 * disn_pack_kn of fold1/conv2, fold1/conv3, fold2/conv2 and the local      *
 * fold2/conv1 [1984][512].  g_w4_point = disn_pack_kn of rows 0..511 of the*
 * global fold2/conv1 [1536][512]; g_w4_global = its rows 512..1535 as is   *
 * ([1024][512] row-major): that block multiplies a per-image constant and  *
 * is folded into a per-image bias (changes summation order only).          *
 * w1 [3][64], w6 [256] are the TF tensors as is.                           *
 * ---------------------------------------------------------------------- */
typedef struct disn_mlp_weights {
  const float *g_w1, *g_b1, *g_w2, *g_b2, *g_w3, *g_b3;
  const float *g_w4_point, *g_w4_global, *g_b4, *g_w5, *g_b5, *g_w6, *g_b6;
  const float *l_w1, *l_b1, *l_w2, *l_b2, *l_w3, *l_b3;
  const float *l_w4, *l_b4, *l_w5, *l_b5, *l_w6, *l_b6;
  /* optional (NULL = not used): disn_pack_kn_x3 images of g_w2, g_w3, g_w4_point, g_w5, l_w2, l_w3,
   * l_w4, l_w5 (same matrices as the fp32 packs); see disn_vgg_weights_t.conv_w_x3 */
  const void *g_x2, *g_x3, *g_x4_point, *g_x5, *l_x2, *l_x3, *l_x4, *l_x5;
  /* optional, for the *_folded entry points: the local fold2/conv1 matrix [1984][512] as two
   * disn_pack_kn images -- rows 0..511 (point features) and rows 512..1983 (the 1472 gathered
   * channels) -- and their disn_pack_kn_x3 images (optional again) */
  const float *l_w4_point, *l_w4_feat;
  const void *l_x4_point, *l_x4_feat;
  /* optional, for the *_fused entry points: disn_mlp_fused_pack images of the two streams */
  const void *g_fused, *l_fused;
  /* optional: g_w4_global transposed, [512][1024] row-major -- the per-image bias fold as one launch (see fc_w_t) */
  const float* g_w4_global_t;
  /* optional (all eight or none): disn_pack_dense_h2 images of fold1/conv2, fold1/conv3, the point rows of the global
   * fold2/conv1 and fold2/conv2 of the global stream, and of fold1/conv2, fold1/conv3, the WHOLE local fold2/conv1
   * [1984][512] zero-padded to [2048][512] and fold2/conv2 of the local stream.  With them a point set of fewer than 8192 points per image runs
   * its layers through dense_h2.hip (disn_encode_query, disn_query, disn_sdf_mlp): one short launch per layer. */
  const void *g_d2, *g_d3, *g_d4_point, *g_d5, *l_d2, *l_d3, *l_d4, *l_d5;
  /* optional (with g_fused): disn_mlp_fused_feat_pack image of the local stream -- the FEAT form of the fused point
   * MLP for small point sets (disn_query_taps_fused; batched disn_encode_query calls) */
  const void* l_feat;
} disn_mlp_weights_t;

/* scratch for one launch over B images x N points (N per image) */
size_t disn_sdf_mlp_workspace_bytes(int B, int N);

/* pts_rot [B,N,3] ('sample_pc_rot'), embedding [B,1024], feat [B,N,1472]
 * -> sdf [B,N] = global + local; sdf_global / sdf_local optional (NULL ok). */
int disn_sdf_mlp(const disn_mlp_weights_t* w, const float* pts_rot, const float* embedding,
                 const float* feat, int B, int N, float* sdf, float* sdf_global, float* sdf_local,
                 void* ws, size_t ws_bytes, void* stream);

/* Rows D..H in one call: project pts, gather from featmap, run both MLPs.
 * Same result as disn_project + disn_gather + disn_sdf_mlp; the 1472-wide
 * feature only ever exists chunk-wise inside `ws`.
 * pts [B,N,3] is projected; pts_rot [B,N,3] feeds the MLPs (may alias pts). */
size_t disn_query_workspace_bytes(int B, int N);
int disn_query(const disn_mlp_weights_t* w, const float* featmap, const float* embedding,
               const float* trans_mat, const float* pts, const float* pts_rot, int B, int N,
               float* sdf, void* ws, size_t ws_bytes, void* stream);

/* Folded local stream.  sdfprediction_imgfeat/fold2/conv1 (models/sdfnet.py:180-182) is linear in
 * the gathered feature, and the gather (models/model_normalization.py:172-190) is a 4-tap weighted
 * sum of feature-map pixels, so
 *     feat(p) . W_feat = sum_c w_c(p) * (featmap[pixel_c(p)] . W_feat).
 * disn_fold_local multiplies ONE image's feature map with the 1472 feature rows of that layer once,
 * pmap [137*137][512] (28 GFLOP, 38 MB); the *_folded queries then gather 4 x 512 floats per point
 * from pmap instead of 4 x 1472 from featmap and run a 512-deep instead of a 1984-deep layer:
 * 1.5 MFLOP and 15 KB of traffic less per point (pays from ~2e4 points per image; the dense grid
 * has 1.7e7).  Same math re-associated: results differ from disn_query / disn_query_grid by fp32
 * rounding only (measured <= 2e-6 of the output scale; tests/test_gpu_fold.py).
 * w->l_w4_point and w->l_w4_feat must be set. */
size_t disn_fold_local_workspace_bytes(void);
int disn_fold_local(const disn_mlp_weights_t* w, const float* featmap_b, float* pmap, void* ws,
                    size_t ws_bytes, void* stream);
/* disn_query with pmap [B][137*137][512] in place of featmap; workspace as disn_query */
int disn_query_folded(const disn_mlp_weights_t* w, const float* pmap, const float* embedding,
                      const float* trans_mat, const float* pts, const float* pts_rot, int B, int N,
                      float* sdf, void* ws, size_t ws_bytes, void* stream);

/* A non-blocking HIP stream for a host that does not link HIP itself (hipStreamCreateWithFlags(hipStreamNonBlocking)
 * on the current device).  disn_amd.engine.StepPipeline creates the streams of its step contexts with it, in a
 * fixed order: ROCm hands out GPU_MAX_HW_QUEUES hardware queues (default 4, the null stream's included) in creation
 * order and shares them afterwards, and streams that share a queue serialise -- a reproducible assignment instead
 * of the lottery of a framework's stream pool. */
int disn_stream_create(void** stream);
int disn_stream_destroy(void* stream);

/* ---------------------------------------------------------------------- *
 * Concurrency context.  The path mixes MFMA-bound launches (convolutions,  *
 * MLP GEMMs) with an HBM-bound one (the 495 MB fc6-fc8 weight stream) that *
 * do not depend on each other; disn_encode_query runs the point MLPs'      *
 * embedding-independent layers on the context's auxiliary HIP stream,      *
 * forked from and joined back into the caller's `stream` with events, so   *
 * the caller still sees ONE asynchronous operation on `stream`;            *
 * disn_query_grid_ctx pipelines chunks over the two streams.               *
 * A context owns one non-blocking stream and ten events; use one context   *
 * per caller stream (not thread-safe).  disn_encode accepts a context for  *
 * symmetry and ignores it (may be NULL): the encoder alone runs on         *
 * `stream`.                                                                *
 * ---------------------------------------------------------------------- */
typedef struct disn_ctx disn_ctx_t;
int disn_ctx_create(disn_ctx_t** out);
/* Software pipeline of consecutive disn_encode_query calls on DIFFERENT contexts / streams (independent steps):
 * the step of context c starts behind `wait_event` (hipEvent_t; NULL: at once) and records `record_event`
 * (hipEvent_t; NULL: nothing) when its convolution stack is done.  Chaining step k's record to step k+1's wait
 * keeps the convolution stacks of consecutive steps from running at the same time -- their kernels are ONE round of
 * workgroups that each fill a CU and halve each other -- while step k's fc head and point MLPs run beside step
 * k+1's convolutions.  The caller enqueues the steps from ONE host thread in step order (an event must be recorded
 * before it is waited for).  disn_amd.engine.StepPipeline does exactly this. */
int disn_ctx_pipeline(disn_ctx_t* ctx, void* wait_event, void* record_event);
int disn_ctx_destroy(disn_ctx_t* ctx);

/* Rows A, B, C, E for a batch of images: disn_vgg16_forward + disn_build_featmap
 * (models/model_normalization.py:65-78 and :171-183).  featmap [B,137,137,1472]. */
size_t disn_encode_workspace_bytes(int B);
int disn_encode(disn_ctx_t* ctx, const disn_vgg_weights_t* w, const float* img, int B,
                float* resized224, float* const taps[5], float* embedding, float* featmap, void* ws,
                size_t ws_bytes, void* stream);

/* One full evaluation of the reference graph for pred_sdf, i.e. what ONE
 * sess.run([pred_sdf]) executes (test/create_sdf.py:275): rows A..H, nothing cached.
 * B*N <= 65536.  Outputs as disn_encode plus sdf [B,N].  featmap may be NULL: the
 * [B,137,137,1472] map is an intermediate of the graph, not something sess.run returns, and
 * with NULL it is never materialised -- the gather bilinearly up-samples the taps on the fly
 * (same expression, bit-identical sdf; the right trade for N up to ~10^4 per image).
 * B > 1 = B independent requests (image b, its N points, its camera) in one call -- the throughput form
 * (disn_amd.engine.StepPipeline(batch=B), bench.py --batch): the fc weights are read once per call and every launch
 * carries B images.
 * WHICH KERNELS RUN (selection rules; results of different forms agree to fp32 rounding -- all within 1e-5 of the
 * float64 oracle -- and are bit-identical only within one form):
 *   point MLPs  B >= 4 or N >= 8192, featmap == NULL (with w->g_fused and w->l_feat): the FUSED small-set kernels --
 *               split-form gather from the taps, mlp_fused_kernel<local, FEAT>, mlp_fused_kernel<global> with image b's
 *               folded bias row: one launch per stream behind the gather, bit-identical to disn_encode +
 *               disn_query_taps_fused.  ANY N (round 5): the library pads every point set to a multiple of 128 points
 *               with (0, 0, 0) -- what test/create_sdf.py:241,256 pads its last split with -- and discards those results;
 *               the fused kernels scale per point, so the real points' bits do not depend on the padding (only when the
 *               padded B * N exceeds 65536 does such a call fall through to the next rule).
 *               Otherwise, B <= 512 and N < 8192 (and the *_d* images present): the two-term f16 layers, one launch per
 *               layer -- dense_h2.hip (B < 4, or B >= 4 with a feature map asked for and N % 128 != 0), dense_h2w.hip
 *               (B >= 4, N % 128 == 0; disn_dense_h2's rule); otherwise (a feature map asked for at N >= 8192; more than
 *               512 requests of a few points each) the three-term bf16 / f32-input GEMM chain of disn_dense.  disn_query /
 *               disn_sdf_mlp switch at the same N = 8192 per image.
 *   convolutions / fc head  B < 4: conv_h2.hip + one-launch fc rows; B >= 4: conv_h2w.hip (conv2_2 .. conv4_3 and the
 *               14 x 14 layers in SEGMENTED accumulation, round 6: disn_conv3x3_h2's rule) + the split-K fc stream with up
 *               to sixteen batch rows per pass on the fp32 matrix pipe (gemv_mfma_kernel).
 *   gather      from 10 240 points per call on: one wave per point (project_gather_taps_wave_kernel); below: one thread per
 *               (point, four channels).  The SAME bits (the arithmetic is one expression tree; elementwise.hip).
 *   strict mode  vw->strict_forms == 1 (zero-initialise disn_vgg_weights_t: 0 is the default): the B < 4 forms of ALL of the
 *               above for any B -- request b's taps, embedding and sdf are then bit for bit those
 *               of a B = 1 call (tests/test_gpu_model.py::test_strict_mode_runs_the_single_image_forms); see the struct.
 *               Not needed for the 1e-5 bar since round 6 (every form holds it on the 48-set sweep); it buys reproducible
 *               bits across call sizes at ~65 % of the throughput.  Where it does NOT reach: (1) the point MLP
 *               of a call in single-stream mode (ctx == NULL) or with B > 512 falls to the three-term GEMM chain over B * N
 *               rows, whose split-K plan depends on B * N; (2) disn_query / disn_query_folded / disn_query_fused take no
 *               vgg struct: their per-image global-bias fold runs the batched fc form from B >= 4 on -- the same encoder
 *               state queried at B = 1 and at B = 4 gives pred_sdf equal to fp32 rounding (<= 1.2e-6), not bit for bit.
 * Every activation scale is per image (per point inside the fused kernels) on all of these paths, so request b's
 * outputs never depend on the other requests of its call, on its position, or on B beyond the thresholds above (B < 4:
 * bit for bit those of a B = 1 call; B >= 4: bit for bit those of any other call of >= 4 requests of the same N). */
size_t disn_encode_query_workspace_bytes(int B, int N);
int disn_encode_query(disn_ctx_t* ctx, const disn_vgg_weights_t* vw,
                      const disn_mlp_weights_t* mw, const float* img, const float* trans_mat,
                      const float* pts, const float* pts_rot, int B, int N, float* resized224,
                      float* const taps[5], float* embedding, float* featmap, float* sdf, void* ws,
                      size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------- *
 * Row J: dense grid -- test/create_sdf.py:246-256.  Flat index             *
 * k=(iz*(R+1)+iy)*(R+1)+ix -> (x_[ix], y_[iy], z_[iz]) with                *
 * x_=linspace(p0,p3,R+1) etc. evaluated in float64 then cast to float32    *
 * (bit-exact with numpy).  Writes points k0..k1-1 to pts [(k1-k0),3].      *
 * ---------------------------------------------------------------------- */
int disn_grid_points(const double* sdf_params_host, int R, int64_t k0, int64_t k1, float* pts,
                     void* stream);

/* Row J + D..H + the host '/ SDF_WEIGHT' (test/create_sdf.py:285): SDF of
 * grid points k0..k1-1 of ONE image, out[k-k0] = pred_sdf / sdf_weight
 * (IEEE float32 division; pass 1.0f for the un-divided value). */
size_t disn_query_grid_workspace_bytes(int64_t max_points);
int disn_query_grid(const disn_mlp_weights_t* w, const float* featmap, const float* embedding,
                    const float* trans_mat, const double* sdf_params_host, int R, int64_t k0,
                    int64_t k1, float sdf_weight, float* out, void* ws, size_t ws_bytes,
                    void* stream);

/* disn_query_grid with pmap [137*137][512] (disn_fold_local) in place of featmap; workspace as
 * disn_query_grid */
int disn_query_grid_folded(const disn_mlp_weights_t* w, const float* pmap, const float* embedding,
                           const float* trans_mat, const double* sdf_params_host, int R,
                           int64_t k0, int64_t k1, float sdf_weight, float* out, void* ws,
                           size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------- *
 * Fused point MLP (mlp_fused.hip): rows D, F (folded), G, H of one stream  *
 * in ONE persistent kernel -- models/sdfnet.py:71-88 ('sdfprediction') and *
 * :173-186 ('sdfprediction_imgfeat'), sum models/model_normalization.py:204*
 * -- with every activation in registers; only the weights move (L2 -> LDS  *
 * ring -> MFMA).  fp32-accurate products from a two-term fp16 split of     *
 * every operand (x = h + l, 23 significant bits; power-of-two scales per   *
 * layer for the weights and per point for the activations), fp32           *
 * layer -- round 4: per OUTPUT FEATURE -- for the weights and per point    *
 * for the activations), fp32                                               *
 * accumulation.  disn_mlp_fused_pack builds one stream's weight image from *
 * its four MFMA-shaped layers in TF [K][N] layout: fold1/conv2 [64][256],  *
 * fold1/conv3 [256][512], the 512 point rows of fold2/conv1 [512][512],    *
 * fold2/conv2 [512][256]; set w->g_fused / w->l_fused.  The local stream   *
 * gathers from pmap (disn_fold_local) and needs max|pmap| of each image    *
 * (disn_amax) for its activation-scale bound.                              *
 * ---------------------------------------------------------------------- */
size_t disn_mlp_fused_image_bytes(void);
int disn_mlp_fused_pack(const float* w2, const float* w3, const float* w4_point, const float* w5, void* image,
                        void* stream);
/* The FEAT form (round 4): the local stream of a SMALL point set -- B images x N points, N a multiple of 128, B * N <=
 * 65536 -- without a feature map or a folded map (models/sdfnet.py:173-186 on the concat of :180, gather
 * models/model_normalization.py:171-190).  The gather from the taps writes the 1472 features of a point in SPLIT form
 * (two f16 planes of feature * 2^k, one k per image from the taps' maxima: every gathered feature is a convex combination
 * of tap values) and the kernel takes them as 96 extra reduction blocks of fold2/conv1 straight from memory: one launch
 * for the whole batch, every activation in registers.  disn_mlp_fused_feat_pack: as disn_mlp_fused_pack with w4 = the
 * WHOLE fold2/conv1 matrix [512 + 1472][512]; set w->l_feat.  disn_query_taps_fused: rows D..H from the five taps
 * (disn_vgg16_forward / disn_encode outputs) -- gather, local stream, folded global bias, global stream with image b's
 * bias row, sum; same result as disn_query on the feature map of the same taps up to fp32 rounding (|gpu - f64| <= 1e-5
 * on the He-weight configurations, tests/test_gpu_fused.py).  disn_encode_query runs the same launches for calls of
 * >= 4 images (bit-identical to disn_encode + disn_query_taps_fused). */
size_t disn_mlp_fused_feat_image_bytes(void);
int disn_mlp_fused_feat_pack(const float* w2, const float* w3, const float* w4, const float* w5, void* image,
                             void* stream);
size_t disn_query_taps_fused_workspace_bytes(int B, int N);
int disn_query_taps_fused(const disn_mlp_weights_t* w, const float* const taps[5], const float* embedding,
                          const float* trans_mat, const float* pts, const float* pts_rot, int B, int N, float* sdf,
                          void* ws, size_t ws_bytes, void* stream);
/* *out = max |x[i]|, n a multiple of 4 */
int disn_amax(const float* x, int64_t n, float* out, void* stream);
/* disn_query_folded / disn_query_grid_folded through the fused kernels (two launches per image: global
 * stream, then local stream + sum + '/ sdf_weight').  pmap [B][137*137][512], pmap_amax [B].  Same math as
 * the *_folded entry points; differs from them by fp32 rounding (tests/test_gpu_fused.py). */
size_t disn_query_fused_workspace_bytes(int B, int64_t N);
int disn_query_fused(const disn_mlp_weights_t* w, const float* pmap, const float* pmap_amax,
                     const float* embedding, const float* trans_mat, const float* pts, const float* pts_rot,
                     int B, int64_t N, float* sdf, void* ws, size_t ws_bytes, void* stream);
size_t disn_query_grid_fused_workspace_bytes(int64_t max_points);
int disn_query_grid_fused(const disn_mlp_weights_t* w, const float* pmap, const float* pmap_amax,
                          const float* embedding, const float* trans_mat, const double* sdf_params_host, int R,
                          int64_t k0, int64_t k1, float sdf_weight, float* out, void* ws, size_t ws_bytes,
                          void* stream);

/* disn_query_grid, chunk-pipelined over the context's two streams: the HBM-bound front of chunk
 * i+1 (grid points, projection, gather) runs under the MFMA-bound MLP of chunk i.  Same result. */
size_t disn_query_grid_ctx_workspace_bytes(int64_t max_points);
int disn_query_grid_ctx(disn_ctx_t* ctx, const disn_mlp_weights_t* w, const float* featmap,
                        const float* embedding, const float* trans_mat,
                        const double* sdf_params_host, int R, int64_t k0, int64_t k1,
                        float sdf_weight, float* out, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------- *
 * Iso-surface: the reference writes the grid to a .dist     *
 * file and shells out to the closed Vega-FEM binary                        *
 * ./isosurface/computeMarchingCubes <dist> <obj> -i <iso>                  *
 * (test/create_sdf.py:305-323).  Here the grid [(R+1)^3] (x fastest, as    *
 * produced by disn_query_grid) is meshed where it lies: indexed marching   *
 * cubes, one vertex per cut grid edge, case table derived in               *
 * tools/gen_mc_tables.py (crack-free).  Two steps because the sizes are    *
 * data dependent:                                                          *
 *   disn_mc_count -> counts[0] = #vertices, counts[1] = #triangles (uint64,*
 *                    DEVICE memory; read them back, allocate, then)        *
 *   disn_mc_emit  -> verts [nv,3] float32 world coordinates (bbox =        *
 *                    sdf_params), faces [nt,3] int32 0-based, outward      *
 *                    orientation (normals point towards larger values).    *
 * Both calls take the same workspace (disn_mc_workspace_bytes(R)); emit    *
 * relies on what count left in it.                                         *
 * ---------------------------------------------------------------------- */
size_t disn_mc_workspace_bytes(int R);
int disn_mc_count(const float* sdf, int R, float iso, uint64_t* counts, void* ws, size_t ws_bytes,
                  void* stream);
int disn_mc_emit(const float* sdf, const double* sdf_params_host, int R, float iso, float* verts,
                 int32_t* faces, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------- *
 * Training step (BASELINE config 5): what ONE                *
 * sess.run([train_op, losses...]) of train/train_sdf.py:371-387 executes.  *
 * Every variable of the graph (train/train_sdf.py:266-268 minimises over   *
 * ALL globals, the VGG is fine-tuned) lives in ONE flat float32 device     *
 * buffer, each in the reference's own TF layout ([kh,kw,Cin,Cout] =        *
 * [K][N] row-major; biases [N]), at the offsets disn_param_layout returns  *
 * (offsets are multiples of 64 floats; the gaps are never written: keep    *
 * them zero).  Variable order:                                             *
 *   0..25   vgg_16/conv{1..5}/conv{i}_{j}/{weights,biases}                 *
 *   26..31  vgg_16/{fc6,fc7,fc8}/{weights,biases}                          *
 *   32..43  sdfprediction/{fold1/conv1,fold1/conv2,fold1/conv3,            *
 *           fold2/conv1,fold2/conv2,fold2/conv5}/{weights,biases}          *
 *   44..55  sdfprediction_imgfeat/ (same six layers)                       *
 * `grads` and the Adam slots m, v have the same layout, so data-parallel   *
 * training all-reduces ONE buffer and updates with ONE kernel.             *
 * ---------------------------------------------------------------------- */
#define DISN_NUM_VARS 56
typedef struct disn_param_layout {
  int64_t offset[DISN_NUM_VARS]; /* in floats */
  int64_t count[DISN_NUM_VARS];
  int64_t total; /* floats, multiple of 64 */
} disn_param_layout_t;
int disn_param_layout(disn_param_layout_t* out);

/* Forward (rows A..H, activations kept in ws), get_loss
 * (models/model_normalization.py:254-300) and the gradient of overall_loss
 * w.r.t. every variable, written to `grads` (layout above; weight-decay term
 * wd*w included for every '/weights' variable).  B*N <= 65536.
 * gt [B,N] = the fed 'sdf' (sdf_val - 0.003, train/train_sdf.py:375).
 * pred [B,N] = pred_sdf (un-divided).  losses: 5 device floats =
 * {accuracy, sdf_loss_realvalue, sdf_loss, regularization, overall_loss}.
 * compute_bf16: 0 = every product on the f32-input MFMA (implicit-GEMM convolutions).
 *   2 = the same fp32 accuracy, faster: the convolutions of the forward (conv1_1 excepted: direct fp32) AND their
 *   data gradients (dx = conv(dz, mirrored transposed kernel)) through the inference kernels conv_h2.hip /
 *   conv_h2w.hip (two-term f16 split, per-image operand scales; the batched form from four samples on), the
 *   point-MLP forward / data-gradient GEMMs as a three-term bf16 split on the bf16 MFMA pipes.
 *   1 = mixed precision as BASELINE config 5 names it: the point-MLP forward / data-gradient GEMMs multiply in bf16
 *   with fp32 accumulation; the convolutions' forward and data gradients stay on the f16-split kernels of mode 2
 *   (measured faster than a one-term bf16 implicit GEMM); parameters, activations, gradients and the optimizer
 *   stay fp32 ("fp32 master"); weight gradients use bf16 where the 128x128 tile applies and the fp32 MFMA
 *   otherwise.
 * ctx (may be NULL): the HBM-bound side work (weight-norm sum, fc6-fc8 forward and backward) runs on
 *   the context's auxiliary stream under the MFMA-bound GEMMs; the caller still sees one
 *   asynchronous operation on `stream`.
 * head_ready_event (may be NULL): a hipEvent_t recorded on `stream` at the point where every
 *   gradient from offset[26] on (fc6..fc8 and both MLPs, 96 % of the bytes) is final, before the
 *   convolution backward: a data-parallel caller starts reducing that part under the rest of the
 *   step.  Gradients of the MLPs and fc layers are bit-reproducible; those of the convolutions
 *   depend on fp32 atomics (resampler gradient) in their last bits. */
size_t disn_train_workspace_bytes(int B, int N);
int disn_train_step(disn_ctx_t* ctx, const float* params, float* grads, const float* img,
                    const float* trans_mat, const float* pts, const float* pts_rot, const float* gt, int B,
                    int N, float wd, float sdf_weight, float mask_weight, int compute_bf16, float* pred,
                    float* losses, void* head_ready_event, void* ws, size_t ws_bytes, void* stream);

/* tf.train.AdamOptimizer update (train/train_sdf.py:251) on n floats (n % 4 == 0):
 * g = grads*grad_scale; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 * params -= lr_t * m / (sqrt(v) + eps), lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller. */
int disn_adam_update(float* params, const float* grads, float* m, float* v, int64_t n, float lr_t,
                     float beta1, float beta2, float eps, float grad_scale, void* stream);

/* Building blocks of the step (unit-test / composition surface).
 * disn_dense_backward: one [1,1] conv layer out = act(a W + b).  dy [M][N] is the gradient w.r.t.
 *   the layer OUTPUT; with y (the saved post-ReLU output) non-NULL it is masked in place to the
 *   pre-activation gradient first.  Outputs db [N], dw [K][N] (+ wd*w), da [M][K] (NULL: skipped).
 *   a [M][lda] (first K columns), w_kn raw [K][N]; K, N multiples of 64.
 * disn_conv3x3_backward: same for a SAME 3x3 conv, x [B,H,W,Cin], w_hwio [3,3,Cin,Cout],
 *   Cin == 3 (dx must be NULL) or a multiple of 64, Cout a multiple of 64.
 * compute_bf16 (as in disn_train_step; ignored for Cin == 3): 1 = the GEMMs of the block multiply in bf16 (fp32
 *   accumulate), 2 = three-term bf16 split; disn_conv3x3_backward with 1 or 2: dx through conv_h2.hip /
 *   conv_h2w.hip (two-term f16 split, fp32-accurate) as the step does. */
size_t disn_dense_backward_workspace_bytes(int M, int K, int N);
int disn_dense_backward(const float* a, int lda, int K, const float* w_kn, const float* y, float* dy,
                        int M, int N, float wd, int compute_bf16, float* da, float* dw, float* db,
                        void* ws, size_t ws_bytes, void* stream);
size_t disn_conv3x3_backward_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int disn_conv3x3_backward(const float* x, int B, int H, int W, int Cin, const float* w_hwio,
                          const float* y, float* dy, int Cout, float wd, int compute_bf16, float* dx,
                          float* dw, float* db, void* ws, size_t ws_bytes, void* stream);
/* bf16-compute forms of disn_dense / disn_conv3x3 used by the mixed-precision training step
 * (fp32 tensors in HBM, operands rounded to bf16 when staged, fp32 accumulate, fp32 out;
 * v_mfma_f32_32x32x16_bf16).  They take the RAW TF weights ([K][N] / [3,3,Cin,Cout]) and pack them
 * into `ws`.  k1, k2, Cin multiples of 32; N, Cout multiples of 64.
 * nsplit = 1: plain bf16 product.  nsplit = 3: fp32-accurate product on the bf16 pipes (each operand
 * split into three bf16 terms, six of the nine cross terms accumulated: error ~ one fp32 rounding). */
size_t disn_dense_bf16_workspace_bytes(int M, int K, int N);
int disn_dense_bf16(const float* a1, int lda1, int k1, const float* a2, int lda2, int k2, int M,
                    const float* w_kn, const float* bias, int N, int relu, int nsplit, float* out, void* ws,
                    size_t ws_bytes, void* stream);
size_t disn_conv3x3_bf16_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int disn_conv3x3_bf16(const float* in, int B, int H, int W, int Cin, const float* w_hwio,
                      const float* bias, int Cout, int relu, int nsplit, float* out, void* ws,
                      size_t ws_bytes, void* stream);
/* dx [B,H,W,C]: dy routed to the first maximum of each 2x2 window of x, zero elsewhere (H, W even) */
int disn_maxpool2x2_backward(const float* x, const float* dy, int B, int H, int W, int C, float* dx,
                             void* stream);
/* din [B,Hin,Win,C] (= or +=) gradient of disn_resize_bilinear w.r.t. its input, given dout (channels
 * [out_coff, out_coff+C) of a [B,Hout,Wout,out_cstride] tensor).  C, strides multiples of 4.
 * ws: scratch of the separable form used for >= 2x up-sampling (0 bytes otherwise). */
size_t disn_resize_bilinear_backward_workspace_bytes(int B, int Hin, int Win, int C, int Hout, int Wout);
int disn_resize_bilinear_backward(const float* dout, int B, int Hin, int Win, int C, int Hout, int Wout,
                                  int out_cstride, int out_coff, float* din, int accumulate, void* ws,
                                  size_t ws_bytes, void* stream);
/* dfeatmap [B,137,137,1472] = gradient of disn_gather w.r.t. featmap (zeroed here, then fp32 atomics) */
int disn_gather_backward(const float* dfeat, const float* xy, int B, int N, float* dfeatmap,
                         void* stream);

/* ---------------------------------------------------------------------- *
 * Camera head: the estimated-camera source of `trans_mat`.  *
 * models/posenet.py:91-124 get_cam_mat on the VGG embedding [B,1024]       *
 * (towers scale 1024-64-32-1, ortho6d 1024-512-256-6, translation          *
 * 1024-128-64-3 + const, 6-D -> rotation by Gram-Schmidt :22-36) and       *
 * cam_est/model_cam.py:102-103 pred_trans_mat = pred_RT @ K^T.             *
 * Weights: tf_util.fully_connected variables                               *
 * 'cameraprediction/<tower>/fc{1,2,3}/{weights [in,out], biases [out]}'    *
 * as they are.  K_host: 9 floats (3x3 row-major, HOST memory), NULL = the  *
 * reference's constant (model_cam.py:28).  Outputs rotation [B,3,3],       *
 * translation [B,3], RT [B,4,3], trans_mat [B,4,3].                        *
 * ---------------------------------------------------------------------- */
typedef struct disn_cam_weights {
  const float *s_w1, *s_b1, *s_w2, *s_b2, *s_w3, *s_b3; /* scale */
  const float *r_w1, *r_b1, *r_w2, *r_b2, *r_w3, *r_b3; /* ortho6d */
  const float *t_w1, *t_b1, *t_w2, *t_b2, *t_w3, *t_b3; /* translation */
} disn_cam_weights_t;
int disn_cam_head(const disn_cam_weights_t* w, const float* embedding, const float* K_host, int B,
                  float* rotation, float* translation, float* RT, float* trans_mat, void* stream);

/* ---------------------------------------------------------------------- *
 * Host utility (ABI 9): EQUALISED inference weights.  An exact power-of-two *
 * re-parametrisation of the hidden channels of the network                  *
 * (models/model_normalization.py:74-78,171-204; models/sdfnet.py:71-88,     *
 * 173-186), applied IN PLACE to a host COPY of the variables in their TF    *
 * layouts before they are uploaded / packed (the training state and         *
 * checkpoints keep the true values).  Channel f of every hidden layer is    *
 * multiplied by c[f] = the power of two (<= 2^16) that brings the largest   *
 * entry of its weight column (rows already divided by the producer's        *
 * factors) to the binade of the layer's MEDIAN column; every consumer row   *
 * of that channel (next convolution, fc6, the tap rows of                   *
 * sdfprediction_imgfeat/fold2/conv1, the next MLP layer) is divided by it.  *
 * ReLU / max-pool / resize / resampler commute with a positive per-channel  *
 * factor and powers of two are exact: the network function and every fp32   *
 * rounding are unchanged; what changes is that the channels of a hidden     *
 * tensor have comparable magnitudes, which the per-image power-of-two       *
 * operand scale of the two-term f16 kernels needs (22 bits down to 2^-17 of *
 * the tensor maximum; trained channel gains may span more: DESIGN 4k).      *
 * Layers: 13 convolutions, then fold1/conv1..fold2/conv2 of both streams;   *
 * fc6..fc8 and fold2/conv5 produce no equalised channels.                   *
 * With such weights the TAPS a library call returns (and everything         *
 * gathered from them) are in equalised units: true = value / tap_scale[ch]  *
 * (disn_scale_channels), ch = the 1472 concatenated tap channels.           *
 * span_log2 (optional, 23 floats): log2(largest / smallest factor) of each  *
 * layer = the spread of channel gains the weights carried.  Columns more    *
 * than 2^16 BELOW the median are scaled up by 2^16 only (a nearly dead      *
 * column with an ordinary bias must not become the tensor's maximum); they  *
 * keep a residual gain -- still exact, full precision down to 2^-33.        *
 * ---------------------------------------------------------------------- */
typedef struct disn_eq_weights {
  float* conv_w[13]; /* HOST, [3,3,Cin,Cout], modified in place */
  float* conv_b[13];
  float* fc6_w;      /* HOST, [7,7,512,4096]: rows divided by conv5_3's factors; NULL = not present */
  float* mlp_w[2][6]; /* [0]: sdfprediction, [1]: sdfprediction_imgfeat; fold1/conv1..3, fold2/conv1, conv2, conv5 */
  float* mlp_b[2][6];
  int num_classes;   /* embedding width (1024): sdfprediction/fold2/conv1 has 512 + num_classes rows */
} disn_eq_weights_t;
int disn_equalise_weights(const disn_eq_weights_t* w, float* tap_scale_host, float* span_log2_host);

/* out[r][c] = in[r][c] * scale[c] (invert = 0) or in[r][c] / scale[c] (invert != 0; IEEE divide -- exact for the
 * powers of two of disn_equalise_weights): rows x C floats, device pointers, in == out allowed.  Converts taps
 * (C = the tap's channels, scale = tap_scale + the tap's offset) or gathered features (C = 1472) between equalised and
 * true units. */
int disn_scale_channels(const float* in, int64_t rows, int C, const float* scale, int invert, float* out, void* stream);

/* Host utility: Wavefront .obj writer ("v x y z" / "f a b c", 1-based) for HOST arrays; the
 * reference's output artefact (test/create_sdf.py:311).  Returns 0, or DISN_E_ARG on I/O error. */
int disn_write_obj(const char* path, const float* verts_host, int64_t nv, const int32_t* faces_host,
                   int64_t nf);

/* Host utility (no device work): CRC-32C of a HOST buffer, continuing from `crc` (0 to start);
 * the checksum of TensorFlow's table blocks and tensor-bundle entries, used by the
 * TensorFlow-free checkpoint reader/writer (train/train_sdf.py:285-299, test/create_sdf.py:180-192). */
uint32_t disn_crc32c(const void* data_host, size_t n, uint32_t crc);

#ifdef __cplusplus
}
#endif
#endif /* DISN_AMD_H */
