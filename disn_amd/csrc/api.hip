// extern "C" entry points of include/disn_amd.h: argument checking, workspace carving and
// the launch sequences.  No allocation, no global mutable state, no host synchronisation.
#include "../../include/disn_amd.h"

#include "kernels.hpp"
#include "tuning.hpp"

#include <new>

using namespace disn;

#define DISN_TRY(expr)                    \
  do {                                    \
    hipError_t _e = (expr);               \
    if (_e != hipSuccess) return (int)_e; \
  } while (0)

namespace {

struct Bump {  // carve a caller-provided workspace; base == nullptr just measures
  char* base;
  size_t off;
  explicit Bump(void* b) : base(static_cast<char*>(b)), off(0) {}
  float* take(size_t bytes) {
    off = (off + 255) & ~size_t(255);
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += bytes;
    return p;
  }
};

// ---- VGG-16 -----------------------------------------------------------------
struct VggLayer {
  int cin, cout, hw, tap;  // tap index or -1
};
const VggLayer kVgg[13] = {
    {3, 64, 224, -1},   {64, 64, 224, 0},    {64, 128, 112, -1},  {128, 128, 112, 1},
    {128, 256, 56, -1}, {256, 256, 56, -1},  {256, 256, 56, 2},   {256, 512, 28, -1},
    {512, 512, 28, -1}, {512, 512, 28, 3},   {512, 512, 14, -1},  {512, 512, 14, -1},
    {512, 512, 14, 4}};
const bool kPoolAfter[13] = {false, true, false, true, false, false, true,
                             false, false, true, false, false, true};

inline int conv_k(int cin) { return cin == 3 ? 32 : 9 * cin; }

struct VggWs {
  float *resized, *bufA, *bufB, *bufP, *gemm_ws, *fc_ws, *fc6, *fc7;
  float* amax;  // [14][B][64]: slot group (i, b) = max |input of conv layer i| of image b (conv_h2.hip), written by
                // layer i - 1: per image, so that an image's scales (hence every bit of its result) do not depend
                // on the batch it travels in
  size_t total;
};

VggWs vgg_layout(void* ws, int B, int num_classes) {
  Bump b(ws);
  VggWs w;
  const size_t big = (size_t)B * 224 * 224 * 64 * sizeof(float);
  w.resized = b.take((size_t)B * 224 * 224 * 3 * sizeof(float));
  w.bufA = b.take(big);
  w.bufB = b.take(big / 4);  // only used from conv3 on (56x56x256 = big/4)
  w.bufP = b.take(big / 4);
  size_t gws = 0;
  for (const VggLayer& L : kVgg) {
    const GemmPlan pl = gemm_plan(B * L.hw * L.hw, L.cout, conv_k(L.cin));
    if (pl.ws_bytes > gws) gws = pl.ws_bytes;
    if (L.cin != 3) {
      const size_t bw = gemm_bf16_ws_bytes(B * L.hw * L.hw, L.cout, 9 * L.cin);
      if (bw > gws) gws = bw;
    }
  }
  w.gemm_ws = b.take(gws);
  size_t fws = gemv_ws_bytes(B, 25088, 4096);
  const size_t f7 = gemv_ws_bytes(B, 4096, 4096), f8 = gemv_ws_bytes(B, 4096, num_classes);
  if (f7 > fws) fws = f7;
  if (f8 > fws) fws = f8;
  w.fc_ws = b.take(fws);
  w.fc6 = b.take((size_t)B * 4096 * sizeof(float));
  w.fc7 = b.take((size_t)B * 4096 * sizeof(float));
  w.amax = b.take((size_t)14 * B * 64 * sizeof(float));
  w.total = (b.off + 255) & ~size_t(255);
  return w;
}

bool x3_enabled();

int conv3x3_impl(const float* in, int B, int H, int W, int Cin, const float* w_packed,
                 const float* bias, int Cout, int relu, float* out, float* ws, size_t ws_bytes,
                 hipStream_t st, const void* x3 = nullptr, float* pool_out = nullptr,
                 bool* pooled = nullptr) {
  if (pooled) *pooled = false;
  if (!in || !w_packed || !bias || !out || B <= 0 || H <= 0 || W <= 0) return DISN_E_ARG;
  if (!(Cin == 3 || Cin % 32 == 0) || Cout % 64 != 0 || H >= 32768 || W >= 32768)
    return DISN_E_SHAPE;
  GemmParams p{};
  p.a1 = in;
  p.H = H; p.W = W; p.Cin = Cin;
  p.M = B * H * W; p.N = Cout; p.K = conv_k(Cin);
  p.bp = w_packed; p.bias = bias; p.rows_per_bias = 0;
  p.out = out; p.ldc = Cout; p.relu = relu;
  // measured at B = 1 and B = 8 (tools/bf16_time.py, prepacked): the three-term kernel wins on every
  // layer with Cin >= 64
  if (x3 && Cin != 3 && x3_enabled()) {
    DISN_TRY(gemm_bf16_launch(p, GEMM_CONV3, x3, ws, ws ? ws_bytes : 0, st, 3, pool_out, pooled));
    return 0;
  }
  const GemmPlan pl = gemm_plan(p.M, p.N, p.K, ws ? ws_bytes : 0);
  DISN_TRY(gemm_launch(p, Cin == 3 ? GEMM_CONV3_C3 : GEMM_CONV3, pl, ws, st));
  return 0;
}

// ---- point MLP ----------------------------------------------------------------
const int kChunk = 65536;  // points per MLP pass inside disn_query / disn_sdf_mlp

const int kH2Imgs = 512;  // images per call the dense_h2 path keeps maximum slots for (512 x 128 points = kChunk)

struct MlpWs {
  // local stream: e1l -> h256 -> h512a -> (with feat) h512b -> l5
  // global stream: e1g -> g256 -> g512 -> (with the folded bias) h512b -> g5
  float *e1g, *e1l, *h256, *h512a, *h512b, *g256, *g512, *g5, *l5, *gemm_ws;
  // split global fold2/conv1 (disn_encode_query): g512 . W4_point before the embedding exists, the
  // folded bias + ReLU once it does; the tail then runs beside phase 1, on its own GEMM scratch
  float *g4pre, *zero512, *gemm_ws2;
  // dense_h2 path (small point sets): 16 x 64 activation-maximum slots, directly in front of zero512 so that the
  // chain's first launch clears both in one go.  Slot groups: 0 e1g, 1 e1l, 2 g256, 3 h256, 4 g512, 5 h512a,
  // 6 g4pre, 7 feat, 8 h512b (local fold2/conv1 out).  One set of 1024 floats per image (the maxima are per image so
  // that a batch gives every image exactly what it would get alone), kH2Imgs sets back to back, zero512 behind.
  float* amax;
  size_t gemm_ws_bytes, total;
};

size_t mlp_gemm_ws(int n) {
  // the split-K plan depends on the row count; a ragged last chunk may want slabs although the
  // full chunk does not, so size for every power-of-two row count up to n (and n itself).
  // mlp_chunk() additionally plans WITHIN this capacity, so the size is only a speed matter.
  size_t m = 0;
  const int shapes[5][2] = {{256, 64}, {512, 256}, {512, 512}, {512, 1984}, {256, 512}};
  for (int rows = 1;; rows *= 2) {
    const int r = rows < n ? rows : n;
    for (auto& s : shapes) {
      const GemmPlan pl = gemm_plan(r, s[0], s[1]);
      if (pl.ws_bytes > m) m = pl.ws_bytes;
      const size_t bw = gemm_bf16_ws_bytes(r, s[0], s[1]);  // split-K partials of the 3xBF16 kernel
      if (bw > m) m = bw;
    }
    if (rows >= n) break;
  }
  return m;
}

MlpWs mlp_layout(Bump& b, int n, bool split_g4 = false) {
  MlpWs w;
  const size_t f = sizeof(float);
  w.e1g = b.take((size_t)n * 64 * f);
  w.e1l = b.take((size_t)n * 64 * f);
  w.h256 = b.take((size_t)n * 256 * f);
  w.h512a = b.take((size_t)n * 512 * f);
  w.h512b = b.take((size_t)n * 512 * f);
  w.g256 = b.take((size_t)n * 256 * f);
  w.g512 = b.take((size_t)n * 512 * f);
  w.g5 = b.take((size_t)n * 256 * f);
  w.l5 = b.take((size_t)n * 256 * f);
  w.gemm_ws_bytes = mlp_gemm_ws(n);
  w.gemm_ws = b.take(w.gemm_ws_bytes);
  w.g4pre = split_g4 ? b.take((size_t)n * 512 * f) : nullptr;
  w.amax = b.take((size_t)kH2Imgs * 16 * 64 * f);
  w.zero512 = b.take(512 * f);
  w.gemm_ws2 = split_g4 ? b.take(w.gemm_ws_bytes) : nullptr;
  w.total = b.off;
  return w;
}

bool mlp_weights_ok(const disn_mlp_weights_t* w) {
  if (!w) return false;
  const float* const* p = reinterpret_cast<const float* const*>(w);
  for (size_t i = 0; i < 25; ++i)  // g_w1 .. l_b6; the *_x* images that follow are optional
    if (!p[i]) return false;
  return true;
}

// fp32-accurate products on the bf16 MFMA pipes (three bf16 terms per operand, gemm_bf16_mfma.hip)
// wherever the caller supplied the 3-plane weight image
bool x3_enabled() { return tune::x3 != 0; }

// The point-MLP layers of a SMALL point set (a few thousand rows: launch- and latency-bound in the GEMM kernels)
// through dense_h2.hip when the weights carry its images: one short launch per layer, the two streams' same-shaped
// layers paired in one launch, the per-image bias + ReLU of the split global fold2/conv1 applied by the consumer.
bool mlp_h2(const disn_mlp_weights_t* w, long n) {
  return x3_enabled() && n < 8192 && w->g_d2 && w->g_d3 && w->g_d4_point && w->g_d5 && w->l_d2 && w->l_d3 &&
         w->l_d4 && w->l_d5;
}

DenseH2Prob h2_prob(const float* a, int K, const void* img, const float* bias, int N, int relu, const float* in_amax,
                    float* out, float* out_amax, int n) {
  DenseH2Prob p{};
  p.a = a; p.lda = K; p.k1 = K; p.wimg = static_cast<const unsigned char*>(img); p.bias = bias;
  p.in_amax = in_amax; p.out = out; p.ldc = N; p.out_amax = out_amax; p.M = n; p.N = N; p.K = K; p.relu = relu;
  return p;
}

// The h2 phases work on `imgs` images of n rows each (rows image-major, first row o): ONE launch per layer for the
// whole batch when n is a multiple of 64 (a 64-row tile then lies in one image and picks that image's maxima),
// otherwise the caller loops over the images.  Either way the activation maxima -- hence the scales, hence every bit
// of an image's result -- are the image's own: independent of the batch and of the entry point that runs the layers.
float* h2_slots(const MlpWs& s, int b) { return s.amax + (size_t)b * 1024; }
constexpr int kFeatMaxSlot = 576;  // floats 576 .. 1023 of a set: up to 448 per-workgroup maxima of the gather

DenseH2Prob h2_batched(DenseH2Prob p, int imgs, int n, bool no_wide = false) {
  if (imgs > 1) { p.M = imgs * n; p.amax_rows = n; p.amax_stride = 1024; }
  p.no_wide = no_wide ? 1 : 0;   // "strict" (disn_vgg_weights_t.strict_forms = 1): never the batched form of dense_h2w.hip
  return p;
}

// fold1 of BOTH streams: embedding (+ its maxima per image, + clearing the later layers' slots and the zero bias
// row), then conv2 and conv3 of the two streams as two paired launches
int mlp_fold1_h2(const disn_mlp_weights_t* w, const float* pts_rot, int n, const MlpWs& s, int b, size_t o, int imgs,
                 hipStream_t st, bool nw = false) {
  float* A = h2_slots(s, b);
  DISN_TRY(pt_embed_launch(pts_rot, (int64_t)imgs * n, w->g_w1, w->g_b1, w->l_w1, w->l_b1, s.e1g + o * 64, s.e1l + o * 64, st,
                           A, b == 0 ? s.zero512 : nullptr, b == 0 ? 512 : 0, imgs));
  DenseH2Prob p2[2] = {h2_batched(h2_prob(s.e1g + o * 64, 64, w->g_d2, w->g_b2, 256, 1, A + 0, s.g256 + o * 256, A + 128, n), imgs, n, nw),
                       h2_batched(h2_prob(s.e1l + o * 64, 64, w->l_d2, w->l_b2, 256, 1, A + 64, s.h256 + o * 256, A + 192, n), imgs, n, nw)};
  DISN_TRY(dense_h2_launch(p2, 2, st));
  DenseH2Prob p3[2] = {h2_batched(h2_prob(s.g256 + o * 256, 256, w->g_d3, w->g_b3, 512, 1, A + 128, s.g512 + o * 512, A + 256, n), imgs, n, nw),
                       h2_batched(h2_prob(s.h256 + o * 256, 256, w->l_d3, w->l_b3, 512, 1, A + 192, s.h512a + o * 512, A + 320, n), imgs, n, nw)};
  DISN_TRY(dense_h2_launch(p3, 2, st));
  return 0;
}

// the point half of the global fold2/conv1: g512 . W4_point, no bias, no ReLU -> `pre` (+ its maximum)
int mlp_g4_pre_h2(const disn_mlp_weights_t* w, int n, const MlpWs& s, int b, size_t o, int imgs, float* pre,
                  hipStream_t st, bool nw = false) {
  float* A = h2_slots(s, b);
  const DenseH2Prob p = h2_batched(h2_prob(s.g512 + o * 512, 512, w->g_d4_point, s.zero512, 512, 0, A + 256, pre, A + 384, n), imgs, n, nw);
  DISN_TRY(dense_h2_launch(&p, 1, st));
  return 0;
}

// local fold2/conv1 on [h512a | feat] read in place, fold2/conv2.
// feat rows have feat_ld floats: 1472 (K = 1984, 64-column chunks) or 1536 with zero padding (K = 2048, 256-column
// chunks).  A k-wave sums its k16 blocks in ascending order either way and the padding adds exact zeros: the two
// forms give the same bits.  l_d4 is the [1984][512] matrix packed with 2048 rows (zero rows at the end).
// feat_amax_done: the gather left its per-workgroup maxima of |feat| in the set's free tail (project_gather_taps_kernel)
int mlp_phase1_h2(const disn_mlp_weights_t* w, int n, const float* feat, int feat_ld, const MlpWs& s, int b, size_t o,
                  int imgs, hipStream_t st, bool feat_amax_done = false, bool nw = false) {
  float* A = h2_slots(s, b);
  if (!feat_amax_done) DISN_TRY(amax64_accumulate_launch(feat, (size_t)n * feat_ld, A + 448, st, imgs, 1024));
  DenseH2Prob p4 = h2_prob(s.h512a + o * 512, 512 + feat_ld, w->l_d4, w->l_b4, 512, 1, A + 320, s.h512b + o * 512,
                           A + 512, n);
  p4.lda = 512; p4.k1 = 512; p4.a2 = feat; p4.lda2 = feat_ld; p4.in_amax2 = A + 448; p4.Kimg = 2048;
  if (feat_amax_done) { p4.in_amax2 = A + kFeatMaxSlot; p4.in_amax2_n = project_gather_taps_amax_blocks(n, feat_ld); }
  p4 = h2_batched(p4, imgs, n, nw);
  DISN_TRY(dense_h2_launch(&p4, 1, st));
  const DenseH2Prob p5 = h2_batched(h2_prob(s.h512b + o * 512, 512, w->l_d5, w->l_b5, 256, 1, A + 512, s.l5 + o * 256, nullptr, n), imgs, n, nw);
  DISN_TRY(dense_h2_launch(&p5, 1, st));
  return 0;
}

// global fold2/conv2 on relu(pre + the image's bias row): the deferred bias + ReLU
int mlp_g5_h2(const disn_mlp_weights_t* w, int n, const float* pre, const float* gbias_b, const MlpWs& s, int b, size_t o,
              int imgs, hipStream_t st, bool nw = false) {
  DenseH2Prob p = h2_prob(pre, 512, w->g_d5, w->g_b5, 256, 1, h2_slots(s, b) + 384, s.g5 + o * 256, nullptr, n);
  p.in_bias = gbias_b;
  if (imgs > 1) p.in_bias_rows = n;
  p = h2_batched(p, imgs, n, nw);
  DISN_TRY(dense_h2_launch(&p, 1, st));
  return 0;
}

int dense_layer(const float* a1, int lda1, int k1, const float* a2, int lda2, int K, int n,
                const float* bp, const float* bias, int N, float* out, float* ws, size_t ws_bytes,
                hipStream_t st, const void* x3 = nullptr, int relu = 1) {
  GemmParams p{};
  p.a1 = a1; p.lda1 = lda1; p.k1 = k1; p.a2 = a2; p.lda2 = lda2;
  p.M = n; p.N = N; p.K = K;
  p.bp = bp; p.bias = bias; p.rows_per_bias = 0;
  p.out = out; p.ldc = N; p.relu = relu;
  // measured in the step (profiles/r01k_infer_step_trace.txt): at a 2048-point batch the 64..512-deep
  // layers are launch-latency bound and the f32-input kernel's stream-K plan is faster (14.6 vs 23 us);
  // the three-term kernel wins from ~8k rows on, and on the 1984-deep layer always (42 vs 54 us)
  if (x3 && x3_enabled() && (n >= 8192 || K >= 1024)) {
    DISN_TRY(gemm_bf16_launch(p, GEMM_DENSE, x3, ws, ws ? ws_bytes : 0, st, 3));
    return 0;
  }
  const GemmPlan pl = gemm_plan(n, N, K, ws ? ws_bytes : 0);
  DISN_TRY(gemm_launch(p, GEMM_DENSE, pl, ws, st));
  return 0;
}

// The MLPs in three dependency phases (the phases of one point set may run on different streams):
//   phase 0 -- needs only the points: fold1 of both streams  (models/sdfnet.py:71-76,173-178)
//   phase 1 -- needs the feature map: local fold2/conv1 on [point512 | feat1472], fold2/conv2 (:180-184)
//   phase 2 -- needs the embedding: global fold2/conv1 with the per-image folded bias, fold2/conv2,
//              both fold2/conv5 and the sum  (:78-88, :186; models/model_normalization.py:204)
// fold1/conv2, conv3 of one stream (after pt_embed); `gws`: the GEMM scratch this stream may use
int mlp_fold1_local(const disn_mlp_weights_t* w, int n, const MlpWs& s, float* gws, hipStream_t st) {
  int rc;
  if ((rc = dense_layer(s.e1l, 64, 64, nullptr, 0, 64, n, w->l_w2, w->l_b2, 256, s.h256, gws, s.gemm_ws_bytes, st, w->l_x2))) return rc;
  return dense_layer(s.h256, 256, 256, nullptr, 0, 256, n, w->l_w3, w->l_b3, 512, s.h512a, gws, s.gemm_ws_bytes, st, w->l_x3);
}
int mlp_fold1_global(const disn_mlp_weights_t* w, int n, const MlpWs& s, float* gws, hipStream_t st) {
  int rc;
  if ((rc = dense_layer(s.e1g, 64, 64, nullptr, 0, 64, n, w->g_w2, w->g_b2, 256, s.g256, gws, s.gemm_ws_bytes, st, w->g_x2))) return rc;
  return dense_layer(s.g256, 256, 256, nullptr, 0, 256, n, w->g_w3, w->g_b3, 512, s.g512, gws, s.gemm_ws_bytes, st, w->g_x3);
}
int mlp_phase0(const disn_mlp_weights_t* w, const float* pts_rot, int n, const MlpWs& s,
               hipStream_t st) {
  int rc;
  DISN_TRY(pt_embed_launch(pts_rot, n, w->g_w1, w->g_b1, w->l_w1, w->l_b1, s.e1g, s.e1l, st));
  if ((rc = mlp_fold1_local(w, n, s, s.gemm_ws, st))) return rc;
  return mlp_fold1_global(w, n, s, s.gemm_ws, st);
}

int mlp_phase1(const disn_mlp_weights_t* w, int n, const float* feat, const MlpWs& s,
               hipStream_t st) {
  int rc;
  if ((rc = dense_layer(s.h512a, 512, 512, feat, DISN_FEAT_DIM, 512 + DISN_FEAT_DIM, n, w->l_w4, w->l_b4, 512, s.h512b, s.gemm_ws, s.gemm_ws_bytes, st, w->l_x4))) return rc;
  if ((rc = dense_layer(s.h512b, 512, 512, nullptr, 0, 512, n, w->l_w5, w->l_b5, 256, s.l5, s.gemm_ws, s.gemm_ws_bytes, st, w->l_x5))) return rc;
  return 0;
}

// B images x N points (rows image-major); gbias [B][512]
int mlp_phase2(const disn_mlp_weights_t* w, int B, int N, const float* gbias, float* sdf,
               float* sdf_g, float* sdf_l, float out_div, const MlpWs& s, hipStream_t st) {
  int rc;
  for (int b = 0; b < B; ++b) {
    const size_t o = (size_t)b * N;
    if ((rc = dense_layer(s.g512 + o * 512, 512, 512, nullptr, 0, 512, N, w->g_w4_point,
                          gbias + (size_t)b * 512, 512, s.h512b + o * 512, s.gemm_ws,
                          s.gemm_ws_bytes, st, w->g_x4_point)))
      return rc;
  }
  const int n = B * N;
  if ((rc = dense_layer(s.h512b, 512, 512, nullptr, 0, 512, n, w->g_w5, w->g_b5, 256, s.g5, s.gemm_ws, s.gemm_ws_bytes, st, w->g_x5))) return rc;
  DISN_TRY(final_dot_launch(s.g5, s.l5, n, w->g_w6, w->g_b6, w->l_w6, w->l_b6, sdf, sdf_g, sdf_l,
                            out_div, st));
  return 0;
}

// global fold2/conv1 in two halves.  g4_pre: the 512-deep product on the point features, no bias, no
// ReLU -- needs only phase 0.  phase2_split: + the per-image folded bias, ReLU (the same fp32 add the
// fused epilogue does), fold2/conv2 on s.gemm_ws2, then -- behind `joined` -- both conv5 and the sum.
int mlp_g4_pre(const disn_mlp_weights_t* w, int n, const MlpWs& s, float* gws, hipStream_t st, bool zeroed = false) {
  if (!zeroed) DISN_TRY(hipMemsetAsync(s.zero512, 0, 512 * sizeof(float), st));  // on this stream: no cross-stream order
  return dense_layer(s.g512, 512, 512, nullptr, 0, 512, n, w->g_w4_point, s.zero512, 512, s.g4pre,
                     gws, s.gemm_ws_bytes, st, w->g_x4_point, 0);
}

int mlp_phase2_split(const disn_mlp_weights_t* w, int B, int N, const float* gbias, float* sdf,
                     const MlpWs& s, hipStream_t st, hipEvent_t joined) {
  int rc;
  const int n = B * N;
  DISN_TRY(splitk_reduce_launch(s.g4pre, 1, n, 512, gbias, N, 1, s.g512, 512, st));
  if ((rc = dense_layer(s.g512, 512, 512, nullptr, 0, 512, n, w->g_w5, w->g_b5, 256, s.g5, s.gemm_ws2, s.gemm_ws_bytes, st, w->g_x5))) return rc;
  DISN_TRY(hipStreamWaitEvent(st, joined, 0));
  DISN_TRY(final_dot_launch(s.g5, s.l5, n, w->g_w6, w->g_b6, w->l_w6, w->l_b6, sdf, nullptr, nullptr,
                            1.0f, st));
  return 0;
}

// phase 1 with the folded local fold2/conv1 (disn_fold_local): 512-deep product on the point
// features, then + resampled pmap rows + bias, ReLU in one gather pass; no [n,1472] feature rows
int mlp_phase1_folded(const disn_mlp_weights_t* w, int n, const float* pmap_b, const float* trans_mat_b,
                      const float* pts, const MlpWs& s, hipStream_t st) {
  int rc;
  if ((rc = dense_layer(s.h512a, 512, 512, nullptr, 0, 512, n, w->l_w4_point, s.zero512, 512, s.h512b,
                        s.gemm_ws, s.gemm_ws_bytes, st, w->l_x4_point, 0)))
    return rc;
  DISN_TRY(gather_fold_launch(pmap_b, trans_mat_b, pts, n, s.h512b, w->l_b4, s.h512b, st));
  return dense_layer(s.h512b, 512, 512, nullptr, 0, 512, n, w->l_w5, w->l_b5, 256, s.l5, s.gemm_ws,
                     s.gemm_ws_bytes, st, w->l_x5);
}

int mlp_chunk_folded(const disn_mlp_weights_t* w, const float* pts, const float* pts_rot, int n,
                     const float* gbias, const float* pmap_b, const float* trans_mat_b, float* sdf,
                     float out_div, const MlpWs& s, hipStream_t st) {
  int rc;
  if ((rc = mlp_phase0(w, pts_rot, n, s, st))) return rc;
  if ((rc = mlp_phase1_folded(w, n, pmap_b, trans_mat_b, pts, s, st))) return rc;
  return mlp_phase2(w, 1, n, gbias, sdf, nullptr, nullptr, out_div, s, st);
}

// both MLP streams for n points of ONE image on one stream (gbias = that image's folded bias row)
int mlp_chunk(const disn_mlp_weights_t* w, const float* pts_rot, int n, const float* gbias,
              const float* feat, float* sdf, float* sdf_g, float* sdf_l, float out_div,
              const MlpWs& s, hipStream_t st) {
  int rc;
  if (mlp_h2(w, n)) {  // the same launches as disn_encode_query, on one stream (bit-identical results)
    if ((rc = mlp_fold1_h2(w, pts_rot, n, s, 0, 0, 1, st))) return rc;
    if ((rc = mlp_phase1_h2(w, n, feat, DISN_FEAT_DIM, s, 0, 0, 1, st))) return rc;
    if ((rc = mlp_g4_pre_h2(w, n, s, 0, 0, 1, s.h512a, st))) return rc;   // h512a is free once the local fold2/conv1 ran
    if ((rc = mlp_g5_h2(w, n, s.h512a, gbias, s, 0, 0, 1, st))) return rc;
    DISN_TRY(final_dot_launch(s.g5, s.l5, n, w->g_w6, w->g_b6, w->l_w6, w->l_b6, sdf, sdf_g, sdf_l, out_div, st));
    return 0;
  }
  if ((rc = mlp_phase0(w, pts_rot, n, s, st))) return rc;
  if ((rc = mlp_phase1(w, n, feat, s, st))) return rc;
  return mlp_phase2(w, 1, n, gbias, sdf, sdf_g, sdf_l, out_div, s, st);
}

const int kMapPixels = DISN_IMG_H * DISN_IMG_W;

const int kFeatPad = 1536;   // gathered feature rows zero-padded to a multiple of 256 columns (dense_h2 chunks)

// ---- the fused point MLP of a SMALL point set (round 4; models/sdfnet.py:71-90,173-186 + model_normalization.py:171-204)
// B images x N points (N % 128 == 0), no feature map and no folded map: the gather from the taps writes the 1472
// features in split form (one power-of-two scale per image from the taps' maxima), mlp_fused_kernel<local, FEAT> takes
// them as 96 extra reduction blocks of fold2/conv1, mlp_fused_kernel<global> runs with the image's folded bias row and
// adds the local sums: two launches per call behind the gather, every activation in registers.
bool fused_small_ok(const disn_mlp_weights_t* w, int B, int N) {
  return x3_enabled() && tune::fused_small != 0 && w->g_fused && w->l_feat && N % 128 == 0 && (long)B * N <= kChunk;
}
// tap_slots[k]: image 0's 64 activation-maximum slots of tap k (image b's are slot_stride floats further each);
// featmax: B floats of scratch; lsum: B * N floats of scratch
int fused_small_local(const disn_mlp_weights_t* w, float* const taps[5], const float* const tap_slots[5],
                      size_t slot_stride, const float* trans_mat, const float* pts, const float* pts_rot, int B, int N,
                      float* feat_split, float* featmax, float* lsum, hipStream_t st) {
  // (the one-wave-per-point gather takes the images' tap maxima from the slots itself and leaves them in featmax)
  const bool slots_in_gather = project_gather_taps_takes_slots(B, N, kFeatPad);
  if (!slots_in_gather) DISN_TRY(tap_amax_launch(tap_slots, slot_stride, B, featmax, st));
  DISN_TRY(project_gather_taps_launch(taps, trans_mat, pts, B, N, 0, 5, feat_split, st, kFeatPad, nullptr, 0, 0, featmax,
                                      slots_in_gather ? tap_slots : nullptr, slot_stride));
  DISN_TRY(mlp_fused_small_launch(true, w->l_feat, w->l_w1, w->l_b1, w->l_b2, w->l_b3, w->l_b4, w->l_b5, w->l_w6, w->l_b6,
                                  pts_rot, N, B, feat_split, kFeatPad, featmax, nullptr, lsum, 1.0f, st));
  return 0;
}
int fused_small_global(const disn_mlp_weights_t* w, const float* gbias, const float* pts_rot, int B, int N,
                       const float* lsum, float* sdf, float out_div, hipStream_t st) {
  DISN_TRY(mlp_fused_small_launch(false, w->g_fused, w->g_w1, w->g_b1, w->g_b2, w->g_b3, gbias, w->g_b5, w->g_w6, w->g_b6,
                                  pts_rot, N, B, nullptr, 0, nullptr, lsum, sdf, out_div, st));
  return 0;
}


struct QueryWs {
  float *gbias, *gemv_ws, *feat, *pts;
  MlpWs mlp;
  size_t total;
};

QueryWs query_layout(void* ws, int B, int chunk, bool need_feat, bool need_pts, bool split_g4 = false) {
  Bump b(ws);
  QueryWs q;
  q.gbias = b.take((size_t)B * 512 * sizeof(float));
  q.gemv_ws = b.take(gemv_ws_bytes(B, DISN_EMBED_DIM, 512));
  q.feat = need_feat ? b.take((size_t)chunk * kFeatPad * sizeof(float)) : nullptr;
  q.pts = need_pts ? b.take((size_t)chunk * 3 * sizeof(float)) : nullptr;
  q.mlp = mlp_layout(b, chunk, split_g4);
  q.total = (b.off + 255) & ~size_t(255);
  return q;
}

inline int chunk_for(long n) { return (int)(n < kChunk ? (n > 0 ? n : 1) : kChunk); }

bool grid_spec(const double* p, int R, GridSpec* g) {
  if (!p || R < 1) return false;
  g->res = R + 1;
  for (int a = 0; a < 3; ++a) {
    g->start[a] = p[a];
    g->stop[a] = p[a + 3];
    g->step[a] = (p[a + 3] - p[a]) / (double)R;  // numpy.linspace: delta / div
  }
  return true;
}

}  // namespace

extern "C" {

int disn_abi_version(void) { return DISN_ABI_VERSION; }

size_t disn_pack_kn_x3_bytes(int K, int N) {
  if (K <= 0 || N <= 0 || N % 32) return 0;
  return (size_t)3 * ((K + 31) & ~31) * N * 2;
}

int disn_pack_kn_x3(const float* w_kn, int K, int N, void* packed, void* stream) {
  if (!w_kn || !packed || K <= 0 || N <= 0) return DISN_E_ARG;
  if (N % 32) return DISN_E_SHAPE;
  DISN_TRY(pack_bf16_launch(w_kn, 0, K, N, packed, (hipStream_t)stream, 3));
  return 0;
}

int disn_pack_kn(const float* w_kn, int K, int N, int Kpad, float* packed, void* stream) {
  if (!w_kn || !packed || K <= 0 || N <= 0) return DISN_E_ARG;
  if (N % 32 || Kpad % 32 || Kpad < K) return DISN_E_SHAPE;
  DISN_TRY(pack_kn_launch(w_kn, K, N, Kpad, packed, (hipStream_t)stream));
  return 0;
}

int disn_resize_bilinear(const float* in, int B, int Hin, int Win, int C, float* out, int Hout,
                         int Wout, int out_cstride, int out_coff, void* stream) {
  if (!in || !out || B <= 0 || Hin <= 0 || Win <= 0 || C <= 0 || Hout <= 0 || Wout <= 0)
    return DISN_E_ARG;
  if (out_coff < 0 || out_coff + C > out_cstride) return DISN_E_SHAPE;
  DISN_TRY(resize_bilinear_launch(in, B, Hin, Win, C, out, Hout, Wout, out_cstride, out_coff,
                                  (hipStream_t)stream));
  return 0;
}

size_t disn_conv3x3_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
  return gemm_plan(B * H * W, Cout, conv_k(Cin)).ws_bytes;
}

int disn_conv3x3(const float* in, int B, int H, int W, int Cin, const float* w_packed,
                 const float* bias, int Cout, int relu, float* out, void* ws, size_t ws_bytes,
                 void* stream) {
  return conv3x3_impl(in, B, H, W, Cin, w_packed, bias, Cout, relu, out, (float*)ws, ws_bytes,
                      (hipStream_t)stream);
}

size_t disn_conv3x3_planned_workspace_bytes(int B, int H, int W, int Cin, int Cout, int bm, int bn, int wgs) {
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
  const int force[3] = {bm, bn, wgs};
  return gemm_plan(B * H * W, Cout, conv_k(Cin), ~size_t(0), force).ws_bytes;
}

int disn_conv3x3_planned(const float* in, int B, int H, int W, int Cin, const float* w_packed,
                         const float* bias, int Cout, int relu, float* out, void* ws, size_t ws_bytes,
                         int bm, int bn, int wgs, void* stream) {
  if (!in || !w_packed || !bias || !out || B <= 0 || H <= 0 || W <= 0) return DISN_E_ARG;
  if (!(Cin == 3 || Cin % 32 == 0) || Cout % 64 != 0 || H >= 32768 || W >= 32768) return DISN_E_SHAPE;
  if (!((bm == 64 || bm == 128) && (bn == 64 || bn == 128) && Cout % bn == 0 && (wgs >= 1 || wgs == -1)))
    return DISN_E_SHAPE;
  GemmParams p{};
  p.a1 = in;
  p.H = H; p.W = W; p.Cin = Cin;
  p.M = B * H * W; p.N = Cout; p.K = conv_k(Cin);
  p.bp = w_packed; p.bias = bias; p.rows_per_bias = 0;
  p.out = out; p.ldc = Cout; p.relu = relu;
  const int force[3] = {bm, bn, wgs};
  const GemmPlan pl = gemm_plan(p.M, p.N, p.K, ~size_t(0), force);
  if (pl.ws_bytes > (ws ? ws_bytes : 0)) return DISN_E_WS;
  DISN_TRY(gemm_launch(p, Cin == 3 ? GEMM_CONV3_C3 : GEMM_CONV3, pl, (float*)ws, (hipStream_t)stream));
  return 0;
}

int disn_maxpool2x2(const float* in, int B, int H, int W, int C, float* out, void* stream) {
  if (!in || !out || B <= 0 || H < 2 || W < 2) return DISN_E_ARG;
  if (C % 4) return DISN_E_SHAPE;
  DISN_TRY(maxpool2x2_launch(in, B, H, W, C, out, (hipStream_t)stream));
  return 0;
}

size_t disn_fc_workspace_bytes(int B, int K, int N) {
  if (B <= 0 || K <= 0 || N <= 0 || N % 256) return 0;
  return gemv_ws_bytes(B, K, N);
}

int disn_fc(const float* x, int B, int K, const float* w_kn, const float* bias, int N, int relu,
            float* out, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !w_kn || !bias || !out || !ws || B <= 0 || K <= 0) return DISN_E_ARG;
  if (N <= 0 || N % 256) return DISN_E_SHAPE;
  if (ws_bytes < gemv_ws_bytes(B, K, N)) return DISN_E_WS;
  DISN_TRY(gemv_launch(x, B, K, w_kn, bias, N, relu, out, (float*)ws, (hipStream_t)stream));
  return 0;
}

int disn_fc_t(const float* x, int B, int K, const float* wt_nk, const float* bias, int N, int relu, float* out,
              void* stream) {
  if (!x || !wt_nk || !bias || !out || B <= 0 || K <= 0 || N <= 0) return DISN_E_ARG;
  if (K % 4) return DISN_E_SHAPE;
  DISN_TRY(gemv_rows_launch(x, B, K, wt_nk, bias, N, relu, out, (hipStream_t)stream));
  return 0;
}

int disn_scale_channels(const float* in, int64_t rows, int C, const float* scale, int invert, float* out,
                        void* stream) {
  if (!in || !scale || !out || rows < 0 || C <= 0) return DISN_E_ARG;
  if (C % 4) return DISN_E_SHAPE;
  if (rows == 0) return 0;
  DISN_TRY(scale_channels_launch(in, rows, C, scale, invert, out, (hipStream_t)stream));
  return 0;
}

int disn_get_loss(const float* pred, const float* gt, int64_t M, float sdf_weight, float mask_weight, float* out5,
                  void* stream) {
  if (!pred || !gt || !out5 || M <= 0 || sdf_weight == 0.0f) return DISN_E_ARG;
  DISN_TRY(loss_reduce_launch(pred, gt, (long)M, sdf_weight, mask_weight, out5, (hipStream_t)stream));
  return 0;
}

size_t disn_dense_workspace_bytes(int M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0 || K % 32 || N % 64) return 0;
  return gemm_plan(M, N, K).ws_bytes;
}

int disn_dense(const float* a1, int lda1, int k1, const float* a2, int lda2, int k2, int M,
               const float* w_packed, const float* bias, int N, int relu, float* out, void* ws,
               size_t ws_bytes, void* stream) {
  if (!a1 || !w_packed || !bias || !out || M <= 0 || k1 <= 0 || k2 < 0 || (k2 > 0 && !a2))
    return DISN_E_ARG;
  if (k1 % 32 || k2 % 32 || N <= 0 || N % 64 || lda1 < k1 || (k2 > 0 && lda2 < k2) || lda1 % 4 ||
      (k2 > 0 && lda2 % 4))
    return DISN_E_SHAPE;
  GemmParams p{};
  p.a1 = a1; p.lda1 = lda1; p.k1 = k1; p.a2 = a2; p.lda2 = lda2;
  p.M = M; p.N = N; p.K = k1 + k2;
  p.bp = w_packed; p.bias = bias; p.rows_per_bias = 0;
  p.out = out; p.ldc = N; p.relu = relu;
  const GemmPlan pl = gemm_plan(M, N, p.K, ws ? ws_bytes : 0);
  DISN_TRY(gemm_launch(p, GEMM_DENSE, pl, (float*)ws, (hipStream_t)stream));
  return 0;
}

size_t disn_vgg16_workspace_bytes(int B) {
  if (B <= 0) return 0;
  return vgg_layout(nullptr, B, DISN_EMBED_DIM).total;
}

}  // extern "C"

// struct disn_ctx (kernels.hpp) as used here -- ev 0: fork, 1..4: grid pipeline buffers, 6: aux done,
// 7: features done, 8: g4_pre done

namespace {

const int kTapHw[5] = {224, 112, 56, 28, 14}, kTapCh[5] = {64, 128, 256, 512, 512};
const int kTapOff[5] = {0, 64, 192, 448, 960};

// disn_encode_query runs on two streams (a tuning build can run the same launches on one).
// Measured on MI355X (tools/overlap_sweep.py, cfg2 step of build r01c): single stream 0.886 ms, MLP
// under the fc head 0.797 ms.  Two other overlaps were tried and removed: the 110 MB tap up-samples
// on the auxiliary stream under the convolutions (0.86-0.98 ms at every throttle: their streamed
// writes disturb the latency-sensitive convolution loads) and a trickle read of the fc6 weights
// into the memory-side cache under conv4/conv5 (no gain up to 200 MB, slower beyond).
bool two_streams() { return tune::overlap != 0; }

bool vgg_weights_ok(const disn_vgg_weights_t* w) {
  if (!w) return false;
  for (int i = 0; i < 13; ++i)
    if (!w->conv_w[i] || !w->conv_b[i]) return false;
  for (int i = 0; i < 3; ++i)
    if (!w->fc_w[i] || !w->fc_b[i]) return false;
  return true;
}

// rows A, B (+E when featmap != nullptr): resize, conv stack, pools, all on `st`.  Returns pool5.
// layers [i0, i1) of the stack; i0 == 0 starts with the resize; *xio carries the current activation between calls
int vgg_features(const disn_vgg_weights_t* w, const float* img, int B, float* resized,
                 float* const taps[5], float* featmap, const VggWs& s, const float** xio,
                 hipStream_t st, int i0 = 0, int i1 = 13) {
  // the single-image kernels (conv_h2.hip) when every layer has its image: each layer's epilogue leaves the
  // maximum of its output in the slots the next layer scales its f16 split by (cleared by the resize launch)
  bool h2 = x3_enabled();
  for (int i = 0; i < 13; ++i) h2 = h2 && w->conv_w_h2[i] != nullptr;
  if (i0 == 0)
    DISN_TRY(resize_bilinear_launch(img, B, DISN_IMG_H, DISN_IMG_W, 3, resized, DISN_VGG_SIZE,
                                    DISN_VGG_SIZE, 3, 0, st, 0, h2 ? s.amax : nullptr, h2 ? 14 * B * 64 : 0));
  const float* x = i0 == 0 ? resized : *xio;
  bool toggle = false;  // (a pool precedes every layer that uses the bufA / bufB toggle first: restarts agree)
  const size_t gws_cap = (size_t)((char*)s.fc_ws - (char*)s.gemm_ws);
  for (int i = 0; i < i1; ++i) {
    const VggLayer& L = kVgg[i];
    float* out = L.tap >= 0 ? taps[L.tap] : (L.hw >= 112 ? s.bufA : (toggle ? s.bufB : s.bufA));
    if (L.tap < 0 && L.hw < 112) toggle = !toggle;
    if (kPoolAfter[i]) toggle = false;
    if (i < i0) continue;  // replay of the buffer choice only
    // a layer the pool follows: when its split-K reduce runs anyway, that pass also emits the pool
    bool pooled = false;
    int rc = 0;
    if (h2 && i == 0) {
      DISN_TRY(conv1_1_direct_launch(x, B, L.hw, L.hw, static_cast<const float*>(w->conv_w_h2[0]), w->conv_b[0], 1, out,
                                     s.amax + (size_t)B * 64, st, 64));
    } else if (h2) {
      DISN_TRY(conv_h2_launch(x, B, L.hw, L.hw, L.cin, w->conv_w_h2[i], w->conv_b[i], L.cout, 1,
                              s.amax + (size_t)B * 64 * i, out, kPoolAfter[i] ? s.bufP : nullptr,
                              s.amax + (size_t)B * 64 * (i + 1), st, w->strict_forms == 1 ? 11 : 0, 64));
      pooled = kPoolAfter[i];
    } else {
      rc = conv3x3_impl(x, B, L.hw, L.hw, L.cin, w->conv_w[i], w->conv_b[i], L.cout, 1, out, s.gemm_ws, gws_cap, st,
                        w->conv_w_x3[i], kPoolAfter[i] ? s.bufP : nullptr, &pooled);
    }
    if (rc) return rc;
    x = out;
    if (L.tap >= 0 && featmap)
      DISN_TRY(resize_bilinear_launch(taps[L.tap], B, kTapHw[L.tap], kTapHw[L.tap], kTapCh[L.tap],
                                      featmap, DISN_IMG_H, DISN_IMG_W, DISN_FEAT_DIM,
                                      kTapOff[L.tap], st, 0));
    if (kPoolAfter[i]) {
      if (!pooled) DISN_TRY(maxpool2x2_launch(x, B, L.hw, L.hw, L.cout, s.bufP, st));
      x = s.bufP;
    }
  }
  *xio = x;
  return 0;
}

// row C: fc6 (7x7 VALID == dense over the NHWC-flattened pool5), fc7, fc8
// (models/CNN/vgg.py:198-214; dropout inactive: is_training=False, model_normalization.py:76)
int fc_layer(const float* x, int B, int K, const float* w_kn, const float* wt_nk, const float* bias, int N, int relu,
             float* out, float* ws, hipStream_t st, bool single_form = false) {
  // the one-launch row form re-reads x (B x K floats) once per wave: right for one to three rows (launch-latency bound
  // layers), 4x the weight bytes in L2 traffic at eight -- a batched call takes the split-K stream kernel for every layer
  // single_form ("strict"): the forms of a call of one row for any B -- a row's sum is the same instruction sequence
  if (wt_nk && (B < tune::conv_wide_min || single_form)) DISN_TRY(gemv_rows_launch(x, B, K, wt_nk, bias, N, relu, out, st));   // (one threshold for every form switch: ADVICE r3)
  else DISN_TRY(gemv_launch(x, B, K, w_kn, bias, N, relu, out, ws, st, single_form));
  return 0;
}

int vgg_head(const disn_vgg_weights_t* w, const float* pool5, int B, float* embedding,
             const VggWs& s, hipStream_t st) {
  int rc;
  const bool sf = w->strict_forms == 1;   // "strict": every row as in a call of one image
  // fc6 (411 MB) stays on the split-K stream kernel: 6.2 TB/s there against 2.8 for the row form (r02i); the
  // 67 / 17 / 2 MB layers are launch-latency bound and take the one-launch row form
  if ((rc = fc_layer(pool5, B, 25088, w->fc_w[0], nullptr, w->fc_b[0], 4096, 1, s.fc6, s.fc_ws, st, sf))) return rc;
  if ((rc = fc_layer(s.fc6, B, 4096, w->fc_w[1], w->fc_w_t[1], w->fc_b[1], 4096, 1, s.fc7, s.fc_ws, st, sf))) return rc;
  return fc_layer(s.fc7, B, 4096, w->fc_w[2], w->fc_w_t[2], w->fc_b[2], w->num_classes, 0, embedding, s.fc_ws, st, sf);
}

// the per-image folded bias of the global fold2/conv1: gbias[b] = embedding[b] . W4_global + b4
int gbias_layer(const disn_mlp_weights_t* w, const float* embedding, int B, float* gbias, float* ws, hipStream_t st,
                bool single_form = false) {
  return fc_layer(embedding, B, DISN_EMBED_DIM, w->g_w4_global, w->g_w4_global_t, w->g_b4, 512, 0, gbias, ws, st, single_form);
}

// Point sets of a fused-small call are padded to a multiple of 128 points per image INSIDE the library (round 5; pad
// points (0, 0, 0) as test/create_sdf.py:241,256 pads its last split, their results discarded): whether a request runs
// the fused kernels no longer depends on N % 128.  The fused kernels scale per POINT and the feature scale
// of an image comes from its taps, so the real points' bits do not depend on the pad points.
inline int pad128(int N) { return (N + 127) & ~127; }

struct EncQueryWs {
  VggWs vgg;
  QueryWs q;
  float *pad_pts, *pad_rot, *pad_sdf;   // [B][pad128(N)][3] x 2, [B][pad128(N)]
  size_t total;
};

EncQueryWs encq_layout(void* ws, int B, int N, int num_classes) {
  EncQueryWs e;
  const int Np = (long)B * pad128(N) <= kChunk ? pad128(N) : N;
  e.vgg = vgg_layout(ws, B, num_classes);
  char* base = ws ? static_cast<char*>(ws) + e.vgg.total : nullptr;
  e.q = query_layout(base, B, B * Np, true, false, true);
  Bump b(base ? base + e.q.total : nullptr);
  e.pad_pts = b.take((size_t)B * Np * 3 * sizeof(float));
  e.pad_rot = b.take((size_t)B * Np * 3 * sizeof(float));
  e.pad_sdf = b.take((size_t)B * Np * sizeof(float));
  e.total = e.vgg.total + e.q.total + ((b.off + 255) & ~size_t(255));
  return e;
}

}  // namespace

extern "C" {

// ROCm binds a stream to one of its hardware queues when the stream first submits work: an empty launch (and a
// wait for it) right after creation makes that binding happen HERE, in creation order, instead of in whatever order
// several host threads happen to issue their first real launches
__global__ void disn_bind_queue_kernel() {}
static hipError_t bind_queue(hipStream_t st) {
  hipLaunchKernelGGL(disn_bind_queue_kernel, dim3(1), dim3(64), 0, st);
  const hipError_t e = hipGetLastError();
  return e != hipSuccess ? e : hipStreamSynchronize(st);
}

int disn_stream_create(void** stream) {
  if (!stream) return DISN_E_ARG;
  hipStream_t st = nullptr;
  DISN_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  DISN_TRY(bind_queue(st));
  *stream = st;
  return 0;
}

int disn_stream_destroy(void* stream) {
  if (!stream) return DISN_E_ARG;
  DISN_TRY(hipStreamSynchronize((hipStream_t)stream));
  DISN_TRY(hipStreamDestroy((hipStream_t)stream));
  return 0;
}

int disn_ctx_create(disn_ctx_t** out) {
  if (!out) return DISN_E_ARG;
  disn_ctx* c = new (std::nothrow) disn_ctx();
  if (!c) return DISN_E_ARG;
  hipError_t e;
#ifdef DISN_TUNING
  if (tune::aux_cu_mode > 0) {  // experiment: the auxiliary stream on a subset of the 256 CUs
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 256; ++i) {
      const int m = tune::aux_cu_mode;
      const bool on = m == 1 ? i < 192 : m == 2 ? (i & 3) != 3 : m == 3 ? i < 128 : m == 4 ? (i & 1) == 0
                    : m == 5 ? i >= 64 : (i & 7) < 6;
      if (on) mask[i >> 5] |= 1u << (i & 31);
    }
    e = hipExtStreamCreateWithCUMask(&c->aux, 8, mask);
  } else
#endif
  e = hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking);
  if (e == hipSuccess) e = bind_queue(c->aux);
  for (int i = 0; i < 10 && e == hipSuccess; ++i)
    e = hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming);
  if (e != hipSuccess) {
    delete c;
    return (int)e;
  }
  *out = c;
  return 0;
}

int disn_ctx_pipeline(disn_ctx_t* c, void* wait_event, void* record_event) {
  if (!c) return DISN_E_ARG;
  c->pipe_wait = (hipEvent_t)wait_event;
  c->pipe_record = (hipEvent_t)record_event;
  return 0;
}

int disn_ctx_destroy(disn_ctx_t* c) {
  if (!c) return DISN_E_ARG;
  (void)hipStreamSynchronize(c->aux);
  for (int i = 0; i < 10; ++i) (void)hipEventDestroy(c->ev[i]);
  (void)hipStreamDestroy(c->aux);
  delete c;
  return 0;
}

int disn_vgg16_forward(const disn_vgg_weights_t* w, const float* img, int B, float* resized224,
                       float* const taps[5], float* embedding, void* ws, size_t ws_bytes,
                       void* stream) {
  if (!vgg_weights_ok(w) || !img || !taps || !embedding || !ws || B <= 0) return DISN_E_ARG;
  for (int i = 0; i < 5; ++i)
    if (!taps[i]) return DISN_E_ARG;
  if (w->num_classes <= 0 || w->num_classes % 256) return DISN_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const VggWs s = vgg_layout(ws, B, w->num_classes);
  if (s.total > ws_bytes) return DISN_E_WS;
  const float* pool5 = nullptr;
  int rc = vgg_features(w, img, B, resized224 ? resized224 : s.resized, taps, nullptr, s,
                        &pool5, st);
  if (rc) return rc;
  return vgg_head(w, pool5, B, embedding, s, st);
}

int disn_vgg16_conv_stack(const disn_vgg_weights_t* w, const float* img, int B, float* resized224,
                          float* const taps[5], float* pool5, void* ws, size_t ws_bytes, void* stream) {
  if (!vgg_weights_ok(w) || !img || !taps || !ws || B <= 0) return DISN_E_ARG;
  for (int i = 0; i < 5; ++i)
    if (!taps[i]) return DISN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const VggWs s = vgg_layout(ws, B, w->num_classes > 0 ? w->num_classes : DISN_EMBED_DIM);
  if (s.total > ws_bytes) return DISN_E_WS;
  const float* p5 = nullptr;
  const int rc = vgg_features(w, img, B, resized224 ? resized224 : s.resized, taps, nullptr, s, &p5, st);
  if (rc) return rc;
  if (pool5) DISN_TRY(hipMemcpyAsync(pool5, p5, (size_t)B * 7 * 7 * 512 * sizeof(float), hipMemcpyDeviceToDevice, st));
  return 0;
}

size_t disn_encode_workspace_bytes(int B) { return disn_vgg16_workspace_bytes(B); }

int disn_encode(disn_ctx_t* ctx, const disn_vgg_weights_t* w, const float* img, int B,
                float* resized224, float* const taps[5], float* embedding, float* featmap, void* ws,
                size_t ws_bytes, void* stream) {
  if (!vgg_weights_ok(w) || !img || !taps || !embedding || !featmap || !ws || B <= 0)
    return DISN_E_ARG;
  for (int i = 0; i < 5; ++i)
    if (!taps[i]) return DISN_E_ARG;
  if (w->num_classes <= 0 || w->num_classes % 256) return DISN_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const VggWs s = vgg_layout(ws, B, w->num_classes);
  if (s.total > ws_bytes) return DISN_E_WS;
  (void)ctx;  // everything runs on `st`: nothing in the encoder alone is worth a second stream
  const float* pool5 = nullptr;
  int rc = vgg_features(w, img, B, resized224 ? resized224 : s.resized, taps, featmap, s, &pool5, st);
  if (rc) return rc;
  return vgg_head(w, pool5, B, embedding, s, st);
}

size_t disn_encode_query_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0 || (long)B * N > kChunk) return 0;
  return encq_layout(nullptr, B, N, DISN_EMBED_DIM).total;
}

int disn_encode_query(disn_ctx_t* ctx, const disn_vgg_weights_t* vw, const disn_mlp_weights_t* mw,
                      const float* img, const float* trans_mat, const float* pts,
                      const float* pts_rot, int B, int N, float* resized224, float* const taps[5],
                      float* embedding, float* featmap, float* sdf, void* ws, size_t ws_bytes,
                      void* stream) {
  if (!ctx || !vgg_weights_ok(vw) || !mlp_weights_ok(mw) || !img || !trans_mat || !pts || !pts_rot ||
      !taps || !embedding || !sdf || !ws || B <= 0 || N <= 0)
    return DISN_E_ARG;
  for (int i = 0; i < 5; ++i)
    if (!taps[i]) return DISN_E_ARG;
  if ((long)B * N > kChunk || vw->num_classes != DISN_EMBED_DIM) return DISN_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int N0 = N;
  const EncQueryWs e = encq_layout(ws, B, N, vw->num_classes);
  if (e.total > ws_bytes) return DISN_E_WS;
  // Two streams.  `st` (the caller's): resize, conv stack, fc6..fc8, the folded global bias, then the
  // short tail of the global MLP stream.  ctx->aux: everything of the MLPs that does not need the
  // embedding -- from the fork on the point-only layers of both streams and the point half of the
  // global fold2/conv1 (small GEMMs in the quantisation holes of the convolutions); behind conv5_3 the
  // gather and the local fold2 layers (MFMA bound), under the 495 MB fc weight stream (HBM bound).
  // Every event record / wait on `st` drains it (~6 us in the kernel trace): there are three.
  const bool two = two_streams();
  hipStream_t ms = two ? ctx->aux : st;
  int rc;
  // Schedule (r02i / r02j traces).  The convolution kernels from conv2 on are ONE round of 112..224 workgroups that
  // each fill a CU: a point-MLP GEMM running beside them takes CUs away and doubles a layer (18 -> 34 us), so
  // nothing runs beside conv2_1 .. conv5_3.  `st`: resize, the convolutions, the HBM-bound fc head, the short
  // global tail.  ctx->aux: from the fork, the point embedding, the GLOBAL stream's fold1 and the point half of its
  // fold2/conv1 -- ~45 us of small launches beside resize / conv1_1 / conv1_2 (VALU work and a two-round kernel,
  // measured unaffected) -- then, behind conv5_3, the LOCAL stream's fold1, the gather and fold2 under the fc head.
  const float* pool5 = nullptr;
  bool gather_on_st = false;
  if (ctx->pipe_wait) DISN_TRY(hipStreamWaitEvent(st, ctx->pipe_wait, 0));  // behind the previous step's convolutions
  // Round 4: a batched call (>= kConvWideMinImages images) with the fused small-set kernels -- nothing runs beside the
  // convolutions; behind conv5_3 the auxiliary stream gathers (split form) and runs the local stream's one launch
  // under the fc head; the global stream's one launch follows the folded bias and adds the local sums.
  bool conv_h2_all = x3_enabled();
  for (int i = 0; i < 13; ++i) conv_h2_all = conv_h2_all && vw->conv_w_h2[i] != nullptr;
  // (a call of one to three requests keeps the layer-by-layer dense_h2 form below 8192 points per request -- its bits
  // are a B = 1 call's; from 8192 points on it takes the fused kernels too: 2.13 -> 1.40 ms for one request of 65536 points
  // against the three-term GEMM chain this shape ran until round 4, tools/encode_query_forms_time.py)
  const int Np = pad128(N);
  // "strict" (disn_vgg_weights_t.strict_forms = 1) in a call of >= 4 requests: the single-image forms of the convolutions
  // (vgg_features) AND of the point-MLP layers (dense_h2.hip's four-k-wave tiles, per-image scales) -- the request's taps
  // and, up to the fc head's form, its pred_sdf are those of the request alone
  const bool strict = vw->strict_forms == 1 && B >= tune::conv_wide_min && mlp_h2(mw, N);
  if (!strict && two && conv_h2_all && !featmap && (B >= tune::conv_wide_min || N >= 8192) && fused_small_ok(mw, B, Np)) {
    float* sdf_out = sdf;
    if (Np != N) {   // pad the point sets (both streams run behind the fork / the convolutions anyway)
      DISN_TRY(restride_rows_launch(pts, B, N, e.pad_pts, Np, 3, st));
      if (pts_rot != pts) DISN_TRY(restride_rows_launch(pts_rot, B, N, e.pad_rot, Np, 3, st));
      pts_rot = pts_rot != pts ? e.pad_rot : e.pad_pts;
      pts = e.pad_pts;
      sdf = e.pad_sdf;
      N = Np;
    }
    rc = vgg_features(vw, img, B, resized224 ? resized224 : e.vgg.resized, taps, nullptr, e.vgg, &pool5, st);
    if (rc) return rc;
    DISN_TRY(hipEventRecord(ctx->ev[7], st));
    DISN_TRY(hipStreamWaitEvent(ctx->aux, ctx->ev[7], 0));
    static const int tap_layer[5] = {1, 3, 6, 9, 12};
    const float* slots[5];
    for (int k = 0; k < 5; ++k) slots[k] = e.vgg.amax + (size_t)B * 64 * (tap_layer[k] + 1);
    if ((rc = fused_small_local(mw, taps, slots, 64, trans_mat, pts, pts_rot, B, N, e.q.feat, e.q.mlp.amax, e.q.mlp.l5,
                                ctx->aux)))
      return rc;
    DISN_TRY(hipEventRecord(ctx->ev[6], ctx->aux));
    if (ctx->pipe_record) DISN_TRY(hipEventRecord(ctx->pipe_record, st));
    if ((rc = vgg_head(vw, pool5, B, embedding, e.vgg, st))) return rc;
    { const int grc = gbias_layer(mw, embedding, B, e.q.gbias, e.q.gemv_ws, st, vw->strict_forms == 1); if (grc) return grc; }
    DISN_TRY(hipStreamWaitEvent(st, ctx->ev[6], 0));
    if ((rc = fused_small_global(mw, e.q.gbias, pts_rot, B, N, e.q.mlp.l5, sdf, 1.0f, st))) return rc;
    if (sdf != sdf_out) DISN_TRY(restride_rows_launch(sdf, B, N, sdf_out, N0, 1, st));
    return 0;
  }
  const bool h2 = two && B <= kH2Imgs && mlp_h2(mw, N);   // small point sets: the dense_h2 layers, image by image
  const int feat_ld = h2 && !featmap ? kFeatPad : DISN_FEAT_DIM;
  const int hb = h2 && N % 64 == 0 ? B : 1;   // images per h2 launch
  if (two) {
    DISN_TRY(hipEventRecord(ctx->ev[0], st));  // fork (orders aux behind the caller's inputs)
    DISN_TRY(hipStreamWaitEvent(ctx->aux, ctx->ev[0], 0));
    // host order: the caller's stream gets resize, conv1_1, conv1_2 first (it must never wait for the host), then
    // the auxiliary stream its launches, then the rest of the stack
    rc = vgg_features(vw, img, B, resized224 ? resized224 : e.vgg.resized, taps, featmap, e.vgg, &pool5, st, 0, 2);
    if (rc) return rc;
    if (h2) {  // fold1 of both streams (paired launches) and the point half of the global fold2/conv1: one launch
               // per layer for the whole batch (hb = B images at once) or image by image (hb = 1)
      for (int b = 0; b < B; b += hb) {
        const size_t o = (size_t)b * N;
        if ((rc = mlp_fold1_h2(mw, pts_rot + o * 3, N, e.q.mlp, b, o, hb, ctx->aux, strict))) return rc;
        if ((rc = mlp_g4_pre_h2(mw, N, e.q.mlp, b, o, hb, e.q.mlp.g4pre + o * 512, ctx->aux, strict))) return rc;
      }
    } else {
      DISN_TRY(pt_embed_launch(pts_rot, B * N, mw->g_w1, mw->g_b1, mw->l_w1, mw->l_b1, e.q.mlp.e1g, e.q.mlp.e1l, ctx->aux));
      if ((rc = mlp_fold1_global(mw, B * N, e.q.mlp, e.q.mlp.gemm_ws2, ctx->aux))) return rc;
      if ((rc = mlp_g4_pre(mw, B * N, e.q.mlp, e.q.mlp.gemm_ws2, ctx->aux))) return rc;
    }
    DISN_TRY(hipEventRecord(ctx->ev[8], ctx->aux));
    rc = vgg_features(vw, img, B, resized224 ? resized224 : e.vgg.resized, taps, featmap, e.vgg, &pool5, st, 2, 13);
    if (rc) return rc;
    // the gather from the taps on the caller's stream, BEFORE the fc head: alone it takes 14 us, under fc6's HBM
    // stream 65-70 us (r02q trace) -- and the local fold2 layers behind it are the critical path of the tail
    // The kernel also leaves max |feat| per image in the slots the local fold2/conv1 reads its scale from (cleared
    // by pt_embed on the auxiliary stream, hence the ev[8] wait first -- recorded a whole convolution stack ago).
    gather_on_st = h2 && !featmap;
    DISN_TRY(hipStreamWaitEvent(st, ctx->ev[8], 0));   // (recorded behind g4_pre, a whole convolution stack ago)
    if (gather_on_st)
      DISN_TRY(project_gather_taps_launch(taps, trans_mat, pts, B, N, 0, 5, e.q.feat, st, feat_ld,
                                          h2_slots(e.q.mlp, 0) + kFeatMaxSlot, 1024));
    if (ctx->pipe_record) DISN_TRY(hipEventRecord(ctx->pipe_record, st));  // the next step's convolutions may start
    DISN_TRY(hipEventRecord(ctx->ev[7], st));
    DISN_TRY(hipStreamWaitEvent(ctx->aux, ctx->ev[7], 0));
    if (!h2 && (rc = mlp_fold1_local(mw, B * N, e.q.mlp, e.q.mlp.gemm_ws, ctx->aux))) return rc;
  } else {
    rc = vgg_features(vw, img, B, resized224 ? resized224 : e.vgg.resized, taps, featmap, e.vgg, &pool5, st);
    if (rc) return rc;
    if ((rc = mlp_phase0(mw, pts_rot, B * N, e.q.mlp, st))) return rc;
    if ((rc = vgg_head(vw, pool5, B, embedding, e.vgg, st))) return rc;
  }
  if (gather_on_st) {
    // done above
  } else if (featmap) {
    const size_t map_stride = (size_t)DISN_IMG_H * DISN_IMG_W * DISN_FEAT_DIM;
    for (int b = 0; b < B; ++b)
      DISN_TRY(project_gather_launch(featmap + b * map_stride, trans_mat + (size_t)b * 12,
                                     pts + (size_t)b * N * 3, N,
                                     e.q.feat + (size_t)b * N * DISN_FEAT_DIM, ms));
  } else {  // no map: up-sample the taps at the touched pixels (bit-identical), all images in one launch
    DISN_TRY(project_gather_taps_launch(taps, trans_mat, pts, B, N, 0, 5, e.q.feat, ms, feat_ld));
  }
  if (h2) {
    for (int b = 0; b < B; b += hb) {
      const size_t o = (size_t)b * N;
      if ((rc = mlp_phase1_h2(mw, N, e.q.feat + o * feat_ld, feat_ld, e.q.mlp, b, o, hb, ms, gather_on_st, strict))) return rc;
    }
  } else if ((rc = mlp_phase1(mw, B * N, e.q.feat, e.q.mlp, ms))) return rc;
  if (two) {
    DISN_TRY(hipEventRecord(ctx->ev[6], ctx->aux));
    if ((rc = vgg_head(vw, pool5, B, embedding, e.vgg, st))) return rc;
  }
  { const int grc = gbias_layer(mw, embedding, B, e.q.gbias, e.q.gemv_ws, st, vw->strict_forms == 1); if (grc) return grc; }
  if (h2) {  // global fold2/conv2 on relu(pre + bias) per image, then -- behind ev[6] -- both fold2/conv5 and the sum
    for (int b = 0; b < B; b += hb) {
      const size_t o = (size_t)b * N;
      if ((rc = mlp_g5_h2(mw, N, e.q.mlp.g4pre + o * 512, e.q.gbias + (size_t)b * 512, e.q.mlp, b, o, hb, st, strict))) return rc;
    }
    DISN_TRY(hipStreamWaitEvent(st, ctx->ev[6], 0));
    DISN_TRY(final_dot_launch(e.q.mlp.g5, e.q.mlp.l5, (int64_t)B * N, mw->g_w6, mw->g_b6, mw->l_w6, mw->l_b6, sdf, nullptr,
                              nullptr, 1.0f, st));
    return 0;
  }
  if (two)  // bias + ReLU of the split layer, fold2/conv2, then -- behind ev[6] -- the final sum
    return mlp_phase2_split(mw, B, N, e.q.gbias, sdf, e.q.mlp, st, ctx->ev[6]);
  return mlp_phase2(mw, B, N, e.q.gbias, sdf, nullptr, nullptr, 1.0f, e.q.mlp, st);
}

int disn_build_featmap(const float* const taps[5], int B, float* featmap, void* stream) {
  if (!taps || !featmap || B <= 0) return DISN_E_ARG;
  const int hw[5] = {224, 112, 56, 28, 14}, ch[5] = {64, 128, 256, 512, 512};
  int coff = 0;
  for (int i = 0; i < 5; ++i) {
    if (!taps[i]) return DISN_E_ARG;
    DISN_TRY(resize_bilinear_launch(taps[i], B, hw[i], hw[i], ch[i], featmap, DISN_IMG_H,
                                    DISN_IMG_W, DISN_FEAT_DIM, coff, (hipStream_t)stream));
    coff += ch[i];
  }
  return 0;
}

int disn_project(const float* pts, const float* trans_mat, int B, int N, float* xy, void* stream) {
  if (!pts || !trans_mat || !xy || B <= 0 || N <= 0) return DISN_E_ARG;
  DISN_TRY(project_launch(pts, trans_mat, B, N, xy, (hipStream_t)stream));
  return 0;
}

int disn_gather(const float* featmap, const float* xy, int B, int N, float* feat, void* stream) {
  if (!featmap || !xy || !feat || B <= 0 || N <= 0) return DISN_E_ARG;
  DISN_TRY(gather_launch(featmap, xy, B, N, feat, (hipStream_t)stream));
  return 0;
}

int disn_gather_taps(const float* const taps[5], const float* trans_mat, const float* pts, int B, int N,
                     float* feat, void* stream) {
  if (!taps || !trans_mat || !pts || !feat || B <= 0 || N <= 0) return DISN_E_ARG;
  for (int i = 0; i < 5; ++i)
    if (!taps[i]) return DISN_E_ARG;
  DISN_TRY(project_gather_taps_launch(taps, trans_mat, pts, B, N, 0, 5, feat, (hipStream_t)stream));
  return 0;
}

int disn_gather_taps_split(const float* const taps[5], const float* trans_mat, const float* pts, int B, int N,
                           const float* feat_amax, void* feat_split, void* stream) {
  if (!taps || !trans_mat || !pts || !feat_amax || !feat_split || B <= 0 || N <= 0) return DISN_E_ARG;
  for (int i = 0; i < 5; ++i)
    if (!taps[i]) return DISN_E_ARG;
  DISN_TRY(project_gather_taps_launch(taps, trans_mat, pts, B, N, 0, 5, static_cast<float*>(feat_split),
                                      (hipStream_t)stream, kFeatPad, nullptr, 0, 0, feat_amax));
  return 0;
}

int disn_gather_fold(const float* pmap_b, const float* trans_mat_b, const float* pts, int N, const float* pre,
                     const float* bias, float* h, void* stream) {
  if (!pmap_b || !trans_mat_b || !pts || !pre || !bias || !h || N <= 0) return DISN_E_ARG;
  DISN_TRY(gather_fold_launch(pmap_b, trans_mat_b, pts, N, pre, bias, h, (hipStream_t)stream));
  return 0;
}

size_t disn_sdf_mlp_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  return query_layout(nullptr, B, chunk_for(N), false, false).total;
}

int disn_sdf_mlp(const disn_mlp_weights_t* w, const float* pts_rot, const float* embedding,
                 const float* feat, int B, int N, float* sdf, float* sdf_global, float* sdf_local,
                 void* ws, size_t ws_bytes, void* stream) {
  if (!mlp_weights_ok(w) || !pts_rot || !embedding || !feat || !sdf || !ws || B <= 0 || N <= 0)
    return DISN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int chunk = chunk_for(N);
  const QueryWs q = query_layout(ws, B, chunk, false, false);
  if (q.total > ws_bytes) return DISN_E_WS;
  { const int grc = gbias_layer(w, embedding, B, q.gbias, q.gemv_ws, st); if (grc) return grc; }
  for (int b = 0; b < B; ++b)
    for (int n0 = 0; n0 < N; n0 += chunk) {
      const int n = (N - n0) < chunk ? (N - n0) : chunk;
      const size_t o = (size_t)b * N + n0;
      const int rc = mlp_chunk(w, pts_rot + o * 3, n, q.gbias + (size_t)b * 512,
                               feat + o * DISN_FEAT_DIM, sdf + o, sdf_global ? sdf_global + o : nullptr,
                               sdf_local ? sdf_local + o : nullptr, 1.0f, q.mlp, st);
      if (rc) return rc;
    }
  return 0;
}

size_t disn_query_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  return query_layout(nullptr, B, chunk_for(N), true, false).total;
}

int disn_query(const disn_mlp_weights_t* w, const float* featmap, const float* embedding,
               const float* trans_mat, const float* pts, const float* pts_rot, int B, int N,
               float* sdf, void* ws, size_t ws_bytes, void* stream) {
  if (!mlp_weights_ok(w) || !featmap || !embedding || !trans_mat || !pts || !pts_rot || !sdf ||
      !ws || B <= 0 || N <= 0)
    return DISN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int chunk = chunk_for(N);
  const QueryWs q = query_layout(ws, B, chunk, true, false);
  if (q.total > ws_bytes) return DISN_E_WS;
  { const int grc = gbias_layer(w, embedding, B, q.gbias, q.gemv_ws, st); if (grc) return grc; }
  const size_t map_stride = (size_t)DISN_IMG_H * DISN_IMG_W * DISN_FEAT_DIM;
  for (int b = 0; b < B; ++b)
    for (int n0 = 0; n0 < N; n0 += chunk) {
      const int n = (N - n0) < chunk ? (N - n0) : chunk;
      const size_t o = (size_t)b * N + n0;
      DISN_TRY(project_gather_launch(featmap + b * map_stride, trans_mat + (size_t)b * 12,
                                     pts + o * 3, n, q.feat, st));
      const int rc = mlp_chunk(w, pts_rot + o * 3, n, q.gbias + (size_t)b * 512, q.feat, sdf + o,
                               nullptr, nullptr, 1.0f, q.mlp, st);
      if (rc) return rc;
    }
  return 0;
}

// ---- folded local stream (include/disn_amd.h) -------------------------------------------------

static size_t fold_gemm_ws() {
  const size_t a = gemm_plan(kMapPixels, 512, DISN_FEAT_DIM).ws_bytes;
  const size_t b = gemm_bf16_ws_bytes(kMapPixels, 512, DISN_FEAT_DIM);
  return ((a > b ? a : b) + 255) & ~size_t(255);
}

size_t disn_fold_local_workspace_bytes(void) { return 2048 + fold_gemm_ws(); }

int disn_fold_local(const disn_mlp_weights_t* w, const float* featmap_b, float* pmap, void* ws,
                    size_t ws_bytes, void* stream) {
  if (!mlp_weights_ok(w) || !w->l_w4_point || !w->l_w4_feat || !featmap_b || !pmap || !ws)
    return DISN_E_ARG;
  if (ws_bytes < disn_fold_local_workspace_bytes()) return DISN_E_WS;
  hipStream_t st = (hipStream_t)stream;
  float* zero = static_cast<float*>(ws);
  DISN_TRY(hipMemsetAsync(zero, 0, 512 * sizeof(float), st));
  return dense_layer(featmap_b, DISN_FEAT_DIM, DISN_FEAT_DIM, nullptr, 0, DISN_FEAT_DIM, kMapPixels,
                     w->l_w4_feat, zero, 512, pmap, reinterpret_cast<float*>(static_cast<char*>(ws) + 2048),
                     ws_bytes - 2048, st, w->l_x4_feat, 0);
}

int disn_query_folded(const disn_mlp_weights_t* w, const float* pmap, const float* embedding,
                      const float* trans_mat, const float* pts, const float* pts_rot, int B, int N,
                      float* sdf, void* ws, size_t ws_bytes, void* stream) {
  if (!mlp_weights_ok(w) || !w->l_w4_point || !pmap || !embedding || !trans_mat || !pts || !pts_rot ||
      !sdf || !ws || B <= 0 || N <= 0)
    return DISN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int chunk = chunk_for(N);
  const QueryWs q = query_layout(ws, B, chunk, true, false);
  if (q.total > ws_bytes) return DISN_E_WS;
  DISN_TRY(hipMemsetAsync(q.mlp.zero512, 0, 512 * sizeof(float), st));
  { const int grc = gbias_layer(w, embedding, B, q.gbias, q.gemv_ws, st); if (grc) return grc; }
  for (int b = 0; b < B; ++b)
    for (int n0 = 0; n0 < N; n0 += chunk) {
      const int n = (N - n0) < chunk ? (N - n0) : chunk;
      const size_t o = (size_t)b * N + n0;
      const int rc = mlp_chunk_folded(w, pts + o * 3, pts_rot + o * 3, n, q.gbias + (size_t)b * 512,
                                      pmap + (size_t)b * kMapPixels * 512, trans_mat + (size_t)b * 12,
                                      sdf + o, 1.0f, q.mlp, st);
      if (rc) return rc;
    }
  return 0;
}

int disn_query_grid_folded(const disn_mlp_weights_t* w, const float* pmap, const float* embedding,
                           const float* trans_mat, const double* sdf_params_host, int R,
                           int64_t k0, int64_t k1, float sdf_weight, float* out, void* ws,
                           size_t ws_bytes, void* stream) {
  GridSpec g;
  if (!mlp_weights_ok(w) || !w->l_w4_point || !pmap || !embedding || !trans_mat || !out || !ws ||
      !grid_spec(sdf_params_host, R, &g))
    return DISN_E_ARG;
  const int64_t total = (int64_t)g.res * g.res * g.res;
  if (k0 < 0 || k1 > total || k0 >= k1 || sdf_weight == 0.0f) return DISN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int chunk = chunk_for(k1 - k0);
  const QueryWs q = query_layout(ws, 1, chunk, true, true);
  if (q.total > ws_bytes) return DISN_E_WS;
  DISN_TRY(hipMemsetAsync(q.mlp.zero512, 0, 512 * sizeof(float), st));
  { const int grc = gbias_layer(w, embedding, 1, q.gbias, q.gemv_ws, st); if (grc) return grc; }
  for (int64_t k = k0; k < k1; k += chunk) {
    const int n = (int)((k1 - k) < chunk ? (k1 - k) : chunk);
    DISN_TRY(grid_points_launch(g, k, k + n, q.pts, st));
    const int rc = mlp_chunk_folded(w, q.pts, q.pts, n, q.gbias, pmap, trans_mat, out + (k - k0),
                                    sdf_weight, q.mlp, st);
    if (rc) return rc;
  }
  return 0;
}

// ---- fused point MLP (mlp_fused.hip) -----------------------------------------------------------------
size_t disn_mlp_fused_image_bytes(void) { return mlp_fused_image_bytes(); }

int disn_mlp_fused_pack(const float* w2, const float* w3, const float* w4_point, const float* w5, void* image,
                        void* stream) {
  if (!w2 || !w3 || !w4_point || !w5 || !image) return DISN_E_ARG;
  DISN_TRY(mlp_fused_pack_launch(w2, w3, w4_point, w5, image, (hipStream_t)stream));
  return 0;
}

size_t disn_mlp_fused_feat_image_bytes(void) { return mlp_fused_feat_image_bytes(); }

int disn_mlp_fused_feat_pack(const float* w2, const float* w3, const float* w4, const float* w5, void* image,
                             void* stream) {
  if (!w2 || !w3 || !w4 || !w5 || !image) return DISN_E_ARG;
  DISN_TRY(mlp_fused_feat_pack_launch(w2, w3, w4, w5, image, (hipStream_t)stream));
  return 0;
}

namespace {
struct TapsFusedWs {
  float *gbias, *gemv_ws, *feat, *slots, *featmax, *lsum, *pad_pts, *pad_rot, *pad_sdf;
  size_t total;
};
TapsFusedWs taps_fused_layout(void* ws, int B, int N) {
  Bump b(ws);
  TapsFusedWs f;
  f.gbias = b.take((size_t)B * 512 * sizeof(float));
  f.gemv_ws = b.take(gemv_ws_bytes(B, DISN_EMBED_DIM, 512));
  f.feat = b.take((size_t)B * N * kFeatPad * sizeof(float));
  f.slots = b.take((size_t)5 * B * 64 * sizeof(float));
  f.featmax = b.take((size_t)B * sizeof(float));
  f.lsum = b.take((size_t)B * N * sizeof(float));
  f.pad_pts = b.take((size_t)B * N * 3 * sizeof(float));
  f.pad_rot = b.take((size_t)B * N * 3 * sizeof(float));
  f.pad_sdf = b.take((size_t)B * N * sizeof(float));
  f.total = (b.off + 255) & ~size_t(255);
  return f;
}
}  // namespace

size_t disn_query_taps_fused_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0 || (long)B * pad128(N) > kChunk) return 0;
  return taps_fused_layout(nullptr, B, pad128(N)).total;
}

int disn_query_taps_fused(const disn_mlp_weights_t* w, const float* const taps[5], const float* embedding,
                          const float* trans_mat, const float* pts, const float* pts_rot, int B, int N, float* sdf,
                          void* ws, size_t ws_bytes, void* stream) {
  if (!mlp_weights_ok(w) || !w->g_fused || !w->l_feat || !taps || !embedding || !trans_mat || !pts || !pts_rot || !sdf ||
      !ws || B <= 0 || N <= 0)
    return DISN_E_ARG;
  for (int i = 0; i < 5; ++i)
    if (!taps[i]) return DISN_E_ARG;
  const int N0 = N, Np = pad128(N);
  if ((long)B * Np > kChunk) return DISN_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const TapsFusedWs f = taps_fused_layout(ws, B, Np);
  if (f.total > ws_bytes) return DISN_E_WS;
  float* sdf_out = sdf;
  if (Np != N) {   // pad points (0, 0, 0) in, the first N results out (see pad128)
    DISN_TRY(restride_rows_launch(pts, B, N, f.pad_pts, Np, 3, st));
    if (pts_rot != pts) DISN_TRY(restride_rows_launch(pts_rot, B, N, f.pad_rot, Np, 3, st));
    pts_rot = pts_rot != pts ? f.pad_rot : f.pad_pts;
    pts = f.pad_pts;
    sdf = f.pad_sdf;
    N = Np;
  }
  // the taps' exact maxima, per image (inside disn_encode_query they come out of the convolutions' epilogues: the
  // same numbers, hence the same split scale and the same bits)
  static const int hw[5] = {224, 112, 56, 28, 14}, ch[5] = {64, 128, 256, 512, 512};
  DISN_TRY(hipMemsetAsync(f.slots, 0, (size_t)5 * B * 64 * sizeof(float), st));
  const float* slots[5];
  for (int k = 0; k < 5; ++k) {
    slots[k] = f.slots + (size_t)k * B * 64;
    DISN_TRY(amax64_accumulate_launch(taps[k], (size_t)hw[k] * hw[k] * ch[k], f.slots + (size_t)k * B * 64, st, B, 64));
  }
  float* tp[5];
  for (int k = 0; k < 5; ++k) tp[k] = const_cast<float*>(taps[k]);
  int rc = fused_small_local(w, tp, slots, 64, trans_mat, pts, pts_rot, B, N, f.feat, f.featmax, f.lsum, st);
  if (rc) return rc;
  { const int grc = gbias_layer(w, embedding, B, f.gbias, f.gemv_ws, st); if (grc) return grc; }
  if ((rc = fused_small_global(w, f.gbias, pts_rot, B, N, f.lsum, sdf, 1.0f, st))) return rc;
  if (sdf != sdf_out) DISN_TRY(restride_rows_launch(sdf, B, N, sdf_out, N0, 1, st));
  return 0;
}

int disn_amax(const float* x, int64_t n, float* out, void* stream) {
  if (!x || !out || n <= 0) return DISN_E_ARG;
  if (n % 4) return DISN_E_SHAPE;
  DISN_TRY(amax_launch(x, (size_t)n, out, (hipStream_t)stream));
  return 0;
}

namespace {
struct FusedWs {
  float *gbias, *gemv_ws, *gsum;
  size_t total;
};
FusedWs fused_layout(void* ws, int B, int64_t n_max) {
  Bump b(ws);
  FusedWs f;
  f.gbias = b.take((size_t)B * 512 * sizeof(float));
  f.gemv_ws = b.take(gemv_ws_bytes(B, DISN_EMBED_DIM, 512));
  f.gsum = b.take((size_t)n_max * sizeof(float));
  f.total = (b.off + 255) & ~size_t(255);
  return f;
}
bool fused_ok(const disn_mlp_weights_t* w) { return mlp_weights_ok(w) && w->g_fused && w->l_fused; }

// both streams for n points of one image: the global stream's sums go through ws, the local kernel adds them
int fused_streams(const disn_mlp_weights_t* w, const float* gbias_b, const float* pmap_b,
                  const float* pmap_amax_b, const float* trans_mat_b, const float* pts, const float* pts_rot,
                  const GridSpec* grid, long long k0, long long n, float* gsum, float* out, float out_div,
                  hipStream_t st) {
  DISN_TRY(mlp_fused_launch(false, w->g_fused, w->g_w1, w->g_b1, w->g_b2, w->g_b3, gbias_b, w->g_b5, w->g_w6,
                            w->g_b6, nullptr, pts_rot, grid, k0, n, nullptr, nullptr, nullptr, nullptr, gsum,
                            1.0f, st));
  DISN_TRY(mlp_fused_launch(true, w->l_fused, w->l_w1, w->l_b1, w->l_b2, w->l_b3, w->l_b4, w->l_b5, w->l_w6,
                            w->l_b6, pts, pts_rot, grid, k0, n, trans_mat_b, pmap_b, pmap_amax_b, gsum, out,
                            out_div, st));
  return 0;
}
}  // namespace

size_t disn_query_fused_workspace_bytes(int B, int64_t N) {
  if (B <= 0 || N <= 0) return 0;
  return fused_layout(nullptr, B, N).total;
}

int disn_query_fused(const disn_mlp_weights_t* w, const float* pmap, const float* pmap_amax,
                     const float* embedding, const float* trans_mat, const float* pts, const float* pts_rot,
                     int B, int64_t N, float* sdf, void* ws, size_t ws_bytes, void* stream) {
  if (!fused_ok(w) || !pmap || !pmap_amax || !embedding || !trans_mat || !pts || !pts_rot || !sdf || !ws ||
      B <= 0 || N <= 0)
    return DISN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const FusedWs f = fused_layout(ws, B, N);
  if (f.total > ws_bytes) return DISN_E_WS;
  { const int grc = gbias_layer(w, embedding, B, f.gbias, f.gemv_ws, st); if (grc) return grc; }
  for (int b = 0; b < B; ++b) {
    const size_t o = (size_t)b * N;
    const int rc = fused_streams(w, f.gbias + (size_t)b * 512, pmap + (size_t)b * kMapPixels * 512, pmap_amax + b,
                                 trans_mat + (size_t)b * 12, pts + o * 3, pts_rot + o * 3, nullptr, 0, N, f.gsum,
                                 sdf + o, 1.0f, st);
    if (rc) return rc;
  }
  return 0;
}

size_t disn_query_grid_fused_workspace_bytes(int64_t max_points) {
  if (max_points <= 0) return 0;
  return fused_layout(nullptr, 1, max_points).total;
}

int disn_query_grid_fused(const disn_mlp_weights_t* w, const float* pmap, const float* pmap_amax,
                          const float* embedding, const float* trans_mat, const double* sdf_params_host, int R,
                          int64_t k0, int64_t k1, float sdf_weight, float* out, void* ws, size_t ws_bytes,
                          void* stream) {
  GridSpec g;
  if (!fused_ok(w) || !pmap || !pmap_amax || !embedding || !trans_mat || !out || !ws ||
      !grid_spec(sdf_params_host, R, &g))
    return DISN_E_ARG;
  const int64_t total = (int64_t)g.res * g.res * g.res;
  if (k0 < 0 || k1 > total || k0 >= k1 || sdf_weight == 0.0f) return DISN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const FusedWs f = fused_layout(ws, 1, k1 - k0);
  if (f.total > ws_bytes) return DISN_E_WS;
  { const int grc = gbias_layer(w, embedding, 1, f.gbias, f.gemv_ws, st); if (grc) return grc; }
  // one launch per stream over the whole range: no chunking, no per-point workspace but the global sums
  return fused_streams(w, f.gbias, pmap, pmap_amax, trans_mat, nullptr, nullptr, &g, k0, k1 - k0, f.gsum, out,
                       sdf_weight, st);
}

int disn_grid_points(const double* sdf_params_host, int R, int64_t k0, int64_t k1, float* pts,
                     void* stream) {
  GridSpec g;
  if (!pts || !grid_spec(sdf_params_host, R, &g)) return DISN_E_ARG;
  const int64_t total = (int64_t)g.res * g.res * g.res;
  if (k0 < 0 || k1 > total || k0 >= k1) return DISN_E_ARG;
  DISN_TRY(grid_points_launch(g, k0, k1, pts, (hipStream_t)stream));
  return 0;
}

size_t disn_query_grid_workspace_bytes(int64_t max_points) {
  if (max_points <= 0) return 0;
  return query_layout(nullptr, 1, chunk_for(max_points), true, true).total;
}

int disn_query_grid(const disn_mlp_weights_t* w, const float* featmap, const float* embedding,
                    const float* trans_mat, const double* sdf_params_host, int R, int64_t k0,
                    int64_t k1, float sdf_weight, float* out, void* ws, size_t ws_bytes,
                    void* stream) {
  GridSpec g;
  if (!mlp_weights_ok(w) || !featmap || !embedding || !trans_mat || !out || !ws ||
      !grid_spec(sdf_params_host, R, &g))
    return DISN_E_ARG;
  const int64_t total = (int64_t)g.res * g.res * g.res;
  if (k0 < 0 || k1 > total || k0 >= k1 || sdf_weight == 0.0f) return DISN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int chunk = chunk_for(k1 - k0);
  const QueryWs q = query_layout(ws, 1, chunk, true, true);
  if (q.total > ws_bytes) return DISN_E_WS;
  { const int grc = gbias_layer(w, embedding, 1, q.gbias, q.gemv_ws, st); if (grc) return grc; }
  for (int64_t k = k0; k < k1; k += chunk) {
    const int n = (int)((k1 - k) < chunk ? (k1 - k) : chunk);
    DISN_TRY(grid_points_launch(g, k, k + n, q.pts, st));
    DISN_TRY(project_gather_launch(featmap, trans_mat, q.pts, n, q.feat, st));
    // sample_pc == sample_pc_rot on this caller (test/create_sdf.py:268-269)
    const int rc = mlp_chunk(w, q.pts, n, q.gbias, q.feat, out + (k - k0), nullptr, nullptr,
                             sdf_weight, q.mlp, st);
    if (rc) return rc;
  }
  return 0;
}

// Same result as disn_query_grid, chunk-pipelined over two streams: the HBM-bound front of chunk
// i+1 (grid points, projection, gather: ~0.3 ms per 65536 points) runs on ctx->aux into the other
// half of a double buffer while the MFMA-bound MLP of chunk i (~2 ms) runs on `stream`.
size_t disn_query_grid_ctx_workspace_bytes(int64_t max_points) {
  if (max_points <= 0) return 0;
  const int chunk = chunk_for(max_points);
  return query_layout(nullptr, 1, chunk, true, true).total +
         (((size_t)chunk * (DISN_FEAT_DIM + 3) * sizeof(float) + 1023) & ~size_t(255));
}

int disn_query_grid_ctx(disn_ctx_t* ctx, const disn_mlp_weights_t* w, const float* featmap,
                        const float* embedding, const float* trans_mat,
                        const double* sdf_params_host, int R, int64_t k0, int64_t k1,
                        float sdf_weight, float* out, void* ws, size_t ws_bytes, void* stream) {
  GridSpec g;
  if (!ctx || !mlp_weights_ok(w) || !featmap || !embedding || !trans_mat || !out || !ws ||
      !grid_spec(sdf_params_host, R, &g))
    return DISN_E_ARG;
  const int64_t total = (int64_t)g.res * g.res * g.res;
  if (k0 < 0 || k1 > total || k0 >= k1 || sdf_weight == 0.0f) return DISN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int chunk = chunk_for(k1 - k0);
  const QueryWs q = query_layout(ws, 1, chunk, true, true);
  if (disn_query_grid_ctx_workspace_bytes(k1 - k0) > ws_bytes) return DISN_E_WS;
  float* feat[2] = {q.feat, reinterpret_cast<float*>(static_cast<char*>(ws) + q.total)};
  float* pts[2] = {q.pts, feat[1] + (size_t)chunk * DISN_FEAT_DIM};
  // events: 0 fork, 1/2 buffer ready (aux -> main), 3/4 buffer free (main -> aux)
  DISN_TRY(hipEventRecord(ctx->ev[0], st));
  DISN_TRY(hipStreamWaitEvent(ctx->aux, ctx->ev[0], 0));
  { const int grc = gbias_layer(w, embedding, 1, q.gbias, q.gemv_ws, st); if (grc) return grc; }
  int i = 0;
  for (int64_t k = k0; k < k1; k += chunk, ++i) {
    const int n = (int)((k1 - k) < chunk ? (k1 - k) : chunk);
    const int b = i & 1;
    if (i >= 2) DISN_TRY(hipStreamWaitEvent(ctx->aux, ctx->ev[3 + b], 0));
    DISN_TRY(grid_points_launch(g, k, k + n, pts[b], ctx->aux));
    DISN_TRY(project_gather_launch(featmap, trans_mat, pts[b], n, feat[b], ctx->aux));
    DISN_TRY(hipEventRecord(ctx->ev[1 + b], ctx->aux));
    DISN_TRY(hipStreamWaitEvent(st, ctx->ev[1 + b], 0));
    const int rc = mlp_chunk(w, pts[b], n, q.gbias, feat[b], out + (k - k0), nullptr, nullptr,
                             sdf_weight, q.mlp, st);
    if (rc) return rc;
    DISN_TRY(hipEventRecord(ctx->ev[3 + b], st));
  }
  return 0;
}

size_t disn_mc_workspace_bytes(int R) { return (R < 1 || R > 1290) ? 0 : mc_ws_bytes(R); }

int disn_mc_count(const float* sdf, int R, float iso, uint64_t* counts, void* ws, size_t ws_bytes,
                  void* stream) {
  if (!sdf || !counts || !ws || R < 1) return DISN_E_ARG;
  if (R > 1290) return DISN_E_SHAPE;  // 3*(R+1)^3 edge slots must fit the 32-bit scan
  if (ws_bytes < mc_ws_bytes(R)) return DISN_E_WS;
  DISN_TRY(mc_count_launch(sdf, R, iso, reinterpret_cast<unsigned long long*>(counts), ws,
                           (hipStream_t)stream));
  return 0;
}

int disn_mc_emit(const float* sdf, const double* sdf_params_host, int R, float iso, float* verts,
                 int32_t* faces, void* ws, size_t ws_bytes, void* stream) {
  GridSpec g;
  if (!sdf || !verts || !faces || !ws || !grid_spec(sdf_params_host, R, &g)) return DISN_E_ARG;
  if (R > 1290) return DISN_E_SHAPE;
  if (ws_bytes < mc_ws_bytes(R)) return DISN_E_WS;
  DISN_TRY(mc_emit_launch(sdf, g, iso, verts, faces, ws, (hipStream_t)stream));
  return 0;
}

}  // extern "C"

#ifdef DISN_TUNING
namespace disn {
namespace tune {
int x3 = 1, overlap = 1, bf_splits = 0, skip_pack = 0, fused_safe = 0;
int gemm_force[3] = {0, 0, 0};
int gemv_wgs = 0;
int dense_mb = 0, dense_nw = 0, dense_kpw = 0;
int conv_occ = 0, conv_occ_mask = 7, conv_occ_min = 384;
int aux_cu_mode = 0;
int conv_img_major = -1;
int conv_wide_min = 4;
int l4_ranges = 2;
int gather_l16 = 0;
int densew_m64 = -1, densew_c128 = -1;
int conv11_wgs = 0;
int conv11_rt = 0;
int tn_interleave = -1;
int conv5_whole = 1;
int fused_small = 1;
int gemv_rows_cfg = 0;
long long* ch2_stamps = nullptr;
}
}  // namespace disn
extern "C" int disn_tuning_set_ptr(int key, void* p) {
  if (key != 0) return DISN_E_ARG;
  disn::tune::ch2_stamps = static_cast<long long*>(p);
  return 0;
}
// tuning builds only (build.py --tuning -> libdisn_amd_tuning.so): 0 x3, 1 overlap, 2 bf_splits, 3 skip_pack,
// 4 fused_safe, 5-7 gemm_force, 8 gemv_wgs, 9 dense_mb, 10 dense_nw, 11 dense_kpw, 12 conv_occ, 13 conv_occ_mask, 14 conv_occ_min, 15 aux_cu_mode, 16 conv_img_major, 17 conv_wide_min, 18 l4_ranges, 19 gather_l16, 20 densew_m64, 21 densew_c128, 22 conv11_wgs, 23 tn_interleave, 24 conv5_whole, 25 fused_small, 26 conv11_rt, 27 gemv_rows_cfg
extern "C" int disn_tuning_set(int key, int value) {
  int* k[28] = {&disn::tune::x3, &disn::tune::overlap, &disn::tune::bf_splits, &disn::tune::skip_pack,
                &disn::tune::fused_safe, &disn::tune::gemm_force[0], &disn::tune::gemm_force[1],
                &disn::tune::gemm_force[2], &disn::tune::gemv_wgs, &disn::tune::dense_mb,
                &disn::tune::dense_nw, &disn::tune::dense_kpw, &disn::tune::conv_occ, &disn::tune::conv_occ_mask,
                &disn::tune::conv_occ_min, &disn::tune::aux_cu_mode,
                &disn::tune::conv_img_major, &disn::tune::conv_wide_min, &disn::tune::l4_ranges, &disn::tune::gather_l16, &disn::tune::densew_m64, &disn::tune::densew_c128,
                &disn::tune::conv11_wgs, &disn::tune::tn_interleave, &disn::tune::conv5_whole, &disn::tune::fused_small, &disn::tune::conv11_rt, &disn::tune::gemv_rows_cfg};
  if (key < 0 || key > 27) return DISN_E_ARG;
  *k[key] = value;
  return 0;
}
#endif
