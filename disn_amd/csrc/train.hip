// Training step of the path (BASELINE config 5): forward with every activation kept,
// the losses of get_loss (models/model_normalization.py:254-300), the gradient of every variable
// (train/train_sdf.py:266-268 minimises over ALL globals: the VGG is fine-tuned), and the
// tf.train.AdamOptimizer(beta1=0.5) update (:251).  Parameters live in ONE flat device buffer in the
// reference's own variable layouts (disn_param_layout), gradients / Adam slots in buffers of the same
// shape: the data-parallel all-reduce is one collective over `grads`, the update one kernel.
// No allocation, no host synchronisation; everything is enqueued on `stream`.
#include "../../include/disn_amd.h"

#include "kernels.hpp"

using namespace disn;

#define DISN_TRY(expr)                    \
  do {                                    \
    hipError_t _e = (expr);               \
    if (_e != hipSuccess) return (int)_e; \
  } while (0)
#define DISN_RC(expr)       \
  do {                      \
    const int _rc = (expr); \
    if (_rc) return _rc;    \
  } while (0)

namespace {

struct Bump {
  char* base;
  size_t off;
  explicit Bump(void* b) : base(static_cast<char*>(b)), off(0) {}
  float* take(size_t floats) {
    off = (off + 255) & ~size_t(255);
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += floats * sizeof(float);
    return p;
  }
};

struct ConvL {
  int cin, cout, hw, tap;
  bool pool;
};
const ConvL kConv[13] = {
    {3, 64, 224, -1, false},   {64, 64, 224, 0, true},    {64, 128, 112, -1, false},
    {128, 128, 112, 1, true},  {128, 256, 56, -1, false}, {256, 256, 56, -1, false},
    {256, 256, 56, 2, true},   {256, 512, 28, -1, false}, {512, 512, 28, -1, false},
    {512, 512, 28, 3, true},   {512, 512, 14, -1, false}, {512, 512, 14, -1, false},
    {512, 512, 14, 4, true}};
const int kTapHw[5] = {224, 112, 56, 28, 14}, kTapCh[5] = {64, 128, 256, 512, 512};
const int kTapOff[5] = {0, 64, 192, 448, 960};
inline int conv_kpad(int cin) { return cin == 3 ? 32 : 9 * cin; }

// variable order of the flat parameter buffer
//   0..25  vgg_16/convX/convX_Y/{weights,biases}  (2 per layer)
//   26..31 vgg_16/fc6, fc7, fc8 {weights,biases}
//   32..43 sdfprediction/{fold1/conv1,2,3, fold2/conv1,2,5}/{weights,biases}
//   44..55 sdfprediction_imgfeat/ same
enum { V_FC = 26, V_G = 32, V_L = 44 };
const int kMlpK[2][6] = {{3, 64, 256, 1536, 512, 256}, {3, 64, 256, 1984, 512, 256}};
const int kMlpN[6] = {64, 256, 512, 512, 256, 1};

void build_layout(disn_param_layout_t* L) {
  int64_t off = 0;
  auto put = [&](int idx, int64_t count) {
    L->offset[idx] = off;
    L->count[idx] = count;
    off += (count + 63) & ~int64_t(63);
  };
  for (int i = 0; i < 13; ++i) {
    put(2 * i, (int64_t)9 * kConv[i].cin * kConv[i].cout);
    put(2 * i + 1, kConv[i].cout);
  }
  const int64_t fk[3] = {25088, 4096, 4096}, fn[3] = {4096, 4096, DISN_EMBED_DIM};
  for (int i = 0; i < 3; ++i) {
    put(V_FC + 2 * i, fk[i] * fn[i]);
    put(V_FC + 2 * i + 1, fn[i]);
  }
  for (int s = 0; s < 2; ++s)
    for (int i = 0; i < 6; ++i) {
      put((s ? V_L : V_G) + 2 * i, (int64_t)kMlpK[s][i] * kMlpN[i]);
      put((s ? V_L : V_G) + 2 * i + 1, kMlpN[i]);
    }
  L->total = off;
}

int dense_fwd(const float* a1, int lda1, int k1, const float* a2, int lda2, int K, int M,
              const float* bp, const float* bias, int N, int relu, float* out, float* ws,
              size_t ws_bytes, hipStream_t st, int ns = 0, int rows_per_bias = 0) {
  // ns != 0: bp is a pack_bf16_launch image; 1 = bf16 multiply, 3 = three-term split (fp32-accurate)
  // rows_per_bias != 0: bias is [M / rows_per_bias][N], one row per image
  GemmParams p{};
  p.a1 = a1; p.lda1 = lda1; p.k1 = k1; p.a2 = a2; p.lda2 = lda2;
  p.M = M; p.N = N; p.K = K;
  p.bp = bp; p.bias = bias; p.rows_per_bias = rows_per_bias;
  p.out = out; p.ldc = N; p.relu = relu;
  if (ns) {
    DISN_TRY(gemm_bf16_launch(p, GEMM_DENSE, bp, ws, ws ? ws_bytes : 0, st, ns));
    return 0;
  }
  const GemmPlan pl = gemm_plan(M, N, K, ws ? ws_bytes : 0);
  DISN_TRY(gemm_launch(p, GEMM_DENSE, pl, ws, st));
  return 0;
}

int conv_fwd(const float* in, int B, int H, int W, int Cin, const float* bp, const float* bias, int Cout,
             int relu, float* out, float* ws, size_t ws_bytes, hipStream_t st, int ns = 0) {
  GemmParams p{};
  p.a1 = in;
  p.H = H; p.W = W; p.Cin = Cin;
  p.M = B * H * W; p.N = Cout; p.K = conv_kpad(Cin);
  p.bp = bp; p.bias = bias; p.rows_per_bias = 0;
  p.out = out; p.ldc = Cout; p.relu = relu;
  if (ns && Cin != 3) {
    DISN_TRY(gemm_bf16_launch(p, GEMM_CONV3, bp, ws, ws ? ws_bytes : 0, st, ns));
    return 0;
  }
  const GemmPlan pl = gemm_plan(p.M, p.N, p.K, ws ? ws_bytes : 0);
  DISN_TRY(gemm_launch(p, Cin == 3 ? GEMM_CONV3_C3 : GEMM_CONV3, pl, ws, st));
  return 0;
}

size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

// scratch shared by the backward building blocks
struct BwdWs {
  float *wT, *zero, *gemm_ws, *tn_ws, *red_ws;
  size_t gemm_ws_bytes, total;
  int ns = 0;  // GEMM arithmetic: 0 f32-input MFMA; 1 bf16 multiply; 3 three-term bf16 split (fp32-accurate)
};

// capacity for: one packed transposed weight (wt_floats), GEMMs with up to max_m rows
BwdWs bwd_layout(Bump& b, size_t wt_floats, long max_m, size_t gemm_ws_bytes, size_t red_bytes) {
  BwdWs w;
  w.wT = b.take(wt_floats * 3 / 2);  // room for the 6-byte three-term image
  w.zero = b.take(4096);
  w.gemm_ws_bytes = gemm_ws_bytes;
  w.gemm_ws = b.take(gemm_ws_bytes / sizeof(float) + 1);
  w.tn_ws = b.take(gemm_tn_ws_bytes(max_m, 128, 128) / sizeof(float));
  w.red_ws = b.take(red_bytes / sizeof(float) + 1);
  (void)max_m;
  w.total = (b.off + 255) & ~size_t(255);
  return w;
}

// dense layer backward: dz [M][N] is the gradient w.r.t. the layer's PRE-activation output
// (already ReLU-masked); computes dW [K][N] (+ wd W) with a [M][lda] (K columns), and, when da != null,
// dA [M][K] = dz W^T.  W raw [K][N].
int dense_bwd(const float* a, int lda, int K, const float* w_kn, const float* dz, long M, int N,
              float wd, float* da, float* dw, const BwdWs& s, hipStream_t st,
              const float* prepacked = nullptr) {
  TnParams t{};
  t.a = a; t.lda = lda; t.b = dz; t.ldb = N; t.M = M; t.P = K; t.Q = N;
  t.c = dw; t.ldc = N; t.Cin = 0; t.l2 = wd; t.wcur = w_kn; t.bf16 = s.ns == 1;
  DISN_TRY(gemm_tn_launch(t, s.tn_ws, st));
  if (da) {
    const float* wt = prepacked;  // W^T in fragment order: packed at the start of the step, or here
    if (!wt) {
      if (s.ns)
        DISN_TRY(pack_bf16_launch(w_kn, 1, K, N, s.wT, st, s.ns));
      else
        DISN_TRY(pack_kn_T_launch(w_kn, K, N, s.wT, st));
      wt = s.wT;
    }
    DISN_RC(dense_fwd(dz, N, N, nullptr, 0, N, (int)M, wt, s.zero, K, 0, da, s.gemm_ws,
                      s.gemm_ws_bytes, st, s.ns));
  }
  return 0;
}

// 3x3 conv backward; dz [B,H,W,Cout] already ReLU-masked; x [B,H,W,Cin]; w raw [3,3,Cin,Cout]
// col: [B*H*W][64] scratch, only for Cin == 3
// h2img != nullptr: the data gradient through conv_h2.hip / conv_h2w.hip -- dx = conv(dz, w') with the mirrored,
// transposed kernel's two-term f16 image (conv_h2_pack_launch(flip_t)), dz split per image against its own maximum
// (amax: B x 64 slots) -- instead of the implicit-GEMM kernels on `prepacked`
// amax_x (with amax): B x 64 floats whose maximum bounds |x| -- then the weight gradient of the fp32-accurate mode
// (s.ns == 3) runs as a two-term f16 split too (gemm_tn_mfma.hip, mode 2) instead of on the fp32 MFMA
int conv_bwd(const float* x, int B, int H, int W, int Cin, const float* w, const float* dz, int Cout,
             float wd, float* dx, float* dw, float* col, const BwdWs& s, hipStream_t st,
             const float* prepacked = nullptr, const void* h2img = nullptr, float* amax = nullptr,
             const float* amax_x = nullptr, bool amax_ready = false) {
  const long M = (long)B * H * W;
  // maxima of dz per image: the data gradient's operand scales, the weight gradient's bound (amax_ready: already
  // written by the ReLU-mask / bias-gradient pass, relu_bwd_colsum_launch)
  if (amax && Cin != 3 && !amax_ready) {
    DISN_TRY(hipMemsetAsync(amax, 0, (size_t)B * 64 * sizeof(float), st));
    DISN_TRY(amax64_accumulate_launch(dz, (size_t)H * W * Cout, amax, st, B, 64));
  }
  if (Cin == 3) {
    DISN_TRY(im2col_c3_launch(x, B, H, W, col, st));
    TnParams t{};
    t.a = col; t.lda = 64; t.b = dz; t.ldb = Cout; t.M = M; t.P = 64; t.Q = Cout;
    t.c = s.wT; t.ldc = Cout; t.Cin = 0; t.l2 = 0.f; t.wcur = nullptr;  // [64][Cout], 27 rows used
    DISN_TRY(gemm_tn_launch(t, s.tn_ws, st));
    DISN_TRY(axpby_launch(s.wT, w, wd, (size_t)27 * Cout, dw, st));
  } else {
    TnParams t{};
    t.a = x; t.lda = Cin; t.b = dz; t.ldb = Cout; t.M = M; t.P = 9 * Cin; t.Q = Cout;
    t.c = dw; t.ldc = Cout; t.H = H; t.W = W; t.Cin = Cin; t.l2 = wd; t.wcur = w; t.bf16 = s.ns == 1;
    if (s.ns == 3 && amax && amax_x) {
      t.bf16 = 2;
      t.amax_a = amax_x; t.amax_a_n = B * 64;
      t.amax_b = amax; t.amax_b_n = B * 64;
    }
    DISN_TRY(gemm_tn_launch(t, s.tn_ws, st));
  }
  if (dx && h2img) {
    DISN_TRY(conv_h2_launch(dz, B, H, W, Cout, h2img, s.zero, Cin, 0, amax, dx, nullptr, nullptr, st, 18, 64));
  } else if (dx) {
    const float* wt = prepacked;
    if (!wt) {
      if (s.ns)
        DISN_TRY(pack_bf16_launch(w, 2, Cin, Cout, s.wT, st, s.ns));
      else
        DISN_TRY(pack_conv_bwd_launch(w, Cin, Cout, s.wT, st));
      wt = s.wT;
    }
    DISN_RC(conv_fwd(dz, B, H, W, Cout, wt, s.zero, Cin, 0, dx, s.gemm_ws, s.gemm_ws_bytes, st, s.ns));
  }
  return 0;
}

// ---- the full step ------------------------------------------------------------------
struct TrainWs {
  // packed forward weights
  float* conv_p[13];
  float *g_p2, *g_p3, *g_p4, *g_p5, *l_p2, *l_p3, *l_p4, *l_p5;
  // packed weights of the data-gradient GEMMs (flipped conv kernels, transposed MLP weights)
  float* conv_bT[13];
  float *g_t2, *g_t3, *g_t4, *g_t5, *l_t2, *l_t3, *l_t4, *l_t4f, *l_t5;
  // activations
  float *resized, *act[13], *pooled[13], *h6, *h7, *emb, *gbias, *xy, *feat;
  float *g1, *l1, *g2, *l2, *g3, *l3, *g4, *l4, *g5, *l5;
  // gradients
  float *dpred, *d5, *d4, *d3, *d2, *d1, *dfeat, *dmap, *dgbias, *demb, *dz7, *dz6, *dpool5, *gA, *gB;
  float *col, *fc_ws, *sumsq_ws, *red_aux;
  // forward convolutions through conv_h2.hip / conv_h2w.hip: two-term f16 weight images (re-packed every step) and
  // the activation-maximum slots [14][B][64] of the layer chain
  float* conv_h2img[13];
  float* conv_h2bT[13];   // the same for the data gradients (mirrored taps, transposed channels)
  float* amax;
  float* amax_bwd;        // [13][B][64]: per-image maxima of every layer gradient (written by the ReLU-mask pass)
  float* wmax;            // [16]: max |w| of the 12 packed convolution tensors
  BwdWs bw;
  size_t total;
};

size_t train_gemm_ws(int B, long M) {
  size_t m = 0;
  // forward + backward-data GEMM shapes (rows, N, K)
  for (int i = 0; i < 13; ++i) {
    const ConvL& L = kConv[i];
    const int rows = B * L.hw * L.hw;
    m = max_sz(m, gemm_plan(rows, L.cout, conv_kpad(L.cin)).ws_bytes);
    if (i > 0) {
      m = max_sz(m, gemm_plan(rows, L.cin, 9 * L.cout).ws_bytes);
      m = max_sz(m, gemm_bf16_ws_bytes(rows, L.cout, 9 * L.cin));  // split-K partials of the bf16 path
      m = max_sz(m, gemm_bf16_ws_bytes(rows, L.cin, 9 * L.cout));
    }
  }
  const int shapes[9][2] = {{256, 64}, {512, 256}, {512, 512}, {512, 1984}, {256, 512},
                            {64, 256}, {512, 256}, {1472, 512}, {256, 512}};
  for (auto& s : shapes) {
    m = max_sz(m, gemm_plan((int)M, s[0], s[1]).ws_bytes);
    m = max_sz(m, gemm_bf16_ws_bytes((int)M, s[0], s[1]));
    if (B > 0) {
      m = max_sz(m, gemm_plan((int)(M / B), s[0], s[1]).ws_bytes);
      m = max_sz(m, gemm_bf16_ws_bytes((int)(M / B), s[0], s[1]));
    }
  }
  return m;
}

TrainWs train_layout(void* ws, int B, int N) {
  Bump b(ws);
  TrainWs t;
  const long M = (long)B * N;
  // (x 3/2: the three-term bf16 image is 6 bytes per weight)
  for (int i = 0; i < 13; ++i) t.conv_p[i] = b.take((size_t)conv_kpad(kConv[i].cin) * kConv[i].cout * 3 / 2);
  t.g_p2 = b.take(64 * 256 * 3 / 2); t.g_p3 = b.take(256 * 512 * 3 / 2); t.g_p4 = b.take(512 * 512 * 3 / 2); t.g_p5 = b.take(512 * 256 * 3 / 2);
  t.l_p2 = b.take(64 * 256 * 3 / 2); t.l_p3 = b.take(256 * 512 * 3 / 2); t.l_p4 = b.take(1984 * 512 * 3 / 2); t.l_p5 = b.take(512 * 256 * 3 / 2);
  t.conv_bT[0] = nullptr;
  for (int i = 1; i < 13; ++i) t.conv_bT[i] = b.take((size_t)9 * kConv[i].cin * kConv[i].cout * 3 / 2);
  t.g_t2 = b.take(64 * 256 * 3 / 2); t.g_t3 = b.take(256 * 512 * 3 / 2); t.g_t4 = b.take(512 * 512 * 3 / 2); t.g_t5 = b.take(512 * 256 * 3 / 2);
  t.l_t2 = b.take(64 * 256 * 3 / 2); t.l_t3 = b.take(256 * 512 * 3 / 2); t.l_t4 = b.take(512 * 512 * 3 / 2);
  t.l_t4f = b.take(1472 * 512 * 3 / 2); t.l_t5 = b.take(512 * 256 * 3 / 2);
  t.resized = b.take((size_t)B * 224 * 224 * 3);
  for (int i = 0; i < 13; ++i) {
    const ConvL& L = kConv[i];
    t.act[i] = b.take((size_t)B * L.hw * L.hw * L.cout);
    t.pooled[i] = L.pool ? b.take((size_t)B * (L.hw / 2) * (L.hw / 2) * L.cout) : nullptr;
  }
  t.h6 = b.take((size_t)B * 4096); t.h7 = b.take((size_t)B * 4096);
  t.emb = b.take((size_t)B * DISN_EMBED_DIM); t.gbias = b.take((size_t)B * 512);
  t.xy = b.take((size_t)M * 2); t.feat = b.take((size_t)M * DISN_FEAT_DIM);
  t.g1 = b.take((size_t)M * 64); t.l1 = b.take((size_t)M * 64);
  t.g2 = b.take((size_t)M * 256); t.l2 = b.take((size_t)M * 256);
  t.g3 = b.take((size_t)M * 512); t.l3 = b.take((size_t)M * 512);
  t.g4 = b.take((size_t)M * 512); t.l4 = b.take((size_t)M * 512);
  t.g5 = b.take((size_t)M * 256); t.l5 = b.take((size_t)M * 256);
  t.dpred = b.take((size_t)M);
  t.d5 = b.take((size_t)M * 256); t.d4 = b.take((size_t)M * 512); t.d3 = b.take((size_t)M * 512);
  t.d2 = b.take((size_t)M * 256); t.d1 = b.take((size_t)M * 64);
  t.dfeat = b.take((size_t)M * DISN_FEAT_DIM);
  t.dmap = b.take((size_t)B * 137 * 137 * DISN_FEAT_DIM);
  t.dgbias = b.take((size_t)B * 512); t.demb = b.take((size_t)B * DISN_EMBED_DIM);
  t.dz7 = b.take((size_t)B * 4096); t.dz6 = b.take((size_t)B * 4096);
  t.dpool5 = b.take((size_t)B * 25088);
  t.gA = b.take((size_t)B * 224 * 224 * 64); t.gB = b.take((size_t)B * 224 * 224 * 64);
  t.col = b.take((size_t)B * 224 * 224 * 64);
  size_t fws = gemv_ws_bytes(B, 25088, 4096);
  fws = max_sz(fws, gemv_ws_bytes(B, 4096, 4096));
  fws = max_sz(fws, gemv_ws_bytes(B, 4096, DISN_EMBED_DIM));
  fws = max_sz(fws, gemv_ws_bytes(B, DISN_EMBED_DIM, 512));
  t.fc_ws = b.take(fws / sizeof(float) + 1);
  t.sumsq_ws = b.take(32 * 256);
  t.conv_h2img[0] = nullptr;
  for (int i = 1; i < 13; ++i) t.conv_h2img[i] = b.take(conv_h2_image_bytes(kConv[i].cin, kConv[i].cout) / sizeof(float) + 1);
  t.conv_h2bT[0] = nullptr;
  for (int i = 1; i < 13; ++i) t.conv_h2bT[i] = b.take(conv_h2_image_bytes(kConv[i].cout, kConv[i].cin) / sizeof(float) + 1);
  t.amax = b.take((size_t)27 * B * 64);       // 14 slot sets of the forward chain + 13 of the layer gradients
  t.amax_bwd = t.amax + (size_t)14 * B * 64;  // [13][B][64], zeroed with the others by the resize launch
  t.wmax = b.take(16);
  t.red_aux = b.take(colsum_ws_bytes(B, 4096) / sizeof(float) + 1);
  size_t red = colsum_ws_bytes(M, 512);
  for (int i = 0; i < 13; ++i)  // bias-gradient partials of every conv layer (chunks x Cout)
    red = max_sz(red, colsum_ws_bytes((long)B * kConv[i].hw * kConv[i].hw, kConv[i].cout));
  red = max_sz(red, colsum_ws_bytes(B, 4096));
  red = max_sz(red, max_sz(final_bwd_ws_bytes(M), colsum_ws_bytes(M, DISN_FEAT_DIM)));
  t.bw = bwd_layout(b, (size_t)9 * 512 * 512, M, train_gemm_ws(B, M), red);
  t.total = (b.off + 255) & ~size_t(255);
  return t;
}

}  // namespace

extern "C" {

int disn_param_layout(disn_param_layout_t* out) {
  if (!out) return DISN_E_ARG;
  build_layout(out);
  return 0;
}

// ---- building blocks (unit-test / composition surface) -------------------------------
size_t disn_dense_backward_workspace_bytes(int M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0 || K % 64 || N % 64) return 0;
  Bump b(nullptr);
  const size_t g = max_sz(max_sz(gemm_plan(M, K, N).ws_bytes, gemm_bf16_ws_bytes(M, K, N)), 256);
  return bwd_layout(b, (size_t)K * N, M, g, colsum_ws_bytes(M, N)).total;
}

int disn_dense_backward(const float* a, int lda, int K, const float* w_kn, const float* y, float* dy,
                        int M, int N, float wd, int compute_bf16, float* da, float* dw, float* db,
                        void* ws, size_t ws_bytes, void* stream) {
  if (!a || !w_kn || !dy || !dw || !db || !ws || M <= 0 || K <= 0 || N <= 0) return DISN_E_ARG;
  if (K % 64 || N % 64 || lda < K || lda % 4) return DISN_E_SHAPE;
  if (ws_bytes < disn_dense_backward_workspace_bytes(M, K, N)) return DISN_E_WS;
  hipStream_t st = (hipStream_t)stream;
  Bump b(ws);
  BwdWs s = bwd_layout(b, (size_t)K * N, M,
                       max_sz(max_sz(gemm_plan(M, K, N).ws_bytes, gemm_bf16_ws_bytes(M, K, N)), 256),
                       colsum_ws_bytes(M, N));
  s.ns = compute_bf16 == 2 ? 3 : (compute_bf16 ? 1 : 0);
  DISN_TRY(hipMemsetAsync(s.zero, 0, 4096 * sizeof(float), st));
  DISN_TRY(relu_bwd_colsum_launch(dy, y, M, N, y != nullptr, db, s.red_ws, st));
  return dense_bwd(a, lda, K, w_kn, dy, M, N, wd, da, dw, s, st);
}

size_t disn_conv3x3_backward_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout % 64 || !(Cin == 3 || Cin % 64 == 0)) return 0;
  Bump b(nullptr);
  const int M = B * H * W;
  const size_t g = Cin == 3 ? 256
                            : max_sz(max_sz(gemm_plan(M, Cin, 9 * Cout).ws_bytes,
                                            gemm_bf16_ws_bytes(M, Cin, 9 * Cout)), 256);
  if (Cin == 3) b.take((size_t)M * 64);
  else {  // the data gradient's two-term f16 image and its B x 64 maxima (compute_bf16 != 0)
    b.take(conv_h2_image_bytes(Cout, Cin) / sizeof(float) + 1);
    b.take((size_t)B * 128);
  }
  return bwd_layout(b, (size_t)9 * (Cin == 3 ? 64 : Cin) * Cout, M, g, colsum_ws_bytes(M, Cout)).total;
}

int disn_conv3x3_backward(const float* x, int B, int H, int W, int Cin, const float* w_hwio,
                          const float* y, float* dy, int Cout, float wd, int compute_bf16, float* dx,
                          float* dw, float* db, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !w_hwio || !dy || !dw || !db || !ws || B <= 0 || H <= 0 || W <= 0) return DISN_E_ARG;
  if (Cout % 64 || !(Cin == 3 || Cin % 64 == 0) || (Cin == 3 && dx)) return DISN_E_SHAPE;
  if (ws_bytes < disn_conv3x3_backward_workspace_bytes(B, H, W, Cin, Cout)) return DISN_E_WS;
  hipStream_t st = (hipStream_t)stream;
  Bump b(ws);
  const int M = B * H * W;
  float* col = Cin == 3 ? b.take((size_t)M * 64) : nullptr;
  float* h2img = Cin == 3 ? nullptr : b.take(conv_h2_image_bytes(Cout, Cin) / sizeof(float) + 1);
  float* h2amax = Cin == 3 ? nullptr : b.take((size_t)B * 128);
  const size_t g = Cin == 3 ? 256
                            : max_sz(max_sz(gemm_plan(M, Cin, 9 * Cout).ws_bytes,
                                            gemm_bf16_ws_bytes(M, Cin, 9 * Cout)), 256);
  BwdWs s = bwd_layout(b, (size_t)9 * (Cin == 3 ? 64 : Cin) * Cout, M, g, colsum_ws_bytes(M, Cout));
  s.ns = Cin == 3 ? 0 : (compute_bf16 == 2 ? 3 : (compute_bf16 ? 1 : 0));
  DISN_TRY(hipMemsetAsync(s.zero, 0, 4096 * sizeof(float), st));
  DISN_TRY(relu_bwd_colsum_launch(dy, y, M, Cout, y != nullptr, db, s.red_ws, st));
  // as in disn_train_step: with compute_bf16 != 0 the data gradient runs through conv_h2.hip / conv_h2w.hip
  const bool h2 = compute_bf16 != 0 && dx && Cin != 3 && conv_h2_supported(H, W, Cout, Cin);
  if (h2) {
    float* scratch = h2img + (size_t)Cin * 9 * Cout + 2;
    DISN_TRY(conv_h2_pack_launch(w_hwio, Cout, Cin, h2img, scratch, st, 9, 1));
  }
  const bool maxima = compute_bf16 != 0 && Cin != 3;   // dz maxima: data gradient and (mode 2) weight gradient
  if (maxima && compute_bf16 == 2) {
    DISN_TRY(hipMemsetAsync(h2amax + (size_t)B * 64, 0, (size_t)B * 64 * sizeof(float), st));
    DISN_TRY(amax64_accumulate_launch(x, (size_t)H * W * Cin, h2amax + (size_t)B * 64, st, B, 64));
  }
  return conv_bwd(x, B, H, W, Cin, w_hwio, dy, Cout, wd, dx, dw, col, s, st, nullptr, h2 ? h2img : nullptr,
                  maxima ? h2amax : nullptr, maxima && compute_bf16 == 2 ? h2amax + (size_t)B * 64 : nullptr);
}

int disn_maxpool2x2_backward(const float* x, const float* dy, int B, int H, int W, int C, float* dx,
                             void* stream) {
  if (!x || !dy || !dx || B <= 0 || H < 2 || W < 2) return DISN_E_ARG;
  if (C % 4 || H % 2 || W % 2) return DISN_E_SHAPE;
  DISN_TRY(maxpool_bwd_launch(x, dy, B, H, W, C, dx, (hipStream_t)stream));
  return 0;
}

size_t disn_resize_bilinear_backward_workspace_bytes(int B, int Hin, int Win, int C, int Hout, int Wout) {
  if (B <= 0 || Hin <= 0 || Win <= 0 || C <= 0 || Hout <= 0 || Wout <= 0) return 0;
  return resize_bwd_ws_bytes(B, Hin, Win, C, Hout, Wout);
}

int disn_resize_bilinear_backward(const float* dout, int B, int Hin, int Win, int C, int Hout, int Wout,
                                  int out_cstride, int out_coff, float* din, int accumulate, void* ws,
                                  size_t ws_bytes, void* stream) {
  if (!dout || !din || B <= 0 || Hin <= 0 || Win <= 0 || C <= 0 || Hout <= 0 || Wout <= 0)
    return DISN_E_ARG;
  if (C % 4 || out_cstride % 4 || out_coff % 4 || out_coff < 0 || out_coff + C > out_cstride)
    return DISN_E_SHAPE;
  const size_t need = resize_bwd_ws_bytes(B, Hin, Win, C, Hout, Wout);
  if (need > 0 && (!ws || ws_bytes < need)) return DISN_E_WS;
  DISN_TRY(resize_bwd_launch(dout, B, Hin, Win, C, Hout, Wout, out_cstride, out_coff, din, accumulate,
                             need > 0 ? (float*)ws : nullptr, (hipStream_t)stream));
  return 0;
}

int disn_gather_backward(const float* dfeat, const float* xy, int B, int N, float* dfeatmap,
                         void* stream) {
  if (!dfeat || !xy || !dfeatmap || B <= 0 || N <= 0) return DISN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  DISN_TRY(hipMemsetAsync(dfeatmap, 0, (size_t)B * 137 * 137 * DISN_FEAT_DIM * sizeof(float), st));
  DISN_TRY(gather_bwd_launch(dfeat, xy, B, N, dfeatmap, st));
  return 0;
}

int disn_adam_update(float* params, const float* grads, float* m, float* v, int64_t n, float lr_t,
                     float beta1, float beta2, float eps, float grad_scale, void* stream) {
  if (!params || !grads || !m || !v || n <= 0) return DISN_E_ARG;
  if (n % 4) return DISN_E_SHAPE;
  DISN_TRY(adam_launch(params, grads, m, v, (size_t)n, lr_t, beta1, beta2, eps, grad_scale,
                       (hipStream_t)stream));
  return 0;
}

// ---- the step -----------------------------------------------------------------------
size_t disn_train_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0 || (long)B * N > 65536) return 0;
  return train_layout(nullptr, B, N).total;
}

int disn_train_step(disn_ctx_t* ctx, const float* params, float* grads, const float* img,
                    const float* trans_mat, const float* pts, const float* pts_rot, const float* gt, int B,
                    int N, float wd, float sdf_weight, float mask_weight, int compute_bf16, float* pred,
                    float* losses, void* head_ready_event, void* ws, size_t ws_bytes, void* stream) {
  if (!params || !grads || !img || !trans_mat || !pts || !pts_rot || !gt || !pred || !losses || !ws ||
      B <= 0 || N <= 0)
    return DISN_E_ARG;
  if ((long)B * N > 65536) return DISN_E_SHAPE;
  const TrainWs t = train_layout(ws, B, N);
  if (ws_bytes < t.total) return DISN_E_WS;
  hipStream_t st = (hipStream_t)stream;
  // HBM-bound side work (weight-norm sum, fc6-fc8 forward and backward: ~1.3 GB of weight traffic)
  // runs on the context's auxiliary stream under MFMA-bound GEMMs of the main stream
  hipStream_t as = ctx ? ctx->aux : st;
  disn_param_layout_t L;
  build_layout(&L);
  auto P = [&](int idx) { return params + L.offset[idx]; };
  auto G = [&](int idx) { return grads + L.offset[idx]; };
  const long M = (long)B * N;
  BwdWs s = t.bw;
  const int bf = compute_bf16 == 2 ? 3 : (compute_bf16 ? 1 : 0);  // planes of the weight images
  s.ns = bf;
  float* gws = s.gemm_ws;
  const size_t gwb = s.gemm_ws_bytes;

  // ---------------- forward ----------------
  // The 13 convolutions of the forward (models/CNN/vgg.py:187-196) through the inference kernels -- conv1_1 direct,
  // conv_h2.hip / conv_h2w.hip (two-term f16 split: fp32-accurate, pools fused) -- in the fp32-accurate and the
  // mixed-precision mode alike: at eight samples 0.8 ms instead of 2.6 (three-term) / ~1.0 ms (bf16 one-term), and the
  // kept activations are the fp32-accurate ones.  compute_bf16 == 0 (every product on the f32-input MFMA) keeps the
  // implicit-GEMM forward.  The weight images are re-packed every step (the weights change every step).
  const bool h2fwd = compute_bf16 != 0;
  // weights in MFMA fragment order (they change every step): fp32, or bf16 in the same storage
  // all 41 re-packs (forward + data-gradient views) in one launch
  {
    PackJobs jobs{};
    if (!h2fwd) pack_job_add(jobs, P(0), t.conv_p[0], 0, 27, 64, 0);  // conv1_1 (K = 27) stays on the fp32 path
    for (int i = 1; i < 13; ++i) {
      if (!h2fwd) pack_job_add(jobs, P(2 * i), t.conv_p[i], 0, 9 * kConv[i].cin, kConv[i].cout, bf);
      if (!h2fwd) pack_job_add(jobs, P(2 * i), t.conv_bT[i], 2, kConv[i].cin, kConv[i].cout, bf);
    }
    const float* gw4 = P(V_G + 6);  // rows 0..511 point part, 512..1535 global part
    const float* lw4 = P(V_L + 6);  // rows 0..511 point part, 512..1983 image-feature part
    pack_job_add(jobs, P(V_G + 2), t.g_p2, 0, 64, 256, bf);
    pack_job_add(jobs, P(V_G + 4), t.g_p3, 0, 256, 512, bf);
    pack_job_add(jobs, gw4, t.g_p4, 0, 512, 512, bf);
    pack_job_add(jobs, P(V_G + 8), t.g_p5, 0, 512, 256, bf);
    pack_job_add(jobs, P(V_L + 2), t.l_p2, 0, 64, 256, bf);
    pack_job_add(jobs, P(V_L + 4), t.l_p3, 0, 256, 512, bf);
    pack_job_add(jobs, lw4, t.l_p4, 0, 1984, 512, bf);
    pack_job_add(jobs, P(V_L + 8), t.l_p5, 0, 512, 256, bf);
    pack_job_add(jobs, P(V_G + 2), t.g_t2, 1, 64, 256, bf);
    pack_job_add(jobs, P(V_G + 4), t.g_t3, 1, 256, 512, bf);
    pack_job_add(jobs, gw4, t.g_t4, 1, 512, 512, bf);
    pack_job_add(jobs, P(V_G + 8), t.g_t5, 1, 512, 256, bf);
    pack_job_add(jobs, P(V_L + 2), t.l_t2, 1, 64, 256, bf);
    pack_job_add(jobs, P(V_L + 4), t.l_t3, 1, 256, 512, bf);
    pack_job_add(jobs, lw4, t.l_t4, 1, 512, 512, bf);
    pack_job_add(jobs, lw4 + (size_t)512 * 512, t.l_t4f, 1, 1472, 512, bf);
    pack_job_add(jobs, P(V_L + 8), t.l_t5, 1, 512, 256, bf);
    DISN_TRY(pack_multi_launch(jobs, st));
  }
  DISN_TRY(hipMemsetAsync(s.zero, 0, 4096 * sizeof(float), st));
  if (ctx) {
    DISN_TRY(hipEventRecord(ctx->ev[0], st));
    DISN_TRY(hipStreamWaitEvent(as, ctx->ev[0], 0));
  }
  {  // regularization loss: depends on the parameters only
    SumsqSegs segs{};
    int n = 0;
    for (int i = 0; i < 13; ++i) { segs.off[n] = L.offset[2 * i]; segs.cnt[n++] = L.count[2 * i]; }
    for (int i = 0; i < 3; ++i) { segs.off[n] = L.offset[V_FC + 2 * i]; segs.cnt[n++] = L.count[V_FC + 2 * i]; }
    for (int i = 0; i < 6; ++i) { segs.off[n] = L.offset[V_G + 2 * i]; segs.cnt[n++] = L.count[V_G + 2 * i]; }
    for (int i = 0; i < 6; ++i) { segs.off[n] = L.offset[V_L + 2 * i]; segs.cnt[n++] = L.count[V_L + 2 * i]; }
    segs.n = n;
    DISN_TRY(sumsq_launch(params, segs, 0.5f * wd, losses + 3, t.sumsq_ws, as));
  }

  if (h2fwd) {  // 12 forward + 12 data-gradient images (mirrored taps, transposed channels): three launches in all
    ConvH2PackJobs cj{};
    cj.wmax = t.wmax;
    for (int i = 1; i < 13; ++i) {
      conv_h2_pack_job_add(cj, P(2 * i), kConv[i].cin, kConv[i].cout, t.conv_h2img[i], 0, i - 1);
      conv_h2_pack_job_add(cj, P(2 * i), kConv[i].cin, kConv[i].cout, t.conv_h2bT[i], 1, i - 1);
    }
    DISN_TRY(conv_h2_pack_multi_launch(cj, st));
  }
  DISN_TRY(resize_bilinear_launch(img, B, DISN_IMG_H, DISN_IMG_W, 3, t.resized, DISN_VGG_SIZE,
                                  DISN_VGG_SIZE, 3, 0, st, 0, h2fwd ? t.amax : nullptr, h2fwd ? 27 * B * 64 : 0));
  const float* x = t.resized;
  for (int i = 0; i < 13; ++i) {
    const ConvL& c = kConv[i];
    if (h2fwd && i == 0) {
      DISN_TRY(conv1_1_direct_launch(x, B, c.hw, c.hw, P(0), P(1), 1, t.act[0], t.amax + (size_t)B * 64, st, 64));
    } else if (h2fwd) {
      DISN_TRY(conv_h2_launch(x, B, c.hw, c.hw, c.cin, t.conv_h2img[i], P(2 * i + 1), c.cout, 1,
                              t.amax + (size_t)B * 64 * i, t.act[i], c.pool ? t.pooled[i] : nullptr,
                              t.amax + (size_t)B * 64 * (i + 1), st, 18, 64));
    } else {
      DISN_RC(conv_fwd(x, B, c.hw, c.hw, c.cin, t.conv_p[i], P(2 * i + 1), c.cout, 1, t.act[i], gws, gwb, st, bf));
    }
    x = t.act[i];
    if (c.pool) {
      if (!h2fwd) DISN_TRY(maxpool2x2_launch(x, B, c.hw, c.hw, c.cout, t.pooled[i], st));
      x = t.pooled[i];
    }
  }
  const float* pool5 = x;
  if (ctx) {
    DISN_TRY(hipEventRecord(ctx->ev[1], st));
    DISN_TRY(hipStreamWaitEvent(as, ctx->ev[1], 0));
  }
  DISN_TRY(gemv_launch(pool5, B, 25088, P(V_FC), P(V_FC + 1), 4096, 1, t.h6, t.fc_ws, as));
  DISN_TRY(gemv_launch(t.h6, B, 4096, P(V_FC + 2), P(V_FC + 3), 4096, 1, t.h7, t.fc_ws, as));
  DISN_TRY(gemv_launch(t.h7, B, 4096, P(V_FC + 4), P(V_FC + 5), DISN_EMBED_DIM, 0, t.emb, t.fc_ws, as));
  // folded global block: gbias[b] = emb[b] . W4[512:1536] + b4
  DISN_TRY(gemv_launch(t.emb, B, DISN_EMBED_DIM, P(V_G + 6) + (size_t)512 * 512, P(V_G + 7), 512, 0,
                       t.gbias, t.fc_ws, as));
  if (ctx) DISN_TRY(hipEventRecord(ctx->ev[2], as));
  DISN_TRY(project_launch(pts, trans_mat, B, N, t.xy, st));
  // rows E + F without the [B,137,137,1472] map (880 MB at B = 8): the five taps are up-sampled at the
  // pixels each point touches, bit-identical to resize -> gather; the backward needs xy and dfeat only
  {
    const float* tb[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 13; ++i)
      if (kConv[i].tap >= 0) tb[kConv[i].tap] = t.act[i];
    DISN_TRY(project_gather_taps_launch(tb, trans_mat, pts, B, N, 0, 5, t.feat, st));
  }
  DISN_TRY(pt_embed_launch(pts_rot, M, P(V_G), P(V_G + 1), P(V_L), P(V_L + 1), t.g1, t.l1, st));
  DISN_RC(dense_fwd(t.l1, 64, 64, nullptr, 0, 64, (int)M, t.l_p2, P(V_L + 3), 256, 1, t.l2, gws, gwb, st, bf));
  DISN_RC(dense_fwd(t.l2, 256, 256, nullptr, 0, 256, (int)M, t.l_p3, P(V_L + 5), 512, 1, t.l3, gws, gwb, st, bf));
  DISN_RC(dense_fwd(t.l3, 512, 512, t.feat, DISN_FEAT_DIM, 1984, (int)M, t.l_p4, P(V_L + 7), 512, 1, t.l4, gws, gwb, st, bf));
  DISN_RC(dense_fwd(t.l4, 512, 512, nullptr, 0, 512, (int)M, t.l_p5, P(V_L + 9), 256, 1, t.l5, gws, gwb, st, bf));
  DISN_RC(dense_fwd(t.g1, 64, 64, nullptr, 0, 64, (int)M, t.g_p2, P(V_G + 3), 256, 1, t.g2, gws, gwb, st, bf));
  DISN_RC(dense_fwd(t.g2, 256, 256, nullptr, 0, 256, (int)M, t.g_p3, P(V_G + 5), 512, 1, t.g3, gws, gwb, st, bf));
  if (ctx) DISN_TRY(hipStreamWaitEvent(st, ctx->ev[2], 0));  // embedding, gbias, regularization
  // the folded per-image term is the bias row of the sample: one launch for all samples on the bf16 / three-term kernels
  // (gemm_bf16_mfma.hip takes a bias row per rows_per_bias rows); the fp32-MFMA GEMM has one bias row: a launch per sample
  if (bf) {
    DISN_RC(dense_fwd(t.g3, 512, 512, nullptr, 0, 512, (int)M, t.g_p4, t.gbias, 512, 1, t.g4, gws, gwb, st, bf, N));
  } else {
    for (int b = 0; b < B; ++b) {
      const size_t o = (size_t)b * N * 512;
      DISN_RC(dense_fwd(t.g3 + o, 512, 512, nullptr, 0, 512, N, t.g_p4, t.gbias + (size_t)b * 512, 512, 1,
                        t.g4 + o, gws, gwb, st, bf));
    }
  }
  DISN_RC(dense_fwd(t.g4, 512, 512, nullptr, 0, 512, (int)M, t.g_p5, P(V_G + 9), 256, 1, t.g5, gws, gwb, st, bf));
  DISN_TRY(final_dot_launch(t.g5, t.l5, M, P(V_G + 10), P(V_G + 11), P(V_L + 10), P(V_L + 11), pred,
                            nullptr, nullptr, 1.0f, st));

  // ---------------- losses (losses[3] was written by the weight-norm pass above) ----------------
  DISN_TRY(loss_reduce_launch(pred, gt, M, sdf_weight, mask_weight, losses, st));
  DISN_TRY(loss_grad_launch(pred, gt, M, sdf_weight, mask_weight, t.dpred, st));

  // ---------------- backward: point MLPs ----------------
  for (int sidx = 0; sidx < 2; ++sidx) {
    const bool loc = sidx == 1;
    const int V = loc ? V_L : V_G;
    const float *h1 = loc ? t.l1 : t.g1, *h2 = loc ? t.l2 : t.g2, *h3 = loc ? t.l3 : t.g3;
    const float *h4 = loc ? t.l4 : t.g4, *h5 = loc ? t.l5 : t.g5;
    // fold2/conv5 + ReLU of fold2/conv2
    DISN_TRY(final_bwd_launch(h5, t.dpred, M, P(V + 10), t.d5, G(V + 10), G(V + 11), G(V + 9), wd,
                              s.red_ws, st));
    // fold2/conv2 (512 -> 256)
    DISN_RC(dense_bwd(h4, 512, 512, P(V + 8), t.d5, M, 256, wd, t.d4, G(V + 8), s, st, loc ? t.l_t5 : t.g_t5));
    DISN_TRY(relu_bwd_colsum_launch(t.d4, h4, M, 512, 1, G(V + 7), s.red_ws, st));
    // fold2/conv1: rows 0..511 of W multiply the point feature, the rest the image feature
    DISN_RC(dense_bwd(h3, 512, 512, P(V + 6), t.d4, M, 512, wd, t.d3, G(V + 6), s, st, loc ? t.l_t4 : t.g_t4));
    if (loc) {
      const float* wf = P(V + 6) + (size_t)512 * 512;
      DISN_RC(dense_bwd(t.feat, DISN_FEAT_DIM, DISN_FEAT_DIM, wf, t.d4, M, 512, wd, t.dfeat,
                        G(V + 6) + (size_t)512 * 512, s, st, t.l_t4f));
    } else {
      const float* wg = P(V + 6) + (size_t)512 * 512;  // [1024][512]
      DISN_TRY(image_colsum_launch(t.d4, B, N, 512, t.dgbias, s.red_ws, st));
      DISN_TRY(fc_bwd_launch(t.emb, t.dgbias, B, DISN_EMBED_DIM, 512, wg, wd, G(V + 6) + (size_t)512 * 512, nullptr,
                             t.demb, st));
    }
    DISN_TRY(relu_bwd_colsum_launch(t.d3, h3, M, 512, 1, G(V + 5), s.red_ws, st));
    // fold1/conv3 (256 -> 512), conv2 (64 -> 256), conv1 (3 -> 64)
    DISN_RC(dense_bwd(h2, 256, 256, P(V + 4), t.d3, M, 512, wd, t.d2, G(V + 4), s, st, loc ? t.l_t3 : t.g_t3));
    DISN_TRY(relu_bwd_colsum_launch(t.d2, h2, M, 256, 1, G(V + 3), s.red_ws, st));
    DISN_RC(dense_bwd(h1, 64, 64, P(V + 2), t.d2, M, 256, wd, t.d1, G(V + 2), s, st, loc ? t.l_t2 : t.g_t2));
    DISN_TRY(relu_bwd_colsum_launch(t.d1, h1, M, 64, 1, G(V + 1), s.red_ws, st));
    DISN_TRY(embed_bwd_launch(pts_rot, t.d1, M, G(V), P(V), wd, s.red_ws, st));
    if (!loc) {
      // ------------ backward: fc8, fc7, fc6 (needs only d(embedding) of the global stream) ------------
      // on the auxiliary stream, under the local stream's GEMMs
      if (ctx) {
        DISN_TRY(hipEventRecord(ctx->ev[3], st));
        DISN_TRY(hipStreamWaitEvent(as, ctx->ev[3], 0));
      }
      float* r2 = t.red_aux;
      DISN_TRY(relu_bwd_colsum_launch(t.demb, nullptr, B, DISN_EMBED_DIM, 0, G(V_FC + 5), r2, as));
      DISN_TRY(fc_bwd_launch(t.h7, t.demb, B, 4096, DISN_EMBED_DIM, P(V_FC + 4), wd, G(V_FC + 4), t.h7, t.dz7, as));
      DISN_TRY(relu_bwd_colsum_launch(t.dz7, nullptr, B, 4096, 0, G(V_FC + 3), r2, as));
      DISN_TRY(fc_bwd_launch(t.h6, t.dz7, B, 4096, 4096, P(V_FC + 2), wd, G(V_FC + 2), t.h6, t.dz6, as));
      DISN_TRY(relu_bwd_colsum_launch(t.dz6, nullptr, B, 4096, 0, G(V_FC + 1), r2, as));
      DISN_TRY(fc_bwd_launch(pool5, t.dz6, B, 25088, 4096, P(V_FC), wd, G(V_FC), nullptr, t.dpool5, as));
      if (ctx) DISN_TRY(hipEventRecord(ctx->ev[4], as));
    }
  }

  // ---------------- backward: image feature map ----------------
  DISN_TRY(hipMemsetAsync(t.dmap, 0, (size_t)B * 137 * 137 * DISN_FEAT_DIM * sizeof(float), st));
  DISN_TRY(gather_bwd_launch(t.dfeat, t.xy, B, N, t.dmap, st));
  if (ctx) DISN_TRY(hipStreamWaitEvent(st, ctx->ev[4], 0));
  // every gradient from offset[26] on (fc6..fc8 and both MLPs: 96 % of the bytes) is final here;
  // the caller may start reducing that part while the convolution backward below still runs
  if (head_ready_event) DISN_TRY(hipEventRecord((hipEvent_t)head_ready_event, st));

  // ---------------- backward: conv stack ----------------
  const float* dcur = t.dpool5;  // gradient w.r.t. the input of the layer above
  float* bufs[2] = {t.gA, t.gB};
  int which = 0;
  for (int i = 12; i >= 0; --i) {
    const ConvL& c = kConv[i];
    float* dy;
    if (c.pool) {
      dy = bufs[which];
      which ^= 1;
      DISN_TRY(maxpool_bwd_launch(t.act[i], dcur, B, c.hw, c.hw, c.cout, dy, st));
      // t.col ([B,224,224,64] floats, free until conv1_1) is the row-pass scratch (<= B*56*137*256)
      DISN_TRY(resize_bwd_launch(t.dmap, B, c.hw, c.hw, c.cout, DISN_IMG_H, DISN_IMG_W, DISN_FEAT_DIM,
                                 kTapOff[c.tap], dy, 1, t.col, st));
    } else {
      dy = const_cast<float*>(dcur);
    }
    const long rows = (long)B * c.hw * c.hw;
    float* amax_i = h2fwd && i > 0 ? t.amax_bwd + (size_t)i * B * 64 : nullptr;
    bool amax_ready = false;
    DISN_TRY(relu_bwd_colsum_launch(dy, t.act[i], rows, c.cout, 1, G(2 * i + 1), s.red_ws, st, amax_i,
                                    (long)c.hw * c.hw, &amax_ready));
    const float* xin = i == 0 ? t.resized : (kConv[i - 1].pool ? t.pooled[i - 1] : t.act[i - 1]);
    float* dx = nullptr;
    if (i > 0) {
      dx = bufs[which];
      if (dx == dy) dx = bufs[which ^ 1];
    }
    DISN_RC(conv_bwd(xin, B, c.hw, c.hw, c.cin, P(2 * i), dy, c.cout, wd, dx, G(2 * i), t.col, s, st,
                     t.conv_bT[i], h2fwd && i > 0 ? t.conv_h2bT[i] : nullptr, amax_i,
                     h2fwd && i > 0 ? t.amax + (size_t)B * 64 * i : nullptr, amax_ready));
    if (dx) {
      which = (dx == bufs[0]) ? 1 : 0;
      dcur = dx;
    }
  }
  return 0;
}

}  // extern "C"
