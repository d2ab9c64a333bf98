// Internal launcher declarations shared by the .hip translation units.
// Public C ABI: include/disn_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace disn {

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

// max_ws bounds the stream-K partial slabs the plan may use (the plan is a pure speed choice)
GemmPlan gemm_plan(int M, int N, int K, size_t max_ws = ~size_t(0));
hipError_t gemm_launch(const GemmParams& p, GemmMode mode, const GemmPlan& plan, float* ws,
                       hipStream_t st);
hipError_t pack_kn_launch(const float* w, int K, int N, int Kpad, float* packed, hipStream_t st);
// out[m][n] = act(sum_s ws[s][m][n] + bias[row(m)][n])
hipError_t splitk_reduce_launch(const float* ws, int S, int M, int N, const float* bias,
                                int rows_per_bias, int relu, float* out, int ldc, hipStream_t st);

// ---- gemv.hip ------------------------------------------------------------
int gemv_splits(int K, int N);
size_t gemv_ws_bytes(int B, int K, int N);
// out[b][n] = act(sum_k x[b][k] W[k][n] + bias[n]); N % 256 == 0
hipError_t gemv_launch(const float* x, int B, int K, const float* w_kn, const float* bias, int N,
                       int relu, float* out, float* ws, hipStream_t st);

// ---- elementwise.hip (compiled with -ffp-contract=off) --------------------
// max_blocks > 0 caps the grid (grid-stride): a background launch that should trickle under
// MFMA-bound work instead of flooding the CUs
hipError_t resize_bilinear_launch(const float* in, int B, int Hin, int Win, int C, float* out,
                                  int Hout, int Wout, int out_cstride, int out_coff,
                                  hipStream_t st, int max_blocks = 0);
hipError_t maxpool2x2_launch(const float* in, int B, int H, int W, int C, float* out,
                             hipStream_t st);
hipError_t project_launch(const float* pts, const float* trans_mat, int B, int N, float* xy,
                          hipStream_t st);
hipError_t gather_launch(const float* featmap, const float* xy, int B, int N, float* feat,
                         hipStream_t st);
// project + gather for a chunk of ONE image (points are a slice of image b's points)
hipError_t project_gather_launch(const float* featmap_b, const float* trans_mat_b, const float* pts,
                                 int n, float* feat, hipStream_t st);
struct GridSpec {
  double start[3], step[3], stop[3];
  int res;  // R+1
};
hipError_t grid_points_launch(const GridSpec& g, int64_t k0, int64_t k1, float* pts,
                              hipStream_t st);
hipError_t scale_div_launch(const float* in, float divisor, int64_t n, float* out, hipStream_t st);

// ---- marching_cubes.hip (compiled with -ffp-contract=off) ------------------------
size_t mc_ws_bytes(int R);
// counts[0] = vertices, counts[1] = triangles (device memory); fills ws for mc_emit_launch
hipError_t mc_count_launch(const float* vol, int R, float iso, unsigned long long* counts, void* ws,
                           hipStream_t st);
hipError_t mc_emit_launch(const float* vol, const GridSpec& g, float iso, float* verts, int* faces,
                          void* ws, hipStream_t st);

// ---- mlp_small.hip ---------------------------------------------------------
// relu(p . W1 + b1) for both streams: pts [M][3] -> out_g [M][64], out_l [M][64]
hipError_t pt_embed_launch(const float* pts, int64_t M, const float* g_w1, const float* g_b1,
                           const float* l_w1, const float* l_b1, float* out_g, float* out_l,
                           hipStream_t st);
// sdf[m] = (g5[m].g_w6 + g_b6) + (l5[m].l_w6 + l_b6); optional separate outputs
hipError_t final_dot_launch(const float* g5, const float* l5, int64_t M, const float* g_w6,
                            const float* g_b6, const float* l_w6, const float* l_b6, float* sdf,
                            float* sdf_g, float* sdf_l, float out_div, hipStream_t st);

}  // namespace disn
