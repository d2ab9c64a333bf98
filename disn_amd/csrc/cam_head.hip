// Camera head: models/posenet.py:91-124 get_cam_mat on the 1024-d VGG embedding and
// cam_est/model_cam.py:102-103 pred_trans_mat = pred_RT @ K^T -- the estimated `trans_mat` input of
// the SDF path.  0.9 M MACs and 2.9 MB of weights per image: one workgroup per image runs the three
// towers layer by layer through LDS (thread = output column, weight rows read coalesced), thread 0
// finishes with the 6-D -> rotation Gram-Schmidt (models/posenet.py:22-36).  Launch-latency bound.
#include "../../include/disn_amd.h"

#include "kernels.hpp"

namespace disn {

struct CamK {
  float k[9];
};

__device__ __forceinline__ void cam_normalize(float* v) {
  float mag = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  mag = fmaxf(mag, 1e-8f);
  v[0] /= mag; v[1] /= mag; v[2] /= mag;
}

__device__ __forceinline__ void cam_cross(const float* a, const float* b, float* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

__global__ __launch_bounds__(256) void cam_head_kernel(const disn_cam_weights_t w,
                                                       const float* __restrict__ embedding, CamK K,
                                                       float* __restrict__ rotation,
                                                       float* __restrict__ translation,
                                                       float* __restrict__ RT,
                                                       float* __restrict__ trans_mat) {
  __shared__ float emb[1024], h1[704], h2[352], o3[10];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int k = tid; k < 1024; k += 256) emb[k] = embedding[(size_t)b * 1024 + k];
  __syncthreads();
  // layer 1: [scale 64 | ortho6d 512 | translation 128] columns, K = 1024, ReLU
  for (int n = tid; n < 704; n += 256) {
    const float* W;
    const float* bias;
    int ld, c;
    if (n < 64) { W = w.s_w1; bias = w.s_b1; ld = 64; c = n; }
    else if (n < 576) { W = w.r_w1; bias = w.r_b1; ld = 512; c = n - 64; }
    else { W = w.t_w1; bias = w.t_b1; ld = 128; c = n - 576; }
    float acc = 0.f;
    for (int k = 0; k < 1024; ++k) acc += emb[k] * W[(size_t)k * ld + c];
    h1[n] = fmaxf(acc + bias[c], 0.f);
  }
  __syncthreads();
  // layer 2: 64 -> 32, 512 -> 256, 128 -> 64, ReLU
  for (int n = tid; n < 352; n += 256) {
    const float *W, *bias, *x;
    int ld, c, kin;
    if (n < 32) { W = w.s_w2; bias = w.s_b2; ld = 32; c = n; x = h1; kin = 64; }
    else if (n < 288) { W = w.r_w2; bias = w.r_b2; ld = 256; c = n - 32; x = h1 + 64; kin = 512; }
    else { W = w.t_w2; bias = w.t_b2; ld = 64; c = n - 288; x = h1 + 576; kin = 128; }
    float acc = 0.f;
    for (int k = 0; k < kin; ++k) acc += x[k] * W[(size_t)k * ld + c];
    h2[n] = fmaxf(acc + bias[c], 0.f);
  }
  __syncthreads();
  // layer 3 (linear): 32 -> 1, 256 -> 6, 64 -> 3
  if (tid < 10) {
    const float *W, *bias, *x;
    int ld, c, kin;
    if (tid < 1) { W = w.s_w3; bias = w.s_b3; ld = 1; c = 0; x = h2; kin = 32; }
    else if (tid < 7) { W = w.r_w3; bias = w.r_b3; ld = 6; c = tid - 1; x = h2 + 32; kin = 256; }
    else { W = w.t_w3; bias = w.t_b3; ld = 3; c = tid - 7; x = h2 + 288; kin = 64; }
    float acc = 0.f;
    for (int k = 0; k < kin; ++k) acc += x[k] * W[(size_t)k * ld + c];
    o3[tid] = acc + bias[c];
  }
  __syncthreads();
  if (tid == 0) {
    const float s = o3[0];
    float x[3] = {o3[1], o3[2], o3[3]}, yr[3] = {o3[4], o3[5], o3[6]}, z[3], y[3];
    cam_normalize(x);
    cam_cross(x, yr, z);
    cam_normalize(z);
    cam_cross(z, x, y);
    // rotation matrix columns (x, y, z), scaled by s (pred_scale = s * I3 on the left)
    float R[4][3];
    for (int i = 0; i < 3; ++i) { R[i][0] = s * x[i]; R[i][1] = s * y[i]; R[i][2] = s * z[i]; }
    R[3][0] = o3[7] + -0.00193892f; R[3][1] = o3[8] + 0.00169222f; R[3][2] = o3[9] + 1.3949631f;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) rotation[(size_t)b * 9 + i * 3 + j] = R[i][j];
    for (int j = 0; j < 3; ++j) translation[(size_t)b * 3 + j] = R[3][j];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 3; ++j) {
        RT[(size_t)b * 12 + i * 3 + j] = R[i][j];
        // (RT @ K^T)[i][j] = sum_c RT[i][c] * K[j][c]
        float t = R[i][0] * K.k[j * 3 + 0];
        t += R[i][1] * K.k[j * 3 + 1];
        t += R[i][2] * K.k[j * 3 + 2];
        trans_mat[(size_t)b * 12 + i * 3 + j] = t;
      }
  }
}

}  // namespace disn

extern "C" int disn_cam_head(const disn_cam_weights_t* w, const float* embedding, const float* K_host,
                             int B, float* rotation, float* translation, float* RT, float* trans_mat,
                             void* stream) {
  if (!w || !embedding || !rotation || !translation || !RT || !trans_mat || B <= 0) return DISN_E_ARG;
  const float* const* p = reinterpret_cast<const float* const*>(w);
  for (size_t i = 0; i < sizeof(disn_cam_weights_t) / sizeof(const float*); ++i)
    if (!p[i]) return DISN_E_ARG;
  disn::CamK K;
  const float kd[9] = {149.84375f, 0.f, 68.5f, 0.f, 149.84375f, 68.5f, 0.f, 0.f, 1.f};  // model_cam.py:28
  for (int i = 0; i < 9; ++i) K.k[i] = K_host ? K_host[i] : kd[i];
  hipLaunchKernelGGL(disn::cam_head_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, *w, embedding, K,
                     rotation, translation, RT, trans_mat);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
