// Shared by conv_h2.hip and dense_h2.hip: fragment vector types, the power-of-two scale of the two-term f16 split,
// and the hardware-row <-> logical-row map that makes the A-fragment ds_read_b128 conflict-free (see conv_h2.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace disn {

typedef _Float16 ch_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 ch_h4 __attribute__((ext_vector_type(4)));
typedef float ch_f16v __attribute__((ext_vector_type(16)));

namespace ch2 {
// power of two s with amax * s in [2^target, 2^(target+1)); 1 for amax == 0 / non-finite / extreme
__host__ __device__ inline float pow2_scale(float amax, int target_exp) {
  union { float f; unsigned u; } a;
  a.f = amax;
  const int e = (int)((a.u >> 23) & 0xffu) - 127;
  if (!(amax > 0.f) || e > 100 || e < -100) return 1.0f;
  a.u = (unsigned)(127 + target_exp - e) << 23;
  return a.f;
}
// logical row of hardware row i (0..31) of a 32-row block
__host__ __device__ constexpr int sigma(int i) {
  return i < 4 ? i : (i < 12 ? i + 12 : (i < 16 ? i - 8 : (i < 20 ? i + 8 : (i < 28 ? i - 12 : i))));
}
// first logical row of the four held by accumulator quad q (registers 4q..4q+3) of lane half g
__host__ __device__ constexpr int quad_row(int q, int g) {
  return sigma(8 * q + 4 * g);
}
}  // namespace ch2

// one 3x3 SAME convolution launch of conv_h2.hip / conv_h2w.hip
struct ConvH2Dev {
  const float* in;            // [B][H][W][Cin]
  const unsigned char* wimg;  // conv_h2_pack image
  const float* bias;          // [Cout]
  const float* in_amax;       // 64 floats whose maximum is max |in|
  float* out;                 // [B][H][W][Cout]
  float* pool_out;            // [B][H/2][W/2][Cout] or nullptr
  float* out_amax;            // 64 floats (zeroed by the caller): atomic max |out| spread over the slots, or nullptr
  int B, H, W, Cin, Cout;
  int tiles_x, tiles_y;
  int relu;
  int amax_stride;            // floats between the slot groups of consecutive images (0: one group for the whole batch)
  int img_major;              // tile order [image][n-tile][patch] instead of [n-tile][image][patch] (see the launcher)
  long long* stamps;  // tuning builds: 16 clock stamps per workgroup (nullptr in the product)
};

}  // namespace disn
