// Host-side helpers of the C ABI (no device code).
//   disn_crc32c -- CRC-32C (Castagnoli) as used by TensorFlow's table / tensor-bundle files
//                  (disn_amd/tf_checkpoint.py); SSE4.2 crc32 instruction, table fallback.
#include "../../include/disn_amd.h"

#include <cstdint>
#include <cstring>

#if defined(__SSE4_2__)
#include <nmmintrin.h>
#endif

namespace {
uint32_t crc_table[256];
bool table_ready = false;
void make_table() {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
    crc_table[i] = c;
  }
  table_ready = true;
}
}  // namespace

extern "C" uint32_t disn_crc32c(const void* data, size_t n, uint32_t crc) {
  const unsigned char* p = static_cast<const unsigned char*>(data);
  uint32_t c = crc ^ 0xFFFFFFFFu;
#if defined(__SSE4_2__)
  uint64_t c64 = c;
  while (n >= 8) {
    uint64_t v;
    std::memcpy(&v, p, 8);
    c64 = _mm_crc32_u64(c64, v);
    p += 8;
    n -= 8;
  }
  c = static_cast<uint32_t>(c64);
  while (n--) c = _mm_crc32_u8(c, *p++);
#else
  if (!table_ready) make_table();
  while (n--) c = crc_table[(c ^ *p++) & 0xFF] ^ (c >> 8);
#endif
  return c ^ 0xFFFFFFFFu;
}

// Wavefront .obj: "v x y z" (9 significant digits: float32 round-trips) and "f a b c" (1-based).
// Formats into a large buffer; ~1 M lines/s.
#include <cstdio>
#include <string>

extern "C" int disn_write_obj(const char* path, const float* verts, int64_t nv, const int32_t* faces,
                              int64_t nf) {
  if (!path || (nv > 0 && !verts) || (nf > 0 && !faces) || nv < 0 || nf < 0) return DISN_E_ARG;
  std::FILE* f = std::fopen(path, "wb");
  if (!f) return DISN_E_ARG;
  std::string buf;
  buf.reserve(1 << 22);
  char line[128];
  bool ok = true;
  auto flush = [&]() {
    if (!buf.empty()) ok = ok && std::fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    buf.clear();
  };
  for (int64_t i = 0; i < nv; ++i) {
    const int n = std::snprintf(line, sizeof line, "v %.9g %.9g %.9g\n", verts[3 * i], verts[3 * i + 1],
                                verts[3 * i + 2]);
    buf.append(line, n);
    if (buf.size() > (1u << 22) - 256) flush();
  }
  for (int64_t i = 0; i < nf; ++i) {
    const int n = std::snprintf(line, sizeof line, "f %d %d %d\n", faces[3 * i] + 1, faces[3 * i + 1] + 1,
                                faces[3 * i + 2] + 1);
    buf.append(line, n);
    if (buf.size() > (1u << 22) - 256) flush();
  }
  flush();
  ok = (std::fclose(f) == 0) && ok;
  return ok ? 0 : DISN_E_ARG;
}

// ---------------------------------------------------------------------------------------------------
// disn_equalise_weights (ABI 9): exact power-of-two re-parametrisation of the hidden channels of an
// INFERENCE copy of the weights (models/model_normalization.py:74-78,171-204; models/sdfnet.py:71-88,173-186).
// Hidden channel f of producer layer l gets the factor c_l[f] = 2^e(M_l) / 2^e(max(m_f, 2^-16 M_l)), m_f = the largest
// |entry| of column f of W_l AFTER its rows were divided by the factors of ITS producer, M_l = the MEDIAN of the m_f:
// the column and its bias are multiplied by c_l[f] (<= 2^16; < 1 for columns above the median), every row a consumer
// reads from that channel is divided by it.  ReLU and max-pool are positively homogeneous, resize / resampler / the products are linear, powers of two are
// exact in fp32: the network function is unchanged and so is every fp32 rounding on the way (barring under/overflow);
// what changes is that all channels of a hidden tensor now carry comparable magnitudes, which is what the per-image
// power-of-two ACTIVATION scale of the two-term f16 kernels (conv_h2 / conv_h2w / dense_h2 / dense_h2w; 22 bits down
// to 2^-17 of the tensor's maximum) needs when training left gains of 2^20 and more between channels.
// ---------------------------------------------------------------------------------------------------
#include <algorithm>
#include <cmath>
#include <vector>

namespace {
const int kEqConvCin[13] = {3, 64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512};
const int kEqConvCout[13] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};
const int kEqTapOfConv[13] = {-1, 0, -1, 1, -1, -1, 2, -1, -1, 3, -1, -1, 4};
const int kEqTapOff[5] = {0, 64, 192, 448, 960};
const int kEqMlpCin[6] = {3, 64, 256, 0 /* 1536 or 1984 */, 512, 256};
const int kEqMlpCout[6] = {64, 256, 512, 512, 256, 1};

inline int exp_of(float v) {  // floor(log2(v)) of a positive normal float
  int e;
  std::frexp(v, &e);
  return e - 1;
}

// column factors of W [rows][cols] (already row-scaled): c[f] as above; returns log2 of the largest factor
float eq_factors(const float* w, const float* bias, long rows, int cols, std::vector<float>& c) {
  std::vector<float> m(cols, 0.f);
  for (long r = 0; r < rows; ++r) {
    const float* p = w + r * cols;
    for (int f = 0; f < cols; ++f) {
      const float a = std::fabs(p[f]);
      if (a > m[f]) m[f] = a;
    }
  }
  // reference magnitude: the MEDIAN column maximum (of the columns that are not all zero) -- outlier columns are
  // scaled DOWN to it without limit, small columns up by at most 2^16 (a nearly dead column with an ordinary bias
  // must not become the tensor's maximum through b c)
  std::vector<float> nz;
  nz.reserve(cols);
  for (int f = 0; f < cols; ++f)
    if (m[f] > 0.f && std::isfinite(m[f])) nz.push_back(m[f]);
  c.assign(cols, 1.0f);
  if (nz.empty()) return 0.f;
  std::nth_element(nz.begin(), nz.begin() + nz.size() / 2, nz.end());
  const float med = nz[nz.size() / 2];
  if (med < 1e-30f) return 0.f;
  const int eM = exp_of(med);
  // an UP-scaled column takes its bias along: b c must stay within the layer's largest bias (a nearly dead
  // column with an ordinary bias, scaled up by 2^16, would set the image's activation scale -- the other channels of
  // the two-term f16 kernels would lose ~13 bits to it)
  float bmax = 0.f;
  for (int f = 0; f < cols; ++f)
    if (std::isfinite(bias[f])) bmax = std::fmax(bmax, std::fabs(bias[f]));
  int dmin = 0, dmax = 0;
  for (int f = 0; f < cols; ++f) {
    if (!std::isfinite(m[f])) continue;
    const float mf = std::fmax(m[f], std::ldexp(med, -16));
    int d = eM - exp_of(mf);  // <= 16; negative for columns above the median
    if (d < -100) d = -100;
    while (d > 0 && std::fabs(bias[f]) * std::ldexp(1.0f, d) > bmax) --d;
    c[f] = std::ldexp(1.0f, d);
    dmin = d < dmin ? d : dmin;
    dmax = d > dmax ? d : dmax;
  }
  return (float)(dmax - dmin);
}

void scale_cols(float* w, long rows, int cols, const std::vector<float>& c) {
  for (long r = 0; r < rows; ++r) {
    float* p = w + r * cols;
    for (int f = 0; f < cols; ++f) p[f] *= c[f];
  }
}

// rows of W [outer][nrows_total][cols]: rows r0 .. r0 + c.size() - 1 of every outer slice divided by c
void unscale_rows(float* w, long outer, long nrows_total, int cols, long r0, const std::vector<float>& c) {
  for (long o = 0; o < outer; ++o)
    for (size_t k = 0; k < c.size(); ++k) {
      if (c[k] == 1.0f) continue;
      const float inv = 1.0f / c[k];
      float* p = w + ((o * nrows_total) + r0 + (long)k) * cols;
      for (int f = 0; f < cols; ++f) p[f] *= inv;
    }
}
}  // namespace

extern "C" int disn_equalise_weights(const disn_eq_weights_t* w, float* tap_scale, float* span_log2) {
  if (!w || !tap_scale || w->num_classes <= 0) return DISN_E_ARG;
  for (int i = 0; i < 13; ++i)
    if (!w->conv_w[i] || !w->conv_b[i]) return DISN_E_ARG;
  for (int s = 0; s < 2; ++s)
    for (int l = 0; l < 6; ++l)
      if (!w->mlp_w[s][l] || !w->mlp_b[s][l]) return DISN_E_ARG;
  std::vector<float> c;
  for (int i = 0; i < 13; ++i) {
    const int ci = kEqConvCin[i], co = kEqConvCout[i];
    const float span = eq_factors(w->conv_w[i], w->conv_b[i], 9L * ci, co, c);
    if (span_log2) span_log2[i] = span;
    scale_cols(w->conv_w[i], 9L * ci, co, c);
    for (int f = 0; f < co; ++f) w->conv_b[i][f] *= c[f];
    if (i + 1 < 13) unscale_rows(w->conv_w[i + 1], 9, kEqConvCin[i + 1], kEqConvCout[i + 1], 0, c);
    else if (w->fc6_w) unscale_rows(w->fc6_w, 49, 512, 4096, 0, c);
    const int tap = kEqTapOfConv[i];
    if (tap >= 0) {
      unscale_rows(w->mlp_w[1][3], 1, 1984, 512, 512 + kEqTapOff[tap], c);
      for (int f = 0; f < co; ++f) tap_scale[kEqTapOff[tap] + f] = c[f];
    }
  }
  for (int s = 0; s < 2; ++s) {
    const int k4 = s == 0 ? 512 + w->num_classes : 1984;
    for (int l = 0; l < 5; ++l) {   // fold1/conv1 .. fold2/conv2 produce hidden channels; fold2/conv5 is the output
      const int ci = l == 3 ? k4 : kEqMlpCin[l], co = kEqMlpCout[l];
      const float span = eq_factors(w->mlp_w[s][l], w->mlp_b[s][l], ci, co, c);
      if (span_log2) span_log2[13 + 5 * s + l] = span;
      scale_cols(w->mlp_w[s][l], ci, co, c);
      for (int f = 0; f < co; ++f) w->mlp_b[s][l][f] *= c[f];
      const int cn = l + 1 == 3 ? k4 : kEqMlpCin[l + 1];
      unscale_rows(w->mlp_w[s][l + 1], 1, cn, kEqMlpCout[l + 1], 0, c);
    }
  }
  return 0;
}
