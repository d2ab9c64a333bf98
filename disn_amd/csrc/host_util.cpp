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
