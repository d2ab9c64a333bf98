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
