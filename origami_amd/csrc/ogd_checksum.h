// CRC-32 (IEEE 802.3: zip members, PNG chunks) and Adler-32 (zlib streams) for the host side of libogdewarp.
// Portable forms (slicing by eight; two plain reductions per block) and, on x86-64 CPUs that have them, carry-less
// multiplication (PCLMULQDQ: folding four 128-bit lanes, V. Gopal et al., "Fast CRC Computation for Generic
// Polynomials Using PCLMULQDQ Instruction", Intel 2009, constants for the reflected polynomial 0xEDB88320) and
// SSSE3 multiply-add for the position-weighted sum — chosen at run time, same values either way
// (tests/test_deflate.py: against zlib over random lengths and alignments).
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define OGD_X86_SIMD 1
#else
#define OGD_X86_SIMD 0
#endif

namespace ogd {
namespace checksum {

struct CrcTables {
  uint32_t t[8][256];
  CrcTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int k = 1; k < 8; ++k) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xff];
  }
};

// c: the running register (0xFFFFFFFF at the start, inverted at the end), eight bytes per step
inline uint32_t crc32_tables(uint32_t c, const uint8_t* p, int64_t n) {
  static const CrcTables T;
  for (; n >= 8; n -= 8, p += 8) {
    uint32_t lo, hi;
    std::memcpy(&lo, p, 4);
    std::memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = T.t[7][lo & 0xff] ^ T.t[6][(lo >> 8) & 0xff] ^ T.t[5][(lo >> 16) & 0xff] ^ T.t[4][lo >> 24] ^
        T.t[3][hi & 0xff] ^ T.t[2][(hi >> 8) & 0xff] ^ T.t[1][(hi >> 16) & 0xff] ^ T.t[0][hi >> 24];
  }
  for (; n > 0; --n) c = (c >> 8) ^ T.t[0][(c ^ *p++) & 0xff];
  return c;
}

#if OGD_X86_SIMD
// n >= 64 and a multiple of 16
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_clmul(uint32_t c, const uint8_t* p, int64_t n) {
  alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};  // x^(4*128+32), x^(4*128-32) mod P
  alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};  // x^(128+32), x^(128-32) mod P
  alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};  // x^64 mod P
  alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};  // P, floor(x^64 / P)
  __m128i x1 = _mm_loadu_si128((const __m128i*)(p + 0x00)), x2 = _mm_loadu_si128((const __m128i*)(p + 0x10));
  __m128i x3 = _mm_loadu_si128((const __m128i*)(p + 0x20)), x4 = _mm_loadu_si128((const __m128i*)(p + 0x30));
  x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)c));
  __m128i x0 = _mm_load_si128((const __m128i*)k1k2);
  p += 64;
  n -= 64;
  while (n >= 64) {  // four lanes folded over 64 bytes each
    __m128i x5 = _mm_clmulepi64_si128(x1, x0, 0x00), x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
    __m128i x7 = _mm_clmulepi64_si128(x3, x0, 0x00), x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
    x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
    x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), _mm_loadu_si128((const __m128i*)(p + 0x00)));
    x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), _mm_loadu_si128((const __m128i*)(p + 0x10)));
    x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), _mm_loadu_si128((const __m128i*)(p + 0x20)));
    x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), _mm_loadu_si128((const __m128i*)(p + 0x30)));
    p += 64;
    n -= 64;
  }
  x0 = _mm_load_si128((const __m128i*)k3k4);  // the four lanes into one
  __m128i x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
  while (n >= 16) {  // 16 bytes at a time
    x2 = _mm_loadu_si128((const __m128i*)p);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    p += 16;
    n -= 16;
  }
  // 128 -> 64 bits
  x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
  const __m128i x3m = _mm_setr_epi32(~0, 0, ~0, 0);
  x1 = _mm_srli_si128(x1, 8);
  x1 = _mm_xor_si128(x1, x2);
  x0 = _mm_loadl_epi64((const __m128i*)k5k0);
  x2 = _mm_srli_si128(x1, 4);
  x1 = _mm_and_si128(x1, x3m);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  // Barrett reduction to 32 bits
  x0 = _mm_load_si128((const __m128i*)poly);
  x2 = _mm_and_si128(x1, x3m);
  x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
  x2 = _mm_and_si128(x2, x3m);
  x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  return (uint32_t)_mm_extract_epi32(x1, 1);
}

inline bool cpu_has_clmul() {
  static const bool has = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
  return has;
}
inline bool cpu_has_ssse3() {
  static const bool has = __builtin_cpu_supports("ssse3");
  return has;
}

// sums of a block of n bytes, n a multiple of 16 and at most 5536: *sum = sum of the bytes, *wsum = sum of
// (n - i) * p[i]
__attribute__((target("ssse3"))) inline void adler_block_ssse3(const uint8_t* p, uint32_t n, uint32_t* sum, uint64_t* wsum) {
  const __m128i taps = _mm_setr_epi8(16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1);
  const __m128i zero = _mm_setzero_si128(), ones = _mm_set1_epi16(1);
  __m128i v_s = zero;   // byte sums so far (two 64-bit halves)
  __m128i v_ps = zero;  // sum over the groups of the byte sums BEFORE the group
  __m128i v_t = zero;   // sum over the groups of sum (16 - k) * b_k (four 32-bit lanes)
  for (uint32_t i = 0; i < n; i += 16) {
    const __m128i b = _mm_loadu_si128((const __m128i*)(p + i));
    v_ps = _mm_add_epi64(v_ps, v_s);
    v_s = _mm_add_epi64(v_s, _mm_sad_epu8(b, zero));
    v_t = _mm_add_epi32(v_t, _mm_madd_epi16(_mm_maddubs_epi16(b, taps), ones));
  }
  alignas(16) uint64_t s[2], ps[2];
  alignas(16) uint32_t t[4];
  _mm_store_si128((__m128i*)s, v_s);
  _mm_store_si128((__m128i*)ps, v_ps);
  _mm_store_si128((__m128i*)t, v_t);
  *sum = (uint32_t)(s[0] + s[1]);
  // byte k of group g (G groups) weighs n - 16 g - k = (16 - k) + 16 (G - 1 - g); and the sum over the groups of
  // (G - 1 - g) S_g equals the sum over the groups of the byte sums in front of them (every S_g' is counted once
  // for each later group), which is what v_ps collected
  *wsum = (uint64_t)t[0] + t[1] + t[2] + t[3] + 16 * (ps[0] + ps[1]);
}
#endif

inline uint32_t crc32_update(uint32_t c, const uint8_t* p, int64_t n) {
#if OGD_X86_SIMD
  if (n >= 64 && cpu_has_clmul()) {
    const int64_t m = n & ~(int64_t)15;
    c = crc32_clmul(c, p, m);
    p += m;
    n -= m;
  }
#endif
  return crc32_tables(c, p, n);
}

inline uint32_t crc32(const uint8_t* p, int64_t n) { return ~crc32_update(0xFFFFFFFFu, p, n); }

// Adler-32 continued from `adler` (1 for a fresh sum).
inline uint32_t adler32(uint32_t adler, const uint8_t* p, int64_t n) {
  uint32_t a = adler & 0xffff, b = adler >> 16;
#if OGD_X86_SIMD
  if (cpu_has_ssse3()) {
    while (n >= 16) {
      const uint32_t k = (uint32_t)(n < 5536 ? (n & ~(int64_t)15) : 5536);
      uint32_t sum;
      uint64_t wsum;
      adler_block_ssse3(p, k, &sum, &wsum);
      b = (uint32_t)((b + (uint64_t)k * a + wsum) % 65521);
      a = (a + sum) % 65521;
      p += k;
      n -= k;
    }
  }
#endif
  while (n > 0) {  // blocks of 4096 bytes as two plain reductions: 255 * 4096 * 4097 / 2 + 4096 * 65520 < 2^32
    const uint32_t k = (uint32_t)(n < 4096 ? n : 4096);
    uint32_t sum = 0, wsum = 0;
    for (uint32_t i = 0; i < k; ++i) {
      sum += p[i];
      wsum += (k - i) * p[i];
    }
    b = (b + k * a + wsum) % 65521;
    a = (a + sum) % 65521;
    p += k;
    n -= k;
  }
  return (b << 16) | a;
}

}  // namespace checksum
}  // namespace ogd
