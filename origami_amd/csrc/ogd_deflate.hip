// N4 — the deflate side of the stage's archives (contours.N.zip, dewarp.zip), host code: writing and reading.
// ref: origami/batch/core/io.py:516-518, 565-577 (zipfile.ZIP_DEFLATED members), origami/core/dewarp.py:545-553.
// The entropy coder is csrc/ogd_deflate.h; here: the C ABI around it, one member or all members of an archive in one
// call, with the CRC-32 a zip directory wants for each.
#include "ogd_common.h"
#include "ogd_deflate.h"
#include "ogd_inflate.h"

namespace ogd {
namespace {

// CRC-32 (IEEE 802.3, the zip / PNG polynomial), eight bytes per step
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

uint32_t crc32(const uint8_t* p, int64_t n) {
  static const CrcTables T;
  uint32_t c = 0xFFFFFFFFu;
  for (; n >= 8; n -= 8, p += 8) {
    uint32_t lo, hi;
    std::memcpy(&lo, p, 4);
    std::memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = T.t[7][lo & 0xff] ^ T.t[6][(lo >> 8) & 0xff] ^ T.t[5][(lo >> 16) & 0xff] ^ T.t[4][lo >> 24] ^
        T.t[3][hi & 0xff] ^ T.t[2][(hi >> 8) & 0xff] ^ T.t[1][(hi >> 16) & 0xff] ^ T.t[0][hi >> 24];
  }
  for (; n > 0; --n) c = (c >> 8) ^ T.t[0][(c ^ *p++) & 0xff];
  return ~c;
}

}  // namespace
}  // namespace ogd

using namespace ogd;

extern "C" {

int ogd_deflate_raw(const uint8_t* in, int64_t n, uint8_t* out, int64_t cap, int64_t* n_out) {
  OGD_REQUIRE(n >= 0 && (in || n == 0) && out && n_out && cap >= 0, "bad argument");
  const int64_t size = deflate::deflate_raw(in, n, out, cap);
  OGD_REQUIRE(size >= 0, "deflate: output buffer of %lld bytes too small for %lld bytes of input", (long long)cap, (long long)n);
  *n_out = size;
  return OGD_OK;
}

int ogd_deflate_many(const uint8_t* in, const int64_t* offsets, int n, uint8_t* out, int64_t cap,
                     int64_t* out_offsets, uint32_t* crcs) {
  OGD_REQUIRE(n >= 0 && offsets && out && out_offsets && crcs && cap >= 0, "bad argument");
  out_offsets[0] = 0;
  for (int k = 0; k < n; ++k) {
    const int64_t b = offsets[k], len = offsets[k + 1] - b;
    OGD_REQUIRE(len >= 0 && (in || len == 0), "offsets must be non-decreasing (member %d)", k);
    const int64_t at = out_offsets[k];
    const int64_t size = deflate::deflate_raw(in + b, len, out + at, cap - at);
    OGD_REQUIRE(size >= 0, "deflate: output buffer of %lld bytes too small (member %d)", (long long)cap, k);
    out_offsets[k + 1] = at + size;
    crcs[k] = crc32(in + b, len);
  }
  return OGD_OK;
}

// The reading side (ref: origami/batch/core/io.py:142-144, origami/core/dewarp.py:537-543 read the archives through
// zipfile): member k is the raw deflate stream buf[in_off[k] .. in_off[k] + in_len[k]) (method 8) or those bytes
// themselves (method 0); it goes to out[out_off[k] .. out_off[k + 1]), whose length is the size the archive's
// directory states, and its CRC-32 must be crcs[k].  One call per archive: the decoder is csrc/ogd_inflate.h.
int ogd_inflate_many(const uint8_t* buf, int64_t n_buf, const int64_t* in_off, const int64_t* in_len,
                     const int32_t* methods, int n, uint8_t* out, const int64_t* out_off, const uint32_t* crcs) {
  OGD_REQUIRE(n >= 0 && (n == 0 || (buf && in_off && in_len && methods && out_off && crcs)), "bad argument");
  for (int k = 0; k < n; ++k) {
    const int64_t b = in_off[k], len = in_len[k], size = out_off[k + 1] - out_off[k];
    OGD_REQUIRE(b >= 0 && len >= 0 && b + len <= n_buf && size >= 0 && (out || size == 0), "member %d lies outside its buffers", k);
    uint8_t* dst = out + out_off[k];
    if (methods[k] == 0) {
      OGD_REQUIRE(len == size, "stored member %d: %lld bytes for a size of %lld", k, (long long)len, (long long)size);
      std::memcpy(dst, buf + b, (size_t)size);
    } else {
      OGD_REQUIRE(methods[k] == 8, "member %d: compression method %d", k, methods[k]);
      int64_t n_out = 0, n_used = 0;
      const int st = inflate::inflate_raw(buf + b, len, dst, size, &n_out, &n_used);
      if (st != inflate::OK || n_out != size)
        return fail(OGD_ERR_INVALID, "member %d: deflate stream %s", k,
                    st == inflate::OUTPUT_FULL ? "holds more data than the directory states"
                    : st == inflate::TRUNCATED  ? "is truncated"
                    : st != inflate::OK         ? "is corrupt" : "holds less data than the directory states");
    }
    if (crc32(dst, size) != crcs[k]) return fail(OGD_ERR_INVALID, "member %d: bad CRC-32", k);
  }
  return OGD_OK;
}

}  // extern "C"
