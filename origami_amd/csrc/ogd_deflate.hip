// N4 — the deflate side of the stage's archives (contours.N.zip, dewarp.zip), host code: writing and reading.
// ref: origami/batch/core/io.py:516-518, 565-577 (zipfile.ZIP_DEFLATED members), origami/core/dewarp.py:545-553.
// The entropy coder is csrc/ogd_deflate.h; here: the C ABI around it, one member or all members of an archive in one
// call, with the CRC-32 a zip directory wants for each.
#include "ogd_common.h"
#include "ogd_deflate.h"
#include "ogd_inflate.h"
#include "ogd_lawson.h"  // parallel_ranges
#include <atomic>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace ogd {
namespace {

// CRC-32 (IEEE 802.3, the zip / PNG polynomial): ogd_checksum.h (carry-less multiplication where the CPU has it)
inline uint32_t crc32(const uint8_t* p, int64_t n) { return checksum::crc32(p, n); }

inline void put_be32(uint8_t* p, uint32_t v) {
  p[0] = (uint8_t)(v >> 24), p[1] = (uint8_t)(v >> 16), p[2] = (uint8_t)(v >> 8), p[3] = (uint8_t)v;
}

inline int paeth_pred(int a, int b, int c) {
  const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Sum of |residual| (residuals read as signed bytes) that filter f leaves on a row — the measure of the PNG
// specification's filter heuristic (12.8) and of libpng's default.  prev: the row above, nullptr for the first row.
template <int F>
int64_t png_filter_cost_f(const uint8_t* row, const uint8_t* prev, int nb, int bpp) {
  int64_t sum = 0;
  const int head = std::min(bpp, nb);
  for (int i = 0; i < head; ++i) {  // no left neighbour
    const int up = prev ? prev[i] : 0;
    const uint8_t r = (uint8_t)(row[i] - (F == 2 || F == 4 ? up : F == 3 ? up >> 1 : 0));
    sum += r < 128 ? r : 256 - r;
  }
  if (prev || F < 2) {
    const uint8_t* pv = prev ? prev : row;  // (not read for F < 2)
    int32_t acc = 0;
    for (int i = head; i < nb; ++i) {
      const int x = row[i], a = row[i - bpp], up = pv[i], c = pv[i - bpp];
      const int pred = F == 0 ? 0 : F == 1 ? a : F == 2 ? up : F == 3 ? (a + up) >> 1 : paeth_pred(a, up, c);
      const int r = (uint8_t)(x - pred);
      acc += r < 128 ? r : 256 - r;
    }
    sum += acc;
  } else {  // first row: Up is None, Average halves the left neighbour, Paeth is Sub
    int32_t acc = 0;
    for (int i = head; i < nb; ++i) {
      const int x = row[i], a = row[i - bpp];
      const int r = (uint8_t)(x - (F == 2 ? 0 : F == 3 ? a >> 1 : a));
      acc += r < 128 ? r : 256 - r;
    }
    sum += acc;
  }
  return sum;
}

int64_t png_filter_cost(int f, const uint8_t* row, const uint8_t* prev, int nb, int bpp) {
  switch (f) {
    case 0: return png_filter_cost_f<0>(row, prev, nb, bpp);
    case 1: return png_filter_cost_f<1>(row, prev, nb, bpp);
    case 2: return png_filter_cost_f<2>(row, prev, nb, bpp);
    case 3: return png_filter_cost_f<3>(row, prev, nb, bpp);
    default: return png_filter_cost_f<4>(row, prev, nb, bpp);
  }
}

// dst = filter byte f + the nb filtered bytes of a row.  One plain loop per filter: the inputs of a byte are source
// bytes only, nothing carried from byte to byte, so the compiler vectorises them (Paeth included).
void png_filter_row(int f, const uint8_t* row, const uint8_t* prev, int nb, int bpp, uint8_t* dst) {
  const int head = std::min(bpp, nb);
  if (!prev && f == 2) f = 0;  // the same rows
  if (!prev && f == 4) f = 1;
  dst[0] = (uint8_t)f;
  uint8_t* o = dst + 1;
  switch (f) {
    case 0: std::memcpy(o, row, (size_t)nb); break;
    case 1:
      for (int i = 0; i < head; ++i) o[i] = row[i];
      for (int i = head; i < nb; ++i) o[i] = (uint8_t)(row[i] - row[i - bpp]);
      break;
    case 2:
      for (int i = 0; i < nb; ++i) o[i] = (uint8_t)(row[i] - prev[i]);
      break;
    case 3:
      for (int i = 0; i < head; ++i) o[i] = (uint8_t)(row[i] - ((prev ? prev[i] : 0) >> 1));
      if (prev)
        for (int i = head; i < nb; ++i) o[i] = (uint8_t)(row[i] - ((row[i - bpp] + prev[i]) >> 1));
      else
        for (int i = head; i < nb; ++i) o[i] = (uint8_t)(row[i] - (row[i - bpp] >> 1));
      break;
    default:
      for (int i = 0; i < head; ++i) o[i] = (uint8_t)(row[i] - prev[i]);  // paeth(0, b, 0) = b
      for (int i = head; i < nb; ++i) {
        const int a = row[i - bpp], b = prev[i], c = prev[i - bpp];
        const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
        const int pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
        o[i] = (uint8_t)(row[i] - pred);
      }
      break;
  }
}

// Bytes a PNG file of an H x W x C image needs at most here (signature, IHDR, one IDAT, IEND around a zlib stream).
inline int64_t png_bound(int64_t H, int64_t W, int C) {
  const int64_t raw = H * (W * C + 1);
  return 8 + 25 + 12 + 12 + 2 + 4 + raw + raw / 1000 + 64;
}

// The PNG file of img ([H][W][C] uint8, C = 1 grey or 3 RGB, contiguous); returns its size, -1 if cap is too small.
int64_t png_encode_one(const uint8_t* img, int H, int W, int C, uint8_t* out, int64_t cap, std::vector<uint8_t>& scratch) {
  const int nb = W * C;
  const int64_t raw = (int64_t)H * (nb + 1);
  if (cap < png_bound(H, W, C)) return -1;
  scratch.resize((size_t)raw);
  uint8_t* filt = scratch.data();
  // The filter is chosen per image from up to eight of its rows (cost of the five filters summed over them), not per
  // row: the rows of a scanned line of text are of one kind, and costing every filter on every row was most of the
  // writer's time.
  int best = 0;
  {
    int64_t cost[5] = {0, 0, 0, 0, 0};
    const int probes = std::min(H, 8);
    for (int k = 0; k < probes; ++k) {
      const int y = H <= 8 ? k : (int)(((int64_t)k * (H - 1)) / (probes - 1));
      for (int f = 0; f < 5; ++f)
        cost[f] += png_filter_cost(f, img + (size_t)y * nb, y ? img + (size_t)(y - 1) * nb : nullptr, nb, C);
    }
    for (int f = 1; f < 5; ++f)
      if (cost[f] < cost[best]) best = f;
  }
  for (int y = 0; y < H; ++y)
    png_filter_row(best, img + (size_t)y * nb, y ? img + (size_t)(y - 1) * nb : nullptr, nb, C, filt + (size_t)y * (nb + 1));
  static const uint8_t magic[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  uint8_t* p = out;
  std::memcpy(p, magic, 8);
  p += 8;
  put_be32(p, 13);
  std::memcpy(p + 4, "IHDR", 4);
  put_be32(p + 8, (uint32_t)W);
  put_be32(p + 12, (uint32_t)H);
  p[16] = 8, p[17] = (uint8_t)(C == 1 ? 0 : 2), p[18] = 0, p[19] = 0, p[20] = 0;
  put_be32(p + 21, crc32(p + 4, 17));
  p += 25;
  uint8_t* idat = p;
  std::memcpy(p + 4, "IDAT", 4);
  p[8] = 0x78, p[9] = 0x01;  // zlib header: deflate, 32 K window, no dictionary, fastest
  const int64_t size = deflate::deflate_raw(filt, raw, p + 10, cap - (p + 10 - out) - 4 - 4 - 12);
  if (size < 0) return -1;
  put_be32(p + 10 + size, inflate::adler32(1, filt, raw));
  const int64_t len = 2 + size + 4;
  put_be32(idat, (uint32_t)len);
  put_be32(idat + 8 + len, crc32(idat + 4, 4 + len));
  p = idat + 12 + len;
  put_be32(p, 0);
  std::memcpy(p + 4, "IEND", 4);
  put_be32(p + 8, crc32(p + 4, 4));
  return (p + 12) - out;
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

// PNG files of many small images in one call: the line images of a page (ref: origami/batch/export/lines.py:62-67,
// `im.save(f, format='png', optimize=True)` per line — zlib at level 9, 13 ms per 48 x 3000 line image and 96 % of
// the export stage's time once the images themselves come from one kernel launch).  Rows filtered by the minimum
// sum of absolute residuals, then the entropy coder of ogd_deflate.h (no string matching: scanned text is noise
// around a few grey levels, its filtered rows hold next to no repeats).  Image k: [heights[k]][widths[k]][channels]
// contiguous at images[k]; its file goes to out + out_starts[k] (room: out_starts[k + 1] - out_starts[k], at least
// ogd_png_bound of the image), out_sizes[k] receives the file's size and out_crcs[k] (may be null) its CRC-32, which
// a zip archive that stores the file wants.  Images are independent: up to OGD_THREADS threads.
int64_t ogd_png_bound(int height, int width, int channels) { return png_bound(height, width, channels); }

int ogd_png_encode_many(const uint8_t* const* images, const int32_t* widths, const int32_t* heights,
                        int channels, int n, uint8_t* out, const int64_t* out_starts, int64_t* out_sizes,
                        uint32_t* out_crcs) {
  OGD_REQUIRE(n >= 0 && (n == 0 || (images && widths && heights && out && out_starts && out_sizes)), "bad argument");
  OGD_REQUIRE(channels == 1 || channels == 3, "PNG writer: %d channels (1 or 3)", channels);
  for (int k = 0; k < n; ++k) {
    OGD_REQUIRE(widths[k] >= 1 && heights[k] >= 1 && widths[k] < (1 << 24) && heights[k] < (1 << 24), "image %d: bad size %dx%d", k, widths[k], heights[k]);
    OGD_REQUIRE(images[k] != nullptr, "image %d: null pointer", k);
    OGD_REQUIRE(out_starts[k + 1] - out_starts[k] >= png_bound(heights[k], widths[k], channels), "image %d: output room too small", k);
  }
  std::atomic<int> failed{0};
  parallel_ranges(n, 4, [&](int64_t b, int64_t e) {
    std::vector<uint8_t> scratch;
    for (int64_t k = b; k < e; ++k) {
      const int64_t size = png_encode_one(images[k], heights[k], widths[k], channels, out + out_starts[k],
                                          out_starts[k + 1] - out_starts[k], scratch);
      if (size < 0) failed.store(1);
      out_sizes[k] = size;
      if (out_crcs && size >= 0) out_crcs[k] = crc32(out + out_starts[k], size);
    }
  });
  OGD_REQUIRE(!failed.load(), "PNG writer: an image did not fit its room");
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
