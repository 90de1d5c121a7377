// N4 — page scans stored as PNG: decoded on the host by libogdewarp's own inflate (ogd_inflate.h), host code.
// ref: origami/core/page.py:82-88 (PIL.Image.open -> the pixels every stage works on); SURVEY §8f N4.
//
// A PNG's pixels are one zlib stream cut into IDAT chunks, holding the rows of the image each behind a filter
// byte (PNG specification, "Filtering": None, Sub, Up, Average, Paeth).  Eight-bit grey and RGB images without
// interlacing are decoded here — what scanners and PIL write; everything else (palettes, 16 bits, alpha,
// Adam7) returns OGD_ERR_UNSUPPORTED and stays with PIL.  The zlib Adler-32 is verified; chunk CRCs are not
// (one checksum over the decoded bytes already answers "is this the image that was written").
#include "ogd_common.h"
#include "ogd_inflate.h"
#include <cstring>
#include <memory>
#include <new>
#include <thread>
#include <vector>

namespace ogd {
namespace {

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

struct PngHeader {
  int width = 0, height = 0, channels = 0;
  int64_t first_chunk = 0;  // offset of the chunk after IHDR
};

int read_header(const uint8_t* file, int64_t n, PngHeader& h) {
  static const uint8_t magic[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  OGD_REQUIRE(file != nullptr && n >= 8 + 25 && std::memcmp(file, magic, 8) == 0, "not a PNG file");
  OGD_REQUIRE(be32(file + 8) == 13 && std::memcmp(file + 12, "IHDR", 4) == 0, "PNG: IHDR chunk missing");
  const uint8_t* d = file + 16;
  const uint32_t w = be32(d), ht = be32(d + 4);
  const int depth = d[8], colour = d[9], compression = d[10], filter = d[11], interlace = d[12];
  OGD_REQUIRE(w >= 1 && ht >= 1 && w < (1u << 30) && ht < (1u << 30), "PNG: bad size %ux%u", w, ht);
  OGD_REQUIRE(compression == 0 && filter == 0, "PNG: unknown compression / filter method");
  if (depth != 8 || (colour != 0 && colour != 2) || interlace != 0)
    return fail(OGD_ERR_UNSUPPORTED, "PNG: bit depth %d, colour type %d, interlace %d is left to PIL", depth, colour, interlace);
  h.width = (int)w;
  h.height = (int)ht;
  h.channels = colour == 0 ? 1 : 3;
  h.first_chunk = 8 + 12 + 13;
  return OGD_OK;
}

inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// One row: f = the filtered bytes behind the filter byte `type`, prev = the finished row above (zeros for row 0).
template <int BPP>
inline bool unfilter_row(int type, const uint8_t* f, const uint8_t* prev, uint8_t* o, int64_t rowbytes) {
  switch (type) {
    case 0:
      std::memcpy(o, f, (size_t)rowbytes);
      return true;
    case 1: {
      uint8_t acc[BPP];
      for (int c = 0; c < BPP; ++c) acc[c] = 0;
      for (int64_t i = 0; i + BPP <= rowbytes; i += BPP)
        for (int c = 0; c < BPP; ++c) o[i + c] = acc[c] = (uint8_t)(acc[c] + f[i + c]);
      return true;
    }
    case 2:
      for (int64_t i = 0; i < rowbytes; ++i) o[i] = (uint8_t)(f[i] + prev[i]);
      return true;
    case 3:
      for (int64_t i = 0; i < BPP && i < rowbytes; ++i) o[i] = (uint8_t)(f[i] + (prev[i] >> 1));
      for (int64_t i = BPP; i < rowbytes; ++i) o[i] = (uint8_t)(f[i] + ((o[i - BPP] + prev[i]) >> 1));
      return true;
    case 4:
      for (int64_t i = 0; i < BPP && i < rowbytes; ++i) o[i] = (uint8_t)(f[i] + prev[i]);  // paeth(0, b, 0) = b
      for (int64_t i = BPP; i < rowbytes; ++i) o[i] = (uint8_t)(f[i] + paeth(o[i - BPP], prev[i], prev[i - BPP]));
      return true;
    default:
      return false;
  }
}

// The consumer of the inflate thread's output: rows are unfiltered (and summed for the Adler-32) as soon as the
// decoder has published them.  raw: height rows of (filter byte + rowbytes); out: rows of rowbytes at out_stride.
// Returns the row with an unknown filter type, or -1; stops early when `failed` is set.
template <int BPP>
int unfilter_behind(const uint8_t* raw, int height, int64_t rowbytes, uint8_t* out, int64_t out_stride,
                    const std::atomic<int64_t>& progress, const std::atomic<bool>& finished, uint32_t& adler) {
  std::vector<uint8_t> zero((size_t)rowbytes, 0);
  const uint8_t* prev = zero.data();
  int64_t ready = 0;
  adler = 1;
  for (int y = 0; y < height; ++y) {
    const int64_t need = (int64_t)(y + 1) * (rowbytes + 1);
    while (ready < need) {
      ready = progress.load(std::memory_order_acquire);
      if (ready >= need) break;
      if (finished.load(std::memory_order_acquire)) {
        ready = progress.load(std::memory_order_acquire);
        if (ready < need) return -2;  // the stream ended early (the decoder reports why)
        break;
      }
      std::this_thread::yield();
    }
    const uint8_t* f = raw + (int64_t)y * (rowbytes + 1);
    adler = inflate::adler32(adler, f, rowbytes + 1);
    uint8_t* o = out + (int64_t)y * out_stride;
    if (!unfilter_row<BPP>(f[0], f + 1, prev, o, rowbytes)) return y;
    prev = o;
  }
  return -1;
}

}  // namespace
}  // namespace ogd

using namespace ogd;

extern "C" {

int ogd_inflate_zlib(const uint8_t* in, int64_t n_in, uint8_t* out, int64_t cap, int64_t* n_out) {
  OGD_REQUIRE(in && out && n_out && n_in >= 0 && cap >= 0, "bad argument");
  const int st = inflate::inflate_zlib(in, n_in, out, cap, n_out);
  if (st == inflate::OK) return OGD_OK;
  return fail(OGD_ERR_INVALID, "zlib stream: %s", st == inflate::OUTPUT_FULL ? "more data than the output buffer holds"
                                                  : st == inflate::TRUNCATED ? "truncated" : "corrupt");
}

int ogd_png_info(const uint8_t* file, int64_t n, int* width, int* height, int* channels) {
  OGD_REQUIRE(width && height && channels, "null pointer");
  PngHeader h;
  OGD_TRY(read_header(file, n, h));
  *width = h.width;
  *height = h.height;
  *channels = h.channels;
  return OGD_OK;
}

int ogd_png_decode_u8(const uint8_t* file, int64_t n, uint8_t* out, int64_t out_stride) {
  PngHeader h;
  OGD_TRY(read_header(file, n, h));
  OGD_REQUIRE(out != nullptr, "null pointer");
  const int64_t rowbytes = (int64_t)h.width * h.channels;
  OGD_REQUIRE(out_stride >= rowbytes, "output stride %lld below the row size %lld", (long long)out_stride, (long long)rowbytes);
  // gather the IDAT payloads (usually many chunks of 8 .. 64 KB)
  std::vector<uint8_t> z;
  z.reserve((size_t)n);
  bool ended = false;
  for (int64_t pos = h.first_chunk; pos + 12 <= n;) {
    const uint32_t len = be32(file + pos);
    const uint8_t* type = file + pos + 4;
    OGD_REQUIRE(pos + 12 + (int64_t)len <= n, "PNG: chunk runs past the end of the file");
    if (std::memcmp(type, "IDAT", 4) == 0) {
      z.insert(z.end(), file + pos + 8, file + pos + 8 + len);
    } else if (std::memcmp(type, "IEND", 4) == 0) {
      ended = true;
      break;
    } else if (!(type[0] & 0x20)) {  // an unknown critical chunk (PLTE has no business in these colour types either)
      return fail(OGD_ERR_UNSUPPORTED, "PNG: critical chunk %.4s is left to PIL", (const char*)type);
    }
    pos += 12 + (int64_t)len;
  }
  OGD_REQUIRE(ended && !z.empty(), "PNG: no image data (IDAT / IEND missing)");
  const int64_t raw_bytes = (rowbytes + 1) * h.height;
  std::unique_ptr<uint8_t[]> raw(new (std::nothrow) uint8_t[(size_t)raw_bytes + 8]);  // not zeroed: 48 MB per page
  if (!raw) return fail(OGD_ERR_NOMEM, "out of host memory");
  // the stream is decoded on a second thread; this one follows it row by row (Adler-32 + unfiltering)
  int64_t got = 0;
  uint32_t stored = 0, summed = 1;
  std::atomic<int64_t> progress{0};
  std::atomic<bool> finished{false};
  int st = inflate::OK;
  std::thread decoder([&]() {
    st = inflate::inflate_zlib(z.data(), (int64_t)z.size(), raw.get(), raw_bytes, &got, false, &stored, &progress);
    finished.store(true, std::memory_order_release);
  });
  const int bad_row = h.channels == 1
      ? unfilter_behind<1>(raw.get(), h.height, rowbytes, out, out_stride, progress, finished, summed)
      : unfilter_behind<3>(raw.get(), h.height, rowbytes, out, out_stride, progress, finished, summed);
  decoder.join();
  if (st != inflate::OK || got != raw_bytes)
    return fail(OGD_ERR_INVALID, "PNG: image data %s", st == inflate::OUTPUT_FULL ? "longer than the header says"
                                                       : st == inflate::TRUNCATED ? "truncated"
                                                       : st == inflate::OK ? "shorter than the header says" : "corrupt");
  if (bad_row >= 0) return fail(OGD_ERR_INVALID, "PNG: unknown filter type %d in row %d", raw[(size_t)bad_row * (rowbytes + 1)], bad_row);
  OGD_REQUIRE(summed == stored, "PNG: image data corrupt (Adler-32 mismatch)");
  return OGD_OK;
}

}  // extern "C"
