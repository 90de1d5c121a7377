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
#include <algorithm>
#include <chrono>
#include <cstring>
#include <functional>
#include <memory>
#include <new>
#include <system_error>
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
  // PIL warns above MAX_IMAGE_PIXELS (89 478 485) and raises DecompressionBombError above twice that: leave such
  // headers (a hostile IHDR can claim 2^30 x 2^30) to PIL and its error, before anything is sized from them
  if ((uint64_t)w * ht > 2ull * 89478485ull)
    return fail(OGD_ERR_UNSUPPORTED, "PNG: %ux%u pixels is left to PIL (beyond its decompression-bomb limit)", w, ht);
  h.width = (int)w;
  h.height = (int)ht;
  h.channels = colour == 0 ? 1 : 3;
  h.first_chunk = 8 + 12 + 13;
  return OGD_OK;
}

// The Paeth predictor without branches (page noise makes its comparisons unpredictable): a = left, b = up,
// c = up-left; the distances |p - a|, |p - b|, |p - c| of p = a + b - c are |b - c|, |a - c|, |a + b - 2c|.
inline int paeth(int a, int b, int c) {
  const int db = b - c, da = a - c;
  int pa = db < 0 ? -db : db;
  const int pb = da < 0 ? -da : da;
  const int s = db + da;
  const int pc = s < 0 ? -s : s;
  int r = a;
  r = pb < pa ? b : r;
  pa = pb < pa ? pb : pa;
  r = pc < pa ? c : r;
  return r;
}

// Bytes [b, e) of one row (b, e multiples of BPP or the row's end): f = the filtered bytes behind the filter byte
// `type` (0 .. 4), prev = the finished row above (zeros for row 0).  carry (Average / Paeth, rows done in pieces):
// the left and up-left pixel of position b on entry, of position e on return.
template <int BPP>
inline void unfilter_row(int type, const uint8_t* f, const uint8_t* prev, uint8_t* o, int64_t rowbytes, int64_t b, int64_t e,
                         int* carry = nullptr) {
  (void)rowbytes;
  switch (type) {
    case 0:
      std::memcpy(o + b, f + b, (size_t)(e - b));
      return;
    case 1: {  // whole rows only (no dependence on the row above)
      uint8_t acc[BPP];
      for (int c = 0; c < BPP; ++c) acc[c] = 0;
      for (int64_t i = b; i + BPP <= e; i += BPP)
        for (int c = 0; c < BPP; ++c) o[i + c] = acc[c] = (uint8_t)(acc[c] + f[i + c]);
      return;
    }
    case 2:
      for (int64_t i = b; i < e; ++i) o[i] = (uint8_t)(f[i] + prev[i]);
      return;
    // Average and Paeth feed every output byte into the next pixel: the running left (and up-left) values are kept in
    // locals, so that the chain does not go through memory (o, prev and f may alias as far as the compiler knows)
    case 3: {
      int left[BPP];
      for (int c = 0; c < BPP; ++c) left[c] = carry ? carry[c] : 0;
      for (int64_t i = b; i + BPP <= e; i += BPP)
        for (int c = 0; c < BPP; ++c) {
          left[c] = (f[i + c] + ((left[c] + prev[i + c]) >> 1)) & 255;
          o[i + c] = (uint8_t)left[c];
        }
      if (carry)
        for (int c = 0; c < BPP; ++c) carry[c] = left[c];
      return;
    }
    default: {
      int left[BPP], upleft[BPP];
      for (int c = 0; c < BPP; ++c) {
        left[c] = carry ? carry[c] : 0;
        upleft[c] = carry ? carry[BPP + c] : 0;
      }
      for (int64_t i = b; i + BPP <= e; i += BPP)
        for (int c = 0; c < BPP; ++c) {
          const int up = prev[i + c];
          left[c] = (f[i + c] + paeth(left[c], up, upleft[c])) & 255;
          upleft[c] = up;
          o[i + c] = (uint8_t)left[c];
        }
      if (carry)
        for (int c = 0; c < BPP; ++c) {
          carry[c] = left[c];
          carry[BPP + c] = upleft[c];
        }
      return;
    }
  }
}

// Unfiltering behind the inflate thread, rows dealt round-robin to `nworkers` threads.  Up, Average and Paeth rows
// need the finished row above, byte for byte, so a worker publishes how far its row has got (done[y], in chunks)
// and the worker on the next row stays one chunk behind it: a skewed wavefront over the rows.  Dependencies only
// point upwards and every worker takes its rows top to bottom, so nobody waits on a row that waits on him.
// Scanned pages with correlated noise are mostly Paeth rows — a serial ~14-operation chain per byte that no
// single thread gets through faster than the inflate thread delivers.
struct UnfilterShared {
  const uint8_t* raw;
  int height;
  int64_t rowbytes;
  uint8_t* out;
  int64_t out_stride;
  const std::atomic<int64_t>* progress;  // bytes of raw that are final
  const std::atomic<bool>* finished;     // the decoder has returned
  const uint8_t* zero_row;               // rowbytes zeros: the row above row 0 (allocated before any thread starts)
  std::atomic<int64_t>* done;            // [height] bytes of output row y that are final
  std::atomic<int> bad_row{-1};          // first row with an unknown filter type (-2: the stream ended early)
};

// The followers are faster than the decoder and mostly wait for it; a block of the stream takes it ~50 us.  They
// sleep instead of spinning: a worker process often has a CPU quota, and cycles burnt in a spin are the decoder's.
inline void wait_for_decoder() { std::this_thread::sleep_for(std::chrono::microseconds(40)); }

constexpr int64_t UNFILTER_CHUNK = 1020;  // whole pixels of 1, 2, 3 or 4 bytes

template <int BPP>
void unfilter_worker(UnfilterShared& sh, int worker, int nworkers) {
  int64_t ready = 0;
  for (int y = worker; y < sh.height; y += nworkers) {
    if (sh.bad_row.load(std::memory_order_relaxed) != -1) return;
    const int64_t need = (int64_t)(y + 1) * (sh.rowbytes + 1);
    while (ready < need) {
      ready = sh.progress->load(std::memory_order_acquire);
      if (ready >= need) break;
      if (sh.finished->load(std::memory_order_acquire)) {
        ready = sh.progress->load(std::memory_order_acquire);
        if (ready < need) {
          sh.bad_row.store(-2);
          return;
        }
        break;
      }
      if (sh.bad_row.load(std::memory_order_relaxed) != -1) return;
      wait_for_decoder();
    }
    const uint8_t* f = sh.raw + (int64_t)y * (sh.rowbytes + 1);
    const int type = f[0];
    ++f;
    uint8_t* o = sh.out + (int64_t)y * sh.out_stride;
    const uint8_t* prev = y ? sh.out + (int64_t)(y - 1) * sh.out_stride : sh.zero_row;
    if (type > 4) {
      sh.bad_row.store(y);
      // rows below wait for this one: let them through, the result is discarded anyway
      sh.done[y].store(sh.rowbytes, std::memory_order_release);
      return;
    }
    if (type < 2 || y == 0) {  // None / Sub, or nothing above: the whole row at once
      unfilter_row<BPP>(type, f, prev, o, sh.rowbytes, 0, sh.rowbytes);
      sh.done[y].store(sh.rowbytes, std::memory_order_release);
      continue;
    }
    int carry[2 * BPP];  // left / up-left of the chunk boundary
    for (int c = 0; c < 2 * BPP; ++c) carry[c] = 0;
    for (int64_t b = 0; b < sh.rowbytes; b += UNFILTER_CHUNK) {
      const int64_t e = std::min(sh.rowbytes, b + UNFILTER_CHUNK);
      while (sh.done[y - 1].load(std::memory_order_acquire) < e) {
        if (sh.bad_row.load(std::memory_order_relaxed) != -1) return;
        std::this_thread::yield();
      }
      unfilter_row<BPP>(type, f, prev, o, sh.rowbytes, b, e, carry);
      sh.done[y].store(e, std::memory_order_release);
    }
  }
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

static int png_decode_u8(const uint8_t* file, int64_t n, uint8_t* out, int64_t out_stride) {
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
  // the stream is decoded on one thread, two more unfilter the rows behind it, this one sums the Adler-32
  int64_t got = 0;
  uint32_t stored = 0, summed = 1;
  std::atomic<int64_t> progress{0};
  std::atomic<bool> finished{false};
  int st = inflate::OK;
  std::unique_ptr<std::atomic<int64_t>[]> done(new (std::nothrow) std::atomic<int64_t>[(size_t)h.height]);
  if (!done) return fail(OGD_ERR_NOMEM, "out of host memory");
  for (int y = 0; y < h.height; ++y) done[y].store(0, std::memory_order_relaxed);
  const std::vector<uint8_t> zero_row((size_t)rowbytes, 0);
  UnfilterShared sh;
  sh.raw = raw.get();
  sh.zero_row = zero_row.data();
  sh.height = h.height;
  sh.rowbytes = rowbytes;
  sh.out = out;
  sh.out_stride = out_stride;
  sh.progress = &progress;
  sh.finished = &finished;
  sh.done = done.get();
  auto decode = [&]() {
    st = inflate::inflate_zlib(z.data(), (int64_t)z.size(), raw.get(), raw_bytes, &got, false, &stored, &progress);
    finished.store(true, std::memory_order_release);
  };
  auto unfilter = [&](int k, int of) {
    if (h.channels == 1) unfilter_worker<1>(sh, k, of);
    else unfilter_worker<3>(sh, k, of);
  };
  constexpr int NW = 2;
  std::thread decoder, workers[NW];
  bool threaded = true;
  try {
    decoder = std::thread(decode);
    for (int k = 0; k < NW; ++k) workers[k] = std::thread(unfilter, k, NW);
  } catch (const std::system_error&) {  // no more threads to be had: stop what did start, then do it all here
    threaded = false;
    sh.bad_row.store(-3);
    for (auto& w : workers)
      if (w.joinable()) w.join();
    if (decoder.joinable()) decoder.join();
    else decode();
    sh.bad_row.store(-1);
    for (int y = 0; y < h.height; ++y) done[y].store(0, std::memory_order_relaxed);
    unfilter(0, 1);
  }
  for (int64_t at = 0; at < raw_bytes;) {  // Adler-32 of the filtered rows, as far as they are final
    const int64_t ready = progress.load(std::memory_order_acquire);
    if (ready > at) {
      summed = inflate::adler32(summed, raw.get() + at, ready - at);
      at = ready;
    } else if (finished.load(std::memory_order_acquire)) {
      if (progress.load(std::memory_order_acquire) <= at) break;
    } else {
      wait_for_decoder();
    }
  }
  if (threaded) {
    decoder.join();
    for (auto& w : workers) w.join();
  }
  const int bad_row = sh.bad_row.load();
  if (st != inflate::OK || got != raw_bytes)
    return fail(OGD_ERR_INVALID, "PNG: image data %s", st == inflate::OUTPUT_FULL ? "longer than the header says"
                                                       : st == inflate::TRUNCATED ? "truncated"
                                                       : st == inflate::OK ? "shorter than the header says" : "corrupt");
  if (bad_row >= 0) return fail(OGD_ERR_INVALID, "PNG: unknown filter type %d in row %d", raw[(size_t)bad_row * (rowbytes + 1)], bad_row);
  OGD_REQUIRE(summed == stored, "PNG: image data corrupt (Adler-32 mismatch)");
  return OGD_OK;
}

int ogd_png_decode_u8(const uint8_t* file, int64_t n, uint8_t* out, int64_t out_stride) {
  try {  // no C++ exception may cross the C ABI (std::vector growth in the caller's thread)
    return png_decode_u8(file, n, out, out_stride);
  } catch (const std::bad_alloc&) {
    return fail(OGD_ERR_NOMEM, "PNG: out of host memory");
  } catch (const std::exception& e) {
    return fail(OGD_ERR_INVALID, "PNG: %s", e.what());
  }
}

}  // extern "C"
