// Fuzz harness for origami_amd/csrc/ogd_deflate.h and ogd_checksum.h (host code only): inputs of many kinds and sizes
// are coded by deflate_raw into exact-size heap buffers of every interesting capacity (too small by a byte, by a
// word, just enough, the documented bound) and the streams that come out are decoded by zlib and compared; the
// checksums are compared with zlib's over random lengths and alignments.  The writer must return -1 or a stream
// zlib reads back, never touch memory outside its buffers.
//   g++ -O1 -g -fsanitize=address,undefined tools/deflate_fuzz.cpp -lz -o /tmp/deflate_fuzz && /tmp/deflate_fuzz [rounds]
#include "../origami_amd/csrc/ogd_deflate.h"
#include "../origami_amd/csrc/ogd_checksum.h"
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

static bool inflate_raw_zlib(const uint8_t* z, size_t zn, std::vector<uint8_t>& out, size_t want) {
  z_stream s{};
  if (inflateInit2(&s, -15) != Z_OK) return false;
  out.assign(want + 16, 0);
  s.next_in = const_cast<Bytef*>(z);
  s.avail_in = (uInt)zn;
  s.next_out = out.data();
  s.avail_out = (uInt)out.size();
  const int st = inflate(&s, Z_FINISH);
  const size_t got = s.total_out;
  const bool all_in = s.avail_in == 0;
  inflateEnd(&s);
  out.resize(got);
  return st == Z_STREAM_END && all_in;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 300;
  std::mt19937_64 rng(12345);
  long streams = 0, refused = 0, sums = 0;
  for (int r = 0; r < rounds; ++r) {
    // sizes around the block length and small ones
    static const int64_t sizes[] = {0, 1, 2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 255, 256, 257, 1000, 4095, 65534, 65535, 65536, 65537, 131070, 131071, 200000};
    int64_t n = r < 24 ? sizes[r] : (int64_t)(rng() % (r % 7 == 0 ? 300000 : 5000));
    std::vector<uint8_t> raw((size_t)n);
    const int kind = r % 6;
    for (int64_t i = 0; i < n; ++i) {
      switch (kind) {
        case 0: raw[i] = (uint8_t)rng(); break;                                        // noise: every symbol, long codes
        case 1: raw[i] = (uint8_t)(128 + (int)(rng() % 9) - 4); break;                 // a scanned line: few grey levels
        case 2: raw[i] = 0; break;                                                     // one symbol
        case 3: raw[i] = (uint8_t)(rng() % 2 ? 255 : 0); break;                        // two symbols
        case 4: raw[i] = (uint8_t)((rng() % 1000) ? 200 : rng()); break;               // one common symbol, rare others (deep tree)
        default: raw[i] = (uint8_t)(i < n / 2 ? rng() : 7); break;                     // statistics change inside the input
      }
    }
    const int64_t bound = n + n / 1000 + 64;
    int64_t full = -1;
    {
      uint8_t* in = (uint8_t*)malloc(n ? n : 1);
      if (n) memcpy(in, raw.data(), (size_t)n);
      uint8_t* out = (uint8_t*)malloc((size_t)bound);
      full = ogd::deflate::deflate_raw(in, n, out, bound);
      if (full < 0) { printf("FAIL: bound %lld refused for n=%lld kind %d\n", (long long)bound, (long long)n, kind); return 1; }
      std::vector<uint8_t> back;
      if (!inflate_raw_zlib(out, (size_t)full, back, (size_t)n) || back != raw) { printf("FAIL: zlib does not read the stream back, n=%lld kind %d\n", (long long)n, kind); return 1; }
      ++streams;
      free(out);
      // every capacity near the stream's size, and a few far below it
      const int64_t caps[] = {0, 1, 3, 4, 7, 8, full / 2, full - 9, full - 8, full - 5, full - 4, full - 3, full - 2, full - 1, full, full + 1, full + 3, full + 4, full + 8, full + 23, full + 24, full + 25};
      for (int64_t cap : caps) {
        if (cap < 0) continue;
        uint8_t* o = (uint8_t*)malloc(cap ? (size_t)cap : 1);
        const int64_t got = ogd::deflate::deflate_raw(in, n, o, cap);
        if (got >= 0) {
          if (got > cap) { printf("FAIL: wrote %lld into %lld\n", (long long)got, (long long)cap); return 1; }
          if (!inflate_raw_zlib(o, (size_t)got, back, (size_t)n) || back != raw) { printf("FAIL: stream of capacity %lld unreadable, n=%lld kind %d\n", (long long)cap, (long long)n, kind); return 1; }
          ++streams;
        } else {
          ++refused;
          if (cap >= bound) { printf("FAIL: capacity %lld >= bound refused\n", (long long)cap); return 1; }
        }
        free(o);
      }
      free(in);
    }
    // checksums at random offsets of an exact-size block
    for (int k = 0; k < 8 && n > 0; ++k) {
      const size_t a = rng() % (size_t)n, len = rng() % ((size_t)n - a + 1);
      uint8_t* blk = (uint8_t*)malloc(len ? len : 1);
      if (len) memcpy(blk, raw.data() + a, len);
      const uint32_t c0 = (uint32_t)crc32(0, blk, (uInt)len), a0 = (uint32_t)adler32(1, blk, (uInt)len);
      if (ogd::checksum::crc32(blk, (int64_t)len) != c0) { printf("FAIL: crc32 len %zu\n", len); return 1; }
      if (ogd::checksum::adler32(1u, blk, (int64_t)len) != a0) { printf("FAIL: adler32 len %zu\n", len); return 1; }
      free(blk);
      sums += 2;
    }
  }
  printf("ok: %ld streams read back by zlib, %ld short buffers refused, %ld checksums equal zlib's\n", streams, refused, sums);
  return 0;
}
