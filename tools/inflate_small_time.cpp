// Where the time goes when ogd_inflate.h decodes a few hundred bytes per stream (a page's contour members): the
// whole decode against zlib's, and build_table / pair_literals alone.  Host only:
//   g++ -O2 -std=c++17 -Iorigami_amd/csrc tools/inflate_small_time.cpp -lz -o /tmp/inflate_small_time
#include <chrono>
#include <cstdio>
#include <random>
#include <string>
#include <vector>
#include <zlib.h>
#include "ogd_inflate.h"

using namespace ogd::inflate;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  std::mt19937 rng(1);
  std::vector<std::string> plain;
  std::vector<std::vector<uint8_t>> packed;
  for (int k = 0; k < 400; ++k) {
    std::string s = "POLYGON ((";
    const int n = 8 + rng() % 20;
    for (int i = 0; i < n; ++i) {
      char b[64];
      snprintf(b, sizeof b, "%s%u %u", i ? ", " : "", (unsigned)(100 + rng() % 5900), (unsigned)(100 + rng() % 7900));
      s += b;
    }
    s += "))";
    z_stream z{};
    deflateInit2(&z, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    std::vector<uint8_t> out(s.size() + 64);
    z.next_in = (Bytef*)s.data(); z.avail_in = s.size(); z.next_out = out.data(); z.avail_out = out.size();
    deflate(&z, Z_FINISH);
    out.resize(z.total_out);
    deflateEnd(&z);
    plain.push_back(s);
    packed.push_back(out);
  }
  size_t in_b = 0, out_b = 0;
  for (int k = 0; k < 400; ++k) in_b += packed[k].size(), out_b += plain[k].size();
  printf("400 members, %zu -> %zu bytes, first byte %02x\n", in_b, out_b, packed[0][0]);
  std::vector<uint8_t> buf(4096);
  const int reps = 200;
  double t0 = now();
  for (int r = 0; r < reps; ++r)
    for (int k = 0; k < 400; ++k) {
      int64_t n_out = 0, n_used = 0;
      if (inflate_raw(packed[k].data(), packed[k].size(), buf.data(), plain[k].size() + 512, &n_out, &n_used) != OK || n_out != (int64_t)plain[k].size()) return 1;
    }
  printf("ogd inflate_raw : %.2f us per member\n", (now() - t0) / reps / 400 * 1e6);
  t0 = now();
  for (int r = 0; r < reps; ++r)
    for (int k = 0; k < 400; ++k) {
      z_stream z{};
      inflateInit2(&z, -15);
      z.next_in = packed[k].data(); z.avail_in = packed[k].size(); z.next_out = buf.data(); z.avail_out = buf.size();
      if (inflate(&z, Z_FINISH) != Z_STREAM_END) return 2;
      inflateEnd(&z);
    }
  printf("zlib inflate    : %.2f us per member\n", (now() - t0) / reps / 400 * 1e6);
  // table building alone, on a typical code: fixed-Huffman lengths
  uint8_t lens[288 + 32];
  for (int s = 0; s < 288; ++s) lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
  for (int s = 0; s < 32; ++s) lens[288 + s] = 5;
  static const Payloads P;
  Tables t;
  t0 = now();
  for (int r = 0; r < reps * 400; ++r) {
    build_table(lens, 288, P.lit, LIT_BITS, t.lit, (int)(sizeof t.lit / 4));
    asm volatile("" ::: "memory");
  }
  printf("build_table lit : %.2f us\n", (now() - t0) / reps / 400 * 1e6);
  t0 = now();
  for (int r = 0; r < reps * 400; ++r) {
    pair_literals(t.lit);
    asm volatile("" ::: "memory");
  }
  printf("pair_literals   : %.2f us (on an already paired table)\n", (now() - t0) / reps / 400 * 1e6);
  t0 = now();
  for (int r = 0; r < reps * 400; ++r) {
    build_table(lens + 288, 32, P.dist, DIST_BITS, t.dist, (int)(sizeof t.dist / 4));
    asm volatile("" ::: "memory");
  }
  printf("build_table dist: %.2f us\n", (now() - t0) / reps / 400 * 1e6);
  return 0;
}
