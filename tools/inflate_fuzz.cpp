// Fuzz harness for origami_amd/csrc/ogd_inflate.h (host code only): valid zlib streams of several kinds are decoded
// and compared with zlib's own output, then damaged (flipped bytes, truncation, short output buffers) and decoded
// again — the decoder must answer with an error code or the right bytes, never touch memory outside its buffers.
//   g++ -O1 -g -fsanitize=address,undefined tools/inflate_fuzz.cpp -lz -o /tmp/inflate_fuzz && /tmp/inflate_fuzz [rounds]
// Input and output buffers are exact-size heap blocks, so AddressSanitizer sees any overrun by a single byte.
#include "../origami_amd/csrc/ogd_inflate.h"
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

using namespace ogd::inflate;

static std::vector<uint8_t> deflate_with(const std::vector<uint8_t>& raw, int level, int strategy) {
  z_stream s{};
  deflateInit2(&s, level, Z_DEFLATED, 15, 8, strategy);
  std::vector<uint8_t> out(deflateBound(&s, raw.size()) + 16);
  s.next_in = const_cast<Bytef*>(raw.data());
  s.avail_in = (uInt)raw.size();
  s.next_out = out.data();
  s.avail_out = (uInt)out.size();
  deflate(&s, Z_FINISH);
  out.resize(s.total_out);
  deflateEnd(&s);
  return out;
}

static int run(const std::vector<uint8_t>& z, size_t cap, std::vector<uint8_t>* got) {
  // exact-size heap copies: ASan guards both ends
  uint8_t* in = (uint8_t*)malloc(z.size() ? z.size() : 1);
  memcpy(in, z.data(), z.size());
  uint8_t* out = (uint8_t*)malloc(cap ? cap : 1);
  int64_t n = 0;
  const int st = inflate_zlib(in, (int64_t)z.size(), out, (int64_t)cap, &n);
  if (st == OK && got) got->assign(out, out + n);
  free(in);
  free(out);
  return st;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 200;
  std::mt19937 rng(12345);
  long good = 0, rejected = 0, survived = 0;
  for (int round = 0; round < rounds; ++round) {
    const int kind = round % 5;
    const size_t n = kind == 4 ? rng() % 40 : 1 + rng() % 200000;
    std::vector<uint8_t> raw(n);
    for (size_t i = 0; i < n; ++i) {
      switch (kind) {
        case 0: raw[i] = (uint8_t)rng(); break;                                  // incompressible
        case 1: raw[i] = (uint8_t)(230 + rng() % 8); break;                      // page background: short codes, pairs
        case 2: raw[i] = (uint8_t)("the quick brown fox "[i % 20]); break;       // long matches
        case 3: raw[i] = (uint8_t)((i / 700) % 2 ? rng() % 3 : 0); break;        // runs (distance 1) and noise
        default: raw[i] = (uint8_t)rng(); break;
      }
    }
    static const int strategies[] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE, Z_FILTERED};
    const std::vector<uint8_t> z = deflate_with(raw, (int)(rng() % 10), strategies[rng() % 5]);
    std::vector<uint8_t> got;
    if (run(z, raw.size(), &got) != OK || got != raw) {
      printf("round %d: a valid stream was not decoded\n", round);
      return 1;
    }
    ++good;
    if (raw.size() > 0 && run(z, raw.size() - 1, nullptr) != OUTPUT_FULL) {
      printf("round %d: a short output buffer was not reported\n", round);
      return 1;
    }
    for (int k = 0; k < 40; ++k) {
      std::vector<uint8_t> bad = z;
      const int how = (int)(rng() % 3);
      if (how == 0) {
        bad.resize(rng() % bad.size());
      } else {
        const int flips = 1 + (int)(rng() % 4);
        for (int f = 0; f < flips; ++f) bad[rng() % bad.size()] ^= (uint8_t)(1u << (rng() % 8));
      }
      const size_t cap = how == 2 ? rng() % (raw.size() + 300) : raw.size() + (rng() % 2 ? 0 : 70000);
      std::vector<uint8_t> out;
      const int st = run(bad, cap, &out);
      if (st == OK) {
        ++survived;  // the damage hit nothing that matters (Adler-32 agreed): the bytes must then be right
        if (out != raw) {
          printf("round %d: damaged stream accepted with different contents\n", round);
          return 1;
        }
      } else {
        ++rejected;
      }
    }
  }
  printf("inflate fuzz: %ld valid streams decoded, %ld damaged ones rejected, %ld damaged ones harmless\n", good, rejected, survived);
  return 0;
}
