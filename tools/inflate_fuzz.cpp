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

// ---- hand-made dynamic blocks with incomplete code sets: the verdict must be zlib's --------------------------
struct BitWriter {
  std::vector<uint8_t> out;
  uint32_t acc = 0;
  int n = 0;
  void bits(uint32_t v, int len) {  // LSB first (header fields, extra bits)
    for (int i = 0; i < len; ++i) {
      acc |= ((v >> i) & 1u) << n;
      if (++n == 8) out.push_back((uint8_t)acc), acc = 0, n = 0;
    }
  }
  void code(uint32_t c, int len) {  // Huffman codes go MSB first
    for (int i = len - 1; i >= 0; --i) bits((c >> i) & 1u, 1);
  }
  void flush() {
    if (n) out.push_back((uint8_t)acc), acc = 0, n = 0;
  }
};

static void canonical(const std::vector<int>& lens, std::vector<uint32_t>& codes) {
  int count[16] = {0};
  for (int l : lens) ++count[l];
  count[0] = 0;
  uint32_t next[16] = {0}, c = 0;
  for (int len = 1; len <= 15; ++len) next[len] = c = (c + (uint32_t)count[len - 1]) << 1;
  codes.assign(lens.size(), 0);
  for (size_t s = 0; s < lens.size(); ++s)
    if (lens[s]) codes[s] = next[lens[s]]++;
}

// One final dynamic block in a zlib wrapper: code lengths sent one by one through the code-length code `pre`
// (symbols 0..15 only), then `data` = (symbol, is_distance) pairs without extra bits; `expect` = the bytes it decodes to.
static std::vector<uint8_t> dynamic_block(const std::vector<int>& pre, const std::vector<int>& lit, const std::vector<int>& dist,
                                          const std::vector<std::pair<int, bool>>& data, const std::vector<uint8_t>& expect) {
  static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  BitWriter w;
  w.bits(0x78, 8);
  w.bits(0x01, 8);
  w.bits(1, 1);
  w.bits(2, 2);
  w.bits((uint32_t)lit.size() - 257, 5);
  w.bits((uint32_t)dist.size() - 1, 5);
  w.bits(19 - 4, 4);
  for (int i = 0; i < 19; ++i) w.bits((uint32_t)pre[order[i]], 3);
  std::vector<uint32_t> pc, lc, dc;
  canonical(pre, pc);
  canonical(lit, lc);
  canonical(dist, dc);
  for (int l : lit) w.code(pc[l], pre[l]);
  for (int l : dist) w.code(pc[l], pre[l]);
  for (auto& d : data) d.second ? w.code(dc[d.first], dist[d.first]) : w.code(lc[d.first], lit[d.first]);
  w.flush();
  const uint32_t a = (uint32_t)::adler32(1uL, (const Bytef*)expect.data(), (uInt)expect.size());
  for (int k = 3; k >= 0; --k) w.out.push_back((uint8_t)(a >> (8 * k)));
  return w.out;
}

static int incomplete_sets() {
  auto lens = [](size_t n, std::initializer_list<std::pair<int, int>> set) {
    std::vector<int> v(n, 0);
    for (auto& p : set) v[p.first] = p.second;
    return v;
  };
  const std::vector<int> pre_ok = lens(19, {{0, 1}, {1, 2}, {2, 2}});
  const std::vector<std::pair<int, bool>> aaaa = {{97, false}, {257, false}, {0, true}, {256, false}};
  struct Case {
    const char* name;
    std::vector<uint8_t> z, expect;
  };
  const std::vector<uint8_t> four = {97, 97, 97, 97}, none;
  std::vector<Case> cases;
  // complete literal set, ONE distance code of length 1 (incomplete, allowed): 'a' + match(3, 1)
  cases.push_back({"single distance code", dynamic_block(pre_ok, lens(258, {{97, 1}, {256, 2}, {257, 2}}), lens(1, {{0, 1}}), aaaa, four), four});
  // the same through an incomplete code-length code (three codes of length 2)
  cases.push_back({"incomplete code-length code", dynamic_block(lens(19, {{0, 2}, {1, 2}, {2, 2}}), lens(258, {{97, 1}, {256, 2}, {257, 2}}), lens(1, {{0, 1}}), aaaa, four), four});
  // incomplete literal / length set (three codes of length 2)
  cases.push_back({"incomplete literal set", dynamic_block(pre_ok, lens(258, {{97, 2}, {256, 2}, {257, 2}}), lens(1, {{0, 1}}), aaaa, four), four});
  // incomplete distance set that is not a single code (two codes of length 2)
  cases.push_back({"incomplete distance set", dynamic_block(pre_ok, lens(258, {{97, 1}, {256, 2}, {257, 2}}), lens(2, {{0, 2}, {1, 2}}), aaaa, four), four});
  // a literal set of one code of length 1 (the end-of-block code): allowed, empty output
  cases.push_back({"single literal code", dynamic_block(pre_ok, lens(257, {{256, 1}}), lens(1, {}), {{256, false}}, none), none});
  // no distance codes at all with a complete literal set: allowed while no match is met
  cases.push_back({"no distance codes", dynamic_block(pre_ok, lens(257, {{97, 1}, {256, 1}}), lens(1, {}), {{97, false}, {97, false}, {256, false}}, {97, 97}), {97, 97}});
  for (auto& c : cases) {
    std::vector<uint8_t> zout(64);
    uLongf zn = (uLongf)zout.size();
    const int zst = uncompress(zout.data(), &zn, c.z.data(), (uLong)c.z.size());
    std::vector<uint8_t> got;
    const int st = run(c.z, 64, &got);
    if ((zst == Z_OK) != (st == OK)) {
      printf("%s: zlib says %d, this decoder %d\n", c.name, zst, st);
      return 1;
    }
    if (st == OK && (got != c.expect || got.size() != zn)) {
      printf("%s: accepted with the wrong bytes\n", c.name);
      return 1;
    }
    printf("%-28s %s by both\n", c.name, st == OK ? "accepted" : "rejected");
  }
  return 0;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 200;
  if (incomplete_sets()) return 1;
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
