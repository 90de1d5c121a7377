// Deflate (RFC 1951) writer for the stage's output archives — host code, no dependencies.
//
// ref: origami/batch/core/io.py:516-518, 565-577 and origami/core/dewarp.py:545-553 — contours.N.zip and dewarp.zip are
// written with zipfile.ZIP_DEFLATED.  What goes into them compresses badly by string matching and well by symbol
// statistics: dewarp.zip is 600 KB of float32 node positions (zlib finds next to no repeats and still spends 14 ms
// looking, 85 % of the size remains), a contour is a kilobyte of digits, blanks and commas.  This writer therefore
// does the entropy-coding half of deflate only: every block is a dynamic-Huffman block of literals (no length /
// distance pairs) with a code fitted to the block's byte histogram, or a stored block where that would be larger.
// Any inflater reads the result — it is an ordinary deflate stream — at ~10x zlib's level-1 speed and within a few
// per cent of its size on this data (tests/test_deflate.py compares both).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>

namespace ogd {
namespace deflate {

constexpr int BLOCK = 1 << 16;  // input bytes per block: the histogram follows the data, a header costs ~100 bytes

struct BitWriter {
  uint8_t* p;
  uint8_t* end;
  uint64_t acc = 0;
  int n = 0;
  bool full = false;
  inline void put(uint32_t bits, int len) {  // len <= 32
    acc |= (uint64_t)bits << n;
    n += len;
    if (n >= 32) {
      if (end - p < 4) {
        full = true;
        n = 0;
        acc = 0;
        return;
      }
      const uint32_t w = (uint32_t)acc;
      std::memcpy(p, &w, 4);
      p += 4;
      acc >>= 32;
      n -= 32;
    }
  }
  inline void align() {  // to the next byte boundary, flushing whole bytes
    n = (n + 7) & ~7;
    while (n > 0) {
      if (p >= end) {
        full = true;
        return;
      }
      *p++ = (uint8_t)acc;
      acc >>= 8;
      n -= 8;
    }
  }
};

// Code lengths (<= maxbits) of a Huffman code for freq[0 .. n); symbols with freq 0 get length 0.  Plain two-queue
// construction over the sorted frequencies; if the tree comes out deeper than maxbits the frequencies are halved
// (kept >= 1) and it is built again — a flatter tree, a slightly longer output, no package-merge machinery.
inline void code_lengths(const uint32_t* freq_in, int n, int maxbits, uint8_t* lens) {
  uint32_t freq[320];
  for (int i = 0; i < n; ++i) freq[i] = freq_in[i];
  for (;;) {
    int order[320], m = 0;
    for (int i = 0; i < n; ++i)
      if (freq[i]) order[m++] = i;
    std::memset(lens, 0, (size_t)n);
    if (m == 0) return;
    if (m == 1) {
      lens[order[0]] = 1;
      return;
    }
    std::sort(order, order + m, [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
    // nodes 0 .. m-1: leaves in ascending weight; m .. 2m-2: internal nodes in the order they are made (ascending too)
    uint64_t weight[640];
    int parent[640];
    for (int i = 0; i < m; ++i) weight[i] = freq[order[i]];
    int leaf = 0, inner = m, made = m;
    auto take = [&]() {
      if (leaf < m && (inner >= made || weight[leaf] <= weight[inner])) return leaf++;
      return inner++;
    };
    while (made < 2 * m - 1) {
      const int a = take(), b = take();
      weight[made] = weight[a] + weight[b];
      parent[a] = parent[b] = made;
      ++made;
    }
    int depth[640];
    depth[2 * m - 2] = 0;
    int deepest = 0;
    for (int i = 2 * m - 3; i >= 0; --i) {
      depth[i] = depth[parent[i]] + 1;
      if (i < m && depth[i] > deepest) deepest = depth[i];
    }
    if (deepest <= maxbits) {
      for (int i = 0; i < m; ++i) lens[order[i]] = (uint8_t)depth[i];
      return;
    }
    for (int i = 0; i < n; ++i)
      if (freq[i]) freq[i] = (freq[i] >> 1) | 1;
  }
}

// canonical codes, bit-reversed for deflate's LSB-first packing: code[s] | len << 16
inline void assign_codes(const uint8_t* lens, int n, uint32_t* out) {
  int count[16] = {0};
  for (int i = 0; i < n; ++i) ++count[lens[i]];
  count[0] = 0;
  uint32_t next[16], code = 0;
  for (int len = 1; len <= 15; ++len) {
    code = (code + (uint32_t)count[len - 1]) << 1;
    next[len] = code;
  }
  for (int i = 0; i < n; ++i) {
    const int len = lens[i];
    uint32_t c = len ? next[len]++ : 0, r = 0;
    for (int k = 0; k < len; ++k) r |= ((c >> k) & 1u) << (len - 1 - k);
    out[i] = r | ((uint32_t)len << 16);
  }
}

// One block of literals.  Returns false if the output is full.
inline bool write_block(BitWriter& w, const uint8_t* in, int n, bool last) {
  uint32_t freq[257] = {0};
  {  // four histograms: runs of one value (blank paper) do not serialise on one counter
    uint32_t f1[256] = {0}, f2[256] = {0}, f3[256] = {0};
    int i = 0;
    for (; i + 4 <= n; i += 4) {
      ++freq[in[i]];
      ++f1[in[i + 1]];
      ++f2[in[i + 2]];
      ++f3[in[i + 3]];
    }
    for (; i < n; ++i) ++freq[in[i]];
    for (int s = 0; s < 256; ++s) freq[s] += f1[s] + f2[s] + f3[s];
  }
  freq[256] = 1;  // end of block
  uint8_t lens[257 + 1];
  // literal codes of at most 12 bits: four of them and the up to 7 bits pending fit one 64-bit word (below); the
  // cost against 15 bits is nil on anything but a pathological histogram
  code_lengths(freq, 257, 12, lens);
  uint64_t body_bits = 0;
  for (int s = 0; s < 257; ++s) body_bits += (uint64_t)freq[s] * lens[s];
  // the code lengths, run-length coded with the symbols 16 (repeat), 17 / 18 (zeros); one distance code of length 0
  uint8_t seq[257 + 1];
  std::memcpy(seq, lens, 257);
  seq[257] = 0;  // HDIST = 1 distance code, unused
  uint8_t sym[320], extra[320];
  int ns = 0;
  for (int i = 0; i < 258;) {
    int run = 1;
    while (i + run < 258 && seq[i + run] == seq[i]) ++run;
    if (seq[i] == 0 && run >= 3) {
      const int r = std::min(run, 138);
      sym[ns] = r <= 10 ? 17 : 18;
      extra[ns++] = (uint8_t)(r <= 10 ? r - 3 : r - 11);
      i += r;
    } else if (run >= 4) {  // the value once, then "repeat previous" 3 .. 6 times
      sym[ns] = seq[i];
      extra[ns++] = 0;
      const int r = std::min(run - 1, 6);
      sym[ns] = 16;
      extra[ns++] = (uint8_t)(r - 3);
      i += 1 + r;
    } else {
      sym[ns] = seq[i];
      extra[ns++] = 0;
      ++i;
    }
  }
  uint32_t cfreq[19] = {0};
  for (int i = 0; i < ns; ++i) ++cfreq[sym[i]];
  uint8_t clens[19];
  code_lengths(cfreq, 19, 7, clens);
  static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  int hclen = 19;
  while (hclen > 4 && clens[order[hclen - 1]] == 0) --hclen;
  uint64_t header_bits = 3 + 5 + 5 + 4 + 3 * (uint64_t)hclen;
  for (int i = 0; i < ns; ++i) header_bits += clens[sym[i]] + (sym[i] == 16 ? 2 : sym[i] == 17 ? 3 : sym[i] == 18 ? 7 : 0);
  const uint64_t stored_bits = 3 + 7 + 32 + 8 * (uint64_t)n;  // header, padding (at most), LEN / NLEN, bytes
  if (header_bits + body_bits >= stored_bits) {
    w.put(last ? 1u : 0u, 3);  // BFINAL, BTYPE 00
    w.align();
    if (w.full || w.end - w.p < 4 + n) return false;
    const uint16_t len = (uint16_t)n, nlen = (uint16_t)~n;
    std::memcpy(w.p, &len, 2);
    std::memcpy(w.p + 2, &nlen, 2);
    std::memcpy(w.p + 4, in, (size_t)n);
    w.p += 4 + n;
    return true;
  }
  uint32_t ccode[19], code[257];
  assign_codes(clens, 19, ccode);
  assign_codes(lens, 257, code);
  w.put((last ? 1u : 0u) | (2u << 1), 3);  // BFINAL, BTYPE 10
  w.put(257 - 257, 5);                     // HLIT
  w.put(1 - 1, 5);                         // HDIST
  w.put((uint32_t)(hclen - 4), 4);
  for (int i = 0; i < hclen; ++i) w.put(clens[order[i]], 3);
  for (int i = 0; i < ns; ++i) {
    w.put(ccode[sym[i]] & 0xffff, (int)(ccode[sym[i]] >> 16));
    if (sym[i] == 16) w.put(extra[i], 2);
    else if (sym[i] == 17) w.put(extra[i], 3);
    else if (sym[i] == 18) w.put(extra[i], 7);
  }
  int i = 0;
  // the literals, four per 64-bit word: no branch and one unaligned 8-byte store per four symbols, while the
  // output has room for the whole block body and the store's overshoot (else: the careful loop below all the way)
  if (!w.full && (uint64_t)(w.end - w.p) >= body_bits / 8 + 24) {
    uint64_t acc = w.acc;
    int nb = w.n;  // < 32 pending bits
    uint8_t* p = w.p;
    while (nb >= 8) {  // down to < 8 pending bits
      *p++ = (uint8_t)acc;
      acc >>= 8;
      nb -= 8;
    }
    for (; i + 4 <= n; i += 4) {
      const uint32_t a = code[in[i]], b = code[in[i + 1]], c = code[in[i + 2]], d = code[in[i + 3]];
      // the symbols are paired off first: two steps through the bit position per word instead of four
      const int la = (int)(a >> 16), lc = (int)(c >> 16);
      const uint64_t ab = (uint64_t)(a & 0xffff) | (uint64_t)(b & 0xffff) << la;
      const uint64_t cd = (uint64_t)(c & 0xffff) | (uint64_t)(d & 0xffff) << lc;
      const int lab = la + (int)(b >> 16), lcd = lc + (int)(d >> 16);
      acc |= ab << nb;
      acc |= cd << (nb + lab);
      nb += lab + lcd;  // <= 7 + 48
      std::memcpy(p, &acc, 8);
      p += nb >> 3;
      acc >>= nb & ~7;
      nb &= 7;
    }
    w.p = p;
    w.acc = acc;
    w.n = nb;
  }
  // (the rest, or everything:) two per put where their codes fit 32 bits together (always: 2 x 15)
  for (; i + 2 <= n; i += 2) {
    const uint32_t a = code[in[i]], b = code[in[i + 1]];
    const int la = (int)(a >> 16), lb = (int)(b >> 16);
    w.put((a & 0xffff) | ((b & 0xffff) << la), la + lb);
  }
  if (i < n) w.put(code[in[i]] & 0xffff, (int)(code[in[i]] >> 16));
  w.put(code[256] & 0xffff, (int)(code[256] >> 16));
  return !w.full;
}

// Raw deflate stream of in[0 .. n) into out[0 .. cap); returns the size, or -1 if cap is too small
// (n + n / 1000 + 64 bytes are always enough).
inline int64_t deflate_raw(const uint8_t* in, int64_t n, uint8_t* out, int64_t cap) {
  BitWriter w;
  w.p = out;
  w.end = out + cap;
  int64_t at = 0;
  do {
    const int len = (int)std::min<int64_t>(BLOCK - 1, n - at);  // a stored block holds at most 65 535 bytes
    if (!write_block(w, in + at, len, at + len >= n)) return -1;
    at += len;
  } while (at < n);
  w.align();
  return w.full ? -1 : (int64_t)(w.p - out);
}

}  // namespace deflate
}  // namespace ogd
