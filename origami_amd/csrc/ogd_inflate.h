// Inflate (RFC 1951) and the zlib wrapper (RFC 1950) for page scans stored as PNG — host code, no dependencies.
//
// ref: origami/core/page.py:82-88 — `PIL.Image.open(path)` decodes the page in every stage that looks at pixels;
// for a 6000 x 8000 grey PNG that is 190 ms, three quarters of it zlib's inflate, and more than everything else a
// stage does with the page here.  A deflate stream is one serial bit stream: nothing in it can be
// handed to the GPU, so the lever is the decoder's inner loop.  This one keeps 56..63 bits in a 64-bit buffer
// refilled with one unaligned load, resolves literal / length codes of up to 10 bits and distance codes of up to
// 8 bits with one table look-up (longer codes: one more, in a sub-table; 10 bits because zlib writers start a new
// block, and with it new tables, every 16 K symbols), yields two literals per look-up where both codes fit the
// table index, makes up to three look-ups per refill and copies matches eight bytes at a time.  Near either buffer's end a careful byte-wise variant of the same
// loop takes over, so corrupt or truncated input ends in an error code and never in an out-of-bounds access
// (tools/inflate_fuzz.cpp runs it under AddressSanitizer).
#pragma once
#include "ogd_checksum.h"
#include <atomic>
#include <cstdint>
#include <cstring>

namespace ogd {
namespace inflate {

enum Status { OK = 0, BAD_DATA = -1, OUTPUT_FULL = -2, TRUNCATED = -3 };

constexpr int LIT_BITS = 10, DIST_BITS = 8, PRE_BITS = 7;
// table entry: bits 0-7 code length to consume | 8-12 extra bits (12: literal pair, see pair_literals) | 13 literal
// | 14 end of block | 15 sub-table
// | 16-31 literal value / base length / base distance / sub-table start
constexpr uint32_t F_PAIR = 1u << 12, F_LITERAL = 1u << 13, F_EOB = 1u << 14, F_SUB = 1u << 15;

struct Tables {
  uint32_t lit[(1 << LIT_BITS) + 1024];   // sub-tables behind the primary table; 288 symbols, codes <= 15 bits
  uint32_t dist[(1 << DIST_BITS) + 512];
};

inline uint32_t reverse_bits(uint32_t code, int len) {  // len <= 15
  static const uint8_t nibble[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
  const uint32_t r = ((uint32_t)nibble[code & 15] << 12) | ((uint32_t)nibble[(code >> 4) & 15] << 8) |
                     ((uint32_t)nibble[(code >> 8) & 15] << 4) | nibble[(code >> 12) & 15];
  return r >> (16 - len);
}

// Canonical Huffman decoding table for `n` symbols of code lengths lens[] (0 = unused).  payload[s] holds bits 8..31
// of symbol s's entry.  Returns false for an over-subscribed code, a table that would not fit, and — as zlib's
// inflate_table does ("invalid code lengths set", "invalid literal/lengths set", "invalid distances set") — for an
// INCOMPLETE set, unless it is a single code of length 1 (allowed for literal / length and distance codes, never for the
// code-length code: may_be_single) or has no codes at all (a block without matches has no distance codes).
inline bool build_table(const uint8_t* lens, int n, const uint32_t* payload, int table_bits, uint32_t* table, int cap,
                        bool may_be_single = true) {
  int count[16] = {0};
  for (int s = 0; s < n; ++s) ++count[lens[s]];
  count[0] = 0;
  uint32_t next[16];
  uint32_t code = 0;
  int64_t left = 1;
  for (int len = 1; len <= 15; ++len) {
    left = (left << 1) - count[len];
    if (left < 0) return false;  // over-subscribed
    code = (code + (uint32_t)count[len - 1]) << 1;
    next[len] = code;
  }
  if (left > 0) {  // incomplete
    int used_codes = 0, max_len = 0;
    for (int len = 1; len <= 15; ++len)
      if (count[len]) used_codes += count[len], max_len = len;
    if (used_codes != 0 && !(may_be_single && max_len == 1)) return false;
  }
  const int primary = 1 << table_bits;
  for (int i = 0; i < primary; ++i) table[i] = 0;  // length 0: an unused code (incomplete sets), BAD_DATA if met
  // longest code behind every primary slot that needs a sub-table
  uint8_t sub_len[1 << LIT_BITS];
  std::memset(sub_len, 0, (size_t)primary);
  uint32_t rev[288 + 32];
  {
    uint32_t nx[16];
    std::memcpy(nx, next, sizeof nx);
    for (int s = 0; s < n; ++s) {
      const int len = lens[s];
      if (!len) continue;
      rev[s] = reverse_bits(nx[len]++, len);
      if (len > table_bits) {
        uint8_t& m = sub_len[rev[s] & (uint32_t)(primary - 1)];
        if (len - table_bits > m) m = (uint8_t)(len - table_bits);
      }
    }
  }
  int used = primary;
  for (int i = 0; i < primary; ++i)
    if (sub_len[i]) {
      const int size = 1 << sub_len[i];
      if (used + size > cap) return false;
      table[i] = F_SUB | (uint32_t)table_bits | ((uint32_t)sub_len[i] << 8) | ((uint32_t)used << 16);
      for (int k = 0; k < size; ++k) table[used + k] = 0;
      used += size;
    }
  for (int s = 0; s < n; ++s) {
    const int len = lens[s];
    if (!len) continue;
    if (len <= table_bits) {
      const uint32_t e = (uint32_t)len | (payload[s] << 8);
      for (uint32_t i = rev[s]; i < (uint32_t)primary; i += 1u << len) table[i] = e;
    } else {
      const uint32_t slot = table[rev[s] & (uint32_t)(primary - 1)];
      const int sbits = (int)((slot >> 8) & 31), start = (int)(slot >> 16), rest = len - table_bits;
      const uint32_t e = (uint32_t)rest | (payload[s] << 8);
      for (uint32_t i = rev[s] >> table_bits; i < (1u << sbits); i += 1u << rest) table[start + i] = e;
    }
  }
  return true;
}

// payload = flags >> 8 | extra bits | value << 8 (see the entry layout above, shifted down by 8)
inline uint32_t pay(uint32_t value, uint32_t extra, uint32_t flags) { return extra | (flags >> 8) | (value << 8); }

struct Payloads {
  uint32_t lit[288], dist[32];
  Payloads() {
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lextra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dextra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (uint32_t s = 0; s < 256; ++s) lit[s] = pay(s, 0, F_LITERAL);
    lit[256] = pay(0, 0, F_EOB);
    for (int s = 257; s < 286; ++s) lit[s] = pay(lbase[s - 257], lextra[s - 257], 0);
    lit[286] = lit[287] = pay(0, 31, 0);  // reserved symbols: 31 extra bits marks them invalid
    for (int s = 0; s < 30; ++s) dist[s] = pay(dbase[s], dextra[s], 0);
    dist[30] = dist[31] = pay(0, 31, 0);
  }
};

struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  uint64_t buf = 0;
  int cnt = 0;         // valid bits in buf
  int overrun = 0;     // zero bytes supplied past the end of the input

  inline void refill_fast() {  // needs p + 8 <= end
    uint64_t w;
    std::memcpy(&w, p, 8);
    buf |= w << cnt;
    p += (63 - cnt) >> 3;
    cnt |= 56;
  }
  inline void refill_safe() {
    while (cnt <= 56) {
      if (p < end) {
        buf |= (uint64_t)*p++ << cnt;
      } else {
        ++overrun;  // zeros; a stream that really needs them is truncated (checked by the caller)
      }
      cnt += 8;
    }
  }
  inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
  inline void drop(int n) {
    buf >>= n;
    cnt -= n;
  }
  // bits handed out that the input did not hold?
  inline bool past_end() const { return overrun * 8 > cnt; }
};

inline void store16(uint8_t* p, uint32_t v) {
  const uint16_t w = (uint16_t)v;
  std::memcpy(p, &w, 2);
}

// the literal / length entry for the bits at hand (a sub-table detour consumes the primary bits)
inline uint32_t lookup_lit(Reader& r, const Tables& t) {
  uint32_t e = t.lit[r.peek(LIT_BITS)];
  if (e & F_SUB) {
    r.drop((int)(e & 0xff));
    e = t.lit[(e >> 16) + r.peek((int)((e >> 8) & 31))];
  }
  return e;
}

// Primary entries whose literal is followed, inside the same LIT_BITS bits, by another complete literal code become
// pair entries: bits 0-7 both code lengths together, 8-11 the first code's length, F_PAIR, 16-23 / 24-31 the two
// literals.  Text and noise (scanned pages) code most bytes in 5 .. 8 bits, so most look-ups then yield two bytes
// and the serial chain look-up -> shift -> look-up is walked half as often.
inline void pair_literals(uint32_t* lit) {
  for (uint32_t i = 0; i < (1u << LIT_BITS); ++i) {
    const uint32_t e = lit[i];
    if ((e & (F_LITERAL | F_PAIR)) != F_LITERAL) continue;
    const uint32_t len1 = e & 0xff;
    if (len1 >= (uint32_t)LIT_BITS) continue;
    const uint32_t e2 = lit[i >> len1];  // the bits behind the first code, zero-padded: valid iff the code fits
    if (!(e2 & F_LITERAL)) continue;
    const uint32_t len2 = (e2 & F_PAIR) ? ((e2 >> 8) & 15) : (e2 & 0xff);
    if (len2 == 0 || len1 + len2 > (uint32_t)LIT_BITS) continue;
    lit[i] = (len1 + len2) | (len1 << 8) | F_PAIR | F_LITERAL | (e & 0x00ff0000u) | ((e2 & 0x00ff0000u) << 8);
  }
}

// One block's symbols.  FAST: the caller guarantees >= 16 input bytes and >= 320 output bytes of room at entry of
// every iteration's refill (it re-checks between iterations), so refills and copies need no tests.
template <bool FAST>
inline int decode_symbols(Reader& r, const Tables& t, uint8_t* out0, uint8_t*& out_ref, uint8_t* out_end, bool& done) {
  uint8_t* out = out_ref;
  done = false;
  for (;;) {
    if (FAST) {
      if (r.end - r.p < 16 || out_end - out < 320) break;
      r.refill_fast();
    } else {
      r.refill_safe();
    }
    uint32_t e = lookup_lit(r, t);
    if (FAST) {
      // a run of literals: up to three look-ups (each one or, from a pair entry, two literals) per refill,
      // 3 x 15 bits <= 56; a pair entry's two bytes are stored at once, a single's second byte is overwritten later
      if (e & F_LITERAL) {
        r.drop((int)(e & 0xff));
        store16(out, e >> 16);
        out += 1 + ((e >> 12) & 1);
        e = lookup_lit(r, t);
        if (e & F_LITERAL) {
          r.drop((int)(e & 0xff));
          store16(out, e >> 16);
          out += 1 + ((e >> 12) & 1);
          e = lookup_lit(r, t);
          if (e & F_LITERAL) {
            r.drop((int)(e & 0xff));
            store16(out, e >> 16);
            out += 1 + ((e >> 12) & 1);
            continue;
          }
        }
        // a length or end-of-block code follows with >= 11 bits left: top up so that the pair below fits
        r.refill_fast();
      }
    } else if (e & F_LITERAL) {
      // careful variant: one literal at a time (the first of a pair entry)
      const int len = (e & F_PAIR) ? (int)((e >> 8) & 15) : (int)(e & 0xff);
      if (len == 0) return BAD_DATA;
      r.drop(len);
      if (r.past_end()) return TRUNCATED;
      if (out >= out_end) return OUTPUT_FULL;
      *out++ = (uint8_t)(e >> 16);
      continue;
    }
    if ((e & 0xff) == 0) return BAD_DATA;  // unused code
    r.drop((int)(e & 0xff));
    if (e & F_EOB) {
      if (r.past_end()) return TRUNCATED;
      done = true;
      break;
    }
    const int lx = (int)((e >> 8) & 31);
    if (lx == 31) return BAD_DATA;
    const uint32_t length = (e >> 16) + r.peek(lx);
    r.drop(lx);
    uint32_t d = t.dist[r.peek(DIST_BITS)];
    if (d & F_SUB) {
      r.drop((int)(d & 0xff));
      d = t.dist[(d >> 16) + r.peek((int)((d >> 8) & 31))];
    }
    if ((d & 0xff) == 0) return BAD_DATA;
    r.drop((int)(d & 0xff));
    const int dx = (int)((d >> 8) & 31);
    if (dx == 31) return BAD_DATA;
    if (!FAST && r.cnt < dx) r.refill_safe();  // 15 + 5 + 15 + 13 = 48 <= 56 bits: only after sub-table detours
    const uint32_t distance = (d >> 16) + r.peek(dx);
    r.drop(dx);
    if (!FAST && r.past_end()) return TRUNCATED;
    if (distance > (uint32_t)(out - out0)) return BAD_DATA;
    const uint8_t* src = out - distance;
    if (FAST) {
      uint8_t* const stop = out + length;
      if (distance >= 8) {
        do {
          std::memcpy(out, src, 8);
          out += 8;
          src += 8;
        } while (out < stop);
      } else if (distance == 1) {
        std::memset(out, *src, (length + 7) & ~7u);
      } else {
        do {
          *out++ = *src++;
        } while (out < stop);
      }
      out = stop;  // the room test above covers the up to 7 bytes written past `stop`
    } else {
      if ((uint32_t)(out_end - out) < length) return OUTPUT_FULL;
      for (uint32_t k = 0; k < length; ++k) out[k] = src[k];
      out += length;
    }
  }
  out_ref = out;
  return OK;
}

// A raw deflate stream.  *n_out: bytes written; *n_used: input bytes consumed (the next byte boundary after the
// final block).
// progress (may be null): after every block the number of output bytes that are final is stored there (release),
// for a consumer thread that works on the output while the stream is still being decoded.
inline int inflate_raw(const uint8_t* in, int64_t n_in, uint8_t* out0, int64_t cap, int64_t* n_out, int64_t* n_used,
                       std::atomic<int64_t>* progress = nullptr) {
  static const Payloads P;
  static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  Reader r;
  r.p = in;
  r.end = in + n_in;
  uint8_t* out = out0;
  uint8_t* const out_end = out0 + cap;
  Tables t;
  for (bool last = false; !last;) {
    r.refill_safe();
    last = r.peek(1);
    const uint32_t type = (r.peek(3) >> 1);
    r.drop(3);
    if (r.past_end()) return TRUNCATED;
    if (type == 0) {  // stored: rewind to the byte boundary
      r.drop(r.cnt & 7);
      const int back = r.cnt / 8 - r.overrun;  // whole input bytes still in the bit buffer
      if (back < 0) return TRUNCATED;
      const uint8_t* p = r.p - back;
      r.buf = 0;
      r.cnt = 0;
      r.overrun = 0;
      if (r.end - p < 4) return TRUNCATED;
      const uint32_t len = p[0] | (p[1] << 8), nlen = p[2] | (p[3] << 8);
      if ((len ^ nlen) != 0xffff) return BAD_DATA;
      p += 4;
      if ((uint32_t)(r.end - p) < len) return TRUNCATED;
      if ((int64_t)(out_end - out) < (int64_t)len) return OUTPUT_FULL;
      std::memcpy(out, p, len);
      out += len;
      r.p = p + len;
      if (progress) progress->store(out - out0, std::memory_order_release);
      continue;
    }
    if (type == 3) return BAD_DATA;
    uint8_t lens[288 + 32];
    int nlit, ndist;
    if (type == 1) {
      nlit = 288;
      ndist = 32;
      for (int s = 0; s < 288; ++s) lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
      for (int s = 0; s < 32; ++s) lens[288 + s] = 5;
    } else {
      r.refill_safe();
      nlit = (int)r.peek(5) + 257;
      r.drop(5);
      ndist = (int)r.peek(5) + 1;
      r.drop(5);
      const int npre = (int)r.peek(4) + 4;
      r.drop(4);
      if (nlit > 286 || ndist > 30) return BAD_DATA;
      uint8_t plens[19] = {0};
      for (int i = 0; i < npre; ++i) {
        r.refill_safe();
        plens[order[i]] = (uint8_t)r.peek(3);
        r.drop(3);
      }
      if (r.past_end()) return TRUNCATED;
      uint32_t ppay[19], ptab[1 << PRE_BITS];
      for (uint32_t s = 0; s < 19; ++s) ppay[s] = s << 8;
      if (!build_table(plens, 19, ppay, PRE_BITS, ptab, 1 << PRE_BITS, false)) return BAD_DATA;
      int i = 0;
      while (i < nlit + ndist) {
        r.refill_safe();
        const uint32_t e = ptab[r.peek(PRE_BITS)];
        if ((e & 0xff) == 0) return BAD_DATA;
        r.drop((int)(e & 0xff));
        const int sym = (int)(e >> 16);
        if (sym < 16) {
          lens[i++] = (uint8_t)sym;
          continue;
        }
        int rep, val = 0;
        if (sym == 16) {
          if (i == 0) return BAD_DATA;
          val = lens[i - 1];
          rep = 3 + (int)r.peek(2);
          r.drop(2);
        } else if (sym == 17) {
          rep = 3 + (int)r.peek(3);
          r.drop(3);
        } else {
          rep = 11 + (int)r.peek(7);
          r.drop(7);
        }
        if (i + rep > nlit + ndist) return BAD_DATA;
        while (rep--) lens[i++] = (uint8_t)val;
      }
      if (r.past_end()) return TRUNCATED;
      if (lens[256] == 0) return BAD_DATA;  // no end-of-block code
      // the distance lengths follow the literal lengths directly: move them to their fixed place
      std::memmove(lens + 288, lens + nlit, (size_t)ndist);
      for (int s = nlit; s < 288; ++s) lens[s] = 0;
      for (int s = ndist; s < 32; ++s) lens[288 + s] = 0;
      nlit = 288;
      ndist = 32;
    }
    if (!build_table(lens, nlit, P.lit, LIT_BITS, t.lit, (int)(sizeof t.lit / sizeof t.lit[0]))) return BAD_DATA;
    pair_literals(t.lit);
    if (!build_table(lens + 288, ndist, P.dist, DIST_BITS, t.dist, (int)(sizeof t.dist / sizeof t.dist[0]))) return BAD_DATA;
    for (;;) {
      bool done;
      int st = decode_symbols<true>(r, t, out0, out, out_end, done);
      if (st != OK) return st;
      if (done) break;
      // near the end of either buffer: finish the block (or get clear of the edge) carefully
      st = decode_symbols<false>(r, t, out0, out, out_end, done);
      if (st != OK) return st;
      if (done) break;
    }
    if (progress) progress->store(out - out0, std::memory_order_release);
  }
  // bytes of input not consumed: whole bytes left in the bit buffer go back
  const int back = r.cnt / 8 - r.overrun;
  if (back < 0) return TRUNCATED;
  *n_out = out - out0;
  *n_used = (r.p - back) - in;
  return OK;
}

// Adler-32 continued from `adler` (1 for a fresh sum): ogd_checksum.h (SSSE3 where the CPU has it).
inline uint32_t adler32(uint32_t adler, const uint8_t* p, int64_t n) { return checksum::adler32(adler, p, n); }

// zlib stream (2-byte header, deflate data, Adler-32 of the output)
// verify: check the Adler-32 here; otherwise *adler_out (may be null) receives the stored value for a caller that
// sums the output itself (e.g. on another thread, behind `progress`).
inline int inflate_zlib(const uint8_t* in, int64_t n_in, uint8_t* out, int64_t cap, int64_t* n_out, bool verify = true,
                        uint32_t* adler_out = nullptr, std::atomic<int64_t>* progress = nullptr) {
  if (n_in < 6) return TRUNCATED;
  if ((in[0] & 0x0f) != 8 || (in[0] >> 4) > 7 || ((in[0] << 8) | in[1]) % 31 != 0 || (in[1] & 0x20)) return BAD_DATA;
  int64_t used = 0;
  const int st = inflate_raw(in + 2, n_in - 2, out, cap, n_out, &used, progress);
  if (st != OK) return st;
  const uint8_t* tail = in + 2 + used;
  if (n_in - 2 - used < 4) return TRUNCATED;
  const uint32_t want = ((uint32_t)tail[0] << 24) | ((uint32_t)tail[1] << 16) | ((uint32_t)tail[2] << 8) | tail[3];
  if (adler_out) *adler_out = want;
  if (!verify) return OK;
  return adler32(1, out, *n_out) == want ? OK : BAD_DATA;
}

}  // namespace inflate
}  // namespace ogd
