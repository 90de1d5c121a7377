// N4 — decoding the page image (ref: origami/core/page.py:82-84, `PIL.Image.open(path).convert("L")`): the
// 0.15 - 0.2 s libjpeg spends per broadsheet scan are what every stage that opens a page pays first.
// This is a baseline JPEG decoder for 8-bit Huffman files: one component (grayscale), or
// three (YCbCr, one interleaved scan, luma sampled 1x1, 2x1 or 2x2) converted to "L" the way PIL does it after
// libjpeg's fancy upsampling and YCbCr -> RGB.  Its pixels are the ones `Image.open(f).convert("L")` gives, bit
// for bit (tests/ compare with PIL itself).  Scans with restart intervals (DRI) go through the same passes.
// Anything else (progressive, arithmetic coding, 12 bit, CMYK, Adobe files) returns OGD_ERR_UNSUPPORTED and the
// caller keeps PIL.
//
// Restated from ITU T.81 and libjpeg's conventions: Huffman decoding (F.2.2), DC prediction, dequantisation,
// inverse DCT "islow" (13-bit constants, 2 extra bits after the column pass, final shift by 18), sample =
// range_limit[(v >> 18 rounded) & 1023] with libjpeg's table (wraps, does not clamp, beyond +-384).
//
// The entropy-coded segment is one long bit string; where a block starts is known only by decoding everything
// before it.  It is decoded in parallel all the same, by iterating to a fixed point:
//   D1 unstuff        0xFF 0x00 -> 0xFF, RSTn markers dropped (count per 4 KB chunk, scan, compact)
//   D2 decode_chunks  thread per 512-bit chunk: starts from the state (bit position, coefficient index) at
//                     which its predecessor LEFT its chunk in the previous round (round 0: the chunk border,
//                     index 0), decodes to the end of its own chunk and publishes its exit state.  Chunk 0 always
//                     starts right, so after round r chunks 0..r are exact; Huffman streams re-synchronise
//                     within a few dozen symbols, so in practice everything is exact after a handful of
//                     rounds.  The host stops when a round changes no state.  From the second round on only chunks
//                     whose entry changed (or whose exit another thread's forward walk wrote) are walked again.
//                     With restart intervals a walk that ends an MCU just in front of an interval start (host
//                     list) continues there with a fresh block.  Blank paper (a periodic bit string: period_kernel)
//                     is crossed by arithmetic instead of look-ups.
//   D3 scan of the blocks completed per chunk -> first block of every chunk
//   D4 decode_chunks<WRITE>  the same walk, now storing DC differences and AC coefficients
//   D5 inclusive scan of the DC differences per component (hipCUB; restart intervals: minus the sum in front
//      of the interval), D6 idct_kernel: thread per 8x8 block, D7 ycc_to_l_kernel (three components)
#include "ogd_common.h"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <chrono>
#include <cstring>
#include <vector>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace ogd {
namespace {

constexpr int CHUNK_BITS = 512;
constexpr int UN_CHUNK = 4096;

struct DecHuff {        // one Huffman table, decoder form
  uint16_t look8[256];  // len << 8 | symbol for codes of <= 8 bits, 0 otherwise
  int32_t maxcode[17];  // largest code of length l (-1: none)
  int32_t valoff[17];   // symbol index of the first code of length l, minus that code
  uint8_t vals[256];
};
struct DecTables {
  DecHuff dc[2], ac[2];
  uint16_t quant[3][64];         // natural order, per component
  uint8_t blk_comp[16];          // component of block m of an MCU
  uint8_t dc_sel[4], ac_sel[4];  // per component: index into dc[] / ac[]
  int32_t bpm;                   // blocks per MCU
  int32_t n0;                    // blocks of component 0 per MCU (h0 * v0)
  int32_t base[3];               // first coefficient slot of each component
  int32_t ntot;                  // slots in all
  uint32_t dc_mask, ac_mask;     // bit m: block m of an MCU uses dc[1] / ac[1]
};

static bool build_dec_huff(const uint8_t* bits, const uint8_t* vals, int nvals, DecHuff& h) {
  std::memset(&h, 0, sizeof h);
  int code = 0, k = 0;
  for (int l = 1; l <= 16; ++l) {
    h.valoff[l] = k - code;
    h.maxcode[l] = -1;
    for (int i = 0; i < bits[l - 1]; ++i, ++k, ++code) {
      if (k >= nvals || code >= (1 << l)) return false;
      if (l <= 8) {
        const int lo = code << (8 - l), hi = lo + (1 << (8 - l));
        for (int c = lo; c < hi; ++c) h.look8[c] = (uint16_t)((l << 8) | vals[k]);
      }
      h.maxcode[l] = code;
    }
    code <<= 1;
  }
  if (k != nvals) return false;
  std::memcpy(h.vals, vals, (size_t)nvals);
  return true;
}

// ---- D1 ------------------------------------------------------------------------------------------------

// Bytes that are not entropy-coded data: the 0x00 stuffed behind a 0xFF data byte, and both bytes of an RSTn
// marker (in the raw segment 0xFF is followed by 0x00 or, in files with restart intervals, by 0xD0 - 0xD7 only:
// the host has checked that).
__device__ __forceinline__ bool is_stuffing(const uint8_t* __restrict__ raw, uint64_t i, uint64_t n) {
  const uint32_t b = raw[i];
  if (i > 0 && raw[i - 1] == 0xFF && (b == 0 || (b & 0xF8u) == 0xD0u)) return true;
  return b == 0xFFu && i + 1 < n && (raw[i + 1] & 0xF8u) == 0xD0u;
}

// `markers` (may be null): set to 1 if the segment holds a 0xFF that is not followed by a stuffed zero — a marker, or
// the segment's last byte.  The host has then taken the end of the scan on trust (PARSE_TRUSTING) and has to look.
__global__ __launch_bounds__(256) void unstuff_count_kernel(const uint8_t* __restrict__ raw, uint64_t n, int nchunk,
                                                            uint64_t* __restrict__ counts, int* __restrict__ markers) {
  typedef hipcub::BlockReduce<int, 256> Reduce;
  __shared__ typename Reduce::TempStorage tmp;
  const int ch = blockIdx.x;
  int c = 0;
  if (ch < nchunk) {
    const uint64_t i0 = (uint64_t)ch * UN_CHUNK + threadIdx.x * 16;
    bool marker = false;
    for (int k = 0; k < 16; ++k)
      if (i0 + k < n) {
        c += is_stuffing(raw, i0 + k, n);
        marker |= raw[i0 + k] == 0xFF && (i0 + k + 1 >= n || raw[i0 + k + 1] != 0x00);
      }
    if (markers != nullptr && marker) *markers = 1;
  }
  const int total = Reduce(tmp).Sum(c);
  if (threadIdx.x == 0) counts[ch] = ch < nchunk ? (uint64_t)total : 0;  // counts[nchunk]: sentinel
}

__global__ __launch_bounds__(256) void unstuff_compact_kernel(const uint8_t* __restrict__ raw, uint64_t n,
                                                              const uint64_t* __restrict__ removed_before,
                                                              uint8_t* __restrict__ out) {
  typedef hipcub::BlockScan<int, 256> Scan;
  __shared__ typename Scan::TempStorage tmp;
  const uint64_t i0 = (uint64_t)blockIdx.x * UN_CHUNK + threadIdx.x * 16;
  int c = 0;
  uint32_t drop = 0;
  for (int k = 0; k < 16; ++k)
    if (i0 + k < n && is_stuffing(raw, i0 + k, n)) {
      drop |= 1u << k;
      ++c;
    }
  int before = 0;
  Scan(tmp).ExclusiveSum(c, before);
  uint64_t o = i0 - removed_before[blockIdx.x] - (uint64_t)before;
  for (int k = 0; k < 16 && i0 + k < n; ++k)
    if (!(drop >> k & 1)) out[o++] = raw[i0 + k];
}

// ---- D2 / D4 -------------------------------------------------------------------------------------------

// 64 bits of the stream starting at the byte that holds bit p (the buffer is padded with zero bytes)
__device__ __forceinline__ uint64_t window(const uint8_t* __restrict__ s, uint64_t p) {
  typedef uint64_t u64u __attribute__((aligned(1)));
  const uint64_t w = *reinterpret_cast<const u64u*>(s + (p >> 3));
  return __builtin_bswap64(w) << (p & 7);  // >= 57 valid bits at the top
}

// One Huffman symbol from the top of `w`: returns the symbol, `len` = code length (an impossible code,
// seen only on a speculative walk, is taken as symbol 0 of 16 bits: any deterministic choice will do).
__device__ __forceinline__ int decode_symbol(const uint16_t* __restrict__ look8, const int32_t* __restrict__ maxcode,
                                             const int32_t* __restrict__ valoff, const uint8_t* __restrict__ vals,
                                             uint64_t w, int& len) {
  const uint32_t c16 = (uint32_t)(w >> 48);
  const uint32_t e = look8[c16 >> 8];
  if (e) {
    len = (int)(e >> 8);
    return (int)(e & 255u);
  }
  // longer codes: all eight length tests at once (the loads do not depend on each other; a wave nearly always
  // holds a lane with a long code, so a loop with an early exit costs every symbol its full chain of LDS reads)
  uint32_t fits = 0;
#pragma unroll
  for (int l = 9; l <= 16; ++l) fits |= (uint32_t)((int)(c16 >> (16 - l)) <= maxcode[l]) << (l - 9);
  if (fits) {
    const int l = 8 + __ffs((int)fits);
    len = l;
    return vals[((int)(c16 >> (16 - l)) + valoff[l]) & 255];
  }
  len = 16;
  return 0;
}

// Blank paper is the stream's hard case: every MCU the same few bits, a parse at the wrong phase is self-consistent,
// and the right one has to be carried in from the last printed block by ONE walk.  That walk need not decode: where
// the bit string has period q and an MCU of L bits (q | L) has just been decoded inside the periodic stretch, the
// MCUs that follow are the same bits, so they end at multiples of L — arithmetic instead of Huffman look-ups.
// period_kernel finds the stretches: per[c] = the smallest q <= 32 with bit[j] == bit[j + q] for all j in the
// window [512 c, 512 c + PERIOD_WINDOW) of chunk c, 0 if there is none (or the window would pass the end of the
// stream).  The window reaches LATTICE_MAX bits past an MCU that begins in front of the next chunk border, so a walk
// that enters chunk c + 1 on the lattice finds its first MCU there covered by chunk c's window and the following ones
// by chunk c + 1's.
constexpr int PERIOD_WINDOW = CHUNK_BITS + 128;  // whole 64-bit words
constexpr int LATTICE_MAX = 64;                   // longest MCU taken as a lattice step: 2 * LATTICE_MAX <= 128

__global__ __launch_bounds__(256) void period_kernel(const uint8_t* __restrict__ s, uint64_t total_bits, int nchunk,
                                                     uint8_t* __restrict__ per /* [nchunk + 16] */) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nchunk + 16) return;
  uint32_t q = 0;
  if (i < nchunk && (uint64_t)i * CHUNK_BITS + PERIOD_WINDOW + 64 <= total_bits) {
    const uint64_t* wp = reinterpret_cast<const uint64_t*>(s + (size_t)i * (CHUNK_BITS / 8));  // 64-byte aligned
    uint64_t a = __builtin_bswap64(wp[0]), b = __builtin_bswap64(wp[1]);
    uint32_t cand = 0;
#pragma unroll
    for (int k = 1; k <= 32; ++k) cand |= (uint32_t)(a == ((a << k) | (b >> (64 - k)))) << (k - 1);
    for (int t = 1; t < PERIOD_WINDOW / 64 && cand; ++t) {  // text: cand is 0 after the first word
      a = b;
      b = __builtin_bswap64(wp[t + 1]);
      for (uint32_t rest = cand; rest; rest &= rest - 1) {
        const int k = __ffs((int)rest);
        if (a != ((a << k) | (b >> (64 - k)))) cand &= ~(1u << (k - 1));
      }
    }
    if (cand) q = (uint32_t)__ffs((int)cand);
  }
  per[i] = (uint8_t)q;
}

struct ChunkState {
  uint64_t p;  // bit position of the next symbol
  int32_t k;   // index of the next coefficient of the current block (0: a DC symbol comes next)
  int32_t m;   // index of the current block inside its MCU
};

// Walks chunk `i` from the exit state of chunk i - 1.  WRITE = false: publishes the exit state (and whether it
// changed) and the number of blocks completed.  WRITE = true: stores the coefficients.  MULTI = false: one
// component, one block per MCU, one pair of tables (the usual scan: no per-block table look-ups).
// RST = true: the scan has restart intervals; `rst_bits` lists the bit positions at which intervals 1, 2, ... begin
// in the compacted stream.  After an MCU whose end lies less than 8 bits, all of them 1, in front of such a position
// the walk continues there with a fresh block (no code word is all ones, so those bits cannot be another MCU).
// The write pass counts the restarts it takes and checks that each comes after exactly `restart` MCUs.
template <bool WRITE, bool MULTI, bool RST>
__global__ __launch_bounds__(256) void decode_chunks_kernel(const uint8_t* __restrict__ s, uint64_t total_bits,
                                                            int nchunk, const DecTables* __restrict__ tabs,
                                                            const ChunkState* __restrict__ entry,  // [nchunk]: exit of i - 1
                                                            ChunkState* __restrict__ exit_state,    // [nchunk]
                                                            uint64_t* __restrict__ nblocks, int* __restrict__ changed,
                                                            const uint64_t* __restrict__ first_block, int64_t nmcu,
                                                            int16_t* __restrict__ coef, int32_t* __restrict__ dcdiff,
                                                            int forward, const uint64_t* __restrict__ rst_bits, int nrst,
                                                            int restart, int* __restrict__ rst_check,
                                                            const uint8_t* __restrict__ dirty, uint8_t* __restrict__ touched,
                                                            const uint8_t* __restrict__ per) {
  constexpr bool LATTICE = !WRITE;  // the state passes: see period_kernel
  __shared__ DecTables t;
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(tabs);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&t);
    for (int i = threadIdx.x; i < (int)(sizeof(DecTables) / 4); i += 256) dst[i] = src[i];
    __syncthreads();
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nchunk) return;
  // State pass: a chunk whose entry state is the one it was last walked from, and whose published exit nobody else
  // has written since, would reproduce that exit: skipped (shift_states_kernel keeps the books).
  if (!WRITE && !dirty[i]) return;
  uint64_t end = min((uint64_t)(i + 1) * CHUNK_BITS, total_bits);
  // The write pass stops where the state pass left the chunk, which is where the next chunk's thread begins: a
  // lattice step (below) ends on an MCU border, up to an MCU past the first symbol border behind the chunk.
  if (WRITE && i + 1 < nchunk) end = entry[i + 1].p;
  ChunkState st = entry[i];
  uint64_t p = st.p;
  int k = st.k, m = MULTI ? st.m : 0;
  uint64_t blocks = 0;
  // blocks before this chunk: whole MCUs + m
  int64_t mcu = WRITE ? (MULTI ? (int64_t)(first_block[i] / (uint64_t)t.bpm) : (int64_t)first_block[i]) : 0;
  const size_t ntot = (size_t)t.ntot;
  int c = i;  // the chunk being walked (a state pass may carry on into the following ones, see below)
  int ri = 0;                 // the next interval start behind p (a state at an MCU border has taken the restart at p)
  uint64_t next_rst = ~0ull;  // its bit position
  int taken = 0;              // write pass: restarts taken on this walk, and whether one came after the wrong number of MCUs
  bool misplaced = false;
  if (RST) {
    int lo = 0, hi = nrst;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (rst_bits[mid] <= p) lo = mid + 1; else hi = mid;
    }
    ri = lo;
    if (ri < nrst) next_rst = rst_bits[ri];
  }
  // the current block's tables and coefficient slot (component-major, stream order inside a component): looked up
  // when the block changes, not per symbol
  const DecHuff* hd = &t.dc[0];
  const DecHuff* ha = &t.ac[0];
  size_t slot = (size_t)mcu;
  const int bpm = t.bpm, n0 = t.n0;
  const uint32_t dc_mask = t.dc_mask, ac_mask = t.ac_mask;
  auto select_block = [&]() {
    if (MULTI) {  // no LDS look-up on the state pass: a lone forward walk pays every one of them in full
      hd = &t.dc[(dc_mask >> m) & 1u];
      ha = &t.ac[(ac_mask >> m) & 1u];
      if (WRITE) {
        const int comp = m < n0 ? 0 : m - n0 + 1;
        slot = (size_t)t.base[comp] + (size_t)(comp == 0 ? mcu * n0 + m : mcu);
      }
    } else if (WRITE) {
      slot = (size_t)mcu;
    }
  };
  select_block();
  uint64_t pb = ~0ull;  // where the current MCU began (if on this walk)
  uint32_t lat = 0, qlat = 0;  // the walk left its last chunk on a lattice of MCUs of `lat` bits, period qlat | lat
  uint32_t qc = 0;             // per[c]
  int walked = 0;              // chunks after the walk's own that were decoded (lattice chunks are not counted)
  // one 64-bit load serves several symbols: w0 holds >= 57 valid bits from bit p0 on, a symbol takes <= 31
  uint64_t p0 = p, w0 = window(s, p);
walk:
  if (LATTICE) qc = per[c];
  while (p < end) {
    if (p - p0 > 57 - 31) {
      p0 = p;
      w0 = window(s, p);
    }
    const uint64_t w = w0 << (p - p0);
    int len;
    if (k == 0) {
      if (!MULTI || m == 0) pb = p;
      const int sz = decode_symbol(hd->look8, hd->maxcode, hd->valoff, hd->vals, w, len) & 15;
      if (WRITE && mcu < nmcu) {
        int v = 0;
        if (sz) {
          v = (int)((w << len) >> (64 - sz));
          if (v < (1 << (sz - 1))) v -= (1 << sz) - 1;
        }
        dcdiff[slot] = v;
      }
      p += (uint64_t)(len + sz);
      k = 1;
    } else {
      const int rs = decode_symbol(ha->look8, ha->maxcode, ha->valoff, ha->vals, w, len);
      const int r = rs >> 4, sz = rs & 15;
      if (sz) {
        k += r;
        if (WRITE && mcu < nmcu && k < 64) {
          int v = (int)((w << len) >> (64 - sz));
          if (v < (1 << (sz - 1))) v -= (1 << sz) - 1;
          // zig-zag position k of the block: [8 groups][slot] x 8 int16 (the encoder's layout)
          coef[((size_t)(k >> 3) * ntot + slot) * 8 + (k & 7)] = (int16_t)v;
        }
        ++k;
        p += (uint64_t)(len + sz);
      } else if (r == 15) {
        k += 16;
        p += (uint64_t)len;
      } else {  // end of block
        k = 64;
        p += (uint64_t)len;
      }
      if (k >= 64) {
        k = 0;
        ++blocks;
        if (!MULTI) {
          ++mcu;
        } else if (++m >= bpm) {
          m = 0;
          ++mcu;
        }
        select_block();
        if (RST && m == 0) {
          if (next_rst < p) {  // passed without a restart: this walk is on a wrong parse
            while (ri + 1 < nrst && rst_bits[ri + 1] <= p) ++ri;
            // The right parse begins a fresh MCU at every interval start, whatever came before: a state pass goes
            // back to the last one it has passed and is exact from there on (its block count is not, but a walk
            // from an exact entry state never comes here, and that walk writes the count that is used).
            if (!WRITE) {
              p = rst_bits[ri];
              pb = ~0ull;
            }
            ++ri;
            next_rst = ri < nrst ? rst_bits[ri] : ~0ull;
          }
          const uint64_t pad = next_rst - p;
          if (pad < 8 && (pad == 0 || (window(s, p) >> (64 - pad)) == (1ull << pad) - 1)) {
            if (WRITE) {
              ++taken;
              if (mcu != (int64_t)(ri + 1) * restart) misplaced = true;
            }
            p = next_rst;
            pb = ~0ull;  // the MCU in front of p does not end at p
            ++ri;
            next_rst = ri < nrst ? rst_bits[ri] : ~0ull;
          }
        }
        if (LATTICE && m == 0) lat = 0;
        if (LATTICE && m == 0 && pb != ~0ull) {
          const uint64_t L64 = p - pb;
          // (with restart intervals: only if no interval can begin in the stretch — an MCU that ends less than 8
          // bits in front of one is followed by the interval, not by the next period)
          if (qc != 0 && L64 <= LATTICE_MAX && pb >= (uint64_t)c * CHUNK_BITS && (uint32_t)L64 % qc == 0 &&
              (!RST || next_rst >= end + LATTICE_MAX + 8)) {
            // an MCU inside a periodic window, a whole number of periods long: the rest of the chunk repeats it
            lat = (uint32_t)L64;
            qlat = qc;
            if (p < end) {
              const uint32_t n = ((uint32_t)(end - p) + lat - 1) / lat;
              p += (uint64_t)n * lat;
              blocks += (uint64_t)n * (uint64_t)bpm;
            }
          } else {
            if (!MULTI && !RST && L64 <= 28) {
              // No periodic window (the margin between two printed lines, shorter than a chunk): a block's parse is
              // a function of exactly the bits it consumed, so while the next L bits repeat them the next block ends
              // L bits on — counted without decoding, several per load, as long as it ends inside the chunk.
              const uint32_t L = (uint32_t)L64;
              const uint64_t pat = window(s, pb) >> (64 - L);
              const uint32_t per_load = 57u / L;
              for (;;) {
                const uint64_t wq = window(s, p);
                uint32_t j = 0;
                while (j < per_load && p + L <= end && ((wq << (j * L)) >> (64 - L)) == pat) {
                  ++j;
                  p += L;
                }
                blocks += j;
                if (j < per_load) break;
              }
            }
          }
        }
      }
    }
  }
  // One atomic per walk, AFTER the loop.  With `atomicAdd(rst_check, 1)` inside the divergent loop above, the round-1
  // form of this kernel (commit 465d566) decoded long three-component streams to wrong pixels, differently on
  // every run.  Round 2 reproduced it on demand (tools/jpeg_atomic_variants.py, profiles/r02/c_jpeg_atomic_probe.log):
  // not the compiler's atomic optimiser; a plain store in the atomic's place, the atomic with its return value used,
  // or followed by s_waitcnt 0, are all harmless.  Round 3 settled what it is NOT (tools/jpeg_atomic_owner_probe.py,
  // profiles/r03/i_jpeg_owner_probe.log: the faulty pass and a clean copy of the kernel run back to back on the same
  // settled chunk states and are compared coefficient by coefficient): it is not a race of this protocol.  Of 9.6 M /
  // 72 M coefficients 36 000 / 3 000 differ, and EVERY one of them is a non-zero value at a position where the clean
  // pass also stored a non-zero value — no store is missing, none lands where the clean pass wrote nothing, every
  // restart is counted: the parse (code lengths, runs, block and interval borders, hence every address) is right in
  // every walk, only the magnitude bits stored are wrong; a second walk on a coefficient's slot would have to be on
  // another parse and would leave foreign and missing stores.  The damaged coefficients are those of the workgroups
  // with index >= 256 (chunks >= 65536: 195-202 of the 211 such wavefronts of one file, all 16 of another, ~60 of 64
  // lanes each; a file of 767 wavefronts, every one of which executes the atomic, decodes right), whether or not any
  // lane of the wavefront ever took a restart — five of six damaged wavefronts never execute the atomic at all.  And
  // only in that exact code shape: the same kernel with a template flag around the atomic and one store after the
  // loop shows no difference in 86 M coefficients.  Address-right, value-wrong stores in wavefronts that never reach
  // the atomic, selected by their workgroup index, in one instruction schedule only: below the source level (code
  // generation or hardware; not resolved further).  Rule kept in this library: no fire-and-forget atomic inside a
  // walk; count in a register, publish once at the end; tests/test_gpu_jpeg_decode.py keeps the case.
  if (WRITE && RST) {
    if (taken) atomicAdd(rst_check, taken);
    if (misplaced) rst_check[1] = 1;
  }
  if (!WRITE) {
    const ChunkState old = exit_state[c];
    const bool same = old.p == p && old.k == k && old.m == m;
    if (!same) {
      *changed = 1;
      ChunkState ns;
      ns.p = p;
      ns.k = k;
      ns.m = m;
      exit_state[c] = ns;
      if (c != i) touched[c] = 1;  // chunk c's own walk has to confirm this next round
    }
    if (c == i || !same) nblocks[c] = blocks;
    // Where the stream is periodic (a blank margin: every block the same six bits) a wrong parse can be
    // self-consistent, walks never re-synchronise and exactness would advance one chunk per round.  So a walk
    // that arrives at a border in a state other than the published one carries that state on instead of waiting
    // for the next round: by arithmetic through chunks on its lattice (in any round), by decoding through up to
    // `forward` other chunks (none in the first rounds, when most published states are still guesses, then
    // doubling).  Several threads may then write one chunk's state in a round; whatever survives is only a better
    // or worse guess — the loop on the host ends only after a round in which every chunk reproduced its published
    // state from its predecessor's.
    if (!same && c + 1 < nchunk && p < total_bits) {
      ++c;
      if (LATTICE && lat != 0) {
        // On the lattice (k == 0, m == 0, the MCU in front of p lies in a window of period qlat): chunks whose
        // window has that period are crossed eight flags per load.  Their states are written without looking at
        // the published ones; `touched` sends each chunk's own thread over them next round.
        typedef uint64_t u64u __attribute__((aligned(1)));
        for (;;) {
          const uint64_t flags = *reinterpret_cast<const u64u*>(per + c);
          int nok = 0;
          while (nok < 8 && (uint32_t)((flags >> (8 * nok)) & 255u) == qlat) ++nok;
          for (int j = 0; j < nok; ++j, ++c) {  // per[] is 0 from the last chunks on: c stays below nchunk - 1
            const uint64_t border = (uint64_t)(c + 1) * CHUNK_BITS;
            if (RST && next_rst < border + LATTICE_MAX + 8) {  // an interval may begin here: decoded, not stepped over
              nok = 0;
              break;
            }
            uint32_t n = 0;
            if (p < border) {
              n = ((uint32_t)(border - p) + lat - 1) / lat;
              p += (uint64_t)n * lat;
            }
            ChunkState ns;
            ns.p = p;
            ns.k = 0;
            ns.m = 0;
            exit_state[c] = ns;
            nblocks[c] = (uint64_t)n * (uint64_t)bpm;
            touched[c] = 1;
          }
          if (nok < 8) break;
        }
        pb = ~0ull;
      }
      lat = 0;
      if (walked < forward) {  // otherwise chunk c's own thread takes over next round: its entry has changed
        ++walked;
        blocks = 0;
        end = min((uint64_t)(c + 1) * CHUNK_BITS, total_bits);
        p0 = p;
        w0 = window(s, p);
        goto walk;
      }
    }
  }
}


// DC prediction in a scan with restart intervals: `in` holds the inclusive sums of a component's differences over
// the whole scan, and every interval starts from zero: out = in - (the sum just before the interval's first block).
__global__ __launch_bounds__(256) void dc_restart_fix_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                                             int ntot, int base1, int base2, int seg0, int seg12) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= ntot) return;
  const int base = b >= base2 && base2 > 0 ? base2 : (b >= base1 && base1 > 0 ? base1 : 0);
  const int seg = base == 0 ? seg0 : seg12;
  const int local = b - base, start = local - local % seg;
  out[b] = in[b] - (start > 0 ? in[base + start - 1] : 0);
}

// entry[i] = exit[i - 1] for the next round; entry[0] stays (0, 0).  dirty[i]: chunk i has to be walked again — its
// entry changed, or a forward walk of another thread wrote its exit state in the round before, or it has not been
// walked yet (init_states_kernel).  A predecessor that has not been walked yet leaves the entry at its first guess.
__global__ __launch_bounds__(256) void shift_states_kernel(const ChunkState* __restrict__ exit_state, int nchunk,
                                                           ChunkState* __restrict__ entry, uint8_t* __restrict__ dirty,
                                                           uint8_t* __restrict__ touched) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nchunk) return;
  const ChunkState old = entry[i];
  ChunkState z;
  z.p = 0;
  z.k = 0;
  z.m = 0;
  if (i > 0) z = exit_state[i - 1];
  if (z.p == ~0ull) z = old;
  dirty[i] = (uint8_t)(touched[i] || old.p != z.p || old.k != z.k || old.m != z.m || exit_state[i].p == ~0ull);
  touched[i] = 0;
  entry[i] = z;
}

// First guess: every chunk begins a block at its border.  Chunks inside a periodic stretch (per[i] == per[i - 1] != 0)
// sit the first round out: the chunk at the head of the stretch crosses them on its lattice, so their states come
// from ONE guess and agree with each other — guessed one by one (each border is another phase of the period) they
// would change with every round and keep overwriting what the one walk with the right state writes.
__global__ __launch_bounds__(256) void init_states_kernel(int nchunk, ChunkState* __restrict__ entry,
                                                          ChunkState* __restrict__ exit_state,
                                                          uint64_t* __restrict__ nblocks, uint8_t* __restrict__ dirty,
                                                          uint8_t* __restrict__ touched, const uint8_t* __restrict__ per) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i > nchunk) return;
  nblocks[i] = 0;  // [nchunk]: sentinel of the scan
  if (i == nchunk) return;
  ChunkState z;
  z.p = (uint64_t)i * CHUNK_BITS;
  z.k = 0;
  z.m = 0;
  entry[i] = z;
  z.p = ~0ull;  // not walked yet
  exit_state[i] = z;
  dirty[i] = (uint8_t)!(per != nullptr && i > 0 && per[i] != 0 && per[i] == per[i - 1]);
  touched[i] = 0;
}

// ---- D6 ------------------------------------------------------------------------------------------------

__device__ __forceinline__ int range_limit(int x) {
  // libjpeg's post-IDCT table, indexed with (x & 1023); x already includes the +128 level shift implicitly
  const int i = x & 1023;
  if (i < 128) return i + 128;
  if (i < 512) return 255;
  if (i < 896) return 0;
  return i - 896;
}

template <int PASS>
__device__ __forceinline__ void idct8(int& d0, int& d1, int& d2, int& d3, int& d4, int& d5, int& d6, int& d7) {
  constexpr int CB = 13, P1 = 2;
  constexpr int SH = PASS == 1 ? CB - P1 : CB + P1 + 3;
  constexpr int R = 1 << (SH - 1);
  int z2 = d2, z3 = d6;
  int z1 = (z2 + z3) * 4433;
  const int tmp2 = z1 + z3 * (-15137);
  const int tmp3 = z1 + z2 * 6270;
  z2 = d0;
  z3 = d4;
  const int tmp0 = (z2 + z3) << CB, tmp1 = (z2 - z3) << CB;
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  int t0 = d7, t1 = d5, t2 = d3, t3 = d1;
  z1 = t0 + t3;
  z2 = t1 + t2;
  z3 = t0 + t2;
  int z4 = t1 + t3;
  const int z5 = (z3 + z4) * 9633;
  t0 *= 2446;
  t1 *= 16819;
  t2 *= 25172;
  t3 *= 12299;
  z1 *= -7373;
  z2 *= -20995;
  z3 = z3 * (-16069) + z5;
  z4 = z4 * (-3196) + z5;
  t0 += z1 + z3;
  t1 += z2 + z4;
  t2 += z2 + z3;
  t3 += z1 + z4;
  d0 = (tmp10 + t3 + R) >> SH;
  d7 = (tmp10 - t3 + R) >> SH;
  d1 = (tmp11 + t2 + R) >> SH;
  d6 = (tmp11 - t2 + R) >> SH;
  d2 = (tmp12 + t1 + R) >> SH;
  d5 = (tmp12 - t1 + R) >> SH;
  d3 = (tmp13 + t0 + R) >> SH;
  d4 = (tmp13 - t0 + R) >> SH;
}

// Where the samples of a component go: its plane, the part of it that exists, how its blocks are laid out.
struct Planes {
  uint8_t* p[3];
  ptrdiff_t stride[3];
  int W[3], H[3];  // stores are clipped to this
  int mcux;        // MCUs per row
  int h0, v0;      // blocks of component 0 per MCU, horizontally and vertically
};

__global__ __launch_bounds__(256) void idct_kernel(const uint4* __restrict__ coef, const int32_t* __restrict__ dc,
                                                   const DecTables* __restrict__ tabs, const Planes pl) {
  __shared__ int s_q[3][64];  // dequantisation factors in zig-zag order
  __shared__ int s_base[3];
  constexpr uint8_t ZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                              41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                              30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  if (threadIdx.x < 192) s_q[threadIdx.x >> 6][threadIdx.x & 63] = tabs->quant[threadIdx.x >> 6][ZZ[threadIdx.x & 63]];
  if (threadIdx.x < 3) s_base[threadIdx.x] = tabs->base[threadIdx.x];
  __syncthreads();
  const int ntot = tabs->ntot;
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= ntot) return;
  const int comp = b >= s_base[2] && s_base[2] > 0 ? 2 : (b >= s_base[1] && s_base[1] > 0 ? 1 : 0);
  int d[64];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const uint4 v = coef[(size_t)g * ntot + b];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = g * 8 + j;
      const int c = (int)(int16_t)((w[j >> 1] >> (16 * (j & 1))) & 0xffffu);
      d[ZZ[k]] = (k == 0 ? dc[b] : c) * s_q[comp][k];
    }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c)  // pass 1: columns
    idct8<1>(d[c], d[8 + c], d[16 + c], d[24 + c], d[32 + c], d[40 + c], d[48 + c], d[56 + c]);
#pragma unroll
  for (int r = 0; r < 8; ++r)  // pass 2: rows
    idct8<2>(d[r * 8], d[r * 8 + 1], d[r * 8 + 2], d[r * 8 + 3], d[r * 8 + 4], d[r * 8 + 5], d[r * 8 + 6], d[r * 8 + 7]);
  // block position: component 0 holds h0 x v0 blocks per MCU in raster order, the others one
  const int local = b - s_base[comp];
  int bx, by;
  if (comp == 0) {
    const int n0 = pl.h0 * pl.v0;
    const int mcu = local / n0, j = local - mcu * n0;
    const int my = mcu / pl.mcux, mx = mcu - my * pl.mcux;
    bx = mx * pl.h0 + j % pl.h0;
    by = my * pl.v0 + j / pl.h0;
  } else {
    by = local / pl.mcux;
    bx = local - by * pl.mcux;
  }
  const int x0 = bx * 8, y0 = by * 8, W = pl.W[comp], H = pl.H[comp];
  if (x0 >= W) return;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    if (y0 + r >= H) break;
    uint8_t* row = pl.p[comp] + (ptrdiff_t)(y0 + r) * pl.stride[comp] + x0;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      lo |= (uint32_t)range_limit(d[r * 8 + c]) << (8 * c);
      hi |= (uint32_t)range_limit(d[r * 8 + 4 + c]) << (8 * c);
    }
    if (x0 + 8 <= W) {
      typedef uint32_t u32u __attribute__((aligned(1)));
      *reinterpret_cast<u32u*>(row) = lo;
      *reinterpret_cast<u32u*>(row + 4) = hi;
    } else {
      for (int c = 0; c < 8 && x0 + c < W; ++c) row[c] = (uint8_t)((c < 4 ? lo >> (8 * c) : hi >> (8 * (c - 4))) & 255u);
    }
  }
}

// ---- D7: YCbCr planes -> "L" ---------------------------------------------------------------------------------
// libjpeg's fancy ("triangle") chroma upsampling, its YCbCr -> RGB tables and PIL's RGB -> L:
//   h2v2: colsum = 3 * near row + far row (far: the row above for even output rows, below for odd ones; edge rows
//         repeat), out = (3 * colsum[x/2] + colsum[x/2 -+ 1] + (8 | 7)) >> 4 (left neighbour and 8 for even x)
//   h2v1: out = (3 * c[x/2] + c[x/2 -+ 1] + (1 | 2)) >> 2;  edge columns repeat their own sample
//   R = Y + ((91881 Cr + 32768) >> 16), G = Y + ((-22554 Cb - 46802 Cr + 32768) >> 16), B = Y + ((116130 Cb + 32768) >> 16)
//   with Cb, Cr - 128, clamped;  L = (19595 R + 38470 G + 7471 B + 32768) >> 16
__device__ __forceinline__ uint32_t ycc_l(int Y, int cb, int cr) {
  cb -= 128;
  cr -= 128;
  const int R = min(max(Y + ((91881 * cr + 32768) >> 16), 0), 255);
  const int G = min(max(Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16), 0), 255);
  const int B = min(max(Y + ((116130 * cb + 32768) >> 16), 0), 255);
  return (uint32_t)((R * 19595 + G * 38470 + B * 7471 + 0x8000) >> 16);
}

// Thread = four pixels of a row (x0 a multiple of 4).  Plane strides are multiples of 16 bytes (dword loads).
__global__ __launch_bounds__(256) void ycc_to_l_kernel(const uint8_t* __restrict__ py, ptrdiff_t sy,
                                                       const uint8_t* __restrict__ pcb, const uint8_t* __restrict__ pcr,
                                                       ptrdiff_t sc, int cw, int ch, int h0, int v0, int H, int W,
                                                       uint8_t* __restrict__ out, ptrdiff_t so) {
  const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
  if (x0 >= W || y >= H) return;
  const uint32_t yy = *reinterpret_cast<const uint32_t*>(py + (ptrdiff_t)y * sy + x0);
  int cb[4], cr[4];
  if (h0 == 1) {
    const uint32_t b = *reinterpret_cast<const uint32_t*>(pcb + (ptrdiff_t)y * sc + x0);
    const uint32_t r = *reinterpret_cast<const uint32_t*>(pcr + (ptrdiff_t)y * sc + x0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      cb[j] = (int)((b >> (8 * j)) & 255u);
      cr[j] = (int)((r >> (8 * j)) & 255u);
    }
  } else {
    // chroma columns xc - 1 ... xc + 2 serve the four pixels (edge columns repeat)
    const int xc = x0 >> 1;
    const int xi[4] = {max(xc - 1, 0), xc, min(xc + 1, cw - 1), min(xc + 2, cw - 1)};
    const int yc = v0 == 2 ? y >> 1 : y;
    const int yf = v0 == 2 ? min(max(yc + ((y & 1) ? 1 : -1), 0), ch - 1) : yc;
    const uint8_t* const planes[2] = {pcb, pcr};
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      const uint8_t* rn = planes[pi] + (ptrdiff_t)yc * sc;
      const uint8_t* rf = planes[pi] + (ptrdiff_t)yf * sc;
      int cs[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) cs[j] = v0 == 2 ? 3 * rn[xi[j]] + rf[xi[j]] : rn[xi[j]];
      int* o = pi == 0 ? cb : cr;
      if (v0 == 2) {
        o[0] = (3 * cs[1] + cs[0] + 8) >> 4;
        o[1] = (3 * cs[1] + cs[2] + 7) >> 4;
        o[2] = (3 * cs[2] + cs[1] + 8) >> 4;
        o[3] = (3 * cs[2] + cs[3] + 7) >> 4;
      } else {
        o[0] = (3 * cs[1] + cs[0] + 1) >> 2;
        o[1] = (3 * cs[1] + cs[2] + 2) >> 2;
        o[2] = (3 * cs[2] + cs[1] + 1) >> 2;
        o[3] = (3 * cs[2] + cs[3] + 2) >> 2;
      }
    }
  }
  uint32_t l = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) l |= ycc_l((int)((yy >> (8 * j)) & 255u), cb[j], cr[j]) << (8 * j);
  uint8_t* o = out + (ptrdiff_t)y * so + x0;
  if (x0 + 4 <= W && (reinterpret_cast<uintptr_t>(o) & 3) == 0) {
    *reinterpret_cast<uint32_t*>(o) = l;
  } else {
    for (int j = 0; j < 4 && x0 + j < W; ++j) o[j] = (uint8_t)((l >> (8 * j)) & 255u);
  }
}

// ---- host: file structure --------------------------------------------------------------------------------

struct JpegInfo {
  int H = 0, W = 0, ncomp = 0;
  int h0 = 1, v0 = 1;  // luma blocks per MCU (three components)
  bool supported = false;
  const char* why = "";
  size_t scan_begin = 0, scan_end = 0;
  int restart = 0;            // MCUs per restart interval, 0: none
  bool scan_unchecked = false;  // PARSE_TRUSTING took the scan's end from the file's last two bytes (see parse_jpeg)
  std::vector<uint64_t> rst;  // where intervals 1, 2, ... begin: bit positions in the scan without stuffing and markers
  DecTables tabs;
};

#define JP_UNSUP(msg)     \
  do {                    \
    info.why = msg;       \
    return OGD_OK;        \
  } while (0)

// Parses the marker segments.  Returns OGD_ERR_INVALID for something that is not a JPEG file at all; a valid
// file this decoder does not cover leaves info.supported = false with the reason in info.why.
//   PARSE_FULL      walks the scan to its end (every 0xFF of it: 0.5 ms for a broadsheet scan);
//   PARSE_HEADERS   stops at the start of the scan (size and kind are known, nothing said about the scan);
//   PARSE_TRUSTING  for the decode calls: a file without restart intervals that ends in an EOI marker has its scan
//                   end there — unless another marker stands inside, which the first device pass finds out while
//                   it counts the stuffed bytes anyway (info.scan_unchecked; decode_to_device then returns
//                   SCAN_HAS_MARKERS and the caller parses again with PARSE_FULL).
enum ParseMode { PARSE_FULL, PARSE_HEADERS, PARSE_TRUSTING };
constexpr int SCAN_HAS_MARKERS = 1;  // internal, never leaves this file
static int parse_jpeg(const uint8_t* f, size_t n, JpegInfo& info, ParseMode mode = PARSE_FULL) {
  const bool headers_only = mode == PARSE_HEADERS;
  OGD_REQUIRE(n >= 4 && f[0] == 0xFF && f[1] == 0xD8, "not a JPEG file (no SOI marker)");
  uint8_t qt[4][64];
  bool have_q[4] = {false, false, false, false};
  struct RawHuff { uint8_t bits[16]; uint8_t vals[256]; int n = -1; } hd[4], ha[4];
  int tq[3] = {-1, -1, -1}, td[3] = {-1, -1, -1}, ta[3] = {-1, -1, -1}, comp_id[3] = {-1, -1, -1};
  int hs[3] = {1, 1, 1}, vs[3] = {1, 1, 1};
  bool have_sof = false, have_jfif = false;
  int restart = 0;  // MCUs per restart interval (DRI), 0: none
  size_t p = 2;
  while (true) {
    OGD_REQUIRE(p + 4 <= n, "truncated JPEG file");
    OGD_REQUIRE(f[p] == 0xFF, "JPEG marker expected at byte %zu", p);
    while (p < n && f[p] == 0xFF) ++p;  // fill bytes
    OGD_REQUIRE(p < n, "truncated JPEG file");
    const int m = f[p++];
    if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
    OGD_REQUIRE(m != 0xD9, "JPEG file ends before the scan");
    OGD_REQUIRE(p + 2 <= n, "truncated JPEG file");
    const size_t len = ((size_t)f[p] << 8) | f[p + 1];
    OGD_REQUIRE(len >= 2 && p + len <= n, "bad JPEG segment length");
    const uint8_t* s = f + p + 2;
    const size_t sl = len - 2;
    if (m == 0xDB) {
      size_t o = 0;
      while (o < sl) {
        const int pq = s[o] >> 4, id = s[o] & 15;
        OGD_REQUIRE(id < 4, "bad quantisation table id");
        if (pq != 0) JP_UNSUP("16-bit quantisation table");
        OGD_REQUIRE(o + 65 <= sl, "truncated DQT segment");
        static const uint8_t ZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                       41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                       30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
        for (int k = 0; k < 64; ++k) qt[id][ZZ[k]] = s[o + 1 + k];
        have_q[id] = true;
        o += 65;
      }
    } else if (m == 0xC4) {
      size_t o = 0;
      while (o < sl) {
        OGD_REQUIRE(o + 17 <= sl, "truncated DHT segment");
        const int tc = s[o] >> 4, id = s[o] & 15;
        OGD_REQUIRE(tc < 2 && id < 4, "bad Huffman table id");
        RawHuff& h = tc ? ha[id] : hd[id];
        int cnt = 0;
        for (int k = 0; k < 16; ++k) cnt += (h.bits[k] = s[o + 1 + k]);
        OGD_REQUIRE(cnt <= 256 && o + 17 + (size_t)cnt <= sl, "bad DHT segment");
        std::memcpy(h.vals, s + o + 17, (size_t)cnt);
        h.n = cnt;
        o += 17 + (size_t)cnt;
      }
    } else if (m == 0xC0 || m == 0xC1) {
      OGD_REQUIRE(sl >= 6, "truncated SOF segment");
      info.H = (s[1] << 8) | s[2];
      info.W = (s[3] << 8) | s[4];
      info.ncomp = s[5];
      have_sof = true;
      if (s[0] != 8) JP_UNSUP("sample precision other than 8 bits");
      if (info.ncomp != 1 && info.ncomp != 3) JP_UNSUP("neither one nor three components");
      OGD_REQUIRE(sl >= 6 + 3 * (size_t)info.ncomp, "truncated SOF segment");
      for (int c = 0; c < info.ncomp; ++c) {
        comp_id[c] = s[6 + 3 * c];
        hs[c] = s[7 + 3 * c] >> 4;
        vs[c] = s[7 + 3 * c] & 15;
        tq[c] = s[8 + 3 * c];
        OGD_REQUIRE(tq[c] < 4 && hs[c] >= 1 && vs[c] >= 1, "bad component specification");
      }
      if (info.H == 0 || info.W == 0) JP_UNSUP("image height defined by a DNL marker");
      if (info.ncomp == 3) {
        if (comp_id[0] != 1 || comp_id[1] != 2 || comp_id[2] != 3) JP_UNSUP("component ids other than 1, 2, 3 (not YCbCr)");
        if (hs[1] != 1 || vs[1] != 1 || hs[2] != 1 || vs[2] != 1) JP_UNSUP("subsampled luma or oversampled chroma");
        if (!((hs[0] == 1 && vs[0] == 1) || (hs[0] == 2 && vs[0] == 1) || (hs[0] == 2 && vs[0] == 2)))
          JP_UNSUP("luma sampling other than 1x1, 2x1, 2x2");
        // libjpeg upsamples the chroma of narrow images by replication, not with the triangle filter
        if (hs[0] == 2 && (info.W + 1) / 2 <= 2) JP_UNSUP("image too narrow for fancy upsampling");
      }
    } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      if (sl >= 6) {
        info.H = (s[1] << 8) | s[2];
        info.W = (s[3] << 8) | s[4];
        info.ncomp = s[5];
      }
      JP_UNSUP("not a baseline / sequential Huffman file");
    } else if (m == 0xDD) {
      OGD_REQUIRE(sl >= 2, "truncated DRI segment");
      restart = (s[0] << 8) | s[1];
    } else if (m == 0xE0) {
      if (sl >= 5 && std::memcmp(s, "JFIF", 5) == 0) have_jfif = true;
    } else if (m == 0xEE) {
      if (sl >= 5 && std::memcmp(s, "Adobe", 5) == 0) JP_UNSUP("Adobe colour transform marker");
    } else if (m == 0xDA) {
      OGD_REQUIRE(have_sof, "SOS before SOF");
      OGD_REQUIRE(sl >= 1, "bad SOS segment");
      if (s[0] != info.ncomp) JP_UNSUP("scan does not hold all components (non-interleaved file)");
      OGD_REQUIRE(sl >= 4 + 2 * (size_t)info.ncomp, "bad SOS segment");
      for (int c = 0; c < info.ncomp; ++c) {
        OGD_REQUIRE(s[1 + 2 * c] == comp_id[c], "bad SOS segment (component order)");
        td[c] = s[2 + 2 * c] >> 4;
        ta[c] = s[2 + 2 * c] & 15;
        OGD_REQUIRE(td[c] < 4 && ta[c] < 4, "bad Huffman table selector");
      }
      const uint8_t* e = s + 1 + 2 * info.ncomp;
      if (e[0] != 0 || e[1] != 63 || e[2] != 0) JP_UNSUP("spectral selection / successive approximation");
      p += len;
      break;
    }
    p += len;
  }
  if (info.ncomp == 3 && !have_jfif) JP_UNSUP("three components without a JFIF header (colour space not certain)");
  for (int c = 0; c < info.ncomp; ++c)
    OGD_REQUIRE(have_q[tq[c]] && hd[td[c]].n >= 0 && ha[ta[c]].n >= 0, "JPEG scan refers to a table that was not defined");
  if (headers_only) {  // size and kind are known; whether the scan itself is in order is the decode call's business
    info.supported = true;
    return OGD_OK;
  }
  // the entropy-coded segment ends at the first marker that is not stuffing
  size_t e = p;
  uint64_t nstuff = 0;  // stuffed zero bytes so far
  const bool trusting = mode == PARSE_TRUSTING && restart == 0 && n >= p + 3 && f[n - 2] == 0xFF && f[n - 1] == 0xD9;
  if (trusting) {
    e = n - 2;
    info.scan_unchecked = true;
  }
  while (!trusting) {
    // 64 bytes at a time while they hold nothing but data and stuffed zeros (a scan has a 0xFF every ~50 bytes:
    // one memchr call per 0xFF made this walk 2 ms of a broadsheet page's 3); the two reductions vectorise
    while (e + 65 <= n) {
      unsigned marker = 0, stuffed = 0;
#if defined(__SSE2__)
      const __m128i all_ff = _mm_set1_epi8((char)0xFF), all_0 = _mm_setzero_si128();
      for (int i = 0; i < 64; i += 16) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(f + e + i));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(f + e + i + 1));
        const __m128i is_ff = _mm_cmpeq_epi8(a, all_ff), zero_next = _mm_cmpeq_epi8(b, all_0);
        marker |= (unsigned)_mm_movemask_epi8(_mm_andnot_si128(zero_next, is_ff));
        stuffed += (unsigned)__builtin_popcount((unsigned)_mm_movemask_epi8(_mm_and_si128(is_ff, zero_next)));
      }
#else
      for (int i = 0; i < 64; ++i) {
        const unsigned is_ff = f[e + i] == 0xFF, zero_next = f[e + i + 1] == 0x00;
        marker |= is_ff & (zero_next ^ 1u);
        stuffed += is_ff & zero_next;
      }
#endif
      if (marker) break;
      nstuff += stuffed;
      e += 64;
    }
    const uint8_t* ff = (const uint8_t*)std::memchr(f + e, 0xFF, n - e);
    OGD_REQUIRE(ff && (size_t)(ff - f) + 1 < n, "JPEG scan is not terminated by a marker");
    e = (size_t)(ff - f);
    if (f[e + 1] == 0x00) {
      e += 2;
      ++nstuff;
      continue;
    }
    if (restart > 0 && f[e + 1] >= 0xD0 && f[e + 1] <= 0xD7) {  // RSTn: the next interval begins behind it
      if ((int)(f[e + 1] - 0xD0) != (int)(info.rst.size() & 7)) JP_UNSUP("restart markers out of order");
      e += 2;
      info.rst.push_back(8 * ((uint64_t)(e - p) - nstuff - 2 * ((uint64_t)info.rst.size() + 1)));
      continue;
    }
    break;
  }
  if (f[e + 1] != 0xD9) JP_UNSUP("marker inside the scan");  // RSTn without DRI, further scans, fill bytes
  info.scan_begin = p;
  info.scan_end = e;
  info.restart = restart;
  DecTables& T = info.tabs;
  std::memset(&T, 0, sizeof T);
  // at most two DC and two AC tables in use (baseline)
  int dc_ids[2] = {-1, -1}, ac_ids[2] = {-1, -1};
  for (int c = 0; c < info.ncomp; ++c) {
    for (int k = 0; k < 64; ++k) T.quant[c][k] = qt[tq[c]][k];
    int sd = td[c] == dc_ids[0] ? 0 : td[c] == dc_ids[1] ? 1 : dc_ids[0] < 0 ? 0 : dc_ids[1] < 0 ? 1 : -1;
    int sa = ta[c] == ac_ids[0] ? 0 : ta[c] == ac_ids[1] ? 1 : ac_ids[0] < 0 ? 0 : ac_ids[1] < 0 ? 1 : -1;
    if (sd < 0 || sa < 0) JP_UNSUP("more than two Huffman tables of a class in use");
    dc_ids[sd] = td[c];
    ac_ids[sa] = ta[c];
    T.dc_sel[c] = (uint8_t)sd;
    T.ac_sel[c] = (uint8_t)sa;
  }
  for (int k = 0; k < 2; ++k) {
    if (dc_ids[k] >= 0)
      OGD_REQUIRE(build_dec_huff(hd[dc_ids[k]].bits, hd[dc_ids[k]].vals, hd[dc_ids[k]].n, T.dc[k]), "inconsistent Huffman table");
    if (ac_ids[k] >= 0)
      OGD_REQUIRE(build_dec_huff(ha[ac_ids[k]].bits, ha[ac_ids[k]].vals, ha[ac_ids[k]].n, T.ac[k]), "inconsistent Huffman table");
  }
  // a one-component scan is not interleaved: one block per MCU whatever its sampling factors say
  info.h0 = info.ncomp == 3 ? hs[0] : 1;
  info.v0 = info.ncomp == 3 ? vs[0] : 1;
  T.n0 = info.h0 * info.v0;
  T.bpm = info.ncomp == 3 ? T.n0 + 2 : 1;
  for (int m2 = 0; m2 < T.bpm; ++m2) {
    T.blk_comp[m2] = (uint8_t)(m2 < T.n0 ? 0 : m2 - T.n0 + 1);
    T.dc_mask |= (uint32_t)T.dc_sel[T.blk_comp[m2]] << m2;
    T.ac_mask |= (uint32_t)T.ac_sel[T.blk_comp[m2]] << m2;
  }
  const int64_t mcux = (info.W + 8 * info.h0 - 1) / (8 * info.h0), mcuy = (info.H + 8 * info.v0 - 1) / (8 * info.v0);
  const int64_t nmcu = mcux * mcuy;
  OGD_REQUIRE(nmcu * T.bpm < INT_MAX - 256, "too many blocks");
  T.base[0] = 0;
  T.base[1] = info.ncomp == 3 ? (int32_t)(nmcu * T.n0) : 0;
  T.base[2] = info.ncomp == 3 ? (int32_t)(nmcu * T.n0 + nmcu) : 0;
  T.ntot = (int32_t)(nmcu * T.bpm);
  if (restart > 0) {
    const int64_t nint = (nmcu + restart - 1) / restart;
    if ((int64_t)info.rst.size() + 1 != nint) JP_UNSUP("number of restart markers does not match the image");
  } else if (!info.rst.empty()) {
    JP_UNSUP("restart markers without a restart interval");
  }
  info.supported = true;
  return OGD_OK;
}

static int scan_u64(DevBuf& tmp, uint64_t* d, int count, hipStream_t st) {
  size_t bytes = 0;
  OGD_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, d, d, count, st));
  OGD_TRY(tmp.alloc(bytes));
  OGD_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, d, d, count, st));
  return OGD_OK;
}

// D6 (+ D7): coefficients and DC values of all blocks -> the "L" image
static int blocks_to_image(const JpegInfo& info, const DevBuf& dcoef, const DevBuf& ddc, const DecTables* tabs,
                           uint8_t* d_img, ptrdiff_t stride, hipStream_t st) {
  const int H = info.H, W = info.W;
  const int mcux = (W + 8 * info.h0 - 1) / (8 * info.h0);
  const int64_t nblk = info.tabs.ntot;
  const dim3 blk(256);
  DevBuf dplanes;
  Planes pl;
  std::memset(&pl, 0, sizeof pl);
  pl.mcux = mcux;
  pl.h0 = info.h0;
  pl.v0 = info.v0;
  const int cw = (W + info.h0 - 1) / info.h0, ch = (H + info.v0 - 1) / info.v0;  // libjpeg's downsampled size
  if (info.ncomp == 1) {
    pl.p[0] = d_img;
    pl.stride[0] = stride;
    pl.W[0] = W;
    pl.H[0] = H;
  } else {
    const size_t ys = ((size_t)W + 15) & ~(size_t)15, cs = ((size_t)cw + 15) & ~(size_t)15;  // ycc_to_l_kernel loads dwords
    const size_t ysz = ys * H, csz = cs * ch;
    OGD_TRY(dplanes.alloc(ysz + 2 * csz));
    pl.p[0] = dplanes.as<uint8_t>();
    pl.p[1] = pl.p[0] + ysz;
    pl.p[2] = pl.p[1] + csz;
    pl.stride[0] = (ptrdiff_t)ys;
    pl.stride[1] = pl.stride[2] = (ptrdiff_t)cs;
    pl.W[0] = W;
    pl.H[0] = H;
    pl.W[1] = pl.W[2] = cw;
    pl.H[1] = pl.H[2] = ch;
  }
  hipLaunchKernelGGL(idct_kernel, dim3((unsigned)((nblk + 255) / 256)), blk, 0, st, dcoef.as<uint4>(), ddc.as<int32_t>(),
                     tabs, pl);
  OGD_TRY(check_launch("idct_kernel"));
  if (info.ncomp == 3) {
    OGD_REQUIRE(H <= 65535, "image too tall");
    hipLaunchKernelGGL(ycc_to_l_kernel, dim3((unsigned)((W + 1023) / 1024), (unsigned)H), blk, 0, st, pl.p[0], pl.stride[0],
                       pl.p[1], pl.p[2], pl.stride[1], cw, ch, info.h0, info.v0, H, W, d_img, stride);
    OGD_TRY(check_launch("ycc_to_l_kernel"));
  }
  OGD_HIP(hipStreamSynchronize(st));  // dplanes goes back to the block cache on return
  return OGD_OK;
}

static int decode_to_device(const uint8_t* file, const JpegInfo& info, uint8_t* d_img, ptrdiff_t stride, hipStream_t st) {
  const int H = info.H, W = info.W;
  const int mcux = (W + 8 * info.h0 - 1) / (8 * info.h0), mcuy = (H + 8 * info.v0 - 1) / (8 * info.v0);
  const int64_t nmcu = (int64_t)mcux * mcuy;
  const int64_t nblk = info.tabs.ntot;  // coefficient slots = blocks of all components
  const uint64_t nscan = info.scan_end - info.scan_begin;
  OGD_REQUIRE(nscan >= 1 && nscan < ((uint64_t)1 << 40), "empty JPEG scan");

  // D1
  const int64_t nun64 = (int64_t)((nscan + UN_CHUNK - 1) / UN_CHUNK);
  OGD_REQUIRE(nun64 < INT_MAX - 1, "JPEG scan too long");
  const int nun = (int)nun64;
  DevBuf draw, dcnt, dtmp1, dstream, dtabs;
  OGD_TRY(draw.alloc((size_t)nscan + 16));
  OGD_TRY(dcnt.alloc(((size_t)nun + 1) * sizeof(uint64_t)));
  OGD_TRY(dstream.alloc((size_t)nscan + 64));
  OGD_TRY(dtabs.alloc(sizeof(DecTables)));
  OGD_HIP(hipMemcpyAsync(draw.p, file + info.scan_begin, (size_t)nscan, hipMemcpyHostToDevice, st));
  OGD_HIP(hipMemcpyAsync(dtabs.p, &info.tabs, sizeof(DecTables), hipMemcpyHostToDevice, st));
  OGD_HIP(hipMemsetAsync(dstream.p, 0, (size_t)nscan + 64, st));
  DevBuf dmarkers;
  if (info.scan_unchecked) {
    OGD_TRY(dmarkers.alloc(16));
    OGD_HIP(hipMemsetAsync(dmarkers.p, 0, 4, st));
  }
  hipLaunchKernelGGL(unstuff_count_kernel, dim3((unsigned)nun + 1), dim3(256), 0, st, draw.as<uint8_t>(), nscan, nun,
                     dcnt.as<uint64_t>(), info.scan_unchecked ? dmarkers.as<int>() : (int*)nullptr);
  OGD_TRY(check_launch("unstuff_count_kernel"));
  OGD_TRY(scan_u64(dtmp1, dcnt.as<uint64_t>(), nun + 1, st));
  hipLaunchKernelGGL(unstuff_compact_kernel, dim3((unsigned)nun), dim3(256), 0, st, draw.as<uint8_t>(), nscan,
                     dcnt.as<uint64_t>(), dstream.as<uint8_t>());
  OGD_TRY(check_launch("unstuff_compact_kernel"));
  uint64_t removed = 0;
  int markers = 0;
  OGD_HIP(hipMemcpyAsync(&removed, dcnt.as<uint64_t>() + nun, sizeof removed, hipMemcpyDeviceToHost, st));
  if (info.scan_unchecked) OGD_HIP(hipMemcpyAsync(&markers, dmarkers.p, sizeof markers, hipMemcpyDeviceToHost, st));
  OGD_HIP(hipStreamSynchronize(st));
  if (markers) return SCAN_HAS_MARKERS;  // not an error: the caller parses the file again, walking the scan
  const uint64_t total_bits = (nscan - removed) * 8;

  // D2
  const int64_t nchunk64 = (int64_t)((total_bits + CHUNK_BITS - 1) / CHUNK_BITS);
  OGD_REQUIRE(nchunk64 >= 1 && nchunk64 < INT_MAX - 256, "JPEG scan too long");
  const int nchunk = (int)nchunk64;
  DevBuf dentry, dexit, dnb, dflag, dtmp2, ddirty;
  OGD_TRY(dentry.alloc((size_t)nchunk * sizeof(ChunkState)));
  OGD_TRY(dexit.alloc((size_t)nchunk * sizeof(ChunkState)));
  OGD_TRY(dnb.alloc(((size_t)nchunk + 1) * sizeof(uint64_t)));
  OGD_TRY(dflag.alloc(16));
  OGD_TRY(ddirty.alloc(2 * (size_t)nchunk));
  uint8_t* const dirty = ddirty.as<uint8_t>();
  uint8_t* const touched = dirty + nchunk;
  const dim3 cgrid((unsigned)((nchunk + 255) / 256)), cgrid1((unsigned)((nchunk + 1 + 255) / 256)), blk(256);
  const DecTables* tabs = dtabs.as<DecTables>();
  const bool multi = info.tabs.bpm > 1;
  const bool rst = info.restart > 0 && !info.rst.empty();  // a DRI file of one interval is an ordinary scan
  DevBuf dper;  // period of the bit string around each chunk (blank paper), for the state passes
  {
    OGD_TRY(dper.alloc((size_t)nchunk + 16));
    hipLaunchKernelGGL(period_kernel, dim3((unsigned)((nchunk + 16 + 255) / 256)), blk, 0, st, dstream.as<uint8_t>(),
                       total_bits, nchunk, dper.as<uint8_t>());
    OGD_TRY(check_launch("period_kernel"));
  }
  hipLaunchKernelGGL(init_states_kernel, cgrid1, blk, 0, st, nchunk, dentry.as<ChunkState>(), dexit.as<ChunkState>(),
                     dnb.as<uint64_t>(), dirty, touched, dper.as<uint8_t>());
  OGD_TRY(check_launch("init_states_kernel"));
  const auto state_pass = multi ? (rst ? &decode_chunks_kernel<false, true, true> : &decode_chunks_kernel<false, true, false>)
                                : (rst ? &decode_chunks_kernel<false, false, true> : &decode_chunks_kernel<false, false, false>);
  const auto write_pass = multi ? (rst ? &decode_chunks_kernel<true, true, true> : &decode_chunks_kernel<true, true, false>)
                                : (rst ? &decode_chunks_kernel<true, false, true> : &decode_chunks_kernel<true, false, false>);
  DevBuf drst, dcheck;
  const int nrst = rst ? (int)info.rst.size() : 0;
  if (rst) {
    OGD_TRY(drst.alloc(info.rst.size() * sizeof(uint64_t)));
    OGD_TRY(dcheck.alloc(16));
    OGD_HIP(hipMemcpyAsync(drst.p, info.rst.data(), info.rst.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    OGD_HIP(hipMemsetAsync(dcheck.p, 0, 8, st));
  }
  const uint64_t* rst_bits = drst.as<uint64_t>();
  int rounds = 0;
  // A stream whose period is longer than the lattice steps cover (a synthetic pattern: every block the same ~700
  // bits) is walked by single lanes, chunk after chunk: slower than libjpeg on a host core.  Time allowance for the
  // rounds: twice what libjpeg would need for the whole file (~1 ns per sample + 3 ns per scan byte) + 100 ms; past
  // it the file is handed back (a 12 MP 1-pixel checkerboard would take 2.5 s).  Blank paper does not come here:
  // its period is a few bits (period_kernel).
  const auto t_rounds = std::chrono::steady_clock::now();
  const double allowance_ms = 100.0 + 2e-6 * (1.0 * (double)H * W * info.ncomp + 3.0 * (double)nscan);
  while (true) {
    OGD_HIP(hipMemsetAsync(dflag.p, 0, 4, st));
    hipLaunchKernelGGL(state_pass, cgrid, blk, 0, st, dstream.as<uint8_t>(), total_bits, nchunk, tabs,
                       dentry.as<ChunkState>(), dexit.as<ChunkState>(), dnb.as<uint64_t>(), dflag.as<int>(),
                       (const uint64_t*)nullptr, nmcu, (int16_t*)nullptr, (int32_t*)nullptr,
                       rounds < 3 ? 0 : std::min(2048 * (1024 / CHUNK_BITS), (4 * (1024 / CHUNK_BITS)) << std::min(rounds - 3, 12)), rst_bits, nrst, info.restart,
                       (int*)nullptr, dirty, touched, dper.as<uint8_t>());
    OGD_TRY(check_launch("decode_chunks_kernel"));
    int changed = 0;
    OGD_HIP(hipMemcpyAsync(&changed, dflag.p, sizeof changed, hipMemcpyDeviceToHost, st));
    OGD_HIP(hipStreamSynchronize(st));
    ++rounds;
    if (!changed) break;
    // Chunk r is exact after round r, so this ends after nchunk rounds at the latest; real scans settle in 3 - 4.
    // A scan that has not settled after 1024 rounds is damaged or adversarial: leave it to the caller's decoder.
    if (rounds > 1024 || rounds > nchunk + 1)
      return fail(OGD_ERR_RUNTIME, "JPEG chunk states did not settle in %d rounds", rounds);
    if (rounds >= 6) {
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_rounds).count();
      if (ms > allowance_ms)
        return fail(OGD_ERR_RUNTIME, "JPEG chunk states settle too slowly (periodic stream): %d rounds in %.0f ms", rounds, ms);
    }
    hipLaunchKernelGGL(shift_states_kernel, cgrid, blk, 0, st, dexit.as<ChunkState>(), nchunk, dentry.as<ChunkState>(),
                       dirty, touched);
    OGD_TRY(check_launch("shift_states_kernel"));
    if (std::getenv("OGD_JPEG_DEBUG") && std::atoi(std::getenv("OGD_JPEG_DEBUG")) > 1) {  // chunks to walk next round
      std::vector<uint8_t> hd((size_t)nchunk);
      OGD_HIP(hipMemcpyAsync(hd.data(), dirty, (size_t)nchunk, hipMemcpyDeviceToHost, st));
      OGD_HIP(hipStreamSynchronize(st));
      int64_t nd = 0, first = -1, last = -1;
      for (int64_t c = 0; c < nchunk; ++c)
        if (hd[(size_t)c]) {
          ++nd;
          if (first < 0) first = c;
          last = c;
        }
      fprintf(stderr, "ogd_jpeg_decode: after round %d: %lld chunks to walk (%lld .. %lld)\n", rounds, (long long)nd, (long long)first, (long long)last);
    }
  }
  // the last round ran from entry = shifted exits of the round before and reproduced them: entry is exact
  if (std::getenv("OGD_JPEG_DEBUG")) fprintf(stderr, "ogd_jpeg_decode: %d chunks settled in %d rounds\n", nchunk, rounds);

  // D3
  OGD_TRY(scan_u64(dtmp2, dnb.as<uint64_t>(), nchunk + 1, st));
  uint64_t blocks_found = 0;
  OGD_HIP(hipMemcpyAsync(&blocks_found, dnb.as<uint64_t>() + nchunk, sizeof blocks_found, hipMemcpyDeviceToHost, st));
  OGD_HIP(hipStreamSynchronize(st));
  if (blocks_found < (uint64_t)nblk)
    return fail(OGD_ERR_RUNTIME, "JPEG scan holds %llu blocks, the image needs %lld", (unsigned long long)blocks_found,
                (long long)nblk);

  // D4, D5, D6
  DevBuf dcoef, ddc, dtmp3;
  OGD_TRY(dcoef.alloc((size_t)nblk * 64 * sizeof(int16_t)));
  OGD_TRY(ddc.alloc((size_t)nblk * sizeof(int32_t)));
  OGD_HIP(hipMemsetAsync(dcoef.p, 0, (size_t)nblk * 64 * sizeof(int16_t), st));
  hipLaunchKernelGGL(write_pass, cgrid, blk, 0, st, dstream.as<uint8_t>(), total_bits, nchunk, tabs,
                     dentry.as<ChunkState>(), (ChunkState*)nullptr, (uint64_t*)nullptr, (int*)nullptr,
                     dnb.as<uint64_t>(), nmcu, dcoef.as<int16_t>(), ddc.as<int32_t>(), 0, rst_bits, nrst, info.restart,
                     dcheck.as<int>(), (const uint8_t*)nullptr, (uint8_t*)nullptr, (const uint8_t*)nullptr);
  OGD_TRY(check_launch("decode_chunks_kernel<write>"));
  // D5: DC prediction runs per component, in stream order: one inclusive scan per component's slots
  {
    const int ncomp = info.ncomp;
    const int32_t starts[4] = {info.tabs.base[0], ncomp == 3 ? info.tabs.base[1] : (int32_t)nblk,
                               ncomp == 3 ? info.tabs.base[2] : (int32_t)nblk, (int32_t)nblk};
    size_t sb = 0;
    OGD_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, sb, ddc.as<int32_t>(), ddc.as<int32_t>(), (int)nblk, st));
    OGD_TRY(dtmp3.alloc(sb));
    for (int c = 0; c < ncomp; ++c) {
      const int cnt = starts[c + 1] - starts[c];
      size_t sbc = sb;
      OGD_HIP(hipcub::DeviceScan::InclusiveSum(dtmp3.p, sbc, ddc.as<int32_t>() + starts[c], ddc.as<int32_t>() + starts[c], cnt, st));
    }
  }
  DevBuf ddc2;
  if (rst) {
    int check[2] = {0, 0};
    OGD_HIP(hipMemcpyAsync(check, dcheck.p, sizeof check, hipMemcpyDeviceToHost, st));
    OGD_TRY(ddc2.alloc((size_t)nblk * sizeof(int32_t)));
    hipLaunchKernelGGL(dc_restart_fix_kernel, dim3((unsigned)((nblk + 255) / 256)), blk, 0, st, ddc.as<int32_t>(),
                       ddc2.as<int32_t>(), (int)nblk, info.tabs.base[1], info.tabs.base[2], info.restart * info.tabs.n0,
                       info.restart);
    OGD_TRY(check_launch("dc_restart_fix_kernel"));
    OGD_HIP(hipStreamSynchronize(st));
    if (std::getenv("OGD_JPEG_DEBUG")) fprintf(stderr, "ogd_jpeg_decode: restarts taken %d of %d, misplaced %d\n", check[0], nrst, check[1]);
    if (check[0] != nrst || check[1])
      return fail(OGD_ERR_RUNTIME, "JPEG scan: %d of %d restart intervals begin where their markers stand%s", check[0], nrst,
                  check[1] ? ", not all after the right number of blocks" : "");
  }
  OGD_TRY(blocks_to_image(info, dcoef, rst ? ddc2 : ddc, tabs, d_img, stride, st));
  OGD_HIP(hipStreamSynchronize(st));  // the scratch buffers go back to the block cache on return
  return OGD_OK;
}

}  // namespace
// For ogd_jpeg_dewarp_jpeg_gray (ogd_jpeg.hip): parse ONCE (finding the end of a broadsheet scan is a walk over
// every 0xFF of 6 MB, 1.8 ms), then decode into a buffer of the right size.  OGD_ERR_UNSUPPORTED as ogd_jpeg_info says.
// Parse (trusting the file's end first, see parse_jpeg), check the caller's geometry if it has one (H0 > 0), decode
// into d_img, or into dimg (allocated here) if d_img is null.
static int parse_and_decode(const uint8_t* file, size_t n, int H0, int W0, ptrdiff_t stride, uint8_t* d_img, DevBuf* dimg,
                            int* H, int* W, hipStream_t st) {
  for (const ParseMode mode : {PARSE_TRUSTING, PARSE_FULL}) {
    JpegInfo info;
    OGD_TRY(parse_jpeg(file, n, info, mode));
    if (!info.supported) return fail(OGD_ERR_UNSUPPORTED, "JPEG decoder: %s", info.why);
    if (H0 > 0) {
      OGD_REQUIRE(H0 == info.H && W0 == info.W, "output is %dx%d, the file holds %dx%d", W0, H0, info.W, info.H);
      OGD_REQUIRE(stride >= W0, "row stride %td < %d bytes", stride, W0);
    }
    OGD_TRY(ensure_device());
    uint8_t* target = d_img;
    if (target == nullptr) {
      // the careful pass after a trusting one finds the same H x W in the same headers: keep the block
      if (dimg->p == nullptr) OGD_TRY(dimg->alloc((size_t)info.H * info.W));
      target = dimg->as<uint8_t>();
      stride = info.W;
    }
    if (H) *H = info.H;
    if (W) *W = info.W;
    const int status = decode_to_device(file, info, target, stride, st);
    if (status != SCAN_HAS_MARKERS) return status;
  }
  return fail(OGD_ERR_RUNTIME, "JPEG decoder: the scan's markers could not be settled");  // PARSE_FULL never trusts
}

int jpeg_decode_gray_to(DevBuf& dimg, const uint8_t* file, size_t n, int* H, int* W, void* stream) {
  return parse_and_decode(file, n, 0, 0, 0, nullptr, &dimg, H, W, as_stream(stream));
}

}  // namespace ogd

using namespace ogd;

extern "C" {

int ogd_jpeg_info(const uint8_t* file, size_t n, int* H, int* W, int* ncomp, int* supported) {
  OGD_REQUIRE(file && H && W && ncomp && supported, "null pointer");
  JpegInfo info;
  OGD_TRY(parse_jpeg(file, n, info));
  *H = info.H;
  *W = info.W;
  *ncomp = info.ncomp;
  *supported = info.supported ? 1 : 0;
  if (!info.supported) set_error("JPEG decoder: %s", info.why);
  return OGD_OK;
}

int ogd_jpeg_header(const uint8_t* file, size_t n, int* H, int* W, int* ncomp, int* plausible) {
  OGD_REQUIRE(file && H && W && ncomp && plausible, "null pointer");
  JpegInfo info;
  OGD_TRY(parse_jpeg(file, n, info, PARSE_HEADERS));
  *H = info.H;
  *W = info.W;
  *ncomp = info.ncomp;
  *plausible = info.supported ? 1 : 0;
  if (!info.supported) set_error("JPEG decoder: %s", info.why);
  return OGD_OK;
}

int ogd_jpeg_decode_gray(const uint8_t* file, size_t n, uint8_t* out, int H, int W, ptrdiff_t stride) {
  OGD_REQUIRE(file && out, "null pointer");
  OGD_REQUIRE(H > 0 && W > 0, "output is %dx%d", W, H);
  OGD_REQUIRE(stride >= W, "row stride %td < %d bytes", stride, W);
  DevBuf dimg;
  OGD_TRY(parse_and_decode(file, n, H, W, W, nullptr, &dimg, nullptr, nullptr, nullptr));
  OGD_HIP(copy_rows(out, (size_t)stride, dimg.p, (size_t)W, (size_t)W, (size_t)H, hipMemcpyDeviceToHost));
  return OGD_OK;
}

int ogd_jpeg_decode_gray_dev(const uint8_t* file, size_t n, uint8_t* d_out, int H, int W, ptrdiff_t stride, void* stream) {
  OGD_REQUIRE(file && d_out, "null pointer");
  OGD_REQUIRE(H > 0 && W > 0, "output is %dx%d", W, H);
  return parse_and_decode(file, n, H, W, stride, d_out, nullptr, nullptr, nullptr, as_stream(stream));
}

}  // extern "C"
