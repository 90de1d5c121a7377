// N4 — the JPEG file of a dewarped page (ref: origami/batch/core/io.py:606-610, `image.save(path)` of an
// "L" image to annotation.<name>.jpg, called from origami/batch/utils/dewarp.py:27-28): baseline JPEG,
// one 8-bit component, the two standard luminance Huffman tables, no restart markers — byte for byte the
// file PIL / libjpeg-turbo writes for the same pixels and quality (checked in tests/ against PIL itself).
//
// Restated from the published baseline algorithm (ITU T.81) with libjpeg's conventions:
//   samples - 128 -> forward DCT "islow" (13-bit constants, 2 extra bits after the row pass, output scaled
//   by 8) -> q = (|c| + d/2) / d with d = 8 * table entry, sign restored -> zig-zag -> DC difference /
//   (run, size) symbols -> Huffman codes + magnitude bits, MSB first -> 0xFF byte stuffing, last byte
//   padded with ones.  Blocks beyond the right / bottom edge repeat the last column / row.
//
// The entropy coder is sequential only in its bit positions, so it is a scan:
//   J1 dct_quant_kernel    thread per 8x8 block: pixels -> 64 quantised coefficients (int16, zig-zag order)
//                          + the length in bits of the block's AC code
//   J2 block_bits_kernel   thread per block: + the length of its DC code (needs the previous block's DC)
//      exclusive scan of the lengths (hipCUB)
//   J3 emit_kernel         thread per block: writes its codes at its bit offset (atomicOr on the two words
//                          it may share with its neighbours, plain stores between)
//   J4 ff_count / ff_expand  0xFF -> 0xFF 0x00: count per 4 KB chunk, scan, expand
// Coefficients are stored as [8 groups of 8][block] (16 bytes per group and block), so that the threads of a
// wavefront — consecutive blocks — read and write consecutive 16-byte pieces.
// HBM traffic: 1 B/px read, 2 B/px coefficients written and read once, ~0.14 B/px of code.
#include "ogd_common.h"
#include <hipcub/hipcub.hpp>
#include <cstring>
#include <string>
#include <vector>

namespace ogd {

static const uint8_t STD_LUMA_Q[64] = {
    16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
    14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
    18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};

static const uint8_t ZIGZAG[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static const uint8_t DC_BITS[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t DC_VALS[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t AC_BITS[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t AC_VALS[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

// Kernel-argument tables.  quant: divisor d = 8 * q and M = floor(2^32 / d) + 1 per zig-zag position k (for
// n < 2^17: n / d == umulhi(n, M)); huff: code << 8 | length per symbol (length 0: symbol not in the table).
struct QuantTab {
  uint16_t half[64];  // d >> 1
  uint32_t magic[64];
};
struct HuffTab {
  uint32_t dc[12];
  uint32_t ac[256];
};

static void derive_huff(const uint8_t* bits, const uint8_t* vals, int nvals, uint32_t* out, int nout) {
  for (int i = 0; i < nout; ++i) out[i] = 0;
  uint32_t code = 0;
  int k = 0;
  for (int len = 1; len <= 16; ++len) {
    for (int i = 0; i < bits[len - 1]; ++i, ++k) out[vals[k]] = (code++ << 8) | (uint32_t)len;
    code <<= 1;
  }
  (void)nvals;
}

static void quality_table(int quality, uint8_t q_nat[64]) {
  // libjpeg jpeg_quality_scaling + jpeg_add_quant_table(force_baseline)
  quality = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
  const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
  for (int i = 0; i < 64; ++i) {
    long t = ((long)STD_LUMA_Q[i] * scale + 50L) / 100L;
    q_nat[i] = (uint8_t)(t <= 0 ? 1 : (t > 255 ? 255 : t));
  }
}

// ---- J1 ----------------------------------------------------------------------------------------------

template <int PASS>
__device__ __forceinline__ void fdct8(int& d0, int& d1, int& d2, int& d3, int& d4, int& d5, int& d6, int& d7) {
  constexpr int CB = 13, P1 = 2;
  constexpr int SH_EVEN = PASS == 1 ? CB - P1 : CB + P1;
  const int t0 = d0 + d7, t7 = d0 - d7, t1 = d1 + d6, t6 = d1 - d6;
  const int t2 = d2 + d5, t5 = d2 - d5, t3 = d3 + d4, t4 = d3 - d4;
  const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  if (PASS == 1) {
    d0 = (t10 + t11) << P1;
    d4 = (t10 - t11) << P1;
  } else {
    d0 = (t10 + t11 + (1 << (P1 - 1))) >> P1;
    d4 = (t10 - t11 + (1 << (P1 - 1))) >> P1;
  }
  constexpr int R = 1 << (SH_EVEN - 1);
  const int z1e = (t12 + t13) * 4433;
  d2 = (z1e + t13 * 6270 + R) >> SH_EVEN;
  d6 = (z1e + t12 * (-15137) + R) >> SH_EVEN;
  int z1 = t4 + t7, z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
  const int z5 = (z3 + z4) * 9633;
  const int a4 = t4 * 2446, a5 = t5 * 16819, a6 = t6 * 25172, a7 = t7 * 12299;
  z1 *= -7373;
  z2 *= -20995;
  z3 = z3 * (-16069) + z5;
  z4 = z4 * (-3196) + z5;
  d7 = (a4 + z1 + z3 + R) >> SH_EVEN;
  d5 = (a5 + z2 + z4 + R) >> SH_EVEN;
  d3 = (a6 + z2 + z3 + R) >> SH_EVEN;
  d1 = (a7 + z1 + z4 + R) >> SH_EVEN;
}

typedef uint32_t jp_u32u __attribute__((aligned(1)));

__global__ __launch_bounds__(256) void dct_quant_kernel(const uint8_t* __restrict__ img, int H, int W, ptrdiff_t stride,
                                                        int nbx, int nblk, const QuantTab qt, const HuffTab ht,
                                                        uint4* __restrict__ coef, uint32_t* __restrict__ ac_bits) {
  __shared__ uint32_t s_ac[256];
  for (int i = threadIdx.x; i < 256; i += 256) s_ac[i] = ht.ac[i] & 255u;  // code lengths
  __syncthreads();
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= nblk) return;
  const int by = b / nbx, bx = b - by * nbx;
  const int x0 = bx * 8;
  int d[64];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const uint8_t* row = img + (ptrdiff_t)min(by * 8 + r, H - 1) * stride;
    if (x0 + 8 <= W) {
      const uint32_t lo = *reinterpret_cast<const jp_u32u*>(row + x0);
      const uint32_t hi = *reinterpret_cast<const jp_u32u*>(row + x0 + 4);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        d[r * 8 + c] = (int)((lo >> (8 * c)) & 255u) - 128;
        d[r * 8 + 4 + c] = (int)((hi >> (8 * c)) & 255u) - 128;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) d[r * 8 + c] = (int)row[min(x0 + c, W - 1)] - 128;
    }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r)
    fdct8<1>(d[r * 8], d[r * 8 + 1], d[r * 8 + 2], d[r * 8 + 3], d[r * 8 + 4], d[r * 8 + 5], d[r * 8 + 6], d[r * 8 + 7]);
#pragma unroll
  for (int c = 0; c < 8; ++c)
    fdct8<2>(d[c], d[8 + c], d[16 + c], d[24 + c], d[32 + c], d[40 + c], d[48 + c], d[56 + c]);
  // quantise in zig-zag order; eight int16 per 16-byte store
  constexpr uint8_t ZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                              41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                              30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  // ... and count the bits of the AC code on the way (the DC code needs the previous block: J2)
  uint32_t bits = 0;
  int run = 0;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = g * 8 + j;
      const int c = d[ZZ[k]];
      const uint32_t n = (uint32_t)abs(c) + qt.half[k];
      const int q = (int)__umulhi(n, qt.magic[k]);
      const uint32_t v = (uint32_t)(c < 0 ? -q : q) & 0xffffu;
      if (j & 1)
        w[j >> 1] |= v << 16;
      else
        w[j >> 1] = v;
      if (k > 0) {
        if (q == 0) {
          ++run;
        } else {
          const int nb = 32 - __clz(q);
          bits += (uint32_t)(run >> 4) * s_ac[0xF0] + s_ac[((run & 15) << 4) | nb] + (uint32_t)nb;
          run = 0;
        }
      }
    }
    coef[(size_t)g * nblk + b] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  if (run > 0) bits += s_ac[0];
  ac_bits[b] = bits;
}

// ---- J2 / J3: the block coder, once counting and once writing ----------------------------------------

// Appends MSB-first bit strings at a given bit offset of a zero-initialised big-endian stream.
struct BitWriter {
  uint32_t* words;
  uint64_t wi;   // current 32-bit word
  uint64_t acc;  // bits not yet written (low `n` bits), below the bits other blocks own in word wi
  int n;         // number of bits of word wi accounted for (foreign + own)
  bool first;
  __device__ __forceinline__ BitWriter(uint32_t* w, uint64_t bit_off)
      : words(w), wi(bit_off >> 5), acc(0), n((int)(bit_off & 31)), first(true) {}
  __device__ __forceinline__ void put(uint32_t code, int len) {  // len <= 27
    acc = (acc << len) | code;
    n += len;
    if (n >= 32) {
      const uint32_t word = (uint32_t)(acc >> (n - 32));
      const uint32_t be = __builtin_bswap32(word);
      if (first)
        atomicOr(words + wi, be);  // the head of the word may belong to the previous block
      else
        words[wi] = be;
      first = false;
      ++wi;
      n -= 32;
      acc &= (1ull << n) - 1ull;
    }
  }
  __device__ __forceinline__ void flush() {
    if (n > 0) {
      // with first == true the bits above ours in this word are foreign: acc holds only our own (low) bits
      const uint32_t word = (uint32_t)(acc << (32 - n));
      if (word) atomicOr(words + wi, __builtin_bswap32(word));
    }
  }
};

template <class Sink>
__device__ __forceinline__ void code_block(const uint4* __restrict__ p, size_t nblk, int prev_dc,
                                           const uint32_t* __restrict__ hdc, const uint32_t* __restrict__ hac, Sink& sink) {
  // 64 coefficients as eight 16-byte loads (group g of this block: p[g * nblk])
  int run = 0;
#pragma unroll 1
  for (int g = 0; g < 8; ++g) {
    const uint4 v = p[(size_t)g * nblk];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int val = (int)(int16_t)((w[j >> 1] >> (16 * (j & 1))) & 0xffffu);
      if (g == 0 && j == 0) {
        int t = val - prev_dc, t2 = t;
        if (t < 0) {
          t = -t;
          --t2;
        }
        const int nb = t ? 32 - __clz(t) : 0;
        const uint32_t h = hdc[nb];
        sink.put(((h >> 8) << nb) | ((uint32_t)t2 & ((1u << nb) - 1u)), (int)(h & 255u) + nb);
      } else if (val == 0) {
        ++run;
      } else {
        while (run > 15) {
          const uint32_t z = hac[0xF0];
          sink.put(z >> 8, (int)(z & 255u));
          run -= 16;
        }
        int t = val, t2 = val;
        if (t < 0) {
          t = -t;
          --t2;
        }
        const int nb = 32 - __clz(t);
        const uint32_t h = hac[(run << 4) | nb];
        sink.put(((h >> 8) << nb) | ((uint32_t)t2 & ((1u << nb) - 1u)), (int)(h & 255u) + nb);
        run = 0;
      }
    }
  }
  if (run > 0) {
    const uint32_t e = hac[0];
    sink.put(e >> 8, (int)(e & 255u));
  }
}

__device__ __forceinline__ void stage_huff(const HuffTab& ht, uint32_t* s_dc, uint32_t* s_ac) {
  for (int i = threadIdx.x; i < 256; i += blockDim.x) s_ac[i] = ht.ac[i];
  if (threadIdx.x < 12) s_dc[threadIdx.x] = ht.dc[threadIdx.x];
  __syncthreads();
}

__device__ __forceinline__ int dc_of(const uint4* __restrict__ coef, int b) { return (int)(int16_t)(coef[b].x & 0xffffu); }

__global__ __launch_bounds__(256) void block_bits_kernel(const uint4* __restrict__ coef, const uint32_t* __restrict__ ac_bits,
                                                         int nblk, const HuffTab ht, uint64_t* __restrict__ bits) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b > nblk) return;
  if (b == nblk) {  // sentinel: the scan leaves the total there
    bits[b] = 0;
    return;
  }
  const int t = abs(dc_of(coef, b) - (b ? dc_of(coef, b - 1) : 0));
  const int nb = t ? 32 - __clz(t) : 0;
  bits[b] = (uint64_t)(ac_bits[b] + (ht.dc[nb] & 255u) + (uint32_t)nb);
}

__global__ __launch_bounds__(256) void emit_kernel(const uint4* __restrict__ coef, int nblk, const HuffTab ht,
                                                   const uint64_t* __restrict__ bit_off, uint32_t* __restrict__ words) {
  __shared__ uint32_t s_dc[12], s_ac[256];
  stage_huff(ht, s_dc, s_ac);
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= nblk) return;
  const int prev = b ? dc_of(coef, b - 1) : 0;
  BitWriter wr(words, bit_off[b]);
  code_block(coef + b, (size_t)nblk, prev, s_dc, s_ac, wr);
  wr.flush();
}

// ---- J4: byte stuffing ---------------------------------------------------------------------------------
// raw: the code as bytes (nraw of them; the unused low bits of the last byte are padded with ones here).
// Chunks of 4096 bytes, 16 per thread.

constexpr int FF_CHUNK = 4096;

__device__ __forceinline__ uint4 load_raw16(const uint8_t* __restrict__ raw, uint64_t nraw, uint32_t pad_mask, uint64_t i) {
  // raw is allocated in whole chunks and zero beyond nraw
  uint4 v = *reinterpret_cast<const uint4*>(raw + i);
  if (nraw - 1 >= i && nraw - 1 < i + 16) {
    uint32_t* w = reinterpret_cast<uint32_t*>(&v);
    const int k = (int)(nraw - 1 - i);
    w[k >> 2] |= pad_mask << (8 * (k & 3));
  }
  return v;
}

__device__ __forceinline__ int count_ff(uint32_t w) {
  // bytes equal to 0xFF: all eight bits set
  uint32_t t = w & (w >> 4);  // low nibble of each byte: and of both nibbles
  t &= t >> 2;
  t &= t >> 1;
  t &= 0x01010101u;
  return __popc(t);
}

__global__ __launch_bounds__(256) void ff_count_kernel(const uint8_t* __restrict__ raw, uint64_t nraw, uint32_t pad_mask,
                                                       int nchunk, uint64_t* __restrict__ counts) {
  typedef hipcub::BlockReduce<int, 256> Reduce;
  __shared__ typename Reduce::TempStorage tmp;
  const int ch = blockIdx.x;
  int n = 0;
  if (ch < nchunk) {
    const uint64_t i = (uint64_t)ch * FF_CHUNK + threadIdx.x * 16;
    if (i < nraw) {
      const uint4 v = load_raw16(raw, nraw, pad_mask, i);
      n = count_ff(v.x) + count_ff(v.y) + count_ff(v.z) + count_ff(v.w);
    }
  }
  const int total = Reduce(tmp).Sum(n);
  if (threadIdx.x == 0) counts[ch] = ch < nchunk ? (uint64_t)total : 0;  // counts[nchunk] = 0: sentinel for the scan
}

__global__ __launch_bounds__(256) void ff_expand_kernel(const uint8_t* __restrict__ raw, uint64_t nraw, uint32_t pad_mask,
                                                        const uint64_t* __restrict__ chunk_ff, uint8_t* __restrict__ out) {
  typedef hipcub::BlockScan<int, 256> Scan;
  __shared__ typename Scan::TempStorage tmp;
  const uint64_t i = (uint64_t)blockIdx.x * FF_CHUNK + threadIdx.x * 16;
  uint4 v = make_uint4(0, 0, 0, 0);
  int n = 0;
  if (i < nraw) {
    v = load_raw16(raw, nraw, pad_mask, i);
    n = count_ff(v.x) + count_ff(v.y) + count_ff(v.z) + count_ff(v.w);
  }
  int before = 0;
  Scan(tmp).ExclusiveSum(n, before);
  if (i >= nraw) return;
  uint8_t* o = out + i + chunk_ff[blockIdx.x] + (uint64_t)before;
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  const int nb = (int)min((uint64_t)16, nraw - i);
  for (int k = 0; k < nb; ++k) {
    const uint8_t byte = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
    *o++ = byte;
    if (byte == 0xFF) *o++ = 0;
  }
}

// ---- host side -----------------------------------------------------------------------------------------

static void put16(std::vector<uint8_t>& v, int x) {
  v.push_back((uint8_t)(x >> 8));
  v.push_back((uint8_t)x);
}

// SOI, APP0 (JFIF 1.01, aspect 1:1), DQT, SOF0, DHT (DC 0, AC 0), SOS — the sequence libjpeg writes for a
// one-component baseline image.
static std::vector<uint8_t> jpeg_header(int H, int W, const uint8_t q_nat[64]) {
  std::vector<uint8_t> h;
  h.reserve(400);
  put16(h, 0xFFD8);
  put16(h, 0xFFE0);
  put16(h, 16);
  const uint8_t jfif[] = {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
  h.insert(h.end(), jfif, jfif + sizeof jfif);
  put16(h, 0xFFDB);
  put16(h, 67);
  h.push_back(0);
  for (int k = 0; k < 64; ++k) h.push_back(q_nat[ZIGZAG[k]]);
  put16(h, 0xFFC0);
  put16(h, 11);
  h.push_back(8);
  put16(h, H);
  put16(h, W);
  h.push_back(1);
  h.push_back(1);
  h.push_back(0x11);
  h.push_back(0);
  put16(h, 0xFFC4);
  put16(h, 2 + 1 + 16 + 12);
  h.push_back(0x00);
  h.insert(h.end(), DC_BITS, DC_BITS + 16);
  h.insert(h.end(), DC_VALS, DC_VALS + 12);
  put16(h, 0xFFC4);
  put16(h, 2 + 1 + 16 + 162);
  h.push_back(0x10);
  h.insert(h.end(), AC_BITS, AC_BITS + 16);
  h.insert(h.end(), AC_VALS, AC_VALS + 162);
  put16(h, 0xFFDA);
  put16(h, 8);
  h.push_back(1);
  h.push_back(1);
  h.push_back(0x00);
  h.push_back(0);
  h.push_back(63);
  h.push_back(0);
  return h;
}

static int jpeg_encode_dev(const uint8_t* d_img, int H, int W, ptrdiff_t stride, int quality, uint8_t* out, size_t cap,
                           size_t* out_size, hipStream_t st) {
  OGD_REQUIRE(d_img && out_size && (out || cap == 0), "null pointer");
  OGD_REQUIRE(H >= 1 && W >= 1 && H <= 65535 && W <= 65535, "JPEG size %dx%d outside 1..65535", W, H);
  OGD_REQUIRE(stride >= W, "row stride %td < %d bytes", stride, W);
  OGD_REQUIRE(quality >= 1 && quality <= 100, "quality %d outside 1..100", quality);
  OGD_TRY(ensure_device());
  const int nbx = (W + 7) / 8, nby = (H + 7) / 8;
  const int64_t nblk64 = (int64_t)nbx * nby;
  OGD_REQUIRE(nblk64 < (int64_t)INT_MAX - 256, "too many blocks");
  const int nblk = (int)nblk64;

  uint8_t q_nat[64];
  quality_table(quality, q_nat);
  QuantTab qt;
  for (int k = 0; k < 64; ++k) {
    const uint32_t d = 8u * q_nat[ZIGZAG[k]];
    qt.half[k] = (uint16_t)(d >> 1);
    qt.magic[k] = (uint32_t)((1ull << 32) / d) + 1u;
  }
  HuffTab ht;
  derive_huff(DC_BITS, DC_VALS, 12, ht.dc, 12);
  derive_huff(AC_BITS, AC_VALS, 162, ht.ac, 256);

  DevBuf dcoef, dacb, dbits, dscan;
  OGD_TRY(dcoef.alloc((size_t)nblk * 64 * sizeof(int16_t)));
  OGD_TRY(dacb.alloc((size_t)nblk * sizeof(uint32_t)));
  OGD_TRY(dbits.alloc(((size_t)nblk + 1) * sizeof(uint64_t)));
  const unsigned gblk = (unsigned)((nblk + 255) / 256), gblk1 = (unsigned)((nblk + 1 + 255) / 256);
  hipLaunchKernelGGL(dct_quant_kernel, dim3(gblk), dim3(256), 0, st, d_img, H, W, stride, nbx, nblk, qt, ht,
                     dcoef.as<uint4>(), dacb.as<uint32_t>());
  OGD_TRY(check_launch("dct_quant_kernel"));
  hipLaunchKernelGGL(block_bits_kernel, dim3(gblk1), dim3(256), 0, st, dcoef.as<uint4>(), dacb.as<uint32_t>(), nblk, ht,
                     dbits.as<uint64_t>());
  OGD_TRY(check_launch("block_bits_kernel"));
  size_t scan_bytes = 0;
  OGD_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, dbits.as<uint64_t>(), dbits.as<uint64_t>(), nblk + 1, st));
  OGD_TRY(dscan.alloc(scan_bytes));
  OGD_HIP(hipcub::DeviceScan::ExclusiveSum(dscan.p, scan_bytes, dbits.as<uint64_t>(), dbits.as<uint64_t>(), nblk + 1, st));
  uint64_t total_bits = 0;
  OGD_HIP(hipMemcpyAsync(&total_bits, dbits.as<uint64_t>() + nblk, sizeof total_bits, hipMemcpyDeviceToHost, st));
  OGD_HIP(hipStreamSynchronize(st));

  const uint64_t nraw = (total_bits + 7) / 8;
  const uint32_t pad_mask = (total_bits & 7) ? (1u << (8 - (total_bits & 7))) - 1u : 0u;
  const int64_t nchunk64 = (int64_t)((nraw + FF_CHUNK - 1) / FF_CHUNK);
  OGD_REQUIRE(nchunk64 >= 1 && nchunk64 < INT_MAX - 1, "code stream of %llu bytes", (unsigned long long)nraw);
  const int nchunk = (int)nchunk64;
  DevBuf draw, dcnt, dscan2, dout;
  const size_t raw_alloc = (size_t)nchunk * FF_CHUNK + 16;
  OGD_TRY(draw.alloc(raw_alloc));
  OGD_TRY(dcnt.alloc(((size_t)nchunk + 1) * sizeof(uint64_t)));
  OGD_HIP(hipMemsetAsync(draw.p, 0, raw_alloc, st));
  hipLaunchKernelGGL(emit_kernel, dim3(gblk), dim3(256), 0, st, dcoef.as<uint4>(), nblk, ht, dbits.as<uint64_t>(),
                     draw.as<uint32_t>());
  OGD_TRY(check_launch("emit_kernel"));
  hipLaunchKernelGGL(ff_count_kernel, dim3((unsigned)nchunk + 1), dim3(256), 0, st, draw.as<uint8_t>(), nraw, pad_mask,
                     nchunk, dcnt.as<uint64_t>());
  OGD_TRY(check_launch("ff_count_kernel"));
  OGD_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, dcnt.as<uint64_t>(), dcnt.as<uint64_t>(), nchunk + 1, st));
  OGD_TRY(dscan2.alloc(scan_bytes));
  OGD_HIP(hipcub::DeviceScan::ExclusiveSum(dscan2.p, scan_bytes, dcnt.as<uint64_t>(), dcnt.as<uint64_t>(), nchunk + 1, st));
  uint64_t total_ff = 0;
  OGD_HIP(hipMemcpyAsync(&total_ff, dcnt.as<uint64_t>() + nchunk, sizeof total_ff, hipMemcpyDeviceToHost, st));
  OGD_HIP(hipStreamSynchronize(st));

  const std::vector<uint8_t> head = jpeg_header(H, W, q_nat);
  const uint64_t nscan = nraw + total_ff;
  const uint64_t need = head.size() + nscan + 2;
  *out_size = (size_t)need;
  if (need > cap) return fail(OGD_ERR_INVALID, "JPEG needs %llu bytes, the buffer holds %zu", (unsigned long long)need, cap);
  OGD_TRY(dout.alloc((size_t)nscan + 16));
  hipLaunchKernelGGL(ff_expand_kernel, dim3((unsigned)nchunk), dim3(256), 0, st, draw.as<uint8_t>(), nraw, pad_mask,
                     dcnt.as<uint64_t>(), dout.as<uint8_t>());
  OGD_TRY(check_launch("ff_expand_kernel"));
  std::memcpy(out, head.data(), head.size());
  OGD_HIP(hipMemcpyAsync(out + head.size(), dout.p, (size_t)nscan, hipMemcpyDeviceToHost, st));
  OGD_HIP(hipStreamSynchronize(st));
  out[head.size() + nscan] = 0xFF;
  out[head.size() + nscan + 1] = 0xD9;
  return OGD_OK;
}

}  // namespace ogd

using namespace ogd;

extern "C" {

int ogd_jpeg_encode_gray_dev(const uint8_t* d_img, int H, int W, ptrdiff_t stride, int quality, uint8_t* out,
                             size_t cap, size_t* out_size, void* stream) {
  return jpeg_encode_dev(d_img, H, W, stride, quality, out, cap, out_size, as_stream(stream));
}

int ogd_jpeg_encode_gray(const uint8_t* img, int H, int W, ptrdiff_t stride, int quality, uint8_t* out, size_t cap,
                         size_t* out_size) {
  OGD_REQUIRE(img && out_size, "null pointer");
  OGD_REQUIRE(H >= 1 && W >= 1 && H <= 65535 && W <= 65535, "JPEG size %dx%d outside 1..65535", W, H);
  OGD_REQUIRE(stride >= W, "row stride %td < %d bytes", stride, W);
  OGD_TRY(ensure_device());
  DevBuf dimg;
  OGD_TRY(dimg.alloc((size_t)H * W));
  OGD_HIP(copy_rows(dimg.p, (size_t)W, img, (size_t)stride, (size_t)W, (size_t)H, hipMemcpyHostToDevice));
  return jpeg_encode_dev(dimg.as<uint8_t>(), H, W, W, quality, out, cap, out_size, nullptr);
}

int ogd_dewarp_jpeg_gray(const uint8_t* src, int H, int W, ptrdiff_t src_stride, const float* nodes, int h, int w,
                         int s, int quality, uint8_t* out, size_t cap, size_t* out_size) {
  OGD_REQUIRE(src && nodes && out_size, "null pointer");
  OGD_REQUIRE(h >= 1 && w >= 1 && s >= 1, "bad node grid %dx%d cell %d", h, w, s);
  OGD_REQUIRE((int64_t)h * s < 32768 && (int64_t)w * s < 32768, "dense grid too large");
  OGD_REQUIRE(H >= 1 && W >= 1 && src_stride >= W, "bad source image %dx%d stride %td", W, H, src_stride);
  OGD_TRY(ensure_device());
  const int Hd = h * s, Wd = w * s;
  DevBuf dsrc, dnodes, ddst;
  OGD_TRY(dsrc.alloc((size_t)H * W));
  OGD_TRY(dnodes.alloc((size_t)h * w * 2 * sizeof(float)));
  OGD_TRY(ddst.alloc((size_t)Hd * Wd));
  OGD_HIP(copy_rows(dsrc.p, (size_t)W, src, (size_t)src_stride, (size_t)W, (size_t)H, hipMemcpyHostToDevice));
  OGD_HIP(hipMemcpy(dnodes.p, nodes, (size_t)h * w * 2 * sizeof(float), hipMemcpyHostToDevice));
  OGD_TRY(ogd_dewarp_u8_dev(dsrc.as<uint8_t>(), H, W, 1, W, dnodes.as<float>(), h, w, s, 1.f, 1.f, ddst.as<uint8_t>(),
                            Wd, OGD_INTER_LINEAR, nullptr));
  return jpeg_encode_dev(ddst.as<uint8_t>(), Hd, Wd, Wd, quality, out, cap, out_size, nullptr);
}

int ogd_jpeg_dewarp_jpeg_gray(const uint8_t* file, size_t n, const float* nodes, int h, int w, int s, int quality,
                              uint8_t* out, size_t cap, size_t* out_size) {
  OGD_REQUIRE(file && nodes && out_size, "null pointer");
  OGD_REQUIRE(h >= 1 && w >= 1 && s >= 1, "bad node grid %dx%d cell %d", h, w, s);
  OGD_REQUIRE((int64_t)h * s < 32768 && (int64_t)w * s < 32768, "dense grid too large");
  int H = 0, W = 0;
  DevBuf dsrc, dnodes, ddst;
  OGD_TRY(jpeg_decode_gray_to(dsrc, file, n, &H, &W, nullptr));  // one parse of the file; UNSUPPORTED as ogd_jpeg_info
  OGD_REQUIRE(H < 32768 && W < 32768, "scan %dx%d exceeds the int16 coordinate range of remap", W, H);
  const int Hd = h * s, Wd = w * s;
  OGD_TRY(dnodes.alloc((size_t)h * w * 2 * sizeof(float)));
  OGD_TRY(ddst.alloc((size_t)Hd * Wd));
  OGD_HIP(hipMemcpy(dnodes.p, nodes, (size_t)h * w * 2 * sizeof(float), hipMemcpyHostToDevice));
  OGD_TRY(ogd_dewarp_u8_dev(dsrc.as<uint8_t>(), H, W, 1, W, dnodes.as<float>(), h, w, s, 1.f, 1.f, ddst.as<uint8_t>(),
                            Wd, OGD_INTER_LINEAR, nullptr));
  return jpeg_encode_dev(ddst.as<uint8_t>(), Hd, Wd, Wd, quality, out, cap, out_size, nullptr);
}

}  // extern "C"
