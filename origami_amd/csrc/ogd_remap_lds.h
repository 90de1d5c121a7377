// K2, LDS-tiled remap kernel (ref: origami/core/dewarp.py:612-614, 625-627; arithmetic:
// ogd_remap_math.h) — with the map read from memory (MapGlobal) or recomputed from the node grid
// (MapNodes: the fused K1a + K2 of Dewarper.dewarped / dewarp_image, ref: dewarp.py:608-627).
//
// Every wavefront owns a TW x TH tile of the output (TW = 4 * LX pixels, TH = (64 / LX) * RPT rows; the
// library launches 64 x 8; K2 proper on one channel shears the tile so that each of its rows begins on a 64-byte
// sector of the output, see rl_shear) and a private slice of LDS, so there is no workgroup barrier anywhere:
//   1. every lane loads the map of its 4 * RPT pixels (all loads issued before any use) and quantises
//      it to fixed-point tap coordinates, clamped to the image plus a two-pixel zero border;
//   2. the wavefront reduces the bounding box of those taps (packed int16 min / max over DPP);
//   3. that source window is staged in LDS with coalesced dword loads, zero outside the image
//      (BORDER_CONSTANT 0) — each source byte crosses the vector-memory path once per tile instead of
//      once per tap, and a pixel whose taps all miss the image blends border zeros (no inside test);
//   4. the taps are gathered from LDS as aligned dword pairs + v_alignbyte (an unaligned ds_read_u16
//      is served one lane per cycle), blended with v_dot4_u32_u8 / v_dot2_u32_u16 and stored 4 * C
//      bytes per lane.
// The instruction count per pixel, not HBM traffic, is what holds a naive version of this kernel back
// (measured: DESIGN.md), so everything per pixel is branch-free, 24-bit or packed integer arithmetic.
// A tile whose window does not fit the LDS pitch / row budget (extreme or erratic maps) is redone
// from global memory by remap_tile_global; the result is the same bytes either way.
#pragma once
#include "ogd_remap_math.h"
#include "ogd_resize_math.h"

// Round-3 switches (each measured on its own; tools/build_variant.sh builds the other settings)
#ifndef RL_STAGE_SCALAR
#define RL_STAGE_SCALAR 1  // window staging: wave-uniform row addresses on the scalar unit
#endif
#ifndef RL_PK_QUANT
#define RL_PK_QUANT 1  // x and y of a pixel quantised by one v_pk_fma_f32
#endif
#ifndef RL_STAGE_QW_ALL
#define RL_STAGE_QW_ALL 1  // 8-byte window pieces for every kernel (0: only K2 proper with three / four channels)
#endif
#ifndef RL_WHOLE_TILE
#define RL_WHOLE_TILE 1  // tiles that lie inside the output store without per-row tests
#endif

namespace ogd {

typedef float rl_f4 __attribute__((ext_vector_type(4), aligned(8)));
typedef uint32_t rl_u32u __attribute__((aligned(1)));
typedef uint16_t rl_u16u __attribute__((aligned(1)));
typedef uint64_t rl_u64u __attribute__((aligned(1)));
typedef short rl_s2 __attribute__((ext_vector_type(2)));

// Pointers that arrive inside a by-value struct lose their address space (hipcc then emits flat loads);
// these casts give it back.
#if defined(__HIP_DEVICE_COMPILE__)
#define RL_GLOBAL __attribute__((address_space(1)))
#else
#define RL_GLOBAL  // the host pass only parses the kernels
#endif
typedef const float RL_GLOBAL* rl_gf;
typedef const rl_f4 RL_GLOBAL* rl_gf4;
typedef const float2 RL_GLOBAL* rl_gf2;
typedef const uint8_t RL_GLOBAL* rl_gu8;
typedef const rl_u32u RL_GLOBAL* rl_gu32u;
typedef const rl_u64u RL_GLOBAL* rl_gu64u;
#if defined(__HIP_DEVICE_COMPILE__)
typedef const uint32_t __attribute__((address_space(3)))* rl_lds_u32;
#else
typedef const uint32_t* rl_lds_u32;
#endif

__device__ __forceinline__ rl_s2 rl_as_s2(int v) { return __builtin_bit_cast(rl_s2, v); }
__device__ __forceinline__ int rl_as_int(rl_s2 v) { return __builtin_bit_cast(int, v); }

// Packed (int16 x 2) min of `mn` and max of `mx` over the wavefront (results broadcast through lane 63).
// The two chains are interleaved so that each fills the other's DPP wait states; the first four steps
// move with an undefined `old` (every lane has a source lane), which spares the copy of the tied operand.
__device__ __forceinline__ void rl_wave_reduce_minmax(int& mn, int& mx) {
#define RL_STEP(ctrl)                                                                     \
  {                                                                                       \
    const int a = __builtin_amdgcn_mov_dpp(mn, ctrl, 0xF, 0xF, true);                     \
    const int b = __builtin_amdgcn_mov_dpp(mx, ctrl, 0xF, 0xF, true);                     \
    mn = rl_as_int(__builtin_elementwise_min(rl_as_s2(mn), rl_as_s2(a)));                 \
    mx = rl_as_int(__builtin_elementwise_max(rl_as_s2(mx), rl_as_s2(b)));                 \
  }
#define RL_STEP_B(ctrl)                                                                   \
  {                                                                                       \
    const int a = __builtin_amdgcn_update_dpp(mn, mn, ctrl, 0xF, 0xF, false);             \
    const int b = __builtin_amdgcn_update_dpp(mx, mx, ctrl, 0xF, 0xF, false);             \
    mn = rl_as_int(__builtin_elementwise_min(rl_as_s2(mn), rl_as_s2(a)));                 \
    mx = rl_as_int(__builtin_elementwise_max(rl_as_s2(mx), rl_as_s2(b)));                 \
  }
  RL_STEP(0xB1)     // quad_perm [1,0,3,2]
  RL_STEP(0x4E)     // quad_perm [2,3,0,1]
  RL_STEP(0x141)    // row_half_mirror
  RL_STEP(0x140)    // row_mirror
  RL_STEP_B(0x142)  // row_bcast15
  RL_STEP_B(0x143)  // row_bcast31
#undef RL_STEP
#undef RL_STEP_B
  mn = __builtin_amdgcn_readlane(mn, 63);
  mx = __builtin_amdgcn_readlane(mx, 63);
}

// ---- where the map comes from ------------------------------------------------------------------------
// A map source yields the (x, y) sampling position of output pixels: `load` for the 4 * RPT pixels a
// lane owns in its tile (pixels beyond the right / bottom edge of the output repeat the edge pixel; they
// are never stored), `at` for one pixel (slow path).

// K2 proper: the dense map materialised in memory ([Hd][Wd] float2).
struct MapGlobal {
  static constexpr bool SHEAR = true;  // one-channel tiles are sheared onto the sectors of the output (see the kernel)
  static constexpr bool STAGE_QW = true;  // 64-pixel tiles: multi-channel window rows are staged in 8-byte pieces
  const float* __restrict__ map;
  int Hd, Wd;

  __device__ __forceinline__ float2 at(int X, int Y) const {
    const rl_gf p = (rl_gf)map + ((size_t)Y * Wd + X) * 2;
    return make_float2(p[0], p[1]);
  }

  // Xr[r]: column of the lane's first pixel in row pass r (all equal unless the tile is sheared)
  template <int LX, int RPT>
  __device__ __forceinline__ bool load(int, const int (&Xr)[RPT], int Yb, float (&mv)[RPT][8], uint32_t*, int, int, int,
                                       int, float, float, bool& interior) const {
    constexpr int RW = 64 / LX;
    interior = false;  // nothing is known about a map read from memory: always clamp
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int Y = min(Yb + r * RW, Hd - 1), X = Xr[r];
      if (X >= 0 && X + 4 <= Wd) {
        const rl_gf p = (rl_gf)map + ((size_t)Y * Wd + X) * 2;
        const rl_f4 a = *(rl_gf4)p, b = *(rl_gf4)(p + 4);
        mv[r][0] = a.x; mv[r][1] = a.y; mv[r][2] = a.z; mv[r][3] = a.w;
        mv[r][4] = b.x; mv[r][5] = b.y; mv[r][6] = b.z; mv[r][7] = b.w;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float2 v = at(min(max(X + k, 0), Wd - 1), Y);
          mv[r][2 * k] = v.x;
          mv[r][2 * k + 1] = v.y;
        }
      }
    }
    return true;
  }
};

// Fused K1a + K2: the map value is recomputed from the node grid (ogd_resize_math.h, bit-identical
// to grid_fill_kernel) and scaled like dewarp_image does; the dense map never exists.
// UNIT: the scale is (1, 1) (Dewarper.dewarped; v * 1.0f is v, so the multiplies are left out — a kernel of
// its own, because as a run-time test the compiler turns it into selects over both results).
template <bool UNIT>
struct MapNodesT {
  static constexpr bool SHEAR = false;  // the map of a tile is built per tile column: rectangular tiles only
  static constexpr bool STAGE_QW = false;  // 32-pixel tiles: a window row is <= 32 dwords already (measured: 120 -> 125 us with 8-byte pieces)
  const float2* __restrict__ nodes;
  int h, w;
  double inv_s;
  float scx, scy;
  int Hd, Wd;

  __device__ __forceinline__ float2 at(int X, int Y) const {
    const float2 v = dense_at((rl_gf2)nodes, w, tap_x(X, inv_s, w), tap_y(Y, inv_s, h));
    if (UNIT) return v;
    return make_float2(__fmul_rn(v.x, scx), __fmul_rn(v.y, scy));
  }

  // `lds` is the wavefront's window slice (free until the window is staged).  The map of the tile is built
  // in it the way cv2.resize builds it, separably: (0) the node sub-grid the tile touches, (1) one
  // horizontally lerped row per touched node row (lane = tile column: the x tap, with its fp64
  // arithmetic, is computed once per column and wavefront, not once per pixel), (2) per pixel only the
  // vertical lerp of two such rows.  Same operations in the same order as dense_at, hence the same bits.
  // (X0, Y0) is the tile origin.
  // Returns false (wave-uniform) when the sub-grid does not fit the slice (cells of a pixel or two): the
  // caller then takes the per-pixel path for the tile.
  template <int LX, int RPT>
  __device__ __forceinline__ bool load(int X, const int (&)[RPT], int Yb, float (&mv)[RPT][8], uint32_t* lds,
                                       int lds_dwords, int lane, int X0, int Y0, float xmax, float ymax,
                                       bool& interior) const {
    constexpr int RW = 64 / LX, TW = 4 * LX, TH = RW * RPT;
    static_assert(TW <= 32 && TH <= 32, "one tile column per lane of the lower half-wave, one tile row per lane of the upper");
    constexpr int NP = 8;  // node sub-grid pitch: one node per lane (8 x 8)
    // Lower half-wave: x tap of tile column l5; upper half-wave: y tap of tile row l5 (clamped to the tile; the
    // taps are monotone).  One evaluation of the fp64 chain per lane serves both axes (ogd_resize_math.h:
    // tap_x / tap_y differ only in how the ends are clamped).
    const int half = lane >> 5, l5 = lane & 31;
    const int n = half ? h : w;
    const int pos = half ? min(Y0 + min(l5, TH - 1), Hd - 1) : min(X0 + min(l5, TW - 1), Wd - 1);
    float f = (float)(((double)pos + 0.5) * inv_s - 0.5);
    const float fl = floorf(f);
    const int i = (int)fl;
    f = f - fl;
    const int i0 = min(max(i, 0), n - 1);
    // x: a tap clamped at either end loses its fraction and keeps its right neighbour min(i0 + 1, n - 1);
    // y: both rows are clamped, the fraction stays
    const bool xclamp = !half && (i < 0 || i >= n - 1);
    const int i1 = (!half && i < 0) ? min(1, n - 1) : min(max(i + 1, 0), n - 1);
    if (xclamp) f = 0.f;
    const float w0 = 1.f - f, w1 = f;
    const int jlo = __builtin_amdgcn_readfirstlane(i0), jhi = __builtin_amdgcn_readlane(i1, TW - 1);
    const int ilo = __builtin_amdgcn_readlane(i0, 32), ihi = __builtin_amdgcn_readlane(i1, 32 + TH - 1);
    const int nc = jhi - jlo + 1, nr = ihi - ilo + 1;
    // LDS (dwords): nodes float2[8][8] | rows float2[nr][TW] | y taps float4[TH]
    constexpr int o_rows = 2 * NP * NP;
    const int o_yt = o_rows + 2 * nr * TW;
    if (nc > NP || nr > NP || o_yt + 4 * TH > lds_dwords) return false;  // wave-uniform (cells of a few pixels)
    {
      float2* ln = reinterpret_cast<float2*>(lds);
      float2* rows = reinterpret_cast<float2*>(lds + o_rows);
      float4* ytab = reinterpret_cast<float4*>(lds + o_yt);
      const int r = lane >> 3, c = lane & (NP - 1);
      bool ok = true;
      if (r < nr && c < nc) {
        const float2 v = ((rl_gf2)nodes)[(size_t)(ilo + r) * w + jlo + c];
        ln[lane] = v;
        // Every map value of the tile is a convex combination of these nodes (up to rounding): if all of them
        // lie a pixel inside the clamp range, the clamp of step 1 is the identity for the whole tile.
        const float vx = UNIT ? v.x : __fmul_rn(v.x, scx), vy = UNIT ? v.y : __fmul_rn(v.y, scy);
        ok = vx >= -1.f && vx <= xmax && vy >= -1.f && vy <= ymax;  // false for NaN
      }
      interior = __builtin_amdgcn_ballot_w64(!ok) == 0;
      if (half && l5 < TH)
        ytab[l5] = make_float4(__int_as_float((i0 - ilo) * TW), __int_as_float((i1 - ilo) * TW), w0, w1);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (lane < TW) {
        const int c0 = i0 - jlo, c1 = i1 - jlo;
        for (int rr = 0; rr < nr; ++rr) {  // wave-uniform trip count (2..3 for a 25-px cell)
          const float2 a = ln[rr * NP + c0], b = ln[rr * NP + c1];
          rows[rr * TW + lane] = make_float2(lerp2(a.x, w0, b.x, w1), lerp2(a.y, w0, b.y, w1));
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int lx4 = (lane % LX) * 4, ly = lane / LX;
#pragma unroll
      for (int r2 = 0; r2 < RPT; ++r2) {
        const float4 yt = ytab[ly + r2 * RW];
        const float4* t0 = reinterpret_cast<const float4*>(rows + __float_as_int(yt.x) + lx4);
        const float4* t1 = reinterpret_cast<const float4*>(rows + __float_as_int(yt.y) + lx4);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const float4 p = t0[k2], q = t1[k2];  // two pixels: (x, y, x, y)
          float ax = lerp2(p.x, yt.z, q.x, yt.w), ay = lerp2(p.y, yt.z, q.y, yt.w);
          float bx = lerp2(p.z, yt.z, q.z, yt.w), by = lerp2(p.w, yt.z, q.w, yt.w);
          if (!UNIT) {
            ax = __fmul_rn(ax, scx);
            ay = __fmul_rn(ay, scy);
            bx = __fmul_rn(bx, scx);
            by = __fmul_rn(by, scy);
          }
          mv[r2][4 * k2 + 0] = ax;
          mv[r2][4 * k2 + 1] = ay;
          mv[r2][4 * k2 + 2] = bx;
          mv[r2][4 * k2 + 3] = by;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // the slice is re-used for the source window next
    return true;
  }
};

typedef MapNodesT<false> MapNodes;
typedef MapNodesT<true> MapNodesUnit;

// Sheared tiles (K2 proper, one channel).  A wavefront stores 4 * LX bytes per output row, and where those do not
// begin on a 64-byte sector of the output ADDRESS every such store leaves two partial sectors for the memory
// system to merge (tools/kbench, map-stream skeleton of this kernel on a 6075-byte pitch: 83 us, 79 us with
// aligned segments).  So row rr of tile (bx, by) covers the columns [bx * TW - a, bx * TW - a + TW) with
// a = sh0 + d(rr), a = (address of the tile row's first output row + rr * stride) mod 64 up to a multiple of 64:
// sh0 comes from the tile row, d(rr) = rr * step while that stays within a sector's length (step = stride mod 64,
// taken from [-32, 32)), else its residue in [-32, 32) — the rows of a tile then lie within 64 columns of each other
// and the source window grows by that much at most.  Every row is still cut into consecutive TW-column pieces.
// Three channels: a row piece is 64 pixels = 192 bytes = three sectors once it begins on one, i.e. once
// address + 3 * column = 0 (mod 64); 3 is invertible mod 64 (3 * 43 = 129), so the shift in PIXELS that does it is
// a = 43 * address (mod 64), and it drifts from row to row by 43 * stride (mod 64) — the same walk as for one channel
// with the byte address replaced by 43 times it.  (Map and output then align together: the map address of a pixel is
// 8 * its index, and index = 0 (mod 64) is what both conditions say.)  Four channels would leave a 16-pixel walk
// ((address / 4) mod 16); not done.
// MEASURED AND SWITCHED OFF (profiles/r03/e_rgb_shear.log): same bytes (the stride / alignment tests run with it), but
// K2 RGB 143 -> 169-190 us.  The parallelogram is 35-56 pixels wider than the tile, and where the source window of a
// one-channel tile grows by a few dozen BYTES (the source is a tenth of K2's traffic), a three-channel window grows
// 1.5-1.9x on a source that is a fifth of the traffic and is staged dword by dword: the sectors saved on the
// output (WRITE_SIZE is only 3.8 % above the output's own size for RGB) do not pay for it.
#ifndef RL_SHEAR3
#define RL_SHEAR3 0
#endif
__host__ __device__ constexpr bool rl_shear_channels(int C) { return C == 1 || (RL_SHEAR3 && C == 3); }
#define RL_SHEAR_INV(C) ((C) == 3 ? 43u : 1u)

__device__ __forceinline__ int rl_shear(int sh0, int step, int wrap, int rr) {
  const int d = rr * step;
  return sh0 + (wrap ? ((d + 32) & 63) - 32 : d);
}

// First window row of a staging pass: `row` clamped to [0, last] — on wave-uniform values, i.e. on the scalar unit.
__device__ __forceinline__ int rl_stage_row(int row, int last) {
  return __builtin_amdgcn_readfirstlane(max(min(row, last), 0));
}

// A wave-uniform value pinned to scalar registers and cut off from what it was computed from — otherwise the compiler
// folds a scalar row offset back into 64-bit vector multiply-adds per load.
template <class T>
__device__ __forceinline__ T rl_scalar(T v) {
  asm volatile("" : "+s"(v));
  return v;
}

// Wavefronts per workgroup (neighbouring tiles of one tile row): 4; tools/kbench builds other values to measure
#ifndef RL_WPB
#define RL_WPB 4
#endif

// Slow path of one tile: every pixel straight from global memory (same bytes as the LDS path).  Kept as small as
// it is on purpose: called as the last thing the kernel does, with nothing live across the call, and with few enough
// registers of its own that neither side needs scratch memory.
template <int C, int INTERP, int LX, int RPT, class MapSrc>
__device__ __noinline__ void remap_tile_global(const uint8_t* __restrict__ src, int H, int W,
                                               ptrdiff_t ss, const MapSrc ms, int Hd, int Wd,
                                               uint8_t* __restrict__ dst, ptrdiff_t ds, int X0, int Yb,
                                               int sh0, int sh_step, int sh_wrap) {
  constexpr int RW = 64 / LX;
  // rolled loops: this rare path must not set the register budget (and with it the occupancy) of the kernel
#pragma unroll 1
  for (int r = 0; r < RPT; ++r) {
    const int Y = Yb + r * RW;
    if (Y >= Hd) continue;
    const int X = X0 - rl_shear(sh0, sh_step, sh_wrap, (threadIdx.x & 63) / LX + r * RW);
#pragma unroll 1
    for (int k = 0; k < 4 && X + k < Wd; ++k) {
      if (X + k < 0) continue;
      const float2 m = ms.at(X + k, Y);
      uint8_t o[C];
      sample_q<C, INTERP>(src, H, W, ss, quantize<INTERP>(m.x), quantize<INTERP>(m.y), o);
      uint8_t* d = dst + (ptrdiff_t)Y * ds + (ptrdiff_t)(X + k) * C;
      for (int c = 0; c < C; ++c) d[c] = o[c];
    }
  }
}

// LDSW: bytes of LDS per wavefront (the source window budget).
// Register budget: 8 wavefronts per SIMD (64 VGPRs) for one channel, where the 4 KB LDS slices allow that
// many (RL_W1B wavefronts, 80 VGPRs, for tiles of more than 8 pixels per lane, which would spill at 64);
// 5 (96 VGPRs) for three and four channels, whose 8 KB slices admit no more.
#ifndef RL_W1B
#define RL_W1B 6
#endif
#ifndef RL_WMC
#define RL_WMC 5
#endif
#define RL_WAVES_PER_SIMD(C, RPT) ((C) == 1 ? ((RPT) <= 2 ? 8 : RL_W1B) : RL_WMC)
// LDS geometry of a kernel instance
template <int C, int INTERP, int LX, int RPT, int LDSW, class MapSrc>
struct RlGeom {
  static constexpr bool SHEAR = MapSrc::SHEAR && rl_shear_channels(C);
  static constexpr int TAPS = INTERP == OGD_INTER_LINEAR ? 2 : 1;
  static constexpr int TW = 4 * LX;
  // LDS pitch of the source window (bytes, a multiple of 8: windows are staged in 8-byte pieces): tile width x
  // stretch + taps and alignment.  Stretch 5/4 for K2 proper; 3/2 for the 32-pixel tiles of the fused kernel with
  // one and three channels, whose slices have rows to spare — the pitch itself costs nothing (48, 52, 60 bytes:
  // 64.2, 63.7, 64.3 us, profiles/r03/h_pitch.log: LDS bank conflicts are not what holds this kernel), and a page
  // whose map stretches by more than the pitch allows sends the tile down the slow path (18 % of the tiles of
  // configs[4]'s heavily warped page at 5/4, tools/heavy_tiles.py).
#ifndef RL_PF_NUM
#define RL_PF_NUM ((!MapSrc::SHEAR && C != 4) ? 6 : 5)
#endif
  static constexpr int PF = (((TW + (SHEAR ? 64 : 0)) * C * RL_PF_NUM) / 4 + (TAPS + 6) * C + 7) & ~7;
  static constexpr int MAXROWS = LDSW / PF;
  // + 3 dwords of slack (tap gathers read up to 2 dwords past a row), rounded to 16 bytes (b128 accesses in MapNodes)
  static constexpr int WIN_DW = ((PF / 4) * MAXROWS + 3 + 3) & ~3;
};

// Page-size bounds as floats.
struct RlBounds {
  float xmax, ymax;  // W - 1, H - 1
  float qxhi, qyhi;  // upper clamp bounds of the quantised coordinates: MAGIC + QS * W, MAGIC + QS * H
};
constexpr float RL_MAGIC = 12582912.f;  // 1.5 * 2^23; bit pattern 0x4B400000, a multiple of 32

// One tile (t) by one wavefront; `win` is the wavefront's LDS slice.  A tile that does not fit the LDS path goes to
// remap_tile_global before anything is written.
template <int C, int INTERP, int LX, int RPT, int LDSW, class MapSrc>
__device__ __forceinline__ void rl_tile(const uint8_t* __restrict__ src, int H, int W, ptrdiff_t ss, const MapSrc& ms,
                                        int Hd, int Wd, uint8_t* __restrict__ dst, ptrdiff_t ds, int nbx, int sh_step,
                                        int sh_wrap, int sh_bias, uint32_t* win, int t, int lane, const RlBounds bd) {
  typedef RlGeom<C, INTERP, LX, RPT, LDSW, MapSrc> G;
  constexpr bool LIN = INTERP == OGD_INTER_LINEAR;
  constexpr bool SHEAR = G::SHEAR;
  constexpr int TAPS = G::TAPS;
  constexpr int SHIFT = LIN ? 5 : 0;
  constexpr int RW = 64 / LX;   // output rows per wave pass
  constexpr int TH = RW * RPT;  // tile height
  constexpr int TW = 4 * LX;    // tile width
  constexpr int PF = G::PF, MAXROWS = G::MAXROWS, WIN_DW = G::WIN_DW;
  const int lx4 = (lane % LX) * 4, ly = lane / LX;
  const int by = t / nbx, bx = t - by * nbx;
  const int X = bx * TW + lx4;
  const int Yb = by * TH + ly;  // row of pass r: Yb + r * RW
  int Xr[RPT];  // column of the lane's first pixel in row pass r
  int sh0 = 0;
  if (SHEAR) {
    sh0 = (int)(((reinterpret_cast<uintptr_t>(dst) + (ptrdiff_t)(by * TH) * ds) * RL_SHEAR_INV(C)) & 63) + sh_bias;
    bool any = false;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      Xr[r] = X - rl_shear(sh0, sh_step, sh_wrap, ly + r * RW);
      any |= Yb + r * RW < Hd && Xr[r] + 4 > 0 && Xr[r] < Wd;
    }
    if (__builtin_amdgcn_ballot_w64(any) == 0) return;  // a tile past the ends of all its rows
  } else {
#pragma unroll
    for (int r = 0; r < RPT; ++r) Xr[r] = X;
  }
  float mv[RPT][8];
  bool interior;  // wave-uniform: the clamp below cannot change any map value of this tile
  if (!ms.template load<LX, RPT>(X, Xr, Yb, mv, win, WIN_DW, lane, bx * TW, by * TH, bd.xmax, bd.ymax,
                                 interior)) {  // wave-uniform
    remap_tile_global<C, INTERP, LX, RPT>(src, H, W, ss, ms, Hd, Wd, dst, ds, X, Yb, sh0, sh_step, sh_wrap);
    return;
  }

  // 1. + 2. (see the file comment for the clamp)
  // Quantisation without a float->int conversion: for |v| < 2^22, v + 1.5 * 2^23 rounds v to the nearest
  // integer, ties to even, in the adder (that is cvRound), and leaves it in the low mantissa bits.  The
  // product m * 32 is exact, so one fma gives cvRound(m * 32); the clamp (applied to m: the bounds are
  // multiples of 32) keeps the sum inside [2^23, 2^24) and sends NaN / huge values to the border.
  const float xlo = LIN ? -2.f : -1.f, ylo = xlo;  // the upper bounds are W and H (bd)
  constexpr float MAGIC = RL_MAGIC;
  int lp[RPT][4];
  uint32_t wa[RPT][4], wb[RPT][4];
  int mn = 0x7fff7fff, mx = (int)0x80008000;
  // x and y of a pixel are quantised by one packed instruction (v_pk_fma_f32 / v_pk_add_f32: each half rounded as the
  // scalar form).  The clamp comes after it, on the sums: rounding is monotone and the bounds are integers, so
  // med3(fma(m, 32, MAGIC), MAGIC + 32 lo, MAGIC + 32 hi) == fma(med3(m, lo, hi), 32, MAGIC) for every m, huge
  // values and NaN (which med3 sends to its lower bound either way) included.
  constexpr float QS = LIN ? 32.f : 1.f;
  const float qxlo = MAGIC + QS * xlo, qxhi = bd.qxhi, qylo = MAGIC + QS * ylo, qyhi = bd.qyhi;
#pragma unroll
  for (int r = 0; r < RPT; ++r)
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      const f32x2 cxy = {mv[r][k], mv[r][k + 1]}, magic2 = {MAGIC, MAGIC}, k32 = {32.f, 32.f};
      f32x2 qf = LIN ? __builtin_elementwise_fma(cxy, k32, magic2) : cxy + magic2;
      if (!RL_PK_QUANT) {
        qf.x = LIN ? __builtin_fmaf(cxy.x, 32.f, MAGIC) : __fadd_rn(cxy.x, MAGIC);
        qf.y = LIN ? __builtin_fmaf(cxy.y, 32.f, MAGIC) : __fadd_rn(cxy.y, MAGIC);
        asm volatile("" : "+v"(qf.x), "+v"(qf.y));
      }
      mv[r][k] = qf.x;
      mv[r][k + 1] = qf.y;
    }
  if (__builtin_expect(!interior, 0)) {
#pragma unroll
    for (int r = 0; r < RPT; ++r)
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        mv[r][k] = __builtin_amdgcn_fmed3f(mv[r][k], qxlo, qxhi);
        mv[r][k + 1] = __builtin_amdgcn_fmed3f(mv[r][k + 1], qylo, qyhi);
      }
  }
#pragma unroll
  for (int r = 0; r < RPT; ++r)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t qbx = __float_as_uint(mv[r][2 * k]), qby = __float_as_uint(mv[r][2 * k + 1]);
      const int cx = __builtin_amdgcn_sbfe((int)qbx, SHIFT, 16), cy = __builtin_amdgcn_sbfe((int)qby, SHIFT, 16);
      lp[r][k] = __mul24(cy, PF) + (C == 1 ? cx : __mul24(cx, C));
      if (LIN) {
        wa[r][k] = __umul24(qbx & 31, 255) + 32;              // (32 - fx) | fx << 8
        wb[r][k] = __umul24(qby & 31, 64 * 65535) + 64 * 32;  // 64 * ((32 - fy) | fy << 16)
      }
      asm volatile("" : "+v"(lp[r][k]));
      if (LIN) asm volatile("" : "+v"(wa[r][k]), "+v"(wb[r][k]));
      const int pc = (int)__builtin_amdgcn_perm((uint32_t)cy, (uint32_t)cx, 0x05040100u);  // cx | cy << 16
      mn = rl_as_int(__builtin_elementwise_min(rl_as_s2(mn), rl_as_s2(pc)));
      mx = rl_as_int(__builtin_elementwise_max(rl_as_s2(mx), rl_as_s2(pc)));
    }
  rl_wave_reduce_minmax(mn, mx);
  const int mnx = (short)(mn & 0xffff), mny = mn >> 16, mxx = (short)(mx & 0xffff), mxy = mx >> 16;
  const int x0b = (mnx * C) & ~3;
  const int wbytes = (mxx + TAPS) * C - x0b;
  const int nrows = mxy + TAPS - mny;
  if (wbytes > PF || nrows > MAXROWS) {  // wave-uniform
    remap_tile_global<C, INTERP, LX, RPT>(src, H, W, ss, ms, Hd, Wd, dst, ds, X, Yb, sh0, sh_step, sh_wrap);
    return;
  }

  // 3. stage: lanes cover (row, dword) pairs, 32 or 64 dwords per window row
  const int P4 = (wbytes + 3) >> 2, WB = W * C;
  if (x0b >= 0 && x0b + 4 * P4 <= WB && mny >= 0 && mny + nrows <= H && ss < (1 << 24)) {  // wave-uniform
    // window inside the image (all but the tiles on the page border): no per-dword tests, and the
    // loads of up to 8 window rows per lane are all in flight before the first one is written to LDS
    // (one memory round trip per tile instead of one per row pair)
#ifndef RL_STAGE_U
#define RL_STAGE_U 8
#endif
#ifndef RL_STAGE_QW
#define RL_STAGE_QW 1
#endif
#ifndef RL_STAGE_UQ1_K2
#define RL_STAGE_UQ1_K2 6
#endif
#ifndef RL_STAGE_UQ1_FUSED
#define RL_STAGE_UQ1_FUSED 5
#endif
#ifndef RL_STAGE_UQ
#define RL_STAGE_UQ 8  // 8-byte pieces in flight per lane
#endif
    // three / four channels: a window row is 50-70 dwords, i.e. one row per pass of the wavefront, and a 64 x 8 tile's
    // window has 11-13 rows: 12 loads in flight per lane stage it in ONE round trip (8: two)
#ifndef RL_STAGE_U_MC
#define RL_STAGE_U_MC 12
#endif
    constexpr int U = C == 1 ? RL_STAGE_U : RL_STAGE_U_MC;
    const uint8_t* pw = src + ((ptrdiff_t)mny * ss + x0b);  // wave-uniform base + 32-bit lane offsets
    const uint32_t ss32 = (uint32_t)ss;                      // the window lies inside the image: ss > 0
    const rl_gu8 pwg = (rl_gu8)pw;
    if (RL_STAGE_QW && (RL_STAGE_QW_ALL || (MapSrc::STAGE_QW && C > 1)) && x0b + 8 * ((wbytes + 7) >> 3) <= WB) {  // wave-uniform
      // three / four channels: window rows of 50-70 dwords are staged as 8-byte pieces — half the loads and LDS
      // writes, and 26-35 pieces per row let the wavefront take two rows per pass (PF is a multiple of 8 and the
      // slice 16-byte aligned; global loads need no more than their dword alignment)
      static_assert(PF % 8 == 0, "8-byte window pieces need an 8-byte pitch");
      // pieces in flight per lane: what the window of a tile needs in one trip (one channel: 8 columns x 8 rows per
      // pass and ~35 rows for the fused kernel's 32 x 32 tiles, 32 x 2 and ~12 rows for K2's 64 x 8) and no more — the
      // tap addresses and weights of all the lane's pixels are live here, and 8 pieces spilled
      constexpr int UQ = C == 1 ? (MapSrc::SHEAR ? RL_STAGE_UQ1_K2 : RL_STAGE_UQ1_FUSED) : RL_STAGE_UQ;
      const int P8 = (wbytes + 7) >> 3;
      const int sh = P8 <= 8 ? 3 : P8 <= 16 ? 4 : P8 <= 32 ? 5 : 6, cw = 1 << sh, rp = 64 >> sh;
      const int r0 = min(lane >> sh, nrows - 1);
      uint64_t* win8 = reinterpret_cast<uint64_t*>(win) + __mul24(r0, PF / 8);
      const uint32_t goff = __umul24((uint32_t)r0, ss32);
      for (int c = lane & (cw - 1); c < P8; c += cw) {
        for (int rb = 0; rb < nrows; rb += UQ * rp) {  // wave-uniform trip count
          uint64_t v[UQ];
          uint32_t lo = goff + 8u * (uint32_t)c;
#pragma unroll
          for (int u = 0; u < UQ; ++u) {
            if (u && rb + u * rp >= nrows) continue;  // wave-uniform: a pass that begins past the window is skipped
            const int row = rl_stage_row(rb + u * rp, nrows - rp);  // wave-uniform: the row's address is scalar arithmetic
            asm volatile("" : "+v"(lo));  // (in this block, so that the load is scalar base + zero-extended lane offset)
            v[u] = *(rl_gu64u)(rl_scalar(pwg + (uint32_t)row * ss32) + lo);  // row * ss < 2^31
          }
#pragma unroll
          for (int u = 0; u < UQ; ++u) {
            if (u && rb + u * rp >= nrows) continue;
            const int row = rl_stage_row(rb + u * rp, nrows - rp);
            win8[rl_scalar(row * (PF / 8)) + c] = v[u];
          }
        }
      }
    } else {
    const int sh = P4 <= 32 ? 5 : 6, cw = 1 << sh, rp = 64 >> sh;
    // Row u of a pass is rows (row_u, row_u + 1) of the window for the two half-waves (or one row for all lanes):
    // row_u is wave-uniform, so the source address of a load is a scalar base + one per-lane offset computed once,
    // and the LDS address one vector addition (it was add + min + two multiply-adds per load: a sixth of the fused
    // kernel's vector instructions).  Passes that reach past the window re-read and re-write its last rows.
#if RL_STAGE_SCALAR
    const int r0 = min(lane >> sh, nrows - 1);
    uint32_t* const wlane = win + __mul24(r0, PF / 4);
    const uint32_t goff = __umul24((uint32_t)r0, ss32);  // ss < 2^24
    for (int c = lane & (cw - 1); c < P4; c += cw) {
      for (int rb = 0; rb < nrows; rb += U * rp) {  // wave-uniform trip count
        uint32_t v[U];
        uint32_t lo = goff + 4u * (uint32_t)c;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (u && rb + u * rp >= nrows) continue;  // wave-uniform: a pass that begins past the window is skipped
          const int row = rl_stage_row(rb + u * rp, nrows - rp);
          asm volatile("" : "+v"(lo));
          v[u] = *(rl_gu32u)(rl_scalar(pwg + (uint32_t)row * ss32) + lo);  // row * ss < 2^31
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (u && rb + u * rp >= nrows) continue;
          const int row = rl_stage_row(rb + u * rp, nrows - rp);
          wlane[rl_scalar(row * (PF / 4)) + c] = v[u];
        }
      }
    }
#else
    const int r0 = lane >> sh;
    for (int c = lane & (cw - 1); c < P4; c += cw) {
      for (int rb = 0; rb < nrows; rb += U * rp) {
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t row = (uint32_t)min(rb + r0 + u * rp, nrows - 1);
          v[u] = *reinterpret_cast<const rl_u32u*>(pw + (__umul24(row, ss32) + 4u * (uint32_t)c));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int row = min(rb + r0 + u * rp, nrows - 1);
          win[__mul24(row, PF / 4) + c] = v[u];
        }
      }
    }
#endif
    }
  } else {
    const int sh = P4 <= 32 ? 5 : 6, cw = 1 << sh, rp = 64 >> sh;
    for (int c = lane & (cw - 1); c < P4; c += cw) {
      const int xb = x0b + 4 * c;
      const bool full = xb >= 0 && xb + 4 <= WB;
      for (int row = lane >> sh; row < nrows; row += rp) {
        const int y = mny + row;
        const bool yin = (unsigned)y < (unsigned)H;
        const uint8_t* prow = src + (ptrdiff_t)y * ss;
        uint32_t v = 0;
        if (yin && full) {
          v = *reinterpret_cast<const rl_u32u*>(prow + xb);
        } else if (yin) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if ((unsigned)(xb + j) < (unsigned)WB) v |= (uint32_t)prow[xb + j] << (8 * j);
        }
        win[row * (PF / 4) + c] = v;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const uint8_t* wbytes_p = reinterpret_cast<const uint8_t*>(win);
  const int base = __mul24(mny, PF) + x0b;
  // LDS byte address of tap (cy, cx) = lp + lds_rel; slice and window origin are dword aligned, so its low two
  // bits are the byte phase of the tap inside its dword
  const uint32_t lds_rel = (uint32_t)(uintptr_t)(rl_lds_u32)win - (uint32_t)base;
  // output rows: wave-uniform tile origin + 32-bit lane offsets (row strides are < 2^24: check_image)
  uint8_t* const dtile = dst + ((ptrdiff_t)(by * TH) * ds + (ptrdiff_t)(bx * TW) * C);
  const uint32_t ds32 = (uint32_t)ds;
  const uint32_t dlane = __umul24((uint32_t)ly, ds32) + (uint32_t)(lx4 * C);
  const bool whole_tile = by * TH + TH <= Hd && bx * TW + TW <= Wd && ds > 0;  // wave-uniform
  // 4. gather + blend + store
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const int Y = Yb + r * RW;
    uint32_t o[4 * C];  // result bytes in bits 16..23
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int off = lp[r][k] - base;
      const uint8_t* p = wbytes_p + off;
      if (LIN) {
        uint32_t top[C], bot[C];
        if (C == 1) {
          const uint32_t a = (uint32_t)lp[r][k] + lds_rel;
          const rl_lds_u32 q = (rl_lds_u32)(uintptr_t)(a & ~3u);
          const uint32_t t2 = __builtin_amdgcn_alignbyte(q[1], q[0], a);
          const uint32_t b2 = __builtin_amdgcn_alignbyte(q[PF / 4 + 1], q[PF / 4], a);
          top[0] = __builtin_amdgcn_udot4(t2, wa[r][k], 0u, false);
          bot[0] = __builtin_amdgcn_udot4(b2, wa[r][k], 0u, false);
        } else if (C == 3) {
          // the six bytes of the two taps of a row as aligned dwords + v_alignbyte: lo = (r0 g0 b0 r1),
          // hi = (g1 b1 . .); v_perm pairs channel c of both taps for the dot product
          const int i = off >> 2;
          const uint32_t tl = __builtin_amdgcn_alignbyte(win[i + 1], win[i], (uint32_t)off);
          const uint32_t th = __builtin_amdgcn_alignbyte(win[i + 2], win[i + 1], (uint32_t)off);
          const uint32_t bl = __builtin_amdgcn_alignbyte(win[i + PF / 4 + 1], win[i + PF / 4], (uint32_t)off);
          const uint32_t bh = __builtin_amdgcn_alignbyte(win[i + PF / 4 + 2], win[i + PF / 4 + 1], (uint32_t)off);
          top[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(th, tl, 0x0c0c0300u), wa[r][k], 0u, false);
          top[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(th, tl, 0x0c0c0401u), wa[r][k], 0u, false);
          top[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(th, tl, 0x0c0c0502u), wa[r][k], 0u, false);
          bot[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(bh, bl, 0x0c0c0300u), wa[r][k], 0u, false);
          bot[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(bh, bl, 0x0c0c0401u), wa[r][k], 0u, false);
          bot[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(bh, bl, 0x0c0c0502u), wa[r][k], 0u, false);
        } else {
          // four channels: a pixel is one aligned dword, the two taps of a row are neighbours
          const int i = off >> 2;
          const uint32_t t0 = win[i], t1 = win[i + 1], b0 = win[i + PF / 4], b1 = win[i + PF / 4 + 1];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const uint32_t sel = 0x0c0c0400u + 0x0101u * c;  // (channel c of tap 0, channel c of tap 1)
            top[c] = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(t1, t0, sel), wa[r][k], 0u, false);
            bot[c] = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(b1, b0, sel), wa[r][k], 0u, false);
          }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const uint32_t tb = top[c] | (bot[c] << 16);
          typedef unsigned short us2 __attribute__((ext_vector_type(2)));
          o[k * C + c] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, tb), __builtin_bit_cast(us2, wb[r][k]), 64u * 512u, false);
        }
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) o[k * C + c] = (uint32_t)p[c] << 16;
      }
    }
    const int Xs = Xr[r];
    if (!SHEAR && RL_WHOLE_TILE && whole_tile) {  // wave-uniform: no per-row tests, the row's address is a scalar base + the lane's offset
      uint8_t* d = dtile + (ptrdiff_t)(r * RW) * ds + dlane;
#pragma unroll
      for (int j = 0; j < 4 * C; j += 4) {
        const uint32_t lo = __builtin_amdgcn_perm(o[j + 1], o[j], 0x0c0c0602u);
        const uint32_t hi = __builtin_amdgcn_perm(o[j + 3], o[j + 2], 0x06020c0cu);
        *reinterpret_cast<rl_u32u*>(d + j) = lo | hi;
      }
    } else if (Y < Hd && Xs < Wd && Xs + 4 > 0) {
      uint8_t* d = dtile + (int)(dlane + (uint32_t)(r * RW) * ds32 + (uint32_t)((Xs - X) * C));
      if (Xs >= 0 && Xs + 4 <= Wd) {
#pragma unroll
        for (int j = 0; j < 4 * C; j += 4) {
          const uint32_t lo = __builtin_amdgcn_perm(o[j + 1], o[j], 0x0c0c0602u);
          const uint32_t hi = __builtin_amdgcn_perm(o[j + 3], o[j + 2], 0x06020c0cu);
          *reinterpret_cast<rl_u32u*>(d + j) = lo | hi;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4 * C; ++j)
          if ((unsigned)(Xs + j / C) < (unsigned)Wd) d[j] = (uint8_t)(o[j] >> 16);
      }
    }
  }
}

// One tile per wavefront.  Several tiles per wavefront (a loop here, workgroups of 4 x tpw consecutive tiles) were
// measured in round 3 because the counters show the fused kernel's wave slots a third empty: 2 / 4 / 8 tiles per
// wavefront ran the fused kernel in 70.7 / 76.5 / 87 us against 68.6, and the loop alone (its loop-carried scalars
// spill to VGPR lanes) cost K2 proper 10 % (profiles/r03/p_switches3.log).
template <int C, int INTERP, int LX, int RPT, int LDSW, class MapSrc>
__global__ __launch_bounds__(64 * RL_WPB, RL_WAVES_PER_SIMD(C, RPT)) void remap_u8_wave_kernel(
    const uint8_t* __restrict__ src, int H, int W, ptrdiff_t ss, const MapSrc ms, int Hd, int Wd,
    uint8_t* __restrict__ dst, ptrdiff_t ds, int nbx, int nb, int sh_step = 0, int sh_wrap = 0, int sh_bias = 0) {
  constexpr int WIN_DW = RlGeom<C, INTERP, LX, RPT, LDSW, MapSrc>::WIN_DW;
  __shared__ __attribute__((aligned(16))) uint32_t win_all[RL_WPB * WIN_DW];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int t = blockIdx.x * RL_WPB + wave;
  if (t >= nb) return;  // wave-uniform
  constexpr float QS = INTERP == OGD_INTER_LINEAR ? 32.f : 1.f;
  const RlBounds bd = {(float)(W - 1), (float)(H - 1), RL_MAGIC + QS * (float)W, RL_MAGIC + QS * (float)H};
  rl_tile<C, INTERP, LX, RPT, LDSW, MapSrc>(src, H, W, ss, ms, Hd, Wd, dst, ds, nbx, sh_step, sh_wrap, sh_bias,
                                            win_all + wave * WIN_DW, t, lane, bd);
}

}  // namespace ogd
