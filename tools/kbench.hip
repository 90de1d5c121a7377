// kbench — kernel laboratory for K2 (remap): times candidate kernels against the library's
// ogd_remap_u8_dev on one broadsheet page and checks that every candidate is bit-identical to it.
// Development tool, not part of the product; winners are ported into origami_amd/csrc/ (the tile
// shapes the library launches are TileShape<> in ogd_remap.hip).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/kbench.hip \
//         -Lorigami_amd/lib -logdewarp -Wl,-rpath,'$ORIGIN/../origami_amd/lib' -o tools/kbench
//   tools/kbench [W H cell iters]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../include/ogdewarp.h"
#include "../origami_amd/csrc/ogd_remap_math.h"
#include "../origami_amd/csrc/ogd_remap_lds.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)
#define OK(x)                                                       \
  do {                                                              \
    int s_ = (x);                                                   \
    if (s_ != 0) {                                                  \
      fprintf(stderr, "%s: %s\n", #x, ogd_last_error());            \
      exit(2);                                                      \
    }                                                               \
  } while (0)

using namespace ogd;

// the map source of K2 without the sheared tiles of the library's launch: the "wave ..." variants below
struct MapGlobalPlain : MapGlobal {
  static constexpr bool SHEAR = false;
};

typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));

template <bool NT>
__device__ __forceinline__ f4a8 ldmap(const float* p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f4a8*>(p));
  return *reinterpret_cast<const f4a8*>(p);
}

template <bool NT>
__device__ __forceinline__ void st4(uint8_t* p, uint32_t v) {
  typedef uint32_t u32a1 __attribute__((aligned(1)));
  if (NT)
    __builtin_nontemporal_store(v, reinterpret_cast<u32a1*>(p));
  else
    *reinterpret_cast<u32a1*>(p) = v;
}

// logical tile id from the hardware block id: XCD k (= blockIdx % 8 under round-robin dispatch) walks
// a contiguous band of tiles, so vertically adjacent tiles share that XCD's L2.
template <bool XCD>
__device__ __forceinline__ int tile_id(int nb) {
  if (!XCD) return blockIdx.x;
  const int per = (nb + 7) >> 3;
  return (blockIdx.x & 7) * per + (blockIdx.x >> 3);
}

// rows variant: a wavefront covers 256 px of ROWS consecutive rows; all map loads are issued first.
template <int C, int INTERP, int ROWS, bool NT, bool XCD>
__global__ __launch_bounds__(256) void remap_rows(const uint8_t* __restrict__ src, int H, int W,
                                                  ptrdiff_t ss, const float* __restrict__ map, int Hd,
                                                  int Wd, uint8_t* __restrict__ dst, ptrdiff_t ds,
                                                  int nbx, int nb) {
  const int t = tile_id<XCD>(nb);
  if (t >= nb) return;
  const int by = t / nbx, bx = t - by * nbx;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int X = (bx * 64 + lane) * 4;
  const int Y0 = (by * 4 + wave) * ROWS;
  if (X >= Wd || Y0 >= Hd) return;
  if (X + 4 <= Wd) {
    f4a8 m[ROWS][2];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int Y = min(Y0 + r, Hd - 1);
      const float* p = map + ((size_t)Y * Wd + X) * 2;
      m[r][0] = ldmap<NT>(p);
      m[r][1] = ldmap<NT>(p + 4);
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if (Y0 + r < Hd) {
        uint8_t o[4 * C];
        sample<C, INTERP>(src, H, W, ss, m[r][0].x, m[r][0].y, o);
        sample<C, INTERP>(src, H, W, ss, m[r][0].z, m[r][0].w, o + C);
        sample<C, INTERP>(src, H, W, ss, m[r][1].x, m[r][1].y, o + 2 * C);
        sample<C, INTERP>(src, H, W, ss, m[r][1].z, m[r][1].w, o + 3 * C);
        uint8_t* d = dst + (ptrdiff_t)(Y0 + r) * ds + (ptrdiff_t)X * C;
#pragma unroll
        for (int k = 0; k < 4 * C; k += 4)
          st4<NT>(d + k, (uint32_t)o[k] | ((uint32_t)o[k + 1] << 8) | ((uint32_t)o[k + 2] << 16) |
                             ((uint32_t)o[k + 3] << 24));
      }
    }
  } else {
    for (int r = 0; r < ROWS && Y0 + r < Hd; ++r) {
      const float* p = map + ((size_t)(Y0 + r) * Wd + X) * 2;
      uint8_t* d = dst + (ptrdiff_t)(Y0 + r) * ds + (ptrdiff_t)X * C;
      for (int k = 0; X + k < Wd; ++k) {
        uint8_t o[C];
        sample<C, INTERP>(src, H, W, ss, p[2 * k], p[2 * k + 1], o);
        for (int c = 0; c < C; ++c) d[k * C + c] = o[c];
      }
    }
  }
}

// streaming ceiling of this access pattern: read the whole map, write one byte per pixel, no gather
template <int ROWS, bool NT>
__global__ __launch_bounds__(256) void map_stream(const float* __restrict__ map, int Hd, int Wd,
                                                  uint8_t* __restrict__ dst, ptrdiff_t ds, int nbx, int nb) {
  const int t = blockIdx.x;
  const int by = t / nbx, bx = t - by * nbx;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int X = (bx * 64 + lane) * 4;
  const int Y0 = (by * 4 + wave) * ROWS;
  if (X + 4 > Wd || Y0 >= Hd) return;
  f4a8 m[ROWS][2];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int Y = min(Y0 + r, Hd - 1);
    const float* p = map + ((size_t)Y * Wd + X) * 2;
    m[r][0] = ldmap<NT>(p);
    m[r][1] = ldmap<NT>(p + 4);
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
    if (Y0 + r < Hd) {
      const uint32_t v = (uint32_t)(int)m[r][0].x | ((uint32_t)(int)m[r][0].z << 8) |
                         ((uint32_t)(int)m[r][1].x << 16) | ((uint32_t)(int)m[r][1].z << 24);
      st4<NT>(dst + (ptrdiff_t)(Y0 + r) * ds + X, v);
    }
}

// streaming ceiling of K2's own tile shape: the map loads of remap_u8_wave_kernel, one dword out per lane-row
template <int LX, int RPT>
__global__ __launch_bounds__(256) void map_stream_tile(const float* __restrict__ map, int Hd, int Wd,
                                                       uint8_t* __restrict__ dst, ptrdiff_t ds, int nbx, int nb) {
  constexpr int RW = 64 / LX, TH = RW * RPT, TW = 4 * LX;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + wave;
  if (t >= nb) return;
  const int by = t / nbx, bx = t - by * nbx;
  const int X = bx * TW + (lane % LX) * 4, Yb = by * TH + lane / LX;
  if (X + 4 > Wd) return;
  f4a8 m[RPT][2];
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const float* p = map + ((size_t)min(Yb + r * RW, Hd - 1) * Wd + X) * 2;
    m[r][0] = ldmap<false>(p);
    m[r][1] = ldmap<false>(p + 4);
  }
#pragma unroll
  for (int r = 0; r < RPT; ++r)
    if (Yb + r * RW < Hd) {
      const uint32_t v = (uint32_t)(int)m[r][0].x | ((uint32_t)(int)m[r][0].z << 8) |
                         ((uint32_t)(int)m[r][1].x << 16) | ((uint32_t)(int)m[r][1].z << 24);
      st4<false>(dst + (ptrdiff_t)(Yb + r * RW) * ds + X, v);
    }
}

// the same with the workgroup's output (4 neighbouring tiles of one tile row: 4 * TW bytes per row) collected in LDS
// and written as 8-byte chunks aligned to the output ADDRESS: two partial 64-byte sectors per row and workgroup
// instead of two per row and wavefront
template <int LX, int RPT>
__global__ __launch_bounds__(256) void map_stream_tile_wg(const float* __restrict__ map, int Hd, int Wd,
                                                          uint8_t* __restrict__ dst, ptrdiff_t ds, int nbx4) {
  constexpr int RW = 64 / LX, TH = RW * RPT, TW = 4 * LX, RB = 4 * TW;
  __shared__ __attribute__((aligned(16))) uint32_t tile[TH * RB / 4 + 4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int by = blockIdx.x / nbx4, bx4 = blockIdx.x - by * nbx4;
  const int X0 = bx4 * RB;
  const int X = X0 + wave * TW + (lane % LX) * 4, Yb = by * TH + lane / LX;
  f4a8 m[RPT][2];
  const int Xc = min(X, Wd - 4);
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const float* p = map + ((size_t)min(Yb + r * RW, Hd - 1) * Wd + Xc) * 2;
    m[r][0] = ldmap<false>(p);
    m[r][1] = ldmap<false>(p + 4);
  }
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const uint32_t v = (uint32_t)(int)m[r][0].x | ((uint32_t)(int)m[r][0].z << 8) |
                       ((uint32_t)(int)m[r][1].x << 16) | ((uint32_t)(int)m[r][1].z << 24);
    tile[((lane / LX + r * RW) * RB + wave * TW + (lane % LX) * 4) >> 2] = v;
  }
  __syncthreads();
  constexpr int CPR = RB / 8;          // full chunks per row
  constexpr int TPR = 256 / TH;        // threads per row
  const int row = threadIdx.x / TPR, j = threadIdx.x % TPR;
  const int Y = by * TH + row;
  if (Y >= Hd) return;
  const int n = min(RB, Wd - X0);      // valid bytes of this row
  uint8_t* a = dst + (ptrdiff_t)Y * ds + X0;
  const int sg = (int)(reinterpret_cast<uintptr_t>(a) & 7);
  const uint8_t* lrow = reinterpret_cast<const uint8_t*>(tile) + row * RB;
  for (int c = j; c <= CPR; c += TPR) {
    const int o = 8 * c - sg;          // row byte offset of the chunk
    if (o >= 0 && o + 8 <= n) {
      const uint32_t* q = reinterpret_cast<const uint32_t*>(lrow + (o & ~3));
      const uint32_t lo = __builtin_amdgcn_alignbyte(q[1], q[0], (uint32_t)o);
      const uint32_t hi = __builtin_amdgcn_alignbyte(q[2], q[1], (uint32_t)o);
      *reinterpret_cast<uint2*>(a + o) = make_uint2(lo, hi);
    } else {
      for (int b = max(o, 0); b < min(o + 8, n); ++b) a[b] = lrow[b];
    }
  }
}

// the same with every row of a tile moved left until its first output byte stands on a 64-byte sector of the
// output ADDRESS (a sheared tile; SHEAR_MAP: until its first map value stands on a 128-byte line instead)
template <int LX, int RPT, bool SHEAR_MAP>
__global__ __launch_bounds__(256) void map_stream_tile_shear(const float* __restrict__ map, int Hd, int Wd,
                                                             uint8_t* __restrict__ dst, ptrdiff_t ds, int nbx, int nb) {
  constexpr int RW = 64 / LX, TH = RW * RPT, TW = 4 * LX;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + wave;
  if (t >= nb) return;
  const int by = t / nbx, bx = t - by * nbx;
  const int Yb = by * TH + lane / LX;
  f4a8 m[RPT][2];
  int Xr[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const int Y = min(Yb + r * RW, Hd - 1);
    const int a = SHEAR_MAP ? (int)(((reinterpret_cast<uintptr_t>(map) >> 3) + (size_t)Y * Wd + bx * TW) & 15)
                            : (int)((reinterpret_cast<uintptr_t>(dst) + (size_t)Y * ds + bx * TW) & 63);
    Xr[r] = bx * TW + (lane % LX) * 4 - a;
    const float* p = map + ((size_t)Y * Wd + min(max(Xr[r], 0), Wd - 4)) * 2;
    m[r][0] = ldmap<false>(p);
    m[r][1] = ldmap<false>(p + 4);
  }
#pragma unroll
  for (int r = 0; r < RPT; ++r)
    if (Yb + r * RW < Hd && Xr[r] >= 0 && Xr[r] + 4 <= Wd) {
      const uint32_t v = (uint32_t)(int)m[r][0].x | ((uint32_t)(int)m[r][0].z << 8) |
                         ((uint32_t)(int)m[r][1].x << 16) | ((uint32_t)(int)m[r][1].z << 24);
      st4<false>(dst + (ptrdiff_t)(Yb + r * RW) * ds + Xr[r], v);
    }
}

// K1a lab: store policy, rows per thread and block width of the grid fill (copy of grid_fill_kernel's loop)
#include "../origami_amd/csrc/ogd_resize_math.h"
#include "../origami_amd/csrc/ogd_grid_lines.h"
template <int NT, int ROWS, int THREADS>
__global__ __launch_bounds__(THREADS) void grid_fill_variant(const float2* __restrict__ nodes, int h, int w,
                                                             double inv_s, float2* __restrict__ out, int Hd, int Wd,
                                                             int ybeg = 0) {
  // dense rows [ybeg, Hd) go to out rows 0 .. Hd - ybeg - 1
  const int X = (blockIdx.x * THREADS + threadIdx.x) * 2;
  if (X >= Wd) return;
  const bool pair = (X + 1 < Wd);
  const Tap tx0 = tap_x(X, inv_s, w);
  const Tap tx1 = tap_x(pair ? X + 1 : X, inv_s, w);
  const int Y0 = ybeg + blockIdx.y * ROWS, Y1 = min(Y0 + ROWS, Hd);
  int cur0 = -1, cur1 = -1;
  float2 ta0 = {0.f, 0.f}, ta1 = {0.f, 0.f}, tb0 = {0.f, 0.f}, tb1 = {0.f, 0.f};
  for (int Y = Y0; Y < Y1; ++Y) {
    const Tap ty = tap_y(Y, inv_s, h);
    if (ty.i0 != cur0 || ty.i1 != cur1) {
      const float2* r0 = nodes + (size_t)ty.i0 * w;
      const float2* r1 = nodes + (size_t)ty.i1 * w;
      float2 a = r0[tx0.i0], b = r0[tx0.i1], c = r0[tx1.i0], d = r0[tx1.i1];
      ta0 = make_float2(lerp2(a.x, tx0.w0, b.x, tx0.w1), lerp2(a.y, tx0.w0, b.y, tx0.w1));
      ta1 = make_float2(lerp2(c.x, tx1.w0, d.x, tx1.w1), lerp2(c.y, tx1.w0, d.y, tx1.w1));
      a = r1[tx0.i0], b = r1[tx0.i1], c = r1[tx1.i0], d = r1[tx1.i1];
      tb0 = make_float2(lerp2(a.x, tx0.w0, b.x, tx0.w1), lerp2(a.y, tx0.w0, b.y, tx0.w1));
      tb1 = make_float2(lerp2(c.x, tx1.w0, d.x, tx1.w1), lerp2(c.y, tx1.w0, d.y, tx1.w1));
      cur0 = ty.i0;
      cur1 = ty.i1;
    }
    f4a8 v;
    v.x = lerp2(ta0.x, ty.w0, tb0.x, ty.w1);
    v.y = lerp2(ta0.y, ty.w0, tb0.y, ty.w1);
    v.z = lerp2(ta1.x, ty.w0, tb1.x, ty.w1);
    v.w = lerp2(ta1.y, ty.w0, tb1.y, ty.w1);
    float2* o = out + (size_t)(Y - ybeg) * Wd + X;
    if (!pair)
      *o = make_float2(v.x, v.y);
    else if (NT == 1)
      __builtin_nontemporal_store(v, reinterpret_cast<f4a8*>(o));
    else
      *reinterpret_cast<f4a8*>(o) = v;
  }
}

struct Ctx {
  int W, H, Wd, Hd, iters;
  uint8_t *src, *dst, *ref;
  float* map;
  std::vector<uint8_t> h_ref, h_out;
};

// store ceiling: the map's bytes written as fully aligned float4, nothing computed
template <bool NT>
__global__ __launch_bounds__(256) void fill_only(float4* __restrict__ out4, size_t n4) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4* __restrict__ out = reinterpret_cast<f4*>(out4);
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const f4 v = {(float)i, 1.f, 2.f, 3.f};
    if (NT) __builtin_nontemporal_store(v, out + i);
    else out[i] = v;
  }
}

// block b writes the contiguous chunk [b * per, (b + 1) * per) float4, its threads side by side
template <bool NT>
__global__ __launch_bounds__(256) void fill_chunked(float4* __restrict__ out4, size_t n4, size_t per) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4* __restrict__ out = reinterpret_cast<f4*>(out4);
  const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n4 ? b0 + per : n4;
  for (size_t i = b0 + threadIdx.x; i < b1; i += 256) {
    const f4 v = {(float)i, 1.f, 2.f, 3.f};
    if (NT) __builtin_nontemporal_store(v, out + i);
    else out[i] = v;
  }
}

template <class F>
static void run(Ctx& c, const char* name, F launch, bool check = true) {
  if (const char* only = getenv("KB_ONLY"))
    if (!strstr(name, only)) return;
  CK(hipMemset(c.dst, 0xCD, (size_t)c.Hd * c.Wd));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // best of KB_REPS timed loops (clocks settle over tens of ms: one short loop is good to +-5 % only)
  const int reps = getenv("KB_REPS") ? atoi(getenv("KB_REPS")) : 1;
  float ms = 1e30f;
  for (int rep = 0; rep < reps; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < c.iters; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float t = 0;
    CK(hipEventElapsedTime(&t, e0, e1));
    ms = std::min(ms, t / c.iters);
  }
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  // ALGORITHMIC bytes of what was launched: every kernel is priced against its own figure,
  // never against K2's (a fused launch moves 2 B/px, not 10)
  const double px = (double)c.Hd * c.Wd, srcb = (double)c.H * c.W;
  const double k2 = px * 9 + srcb, k1a = px * 8, fused = px + srcb;
  double bytes = k2;
  const std::string nm(name);
  auto has = [&](const char* t) { return nm.find(t) != std::string::npos; };
  if (has("fused")) bytes = fused;
  else if (has("K1a alone") || has("map-stream only") || has("rows R=")) bytes = k1a;
  else if (has("K1a + K2") || has("K1a+K2") || has("bands") || has("pipeline")) bytes = k1a + k2;
  else if (has("map-stream tile")) bytes = px * 9;
  size_t bad = 0;
  if (check) {
    CK(hipMemcpy(c.h_out.data(), c.dst, c.h_out.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < c.h_out.size(); ++i) bad += c.h_out[i] != c.h_ref[i];
  }
  printf("%-34s %8.4f ms  %6.1f MB own  %7.1f GB/s  %5.1f%% of 8 TB/s  %s\n", name, ms, bytes / 1e6, bytes / ms / 1e6,
         bytes / ms / 1e6 / 80.0, check ? (bad ? "MISMATCH" : "ok") : "-");
  if (bad) printf("    %zu bytes differ\n", bad);
  fflush(stdout);
}

int main(int argc, char** argv) {
  Ctx c;
  c.W = argc > 1 ? atoi(argv[1]) : 6000;
  c.H = argc > 2 ? atoi(argv[2]) : 8000;
  const int cell = argc > 3 ? atoi(argv[3]) : 25;
  c.iters = argc > 4 ? atoi(argv[4]) : 20;
  const double A = argc > 5 ? atof(argv[5]) : 0.03;
  OK(ogd_init(0));
  const int margin = 1;
  const int w = (c.W + cell - 1) / cell + 1 + 2 * margin, h = (c.H + cell - 1) / cell + 1 + 2 * margin;
  c.Wd = w * cell;
  c.Hd = h * cell;
  std::vector<float> nodes((size_t)h * w * 2);
  const double B = A, PI2 = 6.283185307179586;
  for (int gy = 0; gy < h; ++gy)
    for (int gx = 0; gx < w; ++gx) {
      const double X = (gx - margin) * cell, Y = (gy - margin) * cell;
      const double y = Y + A * c.W * (cos(PI2 * Y / c.H) - cos(PI2 * Y / c.H + X / c.W));
      const double x = X - B * c.H * (cos(PI2 * X / c.W * 2) - cos(PI2 * X / c.W * 2 + Y / c.H));
      nodes[((size_t)gy * w + gx) * 2] = (float)x;
      nodes[((size_t)gy * w + gx) * 2 + 1] = (float)y;
    }
  std::vector<uint8_t> page((size_t)c.H * c.W);
  uint32_t s = 12345;
  for (auto& v : page) {
    s = s * 1664525u + 1013904223u;
    v = (uint8_t)(s >> 24);
  }
  float* d_nodes;
  CK(hipMalloc(&d_nodes, nodes.size() * 4));
  CK(hipMemcpy(d_nodes, nodes.data(), nodes.size() * 4, hipMemcpyHostToDevice));
  // KB_MAP_OFF (pixels of 8 B) / KB_DST_OFF (bytes): buffers moved off their 128-byte alignment
  const int map_off = getenv("KB_MAP_OFF") ? atoi(getenv("KB_MAP_OFF")) : 0;
  const int dst_off = getenv("KB_DST_OFF") ? atoi(getenv("KB_DST_OFF")) : 0;
  CK(hipMalloc(&c.map, (size_t)c.Hd * c.Wd * 8 + 1024));
  c.map += 2 * map_off;
  CK(hipMalloc(&c.src, page.size()));
  CK(hipMemcpy(c.src, page.data(), page.size(), hipMemcpyHostToDevice));
  CK(hipMalloc(&c.dst, (size_t)c.Hd * c.Wd + 1024));
  c.dst += dst_off;
  OK(ogd_grid_fill_dev(d_nodes, h, w, cell, 1.f, 1.f, c.map, nullptr));
  CK(hipDeviceSynchronize());
  c.h_ref.resize((size_t)c.Hd * c.Wd);
  c.h_out.resize(c.h_ref.size());
  printf("page %dx%d, cell %d, dense %dx%d (%.1f MP), warp amplitude %.3f rad\n", c.W, c.H, cell, c.Wd,
         c.Hd, c.Hd * (double)c.Wd / 1e6, A);

  auto lib = [&] {
    OK(ogd_remap_u8_dev(c.src, c.H, c.W, 1, c.W, c.map, c.Hd, c.Wd, c.dst, c.Wd, OGD_INTER_LINEAR, nullptr));
  };
  lib();
  CK(hipMemcpy(c.h_ref.data(), c.dst, c.h_ref.size(), hipMemcpyDeviceToHost));
  run(c, "library ogd_remap_u8_dev", lib);
  run(c, "library ogd_dewarp_u8_dev (fused)", [&] {
    OK(ogd_dewarp_u8_dev(c.src, c.H, c.W, 1, c.W, d_nodes, h, w, cell, 1.f, 1.f, c.dst, c.Wd, OGD_INTER_LINEAR, nullptr));
  });

  // K1a -> K2 page pipeline, whole map vs row bands through one band buffer (Infinity Cache residency)
  for (int nbands : {1, 2, 3, 4, 6, 8, 12, 16, 32}) {
    char name[64];
    snprintf(name, sizeof name, "pipeline K1a+K2, %d band(s)", nbands);
    const int rows = ((c.Hd + nbands - 1) / nbands + 7) / 8 * 8;
    run(c, name, [&] {
      for (int y0 = 0; y0 < c.Hd; y0 += rows) {
        const int n = std::min(rows, c.Hd - y0);
        OK(ogd_grid_fill_rows_dev(d_nodes, h, w, cell, 1.f, 1.f, y0, n, c.map, nullptr));
        OK(ogd_remap_u8_dev(c.src, c.H, c.W, 1, c.W, c.map, n, c.Wd, c.dst + (size_t)y0 * c.Wd, c.Wd,
                            OGD_INTER_LINEAR, nullptr));
      }
    });
  }
  // the same with plain stores in K1a: the band buffer is rewritten at the same addresses band after band, so
  // it can live in the Infinity Cache (non-temporal stores go past it)
#define BAND_VARIANT(NT, NBANDS)                                                                          \
  {                                                                                                       \
    const int rows = ((c.Hd + NBANDS - 1) / NBANDS + 7) / 8 * 8;                                          \
    run(c, "bands K1a nt=" #NT " + K2, " #NBANDS " band(s)", [&] {                                        \
      for (int y0 = 0; y0 < c.Hd; y0 += rows) {                                                           \
        const int n = std::min(rows, c.Hd - y0);                                                          \
        hipLaunchKernelGGL((grid_fill_variant<NT, 4, 512>), dim3((c.Wd + 1023) / 1024, (n + 3) / 4),      \
                           dim3(512), 0, 0, (const float2*)d_nodes, h, w, 1.0 / cell, (float2*)c.map,     \
                           y0 + n, c.Wd, y0);                                                             \
        OK(ogd_remap_u8_dev(c.src, c.H, c.W, 1, c.W, c.map, n, c.Wd, c.dst + (size_t)y0 * c.Wd, c.Wd,     \
                            OGD_INTER_LINEAR, nullptr));                                                  \
      }                                                                                                   \
    });                                                                                                   \
  }
  BAND_VARIANT(1, 1)
  BAND_VARIANT(0, 1)
  BAND_VARIANT(0, 2)
  BAND_VARIANT(0, 3)
  BAND_VARIANT(0, 4)
  BAND_VARIANT(0, 6)
  BAND_VARIANT(0, 8)
  BAND_VARIANT(1, 4)
  OK(ogd_grid_fill_dev(d_nodes, h, w, cell, 1.f, 1.f, c.map, nullptr));
  CK(hipDeviceSynchronize());

  // K1a lab: alone (check = false: the map is the output) and followed by the library's K2
#define K1A_VARIANT(NT, ROWS, THREADS)                                                                    \
  {                                                                                                       \
    auto k1a = [&] {                                                                                      \
      hipLaunchKernelGGL((grid_fill_variant<NT, ROWS, THREADS>),                                          \
                         dim3((c.Wd + 2 * THREADS - 1) / (2 * THREADS), (c.Hd + ROWS - 1) / ROWS),        \
                         dim3(THREADS), 0, 0, (const float2*)d_nodes, h, w, 1.0 / cell, (float2*)c.map,   \
                         c.Hd, c.Wd);                                                                     \
    };                                                                                                    \
    run(c, "K1a alone nt=" #NT " rows=" #ROWS " threads=" #THREADS, k1a, false);                          \
    run(c, "K1a+K2    nt=" #NT " rows=" #ROWS " threads=" #THREADS, [&] { k1a(); lib(); });               \
  }
  K1A_VARIANT(1, 16, 256)
  K1A_VARIANT(0, 16, 256)
  K1A_VARIANT(1, 8, 256)
  K1A_VARIANT(1, 4, 256)
  K1A_VARIANT(1, 8, 512)
  K1A_VARIANT(1, 4, 512)
  K1A_VARIANT(1, 8, 1024)
  K1A_VARIANT(1, 4, 1024)
  K1A_VARIANT(1, 2, 1024)
  K1A_VARIANT(1, 1, 1024)
  K1A_VARIANT(0, 4, 1024)
  for (int blocks : {2048, 8192, 65536}) {
    char name[64];
    snprintf(name, sizeof name, "K1a alone pure fill nt=1 blocks=%d", blocks);
    run(c, name, [&] { hipLaunchKernelGGL(fill_only<true>, dim3(blocks), dim3(256), 0, 0, (float4*)c.map, (size_t)c.Hd * c.Wd / 2); }, false);
    snprintf(name, sizeof name, "K1a alone pure fill nt=0 blocks=%d", blocks);
    run(c, name, [&] { hipLaunchKernelGGL(fill_only<false>, dim3(blocks), dim3(256), 0, 0, (float4*)c.map, (size_t)c.Hd * c.Wd / 2); }, false);
  }
#define LINES_VARIANT(TW, TR, NTH)                                                                      \
  {                                                                                                       \
    auto k1a = [&] {                                                                                      \
      hipLaunchKernelGGL((grid_fill_lines_kernel<TW, TR, true, NTH>), dim3((c.Wd + 15 + TW - 1) / TW, (c.Hd + TR - 1) / TR), \
                         dim3(NTH), 0, 0, (const float2*)d_nodes, h, w, 1.0 / cell, 1.f, 1.f, (float2*)c.map, 0, \
                         c.Hd, c.Wd);                                                                     \
    };                                                                                                    \
    run(c, "K1a alone lines TW=" #TW " TR=" #TR " threads=" #NTH, k1a, false);                            \
    run(c, "K1a+K2    lines TW=" #TW " TR=" #TR " threads=" #NTH, [&] { k1a(); lib(); });                 \
  }
  LINES_VARIANT(496, 8, 256)
  {  // plain instead of non-temporal stores
    auto k1a = [&] {
      hipLaunchKernelGGL((grid_fill_lines_kernel<496, 8, false, 256>), dim3((c.Wd + 15 + 495) / 496, (c.Hd + 7) / 8),
                         dim3(256), 0, 0, (const float2*)d_nodes, h, w, 1.0 / cell, 1.f, 1.f, (float2*)c.map, 0,
                         c.Hd, c.Wd);
    };
    run(c, "K1a alone lines TW=496 TR=8 threads=256 plain stores", k1a, false);
    run(c, "K1a+K2    lines TW=496 TR=8 threads=256 plain stores", [&] { k1a(); lib(); });
  }
  LINES_VARIANT(496, 8, 512)
  LINES_VARIANT(496, 8, 1024)
  LINES_VARIANT(496, 4, 1024)
  LINES_VARIANT(496, 4, 512)
  LINES_VARIANT(496, 2, 512)
  LINES_VARIANT(496, 12, 768)
  LINES_VARIANT(496, 16, 1024)
  LINES_VARIANT(240, 8, 1024)
  LINES_VARIANT(240, 8, 512)
  LINES_VARIANT(368, 8, 1024)
  LINES_VARIANT(512, 8, 1024)
  for (int kb : {4, 8, 16, 32, 64, 256, 1024}) {
    char name[64];
    const size_t n4 = (size_t)c.Hd * c.Wd / 2, per = (size_t)kb * 1024 / 16;
    snprintf(name, sizeof name, "K1a alone chunk fill nt=1 %d KB per block", kb);
    run(c, name, [&] { hipLaunchKernelGGL(fill_chunked<true>, dim3((unsigned)((n4 + per - 1) / per)), dim3(256), 0, 0, (float4*)c.map, n4, per); }, false);
  }
  for (int blocks : {16384, 32768, 65536, 95800}) {
    char name[64];
    snprintf(name, sizeof name, "K1a alone pure fill nt=1 blocks=%d", blocks);
    run(c, name, [&] { hipLaunchKernelGGL(fill_only<true>, dim3(blocks), dim3(256), 0, 0, (float4*)c.map, (size_t)c.Hd * c.Wd / 2); }, false);
  }
  run(c, "library K1a alone", [&] { OK(ogd_grid_fill_dev(d_nodes, h, w, cell, 1.f, 1.f, c.map, nullptr)); }, false);
  run(c, "library K1a + K2", [&] { OK(ogd_grid_fill_dev(d_nodes, h, w, cell, 1.f, 1.f, c.map, nullptr)); lib(); });

  const int nbx = (c.Wd + 255) / 256;
#define ROWS_VARIANT(R, NT, XCD)                                                                        \
  {                                                                                                     \
    const int nby = (c.Hd + 4 * R - 1) / (4 * R), nb = nbx * nby;                                       \
    const int grid = XCD ? ((nb + 7) / 8) * 8 : nb;                                                     \
    run(c, "rows R=" #R " nt=" #NT " xcd=" #XCD, [&] {                                                  \
      hipLaunchKernelGGL((remap_rows<1, 1, R, NT, XCD>), dim3(grid), dim3(256), 0, 0, c.src, c.H, c.W,  \
                         (ptrdiff_t)c.W, c.map, c.Hd, c.Wd, c.dst, (ptrdiff_t)c.Wd, nbx, nb);           \
    });                                                                                                 \
  }
  ROWS_VARIANT(1, false, false)
  ROWS_VARIANT(2, false, false)
  ROWS_VARIANT(4, false, false)
  ROWS_VARIANT(8, false, false)
  ROWS_VARIANT(2, true, false)
  ROWS_VARIANT(4, true, false)
  ROWS_VARIANT(8, true, false)
  ROWS_VARIANT(2, true, true)
  ROWS_VARIANT(4, true, true)
  ROWS_VARIANT(8, true, true)
  ROWS_VARIANT(4, false, true)
#define WAVE_VARIANT(LX, RPT, LDSW)                                                                     \
  {                                                                                                     \
    const int tw = 4 * LX, th = (64 / LX) * RPT;                                                        \
    const int nbx2 = (c.Wd + tw - 1) / tw, nby = (c.Hd + th - 1) / th, nb = nbx2 * nby;                 \
    run(c, "wave LX=" #LX " RPT=" #RPT " ldsw=" #LDSW, [&] {                                            \
      hipLaunchKernelGGL((remap_u8_wave_kernel<1, 1, LX, RPT, LDSW, MapGlobalPlain>), dim3((nb + RL_WPB - 1) / RL_WPB),    \
                         dim3(64 * RL_WPB), 0, 0, c.src, c.H, c.W, (ptrdiff_t)c.W, MapGlobalPlain{{c.map, c.Hd, c.Wd}},  \
                         c.Hd, c.Wd, c.dst, (ptrdiff_t)c.Wd, nbx2, nb);                                                                    \
    });                                                                                                 \
  }
  {  // nearest sampling through the same map: the memory side of K2 with a fraction of its integer work
    const int nbx2 = (c.Wd + 63) / 64, nby = (c.Hd + 7) / 8, nb = nbx2 * nby;
    run(c, "wave LX=16 RPT=2 nearest", [&] {
      hipLaunchKernelGGL((remap_u8_wave_kernel<1, 0, 16, 2, 4096, MapGlobalPlain>), dim3((nb + RL_WPB - 1) / RL_WPB), dim3(64 * RL_WPB), 0, 0,
                         c.src, c.H, c.W, (ptrdiff_t)c.W, MapGlobalPlain{{c.map, c.Hd, c.Wd}}, c.Hd, c.Wd, c.dst,
                         (ptrdiff_t)c.Wd, nbx2, nb);
    }, false);
  }
  WAVE_VARIANT(16, 2, 4096)
  // sheared tiles (what the library launches for one channel), other shapes
#define WAVE_SHEAR_VARIANT(LX, RPT, LDSW)                                                               \
  {                                                                                                     \
    const int tw = 4 * LX, th = (64 / LX) * RPT;                                                        \
    int step = (int)(c.Wd % 64);                                                                        \
    if (step >= 32) step -= 64;                                                                         \
    const int wrap = std::abs(step) * (th - 1) >= 64, bias = (wrap || step < 0) ? 64 : 0;               \
    const int reach = 63 + bias + (wrap ? 31 : std::max(0, step * (th - 1)));                           \
    const int nbx2 = (c.Wd + reach + tw - 1) / tw, nby = (c.Hd + th - 1) / th, nb = nbx2 * nby;         \
    run(c, "wave sheared LX=" #LX " RPT=" #RPT " ldsw=" #LDSW, [&] {                                    \
      hipLaunchKernelGGL((remap_u8_wave_kernel<1, 1, LX, RPT, LDSW, MapGlobal>),                        \
                         dim3((nb + RL_WPB - 1) / RL_WPB), dim3(64 * RL_WPB), 0, 0, c.src, c.H, c.W,    \
                         (ptrdiff_t)c.W, MapGlobal{c.map, c.Hd, c.Wd}, c.Hd, c.Wd, c.dst,               \
                         (ptrdiff_t)c.Wd, nbx2, nb, step, wrap, bias);                                  \
    });                                                                                                 \
  }
  WAVE_SHEAR_VARIANT(16, 2, 4096)
  WAVE_SHEAR_VARIANT(16, 1, 4096)
  WAVE_SHEAR_VARIANT(16, 4, 4096)
  WAVE_SHEAR_VARIANT(32, 1, 4096)
  WAVE_SHEAR_VARIANT(32, 2, 4096)
  WAVE_SHEAR_VARIANT(16, 2, 3072)
  WAVE_SHEAR_VARIANT(16, 2, 5120)
  WAVE_VARIANT(16, 2, 8192)
  WAVE_VARIANT(16, 4, 4096)
  WAVE_VARIANT(8, 2, 4096)
  WAVE_VARIANT(8, 4, 4096)
  WAVE_VARIANT(32, 2, 4096)
#define FUSED_VARIANT(LX, RPT, LDSW)                                                                    \
  {                                                                                                     \
    const int tw = 4 * LX, th = (64 / LX) * RPT;                                                        \
    const int nbx2 = (c.Wd + tw - 1) / tw, nby = (c.Hd + th - 1) / th, nb = nbx2 * nby;                 \
    const MapNodes mn{(const float2*)d_nodes, h, w, 1.0 / cell, 1.f, 1.f, c.Hd, c.Wd};                   \
    run(c, "fused LX=" #LX " RPT=" #RPT " ldsw=" #LDSW, [&] {                                           \
      hipLaunchKernelGGL((remap_u8_wave_kernel<1, 1, LX, RPT, LDSW, MapNodes>), dim3((nb + RL_WPB - 1) / RL_WPB),     \
                         dim3(64 * RL_WPB), 0, 0, c.src, c.H, c.W, (ptrdiff_t)c.W, mn, c.Hd, c.Wd, c.dst,       \
                         (ptrdiff_t)c.Wd, nbx2, nb);                                                    \
    });                                                                                                 \
  }
  {  // the same kernel with a non-unit scale (the multiplies by scx / scy are executed): timing only
    const int nbx2 = (c.Wd + 31) / 32, nby = (c.Hd + 31) / 32, nb = nbx2 * nby;
    const MapNodes mn{(const float2*)d_nodes, h, w, 1.0 / cell, 1.0000001f, 1.0000001f, c.Hd, c.Wd};
    run(c, "fused LX=8 RPT=4 scaled map", [&] {
      hipLaunchKernelGGL((remap_u8_wave_kernel<1, 1, 8, 4, 4096, MapNodes>), dim3((nb + RL_WPB - 1) / RL_WPB), dim3(64 * RL_WPB), 0, 0,
                         c.src, c.H, c.W, (ptrdiff_t)c.W, mn, c.Hd, c.Wd, c.dst, (ptrdiff_t)c.Wd, nbx2, nb);
    }, false);
  }
  {
    const int nbx2 = (c.Wd + 31) / 32, nby = (c.Hd + 31) / 32, nb = nbx2 * nby;
    const MapNodesUnit mn{(const float2*)d_nodes, h, w, 1.0 / cell, 1.f, 1.f, c.Hd, c.Wd};
    run(c, "fused LX=8 RPT=4 unit-scale kernel", [&] {
      hipLaunchKernelGGL((remap_u8_wave_kernel<1, 1, 8, 4, 4096, MapNodesUnit>), dim3((nb + RL_WPB - 1) / RL_WPB), dim3(64 * RL_WPB), 0, 0,
                         c.src, c.H, c.W, (ptrdiff_t)c.W, mn, c.Hd, c.Wd, c.dst, (ptrdiff_t)c.Wd, nbx2, nb);
    });
  }
  FUSED_VARIANT(8, 4, 4096)
  FUSED_VARIANT(8, 2, 4096)
  FUSED_VARIANT(4, 2, 4096)
  FUSED_VARIANT(8, 3, 4096)
  FUSED_VARIANT(8, 4, 3072)
  FUSED_VARIANT(8, 4, 5120)
#define STREAM_TILE(LX, RPT)                                                                           \
  {                                                                                                     \
    const int tw = 4 * LX, th = (64 / LX) * RPT;                                                        \
    const int nbx2 = (c.Wd + tw - 1) / tw, nby = (c.Hd + th - 1) / th, nb = nbx2 * nby;                 \
    run(c, "map-stream tile LX=" #LX " RPT=" #RPT, [&] {                                                \
      hipLaunchKernelGGL((map_stream_tile<LX, RPT>), dim3((nb + 3) / 4), dim3(256), 0, 0, c.map, c.Hd,  \
                         c.Wd, c.dst, (ptrdiff_t)c.Wd, nbx2, nb);                                       \
    }, false);                                                                                          \
  }
  STREAM_TILE(16, 2)
#define STREAM_TILE_WG(LX, RPT)                                                                        \
  {                                                                                                     \
    const int rb = 16 * LX, th = (64 / LX) * RPT;                                                       \
    const int nbx4 = (c.Wd + rb - 1) / rb, nby = (c.Hd + th - 1) / th;                                  \
    run(c, "map-stream tile wg-store LX=" #LX " RPT=" #RPT, [&] {                                       \
      hipLaunchKernelGGL((map_stream_tile_wg<LX, RPT>), dim3(nbx4 * nby), dim3(256), 0, 0, c.map, c.Hd, \
                         c.Wd, c.dst, (ptrdiff_t)c.Wd, nbx4);                                           \
    }, false);                                                                                          \
  }
  STREAM_TILE_WG(16, 2)
#define STREAM_TILE_SHEAR(LX, RPT, SM)                                                                  \
  {                                                                                                     \
    const int tw = 4 * LX, th = (64 / LX) * RPT;                                                        \
    const int nbx2 = (c.Wd + 63 + tw - 1) / tw, nby = (c.Hd + th - 1) / th, nb = nbx2 * nby;            \
    run(c, "map-stream tile shear LX=" #LX " RPT=" #RPT " map=" #SM, [&] {                              \
      hipLaunchKernelGGL((map_stream_tile_shear<LX, RPT, SM>), dim3((nb + 3) / 4), dim3(256), 0, 0,     \
                         c.map, c.Hd, c.Wd, c.dst, (ptrdiff_t)c.Wd, nbx2, nb);                          \
    }, false);                                                                                          \
  }
  STREAM_TILE_SHEAR(16, 2, false)
  STREAM_TILE_SHEAR(16, 2, true)
  STREAM_TILE_WG(16, 4)
  STREAM_TILE(16, 4)
  STREAM_TILE(32, 2)
  STREAM_TILE(32, 4)
  STREAM_TILE(64, 1)
  STREAM_TILE(64, 2)
  STREAM_TILE(64, 4)
#define STREAM_VARIANT(R, NT)                                                                      \
  {                                                                                                \
    const int nby = (c.Hd + 4 * R - 1) / (4 * R), nb = nbx * nby;                                  \
    run(c, "map-stream only R=" #R " nt=" #NT, [&] {                                               \
      hipLaunchKernelGGL((map_stream<R, NT>), dim3(nb), dim3(256), 0, 0, c.map, c.Hd, c.Wd, c.dst, \
                         (ptrdiff_t)c.Wd, nbx, nb);                                                \
    }, false);                                                                                     \
  }
  STREAM_VARIANT(1, false)
  STREAM_VARIANT(4, false)
  STREAM_VARIANT(4, true)
  STREAM_VARIANT(8, true)
  return 0;
}
