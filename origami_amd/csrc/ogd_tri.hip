// K3 / K1b — piecewise-linear interpolation over a host-built Delaunay triangulation.
// ref: origami/core/lingrid.py:174-189 (Interpolator.__call__: clip, inner, extra, NaN merge),
//      :252-282 (dense lattice variant), :286-290 (scipy LinearNDInterpolator x2);
//      origami/core/dewarp.py:158-161 (Transformer.__call__), origami/batch/detect/dewarp.py:26.
//
// Arithmetic follows scipy.interpolate.LinearNDInterpolator (interpnd / qhull.pyx), in fp64:
//   c0 = T00*(x - r0) + T01*(y - r1);  c1 = T10*(x - r0) + T11*(y - r1);  c2 = (1 - c0) - c1
//   inside  <=>  -eps <= c_j <= 1 + eps for all j,  eps = 100 * DBL_EPSILON
//   value_k = ((c0 * v0k) + c1 * v1k) + c2 * v2k            (no FMA contraction)
//   no containing simplex -> NaN.
// T / r come from scipy's Delaunay.transform (the host "topology" stage); the point locator is a
// uniform bucket grid over the triangulation's bounding box (CSR lists of the triangles that
// overlap each cell, ascending index), so a query tests ~2-4 triangles and takes the first hit.
// A point on a shared edge may resolve to a different neighbour than scipy's walk does; both give
// the same value up to the last bits (the interpolant is continuous).
//
// Triangle record (fp64): {T00, T01, T10, T11, r0, r1, v0[nv], v1[nv], v2[nv]} = 72 B (nv=1), 96 B (nv=2).
#include "ogd_common.h"
#include "ogd_tri_dev.h"
#include "ogd_lawson.h"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

namespace ogd {

template <int NV>
__global__ __launch_bounds__(256) void tri_eval_kernel(TriView inner, int has_inner, TriView extra,
                                                       Bounds b, const double2* __restrict__ q,
                                                       int64_t nq, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  double v[NV];
  const double2 p = q[i];
  eval_point<NV>(inner, has_inner, extra, b, p.x, p.y, v);
#pragma unroll
  for (int k = 0; k < NV; ++k) out[i * NV + k] = v[k];
}

// One wavefront per ring (polygon shell, hole or linestring); lanes stride over its vertices, so a
// wave's queries are spatially coherent and share bucket cells / triangle records in L1.
template <int NV>
__global__ __launch_bounds__(256) void tri_eval_ragged_kernel(TriView inner, int has_inner,
                                                              TriView extra, Bounds b,
                                                              const double2* __restrict__ q,
                                                              const int64_t* __restrict__ offsets,
                                                              int nrings, double* __restrict__ out) {
  const int ring = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  if (ring >= nrings) return;
  const int lane = threadIdx.x & 63;
  const int64_t beg = offsets[ring], end = offsets[ring + 1];
  for (int64_t i = beg + lane; i < end; i += 64) {
    double v[NV];
    const double2 p = q[i];
    eval_point<NV>(inner, has_inner, extra, b, p.x, p.y, v);
#pragma unroll
    for (int k = 0; k < NV; ++k) out[i * NV + k] = v[k];
  }
}

// K1b — barycentric rasteriser on a lattice.  One workgroup per RS_TW x RS_TH lattice tile, RS_PX lattice
// rows per thread.  The tile's (clipped) extent selects a rectangle of bucket cells.  The bucket grid is a
// row-major CSR, so the triangle lists of one cell ROW of that rectangle are one contiguous range of
// cell_tris: the workgroup stages the records of those ranges in LDS — all cell rows of the tile at once when
// they fit (three dependent memory round trips per tile), else row by row in chunks of RS_CAP records — and
// every lattice point tests the staged records of its own cell in list order — the order tri_locate uses, so the rasteriser is
// bit-identical to point queries.  Points the inner triangulation leaves open take a second pass over
// the extrapolation triangulation.
#ifndef OGD_RS_TH
#define OGD_RS_TH 16
#endif
constexpr int RS_TW = 64, RS_TH = OGD_RS_TH, RS_THREADS = 256, RS_PX = RS_TH / (RS_THREADS / RS_TW), RS_CAP = 160;

__device__ __forceinline__ void cell_xy(const TriView& t, double x, double y, int& cx, int& cy) {
  cx = min(max((int)floor((x - t.gx0) * t.inv_cw), 0), t.gnx - 1);  // as cell_of, without the division to split it
  cy = min(max((int)floor((y - t.gy0) * t.inv_ch), 0), t.gny - 1);
}

constexpr int RS_MAXROWS = 8;                                // cell rows staged in one go
constexpr int RS_META = 3 * RS_MAXROWS + 1 + RS_CAP;         // ints: row begin | row length | prefix | triangle ids

// One pass of a tile over one triangulation.  Fast path (the tile's cell rectangle has <= RS_MAXROWS rows and
// <= RS_CAP list entries, i.e. practically always): the list ranges of ALL its cell rows are staged at once —
// three dependent memory round trips per tile (row ranges, triangle ids, records) instead of three per cell
// row.  Otherwise cell row by cell row, in chunks.  Either way every lattice point tests the records of its
// own cell in list order.
template <int NV>
__device__ __forceinline__ void raster_pass(const TriView& t, double* lds, int* meta, double x,
                                            const double (&y)[RS_PX], bool (&found)[RS_PX],
                                            double (&val)[RS_PX][NV], double tx0, double ty0, double tx1,
                                            double ty1) {
  constexpr int REC = 6 + 3 * NV;
  const int tid = threadIdx.x;
  int* rb = meta;
  int* rlen = meta + RS_MAXROWS;
  int* pre = meta + 2 * RS_MAXROWS;  // RS_MAXROWS + 1 entries
  int* ids = meta + 3 * RS_MAXROWS + 1;
  // cell rectangle touched by this tile (workgroup-uniform: from the tile corners; the cell index is monotone)
  int cx0, cy0, cx1, cy1;
  cell_xy(t, tx0, ty0, cx0, cy0);
  cell_xy(t, tx1, ty1, cx1, cy1);
  const int nr = cy1 - cy0 + 1;
  // every open point's own list range (loads in flight while the tile's lists are staged)
  int ocx, ocy[RS_PX], ob[RS_PX], oe[RS_PX];
#pragma unroll
  for (int j = 0; j < RS_PX; ++j) {
    cell_xy(t, x, y[j], ocx, ocy[j]);
    const int c = ocy[j] * t.gnx + ocx;
    ob[j] = found[j] ? 0 : t.cell_start[c];
    oe[j] = found[j] ? 0 : t.cell_start[c + 1];
  }
  __syncthreads();  // the previous pass is done with meta / lds
  if (tid < min(nr, RS_MAXROWS)) {
    const int b = t.cell_start[(cy0 + tid) * t.gnx + cx0];
    rb[tid] = b;
    rlen[tid] = t.cell_start[(cy0 + tid) * t.gnx + cx1 + 1] - b;
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int r = 0; r < min(nr, RS_MAXROWS); ++r) {
      pre[r] = run;
      run += rlen[r];
    }
    pre[min(nr, RS_MAXROWS)] = run;
  }
  __syncthreads();
  const int total = pre[min(nr, RS_MAXROWS)];
  if (nr <= RS_MAXROWS && total <= RS_CAP) {  // workgroup-uniform
    for (int k = tid; k < total; k += RS_THREADS) {
      int r = 0;
      while (k >= pre[r + 1]) ++r;
      ids[k] = t.cell_tris[rb[r] + k - pre[r]];
    }
    __syncthreads();
    for (int idx = tid; idx < total * REC; idx += RS_THREADS) {
      const int k = idx / REC, f = idx - k * REC;
      lds[idx] = t.rec[(size_t)ids[k] * REC + f];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_PX; ++j) {
      if (found[j]) continue;
      const int r = ocy[j] - cy0;
      const int k0 = pre[r] + (ob[j] - rb[r]), k1 = k0 + (oe[j] - ob[j]);
      for (int k = k0; k < k1; ++k)
        if (tri_try<NV>(lds + k * REC, x, y[j], val[j])) {
          found[j] = true;
          break;
        }
    }
    return;
  }
  for (int cy = cy0; cy <= cy1; ++cy) {
    const int b = t.cell_start[cy * t.gnx + cx0], e = t.cell_start[cy * t.gnx + cx1 + 1];
    for (int base = b; base < e; base += RS_CAP) {
      const int n = min(RS_CAP, e - base);
      __syncthreads();  // previous chunk fully consumed
      for (int idx = tid; idx < n * REC; idx += RS_THREADS) {
        const int k = idx / REC, f = idx - k * REC;
        lds[idx] = t.rec[(size_t)t.cell_tris[base + k] * REC + f];
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < RS_PX; ++j) {
        if (found[j] || ocy[j] != cy) continue;
        const int k0 = max(ob[j], base) - base, k1 = min(oe[j], base + n) - base;
        for (int k = k0; k < k1; ++k)
          if (tri_try<NV>(lds + k * REC, x, y[j], val[j])) {
            found[j] = true;
            break;
          }
      }
    }
  }
}

template <int NV, typename OutT>
__global__ __launch_bounds__(RS_THREADS) void tri_rasterize_kernel(
    TriView inner, int has_inner, TriView extra, Bounds b, double x0, double y0, double step, int nx,
    int ny, OutT* __restrict__ out, int64_t stride_x, int64_t stride_y) {
  __shared__ double lds[RS_CAP * (6 + 3 * NV)];
  __shared__ int meta[RS_META];
  const int lx = threadIdx.x % RS_TW, ly = threadIdx.x / RS_TW;
  const int ix = blockIdx.x * RS_TW + lx, iy0 = blockIdx.y * RS_TH + ly;  // rows iy0 + j * (RS_TH / RS_PX)
  constexpr int RSTEP = RS_THREADS / RS_TW;
  // lattice coordinates: x0 + i*step is exact for the integer lattices the reference uses
  double x = x0 + (double)min(ix, nx - 1) * step;
  double y[RS_PX];
  bool found[RS_PX];
  double val[RS_PX][NV];
#pragma unroll
  for (int j = 0; j < RS_PX; ++j) {
    const int iy = iy0 + j * RSTEP;
    y[j] = y0 + (double)min(iy, ny - 1) * step;
    found[j] = !(ix < nx && iy < ny);  // points beyond the lattice take no part
#pragma unroll
    for (int k = 0; k < NV; ++k) val[j][k] = __longlong_as_double(0x7ff8000000000000LL);
  }
  const int ixl = min((int)(blockIdx.x * RS_TW + RS_TW - 1), nx - 1);
  const int iyl = min((int)(blockIdx.y * RS_TH + RS_TH - 1), ny - 1);
  double tx0 = x0 + (double)(blockIdx.x * RS_TW) * step, ty0 = y0 + (double)(blockIdx.y * RS_TH) * step;
  double tx1 = x0 + (double)ixl * step, ty1 = y0 + (double)iyl * step;
  if (step < 0) {
    double t = tx0; tx0 = tx1; tx1 = t;
    t = ty0; ty0 = ty1; ty1 = t;
  }
  if (b.clip) {
    x = clipd(x, b.minx, b.maxx);
#pragma unroll
    for (int j = 0; j < RS_PX; ++j) y[j] = clipd(y[j], b.miny, b.maxy);
    tx0 = clipd(tx0, b.minx, b.maxx);
    tx1 = clipd(tx1, b.minx, b.maxx);
    ty0 = clipd(ty0, b.miny, b.maxy);
    ty1 = clipd(ty1, b.miny, b.maxy);
  }
  if (has_inner) raster_pass<NV>(inner, lds, meta, x, y, found, val, tx0, ty0, tx1, ty1);
  bool open = false;
#pragma unroll
  for (int j = 0; j < RS_PX; ++j) open = open || !found[j];
  if (__syncthreads_or(open)) raster_pass<NV>(extra, lds, meta, x, y, found, val, tx0, ty0, tx1, ty1);
  if (ix < nx) {
#pragma unroll
    for (int j = 0; j < RS_PX; ++j) {
      const int iy = iy0 + j * RSTEP;
      if (iy < ny) {
        OutT* o = out + (int64_t)ix * stride_x + (int64_t)iy * stride_y;
#pragma unroll
        for (int k = 0; k < NV; ++k) o[k] = (OutT)val[j][k];
      }
    }
  }
}

// Ring validity, the O(n^2) part (ref: origami/batch/detect/dewarp.py:35, `geom.is_valid`: a warped polygon
// ring must not touch or cross itself).  One wavefront per ring: segment i is wave-uniform, the lanes take
// the segments j >= i + 2 (the closing pair (0, n-1) is adjacent); orientation signs in fp64 without FMA
// contraction, the arithmetic of origami_amd/geometry.py:_segments_cross.
__device__ __forceinline__ int orient_sign(double px, double py, double qx, double qy, double sx, double sy) {
  const double v = __dsub_rn(__dmul_rn(qx - px, sy - py), __dmul_rn(qy - py, sx - px));
  return (v > 0.0) - (v < 0.0);
}

__device__ __forceinline__ bool on_segment(double px, double py, double qx, double qy, double sx, double sy, int o) {
  return o == 0 && fmin(px, qx) <= sx && sx <= fmax(px, qx) && fmin(py, qy) <= sy && sy <= fmax(py, qy);
}

__global__ __launch_bounds__(256) void rings_self_cross_kernel(const double2* __restrict__ q,
                                                               const int64_t* __restrict__ offsets, int nrings,
                                                               uint8_t* __restrict__ flags) {
  const int ring = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  if (ring >= nrings) return;
  const int lane = threadIdx.x & 63;
  const double2* r = q + offsets[ring];
  const int n = (int)(offsets[ring + 1] - offsets[ring]) - 1;  // segments of the closed ring
  bool hit = false;
  if (n >= 3) {
    for (int i = 0; i + 2 < n; ++i) {
      const double2 a = r[i], b = r[i + 1];
      const int jend = i == 0 ? n - 1 : n;  // segment n-1 is adjacent to segment 0
      for (int j = i + 2 + lane; j < jend; j += 64) {
        const double2 c = r[j], d = r[j + 1];
        const int o1 = orient_sign(a.x, a.y, b.x, b.y, c.x, c.y), o2 = orient_sign(a.x, a.y, b.x, b.y, d.x, d.y);
        const int o3 = orient_sign(c.x, c.y, d.x, d.y, a.x, a.y), o4 = orient_sign(c.x, c.y, d.x, d.y, b.x, b.y);
        const bool proper = o1 * o2 < 0 && o3 * o4 < 0;
        const bool touch = on_segment(a.x, a.y, b.x, b.y, c.x, c.y, o1) || on_segment(a.x, a.y, b.x, b.y, d.x, d.y, o2) ||
                           on_segment(c.x, c.y, d.x, d.y, a.x, a.y, o3) || on_segment(c.x, c.y, d.x, d.y, b.x, b.y, o4);
        hit = hit || proper || touch;
      }
    }
  }
  const bool any = __any(hit);
  if (lane == 0) flags[ring] = any ? 1 : 0;
}

// ---- host: records + bucket grid ----------------------------------------------------------------

__host__ __device__ static inline double edge_fn(double ax, double ay, double bx, double by, double px, double py) {
  return (bx - ax) * (py - ay) - (by - ay) * (px - ax);
}

// true if the triangle cannot touch the axis-aligned rectangle [x0,x1] x [y0,y1]
__host__ __device__ static inline bool tri_misses_rect(const double* X, const double* Y, double x0, double y0,
                                                       double x1, double y1) {
  const double area = edge_fn(X[0], Y[0], X[1], Y[1], X[2], Y[2]);
  const double sgn = area >= 0 ? 1.0 : -1.0;
  for (int k = 0; k < 3; ++k) {
    const int a = k, b = (k + 1) % 3;
    const double e0 = sgn * edge_fn(X[a], Y[a], X[b], Y[b], x0, y0);
    const double e1 = sgn * edge_fn(X[a], Y[a], X[b], Y[b], x1, y0);
    const double e2 = sgn * edge_fn(X[a], Y[a], X[b], Y[b], x0, y1);
    const double e3 = sgn * edge_fn(X[a], Y[a], X[b], Y[b], x1, y1);
    if (e0 < 0 && e1 < 0 && e2 < 0 && e3 < 0) return true;
  }
  return false;
}

// ---- bucket grid, built on the device --------------------------------------------------------------------
// A triangle goes into every cell its (slightly grown) shape touches.  Three passes over the triangles'
// cell rectangles — count, (scan), fill — with atomics, then every cell sorts its short list so that the
// result is the deterministic "ascending triangle index" order the queries and the rasteriser rely on.
// (157 k triangles of a broadsheet transformer: 15 ms on one host core, ~0.2 ms here.)
struct BucketGeom {
  int gnx, gny;
  double bx0, by0, cw, ch, mx, my;
};

// One cell of one triangle's cell rectangle: count or fill.
template <bool FILL>
__device__ __forceinline__ int bucket_visit(const BucketGeom& g, const double* X, const double* Y, bool one, int cx, int cy,
                                            int t, int* __restrict__ counter, int* __restrict__ cell_tris) {
  const double rx0 = g.bx0 + cx * g.cw - g.mx, rx1 = g.bx0 + (cx + 1) * g.cw + g.mx;
  const double ry0 = g.by0 + cy * g.ch - g.my, ry1 = g.by0 + (cy + 1) * g.ch + g.my;
  if (!one && tri_misses_rect(X, Y, rx0, ry0, rx1, ry1)) return 0;
  const int c = cy * g.gnx + cx;
  if (FILL)
    cell_tris[atomicAdd(&counter[c], 1)] = t;  // counter = running cursor of the cell
  else
    atomicAdd(&counter[c + 1], 1);  // counter[c + 1] = list length of cell c
  return 1;
}

// The cell rectangle of triangle t (empty for a degenerate simplex).
struct TriRect {
  double X[3], Y[3];
  int cx0, cy0, wdt;
  long long ncells;
};
__device__ __forceinline__ TriRect tri_rect(const double2* __restrict__ pts, const int32_t* __restrict__ simplices,
                                            const double* __restrict__ rec, int reclen, const BucketGeom& g, int t) {
  TriRect q;
  q.cx0 = q.cy0 = 0;
  q.wdt = 0;
  q.ncells = 0;
  for (int k = 0; k < 3; ++k) q.X[k] = q.Y[k] = 0;
  const double* r = rec + (size_t)t * reclen;
  for (int k = 0; k < 6; ++k)
    if (!isfinite(r[k])) return q;  // degenerate simplex: never contains a point
  const int32_t* s = simplices + 3 * (size_t)t;
  const double2 p0 = pts[s[0]], p1 = pts[s[1]], p2 = pts[s[2]];
  q.X[0] = p0.x; q.X[1] = p1.x; q.X[2] = p2.x;
  q.Y[0] = p0.y; q.Y[1] = p1.y; q.Y[2] = p2.y;
  const double tx0 = fmin(fmin(p0.x, p1.x), p2.x) - g.mx, tx1 = fmax(fmax(p0.x, p1.x), p2.x) + g.mx;
  const double ty0 = fmin(fmin(p0.y, p1.y), p2.y) - g.my, ty1 = fmax(fmax(p0.y, p1.y), p2.y) + g.my;
  q.cx0 = min(g.gnx - 1, max(0, (int)floor((tx0 - g.bx0) / g.cw)));
  const int cx1 = min(g.gnx - 1, max(0, (int)floor((tx1 - g.bx0) / g.cw)));
  q.cy0 = min(g.gny - 1, max(0, (int)floor((ty0 - g.by0) / g.ch)));
  const int cy1 = min(g.gny - 1, max(0, (int)floor((ty1 - g.by0) / g.ch)));
  q.wdt = cx1 - q.cx0 + 1;
  q.ncells = (long long)q.wdt * (cy1 - q.cy0 + 1);
  return q;
}

// Thread per triangle for the usual few-cell rectangles.  A triangle whose rectangle holds more than
// BUCKET_WAVE_MIN cells (the slivers of an extrapolation set, or of the pockets between a warped lattice and its
// hull: thousands of cells each, and neighbours in index, so that a few wavefronts would walk them all while
// the launch waits) goes on a list in the counting pass; bucket_big_kernel walks each with a wavefront of its own.
constexpr int BUCKET_WAVE_MIN = 48;

template <bool FILL>
__global__ __launch_bounds__(256) void bucket_scatter_kernel(const double2* __restrict__ pts,
                                                             const int32_t* __restrict__ simplices,
                                                             const double* __restrict__ rec, int reclen, int nt,
                                                             BucketGeom g, int* __restrict__ counter,
                                                             int* __restrict__ cell_tris,
                                                             unsigned long long* __restrict__ total,
                                                             int* __restrict__ big_list, int* __restrict__ nbig) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nt) return;
  const TriRect q = tri_rect(pts, simplices, rec, reclen, g, t);
  if (q.ncells == 0) return;
  if (q.ncells > BUCKET_WAVE_MIN) {
    if (!FILL) big_list[atomicAdd(nbig, 1)] = t;
    return;
  }
  const bool one = q.ncells == 1;
  const int hgt = (int)(q.ncells / q.wdt);
  unsigned long long mine = 0;
  for (int ry = 0; ry < hgt; ++ry)
    for (int rx = 0; rx < q.wdt; ++rx)
      mine += bucket_visit<FILL>(g, q.X, q.Y, one, q.cx0 + rx, q.cy0 + ry, t, counter, cell_tris);
  if (!FILL) atomicAdd(total, mine);  // 64-bit sum of all list lengths: the host refuses a grid past INT_MAX
}

template <bool FILL>
__global__ __launch_bounds__(256) void bucket_big_kernel(const double2* __restrict__ pts,
                                                         const int32_t* __restrict__ simplices,
                                                         const double* __restrict__ rec, int reclen, BucketGeom g,
                                                         int* __restrict__ counter, int* __restrict__ cell_tris,
                                                         unsigned long long* __restrict__ total,
                                                         const int* __restrict__ big_list, int nbig) {
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (k >= nbig) return;  // wave-uniform
  const int t = big_list[k];
  const TriRect q = tri_rect(pts, simplices, rec, reclen, g, t);
  unsigned long long mine = 0;
  for (long long i = lane; i < q.ncells; i += 64) {
    const int ry = (int)(i / q.wdt), rx = (int)(i - (long long)ry * q.wdt);
    mine += bucket_visit<FILL>(g, q.X, q.Y, false, q.cx0 + rx, q.cy0 + ry, t, counter, cell_tris);
  }
  if (!FILL && mine) atomicAdd(total, mine);
}

// Ascending triangle index inside every cell.  Most lists hold a handful of entries (insertion sort by the
// cell's thread); a cell under a fan of slivers holds hundreds, and is left to bucket_sort_long_kernel.
constexpr int BUCKET_SORT_LONG = 24;
constexpr int BUCKET_SORT_LDS = 2048;  // entries a wavefront ranks through LDS

__device__ __forceinline__ void insertion_sort(int* __restrict__ a, int b, int e) {
  for (int i = b + 1; i < e; ++i) {
    const int v = a[i];
    int j = i - 1;
    while (j >= b && a[j] > v) {
      a[j + 1] = a[j];
      --j;
    }
    a[j + 1] = v;
  }
}

__global__ __launch_bounds__(256) void bucket_sort_kernel(const int* __restrict__ cell_start, int ncell,
                                                          int* __restrict__ cell_tris, int* __restrict__ max_cell,
                                                          int* __restrict__ long_cells, int* __restrict__ nlong) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncell) return;
  const int b = cell_start[c], e = cell_start[c + 1];
  if (e - b > BUCKET_SORT_LONG)
    long_cells[atomicAdd(nlong, 1)] = c;
  else
    insertion_sort(cell_tris, b, e);
  if (e - b > 1) atomicMax(max_cell, e - b);
}

// One wavefront per long list: the entries are distinct, so the rank of an entry is the number of smaller ones.
__global__ __launch_bounds__(256) void bucket_sort_long_kernel(const int* __restrict__ cell_start,
                                                               int* __restrict__ cell_tris,
                                                               const int* __restrict__ long_cells,
                                                               const int* __restrict__ nlong) {
  __shared__ int s_all[4][BUCKET_SORT_LDS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int* sv = s_all[wave];
  const int n = *nlong, nwaves = gridDim.x * 4;
  for (int k = blockIdx.x * 4 + wave; k < n; k += nwaves) {  // wave-uniform
    const int c = long_cells[k];
    const int b = cell_start[c], len = cell_start[c + 1] - b;
    if (len > BUCKET_SORT_LDS) {  // pathological: thousands of triangles over one cell
      if (lane == 0) insertion_sort(cell_tris, b, b + len);
      continue;
    }
    for (int i = lane; i < len; i += 64) sv[i] = cell_tris[b + i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i = lane; i < len; i += 64) {
      const int v = sv[i];
      int rank = 0;
      for (int j = 0; j < len; ++j) rank += sv[j] < v;
      cell_tris[b + rank] = v;
    }
    __builtin_amdgcn_wave_barrier();  // sv is re-used by the next list
  }
}

template <class T>
static int upload(T** d, const std::vector<T>& v) {
  *d = nullptr;
  const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  hipError_t e = hipMalloc((void**)d, bytes);
  if (e != hipSuccess) return fail(OGD_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  if (!v.empty()) OGD_HIP(hipMemcpy(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return OGD_OK;
}

struct BindDevice {
  int prev = -1;
  bool ok = true;
  explicit BindDevice(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
  }
  ~BindDevice() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// Triangle record t = {T00, T01, T10, T11, r0, r1, v0[nv], v1[nv], v2[nv]}: the barycentric transform as given
// (scipy's Delaunay.transform layout) or, without one, the plain inverse of the edge matrix (columns v0 - v2, v1 - v2).
// scipy_lu (with transform == NULL): the transform scipy.spatial.Delaunay would hand out, computed here
// (simplex_transform_one, the source the host runs for ogd_lattice_delaunay / ogd_delaunay_points).
__global__ __launch_bounds__(256) void tri_records_kernel(const double2* __restrict__ pts, const int32_t* __restrict__ simplices,
                                                          const double* __restrict__ transform, const double* __restrict__ vals,
                                                          int nv, int nt, double* __restrict__ recs, int rec, bool scipy_lu) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nt) return;
  const int32_t s0 = simplices[3 * (size_t)t], s1 = simplices[3 * (size_t)t + 1], s2 = simplices[3 * (size_t)t + 2];
  double* r = recs + (size_t)t * rec;
  if (transform) {
    const double* T = transform + 6 * (size_t)t;
    for (int k = 0; k < 6; ++k) r[k] = T[k];
  } else if (scipy_lu) {
    const double2 p0 = pts[s0], p1 = pts[s1], p2 = pts[s2];
    const double a[2] = {p0.x, p0.y}, b[2] = {p1.x, p1.y}, c[2] = {p2.x, p2.y};
    double o[6];
    simplex_transform_one(a, b, c, o);
    for (int k = 0; k < 6; ++k) r[k] = o[k];
  } else {
    const double2 p0 = pts[s0], p1 = pts[s1], p2 = pts[s2];
    const double ax = __dsub_rn(p0.x, p2.x), bx = __dsub_rn(p1.x, p2.x);
    const double ay = __dsub_rn(p0.y, p2.y), by = __dsub_rn(p1.y, p2.y);
    const double det = __dsub_rn(__dmul_rn(ax, by), __dmul_rn(bx, ay));
    r[0] = by / det; r[1] = -bx / det; r[2] = -ay / det; r[3] = ax / det;
    r[4] = p2.x; r[5] = p2.y;
  }
  const int32_t s[3] = {s0, s1, s2};
  for (int j = 0; j < 3; ++j)
    for (int k = 0; k < nv; ++k) r[6 + j * nv + k] = vals[(size_t)s[j] * nv + k];
}

static int check_pair(const ogd_tri* inner, const ogd_tri* extra) {
  OGD_REQUIRE(extra != nullptr, "extra triangulation is NULL");
  OGD_REQUIRE(inner == nullptr || inner->nv == extra->nv, "inner/extra value widths differ (%d vs %d)",
              inner ? inner->nv : 0, extra->nv);
  OGD_REQUIRE(inner == nullptr || inner->device == extra->device, "inner/extra live on different devices");
  return OGD_OK;
}

}  // namespace ogd

using namespace ogd;

extern "C" {

static int tri_create(const double* pts, int n, const int32_t* simplices, int nt, const double* transform,
                      bool scipy_lu, const double* vals, int nv, ogd_tri** out) {
  OGD_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  OGD_REQUIRE(pts && simplices && vals, "null pointer");
  OGD_REQUIRE(n >= 3 && nt >= 1, "need at least 3 points and 1 simplex (got %d, %d)", n, nt);
  OGD_REQUIRE(nv == 1 || nv == 2, "value width %d not supported (1 or 2)", nv);
  for (int64_t k = 0; k < (int64_t)nt * 3; ++k)
    OGD_REQUIRE(simplices[k] >= 0 && simplices[k] < n, "simplex vertex %d out of range", simplices[k]);
  OGD_TRY(ensure_device());

  const int rec = 6 + 3 * nv;
  double bx0 = INFINITY, by0 = INFINITY, bx1 = -INFINITY, by1 = -INFINITY;
  for (int i = 0; i < n; ++i) {
    const double x = pts[2 * i], y = pts[2 * i + 1];
    OGD_REQUIRE(std::isfinite(x) && std::isfinite(y), "point %d is not finite", i);
    bx0 = std::min(bx0, x); bx1 = std::max(bx1, x);
    by0 = std::min(by0, y); by1 = std::max(by1, y);
  }

  // bucket grid: about four cells per triangle (most cells then lie inside one triangle and a query tests one
  // or two records; measured against 1, 2, 8, 16: DESIGN.md §4), aspect-matched, at most 2048 per side
  const double ex = std::max(bx1 - bx0, 1e-300), ey = std::max(by1 - by0, 1e-300);
  double density = 4.0;
  if (const char* dv = std::getenv("OGD_BUCKET_DENSITY")) density = std::max(0.25, std::atof(dv));  // experiments
  double fx = std::sqrt(density * (double)nt * ex / ey), fy = std::sqrt(density * (double)nt * ey / ex);
  const int gnx = (int)std::min(2048.0, std::max(1.0, std::floor(fx)));
  const int gny = (int)std::min(2048.0, std::max(1.0, std::floor(fy)));
  const double cw = ex / gnx, ch = ey / gny;
  const double mx = 1e-7 * cw, my = 1e-7 * ch;  // slack: float error in cell_of + the eps band
  const int64_t ncell = (int64_t)gnx * gny;
  const BucketGeom geom{gnx, gny, bx0, by0, cw, ch, mx, my};

  ogd_tri* h = new (std::nothrow) ogd_tri();
  if (!h) return fail(OGD_ERR_NOMEM, "out of host memory");
  (void)hipGetDevice(&h->device);
  h->nt = nt; h->nv = nv; h->rec = rec;
  h->gnx = gnx; h->gny = gny; h->gx0 = bx0; h->gy0 = by0;
  h->inv_cw = 1.0 / cw; h->inv_ch = 1.0 / ch;
  auto build = [&]() -> int {
    DevBuf dpts, dsimp, dcursor, dscalars, dbig, dtransform, dvals;
    // The four inputs: a triangulation of a few thousand triangles (the angle fields, the extrapolation sets: five of
    // a page's six) packs them into the thread's pinned staging buffer and uploads them with ONE asynchronous copy —
    // four synchronous copies from pageable memory are 250 us of such a handle's 450; a broadsheet lattice (15 MB) goes
    // up from where it lies.
    const size_t b_pts = (size_t)n * 2 * sizeof(double), b_simp = (size_t)nt * 3 * sizeof(int32_t);
    const size_t b_vals = (size_t)n * nv * sizeof(double), b_tr = transform ? (size_t)nt * 6 * sizeof(double) : 0;
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_simp = pad(b_pts), o_vals = o_simp + pad(b_simp), o_tr = o_vals + pad(b_vals), b_all = o_tr + pad(b_tr);
    char* stage = b_all <= ((size_t)1 << 20) ? static_cast<char*>(staging_host(b_all)) : nullptr;
    const double2* p_pts;
    const int32_t* p_simp;
    const double* p_vals;
    const double* p_tr = nullptr;
    if (stage) {
      OGD_TRY(dpts.alloc(b_all));
      std::memcpy(stage, pts, b_pts);
      std::memcpy(stage + o_simp, simplices, b_simp);
      std::memcpy(stage + o_vals, vals, b_vals);
      if (transform) std::memcpy(stage + o_tr, transform, b_tr);
      OGD_HIP(hipMemcpyAsync(dpts.p, stage, b_all, hipMemcpyHostToDevice, nullptr));
      char* base = dpts.as<char>();
      p_pts = reinterpret_cast<const double2*>(base);
      p_simp = reinterpret_cast<const int32_t*>(base + o_simp);
      p_vals = reinterpret_cast<const double*>(base + o_vals);
      if (transform) p_tr = reinterpret_cast<const double*>(base + o_tr);
    } else {
      OGD_TRY(dpts.alloc(b_pts));
      OGD_TRY(dsimp.alloc(b_simp));
      OGD_TRY(dvals.alloc(b_vals));
      OGD_HIP(hipMemcpy(dvals.p, vals, b_vals, hipMemcpyHostToDevice));
      if (transform) {
        OGD_TRY(dtransform.alloc(b_tr));
        OGD_HIP(hipMemcpy(dtransform.p, transform, b_tr, hipMemcpyHostToDevice));
        p_tr = dtransform.as<double>();
      }
      OGD_HIP(hipMemcpy(dpts.p, pts, b_pts, hipMemcpyHostToDevice));
      OGD_HIP(hipMemcpy(dsimp.p, simplices, b_simp, hipMemcpyHostToDevice));
      p_pts = dpts.as<double2>();
      p_simp = dsimp.as<int32_t>();
      p_vals = dvals.as<double>();
    }
    {
      const size_t bytes = (size_t)nt * rec * sizeof(double);
      hipError_t e = hipMalloc((void**)&h->d_rec, bytes);
      if (e != hipSuccess) return fail(OGD_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    OGD_TRY(dcursor.alloc((size_t)ncell * sizeof(int)));
    OGD_TRY(dscalars.alloc(32));  // long long total | int max_cell | int nbig | int nlong
    OGD_TRY(dbig.alloc((size_t)nt * sizeof(int)));
    OGD_HIP(hipMalloc((void**)&h->d_cell_start, (size_t)(ncell + 1) * sizeof(int)));
    OGD_HIP(hipMemsetAsync(h->d_cell_start, 0, (size_t)(ncell + 1) * sizeof(int), nullptr));
    OGD_HIP(hipMemsetAsync(dscalars.p, 0, 32, nullptr));
    unsigned long long* d_total = dscalars.as<unsigned long long>();
    int* d_nbig = reinterpret_cast<int*>(dscalars.as<char>() + 12);
    const dim3 tgrid((unsigned)((nt + 255) / 256)), tblock(256);
    // the triangle records are assembled on the device (a broadsheet Transformer has 157 k of them, 15 MB)
    hipLaunchKernelGGL(tri_records_kernel, tgrid, tblock, 0, nullptr, p_pts, p_simp,
                       p_tr, p_vals, nv, nt,
                       h->d_rec, rec, scipy_lu);
    OGD_TRY(check_launch("tri_records_kernel"));
    hipLaunchKernelGGL(bucket_scatter_kernel<false>, tgrid, tblock, 0, nullptr, p_pts,
                       p_simp, h->d_rec, rec, nt, geom, h->d_cell_start, (int*)nullptr, d_total,
                       dbig.as<int>(), d_nbig);
    OGD_TRY(check_launch("bucket_scatter_kernel<count>"));
    int nbig = 0;
    OGD_HIP(hipMemcpy(&nbig, d_nbig, sizeof nbig, hipMemcpyDeviceToHost));
    OGD_REQUIRE(nbig >= 0 && nbig <= nt, "bucket grid: %d large triangles of %d", nbig, nt);
    const dim3 bgrid((unsigned)((nbig + 3) / 4));
    if (nbig > 0) {
      hipLaunchKernelGGL(bucket_big_kernel<false>, bgrid, tblock, 0, nullptr, p_pts, p_simp,
                         h->d_rec, rec, geom, h->d_cell_start, (int*)nullptr, d_total, dbig.as<int>(), nbig);
      OGD_TRY(check_launch("bucket_big_kernel<count>"));
    }
    unsigned long long total64 = 0;
    OGD_HIP(hipMemcpy(&total64, dscalars.p, sizeof total64, hipMemcpyDeviceToHost));
    OGD_REQUIRE(total64 < (unsigned long long)INT_MAX, "bucket grid overflow");
    const int total = (int)total64;
    // list lengths -> list starts: inclusive scan in place (rocPRIM through hipCUB; a library scan, like a
    // library GEMM, is not worth a hand-written kernel)
    size_t scan_bytes = 0;
    OGD_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, h->d_cell_start, h->d_cell_start, (int)(ncell + 1), nullptr));
    DevBuf dscan;
    OGD_TRY(dscan.alloc(scan_bytes));
    OGD_HIP(hipcub::DeviceScan::InclusiveSum(dscan.p, scan_bytes, h->d_cell_start, h->d_cell_start, (int)(ncell + 1), nullptr));

    OGD_HIP(hipMalloc((void**)&h->d_cell_tris, (size_t)std::max(total, 1) * sizeof(int)));
    OGD_HIP(hipMemcpyAsync(dcursor.p, h->d_cell_start, (size_t)ncell * sizeof(int), hipMemcpyDeviceToDevice, nullptr));
    hipLaunchKernelGGL(bucket_scatter_kernel<true>, tgrid, tblock, 0, nullptr, p_pts,
                       p_simp, h->d_rec, rec, nt, geom, dcursor.as<int>(), h->d_cell_tris,
                       (unsigned long long*)nullptr, (int*)nullptr, (int*)nullptr);
    OGD_TRY(check_launch("bucket_scatter_kernel<fill>"));
    if (nbig > 0) {
      hipLaunchKernelGGL(bucket_big_kernel<true>, bgrid, tblock, 0, nullptr, p_pts, p_simp,
                         h->d_rec, rec, geom, dcursor.as<int>(), h->d_cell_tris, (unsigned long long*)nullptr,
                         dbig.as<int>(), nbig);
      OGD_TRY(check_launch("bucket_big_kernel<fill>"));
    }
    int* d_max = reinterpret_cast<int*>(dscalars.as<char>() + 8);
    int* d_nlong = reinterpret_cast<int*>(dscalars.as<char>() + 16);
    // the fill cursors are spent: their buffer takes the list of long cells
    hipLaunchKernelGGL(bucket_sort_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, nullptr,
                       h->d_cell_start, (int)ncell, h->d_cell_tris, d_max, dcursor.as<int>(), d_nlong);
    OGD_TRY(check_launch("bucket_sort_kernel"));
    hipLaunchKernelGGL(bucket_sort_long_kernel, dim3(256), dim3(256), 0, nullptr, h->d_cell_start, h->d_cell_tris,
                       dcursor.as<int>(), d_nlong);
    OGD_TRY(check_launch("bucket_sort_long_kernel"));
    int max_cell = 0;
    OGD_HIP(hipMemcpy(&max_cell, d_max, sizeof max_cell, hipMemcpyDeviceToHost));  // also: the build is complete
    h->max_cell = std::max(max_cell, total > 0 ? 1 : 0);
    return OGD_OK;
  };
  const int st = build();
  if (st != OGD_OK) {
    ogd_tri_destroy(h);
    return st;
  }
  *out = h;
  return OGD_OK;
}

int ogd_tri_create(const double* pts, int n, const int32_t* simplices, int nt, const double* transform,
                   const double* vals, int nv, ogd_tri** out) {
  return tri_create(pts, n, simplices, nt, transform, false, vals, nv, out);
}

int ogd_tri_create_scipy(const double* pts, int n, const int32_t* simplices, int nt, const double* vals, int nv,
                         ogd_tri** out) {
  return tri_create(pts, n, simplices, nt, nullptr, true, vals, nv, out);
}

int ogd_tri_destroy(ogd_tri* t) {
  if (!t) return OGD_OK;
  {
    BindDevice bind(t->device);
    if (t->d_rec) (void)hipFree(t->d_rec);
    if (t->d_cell_start) (void)hipFree(t->d_cell_start);
    if (t->d_cell_tris) (void)hipFree(t->d_cell_tris);
  }
  delete t;
  return OGD_OK;
}

int ogd_tri_eval_dev(const ogd_tri* inner, const ogd_tri* extra, const double* bounds, const double* d_q,
                     int64_t nq, double* d_out, void* stream) {
  OGD_TRY(check_pair(inner, extra));
  OGD_REQUIRE(nq >= 0, "negative point count");
  if (nq == 0) return OGD_OK;
  OGD_REQUIRE(d_q && d_out, "null pointer");
  OGD_TRY(ensure_device());
  const int64_t nblk = (nq + 255) / 256;
  OGD_REQUIRE(nblk < INT_MAX, "too many points");
  const TriView ve = view_of(extra), vi = inner ? view_of(inner) : ve;
  const Bounds b = make_bounds(bounds);
  const double2* q = reinterpret_cast<const double2*>(d_q);
  if (extra->nv == 1)
    hipLaunchKernelGGL(tri_eval_kernel<1>, dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), vi,
                       inner != nullptr, ve, b, q, nq, d_out);
  else
    hipLaunchKernelGGL(tri_eval_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), vi,
                       inner != nullptr, ve, b, q, nq, d_out);
  return check_launch("tri_eval_kernel");
}

int ogd_tri_eval(const ogd_tri* inner, const ogd_tri* extra, const double* bounds, const double* q,
                 int64_t nq, double* out) {
  OGD_TRY(check_pair(inner, extra));
  OGD_REQUIRE(nq >= 0, "negative point count");
  if (nq == 0) return OGD_OK;
  OGD_REQUIRE(q && out, "null pointer");
  OGD_TRY(ensure_device());
  const size_t qb = (size_t)nq * 2 * sizeof(double), ob = (size_t)nq * extra->nv * sizeof(double);
  DevBuf dq, dout;
  OGD_TRY(dq.alloc(qb));
  OGD_TRY(dout.alloc(ob));
  OGD_HIP(hipMemcpy(dq.p, q, qb, hipMemcpyHostToDevice));
  OGD_TRY(ogd_tri_eval_dev(inner, extra, bounds, dq.as<double>(), nq, dout.as<double>(), nullptr));
  OGD_HIP(hipMemcpy(out, dout.p, ob, hipMemcpyDeviceToHost));
  return OGD_OK;
}

int ogd_tri_eval_ragged_dev(const ogd_tri* inner, const ogd_tri* extra, const double* bounds,
                            const double* d_q, const int64_t* d_offsets, int nrings, double* d_out,
                            void* stream) {
  OGD_TRY(check_pair(inner, extra));
  OGD_REQUIRE(nrings >= 0, "negative ring count");
  if (nrings == 0) return OGD_OK;
  OGD_REQUIRE(d_q && d_offsets && d_out, "null pointer");
  OGD_TRY(ensure_device());
  const TriView ve = view_of(extra), vi = inner ? view_of(inner) : ve;
  const Bounds b = make_bounds(bounds);
  const double2* q = reinterpret_cast<const double2*>(d_q);
  const unsigned nblk = (unsigned)((nrings + 3) / 4);
  if (extra->nv == 1)
    hipLaunchKernelGGL(tri_eval_ragged_kernel<1>, dim3(nblk), dim3(256), 0, as_stream(stream), vi,
                       inner != nullptr, ve, b, q, d_offsets, nrings, d_out);
  else
    hipLaunchKernelGGL(tri_eval_ragged_kernel<2>, dim3(nblk), dim3(256), 0, as_stream(stream), vi,
                       inner != nullptr, ve, b, q, d_offsets, nrings, d_out);
  return check_launch("tri_eval_ragged_kernel");
}

int ogd_tri_eval_ragged(const ogd_tri* inner, const ogd_tri* extra, const double* bounds, const double* q,
                        const int64_t* offsets, int nrings, double* out) {
  OGD_TRY(check_pair(inner, extra));
  OGD_REQUIRE(nrings >= 0, "negative ring count");
  if (nrings == 0) return OGD_OK;
  OGD_REQUIRE(q && offsets && out, "null pointer");
  OGD_REQUIRE(offsets[0] == 0, "offsets[0] must be 0");
  for (int r = 0; r < nrings; ++r)
    OGD_REQUIRE(offsets[r + 1] >= offsets[r], "offsets must be non-decreasing (ring %d)", r);
  const int64_t nq = offsets[nrings];
  if (nq == 0) return OGD_OK;
  OGD_TRY(ensure_device());
  // One wavefront per ring leaves the launch waiting for its longest ring (a 400-vertex ring takes seven
  // passes of a latency-bound walk): rings are cut into pieces of at most 64 vertices, one wavefront each
  // (measured: 12.0 -> 5.1 us for 400 rings, 88.6 -> 63.6 us for 5000).  Same points, same results.
  std::vector<int64_t> pieces;
  pieces.reserve((size_t)nrings + (size_t)(nq / 64) + 1);
  for (int r = 0; r < nrings; ++r)
    for (int64_t a = offsets[r]; a < offsets[r + 1]; a += 64) pieces.push_back(a);
  pieces.push_back(nq);
  const int64_t npieces = (int64_t)pieces.size() - 1;
  OGD_REQUIRE(npieces < INT_MAX, "too many points");
  const size_t qb = (size_t)nq * 2 * sizeof(double), ob = (size_t)nq * extra->nv * sizeof(double);
  const size_t fb = pieces.size() * sizeof(int64_t);
  DevBuf dq, dout, doff;
  OGD_TRY(dq.alloc(qb));
  OGD_TRY(dout.alloc(ob));
  OGD_TRY(doff.alloc(fb));
  OGD_HIP(hipMemcpy(dq.p, q, qb, hipMemcpyHostToDevice));
  OGD_HIP(hipMemcpy(doff.p, pieces.data(), fb, hipMemcpyHostToDevice));
  OGD_TRY(ogd_tri_eval_ragged_dev(inner, extra, bounds, dq.as<double>(), doff.as<int64_t>(), (int)npieces,
                                  dout.as<double>(), nullptr));
  OGD_HIP(hipMemcpy(out, dout.p, ob, hipMemcpyDeviceToHost));
  return OGD_OK;
}

int ogd_rings_self_cross(const double* q, const int64_t* offsets, int nrings, uint8_t* flags) {
  OGD_REQUIRE(nrings >= 0, "negative ring count");
  if (nrings == 0) return OGD_OK;
  OGD_REQUIRE(q && offsets && flags, "null pointer");
  OGD_REQUIRE(offsets[0] == 0, "offsets[0] must be 0");
  for (int r = 0; r < nrings; ++r)
    OGD_REQUIRE(offsets[r + 1] >= offsets[r], "offsets must be non-decreasing (ring %d)", r);
  const int64_t nq = offsets[nrings];
  OGD_TRY(ensure_device());
  DevBuf dq, doff, dfl;
  OGD_TRY(dq.alloc((size_t)nq * 2 * sizeof(double)));
  OGD_TRY(doff.alloc((size_t)(nrings + 1) * sizeof(int64_t)));
  OGD_TRY(dfl.alloc((size_t)nrings));
  if (nq) OGD_HIP(hipMemcpy(dq.p, q, (size_t)nq * 2 * sizeof(double), hipMemcpyHostToDevice));
  OGD_HIP(hipMemcpy(doff.p, offsets, (size_t)(nrings + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(rings_self_cross_kernel, dim3((unsigned)((nrings + 3) / 4)), dim3(256), 0, nullptr,
                     dq.as<double2>(), doff.as<int64_t>(), nrings, dfl.as<uint8_t>());
  OGD_TRY(check_launch("rings_self_cross_kernel"));
  OGD_HIP(hipMemcpy(flags, dfl.p, (size_t)nrings, hipMemcpyDeviceToHost));
  return OGD_OK;
}

int ogd_tri_rasterize_dev(const ogd_tri* inner, const ogd_tri* extra, const double* bounds, double x0,
                          double y0, double step, int nx, int ny, void* d_out, int64_t stride_x,
                          int64_t stride_y, int out_f64, void* stream) {
  OGD_TRY(check_pair(inner, extra));
  OGD_REQUIRE(nx >= 1 && ny >= 1, "empty lattice %dx%d", nx, ny);
  OGD_REQUIRE(std::isfinite(x0) && std::isfinite(y0) && std::isfinite(step), "lattice is not finite");
  OGD_REQUIRE(d_out != nullptr, "null pointer");
  OGD_TRY(ensure_device());
  dim3 block(RS_THREADS), grid((nx + RS_TW - 1) / RS_TW, (ny + RS_TH - 1) / RS_TH);
  OGD_REQUIRE(grid.y <= 65535u, "lattice too tall");
  const TriView ve = view_of(extra), vi = inner ? view_of(inner) : ve;
  const Bounds b = make_bounds(bounds);
  const int hi = inner != nullptr;
  hipStream_t st = as_stream(stream);
#define OGD_RASTER(NV, T)                                                                              \
  hipLaunchKernelGGL((tri_rasterize_kernel<NV, T>), grid, block, 0, st, vi, hi, ve, b, x0, y0, step, nx, \
                     ny, reinterpret_cast<T*>(d_out), stride_x, stride_y)
  if (extra->nv == 1) {
    if (out_f64) OGD_RASTER(1, double); else OGD_RASTER(1, float);
  } else {
    if (out_f64) OGD_RASTER(2, double); else OGD_RASTER(2, float);
  }
#undef OGD_RASTER
  return check_launch("tri_rasterize_kernel");
}

int ogd_tri_rasterize(const ogd_tri* inner, const ogd_tri* extra, const double* bounds, double x0, double y0,
                      double step, int nx, int ny, void* out, int64_t stride_x, int64_t stride_y,
                      int out_f64) {
  OGD_TRY(check_pair(inner, extra));
  OGD_REQUIRE(nx >= 1 && ny >= 1, "empty lattice %dx%d", nx, ny);
  OGD_REQUIRE(out != nullptr, "null pointer");
  const int nv = extra->nv;
  // host form: the output must be one dense block in either (x, y, k) or (y, x, k) order
  const bool xy = stride_y == nv && stride_x == (int64_t)ny * nv;
  const bool yx = stride_x == nv && stride_y == (int64_t)nx * nv;
  OGD_REQUIRE(xy || yx, "host output must be dense [nx][ny][nv] or [ny][nx][nv]");
  OGD_TRY(ensure_device());
  const size_t ob = (size_t)nx * ny * nv * (out_f64 ? sizeof(double) : sizeof(float));
  DevBuf dout;
  OGD_TRY(dout.alloc(ob));
  OGD_TRY(ogd_tri_rasterize_dev(inner, extra, bounds, x0, y0, step, nx, ny, dout.p, stride_x, stride_y,
                                out_f64, nullptr));
  OGD_HIP(hipMemcpy(out, dout.p, ob, hipMemcpyDeviceToHost));
  return OGD_OK;
}

}  // extern "C"
