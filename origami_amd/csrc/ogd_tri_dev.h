// Device-side evaluation of the two-triangulation interpolant (shared by ogd_tri.hip: K3 / K1b, and
// ogd_march.hip: the node-grid march).  ref: origami/core/lingrid.py:174-189; arithmetic of
// scipy.interpolate.LinearNDInterpolator as described at the top of ogd_tri.hip.
#pragma once
#include "ogd_common.h"

struct ogd_tri {
  int device = -1;
  int nt = 0, nv = 0, rec = 0;  // rec = doubles per triangle record
  double* d_rec = nullptr;      // [nt][rec]
  int* d_cell_start = nullptr;  // [gnx*gny + 1]
  int* d_cell_tris = nullptr;   // [cell_start[gnx*gny]]
  int gnx = 1, gny = 1;
  double gx0 = 0, gy0 = 0, inv_cw = 0, inv_ch = 0;
  int max_cell = 0;  // longest cell list
};

namespace ogd {

constexpr double TRI_EPS = 100.0 * 2.220446049250313e-16;

struct TriView {
  const double* rec;
  const int* cell_start;
  const int* cell_tris;
  int nv, reclen, gnx, gny;
  double gx0, gy0, inv_cw, inv_ch;
};

static inline TriView view_of(const ogd_tri* t) {
  TriView v;
  v.rec = t->d_rec;
  v.cell_start = t->d_cell_start;
  v.cell_tris = t->d_cell_tris;
  v.nv = t->nv;
  v.reclen = t->rec;
  v.gnx = t->gnx;
  v.gny = t->gny;
  v.gx0 = t->gx0;
  v.gy0 = t->gy0;
  v.inv_cw = t->inv_cw;
  v.inv_ch = t->inv_ch;
  return v;
}

struct Bounds {
  double minx, miny, maxx, maxy;
  int clip;
};

__device__ __forceinline__ int cell_of(const TriView& t, double x, double y) {
  // NaN coordinates land in cell 0 and match no triangle
  int cx = (int)floor((x - t.gx0) * t.inv_cw);
  int cy = (int)floor((y - t.gy0) * t.inv_ch);
  cx = min(max(cx, 0), t.gnx - 1);
  cy = min(max(cy, 0), t.gny - 1);
  return cy * t.gnx + cx;
}

// Barycentric test of one record; on success writes nv values.
template <int NV>
__device__ __forceinline__ bool tri_try(const double* __restrict__ r, double x, double y, double* out) {
  const double dx = x - r[4], dy = y - r[5];
  const double c0 = __dadd_rn(__dmul_rn(r[0], dx), __dmul_rn(r[1], dy));
  const double c1 = __dadd_rn(__dmul_rn(r[2], dx), __dmul_rn(r[3], dy));
  const double c2 = (1.0 - c0) - c1;
  const double lo = -TRI_EPS, hi = 1.0 + TRI_EPS;
  if (!(c0 >= lo && c0 <= hi && c1 >= lo && c1 <= hi && c2 >= lo && c2 <= hi)) return false;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double v = __dmul_rn(c0, r[6 + k]);
    v = __dadd_rn(v, __dmul_rn(c1, r[6 + NV + k]));
    v = __dadd_rn(v, __dmul_rn(c2, r[6 + 2 * NV + k]));
    out[k] = v;
  }
  return true;
}

template <int NV>
__device__ __forceinline__ bool tri_locate(const TriView& t, double x, double y, double* out) {
  const int c = cell_of(t, x, y);
  const int b = t.cell_start[c], e = t.cell_start[c + 1];
  for (int k = b; k < e; ++k) {
    const double* r = t.rec + (size_t)t.cell_tris[k] * (6 + 3 * NV);
    if (tri_try<NV>(r, x, y, out)) return true;
  }
  return false;
}

// tri_locate for a walk: `hint` is the triangle that held the walk's previous point (-1: none).  It is tried first —
// its record is one cached 72-byte read instead of three dependent ones (cell, list, record), and a step of one
// grid cell stays inside the same triangle three times out of four.  Strictly inside a triangle the answer does
// not depend on how the triangle was found; on a shared edge (within eps) the neighbour may answer, which agrees
// to the last bits (the interpolant is continuous).
template <int NV>
__device__ __forceinline__ bool tri_locate_hint(const TriView& t, double x, double y, double* out, int& hint) {
  if (hint >= 0 && tri_try<NV>(t.rec + (size_t)hint * (6 + 3 * NV), x, y, out)) return true;
  const int c = cell_of(t, x, y);
  const int b = t.cell_start[c], e = t.cell_start[c + 1];
  for (int k = b; k < e; ++k) {
    const int tri = t.cell_tris[k];
    if (tri_try<NV>(t.rec + (size_t)tri * (6 + 3 * NV), x, y, out)) {
      hint = tri;
      return true;
    }
  }
  return false;
}

__device__ __forceinline__ double clipd(double v, double lo, double hi) {
  // numpy.clip = minimum(maximum(v, lo), hi); NaN propagates
  if (v != v) return v;
  return fmin(fmax(v, lo), hi);
}

template <int NV>
__device__ __forceinline__ void eval_point(const TriView& inner, int has_inner, const TriView& extra,
                                           const Bounds& b, double x, double y, double* out) {
  if (b.clip) {
    x = clipd(x, b.minx, b.maxx);
    y = clipd(y, b.miny, b.maxy);
  }
  if (has_inner && tri_locate<NV>(inner, x, y, out)) return;
  if (tri_locate<NV>(extra, x, y, out)) return;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
#pragma unroll
  for (int k = 0; k < NV; ++k) out[k] = nan;
}

// eval_point along a walk; hint.x / hint.y: see tri_locate_hint (inner / extra triangulation).
template <int NV>
__device__ __forceinline__ void eval_point_hint(const TriView& inner, int has_inner, const TriView& extra,
                                                const Bounds& b, double x, double y, double* out, int2& hint) {
  if (b.clip) {
    x = clipd(x, b.minx, b.maxx);
    y = clipd(y, b.miny, b.maxy);
  }
  if (has_inner && tri_locate_hint<NV>(inner, x, y, out, hint.x)) return;
  if (tri_locate_hint<NV>(extra, x, y, out, hint.y)) return;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
#pragma unroll
  for (int k = 0; k < NV; ++k) out[k] = nan;
}

static inline Bounds make_bounds(const double* bounds) {
  Bounds b;
  b.clip = bounds != nullptr;
  b.minx = bounds ? bounds[0] : 0;
  b.miny = bounds ? bounds[1] : 0;
  b.maxx = bounds ? bounds[2] : 0;
  b.maxy = bounds ? bounds[3] : 0;
  return b;
}

}  // namespace ogd
