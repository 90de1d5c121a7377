// Shared by the two host triangulators (ogd_lattice.hip: warped lattices, ogd_delaunay.hip: scattered points):
// Lawson's edge flips over a triangle soup with adjacency, and scipy's barycentric transforms.
//
// Triangles are positively oriented vertex triples T[3t .. 3t+2]; N[3t + k] is the triangle across the edge
// opposite vertex k of triangle t (-1: hull edge).  A predicate provider P offers
//   int incircle(int a, int b, int c, int d)   > 0 iff d lies strictly inside the circle through the positively
//                                              oriented a, b, c; 0 iff the four are cocircular; exact.
#pragma once
#include "ogd_common.h"
#include <algorithm>
#include <cmath>
#include <system_error>
#include <thread>
#include <cstdlib>
#include <vector>

namespace ogd {

// fn(begin, end) over [0, n) on up to four threads (the calling one included); the per-page set-up of the batch stage
// is host-bound and these loops are independent per element.  Small n runs inline.
template <class F>
void parallel_ranges(int64_t n, int64_t grain, F fn) {
  // OGD_THREADS (1 .. 4, read once): the stage runtime lowers it for workers that share few cores
  static const int64_t cap = [] {
    const char* e = getenv("OGD_THREADS");
    const int v = e ? atoi(e) : 4;
    return (int64_t)std::min(std::max(v, 1), 4);
  }();
  const unsigned hw = std::thread::hardware_concurrency();
  const int64_t want = std::min<int64_t>(std::min<int64_t>(cap, hw ? hw : 1), n / std::max<int64_t>(grain, 1));
  if (want <= 1) {
    fn((int64_t)0, n);
    return;
  }
  std::vector<std::thread> pool;
  const int64_t step = (n + want - 1) / want;
  int64_t started = 1;  // ranges 1 .. started - 1 run on threads of their own
  try {
    for (; started < want; ++started) pool.emplace_back(fn, started * step, std::min(n, (started + 1) * step));
  } catch (const std::system_error&) {  // no more threads to be had: the rest runs here
  }
  fn((int64_t)0, std::min(n, step));
  for (int64_t k = started; k < want; ++k) fn(k * step, std::min(n, (k + 1) * step));
  for (auto& t : pool) t.join();
}

// Flips until every edge whose slot (3t + k) is on `stack` (and every edge a flip touches) is locally Delaunay.
// Cocircular quads keep the diagonal they have.  hull_slot (may be null): per vertex v the slot of the hull edge
// v -> next(v); kept current when a flip moves a hull edge to another slot.
template <class P>
int lawson_flips(const P& L, std::vector<int32_t>& T, std::vector<int32_t>& N, std::vector<int32_t>& stack,
                 int64_t& flips, int64_t max_flips, int32_t* hull_slot) {
  while (!stack.empty()) {
    const int s = stack.back();
    stack.pop_back();
    const int t = s / 3, k = s % 3, u = N[s];
    if (u < 0) continue;
    const int a = T[3 * t + k], b = T[3 * t + (k + 1) % 3], c = T[3 * t + (k + 2) % 3];
    int ku = -1;  // vertex of u opposite the shared edge (b, c)
    for (int j = 0; j < 3; ++j)
      if (T[3 * u + j] != b && T[3 * u + j] != c) ku = j;
    if (ku < 0 || N[3 * u + ku] != t) return fail(OGD_ERR_INVALID, "lawson_flips: adjacency corrupt");
    const int d = T[3 * u + ku];
    if (L.incircle(a, b, c, d) <= 0) continue;
    if (++flips > max_flips) return fail(OGD_ERR_INVALID, "lawson_flips: flip limit exceeded");
    // quad a, b, d, c (positive); diagonal b-c becomes a-d:  t = (a, b, d), u = (a, d, c)
    const int n_ab = N[3 * t + (k + 2) % 3];  // across (a, b)
    const int n_ca = N[3 * t + (k + 1) % 3];  // across (c, a)
    int n_bd = -1, n_dc = -1;                 // u = (d, c, b) in some rotation: across (b, d) and (d, c)
    for (int j = 0; j < 3; ++j) {
      if (T[3 * u + j] == c) n_bd = N[3 * u + j];
      if (T[3 * u + j] == b) n_dc = N[3 * u + j];
    }
    T[3 * t + 0] = a; T[3 * t + 1] = b; T[3 * t + 2] = d;
    N[3 * t + 0] = n_bd; N[3 * t + 1] = u; N[3 * t + 2] = n_ab;
    T[3 * u + 0] = a; T[3 * u + 1] = d; T[3 * u + 2] = c;
    N[3 * u + 0] = n_dc; N[3 * u + 1] = n_ca; N[3 * u + 2] = t;
    auto relink = [&](int nb, int old_t, int new_t) {
      if (nb < 0) return;
      for (int j = 0; j < 3; ++j)
        if (N[3 * nb + j] == old_t) {
          // the neighbour may touch old_t on one edge only, but after earlier flips it can also border new_t
          const int x = T[3 * nb + (j + 1) % 3], y = T[3 * nb + (j + 2) % 3];
          const int* tv = &T[3 * new_t];
          const bool hx = tv[0] == x || tv[1] == x || tv[2] == x, hy = tv[0] == y || tv[1] == y || tv[2] == y;
          if (hx && hy) N[3 * nb + j] = new_t;
        }
    };
    relink(n_bd, u, t);
    relink(n_ca, t, u);
    if (hull_slot) {  // hull edges run with the triangles' orientation: b -> d, a -> b, d -> c, c -> a
      if (n_bd < 0) hull_slot[b] = 3 * t + 0;
      if (n_ab < 0) hull_slot[a] = 3 * t + 2;
      if (n_dc < 0) hull_slot[d] = 3 * u + 0;
      if (n_ca < 0) hull_slot[c] = 3 * u + 1;
    }
    stack.push_back(3 * t + 0); stack.push_back(3 * t + 2);
    stack.push_back(3 * u + 0); stack.push_back(3 * u + 1);
  }
  return OGD_OK;
}

// scipy.spatial.Delaunay.transform of ONE triangle: o[0..3] = inverse of [[x0 - x2, x1 - x2], [y0 - y2, y1 - y2]],
// o[4..5] = vertex 2; LU with partial pivoting as LAPACK's dgetrf / dgetrs order the operations; NaN where the
// reciprocal condition number (1-norm) is below 1000 eps (scipy's rcond_limit).  Host and device run this very
// source (no contraction on either side, IEEE division on both): ogd_tri_create_scipy computes on the GPU what
// simplex_transforms computes here, bit for bit.
__host__ __device__ inline void simplex_transform_one(const double* p0, const double* p1, const double* p2, double* o) {
  const double rcond_limit = 1000.0 * 2.220446049250313e-16;
  const double nan = __builtin_nan("");
  const double a00 = p0[0] - p2[0], a01 = p1[0] - p2[0], a10 = p0[1] - p2[1], a11 = p1[1] - p2[1];
  o[4] = p2[0];
  o[5] = p2[1];
  // scipy passes its row-major T to column-major LAPACK: the factored matrix is M = T^T and the solution
  // of M X = I, read back row-major, is X^T = T^-1
  const double m00 = a00, m01 = a10, m10 = a01, m11 = a11;
  const bool swap = fabs(m10) > fabs(m00);
  const double u00 = swap ? m10 : m00, u01 = swap ? m11 : m01, l10n = swap ? m00 : m10, r1 = swap ? m01 : m11;
  bool ok = u00 != 0.0;
  double i00 = nan, i01 = nan, i10 = nan, i11 = nan;
  if (ok) {
    const double l10 = l10n * (1.0 / u00);
    const double u11 = r1 - l10 * u01;
    ok = u11 != 0.0;
    if (ok) {
      const double b00 = swap ? 0.0 : 1.0, b10 = swap ? 1.0 : 0.0;  // column 0 of P I (P: the row swap)
      const double b01 = swap ? 1.0 : 0.0, b11 = swap ? 0.0 : 1.0;  // column 1
      const double y10 = b10 - l10 * b00, y11 = b11 - l10 * b01;
      const double x10 = y10 / u11, x11 = y11 / u11;
      const double x00 = (b00 - u01 * x10) / u00, x01 = (b01 - u01 * x11) / u00;
      i00 = x00; i01 = x10; i10 = x01; i11 = x11;
      const double an0 = fabs(m00) + fabs(m10), an1 = fabs(m01) + fabs(m11);
      const double in0 = fabs(x00) + fabs(x10), in1 = fabs(x01) + fabs(x11);
      const double anorm = an0 > an1 ? an0 : an1, inorm = in0 > in1 ? in0 : in1;
      const double rcond = 1.0 / (anorm * inorm);
      if (!(rcond >= rcond_limit)) ok = false;
    }
  }
  o[0] = ok ? i00 : nan; o[1] = ok ? i01 : nan; o[2] = ok ? i10 : nan; o[3] = ok ? i11 : nan;
}

// scipy.spatial.Delaunay.transform for nt triangles; transform: [nt][3][2].
inline void simplex_transforms(const double* pts, const int32_t* T, int64_t nt, double* transform) {
  parallel_ranges(nt, 16384, [=](int64_t t_begin, int64_t t_end) {
    for (int64_t t = t_begin; t < t_end; ++t)
      simplex_transform_one(pts + 2 * (size_t)T[3 * t], pts + 2 * (size_t)T[3 * t + 1], pts + 2 * (size_t)T[3 * t + 2],
                            transform + 6 * t);
  });
}

}  // namespace ogd
