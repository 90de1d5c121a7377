// N3b — Delaunay triangulation of scattered points on the host, exact, in place of a Qhull run.
//
// ref: origami/core/lingrid.py:286-290 (two scipy LinearNDInterpolator per interpolant: one over the samples,
// one over the extrapolation set of :192-250), reached from origami/core/dewarp.py:108-116 (the two angle
// fields of a page) and :143-161 (the Transformer).  A page costs the reference six Qhull runs; after
// ogd_lattice_delaunay took the Transformer's own lattice, five were left (two sample sets of 10^3..10^4
// points, three extrapolation sets of 10^2..10^3 points, most of them exactly collinear on the box outline)
// at ≈ 6 ms each — more host time than everything the GPU does for the page.
//
// Method: sort the points lexicographically and sweep: every new point lies outside the hull of the earlier
// ones, is joined to the hull edges it sees, and the touched edges are made locally Delaunay at once by
// Lawson flips (ogd_lawson.h).  Both predicates are exact for any finite doubles: a floating-point filter with
// a forward error bound, then the same polynomial evaluated without rounding error as a sum of non-overlapping
// doubles (error-free sums and products in the manner of Dekker / Knuth / Shewchuk).  With exact predicates
// the result IS the Delaunay triangulation, unique unless four points lie on one empty circle; the number of
// such cocircular interior edges is reported (n_ties) and the caller then asks Qhull, whose choice among equally
// valid diagonals cannot be predicted.  Of coincident points the one with the lowest index is triangulated and the
// others are reported as its twins (Qhull's "coplanar" points; which twin Qhull keeps cannot be predicted either,
// so the caller checks that twins carry equal values).  Fewer than three distinct points and all-collinear input
// return OGD_ERR_UNSUPPORTED (Qhull raises QhullError there, which the reference catches).
#include "ogd_common.h"
#include "ogd_lawson.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

#pragma clang fp contract(off)  // the error-free transformations below must not be fused or reassociated

namespace ogd {
namespace {

// ---- sums of non-overlapping doubles ("expansions"), smallest magnitude first ---------------------------------

inline void two_sum(double a, double b, double& s, double& e) {
  s = a + b;
  const double bb = s - a;
  e = (a - (s - bb)) + (b - bb);
}

inline void quick_two_sum(double a, double b, double& s, double& e) {  // |a| >= |b|
  s = a + b;
  e = b - (s - a);
}

inline void two_prod(double a, double b, double& p, double& e) {
  p = a * b;
  e = std::fma(a, b, -p);
}

// h = e + f exactly; zero terms are dropped (an all-zero sum is the single term 0).  h may not alias e or f.
int sum_expansions(int ne, const double* e, int nf, const double* f, double* h) {
  int i = 0, j = 0, nh = 0;
  auto next = [&]() -> double {  // merge by magnitude
    if (j >= nf || (i < ne && std::fabs(e[i]) <= std::fabs(f[j]))) return e[i++];
    return f[j++];
  };
  double q = next();
  if (i + j < ne + nf) {
    double g = next(), s, r;
    quick_two_sum(g, q, s, r);
    q = s;
    if (r != 0.0) h[nh++] = r;
    while (i + j < ne + nf) {
      g = next();
      two_sum(q, g, s, r);
      q = s;
      if (r != 0.0) h[nh++] = r;
    }
  }
  if (q != 0.0 || nh == 0) h[nh++] = q;
  return nh;
}

// h = e * b exactly; zero terms dropped.  h may not alias e.
int scale_expansion(int ne, const double* e, double b, double* h) {
  int nh = 0;
  double q, r;
  two_prod(e[0], b, q, r);
  if (r != 0.0) h[nh++] = r;
  for (int i = 1; i < ne; ++i) {
    double p1, p0, s;
    two_prod(e[i], b, p1, p0);
    two_sum(q, p0, s, r);
    if (r != 0.0) h[nh++] = r;
    quick_two_sum(p1, s, q, r);
    if (r != 0.0) h[nh++] = r;
  }
  if (q != 0.0 || nh == 0) h[nh++] = q;
  return nh;
}

inline int sign_of(int n, const double* e) {
  const double top = e[n - 1];
  return top > 0.0 ? 1 : top < 0.0 ? -1 : 0;
}

// px * qy - qx * py as an expansion of at most 4 terms
int cross_terms(const double* p, const double* q, double* h) {
  double a[2], b[2];
  two_prod(p[0], q[1], a[1], a[0]);
  two_prod(-q[0], p[1], b[1], b[0]);
  return sum_expansions(2, a, 2, b, h);
}

// twice the signed area of (p, q, r) = p x q + q x r + r x p, at most 12 terms
int area_terms(const double* p, const double* q, const double* r, double* h) {
  double a[4], b[4], c[4], ab[8];
  const int na = cross_terms(p, q, a), nb = cross_terms(q, r, b), nc = cross_terms(r, p, c);
  const int nab = sum_expansions(na, a, nb, b, ab);
  return sum_expansions(nab, ab, nc, c, h);
}

// sgn * (px^2 + py^2) * e, at most 8 * ne terms
int lifted_terms(int ne, const double* e, const double* p, double sgn, double* h) {
  double t1[24], t2[48], t3[48];
  int n1 = scale_expansion(ne, e, p[0], t1);
  const int n2 = scale_expansion(n1, t1, sgn * p[0], t2);
  n1 = scale_expansion(ne, e, p[1], t1);
  const int n3 = scale_expansion(n1, t1, sgn * p[1], t3);
  return sum_expansions(n2, t2, n3, t3, h);
}

struct Points {
  const double* p;  // [n][2]
  mutable int64_t exact_calls = 0;

  // sign of (b - a) x (c - a)
  int orient(int ia, int ib, int ic) const {
    const double* a = p + 2 * (size_t)ia, *b = p + 2 * (size_t)ib, *c = p + 2 * (size_t)ic;
    const double l = (b[0] - a[0]) * (c[1] - a[1]), r = (b[1] - a[1]) * (c[0] - a[0]), det = l - r;
    const double bound = 3.3306690738754716e-16 * (std::fabs(l) + std::fabs(r));  // (3 + 16 eps) eps
    if (det > bound) return 1;
    if (det < -bound) return -1;
    ++exact_calls;
    double e[12];
    return sign_of(area_terms(a, b, c, e), e);
  }

  // > 0 iff d lies strictly inside the circle through a, b, c (positively oriented); 0 iff cocircular
  int incircle(int ia, int ib, int ic, int id) const {
    const double* a = p + 2 * (size_t)ia, *b = p + 2 * (size_t)ib, *c = p + 2 * (size_t)ic, *d = p + 2 * (size_t)id;
    const double adx = a[0] - d[0], ady = a[1] - d[1];
    const double bdx = b[0] - d[0], bdy = b[1] - d[1];
    const double cdx = c[0] - d[0], cdy = c[1] - d[1];
    const double bdxcdy = bdx * cdy, cdxbdy = cdx * bdy, alift = adx * adx + ady * ady;
    const double cdxady = cdx * ady, adxcdy = adx * cdy, blift = bdx * bdx + bdy * bdy;
    const double adxbdy = adx * bdy, bdxady = bdx * ady, clift = cdx * cdx + cdy * cdy;
    const double det = alift * (bdxcdy - cdxbdy) + blift * (cdxady - adxcdy) + clift * (adxbdy - bdxady);
    const double permanent = (std::fabs(bdxcdy) + std::fabs(cdxbdy)) * alift +
                             (std::fabs(cdxady) + std::fabs(adxcdy)) * blift +
                             (std::fabs(adxbdy) + std::fabs(bdxady)) * clift;
    const double bound = 1.1102230246251577e-15 * permanent;  // (10 + 96 eps) eps
    if (det > bound) return 1;
    if (det < -bound) return -1;
    ++exact_calls;
    // | x y x^2+y^2 1 | over rows a, b, c, d, expanded along the lifted column
    double m[12], ta[96], tb[96], tc[96], td[96], s1[192], s2[192], s[384];
    const int na = lifted_terms(area_terms(b, c, d, m), m, a, 1.0, ta);
    const int nb = lifted_terms(area_terms(a, c, d, m), m, b, -1.0, tb);
    const int nc = lifted_terms(area_terms(a, b, d, m), m, c, 1.0, tc);
    const int nd = lifted_terms(area_terms(a, b, c, m), m, d, -1.0, td);
    const int n1 = sum_expansions(na, ta, nb, tb, s1), n2 = sum_expansions(nc, tc, nd, td, s2);
    return sign_of(sum_expansions(n1, s1, n2, s2, s), s);
  }
};

#define OGD_UNSUPPORTED(...) return ::ogd::fail(OGD_ERR_UNSUPPORTED, __VA_ARGS__)

int delaunay_points(const double* pts, int64_t n_all, std::vector<int32_t>& T, int64_t* n_ties, int32_t* twin) {
  for (int64_t i = 0; i < 2 * n_all; ++i) {
    if (!std::isfinite(pts[i])) return fail(OGD_ERR_INVALID, "delaunay_points: coordinate %lld is not finite", (long long)i);
    // the incircle polynomial has degree 4: its error-free products must neither overflow nor underflow
    const double m = std::fabs(pts[i]);
    if (m > 0x1p200 || (m != 0.0 && m < 0x1p-200)) OGD_UNSUPPORTED("delaunay_points: coordinate %g outside the exact domain", pts[i]);
  }
  Points L;
  L.p = pts;
  std::vector<int32_t> o((size_t)n_all);
  std::iota(o.begin(), o.end(), 0);
  std::sort(o.begin(), o.end(), [pts](int32_t a, int32_t b) {
    if (pts[2 * (size_t)a] != pts[2 * (size_t)b]) return pts[2 * (size_t)a] < pts[2 * (size_t)b];
    if (pts[2 * (size_t)a + 1] != pts[2 * (size_t)b + 1]) return pts[2 * (size_t)a + 1] < pts[2 * (size_t)b + 1];
    return a < b;  // coincident points: the lowest index first
  });
  // coincident points: the first of each group stays, the others are its twins (in no triangle)
  int64_t n = 0;
  for (int64_t i = 0; i < n_all; ++i) {
    const bool same = n > 0 && pts[2 * (size_t)o[i]] == pts[2 * (size_t)o[n - 1]] && pts[2 * (size_t)o[i] + 1] == pts[2 * (size_t)o[n - 1] + 1];
    if (same && !twin) OGD_UNSUPPORTED("delaunay_points: coincident points %d and %d", o[n - 1], o[i]);
    if (twin) twin[o[i]] = same ? o[n - 1] : -1;
    if (!same) o[n++] = o[i];
  }
  if (n < 3) OGD_UNSUPPORTED("delaunay_points: %lld distinct points", (long long)n);

  // the sorted points up to the first one off the line through o[0], o[1] form a fan around that point
  int64_t k = 2;
  int turn = 0;
  while (k < n && (turn = L.orient(o[0], o[1], o[k])) == 0) ++k;
  if (k == n) OGD_UNSUPPORTED("delaunay_points: all points are collinear");
  if (turn < 0) std::reverse(o.begin(), o.begin() + k);  // so that the chain, then o[k], is a left turn

  T.clear();
  T.reserve((size_t)6 * n);
  std::vector<int32_t> N;
  N.reserve((size_t)6 * n);
  std::vector<int32_t> nxt((size_t)n_all, -1), prv((size_t)n_all, -1), hslot((size_t)n_all, -1), stack;
  int64_t flips = 0;
  const int64_t max_flips = n * 256 + 4096;  // Lawson terminates; this only guards a logic error

  const int32_t q = o[k];
  for (int64_t i = 0; i + 1 < k; ++i) {  // triangle i = (chain i, chain i+1, q)
    const int32_t t = (int32_t)i;
    T.insert(T.end(), {o[i], o[i + 1], q});
    N.insert(N.end(), {i + 2 < k ? t + 1 : -1, t > 0 ? t - 1 : -1, -1});
    hslot[o[i]] = 3 * t + 2;
    nxt[o[i]] = o[i + 1];
    prv[o[i + 1]] = o[i];
    if (t > 0) stack.push_back(3 * t + 1);
  }
  hslot[o[k - 1]] = 3 * (int32_t)(k - 2) + 0;
  nxt[o[k - 1]] = q;
  prv[q] = o[k - 1];
  hslot[q] = 1;
  nxt[q] = o[0];
  prv[o[0]] = q;
  OGD_TRY(lawson_flips(L, T, N, stack, flips, max_flips, hslot.data()));

  int32_t last = q;  // the lexicographically largest point so far: always a hull vertex
  for (int64_t i = k + 1; i < n; ++i) {
    const int32_t p = o[i];
    int32_t lo = last, hi = last;
    while (L.orient(hi, nxt[hi], p) < 0) hi = nxt[hi];
    while (L.orient(prv[lo], lo, p) < 0) lo = prv[lo];
    if (lo == hi) return fail(OGD_ERR_INVALID, "delaunay_points: point %d sees no hull edge", p);
    int32_t prev_t = -1;
    const int32_t first_t = (int32_t)(T.size() / 3);
    for (int32_t v = lo; v != hi;) {
      const int32_t w = nxt[v], t = (int32_t)(T.size() / 3), inner = hslot[v];
      T.insert(T.end(), {v, p, w});  // edges: slot 0 = (p, w), slot 1 = (w, v) the old hull edge, slot 2 = (v, p)
      N.insert(N.end(), {-1, inner / 3, prev_t});
      N[inner] = t;
      if (prev_t >= 0) N[3 * prev_t + 0] = t;
      stack.push_back(3 * t + 1);
      prev_t = t;
      v = w;
    }
    hslot[lo] = 3 * first_t + 2;
    hslot[p] = 3 * prev_t + 0;
    nxt[lo] = p;
    prv[p] = lo;
    nxt[p] = hi;
    prv[hi] = p;
    last = p;
    OGD_TRY(lawson_flips(L, T, N, stack, flips, max_flips, hslot.data()));
  }

  if (n_ties) {
    int64_t ties = 0;
    const int32_t nt = (int32_t)(T.size() / 3);
    for (int32_t t = 0; t < nt; ++t)
      for (int j = 0; j < 3; ++j) {
        const int32_t u = N[3 * t + j];
        if (u <= t) continue;
        const int32_t b = T[3 * t + (j + 1) % 3], c = T[3 * t + (j + 2) % 3];
        int32_t d = -1;
        for (int m = 0; m < 3; ++m)
          if (T[3 * u + m] != b && T[3 * u + m] != c) d = T[3 * u + m];
        if (L.incircle(T[3 * t + j], b, c, d) == 0) ++ties;
      }
    *n_ties = ties;
  }
  return OGD_OK;
}

}  // namespace
}  // namespace ogd

using namespace ogd;

extern "C" {

int ogd_delaunay_points(const double* pts, int64_t n, int32_t* simplices, int64_t cap, int64_t* n_simplices,
                        double* transform, int64_t* n_ties, int32_t* twin) {
  OGD_REQUIRE(pts && simplices && n_simplices, "null pointer");
  OGD_REQUIRE(n >= 0 && n < (1 << 30), "%lld points not supported", (long long)n);
  std::vector<int32_t> T;
  OGD_TRY(delaunay_points(pts, n, T, n_ties, twin));
  const int64_t nt = (int64_t)(T.size() / 3);
  OGD_REQUIRE(nt <= cap, "simplices: room for %lld triangles, need %lld", (long long)cap, (long long)nt);
  std::memcpy(simplices, T.data(), T.size() * sizeof(int32_t));
  *n_simplices = nt;
  if (transform) simplex_transforms(pts, T.data(), nt, transform);
  return OGD_OK;
}

}  // extern "C"
