// N3 — Delaunay triangulation of a warped lattice without a general-purpose hull code (host, exact).
//
// ref: origami/core/dewarp.py:143-161 (Transformer: LinearNDInterpolator over the h x w node positions of the
// dewarping grid) and origami/core/lingrid.py:192-250, 284-293 (ConvexHull + Delaunay of the same points).
// The reference hands the ~80 000 nodes of a broadsheet grid to Qhull twice per page (≈ 0.55 s of the
// stage's 1.2 s host time here).  The nodes are not scattered data: they are a regular lattice pushed
// around by a smooth warp, so a valid triangulation is known up front (every lattice quad split in two,
// plus the slivers between the lattice outline and its convex hull).  Lawson's edge flips turn any
// triangulation of a point set into THE Delaunay triangulation, and on this input only a few percent of the
// edges need one.  All predicates are exact (coordinates scaled to integers, 128 / 256-bit arithmetic behind a
// floating-point filter), so the result equals Qhull's wherever the Delaunay triangulation is unique
// (no four cocircular nodes); cocircular quads keep the diagonal they have.
//
// Inputs outside the exact domain (|coordinate| >= 2^20, or not a multiple of 2^-40: float32 node grids are
// far inside), folded lattices and degenerate outlines return OGD_ERR_UNSUPPORTED: the caller then takes the
// general route (Qhull), so this is an accelerator of the set-up, never a different answer.
#include "ogd_common.h"
#include "ogd_lawson.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <vector>

namespace ogd {
namespace {

typedef __int128 i128;
typedef unsigned __int128 u128;

struct I256 {  // sign-magnitude, magnitude as 4 x 64 bit little endian
  uint64_t m[4];
  bool neg;
};

I256 mul_i128(i128 a, i128 b) {
  I256 r;
  r.neg = (a < 0) != (b < 0);
  const u128 ua = a < 0 ? (u128)(-a) : (u128)a, ub = b < 0 ? (u128)(-b) : (u128)b;
  const uint64_t a0 = (uint64_t)ua, a1 = (uint64_t)(ua >> 64), b0 = (uint64_t)ub, b1 = (uint64_t)(ub >> 64);
  const u128 p00 = (u128)a0 * b0, p01 = (u128)a0 * b1, p10 = (u128)a1 * b0, p11 = (u128)a1 * b1;
  r.m[0] = (uint64_t)p00;
  u128 mid = (p00 >> 64) + (uint64_t)p01 + (uint64_t)p10;
  r.m[1] = (uint64_t)mid;
  u128 hi = (mid >> 64) + (p01 >> 64) + (p10 >> 64) + (uint64_t)p11;
  r.m[2] = (uint64_t)hi;
  r.m[3] = (uint64_t)((hi >> 64) + (p11 >> 64));
  if (!(r.m[0] | r.m[1] | r.m[2] | r.m[3])) r.neg = false;
  return r;
}

int cmp_mag(const I256& a, const I256& b) {
  for (int k = 3; k >= 0; --k)
    if (a.m[k] != b.m[k]) return a.m[k] < b.m[k] ? -1 : 1;
  return 0;
}

I256 add_i256(const I256& a, const I256& b) {
  I256 r;
  if (a.neg == b.neg) {
    u128 c = 0;
    for (int k = 0; k < 4; ++k) {
      c += (u128)a.m[k] + b.m[k];
      r.m[k] = (uint64_t)c;
      c >>= 64;
    }
    r.neg = a.neg;
  } else {
    const int c = cmp_mag(a, b);
    if (c == 0) {
      std::memset(r.m, 0, sizeof r.m);
      r.neg = false;
      return r;
    }
    const I256 &big = c > 0 ? a : b, &small = c > 0 ? b : a;
    uint64_t borrow = 0;
    for (int k = 0; k < 4; ++k) {
      const u128 d = (u128)big.m[k] - small.m[k] - borrow;
      r.m[k] = (uint64_t)d;
      borrow = (uint64_t)(d >> 64) ? 1 : 0;
    }
    r.neg = big.neg;
  }
  return r;
}

int sign_i256(const I256& a) {
  if (!(a.m[0] | a.m[1] | a.m[2] | a.m[3])) return 0;
  return a.neg ? -1 : 1;
}

struct Lattice {
  const double* p;            // [n][2]
  std::vector<int64_t> q;     // [n][2] coordinates * 2^40, exact
  int h, w;

  // sign of (b - a) x (c - a), exact
  int orient(int a, int b, int c) const {
    const double abx = p[2 * b] - p[2 * a], aby = p[2 * b + 1] - p[2 * a + 1];
    const double acx = p[2 * c] - p[2 * a], acy = p[2 * c + 1] - p[2 * a + 1];
    const double l = abx * acy, r = aby * acx, det = l - r;
    const double bound = 3.3306690738754716e-16 * (std::fabs(l) + std::fabs(r));  // (3 + 16 eps) eps
    if (det > bound) return 1;
    if (det < -bound) return -1;
    const i128 X1 = q[2 * b] - q[2 * a], Y1 = q[2 * b + 1] - q[2 * a + 1];
    const i128 X2 = q[2 * c] - q[2 * a], Y2 = q[2 * c + 1] - q[2 * a + 1];
    const i128 d = X1 * Y2 - Y1 * X2;
    return d > 0 ? 1 : d < 0 ? -1 : 0;
  }

  // > 0 iff d lies strictly inside the circle through a, b, c (a, b, c positively oriented), exact
  int incircle(int a, int b, int c, int d) const {
    const double adx = p[2 * a] - p[2 * d], ady = p[2 * a + 1] - p[2 * d + 1];
    const double bdx = p[2 * b] - p[2 * d], bdy = p[2 * b + 1] - p[2 * d + 1];
    const double cdx = p[2 * c] - p[2 * d], cdy = p[2 * c + 1] - p[2 * d + 1];
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
    const i128 AX = q[2 * a] - q[2 * d], AY = q[2 * a + 1] - q[2 * d + 1];
    const i128 BX = q[2 * b] - q[2 * d], BY = q[2 * b + 1] - q[2 * d + 1];
    const i128 CX = q[2 * c] - q[2 * d], CY = q[2 * c + 1] - q[2 * d + 1];
    const I256 t0 = mul_i128(AX * AX + AY * AY, BX * CY - CX * BY);
    const I256 t1 = mul_i128(BX * BX + BY * BY, CX * AY - AX * CY);
    const I256 t2 = mul_i128(CX * CX + CY * CY, AX * BY - BX * AY);
    return sign_i256(add_i256(add_i256(t0, t1), t2));
  }
};

// strictly inside or on the boundary of the positively oriented triangle (a, b, c)
bool in_closed_triangle(const Lattice& L, int a, int b, int c, int v) {
  return L.orient(a, b, v) >= 0 && L.orient(b, c, v) >= 0 && L.orient(c, a, v) >= 0;
}

// Ear clipping of a simple, positively oriented polygon; false if it gets stuck (degenerate outline).
bool triangulate_polygon(const Lattice& L, std::vector<int> poly, std::vector<int32_t>& tris) {
  while (poly.size() > 3) {
    const size_t n = poly.size();
    bool clipped = false;
    for (size_t i = 0; i < n && !clipped; ++i) {
      const int a = poly[(i + n - 1) % n], b = poly[i], c = poly[(i + 1) % n];
      if (L.orient(a, b, c) <= 0) continue;
      bool empty = true;
      for (size_t k = 0; k < n && empty; ++k) {
        const int v = poly[k];
        if (v == a || v == b || v == c) continue;
        if (in_closed_triangle(L, a, b, c, v)) empty = false;
      }
      if (!empty) continue;
      tris.push_back(a); tris.push_back(b); tris.push_back(c);
      poly.erase(poly.begin() + (ptrdiff_t)i);
      clipped = true;
    }
    if (!clipped) return false;
  }
  if (L.orient(poly[0], poly[1], poly[2]) <= 0) return false;
  tris.push_back(poly[0]); tris.push_back(poly[1]); tris.push_back(poly[2]);
  return true;
}

#define OGD_UNSUPPORTED(...) return ::ogd::fail(OGD_ERR_UNSUPPORTED, __VA_ARGS__)

int lattice_delaunay(const double* pts, int h, int w, std::vector<int32_t>& T, std::vector<int32_t>& hull) {
  const int n = h * w;
  Lattice L;
  L.p = pts;
  L.h = h;
  L.w = w;
  L.q.resize((size_t)2 * n);
  for (int i = 0; i < 2 * n; ++i) {
    const double v = pts[i];
    if (!(std::fabs(v) < 1048576.0)) OGD_UNSUPPORTED("coordinate %g outside the exact domain (|x| < 2^20)", v);
    const double s = v * 1099511627776.0;  // 2^40: exact, |s| < 2^60
    const int64_t k = (int64_t)s;
    if ((double)k != s) OGD_UNSUPPORTED("coordinate %.17g is not a multiple of 2^-40", v);
    L.q[i] = k;
  }
  auto id = [w](int i, int j) { return i * w + j; };

  // 1. the lattice quads, two positively oriented triangles each
  const int sgn = L.orient(id(0, 0), id(0, 1), id(1, 1));
  if (sgn == 0) OGD_UNSUPPORTED("degenerate first lattice cell");
  // split[cell]: 0 = diagonal a-c, triangles (a, b, c), (a, c, d); 1 = diagonal b-d, triangles (a, b, d), (b, c, d).
  // A convex cell gets the diagonal that is Delaunay for its own four nodes (a-c on a tie), which leaves the
  // flips below little to do on a smooth warp.  Cells are independent: rows of cells go to a few threads.
  const int cw0 = w - 1, ch0 = h - 1;
  std::vector<uint8_t> split((size_t)ch0 * cw0);
  T.assign((size_t)6 * ch0 * cw0, 0);
  T.reserve((size_t)6 * n);
  std::atomic<int> folded{-1};  // first cell (row-major index) found folded or degenerate
  parallel_ranges(ch0, 32, [&](int64_t i0, int64_t i1) {
    for (int i = (int)i0; i < (int)i1; ++i)
      for (int j = 0; j < cw0; ++j) {
        int a = id(i, j), b = id(i, j + 1), c = id(i + 1, j + 1), d = id(i + 1, j);
        if (sgn < 0) std::swap(b, d);  // mirrored lattice: (a, d, c, b) is the positive order
        const bool ac = L.orient(a, b, c) > 0 && L.orient(a, c, d) > 0;
        const bool bd = L.orient(a, b, d) > 0 && L.orient(b, c, d) > 0;
        const int cell = i * cw0 + j;
        if (!ac && !bd) {
          int seen = folded.load();
          while ((seen < 0 || cell < seen) && !folded.compare_exchange_weak(seen, cell)) {
          }
          continue;
        }
        const bool use_bd = ac && bd ? L.incircle(a, b, c, d) > 0 : bd;
        split[(size_t)cell] = use_bd;
        int32_t* t = &T[(size_t)6 * cell];
        if (use_bd) {
          t[0] = a; t[1] = b; t[2] = d; t[3] = b; t[4] = c; t[5] = d;
        } else {
          t[0] = a; t[1] = b; t[2] = c; t[3] = a; t[4] = c; t[5] = d;
        }
      }
  });
  if (folded.load() >= 0) OGD_UNSUPPORTED("lattice cell (%d, %d) is folded or degenerate", folded.load() / cw0, folded.load() % cw0);

  // 2. outline in positive order
  std::vector<int> ring;
  ring.reserve((size_t)2 * (h + w));
  for (int j = 0; j < w; ++j) ring.push_back(id(0, j));
  for (int i = 1; i < h; ++i) ring.push_back(id(i, w - 1));
  for (int j = w - 2; j >= 0; --j) ring.push_back(id(h - 1, j));
  for (int i = h - 2; i >= 1; --i) ring.push_back(id(i, 0));
  if (sgn < 0) std::reverse(ring.begin(), ring.end());
  const int m = (int)ring.size();

  // 3. strict convex hull of the outline (monotone chain, exact turns)
  std::vector<int> order(ring);
  std::sort(order.begin(), order.end(), [&](int a, int b) {
    return pts[2 * a] != pts[2 * b] ? pts[2 * a] < pts[2 * b] : pts[2 * a + 1] < pts[2 * b + 1];
  });
  for (int k = 1; k < m; ++k)
    if (pts[2 * order[k]] == pts[2 * order[k - 1]] && pts[2 * order[k] + 1] == pts[2 * order[k - 1] + 1])
      OGD_UNSUPPORTED("coincident outline nodes");
  std::vector<int> hv;
  for (int pass = 0; pass < 2; ++pass) {
    const size_t base = hv.size();
    for (int k = 0; k < m; ++k) {
      const int v = pass == 0 ? order[k] : order[m - 1 - k];
      while (hv.size() >= base + 2 && L.orient(hv[hv.size() - 2], hv[hv.size() - 1], v) <= 0) hv.pop_back();
      hv.push_back(v);
    }
    hv.pop_back();  // the last point of a chain starts the next one
  }
  if (hv.size() < 3) OGD_UNSUPPORTED("outline has no area");

  // hull vertices must appear along the outline in the same cyclic order (true for a simple outline)
  std::vector<int> pos_of((size_t)n, -1);
  for (int k = 0; k < m; ++k) pos_of[ring[k]] = k;
  const int nh = (int)hv.size();
  int start = 0;
  for (int k = 1; k < nh; ++k)
    if (pos_of[hv[k]] < pos_of[hv[start]]) start = k;
  std::vector<int> hpos((size_t)nh);
  for (int k = 0; k < nh; ++k) hpos[k] = pos_of[hv[(start + k) % nh]];
  for (int k = 1; k < nh; ++k)
    if (hpos[k] <= hpos[k - 1]) OGD_UNSUPPORTED("outline is not simple (hull order differs from outline order)");
  hull.resize((size_t)nh);
  for (int k = 0; k < nh; ++k) hull[k] = ring[hpos[k]];

  // 4. pockets between the outline and each hull edge; outline nodes exactly ON a hull edge split a pocket
  for (int k = 0; k < nh; ++k) {
    const int s = hpos[k], e = k + 1 < nh ? hpos[k + 1] : hpos[0] + m;
    if (e - s < 2) continue;
    const int A = ring[s % m], B = ring[e % m];
    int from = s;
    for (int t = s + 1; t <= e; ++t) {
      const int v = ring[t % m];
      int o = t == e ? 0 : L.orient(A, B, v);
      if (o < 0) OGD_UNSUPPORTED("outline node outside its hull edge");
      if (o != 0) continue;
      if (t - from >= 2) {  // pocket polygon: from -> t along the hull edge, back along the reversed outline
        std::vector<int> poly;
        poly.push_back(ring[from % m]);
        for (int u = t; u > from; --u) poly.push_back(ring[u % m]);
        if (!triangulate_polygon(L, poly, T)) OGD_UNSUPPORTED("degenerate pocket between outline and hull");
      }
      from = t;
    }
  }

  // 5. adjacency: N[3t + k] = triangle across the edge opposite vertex k of triangle t (-1: hull edge)
  const int nt = (int)(T.size() / 3);
  std::vector<int32_t> N((size_t)3 * nt, -1);
  {
    // Inside the lattice the neighbour across each cell side follows from the two cells' splits.  With the corners
    // named a, b, c, d in positive order, side ab faces the neighbour's side cd and side bc its side da; the slot
    // (triangle, opposite vertex) that holds a side: split 0: ab (t0, 2), bc (t0, 0), cd (t1, 0), da (t1, 1);
    // split 1: ab (t0, 2), bc (t1, 2), cd (t1, 0), da (t0, 1).  The diagonal joins t0 and t1.
    // Only the outline's sides and the triangles of the hull pockets are matched by sorting.
    static const int8_t side_tri[2][4] = {{0, 0, 1, 1}, {0, 1, 1, 0}}, side_slot[2][4] = {{2, 0, 0, 1}, {2, 2, 0, 1}};
    struct HalfEdge { uint64_t key; int32_t slot; };
    std::vector<HalfEdge> he;
    const int cw = w - 1, ch = h - 1;
    // neighbour cell across side ab, bc, cd, da as (di, dj); mirrored lattices have b below a instead of right of it
    const int di[2][4] = {{-1, 0, 1, 0}, {0, 1, 0, -1}}, dj[2][4] = {{0, 1, 0, -1}, {-1, 0, 1, 0}};
    const int mir = sgn < 0;
    auto cell_links = [&](int i, int j, std::vector<HalfEdge>* outline) {
      const int c = i * cw + j, sp = split[c];
      N[3 * (2 * c) + (sp ? 0 : 1)] = 2 * c + 1;
      N[3 * (2 * c + 1) + (sp ? 1 : 2)] = 2 * c;
      for (int side = 0; side < 4; ++side) {
        const int slot = 3 * (2 * c + side_tri[sp][side]) + side_slot[sp][side];
        const int ni = i + di[mir][side], nj = j + dj[mir][side];
        if (ni < 0 || nj < 0 || ni >= ch || nj >= cw) {  // outline: a pocket triangle or nothing
          if (outline) {
            const int t = slot / 3, k = slot % 3;
            const uint32_t x = (uint32_t)T[3 * t + (k + 1) % 3], y = (uint32_t)T[3 * t + (k + 2) % 3];
            outline->push_back({((uint64_t)std::min(x, y) << 32) | std::max(x, y), slot});
          }
          continue;
        }
        const int nc = ni * cw + nj, nsp = split[nc], nside = (side + 2) % 4;
        N[slot] = 2 * nc + side_tri[nsp][nside];
      }
    };
    // the cells on the rim here (their outline sides go through the sort below), the inner rows on a few threads
    for (int i = 0; i < ch; ++i)
      for (int j = 0; j < cw; ++j)
        if (i == 0 || j == 0 || i == ch - 1 || j == cw - 1) cell_links(i, j, &he);
    parallel_ranges(ch, 32, [&](int64_t i0, int64_t i1) {
      for (int i = std::max<int>((int)i0, 1); i < std::min<int>((int)i1, ch - 1); ++i)
        for (int j = 1; j < cw - 1; ++j) cell_links(i, j, nullptr);
    });
    for (int t = 2 * ch * cw; t < nt; ++t)
      for (int k = 0; k < 3; ++k) {
        const uint32_t x = (uint32_t)T[3 * t + (k + 1) % 3], y = (uint32_t)T[3 * t + (k + 2) % 3];
        he.push_back({((uint64_t)std::min(x, y) << 32) | std::max(x, y), 3 * t + k});
      }
    std::sort(he.begin(), he.end(), [](const HalfEdge& x, const HalfEdge& y) { return x.key < y.key; });
    for (size_t k = 0; k + 1 < he.size(); ++k)
      if (he[k].key == he[k + 1].key) {
        if (k + 2 < he.size() && he[k + 2].key == he[k].key) OGD_UNSUPPORTED("edge shared by three triangles");
        N[he[k].slot] = he[k + 1].slot / 3;
        N[he[k + 1].slot] = he[k].slot / 3;
        ++k;
      }
  }

  // 6. Lawson flips until every interior edge is locally Delaunay
  // The first look at every interior edge is independent of the others: done on a few threads, and only the
  // edges that fail go on the flip stack (a flip pushes the edges whose quadrilateral it changed).
  std::vector<uint8_t> bad((size_t)3 * nt, 0);
  parallel_ranges((int64_t)3 * nt, 32768, [&](int64_t s0, int64_t s1) {
    for (int64_t s = s0; s < s1; ++s) {
      const int t = (int)(s / 3), k = (int)(s % 3), u = N[s];
      if (u <= t) continue;  // hull edge, or the edge is looked at from its other triangle
      const int a = T[3 * t + k], b = T[3 * t + (k + 1) % 3], c = T[3 * t + (k + 2) % 3];
      int d = -1;
      for (int j = 0; j < 3; ++j)
        if (T[3 * u + j] != b && T[3 * u + j] != c) d = T[3 * u + j];
      bad[s] = d < 0 || L.incircle(a, b, c, d) > 0;
    }
  });
  std::vector<int32_t> stack;
  for (int s = 3 * nt - 1; s >= 0; --s)
    if (bad[s]) stack.push_back(s);
  int64_t flips = 0;
  const int64_t max_flips = (int64_t)nt * 64 + 1024;  // Lawson terminates; this only guards a logic error
  OGD_TRY(lawson_flips(L, T, N, stack, flips, max_flips, nullptr));
  return OGD_OK;
}

}  // namespace
}  // namespace ogd

using namespace ogd;

extern "C" {

int ogd_lattice_delaunay(const double* pts, int h, int w, int32_t* simplices, int64_t cap, int64_t* n_simplices,
                         int32_t* hull, int* n_hull, double* transform) {
  OGD_REQUIRE(pts && simplices && n_simplices && hull && n_hull, "null pointer");
  OGD_REQUIRE(h >= 2 && w >= 2 && (int64_t)h * w < (1 << 30), "lattice %dx%d not supported", h, w);
  std::vector<int32_t> T, H;
  OGD_TRY(lattice_delaunay(pts, h, w, T, H));
  const int64_t nt = (int64_t)(T.size() / 3);
  OGD_REQUIRE(nt <= cap, "simplices: room for %lld triangles, need %lld", (long long)cap, (long long)nt);
  std::memcpy(simplices, T.data(), T.size() * sizeof(int32_t));
  *n_simplices = nt;
  std::memcpy(hull, H.data(), H.size() * sizeof(int32_t));
  *n_hull = (int)H.size();
  if (transform) simplex_transforms(pts, T.data(), nt, transform);
  return OGD_OK;
}

}  // extern "C"
