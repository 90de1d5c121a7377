// WKT coordinate text of contour rings, written and read, and ring areas — host code, no device work.
// ref: origami/batch/detect/dewarp.py:39 (`geom.wkt`), origami/batch/core/io.py:565-577 (one .wkt member per
// contour).  The reference formats through shapely 1.7 -> GEOS 3.8 WKTWriter (C++): trim on, 16 significant
// digits of the stream's default float format, i.e. printf("%.16g"); "nan" / "inf" / "-inf" spelled as
// origami_amd/geometry.py:format_number does.  A broadsheet page has ~2e4 .. 3e5 contour vertices, so the
// per-number formatting is native here as it is in the reference.
#include "ogd_common.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace ogd {

// printf("%.16g") for the magnitudes page coordinates have, 1e-4 <= |v| < 1e15, without printf: the decimal point
// is placed by comparing against exact powers of ten, the 16 digits are round-half-even(m * 10^p / 2^s) in 128-bit
// integers (v = m * 2^-s exactly, p <= 20), which is the exact decimal expansion of the double correctly rounded —
// what glibc prints.  Returns 0 for anything else (the caller then asks snprintf).
static inline int put_fixed16(char* out, double v) {
  static const double P10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
  static const double N10[5] = {1e0, 1e-1, 1e-2, 1e-3, 1e-4};  // only 1e0 is exact: used as a first guess, checked below
  const double a = std::fabs(v);
  if (!(a >= 1e-4 && a < 1e15)) return 0;
  int k;  // decimal digits in front of the point (<= 0: zeros behind it)
  if (a >= 1.0) {
    k = 1;
    while (k < 16 && a >= P10[k]) ++k;
  } else {
    k = 0;
    while (k > -4 && a < N10[1 - k]) --k;  // a guess within one of the truth
  }
  uint64_t bits;
  std::memcpy(&bits, &a, sizeof bits);
  const int be = (int)(bits >> 52);
  if (be == 0) return 0;
  const uint64_t m = (bits & ((1ull << 52) - 1)) | (1ull << 52);
  const int s = 1075 - be;  // a = m * 2^-s; a < 2^50 -> s >= 3; a >= 1e-4 -> s <= 66
  typedef unsigned __int128 u128;
  uint64_t digits = 0;
  for (int attempt = 0; attempt < 3; ++attempt) {
    const int p = 16 - k;  // 1 .. 20
    u128 t = m;
    for (int i = 0; i < p; ++i) t *= 10;  // < 2^53 * 10^20 < 2^120
    const u128 q = t >> s, rem = t - (q << s), half = (u128)1 << (s - 1);
    uint64_t n = (uint64_t)q;
    if (rem > half || (rem == half && (n & 1))) ++n;
    if (n >= 10000000000000000ull) {  // 17 digits: the point sits one place further right
      ++k;
      continue;
    }
    if (n < 1000000000000000ull) {  // 15 digits: one place further left
      --k;
      continue;
    }
    digits = n;
    break;
  }
  if (!digits || k > 15 || k < -3) return 0;  // |v| rounds to 1e15 (scientific in %g) or the guesses failed
  char d[16];
  for (int i = 15; i >= 0; --i) {
    d[i] = (char)('0' + digits % 10);
    digits /= 10;
  }
  int last = 15;
  while (last > 0 && d[last] == '0') --last;  // %g drops trailing zeros
  char* o = out;
  if (std::signbit(v)) *o++ = '-';
  if (k > 0) {
    for (int i = 0; i < k; ++i) *o++ = d[i];  // digits past `last` in front of the point are zeros that stay
    if (last >= k) {
      *o++ = '.';
      for (int i = k; i <= last; ++i) *o++ = d[i];
    }
  } else {
    *o++ = '0';
    *o++ = '.';
    for (int i = 0; i < -k; ++i) *o++ = '0';
    for (int i = 0; i <= last; ++i) *o++ = d[i];
  }
  return (int)(o - out);
}

static inline int put_number(char* p, double v) {
  if (const int n = put_fixed16(p, v)) return n;
  if (std::isnan(v)) {
    std::memcpy(p, "nan", 3);
    return 3;
  }
  if (std::isinf(v)) {
    if (v > 0) {
      std::memcpy(p, "inf", 3);
      return 3;
    }
    std::memcpy(p, "-inf", 4);
    return 4;
  }
  return std::snprintf(p, 32, "%.16g", v);
}

// ---- reading -----------------------------------------------------------------------------------------------------
// ref: origami/batch/core/io.py:142-144 (`shapely.wkt.loads` on every .wkt member: GEOS' C++ WKTReader).

struct WktCursor {
  const char* p;
  const char* end;
  void blanks() {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
  }
  bool take(char c) {
    blanks();
    if (p < end && *p == c) {
      ++p;
      return true;
    }
    return false;
  }
  bool word(const char* w) {  // case-insensitive keyword
    blanks();
    const char* q = p;
    for (; *w; ++w, ++q)
      if (q >= end || (*q | 0x20) != *w) return false;
    if (q < end && ((*q | 0x20) >= 'a' && (*q | 0x20) <= 'z')) return false;
    p = q;
    return true;
  }
};

struct WktOut {
  double* pts;
  int64_t cap_pts, n_pts = 0;
  int64_t* ring_offsets;
  int32_t* ring_geom;
  int32_t* ring_part;
  int64_t cap_rings, n_rings = 0;
};

// "(x y [z [m]], ...)" or "()"; false on anything else
bool parse_ring(WktCursor& c, WktOut& o, int geom, int part) {
  if (!c.take('(')) return false;
  if (o.n_rings >= o.cap_rings) return false;
  o.ring_geom[o.n_rings] = geom;
  o.ring_part[o.n_rings] = part;
  if (!c.take(')')) {
    do {
      double v[4];
      int k = 0;
      for (; k < 4; ++k) {
        c.blanks();
        if (c.p >= c.end || *c.p == ',' || *c.p == ')') break;
        char* stop = nullptr;
        v[k] = std::strtod(c.p, &stop);  // the buffer ends in NUL (ogd_wkt_parse requires it)
        if (stop == c.p || stop > c.end) return false;
        c.p = stop;
      }
      if (k < 2 || o.n_pts >= o.cap_pts) return false;
      o.pts[2 * o.n_pts] = v[0];
      o.pts[2 * o.n_pts + 1] = v[1];
      ++o.n_pts;
    } while (c.take(','));
    if (!c.take(')')) return false;
  }
  o.ring_offsets[++o.n_rings] = o.n_pts;
  return true;
}

bool parse_polygon_body(WktCursor& c, WktOut& o, int geom, int part) {
  if (!c.take('(')) return false;
  do {
    if (!parse_ring(c, o, geom, part)) return false;
  } while (c.take(','));
  return c.take(')');
}

}  // namespace ogd

using namespace ogd;

extern "C" {

int ogd_wkt_ring_text(const double* pts, const int64_t* offsets, int nrings, char* out, int64_t cap,
                      int64_t* text_offsets) {
  OGD_REQUIRE(nrings >= 0, "negative ring count");
  OGD_REQUIRE(text_offsets != nullptr, "null pointer");
  text_offsets[0] = 0;
  if (nrings == 0) return OGD_OK;
  OGD_REQUIRE(pts && offsets && out, "null pointer");
  OGD_REQUIRE(offsets[0] == 0, "offsets[0] must be 0");
  int64_t pos = 0;
  for (int r = 0; r < nrings; ++r) {
    const int64_t b = offsets[r], e = offsets[r + 1];
    OGD_REQUIRE(e >= b, "offsets must be non-decreasing (ring %d)", r);
    // "(" + per point at most 2 * 24 characters, a blank and ", " + ")"
    OGD_REQUIRE(pos + (e - b) * 52 + 2 <= cap, "text buffer too small (%lld bytes)", (long long)cap);
    out[pos++] = '(';
    for (int64_t i = b; i < e; ++i) {
      if (i > b) {
        out[pos++] = ',';
        out[pos++] = ' ';
      }
      pos += put_number(out + pos, pts[2 * i]);
      out[pos++] = ' ';
      pos += put_number(out + pos, pts[2 * i + 1]);
    }
    out[pos++] = ')';
    text_offsets[r + 1] = pos;
  }
  return OGD_OK;
}

int ogd_wkt_parse(const char* text, const int64_t* text_offsets, int n, int8_t* kinds, double* pts, int64_t cap_pts,
                  int64_t* ring_offsets, int32_t* ring_geom, int32_t* ring_part, int64_t cap_rings,
                  int64_t* n_rings) {
  OGD_REQUIRE(n >= 0, "negative text count");
  OGD_REQUIRE(text_offsets && ring_offsets && n_rings, "null pointer");
  ring_offsets[0] = 0;
  *n_rings = 0;
  if (n == 0) return OGD_OK;
  OGD_REQUIRE(text && kinds && pts && ring_geom && ring_part, "null pointer");
  OGD_REQUIRE(text[text_offsets[n]] == 0, "the text buffer must end in a NUL byte");
  WktOut o;
  o.pts = pts;
  o.cap_pts = cap_pts;
  o.ring_offsets = ring_offsets;
  o.ring_geom = ring_geom;
  o.ring_part = ring_part;
  o.cap_rings = cap_rings;
  for (int g = 0; g < n; ++g) {
    OGD_REQUIRE(text_offsets[g + 1] >= text_offsets[g], "text offsets must be non-decreasing");
    WktCursor c{text + text_offsets[g], text + text_offsets[g + 1]};
    const int kind = c.word("multipolygon") ? 2 : c.word("polygon") ? 0 : c.word("linestring") ? 1 : -1;
    bool ok = kind >= 0;
    if (ok) {
      kinds[g] = (int8_t)kind;
      c.word("z");
      if (!c.word("empty")) {
        if (kind == 1) {
          ok = parse_ring(c, o, g, 0);
        } else if (kind == 0) {
          ok = parse_polygon_body(c, o, g, 0);
        } else {
          ok = c.take('(');
          for (int part = 0; ok; ++part) {
            ok = parse_polygon_body(c, o, g, part);
            if (!ok || !c.take(',')) break;
          }
          ok = ok && c.take(')');
        }
      }
      c.blanks();
      ok = ok && c.p == c.end;
    }
    if (!ok) return fail(OGD_ERR_UNSUPPORTED, "WKT text %d not understood (or buffers too small): %.40s", g, text + text_offsets[g]);
  }
  *n_rings = o.n_rings;
  return OGD_OK;
}

int ogd_rings_area_signed(const double* pts, const int64_t* offsets, int nrings, double* out) {
  OGD_REQUIRE(nrings >= 0, "negative ring count");
  if (nrings == 0) return OGD_OK;
  OGD_REQUIRE(pts && offsets && out, "null pointer");
  for (int r = 0; r < nrings; ++r) {
    const int64_t b = offsets[r], n = offsets[r + 1] - b;
    OGD_REQUIRE(n >= 0, "offsets must be non-decreasing (ring %d)", r);
    double total = 0.0;
    if (n >= 3) {
      const double* q = pts + 2 * b;
      const double x0 = q[0];
      for (int64_t i = 1; i < n - 1; ++i) total += (q[2 * i] - x0) * (q[2 * (i - 1) + 1] - q[2 * (i + 1) + 1]);
    }
    out[r] = total / 2.0;
  }
  return OGD_OK;
}

}  // extern "C"
