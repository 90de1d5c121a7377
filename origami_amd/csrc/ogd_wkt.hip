// WKT coordinate text of contour rings — host code, no device work.
// ref: origami/batch/detect/dewarp.py:39 (`geom.wkt`), origami/batch/core/io.py:565-577 (one .wkt member per
// contour).  The reference formats through shapely 1.7 -> GEOS 3.8 WKTWriter (C++): trim on, 16 significant
// digits of the stream's default float format, i.e. printf("%.16g"); "nan" / "inf" / "-inf" spelled as
// origami_amd/geometry.py:format_number does.  A broadsheet page has ~2e4 .. 3e5 contour vertices, so the
// per-number formatting is native here as it is in the reference.
#include "ogd_common.h"
#include <cmath>
#include <cstring>

namespace ogd {

static inline int put_number(char* p, double v) {
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

}  // extern "C"
