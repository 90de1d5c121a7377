// N1 — per-line images of a dewarped page, straight from the warped page.
// ref: origami/core/block.py:223-259 (Line.warped_grid + Line.dewarped_image): for every text line the
// reference builds a small dewarped lattice (np.linspace along `up` and `right` from p0), maps it to the
// warped page with Grid.inverse_yx (scipy RegularGridInterpolator over the dense grid, ref:
// origami/core/dewarp.py:561-576), casts to float32, cv2.remap()s the page through it and flips the rows.
// Thousands of 48 x L remaps per page on 8 threads there; here ONE launch over all lines of a page, with
// the page resident on the device (ogd_page) and neither the dense grid nor the line lattices ever
// materialised.  Arithmetic: ogd_resize_math.h (dense grid, fp64 inverse), ogd_remap_math.h (cv2.remap).
#include "ogd_common.h"
#include "ogd_remap_math.h"
#include "ogd_resize_math.h"
#include <algorithm>
#include <new>
#include <vector>

struct ogd_page {
  int device = -1;
  int H = 0, W = 0, C = 0;
  ptrdiff_t stride = 0;
  uint8_t* d = nullptr;
};

namespace ogd {

struct LineRec {        // one line, fp64, dewarped pixel coordinates
  double p0x, p0y;      // origin
  double rx, ry;        // `right` (end of the x lattice)
  double ux, uy;        // `up` (end of the y lattice)
  double sxx, sxy;      // x step = right / (xsteps - 1)   (np.linspace: step = delta / div)
  double syx, syy;      // y step = up / (ysteps - 1)
  int64_t offset;       // first output byte
  int xsteps, ysteps;
};

template <int C, int INTERP>
__global__ __launch_bounds__(256) void line_images_kernel(const uint8_t* __restrict__ src, int H, int W,
                                                          ptrdiff_t ss, const float2* __restrict__ nodes,
                                                          int h, int w, int s, double inv_s,
                                                          const LineRec* __restrict__ lines,
                                                          uint8_t* __restrict__ out) {
  const LineRec L = lines[blockIdx.y];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= L.xsteps * L.ysteps) return;
  const int r = idx / L.xsteps, j = idx - r * L.xsteps;
  const int i = L.ysteps - 1 - r;  // pixels[::-1, :]
  // np.linspace([0, 0], v, n): k * step, the last sample exactly v
  const double yx = i == L.ysteps - 1 && i > 0 ? L.ux : __dmul_rn((double)i, L.syx);
  const double yy = i == L.ysteps - 1 && i > 0 ? L.uy : __dmul_rn((double)i, L.syy);
  const double xx = j == L.xsteps - 1 && j > 0 ? L.rx : __dmul_rn((double)j, L.sxx);
  const double xy = j == L.xsteps - 1 && j > 0 ? L.ry : __dmul_rn((double)j, L.sxy);
  const double px = __dadd_rn(__dadd_rn(yx, L.p0x), xx), py = __dadd_rn(__dadd_rn(yy, L.p0y), xy);
  const double2 wq = grid_inverse_at(nodes, h, w, s, inv_s, py, px);  // warped page (x, y)
  uint8_t o[C];
  sample_q<C, INTERP>(src, H, W, ss, quantize<INTERP>((float)wq.x), quantize<INTERP>((float)wq.y), o);
  uint8_t* d = out + L.offset + (int64_t)idx * C;
#pragma unroll
  for (int c = 0; c < C; ++c) d[c] = o[c];
}

}  // namespace ogd

using namespace ogd;

extern "C" {

int ogd_page_create(const uint8_t* src, int H, int W, int C, ptrdiff_t src_stride, ogd_page** out) {
  OGD_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  OGD_REQUIRE(src != nullptr, "null pointer");
  OGD_REQUIRE(H >= 1 && W >= 1 && H < 32768 && W < 32768, "page %dx%d not supported", W, H);
  OGD_REQUIRE(C == 1 || C == 3 || C == 4, "%d channels not supported (1, 3 or 4)", C);
  OGD_REQUIRE(src_stride >= (ptrdiff_t)W * C, "row stride %td < %d bytes", src_stride, W * C);
  OGD_TRY(ensure_device());
  ogd_page* p = new (std::nothrow) ogd_page();
  if (!p) return fail(OGD_ERR_NOMEM, "out of host memory");
  (void)hipGetDevice(&p->device);
  p->H = H; p->W = W; p->C = C;
  p->stride = (ptrdiff_t)W * C;  // packed on the device
  hipError_t e = hipMalloc((void**)&p->d, (size_t)p->stride * H);
  if (e != hipSuccess) {
    delete p;
    return fail(OGD_ERR_NOMEM, "hipMalloc(%zu) failed: %s", (size_t)W * C * H, hipGetErrorString(e));
  }
  e = copy_rows(p->d, p->stride, src, src_stride, (size_t)W * C, H, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(p->d);
    delete p;
    return fail(OGD_ERR_HIP, "page upload failed: %s", hipGetErrorString(e));
  }
  *out = p;
  return OGD_OK;
}

int ogd_page_destroy(ogd_page* p) {
  if (!p) return OGD_OK;
  int prev = -1;
  if (hipGetDevice(&prev) == hipSuccess && prev != p->device) (void)hipSetDevice(p->device);
  if (p->d) (void)hipFree(p->d);
  if (prev >= 0 && prev != p->device) (void)hipSetDevice(prev);
  delete p;
  return OGD_OK;
}

int ogd_page_dewarp_u8(const ogd_page* page, const float* nodes, int h, int w, int s, float sx, float sy,
                       uint8_t* dst, ptrdiff_t dst_stride, int interp) {
  OGD_REQUIRE(page && nodes && dst, "null pointer");
  OGD_REQUIRE(h >= 1 && w >= 1 && s >= 1, "bad node grid %dx%d cell %d", h, w, s);
  OGD_REQUIRE((int64_t)h * s < 32768 && (int64_t)w * s < 32768, "dense grid too large");
  OGD_TRY(ensure_device());
  const int Hd = h * s, Wd = w * s;
  const size_t nb = (size_t)h * w * 2 * sizeof(float), drow = (size_t)Wd * page->C;
  OGD_REQUIRE(dst_stride >= (ptrdiff_t)drow, "dst: row stride %td < %zu bytes", dst_stride, drow);
  DevBuf dn, ddst;
  OGD_TRY(dn.alloc(nb));
  OGD_TRY(ddst.alloc(drow * Hd));
  OGD_HIP(hipMemcpy(dn.p, nodes, nb, hipMemcpyHostToDevice));
  OGD_TRY(ogd_dewarp_u8_dev(page->d, page->H, page->W, page->C, page->stride, dn.as<float>(), h, w, s, sx, sy,
                            ddst.as<uint8_t>(), (ptrdiff_t)drow, interp, nullptr));
  OGD_HIP(copy_rows(dst, dst_stride, ddst.p, drow, drow, Hd, hipMemcpyDeviceToHost));
  return OGD_OK;
}

int ogd_line_images_u8(const ogd_page* page, const float* nodes, int h, int w, int s, const double* lines,
                       const int32_t* xsteps, const int32_t* ysteps, int n, int interp, uint8_t* out,
                       const int64_t* offsets) {
  OGD_REQUIRE(page != nullptr, "page is NULL");
  OGD_REQUIRE(n >= 0, "negative line count");
  if (n == 0) return OGD_OK;
  OGD_REQUIRE(nodes && lines && xsteps && ysteps && out && offsets, "null pointer");
  OGD_REQUIRE(h >= 1 && w >= 1 && s >= 1, "bad node grid %dx%d cell %d", h, w, s);
  OGD_REQUIRE((int64_t)h * s >= 2 && (int64_t)w * s >= 2, "dense grid must be at least 2x2");
  OGD_REQUIRE((int64_t)h * s < INT_MAX && (int64_t)w * s < INT_MAX, "dense grid too large");
  OGD_REQUIRE(n <= 65535, "at most 65535 lines per call (got %d)", n);
  OGD_REQUIRE(interp == OGD_INTER_NEAREST || interp == OGD_INTER_LINEAR, "interp %d not supported", interp);
  OGD_REQUIRE(offsets[0] == 0, "offsets[0] must be 0");
  const int C = page->C;
  std::vector<LineRec> recs((size_t)n);
  int64_t maxpix = 0;
  for (int l = 0; l < n; ++l) {
    OGD_REQUIRE(xsteps[l] >= 1 && ysteps[l] >= 1, "line %d: steps must be >= 1", l);
    const int64_t pix = (int64_t)xsteps[l] * ysteps[l];
    OGD_REQUIRE(pix < INT_MAX, "line %d: image too large", l);
    OGD_REQUIRE(offsets[l + 1] - offsets[l] == pix * C, "line %d: offsets do not match %d x %d x %d bytes", l,
                ysteps[l], xsteps[l], C);
    maxpix = std::max(maxpix, pix);
    const double* v = lines + 6 * (size_t)l;
    LineRec& r = recs[l];
    r.p0x = v[0]; r.p0y = v[1]; r.rx = v[2]; r.ry = v[3]; r.ux = v[4]; r.uy = v[5];
    const double dx = xsteps[l] > 1 ? (double)(xsteps[l] - 1) : 1.0, dy = ysteps[l] > 1 ? (double)(ysteps[l] - 1) : 1.0;
    r.sxx = r.rx / dx; r.sxy = r.ry / dx; r.syx = r.ux / dy; r.syy = r.uy / dy;
    r.offset = offsets[l];
    r.xsteps = xsteps[l]; r.ysteps = ysteps[l];
  }
  OGD_TRY(ensure_device());
  const size_t nb = (size_t)h * w * 2 * sizeof(float), ob = (size_t)offsets[n];
  DevBuf dn, dl, dout;
  OGD_TRY(dn.alloc(nb));
  OGD_TRY(dl.alloc(recs.size() * sizeof(LineRec)));
  OGD_TRY(dout.alloc(ob));
  OGD_HIP(hipMemcpy(dn.p, nodes, nb, hipMemcpyHostToDevice));
  OGD_HIP(hipMemcpy(dl.p, recs.data(), recs.size() * sizeof(LineRec), hipMemcpyHostToDevice));
  const dim3 grid((unsigned)((maxpix + 255) / 256), (unsigned)n), block(256);
  const float2* dnodes = dn.as<float2>();
  const double inv_s = 1.0 / (double)s;
#define OGD_LINES(CC, II)                                                                                  \
  hipLaunchKernelGGL((line_images_kernel<CC, II>), grid, block, 0, nullptr, page->d, page->H, page->W,       \
                     page->stride, dnodes, h, w, s, inv_s, dl.as<LineRec>(), dout.as<uint8_t>())
  if (interp == OGD_INTER_LINEAR) {
    if (C == 1) OGD_LINES(1, OGD_INTER_LINEAR); else if (C == 3) OGD_LINES(3, OGD_INTER_LINEAR); else OGD_LINES(4, OGD_INTER_LINEAR);
  } else {
    if (C == 1) OGD_LINES(1, OGD_INTER_NEAREST); else if (C == 3) OGD_LINES(3, OGD_INTER_NEAREST); else OGD_LINES(4, OGD_INTER_NEAREST);
  }
#undef OGD_LINES
  OGD_TRY(check_launch("line_images_kernel"));
  OGD_HIP(hipMemcpy(out, dout.p, ob, hipMemcpyDeviceToHost));
  return OGD_OK;
}

}  // extern "C"
