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

// Warped-page position (x, y) of lattice sample (i, j) of a line, fp64.  ref: block.py:238-245.
__device__ __forceinline__ double2 line_warped_xy(const LineRec& L, int i, int j, const float2* __restrict__ nodes,
                                                  int h, int w, int s, double inv_s) {
  // np.linspace([0, 0], v, n): k * step, the last sample exactly v
  const double yx = i == L.ysteps - 1 && i > 0 ? L.ux : __dmul_rn((double)i, L.syx);
  const double yy = i == L.ysteps - 1 && i > 0 ? L.uy : __dmul_rn((double)i, L.syy);
  const double xx = j == L.xsteps - 1 && j > 0 ? L.rx : __dmul_rn((double)j, L.sxx);
  const double xy = j == L.xsteps - 1 && j > 0 ? L.ry : __dmul_rn((double)j, L.sxy);
  const double px = __dadd_rn(__dadd_rn(yx, L.p0x), xx), py = __dadd_rn(__dadd_rn(yy, L.p0y), xy);
  return grid_inverse_at(nodes, h, w, s, inv_s, py, px);
}

template <int C, int INTERP>
__global__ __launch_bounds__(256) void line_images_kernel(const uint8_t* __restrict__ src, int H, int W,
                                                          ptrdiff_t ss, const float2* __restrict__ nodes,
                                                          int h, int w, int s, double inv_s,
                                                          const LineRec* __restrict__ lines,
                                                          uint8_t* __restrict__ out) {
  // a block is 4 rows x 64 columns of one line image; blocks beyond the line's own extent leave at once
  const LineRec L = lines[blockIdx.z];
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (j >= L.xsteps || r >= L.ysteps) return;
  const int idx = r * L.xsteps + j;
  const int i = L.ysteps - 1 - r;  // pixels[::-1, :]
  const double2 wq = line_warped_xy(L, i, j, nodes, h, w, s, inv_s);
  uint8_t o[C];
  sample_q<C, INTERP>(src, H, W, ss, quantize<INTERP>((float)wq.x), quantize<INTERP>((float)wq.y), o);
  uint8_t* d = out + L.offset + (int64_t)idx * C;
#pragma unroll
  for (int c = 0; c < C; ++c) d[c] = o[c];
}

// Label histogram of a line's lattice: cv2.remap(labels, warped_grid * (sx, sy), INTER_NEAREST) followed by
// np.bincount — the image itself is never written.  ref: batch/detect/lines.py:34-44 (ConfidenceSampler).
__global__ __launch_bounds__(256) void line_label_counts_kernel(const uint8_t* __restrict__ labels, int H, int W,
                                                                ptrdiff_t ss, const float2* __restrict__ nodes,
                                                                int h, int w, int s, double inv_s, float sx,
                                                                float sy, const LineRec* __restrict__ lines,
                                                                int32_t* __restrict__ counts) {
  __shared__ int hist[256];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const LineRec L = lines[blockIdx.z];
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (j < L.xsteps && i < L.ysteps) {
    const double2 wq = line_warped_xy(L, i, j, nodes, h, w, s, inv_s);
    // scale_grid(): float32 array times the size ratio, in float32 (lines.py:18-22)
    const float mx = __fmul_rn((float)wq.x, sx), my = __fmul_rn((float)wq.y, sy);
    uint8_t o[1];
    sample_q<1, OGD_INTER_NEAREST>(labels, H, W, ss, quantize<OGD_INTER_NEAREST>(mx),
                                   quantize<OGD_INTER_NEAREST>(my), o);
    atomicAdd(&hist[o[0]], 1);
  }
  __syncthreads();
  const int c = hist[threadIdx.x];
  if (c) atomicAdd(&counts[(size_t)blockIdx.z * 256 + threadIdx.x], c);
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

int ogd_page_create_jpeg(const uint8_t* file, size_t n, ogd_page** out) {
  OGD_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  OGD_REQUIRE(file != nullptr, "null pointer");
  int H = 0, W = 0, ncomp = 0, supported = 0;
  OGD_TRY(ogd_jpeg_info(file, n, &H, &W, &ncomp, &supported));
  if (!supported) return OGD_ERR_UNSUPPORTED;  // ogd_last_error() already names the reason
  OGD_REQUIRE(H < 32768 && W < 32768, "page %dx%d not supported", W, H);
  OGD_TRY(ensure_device());
  ogd_page* p = new (std::nothrow) ogd_page();
  if (!p) return fail(OGD_ERR_NOMEM, "out of host memory");
  (void)hipGetDevice(&p->device);
  p->H = H; p->W = W; p->C = 1;
  p->stride = (ptrdiff_t)W;
  hipError_t e = hipMalloc((void**)&p->d, (size_t)W * H);
  if (e != hipSuccess) {
    delete p;
    return fail(OGD_ERR_NOMEM, "hipMalloc(%zu) failed: %s", (size_t)W * H, hipGetErrorString(e));
  }
  const int st = ogd_jpeg_decode_gray_dev(file, n, p->d, H, W, W, nullptr);
  if (st != OGD_OK) {
    (void)hipFree(p->d);
    delete p;
    return st;
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

int ogd_page_dewarp_jpeg_gray(const ogd_page* page, const float* nodes, int h, int w, int s, int quality, uint8_t* out,
                              size_t cap, size_t* out_size) {
  OGD_REQUIRE(page && nodes && out_size, "null pointer");
  OGD_REQUIRE(page->C == 1, "a gray page is needed, this one has %d channels", page->C);
  OGD_REQUIRE(h >= 1 && w >= 1 && s >= 1, "bad node grid %dx%d cell %d", h, w, s);
  OGD_REQUIRE((int64_t)h * s < 32768 && (int64_t)w * s < 32768, "dense grid too large");
  OGD_TRY(ensure_device());
  const int Hd = h * s, Wd = w * s;
  const size_t nb = (size_t)h * w * 2 * sizeof(float);
  DevBuf dn, ddst;
  OGD_TRY(dn.alloc(nb));
  OGD_TRY(ddst.alloc((size_t)Hd * Wd));
  OGD_HIP(hipMemcpy(dn.p, nodes, nb, hipMemcpyHostToDevice));
  OGD_TRY(ogd_dewarp_u8_dev(page->d, page->H, page->W, 1, page->stride, dn.as<float>(), h, w, s, 1.f, 1.f,
                            ddst.as<uint8_t>(), (ptrdiff_t)Wd, OGD_INTER_LINEAR, nullptr));
  return ogd_jpeg_encode_gray_dev(ddst.as<uint8_t>(), Hd, Wd, (ptrdiff_t)Wd, quality, out, cap, out_size, nullptr);
}

}  // extern "C"

namespace {

// Validates the line arrays and builds the device records; offsets may be NULL (no per-line output).
int build_line_records(const double* lines, const int32_t* xsteps, const int32_t* ysteps, int n, int C,
                       const int64_t* offsets, std::vector<LineRec>& recs, int& max_x, int& max_y) {
  recs.resize((size_t)n);
  max_x = max_y = 0;
  for (int l = 0; l < n; ++l) {
    OGD_REQUIRE(xsteps[l] >= 1 && ysteps[l] >= 1, "line %d: steps must be >= 1", l);
    const int64_t pix = (int64_t)xsteps[l] * ysteps[l];
    OGD_REQUIRE(pix < INT_MAX, "line %d: image too large", l);
    if (offsets)
      OGD_REQUIRE(offsets[l + 1] - offsets[l] == pix * C, "line %d: offsets do not match %d x %d x %d bytes", l,
                  ysteps[l], xsteps[l], C);
    max_x = std::max(max_x, (int)xsteps[l]);
    max_y = std::max(max_y, (int)ysteps[l]);
    const double* v = lines + 6 * (size_t)l;
    LineRec& r = recs[l];
    r.p0x = v[0]; r.p0y = v[1]; r.rx = v[2]; r.ry = v[3]; r.ux = v[4]; r.uy = v[5];
    const double dx = xsteps[l] > 1 ? (double)(xsteps[l] - 1) : 1.0, dy = ysteps[l] > 1 ? (double)(ysteps[l] - 1) : 1.0;
    r.sxx = r.rx / dx; r.sxy = r.ry / dx; r.syx = r.ux / dy; r.syy = r.uy / dy;
    r.offset = offsets ? offsets[l] : 0;
    r.xsteps = xsteps[l]; r.ysteps = ysteps[l];
  }
  OGD_REQUIRE((max_y + 3) / 4 <= 65535, "line image too tall (%d rows)", max_y);
  return OGD_OK;
}

int check_line_args(const ogd_page* page, const void* nodes, int h, int w, int s, const void* lines,
                    const void* xsteps, const void* ysteps, int n) {
  OGD_REQUIRE(page != nullptr, "page is NULL");
  OGD_REQUIRE(nodes && lines && xsteps && ysteps, "null pointer");
  OGD_REQUIRE(h >= 1 && w >= 1 && s >= 1, "bad node grid %dx%d cell %d", h, w, s);
  OGD_REQUIRE((int64_t)h * s >= 2 && (int64_t)w * s >= 2, "dense grid must be at least 2x2");
  OGD_REQUIRE((int64_t)h * s < INT_MAX && (int64_t)w * s < INT_MAX, "dense grid too large");
  OGD_REQUIRE(n <= 65535, "at most 65535 lines per call (got %d)", n);
  return OGD_OK;
}

}  // namespace

extern "C" {

// Shared body: validates, stages the node grid and the line records, launches into d_out (device memory of
// offsets[n] bytes) on `st`; the kernel has been enqueued, not awaited, when this returns OGD_OK.
static int line_images_launch(const ogd_page* page, const float* nodes, int h, int w, int s, const double* lines,
                              const int32_t* xsteps, const int32_t* ysteps, int n, int interp, uint8_t* d_out,
                              const int64_t* offsets, DevBuf& dn, DevBuf& dl, hipStream_t st) {
  OGD_TRY(check_line_args(page, nodes, h, w, s, lines, xsteps, ysteps, n));
  OGD_REQUIRE(d_out && offsets, "null pointer");
  OGD_REQUIRE(interp == OGD_INTER_NEAREST || interp == OGD_INTER_LINEAR, "interp %d not supported", interp);
  OGD_REQUIRE(offsets[0] == 0, "offsets[0] must be 0");
  const int C = page->C;
  std::vector<LineRec> recs;
  int max_x = 0, max_y = 0;
  OGD_TRY(build_line_records(lines, xsteps, ysteps, n, C, offsets, recs, max_x, max_y));
  OGD_TRY(ensure_device());
  const size_t nb = (size_t)h * w * 2 * sizeof(float);
  OGD_TRY(dn.alloc(nb));
  OGD_TRY(dl.alloc(recs.size() * sizeof(LineRec)));
  // pageable host sources: both copies have left the host buffers when the calls return
  OGD_HIP(hipMemcpyAsync(dn.p, nodes, nb, hipMemcpyHostToDevice, st));
  OGD_HIP(hipMemcpyAsync(dl.p, recs.data(), recs.size() * sizeof(LineRec), hipMemcpyHostToDevice, st));
  const dim3 grid((unsigned)((max_x + 63) / 64), (unsigned)((max_y + 3) / 4), (unsigned)n), block(256);
  const float2* dnodes = dn.as<float2>();
  const double inv_s = 1.0 / (double)s;
#define OGD_LINES(CC, II)                                                                                  \
  hipLaunchKernelGGL((line_images_kernel<CC, II>), grid, block, 0, st, page->d, page->H, page->W,            \
                     page->stride, dnodes, h, w, s, inv_s, dl.as<LineRec>(), d_out)
  if (interp == OGD_INTER_LINEAR) {
    if (C == 1) OGD_LINES(1, OGD_INTER_LINEAR); else if (C == 3) OGD_LINES(3, OGD_INTER_LINEAR); else OGD_LINES(4, OGD_INTER_LINEAR);
  } else {
    if (C == 1) OGD_LINES(1, OGD_INTER_NEAREST); else if (C == 3) OGD_LINES(3, OGD_INTER_NEAREST); else OGD_LINES(4, OGD_INTER_NEAREST);
  }
#undef OGD_LINES
  return check_launch("line_images_kernel");
}

int ogd_line_images_u8(const ogd_page* page, const float* nodes, int h, int w, int s, const double* lines,
                       const int32_t* xsteps, const int32_t* ysteps, int n, int interp, uint8_t* out,
                       const int64_t* offsets) {
  OGD_REQUIRE(n >= 0, "negative line count");
  if (n == 0) return OGD_OK;
  OGD_REQUIRE(out && offsets, "null pointer");
  OGD_TRY(ensure_device());
  const size_t ob = (size_t)offsets[n];
  DevBuf dn, dl, dout;
  OGD_TRY(dout.alloc(ob));
  OGD_TRY(line_images_launch(page, nodes, h, w, s, lines, xsteps, ysteps, n, interp, dout.as<uint8_t>(), offsets, dn,
                             dl, nullptr));
  OGD_HIP(hipMemcpy(out, dout.p, ob, hipMemcpyDeviceToHost));
  return OGD_OK;
}

int ogd_line_images_u8_dev(const ogd_page* page, const float* nodes, int h, int w, int s, const double* lines,
                           const int32_t* xsteps, const int32_t* ysteps, int n, int interp, uint8_t* d_out,
                           const int64_t* offsets, void* stream) {
  OGD_REQUIRE(n >= 0, "negative line count");
  if (n == 0) return OGD_OK;
  DevBuf dn, dl;  // node grid and line records: staged through the block cache, free again once the stream is idle
  OGD_TRY(line_images_launch(page, nodes, h, w, s, lines, xsteps, ysteps, n, interp, d_out, offsets, dn, dl,
                             as_stream(stream)));
  OGD_HIP(hipStreamSynchronize(as_stream(stream)));
  return OGD_OK;
}

int ogd_line_label_counts(const ogd_page* labels, const float* nodes, int h, int w, int s, float sx, float sy,
                          const double* lines, const int32_t* xsteps, const int32_t* ysteps, int n,
                          int32_t* counts) {
  OGD_REQUIRE(n >= 0, "negative line count");
  if (n == 0) return OGD_OK;
  OGD_TRY(check_line_args(labels, nodes, h, w, s, lines, xsteps, ysteps, n));
  OGD_REQUIRE(counts != nullptr, "null pointer");
  OGD_REQUIRE(labels->C == 1, "label map must have one channel (has %d)", labels->C);
  std::vector<LineRec> recs;
  int max_x = 0, max_y = 0;
  OGD_TRY(build_line_records(lines, xsteps, ysteps, n, 1, nullptr, recs, max_x, max_y));
  OGD_TRY(ensure_device());
  const size_t nb = (size_t)h * w * 2 * sizeof(float), cb = (size_t)n * 256 * sizeof(int32_t);
  DevBuf dn, dl, dc;
  OGD_TRY(dn.alloc(nb));
  OGD_TRY(dl.alloc(recs.size() * sizeof(LineRec)));
  OGD_TRY(dc.alloc(cb));
  OGD_HIP(hipMemcpy(dn.p, nodes, nb, hipMemcpyHostToDevice));
  OGD_HIP(hipMemcpy(dl.p, recs.data(), recs.size() * sizeof(LineRec), hipMemcpyHostToDevice));
  OGD_HIP(hipMemsetAsync(dc.p, 0, cb, nullptr));
  const dim3 grid((unsigned)((max_x + 63) / 64), (unsigned)((max_y + 3) / 4), (unsigned)n), block(256);
  hipLaunchKernelGGL(line_label_counts_kernel, grid, block, 0, nullptr, labels->d, labels->H, labels->W,
                     labels->stride, dn.as<float2>(), h, w, s, 1.0 / (double)s, sx, sy, dl.as<LineRec>(),
                     dc.as<int32_t>());
  OGD_TRY(check_launch("line_label_counts_kernel"));
  OGD_HIP(hipMemcpy(counts, dc.p, cb, hipMemcpyDeviceToHost));
  return OGD_OK;
}

}  // extern "C"
