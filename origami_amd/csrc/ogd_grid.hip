// K1a — dense sampling grid from the node grid (ref: origami/core/dewarp.py:514-526), its strided
// sub-sampling (ref: dewarp.py:557-559) and the inverse point map (ref: dewarp.py:561-576).
//
// HBM-bound: 8 B written per dense pixel, 8 B read per node (nodes stay in L1/L2: 628 KB at
// 323 x 243).  Cells of 8 pixels and more go through grid_fill_lines_kernel (ogd_grid_lines.h: every store a
// whole 128-byte line of the output); grid_fill_kernel below serves smaller cells.
// In grid_fill_kernel each thread owns two adjacent columns and walks GF_ROWS rows, so the horizontal
// taps are computed once per thread and the horizontally-lerped node rows are re-used for the dense
// rows of the walk that fall into the same node cell; every store is a 16-byte float4 and a wavefront writes
// 1 KiB of one dense row per instruction.
#include "ogd_common.h"
#include "ogd_resize_math.h"
#include "ogd_grid_lines.h"

namespace ogd {

// 512 threads x 2 columns and 4 rows per workgroup: measured (tools/kbench "K1a") against 8 .. 64 rows and 256 ..
// 1024 threads — fewer rows per thread stream faster (91 vs 103 us alone, 178 vs 195 us with the K2 that follows),
// two rows or plain stores are faster alone but slow that K2 down by more than they gain.
constexpr int GF_THREADS = 512;
constexpr int GF_ROWS = 4;
#ifndef GL_COLS_V
#define GL_COLS_V 496
#endif
#ifndef GL_ROWS_V
#define GL_ROWS_V 8
#endif
#ifndef GL_NT
#define GL_NT true  // non-temporal stores
#endif
constexpr int GL_COLS = GL_COLS_V, GL_ROWS = GL_ROWS_V;  // grid_fill_lines_kernel's tile; its rows must not exceed the cell size
#ifndef GL_THREADS
#define GL_THREADS 256  // 512 is faster alone and slower with the K2 that follows (ogd_grid_lines.h); tools/build_variant_all.sh
#endif

typedef float float4_a8 __attribute__((ext_vector_type(4), aligned(8)));

__global__ __launch_bounds__(GF_THREADS) void grid_fill_kernel(
    const float2* __restrict__ nodes, int h, int w, double inv_s, float scx, float scy,
    float2* __restrict__ out, int ybeg, int yend, int Wd) {
  // dense rows [ybeg, yend) go to out rows 0 .. yend - ybeg - 1
  const int X = (blockIdx.x * GF_THREADS + threadIdx.x) * 2;
  if (X >= Wd) return;
  const bool pair = (X + 1 < Wd);
  const Tap tx0 = tap_x(X, inv_s, w);
  const Tap tx1 = tap_x(pair ? X + 1 : X, inv_s, w);
  const int Y0 = ybeg + blockIdx.y * GF_ROWS;
  const int Y1 = min(Y0 + GF_ROWS, yend);

  int cur0 = -1, cur1 = -1;
  float2 ta0 = {0.f, 0.f}, ta1 = {0.f, 0.f};  // horizontally lerped node row i0 for column X, X+1
  float2 tb0 = {0.f, 0.f}, tb1 = {0.f, 0.f};  // same for node row i1
  for (int Y = Y0; Y < Y1; ++Y) {
    const Tap ty = tap_y(Y, inv_s, h);  // wave-uniform
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
    float4_a8 v;
    v.x = __fmul_rn(lerp2(ta0.x, ty.w0, tb0.x, ty.w1), scx);
    v.y = __fmul_rn(lerp2(ta0.y, ty.w0, tb0.y, ty.w1), scy);
    v.z = __fmul_rn(lerp2(ta1.x, ty.w0, tb1.x, ty.w1), scx);
    v.w = __fmul_rn(lerp2(ta1.y, ty.w0, tb1.y, ty.w1), scy);
    float2* o = out + (size_t)(Y - ybeg) * Wd + X;
    if (pair)
      __builtin_nontemporal_store(v, reinterpret_cast<float4_a8*>(o));
    else
      *o = make_float2(v.x, v.y);
  }
}

__global__ __launch_bounds__(256) void grid_points_strided_kernel(
    const float2* __restrict__ nodes, int h, int w, double inv_s, int step, float2* __restrict__ out,
    int oh, int ow) {
  const int ox = blockIdx.x * blockDim.x + threadIdx.x;
  const int oy = blockIdx.y;
  if (ox >= ow || oy >= oh) return;
  const Tap tx = tap_x(ox * step, inv_s, w);
  const Tap ty = tap_y(oy * step, inv_s, h);
  out[(size_t)oy * ow + ox] = dense_at(nodes, w, tx, ty);
}

__global__ __launch_bounds__(256) void grid_sample_bilinear_kernel(
    const float2* __restrict__ nodes, int h, int w, int s, double inv_s, const double2* __restrict__ q_yx,
    int64_t nq, double2* __restrict__ out_xy) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nq) return;
  out_xy[k] = grid_inverse_at(nodes, h, w, s, inv_s, q_yx[k].x, q_yx[k].y);
}

static int check_nodes(int h, int w, int s) {
  OGD_REQUIRE(h >= 1 && w >= 1, "node grid must be at least 1x1 (got %dx%d)", h, w);
  OGD_REQUIRE(s >= 1, "cell size must be >= 1 (got %d)", s);
  OGD_REQUIRE((int64_t)h * s < INT_MAX && (int64_t)w * s < INT_MAX, "dense grid too large");
  return OGD_OK;
}

}  // namespace ogd

using namespace ogd;

extern "C" {

int ogd_grid_fill_rows_dev(const float* d_nodes, int h, int w, int s, float sx, float sy, int y0,
                           int nrows, float* d_out, void* stream) {
  OGD_TRY(check_nodes(h, w, s));
  OGD_REQUIRE(d_nodes && d_out, "null pointer");
  const int Hd = h * s, Wd = w * s;
  OGD_REQUIRE(y0 >= 0 && nrows >= 0 && (int64_t)y0 + nrows <= Hd, "rows [%d, %d + %d) outside the dense grid (%d rows)", y0, y0, nrows, Hd);
  if (nrows == 0) return OGD_OK;
  OGD_TRY(ensure_device());
  OGD_REQUIRE(reinterpret_cast<uintptr_t>(d_out) % 8 == 0, "dense grid buffer must be 8-byte aligned");
  if (s >= GL_ROWS) {
    // 496 columns x 8 rows per workgroup: measured against 240 .. 2032 columns and 4 .. 24 rows with the K2 that
    // follows (tools/kbench "lines", profiles/r02/m_kbench_k1a_lines.log)
    dim3 lgrid((Wd + 15 + GL_COLS - 1) / GL_COLS, (nrows + GL_ROWS - 1) / GL_ROWS);
    OGD_REQUIRE(lgrid.y <= 65535u, "dense grid too tall");
    hipLaunchKernelGGL((grid_fill_lines_kernel<GL_COLS, GL_ROWS, GL_NT, GL_THREADS>), lgrid, dim3(GL_THREADS), 0, as_stream(stream),
                       reinterpret_cast<const float2*>(d_nodes), h, w, 1.0 / (double)s, sx, sy,
                       reinterpret_cast<float2*>(d_out), y0, y0 + nrows, Wd);
    return check_launch("grid_fill_lines_kernel");
  }
  dim3 grid((Wd + 2 * GF_THREADS - 1) / (2 * GF_THREADS), (nrows + GF_ROWS - 1) / GF_ROWS);
  OGD_REQUIRE(grid.y <= 65535u, "dense grid too tall");
  hipLaunchKernelGGL(grid_fill_kernel, grid, dim3(GF_THREADS), 0, as_stream(stream),
                     reinterpret_cast<const float2*>(d_nodes), h, w, 1.0 / (double)s, sx, sy,
                     reinterpret_cast<float2*>(d_out), y0, y0 + nrows, Wd);
  return check_launch("grid_fill_kernel");
}

int ogd_grid_fill_dev(const float* d_nodes, int h, int w, int s, float sx, float sy, float* d_out,
                      void* stream) {
  OGD_TRY(check_nodes(h, w, s));
  return ogd_grid_fill_rows_dev(d_nodes, h, w, s, sx, sy, 0, h * s, d_out, stream);
}

int ogd_grid_fill(const float* nodes, int h, int w, int s, float sx, float sy, float* out) {
  OGD_TRY(check_nodes(h, w, s));
  OGD_REQUIRE(nodes && out, "null pointer");
  OGD_TRY(ensure_device());
  const size_t nb = (size_t)h * w * 2 * sizeof(float);
  const size_t ob = (size_t)h * s * (size_t)w * s * 2 * sizeof(float);
  DevBuf dn, dout;
  OGD_TRY(dn.alloc(nb));
  OGD_TRY(dout.alloc(ob));
  OGD_HIP(hipMemcpy(dn.p, nodes, nb, hipMemcpyHostToDevice));
  OGD_TRY(ogd_grid_fill_dev(dn.as<float>(), h, w, s, sx, sy, dout.as<float>(), nullptr));
  OGD_HIP(hipMemcpy(out, dout.p, ob, hipMemcpyDeviceToHost));
  return OGD_OK;
}

int ogd_grid_points_strided(const float* nodes, int h, int w, int s, int step, float* out) {
  OGD_TRY(check_nodes(h, w, s));
  OGD_REQUIRE(nodes && out, "null pointer");
  OGD_REQUIRE(step >= 1, "step must be >= 1");
  OGD_TRY(ensure_device());
  const int oh = (h * s + step - 1) / step, ow = (w * s + step - 1) / step;
  OGD_REQUIRE(oh <= 65535, "too many output rows");
  const size_t nb = (size_t)h * w * 2 * sizeof(float);
  const size_t ob = (size_t)oh * ow * 2 * sizeof(float);
  DevBuf dn, dout;
  OGD_TRY(dn.alloc(nb));
  OGD_TRY(dout.alloc(ob));
  OGD_HIP(hipMemcpy(dn.p, nodes, nb, hipMemcpyHostToDevice));
  dim3 grid((ow + 255) / 256, oh);
  hipLaunchKernelGGL(grid_points_strided_kernel, grid, dim3(256), 0, nullptr, dn.as<float2>(), h, w,
                     1.0 / (double)s, step, dout.as<float2>(), oh, ow);
  OGD_TRY(check_launch("grid_points_strided_kernel"));
  OGD_HIP(hipMemcpy(out, dout.p, ob, hipMemcpyDeviceToHost));
  return OGD_OK;
}

int ogd_grid_sample_bilinear(const float* nodes, int h, int w, int s, const double* q_yx, int64_t nq,
                             double* out_xy) {
  OGD_TRY(check_nodes(h, w, s));
  OGD_REQUIRE(nodes && (nq == 0 || (q_yx && out_xy)), "null pointer");
  OGD_REQUIRE(nq >= 0, "negative point count");
  OGD_REQUIRE((int64_t)h * s >= 2 && (int64_t)w * s >= 2, "dense grid must be at least 2x2");
  if (nq == 0) return OGD_OK;
  OGD_TRY(ensure_device());
  const size_t nb = (size_t)h * w * 2 * sizeof(float);
  const size_t qb = (size_t)nq * 2 * sizeof(double);
  DevBuf dn, dq, dout;
  OGD_TRY(dn.alloc(nb));
  OGD_TRY(dq.alloc(qb));
  OGD_TRY(dout.alloc(qb));
  OGD_HIP(hipMemcpy(dn.p, nodes, nb, hipMemcpyHostToDevice));
  OGD_HIP(hipMemcpy(dq.p, q_yx, qb, hipMemcpyHostToDevice));
  const int64_t nblk = (nq + 255) / 256;
  OGD_REQUIRE(nblk < INT_MAX, "too many points");
  hipLaunchKernelGGL(grid_sample_bilinear_kernel, dim3((unsigned)nblk), dim3(256), 0, nullptr,
                     dn.as<float2>(), h, w, s, 1.0 / (double)s, dq.as<double2>(), nq,
                     dout.as<double2>());
  OGD_TRY(check_launch("grid_sample_bilinear_kernel"));
  OGD_HIP(hipMemcpy(out_xy, dout.p, qb, hipMemcpyDeviceToHost));
  return OGD_OK;
}

}  // extern "C"
