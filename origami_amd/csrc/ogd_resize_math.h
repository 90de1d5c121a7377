// Device arithmetic of the dense sampling grid: the value cv2.resize(INTER_LINEAR) gives at dense
// pixel (X, Y) of an [h][w] float32 node plane upscaled by the integer cell size s.
// ref: origami/core/dewarp.py:522-523.  Restated from OpenCV 4.2 imgproc resize (generic float path):
//   fx = (float)((X + 0.5) * (1.0 / s) - 0.5); ix = floor(fx); fx -= ix        (double, then float)
//   x axis: ix < 0 -> ix = 0, fx = 0;  ix >= w-1 -> ix = w-1, fx = 0          (table set-up)
//   y axis: taps are rows clip(iy), clip(iy + 1); the weights keep fy          (row clip in the invoker)
//   tmp = S[ix] * (1 - fx) + S[ix + 1] * fx;  dst = tmp0 * (1 - fy) + tmp1 * fy   (fp32, no FMA)
#pragma once
#include <hip/hip_runtime.h>

namespace ogd {

struct Tap {
  int i0, i1;
  float w0, w1;
};

__device__ __forceinline__ Tap tap_x(int X, double inv_s, int n) {
  float f = (float)(((double)X + 0.5) * inv_s - 0.5);
  float fl = floorf(f);
  int i = (int)fl;
  f = f - fl;
  if (i < 0) {
    i = 0;
    f = 0.f;
  }
  if (i >= n - 1) {
    i = n - 1;
    f = 0.f;
  }
  Tap t;
  t.i0 = i;
  t.i1 = min(i + 1, n - 1);
  t.w0 = 1.f - f;
  t.w1 = f;
  return t;
}

__device__ __forceinline__ Tap tap_y(int Y, double inv_s, int n) {
  float f = (float)(((double)Y + 0.5) * inv_s - 0.5);
  float fl = floorf(f);
  int i = (int)fl;
  f = f - fl;
  Tap t;
  t.i0 = min(max(i, 0), n - 1);
  t.i1 = min(max(i + 1, 0), n - 1);
  t.w0 = 1.f - f;
  t.w1 = f;
  return t;
}

// a*wa + b*wb with two roundings of the products and one of the sum (never contracted to an FMA)
__device__ __forceinline__ float lerp2(float a, float wa, float b, float wb) {
  return __fadd_rn(__fmul_rn(a, wa), __fmul_rn(b, wb));
}

// lerp2 on an (x, y) pair: one v_pk_mul_f32 per product and a v_pk_add_f32, every element rounded exactly as
// the scalar form rounds it
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 lerp2_pk(f32x2 a, float wa, f32x2 b, float wb) {
#pragma clang fp contract(off)
  const f32x2 pa = a * wa, pb = b * wb;
  return pa + pb;
}
__device__ __forceinline__ f32x2 mul_pk(f32x2 a, f32x2 b) {
#pragma clang fp contract(off)
  return a * b;
}

// Dense map value at (X, Y); nodes is [h][w] float2.
template <class NodePtr>
__device__ __forceinline__ float2 dense_at(NodePtr nodes, int w, const Tap& tx, const Tap& ty) {
  const NodePtr r0 = nodes + (size_t)ty.i0 * w;
  const NodePtr r1 = nodes + (size_t)ty.i1 * w;
  float2 a = r0[tx.i0], b = r0[tx.i1], c = r1[tx.i0], d = r1[tx.i1];
  float t0x = lerp2(a.x, tx.w0, b.x, tx.w1), t0y = lerp2(a.y, tx.w0, b.y, tx.w1);
  float t1x = lerp2(c.x, tx.w0, d.x, tx.w1), t1y = lerp2(c.y, tx.w0, d.y, tx.w1);
  return make_float2(lerp2(t0x, ty.w0, t1x, ty.w1), lerp2(t0y, ty.w0, t1y, ty.w1));
}

// Inverse point map: bilinear (fp64) sample of the dense grid at fractional (y, x), extrapolating
// linearly from the edge cell.  ref: origami/core/dewarp.py:561-576.  Follows scipy
// RegularGridInterpolator over (arange(Hd), arange(Wd)):
//   i = searchsorted(grid, x, 'left') - 1, clamped to [0, n-2]; t = (x - i) / 1
//   value = sum over (i, j) in [(i0,j0), (i0,j1), (i1,j0), (i1,j1)] of v * ((1 * wy) * wx)
// The dense grid values are recomputed from the nodes (K1a arithmetic).
__device__ __forceinline__ int rgi_index(double x, int n) {
  int i = (int)ceil(x) - 1;  // searchsorted(left) on integers: #{k : k < x} - 1
  if (!(x == x)) i = n - 1;  // NaN sorts last
  i = max(i, 0);
  i = min(i, n - 2);
  return i;
}

template <class NodePtr>
__device__ __forceinline__ double2 grid_inverse_at(NodePtr nodes, int h, int w, int s, double inv_s, double y,
                                                   double x) {
  const int Hd = h * s, Wd = w * s;
  const int iy = rgi_index(y, Hd), ix = rgi_index(x, Wd);
  const double ty = y - (double)iy, tx = x - (double)ix;
  const Tap ax0 = tap_x(ix, inv_s, w), ax1 = tap_x(ix + 1, inv_s, w);
  const Tap ay0 = tap_y(iy, inv_s, h), ay1 = tap_y(iy + 1, inv_s, h);
  // (sharing the node loads when the four pixels lie in one node cell was measured: a 64-px wavefront always
  // straddles a 25-px cell border, both branches run, nothing gained)
  const float2 v00 = dense_at(nodes, w, ax0, ay0), v01 = dense_at(nodes, w, ax1, ay0);
  const float2 v10 = dense_at(nodes, w, ax0, ay1), v11 = dense_at(nodes, w, ax1, ay1);
  const double w00 = __dmul_rn(1.0 - ty, 1.0 - tx), w01 = __dmul_rn(1.0 - ty, tx);
  const double w10 = __dmul_rn(ty, 1.0 - tx), w11 = __dmul_rn(ty, tx);
  double ox = 0.0, oy = 0.0;
  ox = __dadd_rn(ox, __dmul_rn((double)v00.x, w00));
  oy = __dadd_rn(oy, __dmul_rn((double)v00.y, w00));
  ox = __dadd_rn(ox, __dmul_rn((double)v01.x, w01));
  oy = __dadd_rn(oy, __dmul_rn((double)v01.y, w01));
  ox = __dadd_rn(ox, __dmul_rn((double)v10.x, w10));
  oy = __dadd_rn(oy, __dmul_rn((double)v10.y, w10));
  ox = __dadd_rn(ox, __dmul_rn((double)v11.x, w11));
  oy = __dadd_rn(oy, __dmul_rn((double)v11.y, w11));
  return make_double2(ox, oy);
}

}  // namespace ogd
