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

}  // namespace ogd
