// Device arithmetic of cv2.remap on uint8 sources (shared by the remap kernels and tools/kbench.hip).
// ref: origami/core/dewarp.py:612-614, 625-627.  Restated from OpenCV 4.2 imgproc remap (CV_32FC2 map,
// CV_8U source, BORDER_CONSTANT 0); see ogd_remap.hip for the arithmetic in prose.
#pragma once
#include <hip/hip_runtime.h>
#include <climits>
#include <cstddef>
#include <cstdint>
#include "../../include/ogdewarp.h"

namespace ogd {

__device__ __forceinline__ int cv_round(float v) {
  const float r = rintf(v);
  return (r >= -2147483648.f && r < 2147483648.f) ? (int)r : INT_MIN;
}

__device__ __forceinline__ int sat_i16(int v) { return min(max(v, -32768), 32767); }

template <int C>
__device__ __forceinline__ void sample_linear(const uint8_t* __restrict__ src, int H, int W,
                                              ptrdiff_t ss, float mx, float my, uint8_t* o) {
  const int sx = cv_round(__fmul_rn(mx, 32.f)), sy = cv_round(__fmul_rn(my, 32.f));
  const int ix = sat_i16(sx >> 5), iy = sat_i16(sy >> 5);
  const int a1 = sx & 31, b1 = sy & 31, a0 = 32 - a1, b0 = 32 - b1;
  if ((unsigned)ix < (unsigned)max(W - 1, 0) && (unsigned)iy < (unsigned)max(H - 1, 0)) {
    const uint8_t* p = src + (ptrdiff_t)iy * ss + (ptrdiff_t)ix * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int top = a0 * p[c] + a1 * p[C + c];
      const int bot = a0 * p[ss + c] + a1 * p[ss + C + c];
      o[c] = (uint8_t)((b0 * top + b1 * bot + 512) >> 10);
    }
  } else if (ix >= W || ix + 1 < 0 || iy >= H || iy + 1 < 0) {
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = 0;
  } else {
    const bool x0 = ix >= 0, x1 = ix + 1 < W, y0 = iy >= 0, y1 = iy + 1 < H;
    const uint8_t* p = src + (ptrdiff_t)iy * ss + (ptrdiff_t)ix * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int v00 = (x0 && y0) ? p[c] : 0, v01 = (x1 && y0) ? p[C + c] : 0;
      const int v10 = (x0 && y1) ? p[ss + c] : 0, v11 = (x1 && y1) ? p[ss + C + c] : 0;
      o[c] = (uint8_t)((b0 * (a0 * v00 + a1 * v01) + b1 * (a0 * v10 + a1 * v11) + 512) >> 10);
    }
  }
}

template <int C>
__device__ __forceinline__ void sample_nearest(const uint8_t* __restrict__ src, int H, int W,
                                               ptrdiff_t ss, float mx, float my, uint8_t* o) {
  const int ix = sat_i16(cv_round(mx)), iy = sat_i16(cv_round(my));
  if ((unsigned)ix < (unsigned)W && (unsigned)iy < (unsigned)H) {
    const uint8_t* p = src + (ptrdiff_t)iy * ss + (ptrdiff_t)ix * C;
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = p[c];
  } else {
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = 0;
  }
}

template <int C, int INTERP>
__device__ __forceinline__ void sample(const uint8_t* __restrict__ src, int H, int W, ptrdiff_t ss,
                                       float mx, float my, uint8_t* o) {
  if (INTERP == OGD_INTER_LINEAR)
    sample_linear<C>(src, H, W, ss, mx, my, o);
  else
    sample_nearest<C>(src, H, W, ss, mx, my, o);
}

template <int N>
__device__ __forceinline__ void store_bytes(uint8_t* dst, const uint8_t* v) {
  // N is a multiple of 4: dword stores at byte alignment (gfx950 runs in unaligned access mode)
#pragma unroll
  for (int k = 0; k < N; k += 4) {
    uint32_t u = (uint32_t)v[k] | ((uint32_t)v[k + 1] << 8) | ((uint32_t)v[k + 2] << 16) |
                 ((uint32_t)v[k + 3] << 24);
    __builtin_memcpy(dst + k, &u, 4);
  }
}

// ---- pre-quantised form (used by the LDS-tiled kernel) ---------------------------------------------
// quantize<LINEAR>(m) = cvRound(m * 32), quantize<NEAREST>(m) = cvRound(m).  The clamp to +-2^30 replaces
// cvtss2si's INT_MIN for NaN / out-of-range input: every such coordinate, like every |m| >= 32768,
// addresses a tap outside any image this library accepts (H, W < 32768), and v_max_f32 returns the
// non-NaN operand, so NaN lands on the negative bound.  Pixel index = q >> shift without the int16
// saturation: saturation never moves an index across the image edge for H, W < 32768.
template <int INTERP>
__device__ __forceinline__ int quantize(float m) {
  const float a = INTERP == OGD_INTER_LINEAR ? __fmul_rn(m, 32.f) : m;
  return (int)rintf(fminf(fmaxf(a, -1073741824.f), 1073741824.f));
}

// One output pixel from global memory, coordinates already quantised (same bytes as sample<>).
template <int C, int INTERP>
__device__ __forceinline__ void sample_q(const uint8_t* __restrict__ src, int H, int W, ptrdiff_t ss,
                                         int qx, int qy, uint8_t* o) {
  if (INTERP == OGD_INTER_LINEAR) {
    const int ix = qx >> 5, iy = qy >> 5;
    const int a1 = qx & 31, b1 = qy & 31, a0 = 32 - a1, b0 = 32 - b1;
    if (ix >= W || ix + 1 < 0 || iy >= H || iy + 1 < 0) {
#pragma unroll
      for (int c = 0; c < C; ++c) o[c] = 0;
      return;
    }
    const bool x0 = ix >= 0, x1 = ix + 1 < W, y0 = iy >= 0, y1 = iy + 1 < H;
    const uint8_t* p = src + (ptrdiff_t)iy * ss + (ptrdiff_t)ix * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int v00 = (x0 && y0) ? p[c] : 0, v01 = (x1 && y0) ? p[C + c] : 0;
      const int v10 = (x0 && y1) ? p[ss + c] : 0, v11 = (x1 && y1) ? p[ss + C + c] : 0;
      o[c] = (uint8_t)((b0 * (a0 * v00 + a1 * v01) + b1 * (a0 * v10 + a1 * v11) + 512) >> 10);
    }
  } else {
    if ((unsigned)qx < (unsigned)W && (unsigned)qy < (unsigned)H) {
      const uint8_t* p = src + (ptrdiff_t)qy * ss + (ptrdiff_t)qx * C;
#pragma unroll
      for (int c = 0; c < C; ++c) o[c] = p[c];
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) o[c] = 0;
    }
  }
}

}  // namespace ogd
