// K1a with line-aligned stores (ref: origami/core/dewarp.py:514-526, cv2.resize of the node planes).
//
// The dense grid is a write-only stream of 8 B per pixel whose rows are Wd * 8 bytes long: 48 600 B for a
// broadsheet page, so row starts fall anywhere in a 128-byte line.  A wavefront that stores "its" 128 columns
// of a row then writes seven full lines and two partial ones, and a partial line costs the memory system a
// second visit: the column-owning kernel streams at 4.3 TB/s on such rows and at 5.4 TB/s when Wd happens to be
// a multiple of 16 (tools/kbench, 6075 against 6000 columns).  Here the workgroup first builds what does not depend
// on the row — the horizontally interpolated node rows of its column range — in LDS, and then every dense row is
// written in spans that begin on a line of the OUTPUT ADDRESS: the columns a lane stores shift from row to row
// by that row's phase, which costs nothing because the values come from LDS, not from the lane's registers.
// Only the first and the last line of a row are partial (they are shared with the neighbouring rows).
//
// Arithmetic per pixel is that of grid_fill_kernel, operation for operation: tap_x / tap_y, horizontal lerp2 of
// the two node rows, vertical lerp2, multiplication by the scale.
#pragma once
#include "ogd_resize_math.h"

namespace ogd {

typedef float gl_f4 __attribute__((ext_vector_type(4)));

// TW columns (a multiple of 16, at most 512) and TR rows per workgroup of NTHREADS = 256 * RG threads; needs
// TR <= s, so that the band's rows touch at most three node rows.  Threads 256 g .. 256 g + 255 write rows g,
// g + RG, ... of the band.  A store-only kernel reaches 6.7 TB/s with one 16-byte store per thread, 6.0 with two,
// 5.4 with eight (tools/kbench "chunk fill"), and 512 threads do run this kernel faster by itself (64 against 75 us
// per broadsheet page) — but the K2 that reads the map next then takes longer by more than that (169 against 164 us
// for the pair), so the library launches 256 threads (profiles/r02/m_kbench_k1a_lines.log; round 3, with the K2
// launches of a group of pages behind the K1a launches of that group: 158.5 against 154.2 us, profiles/r03/ab_phase_order.log).
// The grid is (ceil((Wd + 15) / TW), ceil(rows / TR)).
template <int TW, int TR, bool NT, int NTHREADS = 256>
__global__ __launch_bounds__(NTHREADS) void grid_fill_lines_kernel(const float2* __restrict__ nodes, int h, int w,
                                                              double inv_s, float scx, float scy,
                                                              float2* __restrict__ out, int ybeg, int yend, int Wd) {
  static_assert(TW % 16 == 0 && TW <= 512, "spans are whole 128-byte lines; one pair of columns per thread");
  static_assert(NTHREADS % 256 == 0, "row groups of 256 threads");
  constexpr int RG = NTHREADS / 256;
  constexpr int LW = TW + 16;  // staged columns [X0 - 16, X0 + TW): a row's span begins up to 15 columns left of X0
  __shared__ float2 H[3][LW];
  const int X0 = blockIdx.x * TW;
  const int Y0 = ybeg + blockIdx.y * TR;
  const int Y1 = min(Y0 + TR, yend);
  const int rbase = tap_y(Y0, inv_s, h).i0;  // node rows rbase .. rbase + 2 (clipped) serve rows Y0 .. Y1 - 1
  const float2* n0 = nodes + (size_t)rbase * w;
  const float2* n1 = nodes + (size_t)min(rbase + 1, h - 1) * w;
  const float2* n2 = nodes + (size_t)min(rbase + 2, h - 1) * w;
  for (int cx = threadIdx.x; cx < LW; cx += NTHREADS) {
    const int X = X0 - 16 + cx;
    if (X < 0 || X >= Wd) continue;
    const Tap tx = tap_x(X, inv_s, w);
    float2 a = n0[tx.i0], b = n0[tx.i1];
    H[0][cx] = make_float2(lerp2(a.x, tx.w0, b.x, tx.w1), lerp2(a.y, tx.w0, b.y, tx.w1));
    a = n1[tx.i0], b = n1[tx.i1];
    H[1][cx] = make_float2(lerp2(a.x, tx.w0, b.x, tx.w1), lerp2(a.y, tx.w0, b.y, tx.w1));
    a = n2[tx.i0], b = n2[tx.i1];
    H[2][cx] = make_float2(lerp2(a.x, tx.w0, b.x, tx.w1), lerp2(a.y, tx.w0, b.y, tx.w1));
  }
  __syncthreads();
  const unsigned long long base8 = (unsigned long long)reinterpret_cast<uintptr_t>(out) >> 3;  // in pixels of 8 B
  const int j = threadIdx.x & 255;
  if (j >= TW / 2) return;
  for (int Y = Y0 + (threadIdx.x >> 8); Y < Y1; Y += RG) {
    const Tap ty = tap_y(Y, inv_s, h);
    const float2* ha = H[ty.i0 - rbase];
    const float2* hb = H[ty.i1 - rbase];
    const size_t row = (size_t)(Y - ybeg) * (size_t)Wd;
    const int delta = (int)((base8 + row + (unsigned long long)X0) & 15ull);  // the span begins delta columns left of X0
    float2* orow = out + row;
    const int c = 2 * j - delta + 16;  // staged column of the pair's first pixel, >= 1
    const int X = X0 - 16 + c;
    const bool v0 = X >= 0 && X < Wd, v1 = X + 1 >= 0 && X + 1 < Wd;
    if (!v0 && !v1) continue;
    const float2 a0 = ha[c], a1 = ha[c + 1], b0 = hb[c], b1 = hb[c + 1];
    gl_f4 v;
    v.x = __fmul_rn(lerp2(a0.x, ty.w0, b0.x, ty.w1), scx);
    v.y = __fmul_rn(lerp2(a0.y, ty.w0, b0.y, ty.w1), scy);
    v.z = __fmul_rn(lerp2(a1.x, ty.w0, b1.x, ty.w1), scx);
    v.w = __fmul_rn(lerp2(a1.y, ty.w0, b1.y, ty.w1), scy);
    if (v0 && v1) {
      gl_f4* o = reinterpret_cast<gl_f4*>(orow + X);  // 16-byte aligned: X - X0 + delta is even, the span's start a line
      if (NT) __builtin_nontemporal_store(v, o);
      else *o = v;
    } else if (v0) {
      orow[X] = make_float2(v.x, v.y);
    } else {
      orow[X + 1] = make_float2(v.z, v.w);
    }
  }
}

}  // namespace ogd
