/*
 * cv_oracle.c — TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * Plain-C CPU restatement of the two OpenCV calls on Origami's dewarp path:
 *   origami/core/dewarp.py:522-523   cv2.resize(f32 plane, (w*s, h*s), INTER_LINEAR)
 *   origami/core/dewarp.py:612-614, 625-627   cv2.remap(u8, map f32 (Hd,Wd,2), None, LINEAR | NEAREST)
 *
 * OpenCV (opencv-python 4.2.0.32, docker/pip_requirements.txt:96) is a third-party dependency that is
 * neither vendored in the reference nor installable here.  The arithmetic below restates the
 * published algorithm of OpenCV 4.2 imgproc (resize.cpp generic float path; imgwarp.cpp remap with a
 * CV_32FC2 map, CV_8U source, BORDER_CONSTANT 0).  PARITY UNPINNED: the reference ships no tests,
 * golden vectors or fixtures for these calls, and cv2 cannot be run here to produce any.
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off -fopenmp (oracle/build_oracle.py).
 */
#include <emmintrin.h>
#include <limits.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* cvRound(float) on x86-64 = cvtss2si: round-half-even, INT_MIN when out of range or NaN */
static inline int cv_round_f(float v) { return _mm_cvtss_si32(_mm_set_ss(v)); }

static inline int cv_floor_f(float v) {
  int i = (int)v;
  return i - (v < (float)i);
}

static inline short sat_i16(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

static inline int clip_row(int y, int n) { return y >= 0 ? (y < n ? y : n - 1) : 0; }

/*
 * resize.cpp (INTER_LINEAR, float, cn = 1, dsize = (w*s, h*s)):
 *   scale = 1. / ((double)dsize / ssize)  = 1.0 / s
 *   for dx: fx = (float)((dx + 0.5) * scale - 0.5); sx = cvFloor(fx); fx -= sx;
 *           if (sx < 0) fx = 0, sx = 0;   if (sx >= ssize.width - 1) fx = 0, sx = ssize.width - 1 (xmax = dx)
 *           alpha = {1.f - fx, fx}
 *   for dy: fy likewise, yofs = sy (unclamped), beta = {1.f - fy, fy}
 *   invoker: source rows clip(sy), clip(sy + 1);
 *     HResize: dx < xmax: D = S[sx] * a0 + S[sx + 1] * a1;  dx >= xmax: D = S[sx] * 1
 *     VResize: dst = D0 * b0 + D1 * b1
 */
int orc_resize_linear_f32(const float* src, int h, int w, int s, float* dst, int nthreads) {
  if (!src || !dst || h < 1 || w < 1 || s < 1) return -1;
  const int Wd = w * s, Hd = h * s;
  const double scale = 1.0 / ((double)Wd / (double)w);
  const double scale_y = 1.0 / ((double)Hd / (double)h);
  int* xofs = (int*)malloc(sizeof(int) * Wd);
  float* alpha = (float*)malloc(sizeof(float) * 2 * Wd);
  if (!xofs || !alpha) { free(xofs); free(alpha); return -3; }
  int xmax = Wd;
  for (int dx = 0; dx < Wd; ++dx) {
    float fx = (float)((dx + 0.5) * scale - 0.5);
    int sx = cv_floor_f(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx + 1 >= w) {
      if (dx < xmax) xmax = dx;
      if (sx >= w - 1) { fx = 0; sx = w - 1; }
    }
    xofs[dx] = sx;
    alpha[2 * dx] = 1.f - fx;
    alpha[2 * dx + 1] = fx;
  }
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
#endif
  {
    float* t0 = (float*)malloc(sizeof(float) * Wd);
    float* t1 = (float*)malloc(sizeof(float) * Wd);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int dy = 0; dy < Hd; ++dy) {
      float fy = (float)((dy + 0.5) * scale_y - 0.5);
      int sy = cv_floor_f(fy);
      fy -= sy;
      const float b0 = 1.f - fy, b1 = fy;
      const float* S0 = src + (size_t)clip_row(sy, h) * w;
      const float* S1 = src + (size_t)clip_row(sy + 1, h) * w;
      int dx = 0;
      for (; dx < xmax; ++dx) {
        const int sx = xofs[dx];
        const float a0 = alpha[2 * dx], a1 = alpha[2 * dx + 1];
        t0[dx] = S0[sx] * a0 + S0[sx + 1] * a1;
        t1[dx] = S1[sx] * a0 + S1[sx + 1] * a1;
      }
      for (; dx < Wd; ++dx) {
        const int sx = xofs[dx];
        t0[dx] = S0[sx] * 1.f;
        t1[dx] = S1[sx] * 1.f;
      }
      float* D = dst + (size_t)dy * Wd;
      for (dx = 0; dx < Wd; ++dx) D[dx] = t0[dx] * b0 + t1[dx] * b1;
    }
    free(t0);
    free(t1);
  }
  free(xofs);
  free(alpha);
  return 0;
}

/* BilinearTab_i of imgwarp.cpp initInterTab2D(INTER_LINEAR, fixpt = true): 32x32 entries of 4 shorts */
static short g_tab[32 * 32][4];
static int g_tab_ready = 0;

static void init_tab(void) {
  if (g_tab_ready) return;
  for (int fy = 0; fy < 32; ++fy)
    for (int fx = 0; fx < 32; ++fx) {
      const float tx[2] = {1.f - fx / 32.f, fx / 32.f};
      const float ty[2] = {1.f - fy / 32.f, fy / 32.f};
      short* it = g_tab[fy * 32 + fx];
      int isum = 0;
      for (int k1 = 0; k1 < 2; ++k1)
        for (int k2 = 0; k2 < 2; ++k2) {
          const float v = ty[k1] * tx[k2];
          int iv = cv_round_f(v * 32768.f);
          iv = iv > 32767 ? 32767 : iv; /* saturate_cast<short> */
          it[k1 * 2 + k2] = (short)iv;
          isum += iv;
        }
      if (isum != 32768) {
        /* OpenCV moves the rounding residue into the centre taps; for the 2x2 kernel the only
           in-range candidate is tap [1][1] (the fx = fy = 0 entry becomes {32767, 0, 0, 1}) */
        it[3] = (short)(it[3] - (isum - 32768));
      }
    }
  g_tab_ready = 1;
}

/*
 * imgwarp.cpp remap(), map1 = CV_32FC2, map2 empty, src CV_8UC{1,3,4}, BORDER_CONSTANT, value 0.
 *   linear : sx = cvRound(mx * INTER_TAB_SIZE) ...; XY = saturate<short>(sx >> 5); A = (sy & 31) * 32 + (sx & 31)
 *            remapBilinear: inlier (0 <= sx < W-1, 0 <= sy < H-1): FixedPtCast(sum S*w) = (sum + 2^14) >> 15
 *            wholly outside (sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0): 0;  else per-tap border test
 *   nearest: convertTo(CV_16S) = saturate<short>(cvRound(m)); remapNearest: inside -> S, else 0
 */
int orc_remap_u8(const uint8_t* src, int H, int W, int C, ptrdiff_t sstep, const float* map, int Hd,
                 int Wd, uint8_t* dst, ptrdiff_t dstep, int interp, int nthreads) {
  if (!src || !map || !dst || H < 1 || W < 1 || Hd < 1 || Wd < 1 || C < 1 || C > 4) return -1;
  if (interp != 0 && interp != 1) return -1;
  init_tab();
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int dy = 0; dy < Hd; ++dy) {
    const float* M = map + (size_t)dy * Wd * 2;
    uint8_t* D = dst + (ptrdiff_t)dy * dstep;
    for (int dx = 0; dx < Wd; ++dx, D += C) {
      if (interp == 0) {
        const int sx = sat_i16(cv_round_f(M[2 * dx])), sy = sat_i16(cv_round_f(M[2 * dx + 1]));
        if ((unsigned)sx < (unsigned)W && (unsigned)sy < (unsigned)H) {
          const uint8_t* S = src + (ptrdiff_t)sy * sstep + (ptrdiff_t)sx * C;
          for (int k = 0; k < C; ++k) D[k] = S[k];
        } else {
          for (int k = 0; k < C; ++k) D[k] = 0;
        }
        continue;
      }
      const int fsx = cv_round_f(M[2 * dx] * 32.f), fsy = cv_round_f(M[2 * dx + 1] * 32.f);
      const int sx = sat_i16(fsx >> 5), sy = sat_i16(fsy >> 5);
      const short* w = g_tab[(fsy & 31) * 32 + (fsx & 31)];
      const unsigned width1 = W - 1 > 0 ? W - 1 : 0, height1 = H - 1 > 0 ? H - 1 : 0;
      if ((unsigned)sx < width1 && (unsigned)sy < height1) {
        const uint8_t* S = src + (ptrdiff_t)sy * sstep + (ptrdiff_t)sx * C;
        for (int k = 0; k < C; ++k) {
          const int v = S[k] * w[0] + S[C + k] * w[1] + S[sstep + k] * w[2] + S[sstep + C + k] * w[3];
          D[k] = (uint8_t)((v + (1 << 14)) >> 15);
        }
      } else if (sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0) {
        for (int k = 0; k < C; ++k) D[k] = 0;
      } else {
        const int x0 = sx >= 0 && sx < W, x1 = sx + 1 >= 0 && sx + 1 < W;
        const int y0 = sy >= 0 && sy < H, y1 = sy + 1 >= 0 && sy + 1 < H;
        for (int k = 0; k < C; ++k) {
          const int v0 = (x0 && y0) ? src[(ptrdiff_t)sy * sstep + (ptrdiff_t)sx * C + k] : 0;
          const int v1 = (x1 && y0) ? src[(ptrdiff_t)sy * sstep + (ptrdiff_t)(sx + 1) * C + k] : 0;
          const int v2 = (x0 && y1) ? src[(ptrdiff_t)(sy + 1) * sstep + (ptrdiff_t)sx * C + k] : 0;
          const int v3 = (x1 && y1) ? src[(ptrdiff_t)(sy + 1) * sstep + (ptrdiff_t)(sx + 1) * C + k] : 0;
          const int v = v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3];
          D[k] = (uint8_t)((v + (1 << 14)) >> 15);
        }
      }
    }
  }
  return 0;
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
