// K2 — remap of uint8 pages / label maps through the dense sampling grid, and the fused
// grid-fill + remap.  ref: origami/core/dewarp.py:608-627 (cv2.remap, BORDER_CONSTANT 0).
//
// Arithmetic restated from OpenCV 4.2 imgproc remap (CV_32FC2 map, CV_8U source):
//   linear : sx = cvRound(mx * 32), sy = cvRound(my * 32)        (fp32 product, round-half-even;
//            out of int range / NaN -> INT_MIN as cvtss2si does)
//            ix = sat_i16(sx >> 5), fx = sx & 31 (same for y)
//            dst = (sum_ij w_ij * P(ix+i, iy+j) + 2^14) >> 15, w_ij = 32 * a_i * b_j,
//            a = (32 - fx, fx), b = (32 - fy, fy); P = 0 outside the image.
//            (all w_ij are multiples of 32, so this equals (sum a_i b_j P + 512) >> 10; the table
//             entry fx = fy = 0 is stored as {32767, 0, 0, 1}, which gives the same byte.)
//   nearest: ix = sat_i16(cvRound(mx)), iy likewise; dst = P(ix, iy).
//
// HBM traffic per output pixel: 8 B of map read + C B written + ~C B of source.  K2 proper (materialised
// map) is the LDS-tiled remap_u8_wave_kernel of ogd_remap_lds.h; the fused kernel below recomputes the
// map from the node grid inside the same kernel (MapNodes).
#include "ogd_common.h"
#include "ogd_resize_math.h"
#include "ogd_remap_math.h"
#include "ogd_remap_lds.h"

namespace ogd {

static int check_image(int H, int W, int C, ptrdiff_t stride, const char* what) {
  OGD_REQUIRE(H >= 1 && W >= 1, "%s: empty image (%dx%d)", what, W, H);
  OGD_REQUIRE(H < 32768 && W < 32768, "%s: %dx%d exceeds the int16 coordinate range of remap", what, W, H);
  OGD_REQUIRE(C == 1 || C == 3 || C == 4, "%s: %d channels not supported (1, 3 or 4)", what, C);
  OGD_REQUIRE(stride >= (ptrdiff_t)W * C, "%s: row stride %td < %d bytes", what, stride, W * C);
  OGD_REQUIRE(stride < (ptrdiff_t)1 << 24, "%s: row stride %td too large (the kernels address rows with 24-bit multiplies)", what, stride);
  return OGD_OK;
}

// Launch shapes (measured, tools/kbench): K2 proper streams the map, and 64 x 8 tiles keep its row
// segments long (512 B per wavefront and row); the fused kernel streams nothing, and a square 32 x 32 tile
// has the smaller source window per pixel and fewer x taps / lerped node rows to build per pixel.
// LDSW_MC: LDS bytes per wavefront for 3 / 4 channels (6 KB: six wavefronts per SIMD; the 32 x 32 tile of the
// fused kernel needs 8 KB for its ~40-row window).
// Multi-channel K2 (RGB / RGBA pages, configs[3]): experiment switches, see tools/build_variant.sh
#ifndef K2MC_LX
#define K2MC_LX 16
#endif
#ifndef K2MC_RPT
#define K2MC_RPT 2
#endif
#ifndef K2MC_LDSW
#define K2MC_LDSW 6144
#endif
template <class MapSrc, int C> struct TileShape { static constexpr int LX = 16, RPT = 2, LDSW = 4096; };
#ifndef K2C3_LDSW
#define K2C3_LDSW 8192  // with RL_SHEAR3 only: the sheared three-channel window needs a 504-byte pitch
#endif
template <> struct TileShape<MapGlobal, 3> { static constexpr int LX = K2MC_LX, RPT = K2MC_RPT, LDSW = RL_SHEAR3 ? K2C3_LDSW : K2MC_LDSW; };
template <> struct TileShape<MapGlobal, 4> { static constexpr int LX = K2MC_LX, RPT = K2MC_RPT, LDSW = K2MC_LDSW; };
#ifndef FUSED_LX
#define FUSED_LX 8
#endif
#ifndef FUSED_RPT
#define FUSED_RPT 4
#endif
#ifndef FUSED_LDSW1
#define FUSED_LDSW1 4096
#endif
template <int C> struct TileShape<MapNodes, C> { static constexpr int LX = FUSED_LX, RPT = FUSED_RPT, LDSW = C == 1 ? FUSED_LDSW1 : 8192; };
template <int C> struct TileShape<MapNodesUnit, C> : TileShape<MapNodes, C> {};

template <int C, class MapSrc>
static void launch_remap(int interp, hipStream_t st, const uint8_t* src, int H, int W, ptrdiff_t ss,
                         const MapSrc& ms, int Hd, int Wd, uint8_t* dst, ptrdiff_t ds) {
  constexpr int LX = TileShape<MapSrc, C>::LX, RPT = TileShape<MapSrc, C>::RPT;
  constexpr int TW = 4 * LX, TH = (64 / LX) * RPT;
  constexpr int LDSW = TileShape<MapSrc, C>::LDSW;  // source window budget per wavefront
  // sheared tiles (ogd_remap_lds.h:rl_shear): rows begin up to `reach` columns left of their tile's nominal origin
  int step = 0, wrap = 0, bias = 0, reach = 0;
  if (MapSrc::SHEAR && rl_shear_channels(C)) {
    step = (int)((((ds % 64) + 64) % 64) * RL_SHEAR_INV(C) % 64);
    if (step >= 32) step -= 64;
    wrap = std::abs(step) * (TH - 1) >= 64;
    bias = (wrap || step < 0) ? 64 : 0;
    reach = 63 + bias + (wrap ? 31 : std::max(0, step * (TH - 1)));
  }
  const int nbx = (Wd + reach + TW - 1) / TW, nby = (Hd + TH - 1) / TH, nb = nbx * nby;
  const dim3 grid((nb + RL_WPB - 1) / RL_WPB), block(64 * RL_WPB);
  if (interp == OGD_INTER_LINEAR)
    hipLaunchKernelGGL((remap_u8_wave_kernel<C, OGD_INTER_LINEAR, LX, RPT, LDSW, MapSrc>), grid, block,
                       0, st, src, H, W, ss, ms, Hd, Wd, dst, ds, nbx, nb, step, wrap, bias);
  else
    hipLaunchKernelGGL((remap_u8_wave_kernel<C, OGD_INTER_NEAREST, LX, RPT, LDSW, MapSrc>), grid, block,
                       0, st, src, H, W, ss, ms, Hd, Wd, dst, ds, nbx, nb, step, wrap, bias);
}

template <class MapSrc>
static void launch_remap_c(int C, int interp, hipStream_t st, const uint8_t* src, int H, int W, ptrdiff_t ss,
                           const MapSrc& ms, int Hd, int Wd, uint8_t* dst, ptrdiff_t ds) {
  switch (C) {
    case 1: launch_remap<1>(interp, st, src, H, W, ss, ms, Hd, Wd, dst, ds); break;
    case 3: launch_remap<3>(interp, st, src, H, W, ss, ms, Hd, Wd, dst, ds); break;
    default: launch_remap<4>(interp, st, src, H, W, ss, ms, Hd, Wd, dst, ds); break;
  }
}

}  // namespace ogd

using namespace ogd;

extern "C" {

int ogd_remap_u8_dev(const uint8_t* d_src, int H, int W, int C, ptrdiff_t src_stride,
                     const float* d_map_xy, int Hd, int Wd, uint8_t* d_dst, ptrdiff_t dst_stride,
                     int interp, void* stream) {
  OGD_TRY(check_image(H, W, C, src_stride, "src"));
  OGD_TRY(check_image(Hd, Wd, C, dst_stride, "dst"));
  OGD_REQUIRE(d_src && d_map_xy && d_dst, "null pointer");
  OGD_REQUIRE(interp == OGD_INTER_NEAREST || interp == OGD_INTER_LINEAR, "interp %d not supported", interp);
  OGD_TRY(ensure_device());
  const MapGlobal ms{d_map_xy, Hd, Wd};
  launch_remap_c(C, interp, as_stream(stream), d_src, H, W, src_stride, ms, Hd, Wd, d_dst, dst_stride);
  return check_launch("remap_u8_wave_kernel");
}

int ogd_dewarp_u8_dev(const uint8_t* d_src, int H, int W, int C, ptrdiff_t src_stride,
                      const float* d_nodes, int h, int w, int s, float sx, float sy, uint8_t* d_dst,
                      ptrdiff_t dst_stride, int interp, void* stream) {
  OGD_REQUIRE(h >= 1 && w >= 1 && s >= 1, "bad node grid %dx%d cell %d", h, w, s);
  OGD_REQUIRE((int64_t)h * s < 32768 && (int64_t)w * s < 32768, "dense grid too large");
  const int Hd = h * s, Wd = w * s;
  OGD_TRY(check_image(H, W, C, src_stride, "src"));
  OGD_TRY(check_image(Hd, Wd, C, dst_stride, "dst"));
  OGD_REQUIRE(d_src && d_nodes && d_dst, "null pointer");
  OGD_REQUIRE(interp == OGD_INTER_NEAREST || interp == OGD_INTER_LINEAR, "interp %d not supported", interp);
  OGD_TRY(ensure_device());
  if (sx == 1.f && sy == 1.f) {
    const MapNodesUnit ms{reinterpret_cast<const float2*>(d_nodes), h, w, 1.0 / (double)s, sx, sy, Hd, Wd};
    launch_remap_c(C, interp, as_stream(stream), d_src, H, W, src_stride, ms, Hd, Wd, d_dst, dst_stride);
  } else {
    const MapNodes ms{reinterpret_cast<const float2*>(d_nodes), h, w, 1.0 / (double)s, sx, sy, Hd, Wd};
    launch_remap_c(C, interp, as_stream(stream), d_src, H, W, src_stride, ms, Hd, Wd, d_dst, dst_stride);
  }
  return check_launch("remap_u8_wave_kernel<MapNodes>");
}

int ogd_remap_u8(const uint8_t* src, int H, int W, int C, ptrdiff_t src_stride, const float* map_xy,
                 int Hd, int Wd, uint8_t* dst, ptrdiff_t dst_stride, int interp) {
  OGD_TRY(check_image(H, W, C, src_stride, "src"));
  OGD_TRY(check_image(Hd, Wd, C, dst_stride, "dst"));
  OGD_REQUIRE(src && map_xy && dst, "null pointer");
  OGD_TRY(ensure_device());
  const size_t sb = (size_t)src_stride * H, mb = (size_t)Hd * Wd * 2 * sizeof(float);
  const size_t drow = (size_t)Wd * C;
  DevBuf dsrc, dmap, ddst;
  OGD_TRY(dsrc.alloc(sb));
  OGD_TRY(dmap.alloc(mb));
  OGD_TRY(ddst.alloc(drow * Hd));
  // the last source row may be shorter than the stride in the caller's buffer
  OGD_HIP(copy_rows(dsrc.p, src_stride, src, src_stride, (size_t)W * C, H, hipMemcpyHostToDevice));
  OGD_HIP(hipMemcpy(dmap.p, map_xy, mb, hipMemcpyHostToDevice));
  OGD_TRY(ogd_remap_u8_dev(dsrc.as<uint8_t>(), H, W, C, src_stride, dmap.as<float>(), Hd, Wd,
                           ddst.as<uint8_t>(), (ptrdiff_t)drow, interp, nullptr));
  OGD_HIP(copy_rows(dst, dst_stride, ddst.p, drow, drow, Hd, hipMemcpyDeviceToHost));
  return OGD_OK;
}

int ogd_dewarp_u8(const uint8_t* src, int H, int W, int C, ptrdiff_t src_stride, const float* nodes,
                  int h, int w, int s, float sx, float sy, uint8_t* dst, ptrdiff_t dst_stride,
                  int interp) {
  OGD_REQUIRE(h >= 1 && w >= 1 && s >= 1, "bad node grid %dx%d cell %d", h, w, s);
  OGD_REQUIRE((int64_t)h * s < 32768 && (int64_t)w * s < 32768, "dense grid too large");
  const int Hd = h * s, Wd = w * s;
  OGD_TRY(check_image(H, W, C, src_stride, "src"));
  OGD_TRY(check_image(Hd, Wd, C, dst_stride, "dst"));
  OGD_REQUIRE(src && nodes && dst, "null pointer");
  OGD_TRY(ensure_device());
  const size_t sb = (size_t)src_stride * H, nb = (size_t)h * w * 2 * sizeof(float);
  const size_t drow = (size_t)Wd * C;
  DevBuf dsrc, dn, ddst;
  OGD_TRY(dsrc.alloc(sb));
  OGD_TRY(dn.alloc(nb));
  OGD_TRY(ddst.alloc(drow * Hd));
  OGD_HIP(copy_rows(dsrc.p, src_stride, src, src_stride, (size_t)W * C, H, hipMemcpyHostToDevice));
  OGD_HIP(hipMemcpy(dn.p, nodes, nb, hipMemcpyHostToDevice));
  OGD_TRY(ogd_dewarp_u8_dev(dsrc.as<uint8_t>(), H, W, C, src_stride, dn.as<float>(), h, w, s, sx, sy,
                            ddst.as<uint8_t>(), (ptrdiff_t)drow, interp, nullptr));
  OGD_HIP(copy_rows(dst, dst_stride, ddst.p, drow, drow, Hd, hipMemcpyDeviceToHost));
  return OGD_OK;
}

}  // extern "C"
