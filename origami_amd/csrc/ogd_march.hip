// N2 — the march of the node grid on the device.
// ref: origami/core/dewarp.py:108-140 (Field.get, estimate_extent), :328-339 (grid_shape),
//      :421-439 (grid_h), :441-493 (grid_hv), :176-263 (ShapelyBatchIntersections: nearest crossing of a
//      marching ray with the polyline of the next grid_h row).
//
// The reference (and the first build of this library) walks the grid on the host: one field query per Euler
// step, ~1450 round trips and ~650 numpy segment batches per broadsheet page (0.45 s).  The walk is
// sequential along one axis only: rows of grid_h are independent of each other, and so are the columns of
// grid_hv once grid_h exists.  Here each walk is ONE launch:
//   march_extent_kernel   one workgroup per field: all probe points step until every one has passed the far
//                         edge (the workgroup-wide "any" is the loop condition)                 [grid_shape]
//   march_rows_kernel     one lane per row, n_cols Euler steps                                  [grid_h]
//   march_row_ends_kernel the in-place extension of the first / last node of rows 1.. that the reference's
//                         numpy views cause (dewarp.py:205-211)                                  [make_row_slice]
//   march_rise_kernel     per column slice: max_dy over the rows -> the ray length              [compute_slice]
//   march_columns_kernel  one wavefront per column: per row one field query, then the lanes share the row's
//                         segments, test them in closed form (fp64) and reduce to the nearest hit [grid_hv]
// Arithmetic: positions are float32 and every step is a float32 multiply and add (numpy: f32 array * int,
// f32 += f32); angles come from the fp64 interpolant (ogd_tri_dev.h), are cast to float32 and go through
// cos / sin — numpy's float32 cos and this one (fp64 cos rounded to float32) may differ in the last bit, which
// moves a node by ~1e-6 px per step; the grid tolerance is 1e-3 px.  Segment intersections are the closed
// forms of origami_amd/dewarp.py:BatchIntersections (fp64, no FMA contraction).
#include "ogd_common.h"
#include "ogd_tri_dev.h"
#include <cmath>
#include <mutex>
#include <vector>

namespace ogd {

struct FieldView {
  TriView inner, extra;
  int has_inner, constant;
  double phi0;
  Bounds b;
};

static FieldView field_view(const ogd_tri* inner, const ogd_tri* extra, double phi0, const double* bounds) {
  FieldView f;
  f.constant = extra == nullptr;
  f.phi0 = phi0;
  f.has_inner = inner != nullptr;
  if (extra) {
    f.extra = view_of(extra);
    f.inner = inner ? view_of(inner) : f.extra;
  } else {
    f.extra = TriView{};
    f.inner = TriView{};
  }
  f.b = make_bounds(bounds);
  return f;
}

// Field.get (ref: dewarp.py:118-122): unit direction at a float32 position.
// `hint`: the triangles that answered the walk's previous point (start a walk with (-1, -1)).
__device__ __forceinline__ float2 field_dir(const FieldView& f, float px, float py, int2& hint) {
  double ang = f.phi0;
  if (!f.constant) eval_point_hint<1>(f.inner, f.has_inner, f.extra, f.b, (double)px, (double)py, &ang, hint);
  const float a = (float)ang;  // np.cos(angles, dtype=float32) casts first
  double sn, cs;
  sincos((double)a, &sn, &cs);  // one range reduction for both; a walk is a chain of these, nothing hides their latency
  return make_float2((float)cs, (float)sn);
}

__device__ __forceinline__ float2 euler_step(float2 p, float2 d, float len) {
  return make_float2(__fadd_rn(p.x, __fmul_rn(d.x, len)), __fadd_rn(p.y, __fmul_rn(d.y, len)));
}

// ---- grid_shape -----------------------------------------------------------------------------------------
constexpr int EXT_THREADS = 1024, EXT_MAX_PTS = 4096;

// block 0: field_h along x from starts (0, k*s), k*s < H, until all x >= W; block 1: field_v along y.
__global__ __launch_bounds__(EXT_THREADS) void march_extent_kernel(FieldView fh, FieldView fv, int W, int H, int s,
                                                                  int* __restrict__ n_steps_out) {
  __shared__ float2 pts[EXT_MAX_PTS];
  const int axis = blockIdx.x;  // 0: x, 1: y
  const FieldView& f = axis == 0 ? fh : fv;
  const int span = axis == 0 ? H : W;  // the starts run along the other axis
  const float limit = (float)(axis == 0 ? W : H);
  const int n = (span + s - 1) / s;
  const int max_steps = 2 * (1 + (axis == 0 ? W : H));
  for (int i = threadIdx.x; i < n; i += EXT_THREADS)
    pts[i] = axis == 0 ? make_float2(0.f, (float)(i * s)) : make_float2((float)(i * s), 0.f);
  int2 hints[EXT_MAX_PTS / EXT_THREADS];  // one per point this thread owns
#pragma unroll
  for (int k = 0; k < EXT_MAX_PTS / EXT_THREADS; ++k) hints[k] = make_int2(-1, -1);
  int n_steps = 1;
  while (true) {
    int any = 0;
    for (int i = threadIdx.x; i < n; i += EXT_THREADS) any |= (axis == 0 ? pts[i].x : pts[i].y) < limit;
    if (!__syncthreads_or(any) || n_steps >= max_steps) break;
#pragma unroll
    for (int slot = 0; slot < EXT_MAX_PTS / EXT_THREADS; ++slot) {  // each thread owns its points: no barrier needed
      const int i = threadIdx.x + slot * EXT_THREADS;
      if (i < n) {
        const float2 p = pts[i];
        pts[i] = euler_step(p, field_dir(f, p.x, p.y, hints[slot]), (float)s);
      }
    }
    ++n_steps;
  }
  if (threadIdx.x == 0) n_steps_out[axis] = n_steps;
}

// ---- grid_h ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void march_rows_kernel(FieldView fh, int n_rows, int n_cols, int s,
                                                        float2* __restrict__ gh) {
  const int gy = blockIdx.x * 64 + threadIdx.x;
  if (gy >= n_rows) return;
  float2 p = make_float2(0.f, (float)(gy * s));
  float2* row = gh + (size_t)gy * n_cols;
  int2 hint = make_int2(-1, -1);
  for (int gx = 0; gx < n_cols; ++gx) {
    row[gx] = p;
    p = euler_step(p, field_dir(fh, p.x, p.y, hint), (float)s);
  }
}

// extrapolate(a, b, x) of dewarp.py:164-167 in float32: b + x * (b - a) / |b - a|
__device__ __forceinline__ float2 extrapolate_f32(float2 a, float2 b, float x) {
  float vx = __fsub_rn(b.x, a.x), vy = __fsub_rn(b.y, a.y);
  const float nrm = __fsqrt_rn(__fadd_rn(__fmul_rn(vx, vx), __fmul_rn(vy, vy)));
  vx = __fdiv_rn(vx, nrm);
  vy = __fdiv_rn(vy, nrm);
  return make_float2(__fadd_rn(b.x, __fmul_rn(x, vx)), __fadd_rn(b.y, __fmul_rn(x, vy)));
}

__global__ __launch_bounds__(64) void march_row_ends_kernel(int n_rows, int n_cols, float large, int n_tail,
                                                            float2* __restrict__ gh) {
  const int gy = 1 + blockIdx.x * 64 + threadIdx.x;
  if (gy >= n_rows) return;
  float2* row = gh + (size_t)gy * n_cols;
  row[0] = extrapolate_f32(row[1], row[0], large);
  float2 last = row[n_cols - 1];
  const float2 prev = row[n_cols - 2];
  for (int k = 0; k < n_tail; ++k) last = extrapolate_f32(prev, last, large);
  row[n_cols - 1] = last;
}

// ---- grid_hv --------------------------------------------------------------------------------------------
// step_len[slice] = max(0, max_gy(max_sel gh[gy+1].y - min_sel gh[gy].y)) / cos(60 deg), float32.
__global__ __launch_bounds__(256) void march_rise_kernel(const float2* __restrict__ gh, int n_rows, int n_cols,
                                                         int slice_len, float* __restrict__ step_len) {
  __shared__ float red[256];
  const int c0 = blockIdx.x * slice_len, c1 = min(c0 + slice_len, n_cols);
  float best = 0.f;  // max(0, .)
  for (int gy = threadIdx.x; gy + 1 < n_rows; gy += 256) {
    float hi = -INFINITY, lo = INFINITY;
    for (int gx = c0; gx < c1; ++gx) {
      hi = fmaxf(hi, gh[(size_t)(gy + 1) * n_cols + gx].y);
      lo = fminf(lo, gh[(size_t)gy * n_cols + gx].y);
    }
    best = fmaxf(best, __fsub_rn(hi, lo));
  }
  red[threadIdx.x] = best;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + k]);
    __syncthreads();
  }
  // cos(60 * (pi / 180)) = 0.5000000000000001 is 0.5 in float32: the division is an exact doubling
  if (threadIdx.x == 0) step_len[blockIdx.x] = __fdiv_rn(red[0], 0.5f);
}

struct MarchStatus {
  int fail_col, fail_row;  // first column (lowest index wins) whose ray met no row
  float p0x, p0y, p1x, p1y;
};

// One wavefront per column.  status->fail_col must be INT_MAX at launch.
__global__ __launch_bounds__(256) void march_columns_kernel(FieldView fv, const float2* __restrict__ gh,
                                                            int n_rows, int n_cols, int slice_len,
                                                            const float* __restrict__ step_len,
                                                            float2* __restrict__ hv, MarchStatus* status) {
  const int gx = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gx >= n_cols) return;  // wave-uniform
  const int lane = threadIdx.x & 63;
  const float len = step_len[gx / slice_len];
  float2 p0 = gh[gx];  // row 0
  const int n_seg = n_cols - 1;
  int2 hint = make_int2(-1, -1);
  for (int gy = 0; gy + 1 < n_rows; ++gy) {
    if (lane == 0) hv[(size_t)gy * n_cols + gx] = p0;
    float2 p1 = euler_step(p0, field_dir(fv, p0.x, p0.y, hint), len);
    const double p0x = (double)p0.x, p0y = (double)p0.y;
    const double rx = (double)p1.x - p0x, ry = (double)p1.y - p0y;
    bool found = false;
    for (int row_i = gy + 1; row_i < n_rows && !found; ++row_i) {
      const float2* row = gh + (size_t)row_i * n_cols;
      double best_d = INFINITY, best_x = 0.0, best_y = 0.0;
      int best_j = INT_MAX;
      for (int j = lane; j < n_seg; j += 64) {
        const float2 a = row[j], b = row[j + 1];
        const double ax = (double)a.x, ay = (double)a.y;
        const double sx = (double)b.x - ax, sy = (double)b.y - ay;
        const double qx = ax - p0x, qy = ay - p0y;
        const double den = __dsub_rn(__dmul_rn(rx, sy), __dmul_rn(ry, sx));
        const double t = __ddiv_rn(__dsub_rn(__dmul_rn(qx, sy), __dmul_rn(qy, sx)), den);
        const double u = __ddiv_rn(__dsub_rn(__dmul_rn(qx, ry), __dmul_rn(qy, rx)), den);
        if (den != 0.0 && t >= 0.0 && t <= 1.0 && u >= 0.0 && u <= 1.0) {
          const double x = __dadd_rn(p0x, __dmul_rn(t, rx)), y = __dadd_rn(p0y, __dmul_rn(t, ry));
          const double ex = x - p0x, ey = y - p0y;
          const double d = sqrt(__dadd_rn(__dmul_rn(ex, ex), __dmul_rn(ey, ey)));
          if (d < best_d) {  // ascending j: the first of equal distances stays (np.argmin)
            best_d = d; best_x = x; best_y = y; best_j = j;
          }
        }
      }
      for (int m = 32; m > 0; m >>= 1) {
        const double od = __shfl_xor(best_d, m), ox = __shfl_xor(best_x, m), oy = __shfl_xor(best_y, m);
        const int oj = __shfl_xor(best_j, m);
        if (od < best_d || (od == best_d && oj < best_j)) {
          best_d = od; best_x = ox; best_y = oy; best_j = oj;
        }
      }
      if (best_j != INT_MAX) {  // wave-uniform after the butterfly
        p1 = make_float2((float)best_x, (float)best_y);
        found = true;
      }
    }
    if (!found) {
      if (lane == 0 && atomicMin(&status->fail_col, gx) > gx) {
        // (several failing columns may interleave these writes; the host reports fail_col / fail_row)
        status->fail_row = gy;
        status->p0x = p0.x; status->p0y = p0.y; status->p1x = p1.x; status->p1y = p1.y;
      }
      return;
    }
    p0 = p1;
  }
  if (lane == 0) hv[(size_t)(n_rows - 1) * n_cols + gx] = p0;
}

static int check_field(const ogd_tri* inner, const ogd_tri* extra, const char* what) {
  if (!extra) {
    OGD_REQUIRE(inner == nullptr, "%s: inner triangulation without an extra one", what);
    return OGD_OK;
  }
  OGD_REQUIRE(extra->nv == 1 && (!inner || inner->nv == 1), "%s: angle fields are scalar (nv = 1)", what);
  OGD_REQUIRE(!inner || inner->device == extra->device, "%s: inner/extra live on different devices", what);
  return OGD_OK;
}


// ---- the march proper, on device buffers the caller provides (the library's block cache or the fixed scratch) --
static int check_march_args(const ogd_tri* inner_h, const ogd_tri* extra_h, const ogd_tri* inner_v,
                            const ogd_tri* extra_v, int s, int n_rows, int n_cols, int num_slices,
                            const float* grid_h, const float* grid_hv) {
  OGD_REQUIRE(grid_h && grid_hv, "null pointer");
  OGD_REQUIRE(s >= 1 && n_rows >= 2 && n_cols >= 3, "bad grid %dx%d / cell %d (at least 2 rows, 3 columns)", n_rows,
              n_cols, s);
  OGD_REQUIRE((int64_t)n_rows * s < (1 << 24) && (int64_t)n_cols * s < (1 << 24), "grid too large for float32");
  OGD_REQUIRE(num_slices >= 1, "num_slices %d", num_slices);
  OGD_TRY(check_field(inner_h, extra_h, "field_h"));
  OGD_TRY(check_field(inner_v, extra_v, "field_v"));
  return OGD_OK;
}

// dgh, dhv: [n_rows][n_cols] float2; dlen: one float per column slice; dst: one MarchStatus.  All four are
// overwritten; nothing is allocated here.
static int march_grid_run(const FieldView& fh, const FieldView& fv, int s, int n_rows, int n_cols, int num_slices,
                          float2* dgh, float2* dhv, float* dlen, MarchStatus* dst, float* grid_h, float* grid_hv) {
  const size_t gb = (size_t)n_rows * n_cols * sizeof(float2);
  // make_slices(n, k): ceil(n / k) columns per slice (ref: dewarp.py:170-173)
  const int slice_len = (n_cols + num_slices - 1) / num_slices;
  const int n_slices = (n_cols + slice_len - 1) / slice_len;
  MarchStatus st0{INT_MAX, -1, 0.f, 0.f, 0.f, 0.f};
  OGD_HIP(hipMemcpy(dst, &st0, sizeof st0, hipMemcpyHostToDevice));
  OGD_HIP(hipMemsetAsync(dhv, 0, gb, nullptr));

  hipLaunchKernelGGL(march_rows_kernel, dim3((n_rows + 63) / 64), dim3(64), 0, nullptr, fh, n_rows, n_cols, s, dgh);
  OGD_TRY(check_launch("march_rows_kernel"));
  // the reference asserts on grid_h before it is modified (ref: dewarp.py:437); the row-end extension keeps
  // NaNs NaN, so the check below on the final array is equivalent
  // pieces of k = 3 nodes, stride k - 1: every piece that reaches the row end extends it once more
  int n_tail = 0;
  for (int i = 0; i < n_cols; i += 2)
    if (i != 0 && i + 3 >= n_cols) ++n_tail;
  hipLaunchKernelGGL(march_row_ends_kernel, dim3((n_rows + 62) / 64), dim3(64), 0, nullptr, n_rows, n_cols,
                     (float)(n_cols * s), n_tail, dgh);
  OGD_TRY(check_launch("march_row_ends_kernel"));
  hipLaunchKernelGGL(march_rise_kernel, dim3(n_slices), dim3(256), 0, nullptr, dgh, n_rows, n_cols, slice_len, dlen);
  OGD_TRY(check_launch("march_rise_kernel"));
  hipLaunchKernelGGL(march_columns_kernel, dim3((n_cols + 3) / 4), dim3(256), 0, nullptr, fv, dgh, n_rows, n_cols,
                     slice_len, dlen, dhv, dst);
  OGD_TRY(check_launch("march_columns_kernel"));
  MarchStatus st;
  OGD_HIP(hipMemcpy(&st, dst, sizeof st, hipMemcpyDeviceToHost));
  OGD_HIP(hipMemcpy(grid_h, dgh, gb, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < (size_t)n_rows * n_cols * 2; ++i)
    if (std::isnan(grid_h[i])) return fail(OGD_ERR_ASSERT, "NaN in grid_h (node %zu)", i / 2);
  if (st.fail_col != INT_MAX)
    return fail(OGD_ERR_RUNTIME, "failed to find intersection for i=%d, n=%d, [%g %g] to [%g %g].",
                st.fail_col % slice_len, std::min(slice_len, n_cols - st.fail_col / slice_len * slice_len),
                (double)st.p0x, (double)st.p0y, (double)st.p1x, (double)st.p1y);
  OGD_HIP(hipMemcpy(grid_hv, dhv, gb, hipMemcpyDeviceToHost));
  return OGD_OK;
}

// ---- ogd_march_grid_noalloc: the same march for callers that cannot allocate ---------------------------------
// Its device buffers (both node grids, the slice ray lengths, the status record) live in ONE file-scope device
// array that every GPU the library runs on holds from module load to process end (~4 MiB).  `scratch` below is
// the host-side stand-in of that array, NOT a device address: host code only ever uses the per-device address
// hipGetSymbolAddress returns for it (noalloc_scratch), and the kernels take that address as an argument
// exactly as they take the block-cache buffers in ogd_march_grid.
constexpr int NOALLOC_MAX_NODES = 1 << 18;  // n_rows * n_cols; a 6000 x 8000 page at cell 25 has ~80 k
constexpr int NOALLOC_MAX_SLICES = 1024;
constexpr int NOALLOC_MAX_DEVICES = 64;

struct MarchScratch {
  float2 gh[NOALLOC_MAX_NODES];
  float2 hv[NOALLOC_MAX_NODES];
  float step_len[NOALLOC_MAX_SLICES];
  MarchStatus status;
};

static __device__ __align__(256) unsigned char scratch[sizeof(MarchScratch)];

static std::mutex g_scratch_mutex;
static MarchScratch* g_scratch_ptr[NOALLOC_MAX_DEVICES];  // per device: each GPU has its own copy of the array

// The calling thread's current device must be the one to run on (ensure_device() has seen to that).
static int noalloc_scratch(MarchScratch** out) {
  int dev = -1;
  OGD_HIP(hipGetDevice(&dev));
  OGD_REQUIRE(dev >= 0 && dev < NOALLOC_MAX_DEVICES, "device %d beyond the %d this entry point keeps scratch for", dev,
              NOALLOC_MAX_DEVICES);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    OGD_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail(OGD_ERR_HIP, "hipGetSymbolAddress(scratch) returned a null address on device %d", dev);
    g_scratch_ptr[dev] = static_cast<MarchScratch*>(p);
  }
  *out = g_scratch_ptr[dev];
  return OGD_OK;
}

}  // namespace ogd

using namespace ogd;

extern "C" {

int ogd_march_shape(const ogd_tri* inner_h, const ogd_tri* extra_h, double phi0_h, const ogd_tri* inner_v,
                    const ogd_tri* extra_v, double phi0_v, const double* bounds, int W, int H, int s,
                    int* n_rows, int* n_cols) {
  OGD_REQUIRE(n_rows && n_cols, "null pointer");
  OGD_REQUIRE(W >= 1 && H >= 1 && s >= 1, "bad page %dx%d / cell %d", W, H, s);
  OGD_REQUIRE(W < (1 << 24) && H < (1 << 24), "page %dx%d too large for float32 positions", W, H);
  OGD_TRY(check_field(inner_h, extra_h, "field_h"));
  OGD_TRY(check_field(inner_v, extra_v, "field_v"));
  OGD_REQUIRE((H + s - 1) / s <= EXT_MAX_PTS && (W + s - 1) / s <= EXT_MAX_PTS,
              "more than %d probe points per axis (page %dx%d, cell %d)", EXT_MAX_PTS, W, H, s);
  OGD_TRY(ensure_device());
  DevBuf d;
  OGD_TRY(d.alloc(2 * sizeof(int)));
  const FieldView fh = field_view(inner_h, extra_h, phi0_h, bounds), fv = field_view(inner_v, extra_v, phi0_v, bounds);
  hipLaunchKernelGGL(march_extent_kernel, dim3(2), dim3(EXT_THREADS), 0, nullptr, fh, fv, W, H, s, d.as<int>());
  OGD_TRY(check_launch("march_extent_kernel"));
  int steps[2];
  OGD_HIP(hipMemcpy(steps, d.p, sizeof steps, hipMemcpyDeviceToHost));
  // estimate_extent raises when the walk does not leave the page (ref: dewarp.py:138-139)
  if (steps[0] >= 2 * (1 + W)) return fail(OGD_ERR_RUNTIME, "n_steps exceeded %d", 2 * (1 + W));
  if (steps[1] >= 2 * (1 + H)) return fail(OGD_ERR_RUNTIME, "n_steps exceeded %d", 2 * (1 + H));
  *n_cols = steps[0];  // est_width:  steps of field_h along x
  *n_rows = steps[1];  // est_height: steps of field_v along y
  return OGD_OK;
}

int ogd_march_grid(const ogd_tri* inner_h, const ogd_tri* extra_h, double phi0_h, const ogd_tri* inner_v,
                   const ogd_tri* extra_v, double phi0_v, const double* bounds, int s, int n_rows, int n_cols,
                   int num_slices, float* grid_h, float* grid_hv) {
  OGD_TRY(check_march_args(inner_h, extra_h, inner_v, extra_v, s, n_rows, n_cols, num_slices, grid_h, grid_hv));
  OGD_TRY(ensure_device());
  const size_t gb = (size_t)n_rows * n_cols * sizeof(float2);
  const int slice_len = (n_cols + num_slices - 1) / num_slices;
  const int n_slices = (n_cols + slice_len - 1) / slice_len;
  DevBuf dgh, dhv, dlen, dst;
  OGD_TRY(dgh.alloc(gb));
  OGD_TRY(dhv.alloc(gb));
  OGD_TRY(dlen.alloc(n_slices * sizeof(float)));
  OGD_TRY(dst.alloc(sizeof(MarchStatus)));
  return march_grid_run(field_view(inner_h, extra_h, phi0_h, bounds), field_view(inner_v, extra_v, phi0_v, bounds), s,
                        n_rows, n_cols, num_slices, dgh.as<float2>(), dhv.as<float2>(), dlen.as<float>(),
                        dst.as<MarchStatus>(), grid_h, grid_hv);
}

int ogd_march_grid_noalloc(const ogd_tri* inner_h, const ogd_tri* extra_h, double phi0_h, const ogd_tri* inner_v,
                           const ogd_tri* extra_v, double phi0_v, const double* bounds, int s, int n_rows,
                           int n_cols, int num_slices, float* grid_h, float* grid_hv) {
  OGD_TRY(check_march_args(inner_h, extra_h, inner_v, extra_v, s, n_rows, n_cols, num_slices, grid_h, grid_hv));
  OGD_REQUIRE((int64_t)n_rows * n_cols <= NOALLOC_MAX_NODES,
              "ogd_march_grid_noalloc: %d x %d nodes exceed the fixed scratch of %d nodes; use ogd_march_grid", n_rows,
              n_cols, NOALLOC_MAX_NODES);
  const int slice_len = (n_cols + num_slices - 1) / num_slices;
  OGD_REQUIRE((n_cols + slice_len - 1) / slice_len <= NOALLOC_MAX_SLICES,
              "ogd_march_grid_noalloc: more than %d column slices; use ogd_march_grid", NOALLOC_MAX_SLICES);
  OGD_TRY(ensure_device());
  MarchScratch* sc = nullptr;
  OGD_TRY(noalloc_scratch(&sc));
  return march_grid_run(field_view(inner_h, extra_h, phi0_h, bounds), field_view(inner_v, extra_v, phi0_v, bounds), s,
                        n_rows, n_cols, num_slices, sc->gh, sc->hv, sc->step_len, &sc->status, grid_h, grid_hv);
}

}  // extern "C"
