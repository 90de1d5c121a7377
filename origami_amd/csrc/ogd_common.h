// Shared host-side plumbing of libogdewarp.so (error reporting, lazy device binding, scratch).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include "../../include/ogdewarp.h"

namespace ogd {

// thread-local message of the last failure (ogd_last_error)
void set_error(const char* fmt, ...);
int fail(int status, const char* fmt, ...);

// Lazily binds the calling thread to its device (first call per thread creates the context).
// Returns OGD_OK or a negative status.
int ensure_device();

#define OGD_HIP(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return ::ogd::fail(OGD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),  \
                         __FILE__, __LINE__);                                                 \
  } while (0)

#define OGD_REQUIRE(cond, ...)                                  \
  do {                                                          \
    if (!(cond)) return ::ogd::fail(OGD_ERR_INVALID, __VA_ARGS__); \
  } while (0)

#define OGD_TRY(expr)            \
  do {                           \
    int _s = (expr);             \
    if (_s != OGD_OK) return _s; \
  } while (0)

// Device memory of the host-buffer entry points comes from a per-thread cache of blocks (grow-only,
// best fit): a worker that dewarps page after page pays hipMalloc / hipFree once, not per call.
// ogd_trim() gives the calling thread's idle blocks back.
void* pool_take(size_t bytes);  // nullptr on failure (error already set)
void pool_give(void* p);
// blocks of the calling thread's cache that are handed out right now (for leak tests: ogd_pool_busy)
int pool_busy_count();
// A pinned host buffer of the calling thread for packing several small uploads into one asynchronous copy (a
// synchronous copy from pageable memory costs ~60 us whatever its size).  Valid until the thread's next call; the
// caller synchronises before it returns.  nullptr if pinned memory cannot be had: copy from where the data lies.
void* staging_host(size_t bytes);

// A device buffer held for the duration of one host-buffer entry point.
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) pool_give(p);
  }
  int alloc(size_t bytes) {
    if (p) pool_give(p);  // a second alloc on the same buffer must not strand the first block as busy for ever
    p = pool_take(bytes ? bytes : 16);
    return p ? OGD_OK : OGD_ERR_NOMEM;
  }
  template <class T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

// Host <-> device copies of images with a row stride: one linear copy when rows are packed.
inline hipError_t copy_rows(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width,
                            size_t rows, hipMemcpyKind kind) {
  if (dpitch == width && spitch == width) return hipMemcpy(dst, src, width * rows, kind);
  return hipMemcpy2D(dst, dpitch, src, spitch, width, rows, kind);
}

// ogd_jpeg_dec.hip: parse a scan file once and decode it into `dimg` (allocated here: H * W bytes, packed rows)
int jpeg_decode_gray_to(DevBuf& dimg, const uint8_t* file, size_t n, int* H, int* W, void* stream);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(OGD_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
  return OGD_OK;
}

}  // namespace ogd
