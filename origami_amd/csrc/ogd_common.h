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

// A device buffer owned for the duration of one host-buffer entry point.
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t bytes) {
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      p = nullptr;
      return fail(OGD_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    return OGD_OK;
  }
  template <class T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(OGD_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
  return OGD_OK;
}

}  // namespace ogd
