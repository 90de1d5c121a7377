// Library plumbing: error strings, lazy per-thread device binding, raw device-memory helpers.
// No HIP call is made before the first entry point runs, so the library can be dlopen()ed before
// origami's Pool forks its workers (ref: origami/batch/core/processor.py:483).
#include "ogd_common.h"
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace ogd {

static thread_local char g_err[512] = "";
static thread_local int g_bound_device = -2;  // -2: not bound yet

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(int status, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return status;
}

static int bind_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(OGD_ERR_NODEVICE, "no HIP device visible (%s)",
                e == hipSuccess ? "count = 0" : hipGetErrorString(e));
  if (device < 0) {
    const char* env = getenv("OGD_DEVICE");
    device = env ? atoi(env) : -1;
  }
  if (device < 0) {
    // keep whatever device the hosting process already selected (e.g. torch.cuda.set_device)
    if (hipGetDevice(&device) != hipSuccess) device = 0;
  }
  if (device >= n) return fail(OGD_ERR_INVALID, "device %d out of range (%d visible)", device, n);
  // How a host thread waits for the device.  HIP's default spins; a stage worker waits for milliseconds at a time
  // (the march's walks are latency-bound launches), and with several workers per GPU on a box with two cores per
  // worker the spinning threads take the cores the other workers' host work needs.  OGD_SYNC = spin | yield | block
  // (hipDeviceSchedule*); must be set before the device's context exists, so the first binding thread decides.
  // Read ONCE per process, by the first thread that binds: HIP accepts hipSetDeviceFlags on a device whose context is
  // in use, and a process whose waiting mode was switched under way (a later thread binding after the variable had
  // been set for child processes) hung in hipFree (tests/test_gpu_stage.py, round 3).
  static std::once_flag sync_once;
  std::call_once(sync_once, [] {
    if (const char* mode = getenv("OGD_SYNC")) {
      const unsigned flags = !strcmp(mode, "block") ? hipDeviceScheduleBlockingSync
                             : !strcmp(mode, "yield") ? hipDeviceScheduleYield
                             : !strcmp(mode, "spin") ? hipDeviceScheduleSpin : hipDeviceScheduleAuto;
      if (hipSetDeviceFlags(flags) != hipSuccess) (void)hipGetLastError();  // context exists already: keep its mode
    }
  });
  OGD_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  OGD_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(OGD_ERR_NODEVICE, "device %d is %s; this library carries gfx950 code only", device,
                prop.gcnArchName);
  g_bound_device = device;
  return OGD_OK;
}

int ensure_device() {
  if (g_bound_device >= 0) {
    // the host may have switched devices on this thread since; re-assert ours
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == g_bound_device) return OGD_OK;
    OGD_HIP(hipSetDevice(g_bound_device));
    return OGD_OK;
  }
  return bind_device(-1);
}

// ---- per-thread device block cache ---------------------------------------------------------------
struct PoolBlock {
  void* p;
  size_t bytes;
  int device;
  bool busy;
};

struct Pool {
  std::vector<PoolBlock> blocks;
  ~Pool() {
    // thread exit: hand idle blocks back unless the HIP runtime is already gone
    for (auto& b : blocks)
      if (!b.busy && b.p) (void)hipFree(b.p);
  }
};
static thread_local Pool g_pool;

void* pool_take(size_t bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  PoolBlock* best = nullptr;
  for (auto& b : g_pool.blocks)
    if (!b.busy && b.device == dev && b.bytes >= bytes && (!best || b.bytes < best->bytes)) best = &b;
  // a block more than 4x too large stays free for a larger request (blocks come in 2 MiB granules: the allowance
  // lets small requests share them instead of each taking a new granule on every call)
  if (best && best->bytes <= 4 * bytes + (4u << 20)) {
    best->busy = true;
    return best->p;
  }
  const size_t rounded = (bytes + ((size_t)1 << 21) - 1) & ~(((size_t)1 << 21) - 1);  // 2 MiB granules
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, rounded);
  if (e != hipSuccess) {
    // out of memory: drop this thread's idle blocks and try once more
    for (auto& b : g_pool.blocks)
      if (!b.busy && b.p) {
        (void)hipFree(b.p);
        b.p = nullptr;
        b.bytes = 0;
      }
    e = hipMalloc(&p, rounded);
  }
  if (e != hipSuccess) {
    set_error("hipMalloc(%zu) failed: %s", rounded, hipGetErrorString(e));
    return nullptr;
  }
  for (auto& b : g_pool.blocks)
    if (!b.p) {
      b = PoolBlock{p, rounded, dev, true};
      return p;
    }
  g_pool.blocks.push_back(PoolBlock{p, rounded, dev, true});
  return p;
}

void pool_give(void* p) {
  for (auto& b : g_pool.blocks)
    if (b.p == p) {
      b.busy = false;
      return;
    }
  (void)hipFree(p);  // not ours (cannot happen through DevBuf)
}

int pool_busy_count() {
  int n = 0;
  for (auto& b : g_pool.blocks) n += b.p && b.busy;
  return n;
}

struct Staging {
  void* p = nullptr;
  size_t bytes = 0;
  ~Staging() {
    if (p) (void)hipHostFree(p);
  }
};
static thread_local Staging g_staging;

void* staging_host(size_t bytes) {
  if (bytes <= g_staging.bytes) return g_staging.p;
  if (g_staging.p) (void)hipHostFree(g_staging.p);
  g_staging.p = nullptr;
  g_staging.bytes = 0;
  const size_t rounded = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  void* p = nullptr;
  if (hipHostMalloc(&p, rounded, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  g_staging.p = p;
  g_staging.bytes = rounded;
  return p;
}

static size_t pool_trim() {
  size_t freed = 0;
  for (auto& b : g_pool.blocks)
    if (!b.busy && b.p) {
      (void)hipFree(b.p);
      freed += b.bytes;
      b.p = nullptr;
      b.bytes = 0;
    }
  return freed;
}

}  // namespace ogd

using namespace ogd;

extern "C" {

const char* ogd_last_error(void) { return g_err; }

int ogd_version(void) { return 100; }

int ogd_device_count(int* count) {
  OGD_REQUIRE(count != nullptr, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(OGD_ERR_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return OGD_OK;
}

int ogd_init(int device) { return bind_device(device); }

int ogd_device_pci_bus_id(int device, char* out, int cap) {
  OGD_REQUIRE(out != nullptr && cap >= 16, "out must hold at least 16 bytes");
  out[0] = 0;
  hipError_t e = hipDeviceGetPCIBusId(out, cap, device);  // no context is created
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(OGD_ERR_NODEVICE, "hipDeviceGetPCIBusId(%d): %s", device, hipGetErrorString(e));
  }
  return OGD_OK;
}

int ogd_trim(size_t* freed_bytes) {
  const size_t n = pool_trim();
  if (freed_bytes) *freed_bytes = n;
  return OGD_OK;
}

int ogd_pool_busy(int* blocks) {
  OGD_REQUIRE(blocks != nullptr, "blocks is NULL");
  *blocks = pool_busy_count();
  return OGD_OK;
}

int ogd_dev_alloc(void** dptr, size_t bytes) {
  OGD_REQUIRE(dptr != nullptr, "dptr is NULL");
  OGD_TRY(ensure_device());
  *dptr = nullptr;
  hipError_t e = hipMalloc(dptr, bytes ? bytes : 16);
  if (e != hipSuccess)
    return fail(OGD_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  return OGD_OK;
}

int ogd_dev_free(void* dptr) {
  if (!dptr) return OGD_OK;
  OGD_TRY(ensure_device());
  OGD_HIP(hipFree(dptr));
  return OGD_OK;
}

int ogd_memcpy_h2d(void* dptr, const void* hptr, size_t bytes, void* stream) {
  OGD_REQUIRE(dptr && hptr, "null pointer");
  OGD_TRY(ensure_device());
  OGD_HIP(hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, as_stream(stream)));
  return OGD_OK;
}

int ogd_memcpy_d2h(void* hptr, const void* dptr, size_t bytes, void* stream) {
  OGD_REQUIRE(dptr && hptr, "null pointer");
  OGD_TRY(ensure_device());
  OGD_HIP(hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
  return OGD_OK;
}

int ogd_stream_sync(void* stream) {
  OGD_TRY(ensure_device());
  OGD_HIP(hipStreamSynchronize(as_stream(stream)));
  return OGD_OK;
}

int ogd_memcpy2d_h2d(void* dptr, size_t dpitch, const void* hptr, size_t hpitch, size_t width, size_t rows,
                     void* stream) {
  OGD_REQUIRE(dptr && hptr, "null pointer");
  OGD_REQUIRE(dpitch >= width && hpitch >= width, "pitch shorter than a row");
  OGD_TRY(ensure_device());
  if (dpitch == width && hpitch == width)
    OGD_HIP(hipMemcpyAsync(dptr, hptr, width * rows, hipMemcpyHostToDevice, as_stream(stream)));
  else
    OGD_HIP(hipMemcpy2DAsync(dptr, dpitch, hptr, hpitch, width, rows, hipMemcpyHostToDevice, as_stream(stream)));
  return OGD_OK;
}

int ogd_memcpy2d_d2h(void* hptr, size_t hpitch, const void* dptr, size_t dpitch, size_t width, size_t rows,
                     void* stream) {
  OGD_REQUIRE(dptr && hptr, "null pointer");
  OGD_REQUIRE(dpitch >= width && hpitch >= width, "pitch shorter than a row");
  OGD_TRY(ensure_device());
  if (dpitch == width && hpitch == width)
    OGD_HIP(hipMemcpyAsync(hptr, dptr, width * rows, hipMemcpyDeviceToHost, as_stream(stream)));
  else
    OGD_HIP(hipMemcpy2DAsync(hptr, hpitch, dptr, dpitch, width, rows, hipMemcpyDeviceToHost, as_stream(stream)));
  return OGD_OK;
}

int ogd_mem_info(size_t* free_bytes, size_t* total_bytes) {
  OGD_TRY(ensure_device());
  size_t f = 0, t = 0;
  OGD_HIP(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return OGD_OK;
}

// ---- pinned host memory, streams, events: what a caller needs to overlap copies with kernels -----------

int ogd_host_alloc(void** hptr, size_t bytes) {
  OGD_REQUIRE(hptr != nullptr, "hptr is NULL");
  OGD_TRY(ensure_device());
  *hptr = nullptr;
  hipError_t e = hipHostMalloc(hptr, bytes ? bytes : 16, hipHostMallocDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(OGD_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  }
  return OGD_OK;
}

int ogd_host_free(void* hptr) {
  if (!hptr) return OGD_OK;
  OGD_TRY(ensure_device());
  OGD_HIP(hipHostFree(hptr));
  return OGD_OK;
}

int ogd_stream_create(void** stream) {
  OGD_REQUIRE(stream != nullptr, "stream is NULL");
  OGD_TRY(ensure_device());
  hipStream_t s = nullptr;
  // non-blocking: the host-buffer entry points work on the null stream and must not serialise a pipeline
  OGD_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = s;
  return OGD_OK;
}

int ogd_stream_destroy(void* stream) {
  if (!stream) return OGD_OK;
  OGD_TRY(ensure_device());
  OGD_HIP(hipStreamDestroy(as_stream(stream)));
  return OGD_OK;
}

int ogd_stream_wait_event(void* stream, void* event) {
  OGD_REQUIRE(event != nullptr, "event is NULL");
  OGD_TRY(ensure_device());
  OGD_HIP(hipStreamWaitEvent(as_stream(stream), reinterpret_cast<hipEvent_t>(event), 0));
  return OGD_OK;
}

int ogd_event_create(void** event, int timing) {
  OGD_REQUIRE(event != nullptr, "event is NULL");
  OGD_TRY(ensure_device());
  hipEvent_t e = nullptr;
  OGD_HIP(hipEventCreateWithFlags(&e, timing ? hipEventDefault : hipEventDisableTiming));
  *event = e;
  return OGD_OK;
}

int ogd_event_destroy(void* event) {
  if (!event) return OGD_OK;
  OGD_TRY(ensure_device());
  OGD_HIP(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
  return OGD_OK;
}

int ogd_event_record(void* event, void* stream) {
  OGD_REQUIRE(event != nullptr, "event is NULL");
  OGD_TRY(ensure_device());
  OGD_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(event), as_stream(stream)));
  return OGD_OK;
}

int ogd_event_sync(void* event) {
  OGD_REQUIRE(event != nullptr, "event is NULL");
  OGD_TRY(ensure_device());
  OGD_HIP(hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)));
  return OGD_OK;
}

int ogd_event_query(void* event, int* done) {
  OGD_REQUIRE(event != nullptr && done != nullptr, "null pointer");
  OGD_TRY(ensure_device());
  hipError_t e = hipEventQuery(reinterpret_cast<hipEvent_t>(event));
  if (e == hipErrorNotReady) {
    (void)hipGetLastError();
    *done = 0;
    return OGD_OK;
  }
  OGD_HIP(e);
  *done = 1;
  return OGD_OK;
}

int ogd_event_elapsed_ms(void* start, void* stop, float* ms) {
  OGD_REQUIRE(start && stop && ms, "null pointer");
  OGD_TRY(ensure_device());
  OGD_HIP(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
  return OGD_OK;
}

}  // extern "C"
