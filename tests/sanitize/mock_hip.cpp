// A stand-in for the HIP runtime, for SANITIZER builds of the host layer only
// (tests/test_host_sanitizers.py): ngp_api.hip and the launchers of ngp_kernels.hip are compiled
// host-only (hipcc --cuda-host-only) with -fsanitize=thread or address,undefined and linked against
// this file instead of libamdhip64.  "Device" memory is zeroed host memory, copies are memcpy,
// streams and events are inert handles, kernel launches do nothing.  What runs for real is every
// line of host code behind the C-ABI: the context mutex, the caching allocator, job / factor
// lifetimes, the staging code, the error paths — entered from several threads at once.
// Nothing here is part of the product; libngp.so never links it.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <unordered_map>

extern "C" {
typedef int hipError_t;
typedef struct ihipStream_t *hipStream_t;
typedef struct ihipEvent_t *hipEvent_t;
struct dim3_ { uint32_t x, y, z; };

static std::mutex g_mu;
static std::unordered_map<void *, size_t> g_live;      // allocation -> bytes
static std::atomic<long> g_launches{0}, g_bad_free{0}, g_oob{0}, g_sync_us{0};

static bool inside(const void *p, size_t n) {
    // host pointers (stack / heap of the caller) are not tracked: only check "device" ones
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &kv : g_live) {
        const char *b = (const char *)kv.first;
        if ((const char *)p >= b && (const char *)p < b + kv.second)
            return (const char *)p + n <= b + kv.second;
    }
    return true;
}

hipError_t hipGetDeviceCount(int *n) { *n = 1; return 0; }
hipError_t hipGetDevice(int *d) { *d = 0; return 0; }
hipError_t hipFuncSetAttribute(const void *, int, int) { return 0; }
hipError_t hipSetDevice(int) { return 0; }
hipError_t hipMemGetInfo(size_t *fr, size_t *tot) { *fr = *tot = (size_t)2 << 30; return 0; }
hipError_t hipMalloc(void **p, size_t n) {
    void *q = calloc(n ? n : 1, 1);
    if (!q) return 2;
    std::lock_guard<std::mutex> lk(g_mu);
    g_live[q] = n;
    *p = q;
    return 0;
}
// page-locked host memory: plain host memory here (the staging vectors of the jobs, ngp_api.hip
// PinnedPool — reachable from its process-lifetime pool, so not a leak)
hipError_t hipHostMalloc(void **p, size_t n, unsigned) {
    *p = malloc(n ? n : 1);
    return *p ? 0 : 2;
}
hipError_t hipHostFree(void *p) { free(p); return 0; }
hipError_t hipFree(void *p) {
    if (!p) return 0;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_live.erase(p)) { ++g_bad_free; return 1; }
    }
    free(p);
    return 0;
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, int) {
    if (!inside(d, n) || !inside(s, n)) { ++g_oob; return 1; }
    memcpy(d, s, n);
    return 0;
}
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int k, hipStream_t) { return hipMemcpy(d, s, n, k); }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) {
    if (!inside(d, n)) { ++g_oob; return 1; }
    memset(d, v, n);
    return 0;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)malloc(8); return 0; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return 0; }
// a "busy device": every synchronisation takes mock_hip_set_sync_delay_us microseconds, so that
// concurrent callers pile up behind the one that holds the context (combine_stress.cpp)
hipError_t hipStreamSynchronize(hipStream_t) {
    const long us = g_sync_us.load();
    if (us > 0) std::this_thread::sleep_for(std::chrono::microseconds(us));
    return 0;
}
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)malloc(8); return 0; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return 0; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return 0; }
hipError_t hipGetLastError(void) { return 0; }
const char *hipGetErrorString(hipError_t) { return "mock HIP runtime"; }
hipError_t hipLaunchKernel(const void *, dim3_, dim3_, void **, size_t, hipStream_t) { ++g_launches; return 0; }

// kernel<<<...>>> lowering
struct CallCfg { dim3_ g, b; size_t shm; hipStream_t s; };
static thread_local CallCfg t_cfg;
hipError_t __hipPushCallConfiguration(dim3_ g, dim3_ b, size_t shm, hipStream_t s) { t_cfg = {g, b, shm, s}; return 0; }
hipError_t __hipPopCallConfiguration(dim3_ *g, dim3_ *b, size_t *shm, hipStream_t *s) {
    *g = t_cfg.g; *b = t_cfg.b; *shm = t_cfg.shm; *s = t_cfg.s;
    return 0;
}
void **__hipRegisterFatBinary(const void *) { static void *h; return &h; }
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}

// for the driver
long mock_hip_launches(void) { return g_launches.load(); }
void mock_hip_set_sync_delay_us(long us) { g_sync_us.store(us); }
long mock_hip_live_allocations(void) { std::lock_guard<std::mutex> lk(g_mu); return (long)g_live.size(); }
long mock_hip_errors(void) { return g_bad_free.load() + g_oob.load(); }
}
