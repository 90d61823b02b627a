// sim_hip.cpp -- TEST INFRASTRUCTURE ONLY.  Host-memory stand-ins for the HIP runtime calls engine.cpp makes, so that
// the engine's HOST logic (input store, planning, windows and rings, sharding and its exchange schedule) can be
// exercised on a machine without a GPU: "device" memory is malloc'ed host memory, streams run synchronously.
// Linked only into tests/cpp/_build/libfr_simengine.so (tests/sim_tools.py); the product library links the real
// libamdhip64 and has no CPU path.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>

struct ihipStream_t { int dummy; };
struct ihipEvent_t { std::chrono::steady_clock::time_point t; };

extern "C" {

static std::atomic<uint64_t> fr_sim_live_bytes_v{0};
std::atomic<uint64_t> fr_sim_allocs{0}, fr_sim_frees{0};

const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : (e == hipErrorOutOfMemory ? "out of memory" : "simulated HIP error"); }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    std::memset(p, 0, sizeof *p);
    std::strcpy(p->gcnArchName, "gfx950:sim");
    return hipSuccess;
}
// every block is prefixed by its size so that live bytes can be tracked
hipError_t hipMalloc(void **p, size_t n) {
    char *raw = (char *)std::malloc((n ? n : 1) + 16);
    if (!raw) return hipErrorOutOfMemory;
    *(size_t *)raw = n;
    *p = raw + 16;
    std::memset(*p, 0xA5, n);   // uninitialised device memory is not zero: make a read of it show
    ++fr_sim_allocs;
    fr_sim_live_bytes_v += n;
    return hipSuccess;
}
hipError_t hipFree(void *p) {
    if (!p) return hipSuccess;
    char *raw = (char *)p - 16;
    fr_sim_live_bytes_v -= *(size_t *)raw;
    ++fr_sim_frees;
    std::free(raw);
    return hipSuccess;
}
uint64_t fr_sim_live_bytes(void) { return fr_sim_live_bytes_v.load(); }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t) { if (n) std::memmove(dst, src, n); return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) { if (n) std::memmove(dst, src, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t) { if (n) std::memset(dst, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new ihipStream_t{0}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new ihipEvent_t{}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new ihipEvent_t{}; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}

}  // extern "C"

extern "C" {
hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void *) { return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -1; return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = new ihipStream_t{0}; return hipSuccess; }
}
