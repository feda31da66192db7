// SILENT_HOST_ONLY: the host side of libsilent_hip (argument validation, tile / region / tap tables, row programs, walk plans,
// weight-stream packing, workspace layout, the host-pointer twins' staging) compiled WITHOUT a GPU behind it, for the CPU
// container's sanitizer run (pysilent_amd/csrc/build.py --host-asan -> lib/libsilent_hostonly_asan.so, tests/test_sanitizers.py).
//
// Not a product path and never shipped as one: every kernel launch is compiled out (outputs are whatever the "device" buffers
// held), device memory is host memory, streams and events are inert.  What it keeps is every line of host code of
// the six translation units (silent_unity.hip), run under -fsanitize=address,undefined with extents / crops / regions fuzzed from Python, plus two fault
// injectors for the exception barrier of the ABI:
//   silent_host_arm_fault(n)       the n-th NEED_CTX passed from now on throws std::bad_alloc (every entry point has one)
//   silent_host_fail_new_after(n)  the n-th operator new of this library from now on throws std::bad_alloc (vectors of the planners)
#pragma once
#ifdef SILENT_HOST_ONLY

#include <cstdlib>
#include <cstring>
#include <new>

namespace silent_host {
inline long& fault_countdown() {
    static long n = 0;
    return n;
}
inline long& new_countdown() {
    static long n = 0;
    return n;
}
inline void fault_point() {
    long& n = fault_countdown();
    if (n > 0 && --n == 0) throw std::bad_alloc();
}
inline hipError_t Malloc(void** p, size_t n) {
    *p = std::calloc(n ? n : 1, 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t Free(void* p) {
    std::free(p);
    return hipSuccess;
}
inline hipError_t Memcpy(void* d, const void* s, size_t n) {
    if (n) std::memcpy(d, s, n);
    return hipSuccess;
}
inline hipError_t Props(hipDeviceProp_t* prop) {
    std::memset((void*)prop, 0, sizeof(*prop));
    std::snprintf(prop->name, sizeof(prop->name), "host-only build (no GPU)");
    std::snprintf(prop->gcnArchName, sizeof(prop->gcnArchName), "gfx950:host-only");
    prop->multiProcessorCount = 256;
    return hipSuccess;
}
inline int& current_device() {
    static int d = 0;
    return d;
}
}  // namespace silent_host

// this library's own allocations (hidden visibility: other libraries of the process keep theirs)
void* operator new(size_t n) {
    long& c = silent_host::new_countdown();
    if (c > 0 && --c == 0) throw std::bad_alloc();
    void* p = std::malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void* operator new[](size_t n) { return operator new(n); }
void operator delete(void* p) noexcept { std::free(p); }
void operator delete[](void* p) noexcept { std::free(p); }
void operator delete(void* p, size_t) noexcept { std::free(p); }
void operator delete[](void* p, size_t) noexcept { std::free(p); }

// No device code object exists in this build (--offload-host-only): the registration calls the compiler emits into the module
// constructor bind to these (-Wl,-Bsymbolic), and the fat binary they would register is an empty placeholder (-cuid=silenthost
// fixes its name).
extern "C" {
__attribute__((used)) const char __hip_fatbin_silenthost[16] = {0};
void** __hipRegisterFatBinary(const void*) {
    static void* handle = nullptr;
    return &handle;
}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, char*, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void**) {}
}

#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(...) ((void)0)
#define hipMalloc(p, n) silent_host::Malloc((void**)(p), (n))
#define hipFree(p) silent_host::Free((void*)(p))
#define hipMemcpy(d, s, n, kind) silent_host::Memcpy((void*)(d), (const void*)(s), (n))
#define hipMemset(d, v, n) (std::memset((void*)(d), (v), (n)), hipSuccess)
#define hipMemcpyAsync(d, s, n, kind, stream) silent_host::Memcpy((void*)(d), (const void*)(s), (n))
#define hipStreamSynchronize(s) ((void)(s), hipSuccess)
#define hipStreamIsCapturing(s, st) ((void)(s), *(st) = hipStreamCaptureStatusNone, hipSuccess)
#define hipGetLastError() (hipSuccess)
#define hipGetDeviceCount(n) (*(n) = 1, hipSuccess)
#define hipGetDevice(d) (*(d) = silent_host::current_device(), hipSuccess)
#define hipSetDevice(d) (silent_host::current_device() = (d), hipSuccess)
#undef hipGetDeviceProperties
#define hipGetDeviceProperties(p, d) silent_host::Props(p)
#define hipOccupancyMaxActiveBlocksPerMultiprocessor(out, ...) (*(out) = 5, hipSuccess)
#define hipDeviceGetAttribute(v, a, d) (*(v) = 100000, hipSuccess)
#define hipHostMalloc(p, n, f) silent_host::Malloc((void**)(p), (n))
#define hipHostFree(p) silent_host::Free((void*)(p))
#define hipStreamCreateWithFlags(s, f) (*(s) = (hipStream_t)(size_t)8, hipSuccess)
#define hipStreamDestroy(s) ((void)(s), hipSuccess)
#define hipStreamBeginCapture(s, m) ((void)(s), hipSuccess)
#define hipStreamEndCapture(s, g) ((void)(s), *(g) = (hipGraph_t)(size_t)1, hipSuccess)
#define hipGraphInstantiate(e, g, a, b, c) (*(e) = (hipGraphExec_t)(size_t)1, hipSuccess)
#define hipGraphLaunch(e, s) ((void)(e), (void)(s), hipSuccess)
#define hipGraphDestroy(g) ((void)(g), hipSuccess)
#define hipGraphExecDestroy(e) ((void)(e), hipSuccess)
#define hipEventCreate(e) (*(e) = (hipEvent_t)(size_t)1, hipSuccess)
#define hipEventDestroy(e) ((void)(e), hipSuccess)
#define hipEventRecord(e, s) ((void)(e), (void)(s), hipSuccess)
#define hipEventSynchronize(e) ((void)(e), hipSuccess)
#define hipEventElapsedTime(ms, a, b) (*(ms) = 0.0f, hipSuccess)

#define SILENT_FAULT_POINT() silent_host::fault_point()
#else
#define SILENT_FAULT_POINT() ((void)0)
#endif
