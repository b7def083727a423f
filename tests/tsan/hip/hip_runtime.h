// Stub device layer for the CPU-only ThreadSanitizer build of the library's HOST threading code (tests/test_tsan_cpu.py):
// "device" memory is host memory, streams and events complete at once, copies are memcpy.  Only what runtime.hip and
// staging.hip use.  Never part of the product and never used on a GPU box (sanitizers run on the CPU build only).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef struct tsan_stream_s* hipStream_t;
typedef struct tsan_event_s* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocPortable = 1,
       hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000, hipHostRegisterDefault = 0 };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };

inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "stub"; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
inline hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(1); return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)malloc(1); return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* a, int* b) { *a = 0; *b = -1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(1); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)malloc(1); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
    for (size_t i = 0; i < h; ++i) memcpy((char*)d + i * dp, (const char*)s + i * sp, w);
    return hipSuccess;
}
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemset2DAsync(void* d, size_t p, int v, size_t w, size_t h, hipStream_t) {
    for (size_t i = 0; i < h; ++i) memset((char*)d + i * p, v, w);
    return hipSuccess;
}
