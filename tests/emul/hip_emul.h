// tests/emul/hip_emul.h — TEST INFRASTRUCTURE, not part of the product.
// A minimal host stand-in for the HIP runtime and the few device builtins that rtmi_device.hpp / rtmi.hip use, so that
// the scene layout code and the lane-independent kernels (trace_kernel, camera_rays_kernel) can be compiled with g++
// and run one "lane" at a time under AddressSanitizer / UBSan before a changed kernel ever touches a GPU.
// "Device" memory is plain malloc memory, so an out-of-bounds scene access is reported by ASan with a stack trace.
// The wave-cooperative kernels (render / resample) are not emulated.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define __align__(n) alignas(n)
#define HIP_SYMBOL(x) x

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

struct float4 { float x, y, z, w; };

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorEmul = 1 };
typedef void *hipStream_t;
typedef void *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost };
struct hipDeviceProp_t { char name[64]; char gcnArchName[64]; int multiProcessorCount; int clockRate; };

inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorEmul; }
template <typename T> inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyPeer(void *d, int, const void *s, int, size_t n) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
enum { hipHostMallocDefault = 0, hipStreamNonBlocking = 1 };
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorEmul; }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
template <typename T> inline hipError_t hipGetSymbolAddress(void **p, const T &symbol) { *p = (void *)&symbol; return hipSuccess; } // a "device" array is a host array here
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    memset(p, 0, sizeof *p); strcpy(p->name, "host emulation"); strcpy(p->gcnArchName, "none"); p->multiProcessorCount = 1; return hipSuccess;
}
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline const char *hipGetErrorString(hipError_t) { return "emulated error"; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return hipSuccess; }
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }

// one lane after the other: valid for kernels whose lanes do not communicate
template <typename K, typename... A> inline void emul_launch(K kernel, dim3 grid, dim3 block, A... args) {
    gridDim = grid; blockDim = block;
    for (unsigned b = 0; b < grid.x; ++b)
        for (unsigned t = 0; t < block.x; ++t) { blockIdx = dim3(b); threadIdx = dim3(t); kernel(args...); }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emul_launch(kernel, grid, block, __VA_ARGS__)

inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline float __builtin_amdgcn_sqrtf(float x) { return sqrtf(x); }
inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / sqrtf(x); }
inline uint32_t __builtin_amdgcn_readfirstlane(uint32_t x) { return x; }
inline uint32_t __builtin_amdgcn_readlane(uint32_t x, int) { return x; }
// a wave of one lane
inline unsigned long long __ballot(bool b) { return b ? 1ull : 0ull; }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
inline void sincospif(float x, float *s, float *c) { *s = sinf(3.14159265358979323846f * x); *c = cosf(3.14159265358979323846f * x); }
inline void sincospi(double x, double *s, double *c) { *s = sin(3.14159265358979323846 * x); *c = cos(3.14159265358979323846 * x); }
template <typename T> inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
using std::max;
using std::min;
