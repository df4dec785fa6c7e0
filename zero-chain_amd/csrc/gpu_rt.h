// Thin runtime layer: HIP on the product build; a host emulation on the ZK_EMU build.
//
// ZK_EMU is a TEST-ONLY build of the same kernel sources for x86 (tests/emu/), used by the
// CPU-side test-suite to exercise kernel index math, the sort/accumulate/reduce pipeline and
// the C-ABI host logic where no GPU exists.  It is never loaded by the product loader
// (zero_chain_amd/_lib.py refuses anything but the gfx950 library) and never measured.
#pragma once

#ifndef ZK_EMU
// ------------------------------------------------------------------ product: real HIP
#include <hip/hip_runtime.h>
#define ZK_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__)
#define ZK_LAUNCH_SYNC ZK_LAUNCH   // kernels that use __syncthreads()
#define ZK_SHARED __shared__
#define ZK_DYN_SHARED(type, name) extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw[]; \
    type* name = reinterpret_cast<type*>(name##_raw)
#else
// ------------------------------------------------------------------ test-only emulation
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include <mutex>
#include <condition_variable>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__
#ifdef ZK_EMU_NO_FIBERS
#define ZK_SHARED static                /* sanitizer build: blocks run one after another, OS threads share the array */
#else
#define ZK_SHARED static thread_local   /* private to the block a worker of the emulation runs (tests/emu/emu_rt.cpp) */
#endif

struct emu_dim3 {
    unsigned x, y, z;
    emu_dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef emu_dim3 dim3;
extern thread_local emu_dim3 threadIdx, blockIdx;
extern emu_dim3 blockDim, gridDim;
#ifdef ZK_EMU_NO_FIBERS
extern unsigned char* emu_dyn_shared;                /* one block at a time, its OS threads share the pointer */
#else
extern thread_local unsigned char* emu_dyn_shared;   /* of the block the calling worker runs */
#endif
typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };

struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
struct uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
void emu_barrier_wait();
static inline void __syncthreads() { emu_barrier_wait(); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicMin(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old > v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
static inline hipError_t hipFree(void* p) { free(p); return 0; }
static inline hipError_t hipHostMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
static inline hipError_t hipHostFree(void* p) { free(p); return 0; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, int, hipStream_t) {
    for (size_t r = 0; r < height; r++) memcpy((char*)d + r * dpitch, (const char*)s + r * spitch, width);
    return 0;
}
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return 0; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return 0; }
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 0; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return 0; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = 0; return 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }

void emu_launch(emu_dim3 grid, emu_dim3 block, size_t shmem, bool needs_sync, const std::function<void()>& body);
#define ZK_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    emu_launch(grid, block, shmem, false, [&]() { kernel(__VA_ARGS__); })
#define ZK_LAUNCH_SYNC(kernel, grid, block, shmem, stream, ...) \
    emu_launch(grid, block, shmem, true, [&]() { kernel(__VA_ARGS__); })
#define ZK_DYN_SHARED(type, name) type* name = reinterpret_cast<type*>(emu_dyn_shared)
#endif
