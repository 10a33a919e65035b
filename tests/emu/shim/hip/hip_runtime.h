// A stand-in for <hip/hip_runtime.h> that lets the library's HIP sources be compiled as ordinary C++ and run on the
// host, for tests only: the kernels execute lane by lane on fibers that meet at every wave- or block-wide
// operation (ballot, shuffle, readfirstlane, barrier), "device" memory is host memory, streams run synchronously.
// What it is for: running the product's own kernel and host code under a debugger, a sanitizer and the CPU test
// suite, against the same oracle the GPU tests use. What it is not: a way to ship or time anything. The library
// built from it carries another name and is loaded by the tests alone (tests/emulation.py).
#pragma once

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

#ifndef RT_HOST_EMULATION
#define RT_HOST_EMULATION 1
#endif

// ---- language ----
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local          // one copy per work-group: each work-group runs on one host thread
#define __constant__ static const
#define __align__(n) __attribute__((aligned(n)))
#define HIP_SYMBOL(x) (::emu::Symbol{&(x), sizeof(x)})
#define __HIP_MEMORY_SCOPE_AGENT 0

struct alignas(8) uint2 { uint32_t x, y; };
struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { uint2 v; v.x = x; v.y = y; return v; }
static inline float2 make_float2(float x, float y) { float2 v; v.x = x; v.y = y; return v; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

template <class T> static inline T min(T a, T b) { return b < a ? b : a; }
template <class T> static inline T max(T a, T b) { return a < b ? b : a; }

// ---- per-lane intrinsics ----
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline int32_t __float_as_int(float f) { int32_t u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __int_as_float(int32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t __umul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_logf(x) log2f(x)       // v_log_f32 is the base-2 logarithm
static inline uint32_t emu_mbcnt(uint32_t mask, uint32_t add, unsigned shift) {
    const unsigned lane = threadIdx.x & 63u;      // (kernels that use it are one-dimensional)
    if (lane <= shift) return add;
    const unsigned below = lane - shift;
    return add + (uint32_t)__builtin_popcount(below >= 32 ? mask : (mask & ((1u << below) - 1u)));
}
#define __builtin_amdgcn_mbcnt_lo(mask, add) emu_mbcnt((mask), (add), 0u)
#define __builtin_amdgcn_mbcnt_hi(mask, add) emu_mbcnt((mask), (add), 32u)

static inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline uint32_t atomicMax(uint32_t* p, uint32_t v) {
    uint32_t old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), __ATOMIC_RELAXED)

// ---- wave- and block-wide operations: every lane that is still running must arrive at the same one ----
namespace emu {
struct Symbol { const void* address; size_t bytes; };
enum Collective { kBallot = 1, kFirstLane, kShuffleDown, kBarrier };
uint64_t WaveExchange(int kind, uint64_t value, unsigned offset);
void BlockBarrier();
void Launch(dim3 grid, dim3 block, const std::function<void()>& body);
}  // namespace emu
static inline unsigned long long __ballot(int predicate) { return emu::WaveExchange(emu::kBallot, predicate ? 1u : 0u, 0u); }
static inline uint32_t emu_readfirstlane(uint32_t v) { return (uint32_t)emu::WaveExchange(emu::kFirstLane, v, 0u); }
#define __builtin_amdgcn_readfirstlane(v) emu_readfirstlane(v)
template <class T> static inline T __shfl_down(T v, unsigned delta, int width = 64) {
    static_assert(sizeof(T) <= 8, "shuffles of up to 64 bits");
    (void)width;
    uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
    raw = emu::WaveExchange(emu::kShuffleDown, raw, delta);
    T out; memcpy(&out, &raw, sizeof(T)); return out;
}
static inline void __syncthreads() { emu::BlockBarrier(); }

// ---- runtime API: host memory, synchronous streams ----
enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorInvalidSymbol = 13, hipErrorNotReady = 600,
                  hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1 };
struct emu_stream; struct emu_event;
typedef emu_stream* hipStream_t;
typedef emu_event* hipEvent_t;
constexpr unsigned hipStreamNonBlocking = 1u, hipEventDisableTiming = 2u, hipHostMallocDefault = 0u;

const char* hipGetErrorString(hipError_t e);
hipError_t hipGetLastError();
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipDeviceGetAttribute(int* value, hipDeviceAttribute_t attr, int device);
hipError_t hipDeviceSynchronize();
hipError_t emu_malloc(void** p, size_t bytes);
template <class T> static inline hipError_t hipMalloc(T** p, size_t bytes) { return emu_malloc((void**)p, bytes); }
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned flags = 0) { (void)flags; return emu_malloc((void**)p, bytes); }
hipError_t hipFree(void* p);
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s = nullptr);
hipError_t hipMemset(void* dst, int value, size_t bytes);
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s = nullptr);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
// (a __device__ variable is a host variable here; its address enters the table of device allocations, see emu_register_range)
extern "C" void emu_register_range(const void* p, size_t bytes);
static inline hipError_t hipGetSymbolAddress(void** p, emu::Symbol symbol) {
    *p = const_cast<void*>(symbol.address);
    emu_register_range(symbol.address, symbol.bytes);
    return hipSuccess;
}
template <class F> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 1; return hipSuccess; }

// A launch first offers itself, by name and with its arguments laid out as the device's kernel-argument segment (each
// argument at its natural alignment), to a hook a test may have installed (emu_set_launch_hook: tests/emu/isa runs the
// compiler's gfx950 assembly of the kernel instead of its C++ source); without a hook, or when the hook declines, the
// kernel's source runs on fibers as usual.
namespace emu {
template <class T> static inline void PackOne(unsigned char* buf, size_t& at, const T& v) {
    at = (at + alignof(T) - 1) / alignof(T) * alignof(T);
    memcpy(buf + at, &v, sizeof(T));
    at += sizeof(T);
}
bool OfferLaunch(const char* kernel, dim3 grid, dim3 block, const unsigned char* args, size_t bytes);
template <class... A> static inline bool Offer(const char* kernel, dim3 grid, dim3 block, const A&... a) {
    alignas(16) unsigned char buf[1024];
    static_assert((sizeof(A) + ... + 0) + 16 * sizeof...(A) <= sizeof buf, "kernel arguments larger than the packing buffer");
    memset(buf, 0, sizeof buf);
    size_t at = 0;
    (PackOne(buf, at, a), ...);
    return OfferLaunch(kernel, grid, block, buf, at);
}
}  // namespace emu
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    do { if (!::emu::Offer(#kernel, (grid), (block), __VA_ARGS__)) ::emu::Launch((grid), (block), [=]() { kernel(__VA_ARGS__); }); } while (0)
