// tests/hipemu/hip/hip_runtime.h -- TEST INFRASTRUCTURE: a tiny single-process execution model of the HIP
// subset used by welle.io_amd/csrc, so that the *.hip kernel sources can be compiled UNCHANGED with g++ and
// their indexing / barrier logic checked on a machine without a GPU (this build container).  It is not a
// product path, not a fallback and never benchmarked: libdabphy_hip.so (hipcc, gfx950) is the product and
// refuses to initialise without a GPU.  Only tests marked "not gpu" load the emulated library.
//
// Model: blocks run one after another on the calling thread; the threads of a block are ucontext fibers
// scheduled round-robin; __syncthreads() parks a fiber until every live fiber of the block has arrived.
// Wave size is 64.  Cross-lane intrinsics exchange through a per-wave scratch with a wave-level barrier.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <functional>
#include <vector>
#include <chrono>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define HIPEMU 1

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct hipemu_idx { unsigned x, y, z; };
extern hipemu_idx threadIdx, blockIdx, blockDim, gridDim;

typedef int hipError_t;
typedef void* hipStream_t;
typedef struct hipemu_event { std::chrono::steady_clock::time_point t; }* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNoDevice = 100 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem; };

static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { memset(p, 0, sizeof *p); strcpy(p->name, "hipemu"); strcpy(p->gcnArchName, "gfx950:hipemu"); p->multiProcessorCount = 1; return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256); return *p ? hipSuccess : 2; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
static inline hipError_t hipHostGetDevicePointer(void** dp, void* hp, unsigned = 0) { *dp = hp; return hipSuccess; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t = nullptr)
{
    for (size_t r = 0; r < height; r++) memmove((char*)d + r * dpitch, (const char*)s + r * spitch, width);
    return hipSuccess;
}
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
static inline hipError_t hipPointerGetAttributes(void*, const void*) { return hipErrorInvalidValue; }
#define hipStreamNonBlocking 1
#define __noinline__ __attribute__((noinline))
#define hipEventDisableTiming 2
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
#define hipHostMallocDefault 0

// ---- execution engine (hipemu.cpp) ----
void hipemu_launch(dim3 grid, dim3 block, const std::function<void()>& body);
void hipemu_syncthreads();
unsigned hipemu_wave_exchange_impl(unsigned v, int src_lane, bool valid);   // returns v of src_lane
unsigned long long hipemu_ballot_impl(bool p);
// Every rendezvous is also a COMPILER barrier: other fibers run on this thread while one is parked, and they write the
// function-local statics that stand in for LDS.  g++ proves that the address of such a static never escapes and would keep
// its contents in registers across the (opaque) call; the "memory" clobber forbids that.
static inline unsigned hipemu_wave_exchange(unsigned v, int src_lane, bool valid)
{
    asm volatile("" ::: "memory"); const unsigned r = hipemu_wave_exchange_impl(v, src_lane, valid); asm volatile("" ::: "memory"); return r;
}
static inline unsigned long long hipemu_ballot(bool p)
{
    asm volatile("" ::: "memory"); const unsigned long long r = hipemu_ballot_impl(p); asm volatile("" ::: "memory"); return r;
}

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu_launch(dim3(grid), dim3(block), [=]() { kernel(__VA_ARGS__); })

static inline void __syncthreads() { asm volatile("" ::: "memory"); hipemu_syncthreads(); asm volatile("" ::: "memory"); }
static inline int hipemu_lane() { return (int)((threadIdx.x + threadIdx.y * blockDim.x) & 63); }
template <typename T> static inline T __shfl(T v, int src, int width = 64) {
    static_assert(sizeof(T) == 4, "4-byte shuffles only"); unsigned u; memcpy(&u, &v, 4);
    int lane = hipemu_lane(); int s = (lane & ~(width - 1)) | (src & (width - 1));
    u = hipemu_wave_exchange(u, s, true); T r; memcpy(&r, &u, 4); return r; }
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) { return __shfl(v, hipemu_lane() ^ mask, width); }
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) { int l = hipemu_lane(); int s = ((l & (width - 1)) + (int)d < width) ? l + (int)d : l; return __shfl(v, s, 64); }
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) { int l = hipemu_lane(); int s = ((l & (width - 1)) >= (int)d) ? l - (int)d : l; return __shfl(v, s, 64); }
static inline unsigned long long __ballot(int p) { return hipemu_ballot(p != 0); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __ffs(int x) { return __builtin_ffs(x); }
template <typename T> static inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> static inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline void __threadfence() {}
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct char4 { signed char x, y, z, w; };
struct uchar4 { unsigned char x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }

// v_perm_b32: byte i of the result is selected by selector byte i from {s0 (bytes 4..7), s1 (bytes 0..3)}; 0x0c = 0x00
static inline unsigned __builtin_amdgcn_perm(unsigned s0, unsigned s1, unsigned sel) {
    unsigned long long src = ((unsigned long long)s0 << 32) | s1; unsigned r = 0;
    for (int i = 0; i < 4; i++) { unsigned c = (sel >> (8 * i)) & 0xff; unsigned b = c < 8 ? (unsigned)((src >> (8 * c)) & 0xff) : (c == 0x0c ? 0u : 0xffu); r |= b << (8 * i); }
    return r;
}
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline unsigned long long wall_clock64() { static unsigned long long t = 0; return t += 100; }      // (bounded device waits time out quickly here)
static inline void __builtin_amdgcn_fence(int, const char*) {}
#define __HIP_MEMORY_SCOPE_AGENT 4
template <typename T> static inline T __hip_atomic_load(T* p, int, int) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
template <typename T> static inline void __hip_atomic_store(T* p, T v, int, int) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
static inline void __builtin_amdgcn_sched_barrier(int) {}
// v_mov_b32 dpp row_shl:k (dpp_ctrl 0x101..0x10f): lane i reads lane i+k of its row of 16, 0 beyond the row (bound_ctrl)
static inline int __builtin_amdgcn_update_dpp(int old, int src, int dpp_ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    (void)old; (void)row_mask; (void)bank_mask; (void)bound_ctrl;
    const int k = dpp_ctrl & 0xf; const int lane = hipemu_lane();
    const int srcl = ((lane & 15) + k < 16) ? lane + k : lane;
    unsigned v = hipemu_wave_exchange((unsigned)src, srcl, true);
    return ((lane & 15) + k < 16) ? (int)v : 0;
}
