// tools/ubench/copy_f4.hip -- what a plain device copy reaches on this box: the honest denominator next to the 8 TB/s specification the
// FFT stage's roofline fraction is taken against (MI355X_MICROARCH.md: "6.29 TB/s measured (float4 copy, 79 %)").  Not part of the product;
// the same kernel is reachable through the library's test ABI (dabphy_time_copy, include/dabphy_test.h), which is what bench.py calls.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/copy_f4.hip -o tools/ubench/copy_f4
// 16 bytes per lane and request, grid-stride, UNROLL independent requests in flight per lane before the first store; a sweep over the
// grid size (blocks per CU) and the unroll depth, 2 GiB in and 2 GiB out per pass (well beyond the 256 MiB Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));      // (the nontemporal builtins take native vectors)

template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) k_copy(const f4* __restrict__ src, f4* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) v[k] = NT ? __builtin_nontemporal_load(&src[i + k * stride]) : src[i + k * stride];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) { if (NT) __builtin_nontemporal_store(v[k], &dst[i + k * stride]); else dst[i + k * stride] = v[k]; }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// the two halves of a copy on their own: what the memory system gives a pure read stream (summed into one word per work-group) and a
// pure write stream
template <int UNROLL>
__global__ void __launch_bounds__(256) k_read(const f4* __restrict__ src, f4* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
#pragma unroll
        for (int k = 0; k < UNROLL; k++) acc += src[i + k * stride];
    }
    if (acc.x == 12345.0f) dst[blockIdx.x] = acc;       // (never true for the test pattern: keeps the loads alive)
}
__global__ void __launch_bounds__(256) k_write(f4* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    const f4 v = {1.0f, 2.0f, 3.0f, 4.0f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = v;
}
// each work-group copies ONE contiguous piece instead of every gridDim-th KiB
__global__ void __launch_bounds__(256) k_copy_chunked(const f4* __restrict__ src, f4* __restrict__ dst, size_t n)
{
    const size_t per = (n + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) dst[i] = src[i];
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int UNROLL, bool NT>
static int run(const f4* src, f4* dst, size_t n, int blocks, int iters)
{
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k_copy<UNROLL, NT>), dim3(blocks), dim3(256), 0, 0, src, dst, n);
    CHK(hipEventRecord(e0, 0));
    for (int it = 0; it < iters; it++) hipLaunchKernelGGL((k_copy<UNROLL, NT>), dim3(blocks), dim3(256), 0, 0, src, dst, n);
    CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("unroll %d %s blocks %5d: %.3f ms per pass, %.0f GB/s (read + write)\n", UNROLL, NT ? "nt   " : "plain", blocks, ms / iters, 2.0 * n * 16 / (ms / iters * 1e-3) / 1e9);
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    return 0;
}

int main(int argc, char** argv)
{
    const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : (size_t)2048) << 20;
    const size_t n = bytes / 16;
    f4 *src = nullptr, *dst = nullptr;
    CHK(hipMalloc((void**)&src, bytes)); CHK(hipMalloc((void**)&dst, bytes));
    CHK(hipMemset(src, 1, bytes)); CHK(hipMemset(dst, 0, bytes));
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, %zu MiB per buffer\n", p.name, cus, bytes >> 20);
    for (int per_cu : {1, 2, 4, 8, 16, 32}) {
        if (run<1, false>(src, dst, n, cus * per_cu, 10)) return 1;
        if (run<2, false>(src, dst, n, cus * per_cu, 10)) return 1;
        if (run<4, false>(src, dst, n, cus * per_cu, 10)) return 1;
        if (run<1, true>(src, dst, n, cus * per_cu, 10)) return 1;
        if (run<4, true>(src, dst, n, cus * per_cu, 10)) return 1;
    }
    auto timed = [&](const char* what, double bytes, auto&& launch) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int w = 0; w < 3; w++) launch();
        (void)hipEventRecord(e0, 0);
        for (int it = 0; it < 10; it++) launch();
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms per pass, %.0f GB/s\n", what, ms / 10, bytes / (ms / 10 * 1e-3) / 1e9);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    };
    for (int per_cu : {2, 4, 8, 16}) {
        char name[96];
        snprintf(name, sizeof name, "read only, unroll 1, blocks %5d", cus * per_cu);
        timed(name, (double)n * 16, [&]() { hipLaunchKernelGGL((k_read<1>), dim3(cus * per_cu), dim3(256), 0, 0, src, dst, n); });
        snprintf(name, sizeof name, "read only, unroll 4, blocks %5d", cus * per_cu);
        timed(name, (double)n * 16, [&]() { hipLaunchKernelGGL((k_read<4>), dim3(cus * per_cu), dim3(256), 0, 0, src, dst, n); });
        snprintf(name, sizeof name, "write only,          blocks %5d", cus * per_cu);
        timed(name, (double)n * 16, [&]() { hipLaunchKernelGGL(k_write, dim3(cus * per_cu), dim3(256), 0, 0, dst, n); });
        snprintf(name, sizeof name, "copy, one contiguous piece per work-group, blocks %5d (read + write)", cus * per_cu);
        timed(name, 2.0 * n * 16, [&]() { hipLaunchKernelGGL(k_copy_chunked, dim3(cus * per_cu), dim3(256), 0, 0, src, dst, n); });
    }
    CHK(hipFree(src)); CHK(hipFree(dst));
    return 0;
}
