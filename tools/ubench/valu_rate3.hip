// tools/ubench/valu_rate3.hip -- issue rate of the instructions k_demod is made of: packed / plain f32, f64, conversions, reciprocal
// (not part of the product).  build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate3.hip -o /tmp/valu_rate3 ; run on the GPU box.
// Each kernel issues 64 independent instances of ONE instruction per loop iteration (8 register pairs x 8).
#include <hip/hip_runtime.h>
#include <cstdio>

#define OPS(X) \
    X(0,  "v_pk_mul_f32 %0, %0, %1") \
    X(1,  "v_pk_add_f32 %0, %0, %1") \
    X(2,  "v_pk_fma_f32 %0, %0, %1, %2") \
    X(3,  "v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]") \
    X(4,  "v_mul_f64 %0, %0, %1") \
    X(5,  "v_add_f64 %0, %0, %1") \
    X(6,  "v_fma_f64 %0, %0, %1, %2") \
    X(7,  "v_pk_mov_b32 %0, %1, %2") \
    X(8,  "v_lshl_add_u64 %0, %0, 3, %1") \
    X(9,  "v_mov_b64 %0, %1")

#define OPS32(X) \
    X(20, "v_mul_f32 %0, %0, %1") \
    X(21, "v_add_f32 %0, %0, %1") \
    X(22, "v_fma_f32 %0, %0, %1, %2") \
    X(23, "v_rcp_f32 %0, %0") \
    X(24, "v_cvt_i32_f32 %0, %0") \
    X(25, "v_add_f32 %0, |%0|, |%1|") \
    X(26, "v_min_f32 %0, %0, %1") \
    X(27, "v_cndmask_b32 %0, %0, %1, vcc") \
    X(28, "v_cndmask_b32 %0, %0, %1, s[20:21]") \
    X(29, "v_bfi_b32 %0, %1, %0, %2") \
    X(30, "v_mul_f32 %0, -%0, %1") \
    X(31, "v_fmac_f32 %0, %1, %2")

template <int OP>
__global__ void __launch_bounds__(64) k(unsigned* out, int iters, unsigned seed)
{
    double a[8]; float f[16];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = 1.0 + 1e-9 * (seed * (i + 1) + threadIdx.x);
#pragma unroll
    for (int i = 0; i < 16; i++) f[i] = 1.0f + 1e-6f * (seed * (i + 1) + threadIdx.x);
    double b = 1.0 + 1e-12 * seed, c = 1e-13 * seed;
    float fb = 1.0f + 1e-7f * seed, fc = 1e-8f * seed;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
#define X(N, S) if (OP == N) asm volatile(S : "+v"(a[i]) : "v"(b), "v"(c));
                OPS(X)
#undef X
                if (OP == 10) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));
                if (OP == 11) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
#define X(N, S) if (OP == N) asm volatile(S : "+v"(f[i]) : "v"(fb), "v"(fc) : "vcc");
                OPS32(X)
#undef X
            }
        }
    }
    double s = 0; float sf = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
#pragma unroll
    for (int i = 0; i < 16; i++) sf += f[i];
    if (s == 12345.678 && sf == 3.25f) out[0] = 1;
}

template <int OP> void run(const char* name, unsigned* d, int waves)
{
    const int iters = 10000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(1024 * waves), dim3(64), 0, 0, d, 10, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(1024 * waves), dim3(64), 0, 0, d, iters, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 64 * waves;
    printf("%-80s waves/SIMD %d: %.3f ms -> %.2f cycles at 2.4 GHz\n", name, waves, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    fflush(stdout);
}

int main()
{
    unsigned* d; hipMalloc((void**)&d, 64);
    for (int w : {2, 4}) {
#define X(N, S) run<N>(S, d, w);
        OPS(X)
        run<10>("v_cvt_f32_f64", d, w);
        run<11>("v_cvt_f64_f32", d, w);
        OPS32(X)
#undef X
    }
    return 0;
}
