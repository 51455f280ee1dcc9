// tools/ubench/valu_rate.hip -- issue rate of the VALU instructions the Viterbi kernel is made of (not part of the product).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
template <int OP>
__global__ void __launch_bounds__(64) k(unsigned* out, int iters, unsigned seed)
{
    unsigned a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = seed * (i + 1) + threadIdx.x;
    unsigned b = seed ^ 0x12345, c = seed + 77;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 1) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 2) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 3) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 4) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 5) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 6) asm volatile("v_pk_add_u16 %0, %0, %1 op_sel:[1,0] op_sel_hi:[1,1]" : "+v"(a[i]) : "v"(b));
                if (OP == 7) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(c));
                if (OP == 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(unsigned long long*)&a[i & 14]) : "v"(*(unsigned long long*)&a[(i & 14) ^ 2]));
                if (OP == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(unsigned long long*)&a[i & 14]) : "v"(*(unsigned long long*)&a[(i & 14) ^ 2]));
                if (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 11) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            }
        }
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s ^= a[i];
    if (s == 0x7fffffff) out[0] = s;
}

template <int OP> void run(const char* name, unsigned* d, int waves)
{
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(1024 * waves), dim3(64), 0, 0, d, 10, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(1024 * waves), dim3(64), 0, 0, d, iters, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 64 * waves;
    printf("%-34s waves/SIMD %d: %.3f ms -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}

int main()
{
    unsigned* d; hipMalloc((void**)&d, 64);
    for (int w : {1, 2, 4, 5}) {
        run<0>("v_add_u32", d, w); run<1>("v_pk_add_u16", d, w); run<2>("v_pk_min_u16", d, w); run<3>("v_pk_sub_i16", d, w);
        run<4>("v_perm_b32 (vgpr sel)", d, w); run<7>("v_perm_b32 (sgpr sel)", d, w); run<5>("v_and_or_b32", d, w); run<6>("v_pk_add_u16 op_sel", d, w);
        run<8>("v_pk_add_f32", d, w); run<9>("v_pk_mul_f32", d, w); run<10>("v_fma_f32", d, w); run<11>("v_mul_f32", d, w);
    }
    return 0;
}
