// tools/ubench/valu_rate2.hip -- issue rate of candidate VALU instructions for the Viterbi ACS / traceback (not part of the product).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate2.hip -o tools/ubench/valu_rate2 ; run on the GPU box.
// Each kernel issues 64 independent instances of ONE instruction per loop iteration (16 registers x 4), so the figure is the
// issue cost per wave-instruction on a SIMD shared by `waves` resident waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define OPS(X) \
    X(0,  "v_add_u32 %0, %0, %1", "v") \
    X(1,  "v_sub_u32 %0, %0, %1", "v") \
    X(2,  "v_xor_b32 %0, %0, %1", "v") \
    X(3,  "v_and_b32 %0, %0, %1", "v") \
    X(4,  "v_or_b32 %0, %0, %1", "v") \
    X(5,  "v_lshlrev_b32 %0, 3, %0", "v") \
    X(6,  "v_lshrrev_b32 %0, %1, %0", "v") \
    X(7,  "v_min_u32 %0, %0, %1", "v") \
    X(8,  "v_max_i32 %0, %0, %1", "v") \
    X(9,  "v_min_f32 %0, %0, %1", "v") \
    X(10, "v_add_f32 %0, %0, %1", "v") \
    X(11, "v_cndmask_b32 %0, %0, %1, vcc", "v") \
    X(12, "v_bfi_b32 %0, %1, %0, %2", "v") \
    X(13, "v_bfe_u32 %0, %0, 3, 6", "v") \
    X(14, "v_alignbit_b32 %0, %1, %0, 1", "v") \
    X(15, "v_lshl_or_b32 %0, %0, 1, %1", "v") \
    X(16, "v_lshl_add_u32 %0, %0, 1, %1", "v") \
    X(17, "v_add3_u32 %0, %0, %1, %2", "v") \
    X(18, "v_or3_b32 %0, %0, %1, %2", "v") \
    X(19, "v_xad_u32 %0, %0, %1, %2", "v") \
    X(20, "v_and_or_b32 %0, %0, %1, %2", "v") \
    X(21, "v_mad_i32_i24 %0, %0, %1, %2", "v") \
    X(22, "v_mad_u32_u24 %0, %0, %1, %2", "v") \
    X(23, "v_mul_u32_u24 %0, %0, %1", "v") \
    X(24, "v_add_u16 %0, %0, %1", "v") \
    X(25, "v_min_u16 %0, %0, %1", "v") \
    X(26, "v_sub_u16 %0, %0, %1", "v") \
    X(27, "v_min_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1", "v") \
    X(28, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", "v") \
    X(29, "v_pk_add_u16 %0, %0, %1", "v") \
    X(30, "v_pk_min_u16 %0, %0, %1", "v") \
    X(31, "v_pk_sub_i16 %0, %0, %1", "v") \
    X(32, "v_pk_sub_u16 %0, %0, %1 clamp", "v") \
    X(33, "v_pk_max_i16 %0, %0, %1", "v") \
    X(34, "v_pk_ashrrev_i16 %0, 15, %0", "v") \
    X(35, "v_perm_b32 %0, %0, %1, %2", "v") \
    X(36, "v_min3_u32 %0, %0, %1, %2", "v") \
    X(37, "v_med3_i32 %0, %0, %1, %2", "v") \
    X(38, "v_sad_u16 %0, %0, %1, %2", "v") \
    X(39, "v_sad_u8 %0, %0, %1, %2", "v") \
    X(40, "v_dot4_i32_i8 %0, %0, %1, %2", "v") \
    X(41, "v_dot2_u32_u16 %0, %0, %1, %2", "v") \
    X(42, "v_mov_b32 %0, %1", "v") \
    X(43, "v_fma_f32 %0, %0, %1, %2", "v") \
    X(44, "v_sub_f32 %0, %0, %1", "v") \
    X(45, "v_max_f32 %0, %0, %1", "v") \
    X(46, "v_cmp_lt_u32 vcc, %0, %1", "v") \
    X(47, "v_cmp_lt_u16 vcc, %0, %1", "v") \
    X(48, "v_cmp_lt_i16 s[20:21], %0, %1", "v") \
    X(49, "v_add_co_u32 %0, vcc, %0, %1", "v") \
    X(50, "v_pk_mul_lo_u16 %0, %0, %1", "v") \
    X(51, "v_pk_mad_u16 %0, %0, %1, %2", "v") \
    X(52, "v_pk_add_i16 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,1]", "v") \
    X(53, "v_pk_lshlrev_b16 %0, 1, %0", "v") \
    X(54, "v_not_b32 %0, %0", "v") \
    X(55, "v_bfm_b32 %0, %0, %1", "v") \
    X(56, "v_bcnt_u32_b32 %0, %0, %1", "v") \
    X(57, "v_mbcnt_lo_u32_b32 %0, %0, %1", "v") \
    X(58, "v_cvt_pk_u8_f32 %0, %0, %1, %2", "v") \
    X(59, "v_lerp_u8 %0, %0, %1, %2", "v") \
    X(60, "v_msad_u8 %0, %0, %1, %2", "v") \
    X(61, "v_max3_u32 %0, %0, %1, %2", "v") \
    X(62, "v_sub_co_u32 %0, vcc, %0, %1", "v") \
    X(63, "v_ashrrev_i32 %0, 15, %0", "v")

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
#define X(N, S, C) if (OP == N) asm volatile(S : "+v"(a[i]) : "v"(b), "v"(c) : "vcc", "s20", "s21");
                OPS(X)
#undef X
                if (OP == 100) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(*(unsigned long long*)&a[i & 14]) : "v"(b));
                if (OP == 101) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(*(unsigned long long*)&a[i & 14]));
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
    const int iters = 10000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(1024 * waves), dim3(64), 0, 0, d, 10, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(1024 * waves), dim3(64), 0, 0, d, iters, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 64 * waves;
    printf("%-100s waves/SIMD %d: %.3f ms -> %.2f cycles\n", name, waves, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    fflush(stdout);
}

int main()
{
    unsigned* d; hipMalloc((void**)&d, 64);
    for (int w : {4}) {
#define X(N, S, C) run<N>(S, d, w);
        OPS(X)
#undef X
        run<100>("v_lshrrev_b64 (vgpr shift)", d, w);
        run<101>("v_lshlrev_b64 (imm)", d, w);
    }
    return 0;
}
