// welle.io_amd/csrc/dabphy_wave_ops.h -- wave-level primitives written as gfx950 instructions.
// (tests/hipemu/ carries a functional model of this header for the CPU logic tests; the kernels include it with
// angle brackets so the include path decides.)
#pragma once
#include <hip/hip_runtime.h>
#include "dabphy_common.h"

namespace dabphy {

// acc(lane 0) += x(lane 0) + x(lane 1) + ... + x(lane nk-1), one float addition at a time in that order (nk = 8 or 16).
// Lane 0 reads lane k through a row_shl:k DPP source operand, so every step of the dependent chain is exactly one
// v_add_f32_dpp; written as asm because the scheduler otherwise hoists the 15 lane moves of every block and runs
// out of registers.
#define DABPHY_DPP_ADD(K) asm volatile("v_add_f32_dpp %0, %1, %0 row_shl:" #K " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x))
__device__ __forceinline__ float chain16(float acc, float x, int nk)
{
    acc += x;
    // gfx9 DPP hazard: a VGPR written by a VALU instruction needs 2 wait states before a DPP read; inline asm is invisible
    // to the compiler's hazard recogniser, so the gap is made explicit (x may just have been produced by a v_cndmask)
    asm volatile("s_nop 1");
    DABPHY_DPP_ADD(1); DABPHY_DPP_ADD(2); DABPHY_DPP_ADD(3); DABPHY_DPP_ADD(4); DABPHY_DPP_ADD(5); DABPHY_DPP_ADD(6); DABPHY_DPP_ADD(7);
    if (nk > 8) {
        DABPHY_DPP_ADD(8); DABPHY_DPP_ADD(9); DABPHY_DPP_ADD(10); DABPHY_DPP_ADD(11);
        DABPHY_DPP_ADD(12); DABPHY_DPP_ADD(13); DABPHY_DPP_ADD(14); DABPHY_DPP_ADD(15);
    }
    return acc;
}
#undef DABPHY_DPP_ADD

// ---- packed complex arithmetic: one cf32 = one even-aligned VGPR pair, every operation one v_pk_*_f32 whose op_sel /
// neg modifiers do the real/imaginary swizzles (the compiler's own packing of scalar complex code spends one v_mov per
// swizzle).  Same IEEE operations in the same order as cmul/cadd/csub of dabphy_common.h: results are bit-identical.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f c2v(cf32 c) { v2f v; v.x = c.re; v.y = c.im; return v; }
__device__ __forceinline__ cf32 v2c(v2f v) { cf32 c; c.re = v.x; c.im = v.y; return c; }

__device__ __forceinline__ cf32 pk_add(cf32 a, cf32 b) { return v2c(c2v(a) + c2v(b)); }
__device__ __forceinline__ cf32 pk_sub(cf32 a, cf32 b) { return v2c(c2v(a) - c2v(b)); }
// a * b = (a.re b.re - a.im b.im, a.re b.im + a.im b.re).  The multiplies are vector C (the compiler folds the swizzles
// into op_sel); only the add with one negated half needs asm -- back-to-back dependent asm statements would cost an s_nop.
__device__ __forceinline__ cf32 pk_cmul(cf32 a, cf32 b)
{
    const v2f av = c2v(a), bv = c2v(b);
    const v2f p = av.xx * bv;                   // (a.re b.re, a.re b.im)
    const v2f q = av.yy * bv.yx;                // (a.im b.im, a.im b.re)
    v2f r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(p), "v"(q));
    return v2c(r);
}
// a * conj(b) = (a.re b.re - a.im (-b.im), a.re (-b.im) + a.im b.re) = (p.x + q.x, q.y - p.y): negations are exact
__device__ __forceinline__ cf32 pk_cmulc(cf32 a, cf32 b)
{
    const v2f av = c2v(a), bv = c2v(b);
    const v2f p = av.xx * bv;
    const v2f q = av.yy * bv.yx;
    v2f r; asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(q), "v"(p));
    return v2c(r);
}
// a * w for a twiddle w = (1, s) with s = +-0 (tw[0] of the KISS table is (1, -0)): (a.re * 1 - a.im * s, a.re * s + a.im * 1).  The
// products by 1 are exact and a.im * s, a.re * s are exactly +-0 (or NaN), so ONE fused multiply-add per component returns the very
// bits of the two multiplications and the addition -- zero signs, NaN and infinity included (fma rounds once, and nothing here rounds).
// The host checks w.re == 1 when it builds the table.
__device__ __forceinline__ cf32 pk_cmul_unit(cf32 a, cf32 w)
{
    v2f r; const v2f av = c2v(a), wv = c2v(w);
    asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(av), "v"(wv));
    return v2c(r);
}
// a - i b = (a.re + b.im, a.im - b.re)   and   a + i b = (a.re - b.im, a.im + b.re)
__device__ __forceinline__ cf32 pk_sub_ib(cf32 a, cf32 b)
{
    v2f r; const v2f av = c2v(a), bv = c2v(b);
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(av), "v"(bv));
    return v2c(r);
}
__device__ __forceinline__ cf32 pk_add_ib(cf32 a, cf32 b)
{
    v2f r; const v2f av = c2v(a), bv = c2v(b);
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(av), "v"(bv));
    return v2c(r);
}
// float -> int32 as v_cvt_i32_f32 does it (truncate, NaN -> 0, saturate)
__device__ __forceinline__ int cvt_i32_trunc(float x) { int r; asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }

// ---- packed uint16 pairs (Viterbi path metrics): half selections the compiler folds into the op_sel bits of the
// consuming v_pk_*_u16, and the v_perm_b32 selectors 8..11 that expand the sign bit of a 16-bit half into a whole byte
typedef unsigned short u16x2 __attribute__((vector_size(4)));
__device__ __forceinline__ u16x2 pk_dup_lo(u16x2 v) { return __builtin_shufflevector(v, v, 0, 0); }
__device__ __forceinline__ u16x2 pk_dup_hi(u16x2 v) { return __builtin_shufflevector(v, v, 1, 1); }
__device__ __forceinline__ u16x2 pk_swap(u16x2 v) { return __builtin_shufflevector(v, v, 1, 0); }
// a double that is the same in every lane, moved to scalar registers (two v_readfirstlane_b32)
__device__ __forceinline__ double uniform_f64(double x)
{
    unsigned long long u; __builtin_memcpy(&u, &x, 8);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    u = ((unsigned long long)hi << 32) | lo; __builtin_memcpy(&x, &u, 8);
    return x;
}
// an int that is the same in every lane, moved to a scalar register
__device__ __forceinline__ int uniform_i32(int x) { return __builtin_amdgcn_readfirstlane(x); }
// a wave-uniform constant the compiler must keep in a scalar register instead of folding it into literals
__device__ __forceinline__ uint32_t opaque_sgpr(uint32_t x) { asm volatile("" : "+s"(x)); return x; }
// A read-only table as the CONSTANT address space sees it: hipcc turns a wave-uniform load from ordinary global memory that the
// kernel also writes elsewhere into a VECTOR load + s_waitcnt vmcnt(0) + v_readfirstlane (the scalar cache is not coherent with
// vector stores), which drains every store in flight; loads through a constant-address-space pointer become s_load_dword*.
#define DABPHY_CONST_AS __attribute__((address_space(4)))
template <typename T> __device__ __forceinline__ const DABPHY_CONST_AS T* as_constant(const T* p)
{
    return (const DABPHY_CONST_AS T*)(unsigned long long)p;
}

// a per-lane value the compiler must treat as new at this point (keeps loop-invariant address arithmetic from being hoisted into
// registers that then have to live across a register-bound loop)
__device__ __forceinline__ uint32_t opaque_vgpr(uint32_t x) { asm volatile("" : "+v"(x)); return x; }
// (x & wave-uniform mask) | acc in one instruction
__device__ __forceinline__ uint32_t and_or(uint32_t x, uint32_t mask, uint32_t acc)
{
    uint32_t r; asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(mask), "v"(acc)); return r;
}
// bytes (0xff if negative else 0) of the int16 halves [a.lo, b.lo, a.hi, b.hi]
__device__ __forceinline__ uint32_t pk_sign_bytes(u16x2 a, u16x2 b)
{
    uint32_t au, bu; __builtin_memcpy(&au, &a, 4); __builtin_memcpy(&bu, &b, 4);
    return __builtin_amdgcn_perm(bu, au, 0x0b090a08u);       // pool {S0 = b, S1 = a}: 8 = a[15], 10 = b[15], 9 = a[31], 11 = b[31]
}

// x * m for |x|, |m| < 2^23, m wave-uniform: one v_mul_i32_i24 (written as asm: left to itself the compiler re-associates sums of such
// products, loses the operand ranges on the way and ends up with the quarter-rate 32-bit v_mul_lo_u32)
__device__ __forceinline__ int mul_i24(int x, int m) { int r; asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "s"(m), "v"(x)); return r; }
// ({hi, lo} >> sh) & 0xffffffff, 0 <= sh <= 31 (v_alignbit_b32)
__device__ __forceinline__ uint32_t funnel_shr(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }

// x * m + acc for |x|, |m| < 2^23, m wave-uniform (v_mad_i32_i24)
__device__ __forceinline__ int mad_i24(int x, int m, int acc) { int r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "s"(m), "v"(x), "v"(acc)); return r; }

// ---- lanes = trellis states (k_viterbi_sp.hip).  The 64 lanes form 32 pairs that differ in lane bit B; every lane of a pair gets
// x = the value held by the pair's lane with bit B clear and y = the value of the lane with bit B set.  B = 5, 4: v_permlane32_swap /
// v_permlane16_swap of the register with a copy of itself (gfx950); B = 3, 2: row_shr / row_shl DPP moves under a bank mask (the other
// lanes keep their own value); B = 1, 0: quad_perm DPP reads, which the compiler folds into the consuming addition.
typedef unsigned int pair_u32x2 __attribute__((ext_vector_type(2)));
template <int B> __device__ __forceinline__ void pair_values(uint32_t m, uint32_t& x, uint32_t& y)
{
    static_assert(B >= 0 && B <= 5, "lane bit");
    if constexpr (B == 5) { const pair_u32x2 r = __builtin_amdgcn_permlane32_swap(m, m, false, false); x = r[0]; y = r[1]; }
    else if constexpr (B == 4) { const pair_u32x2 r = __builtin_amdgcn_permlane16_swap(m, m, false, false); x = r[0]; y = r[1]; }
    else if constexpr (B == 3) { x = (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x118, 0xf, 0xC, false); y = (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x108, 0xf, 0x3, false); }
    else if constexpr (B == 2) { x = (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x114, 0xf, 0xA, false); y = (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x104, 0xf, 0x5, false); }
    else if constexpr (B == 1) { x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x44, 0xf, 0xf, false); y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0xEE, 0xf, 0xf, false); }
    else { x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0xA0, 0xf, 0xf, false); y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0xF5, 0xf, 0xf, false); }
}
// x * m + acc for |x|, |m| < 2^23, all per lane (v_mad_i32_i24)
__device__ __forceinline__ int mad_i24_vv(int x, int m, int acc) { int r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(x), "v"(acc)); return r; }
// y - x - (bit `lane` of the wave-uniform mask): one v_subb_co_u32 with the mask as its borrow-in (the borrow-out is discarded)
__device__ __forceinline__ uint32_t sub_borrow(uint32_t y, uint32_t x, unsigned long long mask)
{
    uint32_t r; unsigned long long bout;
    asm("v_subb_co_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(bout) : "v"(y), "v"(x), "s"(mask));
    return r;
}
// ---- two code words per wavefront (k_viterbi_sp2.hip): each half of 32 lanes is one code word, every lane holds TWO path metrics and
// the lanes of a half pair up along lane bit B = 4 .. 0.
// B = 4: swap16(r0, r1, a, b): a = the lane's own r0 where its bit 4 is clear, its partner's r1 where it is set; b = its partner's r0 where
//        the bit is clear, its own r1 where it is set (ONE v_permlane16_swap_b32).
// B < 4: partner<B>(g) = the value of g held by the lane's partner: row_ror:8 (B = 3), two bank-masked row moves (B = 2), a quad_perm read
//        (B = 1, 0).
__device__ __forceinline__ void swap16(uint32_t r0, uint32_t r1, uint32_t& a, uint32_t& b)
{
    const pair_u32x2 r = __builtin_amdgcn_permlane16_swap(r0, r1, false, false); a = r[0]; b = r[1];
}
__device__ __forceinline__ uint32_t ubfe(uint32_t x, uint32_t off, uint32_t width) { return __builtin_amdgcn_ubfe(x, off, width); }     // bits [off, off + width) of x (s_bfe_u32 / v_bfe_u32)
__device__ __forceinline__ uint32_t bit_reverse32(uint32_t x) { return __builtin_bitreverse32(x); }     // s_brev_b32 / v_bfrev_b32
// swap32(r0, r1, a, b): a = r0 of the lower 32 lanes followed by r1 of the LOWER 32 lanes (moved up); b = r0 of the UPPER 32 lanes (moved
// down) followed by r1 of the upper 32 lanes (one v_permlane32_swap_b32): the two registers of one half-wave side by side in one register
__device__ __forceinline__ void swap32(uint32_t r0, uint32_t r1, uint32_t& a, uint32_t& b)
{
    const pair_u32x2 r = __builtin_amdgcn_permlane32_swap(r0, r1, false, false); a = r[0]; b = r[1];
}
template <int B> __device__ __forceinline__ uint32_t partner(uint32_t g)
{
    static_assert(B >= 0 && B <= 3, "lane bit");
    if constexpr (B == 3) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g, 0x128, 0xf, 0xf, false);                 // row_ror:8
    else if constexpr (B == 2) {
        const int up = __builtin_amdgcn_update_dpp((int)g, (int)g, 0x114, 0xf, 0xA, false);                                 // lanes with the bit set read lane - 4
        return (uint32_t)__builtin_amdgcn_update_dpp(up, (int)g, 0x104, 0xf, 0x5, false);                                   // ... the others lane + 4
    }
    else if constexpr (B == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g, 0x4E, 0xf, 0xf, false);            // quad_perm [2,3,0,1]
    else return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g, 0xB1, 0xf, 0xf, false);                                    // quad_perm [1,0,3,2]
}
// the value lane `idx` (wave-uniform) holds, as a wave-uniform value (v_readlane_b32)
__device__ __forceinline__ uint32_t lane_get(uint32_t v, uint32_t idx) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)idx); }

// ---- buffer addressing: (wave-uniform base in a 128-bit scalar resource) + (per-lane byte offset, one VGPR) + (wave-uniform byte offset,
// one SGPR).  Global loads and stores of the form uniform_pointer[lane] otherwise cost a 64-bit VGPR address pair per array and a 64-bit
// add per access in a loop that is short of registers.  Raw buffer (stride 0), range = 2 GiB from the base.
typedef __amdgpu_buffer_rsrc_t BufRsrc;
typedef unsigned int buf_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ BufRsrc buf_rsrc(const void* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000); }
// AUX = the instruction's cache-policy bits (gfx940+: 1 = sc0, 2 = nt, 16 = sc1; MI355X_MICROARCH.md "stores of each flavour")
template <int AUX = 0>
__device__ __forceinline__ uint2 buf_load_b64(BufRsrc r, uint32_t lane_bytes, uint32_t uniform_bytes)
{
    const buf_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, lane_bytes, uniform_bytes, AUX); return make_uint2(v.x, v.y);
}
template <int AUX = 0>
__device__ __forceinline__ void buf_store_b64(BufRsrc r, uint32_t lane_bytes, uint32_t uniform_bytes, uint2 v)
{
    buf_u32x2 w; w.x = v.x; w.y = v.y; __builtin_amdgcn_raw_buffer_store_b64(w, r, lane_bytes, uniform_bytes, AUX);
}
__device__ __forceinline__ void buf_store_b32(BufRsrc r, uint32_t lane_bytes, uint32_t uniform_bytes, uint32_t v)
{
    __builtin_amdgcn_raw_buffer_store_b32(v, r, lane_bytes, uniform_bytes, 0);
}
// ... with a range of 4 GiB from the base (LDS-DMA sources of the fused decode: the ring slices of the few ensembles a wave spans)
__device__ __forceinline__ BufRsrc buf_rsrc_4g(const void* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0xffffffffu, 0x00020000); }
// LDS-DMA through a buffer resource: the dword at base + lane_bytes + uniform_bytes lands at lds_wave_base + 4 * lane (inactive lanes
// transfer nothing); no 64-bit address arithmetic, no VGPR pair per request.  Completion is tracked by vmcnt like lds_dma4.
__device__ __forceinline__ void buf_dma4(BufRsrc r, uint32_t lane_bytes, uint32_t uniform_bytes, void* lds_wave_base)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 4, lane_bytes, uniform_bytes, 0, 0);
}

// 127 / x for x in [2^-100, 2^100]: v_rcp_f32 (1 ulp) + one residual correction, 4 instructions instead of the 11 of the
// IEEE division sequence.  DIV127_VARIANT 1 adds a second correction.  dabphy_selftest_div127 compares every variant
// with the correctly rounded quotient for ALL floats of that range on the device it runs on; k_demod only uses the
// variant that test proves exact (tests/test_gpu_parity.py::test_div127_exhaustive).
template <int VARIANT>
__device__ __forceinline__ float div127_fast(float x)
{
    const float y = __builtin_amdgcn_rcpf(x);
    float q = 127.0f * y;
    float r = __builtin_fmaf(-q, x, 127.0f);
    q = __builtin_fmaf(r, y, q);
    if (VARIANT >= 1) { r = __builtin_fmaf(-q, x, 127.0f); q = __builtin_fmaf(r, y, q); }
    return q;
}
constexpr float DIV127_LO = 0x1p-100f, DIV127_HI = 0x1p100f;
// true when every lane of the wave passes
__device__ __forceinline__ bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(p) == __builtin_amdgcn_ballot_w64(true); }

// LDS-DMA: each of the 64 lanes fetches 16 bytes from its own global address + OFF; the wave's 1 KiB lands contiguously at
// lds_wave_base + OFF + 16 * lane without passing through VGPRs (global_load_lds_dwordx4, gfx950).  lds_wave_base must be
// wave-uniform (it travels in M0); OFF is the instruction's immediate offset (< 4096), applied to both addresses, so four
// consecutive KiB share one address computation and one M0 write.  Completion is tracked by vmcnt: lds_dma_wait() + a
// barrier before any lane reads the data.
template <int OFF>
__device__ __forceinline__ void lds_dma16(const void* gptr, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, OFF, 0);
}
// 4-byte variant: lane l's dword lands at lds_wave_base + 4 l (inactive lanes transfer nothing)
__device__ __forceinline__ void lds_dma4(const void* gptr, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
// the wave's LDS reads have returned (so an LDS-DMA issued after this point cannot overtake them)
__device__ __forceinline__ void lds_reads_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// ... until at most N of the wave's memory operations are still in flight (they complete in order: "everything but the
// last N requests has landed")
template <int N> __device__ __forceinline__ void lds_dma_wait_but() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

__device__ __forceinline__ uint32_t u32_max(uint32_t a, uint32_t b) { return __builtin_elementwise_max(a, b); }     // one v_max_u32

// The lanes of a wavefront run in lock step: what one lane wrote to LDS before this point is visible to the others after it without
// any instruction (a scheduling fence for the compiler).  tests/hipemu runs lanes as fibres and switches them here.
// register budget of a kernel: at least n waves per SIMD
#define DABPHY_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n)))
__device__ __forceinline__ void wave_converge() { __builtin_amdgcn_wave_barrier(); }

// accumulate into a double in LDS from many threads (order irrelevant to its users)
__device__ __forceinline__ void lds_add_f64(double* p, double v) { atomicAdd(p, v); }

} // namespace dabphy
