// welle.io_amd/csrc/dabphy_wave_ops.h -- wave-level primitives written as gfx950 instructions.
// (tests/hipemu/ carries a functional model of this header for the CPU logic tests; the kernels include it with
// angle brackets so the include path decides.)
#pragma once
#include <hip/hip_runtime.h>

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

} // namespace dabphy
