// tests/hipemu/dabphy_wave_ops.h -- TEST INFRASTRUCTURE: functional model of welle.io_amd/csrc/dabphy_wave_ops.h for the
// CPU execution model (same results, lane exchange through the emulated wave instead of DPP instructions).
#pragma once
#include <hip/hip_runtime.h>

namespace dabphy {

static inline float chain16(float acc, float x, int nk)
{
    // lane i adds x of lanes i, i+1, ... of its row of 16 (0 beyond the row), exactly like row_shl with bound_ctrl
    unsigned xb; memcpy(&xb, &x, 4);
    const int lane = hipemu_lane();
    for (int k = 0; k < nk; k++) {
        const bool in_row = (lane & 15) + k < 16;
        const unsigned v = hipemu_wave_exchange(xb, in_row ? lane + k : lane, true);
        float f; memcpy(&f, &v, 4);
        acc += in_row ? f : 0.0f;
    }
    return acc;
}

} // namespace dabphy
