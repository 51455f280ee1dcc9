// welle.io_amd/csrc/mix2048.h -- oscillator-corrected sample fetch shared by the synchroniser (k_sync.hip) and the TII
// side kernel (k_tii.hip): OFDMProcessor::getSamples (ofdm-processor.cpp:186-224) in closed form.
#pragma once
#include "dabphy_kernels.h"
#include <dabphy_wave_ops.h>
#include "osc_exact.h"

namespace dabphy {

__device__ __forceinline__ int32_t mod_rate64(int64_t x)
{
    int64_t r = x % INPUT_RATE;
    if (r < 0) r += INPUT_RATE;
    return (int32_t)r;
}

// sample `off` (from absolute position pos) of ensemble stream, oscillator applied: phase (L - (rel+1) f) mod RATE
__device__ __forceinline__ cf32 mixed_sample(const cf32* __restrict__ iq, int64_t ring, int64_t pos, int64_t off,
                                             const cf32* __restrict__ nco, int32_t L, int32_t f, int64_t rel)
{
    const cf32 x = iq[(pos + off) % ring];
    const int32_t ph = (f == 0) ? L : mod_rate64((int64_t)L - (rel + 1) * (int64_t)f);
    return cmul(x, nco[ph]);
}

// 2048 samples starting `off` after pos, phase progression from (L, f) with the first sample at relative index rel0,
// delivered in round-A order: v[8h + j] = x[t + 128h + 256j].  The oscillator values are computed (osc_exact.h); the
// table is only read for the rare sample whose rounding the computation cannot decide.
__device__ __forceinline__ void load_mix2048(cf32 (&v)[16], const cf32* __restrict__ iq, int64_t ring, int64_t pos, int64_t off,
                                             const cf32* __restrict__ nco, int32_t L, int32_t f, int64_t rel0, int t)
{
    int64_t a = (pos + off + t) % ring;
    const int32_t ph0 = mod_rate64((int64_t)L - (rel0 + t + 1) * (int64_t)f);
    const int32_t step = mod_rate64(128LL * f);
    cf32 o[16];
    uint32_t hard = 0;
    {
        dc64 e = osc_exp(ph0);
        const dc64 d = osc_step(128, f);
#pragma unroll
        for (int i = 0; i < 16; i++) { hard |= osc_round(e, o[i]) << i; if (i < 15) e = osc_mul(e, d); }
    }
    if (!wave_all(hard == 0)) {
        int32_t ph = ph0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if ((hard >> i) & 1u) o[i] = nco[ph];
            ph -= step; if (ph < 0) ph += INPUT_RATE;
        }
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        v[(i & 1) * 8 + (i >> 1)] = cmul(iq[a], o[i]);
        a += 128; if (a >= ring) a -= ring;
    }
}

} // namespace dabphy
