// welle.io_amd/csrc/osc_exact.h -- the receiver's oscillator without the table gather.
//
// The reference mixes every input sample with oscillatorTable[localPhase] (ofdm-processor.cpp:93-95 builds the
// 2 048 000-entry table as (float)cos / (float)sin of the double 2 pi i / INPUT_RATE; :211-214 walks it with a step of
// coarse+fine Hz per sample).  On the GPU a step of f entries per sample puts the 64 lanes of a load on 64 different
// cache lines of a 16 MB table: 128 bytes of fabric traffic per 8-byte entry, 16x the IQ stream itself.  This header
// produces the SAME float values arithmetically:
//
//   * exp(j 2 pi i / RATE) in double precision (osc_exp: octant reduction in integers, Taylor polynomials on [0, pi/4],
//     error < 1e-15), advanced from sample to sample by complex multiplication with exp(-j 2 pi k f / RATE) (osc_mul);
//   * a double d within E of the table's double rounds to the table's float unless a float rounding boundary lies within
//     E of d.  osc_round converts d + M and d - M (M = 2^-43 > 6 E for the longest chain used) and
//     reports the sample "hard" when the two floats differ; hard samples -- about one in 10^5 -- are re-read from the
//     table itself, so the result is bit-identical by construction, not by luck.
#pragma once
#include "dabphy_common.h"

namespace dabphy {

struct dc64 { double re, im; };

constexpr double OSC_MARGIN = 0x1p-43;

__host__ __device__ __forceinline__ dc64 osc_mul(dc64 a, dc64 b)
{
    dc64 r;
    r.re = __builtin_fma(a.re, b.re, -(a.im * b.im));
    r.im = __builtin_fma(a.re, b.im, a.im * b.re);
    return r;
}

// exp(j 2 pi i / INPUT_RATE), 0 <= i < INPUT_RATE
__host__ __device__ __forceinline__ dc64 osc_exp(int32_t i)
{
    constexpr int32_t OCT = INPUT_RATE / 8;                     // 256000 phase steps per octant
    const int32_t oct = i / OCT, r = i - oct * OCT;
    const int32_t rr = (oct & 1) ? OCT - r : r;                 // reflected in odd octants: 0 <= rr <= OCT
    const double x = (double)rr * (6.283185307179586476925286766559 / INPUT_RATE);   // [0, pi/4]
    const double z = x * x;
    // Taylor series: truncation < 3e-19 on [0, pi/4]
    double cs = 1.0 / 6402373705728000.0;                       // 1/18!
    cs = __builtin_fma(cs, z, -1.0 / 20922789888000.0);         // -1/16!
    cs = __builtin_fma(cs, z, 1.0 / 87178291200.0);             // 1/14!
    cs = __builtin_fma(cs, z, -1.0 / 479001600.0);              // -1/12!
    cs = __builtin_fma(cs, z, 1.0 / 3628800.0);                 // 1/10!
    cs = __builtin_fma(cs, z, -1.0 / 40320.0);                  // -1/8!
    cs = __builtin_fma(cs, z, 1.0 / 720.0);                     // 1/6!
    cs = __builtin_fma(cs, z, -1.0 / 24.0);                     // -1/4!
    cs = __builtin_fma(cs, z, 0.5);                             // 1/2!
    cs = __builtin_fma(-cs, z, 1.0);                            // cos x = 1 - z (1/2 - z (...))
    double sn = -1.0 / 121645100408832000.0;                    // -1/19!
    sn = __builtin_fma(sn, z, 1.0 / 355687428096000.0);         // 1/17!
    sn = __builtin_fma(sn, z, -1.0 / 1307674368000.0);          // -1/15!
    sn = __builtin_fma(sn, z, 1.0 / 6227020800.0);              // 1/13!
    sn = __builtin_fma(sn, z, -1.0 / 39916800.0);               // -1/11!
    sn = __builtin_fma(sn, z, 1.0 / 362880.0);                  // 1/9!
    sn = __builtin_fma(sn, z, -1.0 / 5040.0);                   // -1/7!
    sn = __builtin_fma(sn, z, 1.0 / 120.0);                     // 1/5!
    sn = __builtin_fma(sn, z, -1.0 / 6.0);                      // -1/3!
    sn = __builtin_fma(sn * z, x, x);                           // sin x = x + x z (...)
    // angle = q pi/2 + x (even octant) or q pi/2 - x (odd octant), q = (oct + 1) / 2
    if (oct & 1) sn = -sn;
    const int q = ((oct + 1) >> 1) & 3;
    dc64 e;
    e.re = (q == 0) ? cs : (q == 1) ? -sn : (q == 2) ? -cs : sn;
    e.im = (q == 0) ? sn : (q == 1) ? cs : (q == 2) ? -sn : -cs;
    return e;
}

// exp(-j 2 pi k f / RATE): the factor that advances the oscillator by k samples at f Hz (the table index DEcreases by f
// per sample, ofdm-processor.cpp:212-213)
__host__ __device__ __forceinline__ dc64 osc_step(int64_t k, int32_t f_hz)
{
    int64_t r = (-(k * (int64_t)f_hz)) % INPUT_RATE;
    if (r < 0) r += INPUT_RATE;
    return osc_exp((int32_t)r);
}

// double -> the table's float; returns nonzero when a rounding boundary is too close to decide (caller reads the table)
__host__ __device__ __forceinline__ uint32_t osc_round(dc64 v, cf32& out)
{
    const float rh = (float)(v.re + OSC_MARGIN), rl = (float)(v.re - OSC_MARGIN);
    const float ih = (float)(v.im + OSC_MARGIN), il = (float)(v.im - OSC_MARGIN);
    out.re = rh; out.im = ih;
    return (uint32_t)(rh != rl) | (uint32_t)(ih != il);
}

} // namespace dabphy
