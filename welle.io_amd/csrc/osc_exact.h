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
#include <cmath>

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

// ---- the unchecked conversion (k_demod's hot path) -------------------------------------------------------------------------------
// osc_round spends more on the question "could this double round the wrong way?" than on the value.  The question has the same answer
// for every sample that reads a given table entry, and a handful of entries are the only ones for which it can be yes:
//   * osc_exp is within 1.4e-16 per component of the true value for EVERY index (tests/native/osc_check.cpp walks all 2 048 000
//     against long double): 2e-16 as a vector; a product of two such values computed by osc_mul adds both errors and at most
//     2.3e-16 of its own (two roundings per component).  In k_demod a symbol's per-thread base is ONE such product (the thread's
//     osc_exp at the chunk's first symbol times the osc_exp step to symbol k: < 6.3e-16) and the samples follow along a tree of
//     depth 4 (steps of 128, 256, 512, 1024 samples, each an osc_exp value: + 4.3e-16 per level): error < 2.4e-15.
//   * only 36 of the table's 2 048 000 entries have a cosine or sine that lies within OSC_UNSAFE_DIST = 2^-45 = 2.8e-14 (twelve times that bound) of the
//     midpoint of two floats (the four entries on the axes, whose sine or cosine is 0 or 1e-16, and a few generic ones;
//     osc_unsafe_list, the same libm calls that build the table).  For every other entry (float)computed == (float)table.
// So the synchroniser, which knows phase and frequency of every sample of a frame before the frame is demodulated, marks the symbols
// whose 2048 samples touch one of those entries (osc_hazard_mask, FrameDesc::osc_hazard: n f = L - u (mod RATE) solved for n per
// entry, about one symbol in 25), k_demod takes the checked path (osc_round + table) for them and a plain double -> float
// conversion for all others.
constexpr double OSC_UNSAFE_DIST = 0x1p-45;
constexpr int OSC_MAX_UNSAFE = 64;

// solutions of  n f == c (mod RATE): n == n0 (mod per), or none (per = 0).  One extended Euclid on (RATE, f mod RATE) yields
// g = gcd and t with t f == g (mod RATE); then n0 = (c / g) t mod (RATE / g).  32-bit throughout (|t| < RATE < 2^21) but the last product.
struct OscCongruence { int32_t n0, per; };
__host__ __device__ inline OscCongruence osc_solve(int32_t f_hz, int32_t c)
{
    OscCongruence r; r.n0 = 0; r.per = 0;
    int32_t fm = f_hz % INPUT_RATE; if (fm < 0) fm += INPUT_RATE;
    c %= INPUT_RATE; if (c < 0) c += INPUT_RATE;
    int32_t r0 = INPUT_RATE, r1 = fm, t0 = 0, t1 = 1;
    while (r1) { const int32_t q = r0 / r1; const int32_t r2 = r0 - q * r1; r0 = r1; r1 = r2; const int32_t t2 = t0 - q * t1; t0 = t1; t1 = t2; }
    const int32_t g = r0;                                          // RATE when fm == 0 (then t0 = 0)
    if (c % g) return r;
    const int32_t P = INPUT_RATE / g;
    int32_t inv = t0 % P; if (inv < 0) inv += P;
    r.n0 = (int32_t)(((int64_t)(c / g) * inv) % P); r.per = P;
    return r;
}

// Which symbols of a frame (bit s of the 76, symbol 0 = PRS) read table entry u in their useful part.  Phase of the sample j behind
// the frame's sync buffer start: (L0 - (j + 1) f_prs) up to the end of the PRS (j < J0 = start_index + T_u), then
// (L1 - (j - J0 + 1) f_sym); the useful part of the PRS is j = start_index .. +2047, of symbol s >= 1 j = J0 + (s - 1) T_s + T_g .. +2047.
// Conservative where enumeration would be long (period <= T_u: every symbol is marked).
__host__ __device__ inline void osc_hazard_entry(uint32_t (&mask)[3], int32_t u, int32_t start_index, int32_t L0, int32_t f_prs, int32_t L1, int32_t f_sym)
{
    {   // PRS: m = j + 1 in [start_index + 1, start_index + T_u],  m f_prs == L0 - u
        const OscCongruence k = osc_solve(f_prs, L0 - u);
        if (k.per) {
            const int64_t lo = (int64_t)start_index + 1, hi = (int64_t)start_index + T_U;
            int64_t m = k.n0; if (m < lo) m += (lo - m + k.per - 1) / k.per * k.per;
            if (m <= hi) mask[0] |= 1u;
        }
    }
    {   // data symbols: m = j - J0 + 1 in [1, 75 T_s],  m f_sym == L1 - u;  (m - 1) = (s - 1) T_s + T_g + n, 0 <= n < T_u
        const OscCongruence k = osc_solve(f_sym, L1 - u);
        if (k.per) {
            if (k.per <= T_U) { mask[0] |= ~1u; mask[1] = ~0u; mask[2] |= 0xfffu; }
            else {
                for (int64_t m = k.n0 ? k.n0 : k.per; m <= 75LL * T_S; m += k.per) {
                    const int32_t q = (int32_t)((m - 1) / T_S), r = (int32_t)((m - 1) % T_S);
                    if (r >= T_G) { const int s = q + 1; mask[s >> 5] |= 1u << (s & 31); }
                }
            }
        }
    }
}

// the table entries whose cosine or sine (as the table builder's libm returns them) lies within OSC_UNSAFE_DIST of the midpoint of
// two adjacent floats; returns their number, -1 when `out` (OSC_MAX_UNSAFE entries, padded with -1) is too small
inline int osc_unsafe_list(int32_t* out)
{
    int n = 0;
    for (int i = 0; i < OSC_MAX_UNSAFE; i++) out[i] = -1;
    for (int i = 0; i < INPUT_RATE; i++) {
        const double v[2] = {cos(2.0 * M_PI * i / INPUT_RATE), sin(2.0 * M_PI * i / INPUT_RATE)};     // ofdm-processor.cpp:93-95
        bool unsafe = false;
        for (int k = 0; k < 2; k++) {
            const float f = (float)v[k];
            const double up = ((double)f + (double)nextafterf(f, INFINITY)) * 0.5, dn = ((double)f + (double)nextafterf(f, -INFINITY)) * 0.5;
            if (fabs(v[k] - up) < OSC_UNSAFE_DIST || fabs(v[k] - dn) < OSC_UNSAFE_DIST) unsafe = true;
        }
        if (unsafe) { if (n == OSC_MAX_UNSAFE) return -1; out[n++] = i; }
    }
    return n;
}

} // namespace dabphy
