// welle.io_amd/csrc/dabphy_common.h -- shared definitions of the MI355X DAB Mode-I PHY library.
//
// Numerics contract: every floating-point expression on the hot path is evaluated in
// the reference's operand types and operation order, without FMA contraction (-ffp-contract=off), so
// that soft bits -- and therefore every decoded byte -- are bit-identical to the reference CPU backend.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Timing-experiment switches (FM_EXP_* in k_viterbi.hip: variants of the fused decoder that skip a stage and decode GARBAGE;
// DEMOD_FORCE_CHECKED in k_demod.hip: one oscillator conversion compiled in) exist for A/B timing on the device only.  They compile only
// into a build that says so (-DDABPHY_EXPERIMENTS, as tools/ use), never into a product library through a stray EXTRA= flag -- and such
// a build announces itself from dabphy_create (DABPHY_WRONG_RESULTS_BUILD).
#if defined(FM_EXP_BROADCAST) || defined(FM_EXP_NOLDS) || defined(FM_EXP_NODMA) || defined(FM_EXP_NOSTORE) || defined(FM_EXP_NOWAIT) || defined(FM_EXP_NOTRACE)
#ifndef DABPHY_EXPERIMENTS
#error "FM_EXP_* switches are timing experiments that decode wrong results: build them with -DDABPHY_EXPERIMENTS"
#endif
#define DABPHY_WRONG_RESULTS_BUILD 1
#endif
#if (defined(DEMOD_EXP_NODEMAP) || defined(DEMOD_EXP_NOFFT) || defined(DEMOD_EXP_OSC_F32)) && !defined(DABPHY_EXPERIMENTS)
#error "DEMOD_EXP_* switches are timing experiments that demodulate wrong results: build them with -DDABPHY_EXPERIMENTS"
#endif
#if defined(DEMOD_FORCE_CHECKED) && !defined(DABPHY_EXPERIMENTS)
#error "DEMOD_FORCE_CHECKED is a timing experiment: build it with -DDABPHY_EXPERIMENTS"
#endif

namespace dabphy {

constexpr int T_U = 2048, T_S = 2552, T_G = 504, T_NULL = 2656, T_F = 196608, L_SYM = 76, K_CARR = 1536;
constexpr int INPUT_RATE = 2048000;
constexpr int SOFT_PER_SYM = 2 * K_CARR;           // 3072
constexpr int SOFT_PER_FRAME = 75 * SOFT_PER_SYM;  // 230400
constexpr int CIF_BITS = 55296;
constexpr int FFT_THREADS = 128;                   // one work-group = one 2048-point transform
constexpr int TII_NERR = 504;                      // TII: candidate delays err = -4 .. 499 (tii-decoder.cpp:349)
constexpr int TII_CARRIER_ROWS = 1537;             // TII: carriers k = -768 .. 768

struct cf32 { float re, im; };

// std::complex<float> operator* as the reference's compiler emits it (libgcc __mulsc3 fast path):
// (a+bi)(c+di) = (ac - bd) + (ad + bc)i, four multiplies, one subtract, one add, no FMA.
__host__ __device__ __forceinline__ cf32 cmul(cf32 x, cf32 y)
{
    cf32 r;
    r.re = x.re * y.re - x.im * y.im;
    r.im = x.re * y.im + x.im * y.re;
    return r;
}
__host__ __device__ __forceinline__ cf32 cconj(cf32 x) { cf32 r; r.re = x.re; r.im = -x.im; return r; }
__host__ __device__ __forceinline__ cf32 cadd(cf32 x, cf32 y) { cf32 r; r.re = x.re + y.re; r.im = x.im + y.im; return r; }
__host__ __device__ __forceinline__ cf32 csub(cf32 x, cf32 y) { cf32 r; r.re = x.re - y.re; r.im = x.im - y.im; return r; }
__host__ __device__ __forceinline__ float l1norm(cf32 z) { return fabsf(z.re) + fabsf(z.im); }  // MathHelper.h:48-51

// ---- libm functions the reference calls on the hot path, restated so the GPU returns the same bits as
// ---- glibc 2.35 (the C library of this image; tests/test_libm_emulation.py checks them against libm).

// hypotf (std::abs(std::complex<float>), phasereference.cpp:214, ofdm-decoder.cpp:250-259): glibc's
// sysdeps/ieee754/flt-32/e_hypotf.c computes sqrt((double)x*x + (double)y*y) in double and rounds once.
__host__ __device__ __forceinline__ float hypotf_exact(float x, float y)
{
    const double dx = (double)x, dy = (double)y;
    return (float)sqrt(dx * dx + dy * dy);
}

// atanf / atan2f (std::arg, ofdm-processor.cpp:450,587-607): glibc 2.35 sysdeps/ieee754/flt-32/s_atanf.c
// and e_atan2f.c (fdlibm single-precision algorithms, Sun Microsystems 1993).  All operations are float.
__host__ __device__ __forceinline__ float fdlibm_atanf(float x)
{
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    union { float f; int32_t i; } u; u.f = x;
    const int32_t hx = u.i; const int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {                       /* |x| >= 2^25 */
        if (ix > 0x7f800000) return x + x;        /* NaN */
        if (hx > 0) return atanhi[3] + atanlo[3];
        return -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {                        /* |x| < 0.4375 */
        if (ix < 0x31000000) return x;            /* |x| < 2^-29 (1e30 + x > 1 always true) */
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {                    /* |x| < 1.1875 */
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const float zz = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return (hx < 0) ? -zz : zz;
}

__host__ __device__ __forceinline__ float fdlibm_atan2f(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
                pi_lo = -8.7422776573e-08f;
    union { float f; int32_t i; } ux, uy; ux.f = x; uy.f = y;
    const int32_t hx = ux.i, hy = uy.i;
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;   /* NaN */
    if (hx == 0x3f800000) return fdlibm_atanf(y);           /* x = 1.0 */
    const int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        switch (m) { case 0: case 1: return y; case 2: return pi + tiny; default: return -pi - tiny; }
    }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) { case 0: return pi_o_4 + tiny; case 1: return -pi_o_4 - tiny;
                         case 2: return 3.0f * pi_o_4 + tiny; default: return -3.0f * pi_o_4 - tiny; }
        } else {
            switch (m) { case 0: return 0.0f; case 1: return -0.0f; case 2: return pi + tiny; default: return -pi - tiny; }
        }
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = fdlibm_atanf(fabsf(y / x));
    switch (m) {
    case 0: return z;
    case 1: { union { float f; uint32_t i; } uz; uz.f = z; uz.i ^= 0x80000000u; return uz.f; }
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

// tables built once on the host with the host libm (same C library as the reference build), uploaded
struct Tables {
    const cf32* tw;          // 2048 forward KISS twiddles (kiss_fft.c:353-364)
    const cf32* ref;         // 2048 PRS reference (phasereference.cpp:45-51)
    const cf32* nco;         // 2 048 000 oscillator phasors (ofdm-processor.cpp:92-94)
    const int16_t* bin2soft; // 2048: FFT bin -> soft-bit index i (perm[i] == bin), -1 for unused bins
    const uint8_t* prbs_bytes; // 1152 bytes: PRBS packed MSB-first (fic-handler.cpp:62-71)
    const int32_t* osc_unsafe; // OSC_MAX_UNSAFE oscillator table indices (padded with -1) the unchecked conversion must not meet (osc_exact.h)
    int n_osc_unsafe;
};

} // namespace dabphy
