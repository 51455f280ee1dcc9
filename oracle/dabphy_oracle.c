/* oracle/dabphy_oracle.c -- TEST INFRASTRUCTURE: plain-C restatement of the welle.io Mode-I PHY hot path.
 *
 * Header comment of dabphy_oracle.h applies: checker only, never on the product path.  Every function
 * cites the reference file:line (relative to /root/reference/src) it restates.  Arithmetic is kept in the
 * reference's types and operation order (float vs double promotions included) so that results are
 * bit-identical to the reference built with g++ -O2 (no -ffast-math, no FMA contraction); this is
 * checked against the real reference in tests/test_oracle_vs_ref.py.  Compile with -ffp-contract=off.
 */
#include "dabphy_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------------- tables */

static orc_cf32 g_tw_fwd[ORC_TU];
static orc_cf32 g_reftable[ORC_TU];
static int16_t g_perm[ORC_K];
static int8_t g_pcodes[24][32];
static orc_cf32* g_nco = NULL;
static uint8_t g_prbs[9216];
static int g_init = 0;

/* phasetable.cpp:24-75 (EN 300 401 table for Mode I): 48 blocks of 32 carriers from k=-768 upward */
static const int8_t PRS_I[48] = {0,1,2,3,0,1,2,3,0,1,2,3,0,1,2,3,0,1,2,3,0,1,2,3, 0,3,2,1,0,3,2,1,0,3,2,1,0,3,2,1,0,3,2,1,0,3,2,1};
static const int8_t PRS_N[48] = {1,2,0,1,3,2,2,3,2,1,2,3,1,2,3,3,2,2,2,1,1,3,1,2, 3,1,1,1,2,2,1,0,2,2,3,3,0,2,1,3,3,3,3,0,3,0,1,1};
/* phasetable.cpp:140-155 */
static const int8_t PRS_H[4][16] = {
    {0,2,0,0,0,0,1,1,2,0,0,0,2,2,1,1}, {0,3,2,3,0,1,3,0,2,1,2,3,2,3,3,0},
    {0,0,0,2,0,2,1,3,2,2,0,2,2,0,1,3}, {0,1,2,1,0,3,3,2,2,3,2,1,2,1,3,2}};

/* phasetable.cpp:172-183: M_PI / 2.0f * (h + n) evaluated in double, returned as float */
static float prs_phi(int k)
{
    int blk = k < 0 ? (k + 768) / 32 : 24 + (k - 1) / 32;
    int kmin = k < 0 ? -768 + 32 * blk : 1 + 32 * (blk - 24);
    int h = PRS_H[PRS_I[blk]][(k - kmin) & 15];
    return (float)(M_PI / 2.0f * (h + PRS_N[blk]));
}

void orc_init(void)
{
    if (g_init) return;
    /* kiss_fft.c:353-364: phase = -2*pi*i/nfft in double, twiddle = (float)cos, (float)sin */
    for (int i = 0; i < ORC_TU; i++) {
        const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
        double phase = -2 * pi * i / ORC_TU;
        g_tw_fwd[i].re = (float)cos(phase);
        g_tw_fwd[i].im = (float)sin(phase);
    }
    /* phasereference.cpp:45-51: cos/sin of a FLOAT argument; the reference calls the C++ overloads
     * (std::cos(float) == cosf) -- verified against the compiled reference in tests */
    memset(g_reftable, 0, sizeof g_reftable);
    for (int i = 1; i <= ORC_K / 2; i++) {
        float phi = prs_phi(i);
        g_reftable[i].re = cosf(phi); g_reftable[i].im = sinf(phi);
        phi = prs_phi(-i);
        g_reftable[ORC_TU - i].re = cosf(phi); g_reftable[ORC_TU - i].im = sinf(phi);
    }
    /* freq-interleaver.cpp:35-59 with V1=511, lwb=256, upb=256+K */
    {
        int16_t tmp[ORC_TU]; int idx = 0;
        tmp[0] = 0;
        for (int i = 1; i < ORC_TU; i++) tmp[i] = (int16_t)((13 * tmp[i - 1] + 511) % ORC_TU);
        for (int i = 0; i < ORC_TU; i++) {
            if (tmp[i] == ORC_TU / 2) continue;
            if (tmp[i] < 256 || tmp[i] > 256 + ORC_K) continue;
            g_perm[idx++] = (int16_t)(tmp[i] - ORC_TU / 2);
        }
    }
    /* protTables.cpp:25-51 (EN 300 401 table 29) generated from its structure: PI_p has 8+p ones; the
     * first of each group of 4 is always set, further ones are added column by column in group order
     * 0,4,2,6,1,5,3,7 */
    {
        static const int order[8] = {0, 4, 2, 6, 1, 5, 3, 7};
        for (int p = 1; p <= 24; p++) {
            int8_t* v = g_pcodes[p - 1];
            memset(v, 0, 32);
            for (int g = 0; g < 8; g++) v[4 * g] = 1;
            for (int r = 0; r < p; r++) v[4 * order[r % 8] + 1 + r / 8] = 1;
        }
    }
    /* fic-handler.cpp:62-71 == energy_dispersal.h:39-49 */
    {
        uint8_t sr[9]; memset(sr, 1, 9);
        for (int i = 0; i < 9216; i++) {
            uint8_t b = sr[8] ^ sr[4];
            for (int j = 8; j > 0; j--) sr[j] = sr[j - 1];
            sr[0] = b; g_prbs[i] = b;
        }
    }
    g_init = 1;
}

const orc_cf32* orc_twiddles_fwd(void) { orc_init(); return g_tw_fwd; }
const orc_cf32* orc_prs_reftable(void) { orc_init(); return g_reftable; }
const int16_t* orc_freq_perm(void) { orc_init(); return g_perm; }
const int8_t* orc_pcodes(int idx) { orc_init(); return g_pcodes[idx]; }
const uint8_t* orc_prbs(int n) { (void)n; orc_init(); return g_prbs; }

const orc_cf32* orc_nco_table(void)
{
    /* ofdm-processor.cpp:92-94 */
    if (!g_nco) {
        orc_cf32* t = (orc_cf32*)malloc(sizeof(orc_cf32) * ORC_INPUT_RATE);
        for (int i = 0; i < ORC_INPUT_RATE; i++) {
            t[i].re = (float)cos(2.0 * M_PI * i / ORC_INPUT_RATE);
            t[i].im = (float)sin(2.0 * M_PI * i / ORC_INPUT_RATE);
        }
        g_nco = t;
    }
    return g_nco;
}

/* ------------------------------------------------------------------------------------------- complex */

/* std::complex<float> operator* (libgcc __mulsc3 fast path): (a+bi)(c+di) = (ac-bd) + (ad+bc)i */
static inline orc_cf32 cmul(orc_cf32 x, orc_cf32 y)
{
    orc_cf32 r;
    r.re = x.re * y.re - x.im * y.im;
    r.im = x.re * y.im + x.im * y.re;
    return r;
}
static inline orc_cf32 cconj(orc_cf32 x) { orc_cf32 r = {x.re, -x.im}; return r; }
static inline float l1_norm(orc_cf32 z) { return fabsf(z.re) + fabsf(z.im); }       /* MathHelper.h:48-51 */
static inline float cabs_(orc_cf32 z) { return hypotf(z.re, z.im); }                 /* std::abs(complex<float>) */
static inline float carg_(orc_cf32 z) { return atan2f(z.im, z.re); }                 /* std::arg(complex<float>) */

/* --------------------------------------------------------------------------------------------- KISS FFT */

/* kiss_fft.c:21-42 */
static void bfly2(orc_cf32* F, size_t fstride, const orc_cf32* tw, int m)
{
    orc_cf32* F2 = F + m; const orc_cf32* tw1 = tw;
    do {
        orc_cf32 t = cmul(*F2, *tw1);
        tw1 += fstride;
        F2->re = F->re - t.re; F2->im = F->im - t.im;
        F->re += t.re; F->im += t.im;
        ++F2; ++F;
    } while (--m);
}

/* kiss_fft.c:44-90 */
static void bfly4(orc_cf32* F, size_t fstride, const orc_cf32* tw, size_t m, int inverse)
{
    const orc_cf32 *tw1 = tw, *tw2 = tw, *tw3 = tw;
    orc_cf32 s[6];
    size_t k = m; const size_t m2 = 2 * m, m3 = 3 * m;
    do {
        s[0] = cmul(F[m], *tw1);
        s[1] = cmul(F[m2], *tw2);
        s[2] = cmul(F[m3], *tw3);
        s[5].re = F->re - s[1].re; s[5].im = F->im - s[1].im;
        F->re += s[1].re; F->im += s[1].im;
        s[3].re = s[0].re + s[2].re; s[3].im = s[0].im + s[2].im;
        s[4].re = s[0].re - s[2].re; s[4].im = s[0].im - s[2].im;
        F[m2].re = F->re - s[3].re; F[m2].im = F->im - s[3].im;
        tw1 += fstride; tw2 += fstride * 2; tw3 += fstride * 3;
        F->re += s[3].re; F->im += s[3].im;
        if (inverse) {
            F[m].re = s[5].re - s[4].im; F[m].im = s[5].im + s[4].re;
            F[m3].re = s[5].re + s[4].im; F[m3].im = s[5].im - s[4].re;
        } else {
            F[m].re = s[5].re + s[4].im; F[m].im = s[5].im - s[4].re;
            F[m3].re = s[5].re - s[4].im; F[m3].im = s[5].im + s[4].re;
        }
        ++F;
    } while (--k);
}

/* kiss_fft.c:232-300 unrolled for nfft=2048: kf_factor gives (4,512)(4,128)(4,32)(4,8)(4,2)(2,1).  The
 * recursion bottoms out copying f[...] in digit-reversed order; the butterflies then run leaf to root. */
void orc_fft2048(const orc_cf32* in, orc_cf32* out, int inverse)
{
    orc_init();
    orc_cf32 tw[ORC_TU];
    for (int i = 0; i < ORC_TU; i++) { tw[i] = g_tw_fwd[i]; if (inverse) tw[i].im = -tw[i].im; }   /* phase *= -1 */
    orc_cf32 buf[ORC_TU];
    for (int j1 = 0; j1 < 4; j1++) for (int j2 = 0; j2 < 4; j2++) for (int j3 = 0; j3 < 4; j3++)
    for (int j4 = 0; j4 < 4; j4++) for (int j5 = 0; j5 < 4; j5++) for (int j6 = 0; j6 < 2; j6++)
        buf[j1 * 512 + j2 * 128 + j3 * 32 + j4 * 8 + j5 * 2 + j6] = in[j1 + 4 * j2 + 16 * j3 + 64 * j4 + 256 * j5 + 1024 * j6];
    for (int b = 0; b < ORC_TU; b += 2) bfly2(buf + b, 1024, tw, 1);
    for (int b = 0; b < ORC_TU; b += 8) bfly4(buf + b, 256, tw, 2, inverse);
    for (int b = 0; b < ORC_TU; b += 32) bfly4(buf + b, 64, tw, 8, inverse);
    for (int b = 0; b < ORC_TU; b += 128) bfly4(buf + b, 16, tw, 32, inverse);
    for (int b = 0; b < ORC_TU; b += 512) bfly4(buf + b, 4, tw, 128, inverse);
    bfly4(buf, 1, tw, 512, inverse);
    if (inverse) {
        const float factor = 1.0f / (float)ORC_TU;                 /* fft.cpp:152-164 */
        for (int i = 0; i < ORC_TU; i++) { buf[i].re *= factor; buf[i].im *= factor; }
    }
    memcpy(out, buf, sizeof buf);
}

/* ------------------------------------------------------------------------------------------ time sync */

/* phasereference.cpp:73-256 */
int orc_find_index(const orc_cf32* v, int method, float* impulse)
{
    orc_init();
    orc_cf32 f[ORC_TU], r[ORC_TU];
    float ir_local[ORC_TU]; float* ir = impulse ? impulse : ir_local;
    float sum = 0;
    orc_fft2048(v, f, 0);
    for (int i = 0; i < ORC_TU; i++) r[i] = cmul(f[i], cconj(g_reftable[i]));
    orc_fft2048(r, r, 1);
    if (method == 0) {                                              /* StrongestPeak :99-129 */
        int maxIndex = -1; float max = -10000;
        for (int i = 0; i < ORC_TU; i++) {
            const float value = cabs_(r[i]);
            sum += value; ir[i] = value;
            if (value > max) { maxIndex = i; max = value; }
        }
        if (sum == 0) return -1;
        if (max < 3 * sum / ORC_TU) return (int)(-fabsf(max * ORC_TU / sum) - 1);
        return maxIndex;
    }
    if (method == 1) {                                              /* EarliestPeakWithBinning :125-211 */
        enum { BIN = 20, NB = (ORC_TU - 1) / BIN };                 /* i + bin_size < Tu: 102 bins, samples 2040..2047 are never looked at */
        float val[NB]; int idx[NB], order[NB];
        float mean = 0;
        for (int k = 0; k < NB; k++) {
            val[k] = 0; idx[k] = -1;
            for (int j = 0; j < BIN; j++) {
                const float value = cabs_(r[BIN * k + j]);
                mean += value; ir[BIN * k + j] = value;
                if (value > val[k]) { val[k] = value; idx[k] = BIN * k + j; }
            }
        }
        for (int i = BIN * NB; i < ORC_TU; i++) ir[i] = 0;          /* the member buffer keeps its initial zeros there */
        mean /= ORC_TU;
        /* std::sort by value, descending (ties -- exact float equality of two bin maxima -- in bin order here) */
        for (int k = 0; k < NB; k++) order[k] = k;
        for (int a = 1; a < NB; a++) { int o = order[a], bpos = a; while (bpos > 0 && val[order[bpos - 1]] < val[o]) { order[bpos] = order[bpos - 1]; bpos--; } order[bpos] = o; }
        const int peak_index = idx[order[0]];
        /* not farther than 500 from the highest peak, then the 4 highest, then those above 3 * mean; min_element over the
         * survivors (an all-zero input leaves bins of index -1 that survive: 0 < 0 is false) */
        {
            int found = 0, mn = 0, kept = 0;
            for (int a = 0; a < NB && kept < 4; a++) {
                const int k = order[a];
                if (abs(idx[k] - peak_index) > 500) continue;
                kept++;
                if (val[k] < 3 * mean) continue;
                if (!found || idx[k] < mn) { mn = idx[k]; found = 1; }
            }
            return found ? mn : -1;
        }
    }
    /* ThresholdBeforePeak :212-252 */
    for (int i = 0; i < ORC_TU; i++) { const float a = cabs_(r[i]); ir[i] = a; sum += a; }
    const int windowsize = 100;
    float pa_local[ORC_TU];
    memset(pa_local, 0, sizeof pa_local);
    float global_max = -10000;
    for (int i = 0; i + windowsize < ORC_TU; i++) {
        float max = -10000;
        for (int j = 0; j < windowsize; j++) { const float value = ir[i + j]; if (value > max) max = value; }
        pa_local[i] = max;
        if (max > global_max) global_max = max;
    }
    if (global_max > 3 * sum / ORC_TU) {
        const float thresh = global_max / 2;
        for (int i = 0; i + windowsize < ORC_TU; i++)
            if (pa_local[i + windowsize] > thresh) return i;
    }
    return -1;
}

/* ofdm-processor.cpp:537-616: method = FreqsyncMethod (0 GetMiddle, 1 CorrelatePRS, 2 PatternOfZeros) */
int orc_coarse_prs_method(const orc_cf32* v, int method)
{
    orc_init();
    orc_cf32 f[ORC_TU];
    orc_fft2048(v, f, 0);
    if (method == 0) {                                               /* getMiddle :618-644 (abs of a complex: std::abs = hypotf) */
        float sum = 0, oldMax = 0; int maxIndex = 0;
        for (int i = 40; i < 1536 + 40; i++) sum += cabs_(f[(ORC_TU / 2 + i) % ORC_TU]);
        for (int i = 40; i < ORC_TU - (1536 - 40); i++) {
            sum -= cabs_(f[(ORC_TU / 2 + i) % ORC_TU]);
            sum += cabs_(f[(ORC_TU / 2 + i + 1536) % ORC_TU]);
            if (sum > oldMax) { sum = oldMax; maxIndex = i; }         /* sic: the running sum is reset, oldMax stays 0 */
        }
        return maxIndex - (ORC_TU - 1536) / 2;
    }
    if (method == 1) {                                               /* CorrelatePRS :547-581 */
        float refArg[24], cv[72 + 24];
        for (int i = 0; i < 24; i++) {                                /* :97-101 */
            const orc_cf32 z = cmul(g_reftable[(ORC_TU + i) % ORC_TU], cconj(g_reftable[(ORC_TU + i + 1) % ORC_TU]));
            refArg[i] = carg_(z);
        }
        for (int i = 0; i < 72 + 24; i++) {
            const int base = ORC_TU - 36 + i;
            cv[i] = carg_(cmul(f[base % ORC_TU], cconj(f[(base + 1) % ORC_TU])));
        }
        float MMax = 0; int index = 100;
        for (int i = 0; i < 72; i++) {
            float sum = 0;
            for (int j = 0; j < 24; j++) {
                sum += (float)abs((int)(refArg[j] * cv[i + j]));      /* ::abs(int) again: the product is truncated first */
                if (sum > MMax) { MMax = sum; index = i; }
            }
        }
        return ORC_TU - 36 + index - ORC_TU;
    }
    int index = 100; float Mmin = 1000;
#define FB(i) f[(i) % ORC_TU]
#define ARGD(a, b) carg_(cmul(FB(a), cconj(FB(b))))
    /* NOTE (reference quirk, pinned by disassembly of the -O2 build and by tests): the unqualified abs()
     * calls of ofdm-processor.cpp:587-607 bind to ::abs(int), so every argument is first truncated to int
     * (cvttsd2si / cvttss2si).  a1, a2, b1 are therefore abs(abs((int)(arg/pi)) - 1) and the other terms
     * are abs((int)arg) in {0..3}. */
    for (int i = ORC_TU - 36; i < ORC_TU + 36; i++) {
        float a1 = (float)abs(abs((int)(ARGD(i + 1, i + 2) / M_PI)) - 1);
        float a2 = (float)abs(abs((int)(ARGD(i + 2, i + 3) / M_PI)) - 1);
        float a3 = (float)abs((int)ARGD(i + 3, i + 4));
        float a4 = (float)abs((int)ARGD(i + 4, i + 5));
        float a5 = (float)abs((int)ARGD(i + 5, i + 6));
        float b1 = (float)abs(abs((int)(ARGD(i + 16 + 1, i + 16 + 3) / M_PI)) - 1);
        float b2 = (float)abs((int)ARGD(i + 16 + 3, i + 16 + 4));
        float b3 = (float)abs((int)ARGD(i + 16 + 4, i + 16 + 5));
        float b4 = (float)abs((int)ARGD(i + 16 + 5, i + 16 + 6));
        float sum = a1 + a2 + a3 + a4 + a5 + b1 + b2 + b3 + b4;
        if (sum < Mmin) { Mmin = sum; index = i; }
    }
#undef ARGD
#undef FB
    return index - ORC_TU;
}

int orc_coarse_prs(const orc_cf32* v) { return orc_coarse_prs_method(v, 2); }

/* -------------------------------------------------------------------------------------------- demod */

void orc_demod_reset(orc_demod_state* st) { memset(st, 0, sizeof *st); }

/* MathHelper.h:43-46 */
static float get_db_over_256(float x) { return (float)(20 * log10((x + 1.0f) / 256.0f)); }

/* ofdm-decoder.cpp:240-266 (method 1) */
static int16_t get_snr(const orc_cf32* v)
{
    float noise = 0, signal = 0;
    const int T_u = ORC_TU, K = ORC_K;
    int16_t low = T_u / 2 - K / 2, high = low + K, i;
    for (i = 70; i < low - 20; i++) noise += cabs_(v[(T_u / 2 + i) % T_u]);
    for (i = high + 20; i < high + 120; i++) noise += cabs_(v[(T_u / 2 + i) % T_u]);
    noise /= (low - 90 + 100);
    for (i = T_u / 2 - K / 4; i < T_u / 2 + K / 4; i++) signal += cabs_(v[(T_u / 2 + i) % T_u]);
    const float dB_signal = get_db_over_256(signal / (K / 2));
    const float dB_noise = get_db_over_256(noise);
    const float snr_new = dB_signal - dB_noise;
    return (int16_t)snr_new;
}

/* ofdm-decoder.cpp:144-166 */
int orc_demod_prs(orc_demod_state* st, const orc_cf32* prs, float* snr_out)
{
    orc_cf32 f[ORC_TU];
    orc_fft2048(prs, f, 0);
    st->snr = (float)(0.7 * st->snr + 0.3 * get_snr(f));
    int fired = 0;
    if (++st->snr_count > 10) { if (snr_out) *snr_out = st->snr; st->snr_count = 0; fired = 1; }
    memcpy(st->phase_ref, f, sizeof f);
    return fired;
}

/* ofdm-decoder.cpp:175-230 */
void orc_demod_symbol(orc_demod_state* st, const orc_cf32* sym, int8_t* soft, orc_cf32* con)
{
    orc_cf32 f[ORC_TU];
    int nc = 0;
    orc_fft2048(sym + ORC_TG, f, 0);
    for (int16_t i = 0; i < ORC_K; i++) {
        int16_t index = g_perm[i];
        if (index < 0) index += ORC_TU;
        const orc_cf32 r1 = cmul(f[index], cconj(st->phase_ref[index]));
        st->phase_ref[index] = f[index];
        const float ab1 = 127.0f / l1_norm(r1);
        /* float -> int8_t conversion: C truncation; NaN (r1 == 0) converts to 0 with cvttss2si on x86-64 */
        float vr = -r1.re * ab1, vi = -r1.im * ab1;
        soft[i] = (vr != vr) ? 0 : (int8_t)vr;
        soft[ORC_K + i] = (vi != vi) ? 0 : (int8_t)vi;
        if (i % 96 == 0 && con) con[nc++] = r1;
    }
}

/* ------------------------------------------------------------------------------------------- Viterbi */

/* viterbi.cpp:36,170-177: Branchtab[i*32+state] = parity((2*state) & poly_i) ? 255 : 0 */
static uint16_t g_branchtab[128];
static int g_vit_init = 0;
static int parity_(int x) { x ^= x >> 16; x ^= x >> 8; x ^= x >> 4; x ^= x >> 2; x ^= x >> 1; return x & 1; }
static void vit_init(void)
{
    static const int polys[4] = {0155, 0117, 0123, 0155};
    if (g_vit_init) return;
    for (int state = 0; state < 32; state++)
        for (int i = 0; i < 4; i++)
            g_branchtab[i * 32 + state] = parity_((2 * state) & polys[i]) ? 255 : 0;
    g_vit_init = 1;
}

/* viterbi.cpp:227-354 */
void orc_viterbi(const int8_t* input, int nbits, uint8_t* output)
{
    vit_init();
    const int nsteps = nbits + 6;
    uint16_t* symbols = (uint16_t*)malloc(sizeof(uint16_t) * 4 * nsteps);
    uint32_t* dec = (uint32_t*)calloc(2 * (size_t)nsteps, sizeof(uint32_t));
    uint8_t* data = (uint8_t*)calloc((size_t)nsteps / 8 + 2, 1);
    uint16_t m1[64], m2[64]; uint16_t *oldm = m1, *newm = m2;
    for (int i = 0; i < 64; i++) m1[i] = 63;                        /* init_viterbi :342-354 */
    m1[0] = 0;
    for (int i = 0; i < 4 * nsteps; i++) {                           /* :233-238 */
        int16_t temp = (int16_t)input[i] + 127;
        if (temp < 0) temp = 0;
        if (temp > 255) temp = 255;
        symbols[i] = (uint16_t)temp;
    }
    for (int s = 0; s < nsteps; s++) {                               /* update_viterbi_blk_GENERIC :285-309 */
        for (int i = 0; i < 32; i++) {                               /* BFLY :248-279 */
            uint16_t metric = 0;
            for (int j = 0; j < 4; j++) metric += (g_branchtab[i + j * 32] ^ symbols[s * 4 + j]);
            const uint16_t max = 4 * 255;
            uint16_t m0 = oldm[i] + metric;
            uint16_t m1_ = oldm[i + 32] + (max - metric);
            uint16_t m2_ = oldm[i] + (max - metric);
            uint16_t m3 = oldm[i + 32] + metric;
            int decision0 = ((int32_t)(m0 - m1_)) > 0;
            int decision1 = ((int32_t)(m2_ - m3)) > 0;
            newm[2 * i] = decision0 ? m1_ : m0;
            newm[2 * i + 1] = decision1 ? m3 : m2_;
            dec[i / 16 + s * 2] |= (uint32_t)(decision0 | decision1 << 1) << ((2 * i) & 31);
        }
        if (newm[0] > 137) {                                         /* renormalize :104-120 */
            uint16_t min = newm[0];
            for (int i = 0; i < 64; i++) if (min > newm[i]) min = newm[i];
            for (int i = 0; i < 64; i++) newm[i] -= min;
        }
        uint16_t* t = oldm; oldm = newm; newm = t;
    }
    {                                                                /* chainback_viterbi :313-339 */
        unsigned endstate = 0; int n = nbits;
        const uint32_t* d = dec + 2 * 6;
        while (n-- != 0) {
            int k = (d[2 * n + ((endstate >> 2) / 32)] >> ((endstate >> 2) % 32)) & 1;
            endstate = (endstate >> 1) | (k << 7);
            data[n >> 3] = (uint8_t)endstate;
        }
    }
    for (int i = 0; i < nbits; i++) output[i] = (data[i >> 3] >> (7 - (i & 7))) & 1;
    free(symbols); free(dec); free(data);
}

/* ---------------------------------------------------------------------------------------- protection */

static const uint8_t PI_X[24] = {1,1,0,0, 1,1,0,0, 1,1,0,0, 1,1,0,0, 1,1,0,0, 1,1,0,0};   /* fic-handler.cpp:39-42 */

static void prot_finish(orc_prot* p)
{
    int n = 0, blocks = 0;
    for (int s = 0; s < 4; s++) if (p->L[s] > 0) { n += p->L[s] * 4 * (8 + p->PI[s]); blocks += p->L[s]; }
    p->n_in = n + 12;
    (void)blocks;
}

int orc_prot_fic(orc_prot* p)
{
    memset(p, 0, sizeof *p);
    p->nbits = 768; p->L[0] = 21; p->PI[0] = 16; p->L[1] = 3; p->PI[1] = 15;
    prot_finish(p); return 0;
}

/* eep-protection.cpp:32-113 */
int orc_prot_eep(orc_prot* p, int bitRate, int profile_b, int level)
{
    memset(p, 0, sizeof *p);
    p->nbits = 24 * bitRate;
    if (!profile_b) {
        switch (level) {
        case 1: p->L[0] = 6 * bitRate / 8 - 3; p->L[1] = 3; p->PI[0] = 24; p->PI[1] = 23; break;
        case 2:
            if (bitRate == 8) { p->L[0] = 5; p->L[1] = 1; p->PI[0] = 13; p->PI[1] = 12; }
            else { p->L[0] = 2 * bitRate / 8 - 3; p->L[1] = 4 * bitRate / 8 + 3; p->PI[0] = 14; p->PI[1] = 13; }
            break;
        case 3: p->L[0] = 6 * bitRate / 8 - 3; p->L[1] = 3; p->PI[0] = 8; p->PI[1] = 7; break;
        case 4: p->L[0] = 4 * bitRate / 8 - 3; p->L[1] = 2 * bitRate / 8 + 3; p->PI[0] = 3; p->PI[1] = 2; break;
        default: return -1;
        }
    } else {
        p->L[0] = 24 * bitRate / 32 - 3; p->L[1] = 3;
        switch (level) {
        case 4: p->PI[0] = 2; p->PI[1] = 1; break;
        case 3: p->PI[0] = 4; p->PI[1] = 3; break;
        case 2: p->PI[0] = 6; p->PI[1] = 5; break;
        case 1: p->PI[0] = 10; p->PI[1] = 9; break;
        default: return -1;
        }
    }
    prot_finish(p); return 0;
}

/* uep-protection.cpp:27-118 (the reference's own table, its row {80,1,...,24,7,12,18} included) merged
 * with dab-constants.cpp:45-109 (CU size per table index): {bitrate, level, CU, L1..L4, PI1..PI4} */
static const int16_t UEP_TAB[64][11] = {
    {32,5,16, 3,4,17,0, 5,3,2,-1}, {32,4,21, 3,3,18,0, 11,6,5,-1}, {32,3,24, 3,4,14,3, 15,9,6,8},
    {32,2,29, 3,4,14,3, 22,13,8,13}, {32,1,35, 3,5,13,3, 24,17,12,17},
    {48,5,24, 4,3,26,3, 5,4,2,3}, {48,4,29, 3,4,26,3, 9,6,4,6}, {48,3,35, 3,4,26,3, 15,10,6,9},
    {48,2,42, 3,4,26,3, 24,14,8,15}, {48,1,52, 3,5,25,3, 24,18,13,18},
    {56,5,29, 6,10,23,3, 5,4,2,3}, {56,4,35, 6,10,23,3, 9,6,4,5}, {56,3,42, 6,12,21,3, 16,7,6,9},
    {56,2,52, 6,10,23,3, 23,13,8,13},
    {64,5,32, 6,9,31,2, 5,3,2,3}, {64,4,42, 6,9,33,0, 11,6,5,-1}, {64,3,48, 6,12,27,3, 16,8,6,9},
    {64,2,58, 6,10,29,3, 23,13,8,13}, {64,1,70, 6,11,28,3, 24,18,12,18},
    {80,5,40, 6,10,41,3, 6,3,2,3}, {80,4,52, 6,10,41,3, 11,6,5,6}, {80,3,58, 6,11,40,3, 16,8,6,7},
    {80,2,70, 6,10,41,3, 23,13,8,13}, {80,1,84, 6,10,41,3, 24,7,12,18},
    {96,5,48, 7,9,53,3, 5,4,2,4}, {96,4,58, 7,10,52,3, 9,6,4,6}, {96,3,70, 6,12,51,3, 16,9,6,10},
    {96,2,84, 6,10,53,3, 22,12,9,12}, {96,1,104, 6,13,50,3, 24,18,13,19},
    {112,5,58, 14,17,50,3, 5,4,2,5}, {112,4,70, 11,21,49,3, 9,6,4,8}, {112,3,84, 11,23,47,3, 16,8,6,9},
    {112,2,104, 11,21,49,3, 23,12,9,14},
    {128,5,64, 12,19,62,3, 5,3,2,4}, {128,4,84, 11,21,61,3, 11,6,5,7}, {128,3,96, 11,22,60,3, 16,9,6,10},
    {128,2,116, 11,21,61,3, 22,12,9,14}, {128,1,140, 11,20,62,3, 24,17,13,19},
    {160,5,80, 11,19,87,3, 5,4,2,4}, {160,4,104, 11,23,83,3, 11,6,5,9}, {160,3,116, 11,24,82,3, 16,8,6,11},
    {160,2,140, 11,21,85,3, 22,11,9,13}, {160,1,168, 11,22,84,3, 24,18,12,19},
    {192,5,96, 11,20,110,3, 6,4,2,5}, {192,4,116, 11,22,108,3, 10,6,4,9}, {192,3,140, 11,24,106,3, 16,10,6,11},
    {192,2,168, 11,20,110,3, 22,13,9,13}, {192,1,208, 11,21,109,3, 24,20,13,24},
    {224,5,116, 12,22,131,3, 8,6,2,6}, {224,4,140, 12,26,127,3, 12,8,4,11}, {224,3,168, 11,20,134,3, 16,10,7,9},
    {224,2,208, 11,22,132,3, 24,16,10,15}, {224,1,232, 11,24,130,3, 24,20,12,20},
    {256,5,128, 11,24,154,3, 6,5,2,5}, {256,4,168, 11,24,154,3, 12,9,5,10}, {256,3,192, 11,27,151,3, 16,10,7,10},
    {256,2,232, 11,22,156,3, 24,14,10,13}, {256,1,280, 11,26,152,3, 24,19,14,18},
    {320,5,160, 11,26,200,3, 8,5,2,6}, {320,4,208, 11,25,201,3, 13,9,5,10}, {320,2,280, 11,26,200,3, 24,17,9,17},
    {384,5,192, 11,27,247,3, 8,6,2,7}, {384,3,280, 11,24,250,3, 16,9,7,10}, {384,1,416, 12,28,245,3, 24,20,14,23}};

/* row `idx` of the short-form table (dab-constants.cpp:45-109): what FIG 0/1's 6-bit table index selects */
int orc_uep_table(int idx, int* bitrate, int* level, int* size_cu)
{
    if (idx < 0 || idx >= 64) return -1;
    *bitrate = UEP_TAB[idx][0]; *level = UEP_TAB[idx][1]; *size_cu = UEP_TAB[idx][2];
    return 0;
}

/* uep-protection.cpp:120-167 (unknown pair falls back to row 1 like the reference) */
int orc_prot_uep(orc_prot* p, int bitRate, int level)
{
    memset(p, 0, sizeof *p);
    p->nbits = 24 * bitRate;
    int idx = -1;
    for (int i = 0; i < 64; i++) if (UEP_TAB[i][0] == bitRate && UEP_TAB[i][1] == level) { idx = i; break; }
    if (idx < 0) idx = 1;
    for (int s = 0; s < 4; s++) { p->L[s] = UEP_TAB[idx][3 + s]; p->PI[s] = UEP_TAB[idx][7 + s] < 0 ? 0 : UEP_TAB[idx][7 + s]; }
    prot_finish(p); return 0;
}

/* fic-handler.cpp:158-191 / eep-protection.cpp:115-148 / uep-protection.cpp:169-233 */
void orc_depuncture(const orc_prot* p, const int8_t* in, int8_t* out)
{
    orc_init();
    int ic = 0, vc = 0;
    memset(out, 0, (size_t)4 * p->nbits + 24);
    for (int s = 0; s < 4; s++) {
        for (int i = 0; i < p->L[s]; i++)
            for (int j = 0; j < 128; j++) {
                if (g_pcodes[p->PI[s] - 1][j % 32] != 0) out[vc] = in[ic++];
                vc++;
            }
    }
    for (int i = 0; i < 24; i++) { if (PI_X[i] != 0) out[vc] = in[ic++]; vc++; }
}

/* MathHelper.h:53-80, returns 1 when the CRC is valid */
int orc_crc16_bits(const uint8_t* in, int size)
{
    static const uint8_t poly[] = {0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0};
    uint8_t b[16]; memset(b, 1, 16);
    for (int i = 0; i < size; i++) {
        uint8_t d = in[i];
        if (i >= size - 16) d ^= 1;
        if ((b[0] ^ d) == 1) { for (int f = 0; f < 15; f++) b[f] = poly[f] ^ b[f + 1]; b[15] = 1; }
        else { memmove(&b[0], &b[1], 15); b[15] = 0; }
    }
    uint16_t crc = 0;
    for (int i = 0; i < 16; i++) crc |= (uint16_t)(b[i] << i);
    return crc == 0;
}

/* fic-handler.cpp:111-230 */
void orc_fic_decode(const int8_t* soft, uint8_t* bits, uint8_t* ok, int* ratio)
{
    orc_prot p; orc_prot_fic(&p);
    int8_t vb[3072 + 24];
    for (int ficno = 0; ficno < 4; ficno++) {
        uint8_t* out = bits + 768 * ficno;
        orc_depuncture(&p, soft + 2304 * ficno, vb);
        orc_viterbi(vb, 768, out);
        for (int i = 0; i < 768; i++) out[i] ^= g_prbs[i];
        for (int i = 0; i < 3; i++) {
            int valid = orc_crc16_bits(out + 256 * i, 256);
            ok[3 * ficno + i] = (uint8_t)valid;
            if (valid) { if (*ratio < 10) (*ratio)++; }
            else if (*ratio > 0) (*ratio)--;
        }
    }
}

/* ----------------------------------------------------------------------------------------- sub-channel */

int orc_subch_init(orc_subch* s, const orc_prot* prot, int length_cu)
{
    memset(s, 0, sizeof *s);
    s->prot = *prot; s->frag = length_cu * 64;
    s->hist = (int8_t*)calloc(16, (size_t)s->frag);
    s->tmp = (int8_t*)calloc(1, (size_t)s->frag);
    s->vit = (int8_t*)calloc(1, (size_t)4 * prot->nbits + 24);
    s->bits = (uint8_t*)calloc(1, (size_t)prot->nbits);
    return 0;
}
void orc_subch_free(orc_subch* s) { free(s->hist); free(s->tmp); free(s->vit); free(s->bits); memset(s, 0, sizeof *s); }

/* dab-audio.cpp:113-164, energy_dispersal.h:35-54, decoder_adapter.cpp:55-67 */
int orc_subch_process(orc_subch* s, const int8_t* data, uint8_t* out)
{
    static const int16_t map[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
    for (int i = 0; i < s->frag; i++) {
        s->tmp[i] = s->hist[(size_t)((s->idx + map[i & 15]) & 15) * s->frag + i];
        s->hist[(size_t)s->idx * s->frag + i] = data[i];
    }
    s->idx = (s->idx + 1) & 15;
    if (s->count <= 15) { s->count++; return 0; }
    orc_depuncture(&s->prot, s->tmp, s->vit);
    orc_viterbi(s->vit, s->prot.nbits, s->bits);
    for (int i = 0; i < s->prot.nbits; i++) s->bits[i] ^= g_prbs[i];
    const int len = s->prot.nbits / 8;
    for (int i = 0; i < len; i++) {
        uint8_t b = 0;
        for (int j = 0; j < 8; j++) b = (uint8_t)((b << 1) | (s->bits[8 * i + j] & 1));
        out[i] = b;
    }
    return len;
}

/* ------------------------------------------------------------------------------------- Reed-Solomon */

/* libs/fec/init_rs.h:6-103 with (symsize 8, gfpoly 0x11D, fcr 0, prim 1, nroots 10, pad 135) */
static uint8_t rs_alpha_to[256], rs_index_of[256], rs_genpoly[11];
static int rs_ready = 0;
#define RS_NN 255
#define RS_NROOTS 10
#define RS_PAD 135
#define RS_A0 255
static int rs_modnn(int x) { while (x >= RS_NN) { x -= RS_NN; x = (x >> 8) + (x & RS_NN); } return x; }
static void rs_init(void)
{
    if (rs_ready) return;
    int sr = 1;
    rs_index_of[0] = RS_A0; rs_alpha_to[RS_A0] = 0;
    for (int i = 0; i < RS_NN; i++) {
        rs_index_of[sr] = (uint8_t)i; rs_alpha_to[i] = (uint8_t)sr;
        sr <<= 1; if (sr & 256) sr ^= 0x11D; sr &= RS_NN;
    }
    rs_genpoly[0] = 1;
    for (int i = 0, root = 0; i < RS_NROOTS; i++, root += 1) {
        rs_genpoly[i + 1] = 1;
        for (int j = i; j > 0; j--) {
            if (rs_genpoly[j] != 0) rs_genpoly[j] = rs_genpoly[j - 1] ^ rs_alpha_to[rs_modnn(rs_index_of[rs_genpoly[j]] + root)];
            else rs_genpoly[j] = rs_genpoly[j - 1];
        }
        rs_genpoly[0] = rs_alpha_to[rs_modnn(rs_index_of[rs_genpoly[0]] + root)];
    }
    for (int i = 0; i <= RS_NROOTS; i++) rs_genpoly[i] = rs_index_of[rs_genpoly[i]];
    rs_ready = 1;
}

/* libs/fec/encode_rs.h (test-signal generation only) */
void orc_rs_encode120(const uint8_t* data, uint8_t* parity)
{
    rs_init();
    memset(parity, 0, RS_NROOTS);
    for (int i = 0; i < RS_NN - RS_NROOTS - RS_PAD; i++) {
        uint8_t feedback = rs_index_of[data[i] ^ parity[0]];
        if (feedback != RS_A0)
            for (int j = 1; j < RS_NROOTS; j++) parity[j] ^= rs_alpha_to[rs_modnn(feedback + rs_genpoly[RS_NROOTS - j])];
        memmove(&parity[0], &parity[1], RS_NROOTS - 1);
        if (feedback != RS_A0) parity[RS_NROOTS - 1] = rs_alpha_to[rs_modnn(feedback + rs_genpoly[0])];
        else parity[RS_NROOTS - 1] = 0;
    }
}

/* libs/fec/decode_rs.h:71-298, no erasures; returns count (-1 = uncorrectable), fills loc[] */
static int rs_decode120(uint8_t* data, int* eras_pos)
{
    int deg_lambda, el, deg_omega, i, j, r, k;
    uint8_t u, q, tmp, num1, num2, den, discr_r;
    uint8_t lambda[RS_NROOTS + 1], s[RS_NROOTS], b[RS_NROOTS + 1], t[RS_NROOTS + 1], omega[RS_NROOTS + 1];
    uint8_t root[RS_NROOTS], reg[RS_NROOTS + 1], loc[RS_NROOTS];
    int syn_error, count;
    (void)u;
    for (i = 0; i < RS_NROOTS; i++) s[i] = data[0];
    for (j = 1; j < RS_NN - RS_PAD; j++)
        for (i = 0; i < RS_NROOTS; i++) {
            if (s[i] == 0) s[i] = data[j];
            else s[i] = data[j] ^ rs_alpha_to[rs_modnn(rs_index_of[s[i]] + (0 + i) * 1)];
        }
    syn_error = 0;
    for (i = 0; i < RS_NROOTS; i++) { syn_error |= s[i]; s[i] = rs_index_of[s[i]]; }
    if (!syn_error) { count = 0; goto finish; }
    memset(&lambda[1], 0, RS_NROOTS); lambda[0] = 1;
    for (i = 0; i < RS_NROOTS + 1; i++) b[i] = rs_index_of[lambda[i]];
    r = 0; el = 0;
    while (++r <= RS_NROOTS) {
        discr_r = 0;
        for (i = 0; i < r; i++)
            if ((lambda[i] != 0) && (s[r - i - 1] != RS_A0)) discr_r ^= rs_alpha_to[rs_modnn(rs_index_of[lambda[i]] + s[r - i - 1])];
        discr_r = rs_index_of[discr_r];
        if (discr_r == RS_A0) { memmove(&b[1], b, RS_NROOTS); b[0] = RS_A0; }
        else {
            t[0] = lambda[0];
            for (i = 0; i < RS_NROOTS; i++) {
                if (b[i] != RS_A0) t[i + 1] = lambda[i + 1] ^ rs_alpha_to[rs_modnn(discr_r + b[i])];
                else t[i + 1] = lambda[i + 1];
            }
            if (2 * el <= r - 1) {
                el = r - el;
                for (i = 0; i <= RS_NROOTS; i++) b[i] = (lambda[i] == 0) ? RS_A0 : (uint8_t)rs_modnn(rs_index_of[lambda[i]] - discr_r + RS_NN);
            } else { memmove(&b[1], b, RS_NROOTS); b[0] = RS_A0; }
            memcpy(lambda, t, RS_NROOTS + 1);
        }
    }
    deg_lambda = 0;
    for (i = 0; i < RS_NROOTS + 1; i++) { lambda[i] = rs_index_of[lambda[i]]; if (lambda[i] != RS_A0) deg_lambda = i; }
    memcpy(&reg[1], &lambda[1], RS_NROOTS);
    count = 0;
    for (i = 1, k = 1 - 1; i <= RS_NN; i++, k = rs_modnn(k + 1)) {        /* iprim = 1 */
        q = 1;
        for (j = deg_lambda; j > 0; j--)
            if (reg[j] != RS_A0) { reg[j] = (uint8_t)rs_modnn(reg[j] + j); q ^= rs_alpha_to[reg[j]]; }
        if (q != 0) continue;
        root[count] = (uint8_t)i; loc[count] = (uint8_t)k;
        if (++count == deg_lambda) break;
    }
    if (deg_lambda != count) { count = -1; goto finish; }
    deg_omega = deg_lambda - 1;
    for (i = 0; i <= deg_omega; i++) {
        tmp = 0;
        for (j = i; j >= 0; j--)
            if ((s[i - j] != RS_A0) && (lambda[j] != RS_A0)) tmp ^= rs_alpha_to[rs_modnn(s[i - j] + lambda[j])];
        omega[i] = rs_index_of[tmp];
    }
    for (j = count - 1; j >= 0; j--) {
        num1 = 0;
        for (i = deg_omega; i >= 0; i--)
            if (omega[i] != RS_A0) num1 ^= rs_alpha_to[rs_modnn(omega[i] + i * root[j])];
        num2 = rs_alpha_to[rs_modnn(root[j] * (0 - 1) + RS_NN)];
        den = 0;
        for (i = (deg_lambda < RS_NROOTS - 1 ? deg_lambda : RS_NROOTS - 1) & ~1; i >= 0; i -= 2)
            if (lambda[i + 1] != RS_A0) den ^= rs_alpha_to[rs_modnn(lambda[i + 1] + i * root[j])];
        if (num1 != 0 && loc[j] >= RS_PAD)
            data[loc[j] - RS_PAD] ^= rs_alpha_to[rs_modnn(rs_index_of[num1] + rs_index_of[num2] + RS_NN - rs_index_of[den])];
    }
finish:
    if (eras_pos) for (i = 0; i < count; i++) eras_pos[i] = loc[i];
    return count;
}

/* dabplus_decoder.cpp:326-359 */
void orc_rs_superframe(uint8_t* sf, int sf_len, int* total_corr, int* uncorr)
{
    rs_init();
    int subch_index = sf_len / 120;
    uint8_t pkt[120]; int corr_pos[10];
    *total_corr = 0; *uncorr = 0;
    for (int i = 0; i < subch_index; i++) {
        for (int pos = 0; pos < 120; pos++) pkt[pos] = sf[pos * subch_index + i];
        int c = rs_decode120(pkt, corr_pos);
        if (c == -1) *uncorr = 1; else *total_corr += c;
        for (int j = 0; j < c; j++) {
            int pos = corr_pos[j] - 135;
            if (pos < 0) continue;
            sf[pos * subch_index + i] = pkt[pos];
        }
    }
}

/* ------------------------------------------------------------------------------------------ DAB+ superframe filter */

/* CalcCRC (tools.cpp:41-72): MSB-first CRC-16, optional initial / final inversion */
uint16_t orc_crc16(const uint8_t* data, int len, int initial_invert, int final_invert, uint16_t poly)
{
    uint16_t crc = initial_invert ? 0xFFFF : 0x0000;
    for (int o = 0; o < len; o++) {
        crc ^= (uint16_t)(data[o] << 8);
        for (int i = 0; i < 8; i++) crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ poly) : (uint16_t)(crc << 1);
    }
    return final_invert ? (uint16_t)~crc : crc;
}

/* SuperframeFilter::CheckSync (dabplus_decoder.cpp:160-213) on a corrected superframe; fills num_aus / au_start */
static int sf_check_sync(const uint8_t* sf, int sf_len, int* num_aus_out, int* au_start)
{
    if (sf[3] == 0x00 && sf[4] == 0x00) return 0;
    uint16_t crc_stored = (uint16_t)(sf[0] << 8 | sf[1]);
    uint16_t crc_calced = orc_crc16(sf + 2, 9, 0, 0, 0x782F);                 /* CalcCRC_FIRE_CODE, tools.cpp:37 */
    if (crc_stored != crc_calced) return 0;
    int dac_rate = sf[2] & 0x40, sbr_flag = sf[2] & 0x20;
    int num_aus = dac_rate ? (sbr_flag ? 3 : 6) : (sbr_flag ? 2 : 4);
    au_start[0] = dac_rate ? (sbr_flag ? 6 : 11) : (sbr_flag ? 5 : 8);
    au_start[num_aus] = sf_len / 120 * 110;
    au_start[1] = sf[3] << 4 | sf[4] >> 4;
    if (num_aus >= 3) au_start[2] = (sf[4] & 0x0F) << 8 | sf[5];
    if (num_aus >= 4) au_start[3] = sf[6] << 4 | sf[7] >> 4;
    if (num_aus == 6) { au_start[4] = (sf[7] & 0x0F) << 8 | sf[8]; au_start[5] = sf[9] << 4 | sf[10] >> 4; }
    *num_aus_out = num_aus;                                                   /* the member is set before the plausibility check */
    for (int i = 0; i < num_aus; i++) if (au_start[i] >= au_start[i + 1]) return 0;
    return 1;
}

/* SuperframeFilter::Feed (dabplus_decoder.cpp:50-157) for one logical frame.  st: frame_count + 5 raw frames, caller
 * allocated (orc_sf_state_bytes(len)) and zeroed.  Returns 1 when the call ran a decode attempt and filled ev; on sync the
 * corrected superframe (5 * len bytes) is written to sf_out. */
int orc_sf_state_bytes(int frame_len) { return 16 + 5 * frame_len; }

int orc_superframe_feed(uint8_t* st, const uint8_t* frame, int len, int frame_index, orc_sf_event* ev, uint8_t* sf_out)
{
    int32_t* frame_count = (int32_t*)st; uint8_t* sf_raw = st + 16;
    const int sf_len = 5 * len;
    if (len < 10 || (5 * len) % 120) return 0;                                /* :59-66: frame ignored */
    if (*frame_count == 5) {
        for (int i = 0; i < 4; i++) memcpy(sf_raw + i * len, sf_raw + (i + 1) * len, len);
    } else (*frame_count)++;
    memcpy(sf_raw + (*frame_count - 1) * len, frame, len);
    if (*frame_count < 5) return 0;
    memset(ev, 0, sizeof *ev);
    ev->cif = frame_index; ev->sf_slot = -1;
    memcpy(sf_out, sf_raw, sf_len);
    int corr, unc; orc_rs_superframe(sf_out, sf_len, &corr, &unc);
    ev->corrected = corr; ev->uncorrectable = unc;
    int num_aus = 0, au_start[7] = {0};
    if (!sf_check_sync(sf_out, sf_len, &num_aus, au_start)) return 1;
    ev->sync = 1; ev->format = sf_out[2]; ev->num_aus = num_aus;
    for (int i = 0; i <= num_aus; i++) ev->au_start[i] = au_start[i];
    for (int i = 0; i < num_aus; i++) {                                       /* :122-131 AU CRC (CRC16-CCITT, both inversions) */
        const uint8_t* au = sf_out + au_start[i]; int au_len = au_start[i + 1] - au_start[i];
        /* an AU shorter than its CRC reads out of bounds in the reference (size_t underflow): treated as a CRC failure here */
        if (au_len < 2) continue;
        uint16_t stored = (uint16_t)(au[au_len - 2] << 8 | au[au_len - 1]);
        if (stored == orc_crc16(au, au_len - 2, 1, 1, 0x1021)) ev->au_crc_ok |= 1 << i;
    }
    *frame_count = 0;                                                         /* :156: wait for a complete new superframe */
    return 1;
}

/* ------------------------------------------------------------------ TII (tii-decoder.cpp:189-383)
 * One call = one pass of TIIDecoder::run over a (NULL symbol, PRS) pair as OFDMProcessor hands them over
 * (ofdm-processor.cpp:381-386,462-466).  The reference decoder drops pairs while its thread is busy; the restatement
 * (like the device path) looks at every frame.
 *
 * Reference behaviour restated literally:
 *  - error_per_correction is an unordered_map<float, uint64_t> (tii-decoder.h:97): the running sum is converted to float,
 *    the frame's float error added, and the result truncated back to uint64;
 *  - std::min_element walks that map in ITS iteration order and keeps the first minimum (tii-decoder.cpp:360-366); the
 *    truncated sums tie often, so the order decides.  It is a property of the C++ library, not of welle.io: the caller
 *    passes it in as rank[cycle][err + 4] (cycle 0: a map filled for the first time, 1: refilled after clear()), taken
 *    from the real container by oracle/tii_order.cpp. */
static uint8_t g_tii_pattern[70];            /* tii-decoder.cpp:29-99: the 70 octets of weight 4 in ascending order, b = 0 is the MSB */
static int g_tii_ready;
static void tii_init(void)
{
    if (g_tii_ready) return;
    int n = 0;
    for (int v = 0; v < 256; v++) if (__builtin_popcount(v) == 4) g_tii_pattern[n++] = (uint8_t)v;
    g_tii_ready = 1;
}
static int tii_pat(int p, int b) { return (g_tii_pattern[p] >> (7 - b)) & 1; }

typedef struct { int32_t num; int32_t cycle; int32_t filled; uint64_t acc[ORC_TII_NERR]; } tii_meas;
struct orc_tii_state { tii_meas m[24 * 70]; };
size_t orc_tii_state_bytes(void) { return sizeof(struct orc_tii_state); }
void orc_tii_reset(orc_tii_state* st) { memset(st, 0, sizeof *st); }

/* CombPattern::generateCarriers (tii-decoder.cpp:106-129), sorted ascending: 4 blocks x 4 pairs */
static void tii_carriers(int comb, int pattern, int* carriers /* 32 */)
{
    static const int off[4] = {-769, -385, 0, 384};
    int n = 0;
    for (int g = 0; g < 4; g++)
        for (int b = 0; b < 8; b++)
            if (tii_pat(pattern, b)) { const int k = 1 + 2 * comb + 48 * b; carriers[n++] = k + off[g]; carriers[n++] = k + off[g] + 1; }
}

int orc_tii_frame(orc_tii_state* st, const orc_cf32* null2656, const orc_cf32* prs2048, const int32_t* rank,
                  orc_tii_event* ev, int max_ev, uint8_t* detect192)
{
    tii_init();
    static _Thread_local orc_cf32 n[ORC_TU], p[ORC_TU];
    orc_fft2048(null2656 + (ORC_TNULL - ORC_TU), n, 0);           /* :235-237 skip the cyclic prefix of the NULL */
    orc_fft2048(prs2048, p, 0);                                    /* :244-245 */
    orc_cf32 bm[192]; float pw[192]; uint8_t det[192];
    for (int i = 0; i < 192; i++) { const orc_cf32 z = p[1 + 2 * i]; pw[i] = z.re * z.re + z.im * z.im; bm[i].re = 0; bm[i].im = 0; }   /* std::norm :272-275 */
    static const int k_start[4] = {2048 - 768, 2048 - 384, 1, 385};
    for (int g = 0; g < 4; g++)
        for (int i = 0; i < 192; i++) {
            const int k = k_start[g];
            const orc_cf32 m = cmul(n[k + 2 * i], cconj(n[k + 2 * i + 1]));    /* :279-287 */
            bm[i].re += m.re; bm[i].im += m.im;
        }
    for (int i = 0; i < 192; i++) det[i] = cabs_(bm[i]) > pw[i] * 0.4f;         /* :298-306, carrier k = 2i + 1 */
    if (detect192) memcpy(detect192, det, 192);
    /* :308-323 cp_count[(c, p)] = number of detected carriers 1 + 2c + 48b with pattern bit b set; "likely" when >= 4 */
    int likely[24 * 70], n_likely = 0;
    for (int c = 0; c < 24; c++)
        for (int q = 0; q < 70; q++) {
            int cnt = 0;
            for (int b = 0; b < 8; b++) cnt += tii_pat(q, b) && det[c + 24 * b];
            if (cnt >= 4) likely[n_likely++] = c * 70 + q;
        }
    int n_ev = 0;
    if (n_likely >= 10) return 0;                                               /* :327 */
    for (int l = 0; l < n_likely; l++) {
        /* analyse_phase :336-383 */
        const int comb = likely[l] / 70, pattern = likely[l] % 70;
        int carriers[32]; float phases_prs[32];
        tii_carriers(comb, pattern, carriers);
        for (int i = 0; i < 32; i += 2) {
            const int ix = carriers[i] < 0 ? 2048 + carriers[i] : carriers[i];
            phases_prs[i] = carg_(p[ix]); phases_prs[i + 1] = phases_prs[i];
        }
        tii_meas* meas = &st->m[likely[l]];
        for (int err = -4; err < 500; err++) {
            float abs_err = 0;
            for (int j = 0; j < 32; j++) {
                const int ix = carriers[j] < 0 ? 2048 + carriers[j] : carriers[j];
                const float pi = (float)M_PI;
                const float theta = 2.0f * pi * err * carriers[j] / 2048.0f;
                const orc_cf32 rot = {1.0f * cosf(theta), 1.0f * sinf(theta)};   /* std::polar(1.0f, theta) */
                const float delta = carg_(cmul(n[ix], rot)) - phases_prs[j];
                abs_err += fabsf(delta);
            }
            meas->acc[err + 4] = (uint64_t)((float)meas->acc[err + 4] + abs_err);  /* uint64 += float */
        }
        meas->filled = 1;
        meas->num++;
        if (meas->num >= 5) {
            const int32_t* rk = rank + ORC_TII_NERR * meas->cycle;
            int best = 0;
            for (int e = 1; e < ORC_TII_NERR; e++)
                if (meas->acc[e] < meas->acc[best] || (meas->acc[e] == meas->acc[best] && rk[e] < rk[best])) best = e;
            if (n_ev < max_ev) {
                ev[n_ev].comb = comb; ev[n_ev].pattern = pattern;
                ev[n_ev].error = (float)meas->acc[best];                          /* m.error = best->second */
                ev[n_ev].delay_samples = (int)(float)(best - 4);                  /* m.delay_samples = best->first */
            }
            n_ev++;
            memset(meas->acc, 0, sizeof meas->acc);                              /* clear() */
            meas->num = 0; meas->cycle = 1;
        }
    }
    return n_ev;
}

/* ------------------------------------------------------------------------------------------ receiver */

typedef struct {
    const orc_cf32* iq; int64_t n, pos;
    const orc_cf32* osc;
    int32_t localPhase; float sLevel;
    int32_t coarse; int16_t fine;
    int failed;
    int32_t bufferContent;
} rx_t;

/* The pull loop of ofdm-processor.cpp:150-160,192-202 against the in-memory InputInterface of
 * oracle/ref_harness.cpp (getSamplesToRead() = min(remaining, 2656); is_ok() = remaining >= 2656). */
static int rx_avail(rx_t* r, int n)
{
    if (n > r->bufferContent) {
        int64_t k = r->n - r->pos; if (k > ORC_TNULL) k = ORC_TNULL;
        r->bufferContent = (int32_t)k;
        if (r->bufferContent < n) { r->failed = 1; return 0; }
    }
    return 1;
}

/* ofdm-processor.cpp:145-184 */
static orc_cf32 rx_get_sample(rx_t* r, int32_t phase)
{
    orc_cf32 temp = {0, 0};
    if (!rx_avail(r, 1)) return temp;
    temp = r->iq[r->pos++]; r->bufferContent--;
    r->localPhase -= phase;
    r->localPhase = (r->localPhase + ORC_INPUT_RATE) % ORC_INPUT_RATE;
    temp = cmul(temp, r->osc[r->localPhase]);
    r->sLevel = (float)(0.00001 * l1_norm(temp) + (1 - 0.00001) * r->sLevel);
    return temp;
}

/* ofdm-processor.cpp:186-224 */
static void rx_get_samples(rx_t* r, orc_cf32* v, int n, int32_t phase)
{
    if (!rx_avail(r, n)) return;
    memcpy(v, r->iq + r->pos, sizeof(orc_cf32) * (size_t)n); r->pos += n; r->bufferContent -= n;
    for (int i = 0; i < n; i++) {
        r->localPhase -= phase;
        r->localPhase = (r->localPhase + ORC_INPUT_RATE) % ORC_INPUT_RATE;
        v[i] = cmul(v[i], r->osc[r->localPhase]);
        r->sLevel = (float)(0.00001 * l1_norm(v[i]) + (1 - 0.00001) * r->sLevel);
    }
}

/* ofdm-processor.cpp:235-501 + lock-step hand-over to ofdm-decoder.cpp:93-130, fic-handler, msc-handler.cpp:129-158 */
int orc_receiver_run(orc_run_io* io)
{
    orc_init();
    rx_t R; memset(&R, 0, sizeof R);
    R.iq = io->iq; R.n = io->n_samples; R.osc = orc_nco_table();
    orc_demod_state* dem = (orc_demod_state*)calloc(1, sizeof *dem);
    orc_subch* subs = (orc_subch*)calloc((size_t)(io->n_subch > 0 ? io->n_subch : 1), sizeof(orc_subch));
    for (int i = 0; i < io->n_subch; i++) { orc_subch_init(&subs[i], &io->subch[i].prot, io->subch[i].length_cu); io->msc_len[i] = 0; }
    orc_cf32* ofdmBuffer = (orc_cf32*)malloc(sizeof(orc_cf32) * ORC_L * ORC_TS);
    orc_cf32* syms = (orc_cf32*)malloc(sizeof(orc_cf32) * (ORC_TU + 75 * ORC_TS));
    int8_t* cif = (int8_t*)calloc(1, 55296);
    int8_t soft[3072]; int8_t fic_soft[9216];
    static float envBuffer[32768];
    orc_cf32 nullSymbol[ORC_TNULL];
    int fic_ratio = 0;
    int frame = 0;
    float currentStrength; int syncBufferIndex, counter; int32_t startIndex;
    io->n_fib = io->n_frames = io->n_snr = io->n_sync_true = io->n_sync_false = io->n_cir = 0;

    R.sLevel = 0;
    for (int i = 0; i < ORC_TF / 2; i++) { rx_get_sample(&R, 0); if (R.failed) goto done; }
notSynced:
    syncBufferIndex = 0; currentStrength = 0;
    for (int i = 0; i < 50; i++) {
        orc_cf32 s = rx_get_sample(&R, 0); if (R.failed) goto done;
        envBuffer[syncBufferIndex] = l1_norm(s);
        currentStrength += envBuffer[syncBufferIndex];
        syncBufferIndex++;
    }
    counter = 0;
    io->n_sync_false++;
    while (currentStrength / 50 > 0.50 * R.sLevel) {
        orc_cf32 s = rx_get_sample(&R, R.coarse + R.fine); if (R.failed) goto done;
        envBuffer[syncBufferIndex] = l1_norm(s);
        currentStrength += envBuffer[syncBufferIndex] - envBuffer[(syncBufferIndex - 50) & 32767];
        syncBufferIndex = (syncBufferIndex + 1) & 32767;
        counter++;
        if (counter > ORC_TF) goto notSynced;
    }
    counter = 0;
    while (currentStrength / 50 < 0.75 * R.sLevel) {
        orc_cf32 s = rx_get_sample(&R, R.coarse + R.fine); if (R.failed) goto done;
        envBuffer[syncBufferIndex] = l1_norm(s);
        currentStrength += envBuffer[syncBufferIndex] - envBuffer[(syncBufferIndex - 50) & 32767];
        syncBufferIndex = (syncBufferIndex + 1) & 32767;
        counter++;
        if (counter > ORC_TNULL + 50) goto notSynced;
    }
SyncOnPhase:
    {
        int64_t buf_pos = R.pos;
        rx_get_samples(&R, ofdmBuffer, ORC_TU, R.coarse + R.fine); if (R.failed) goto done;
        float* ir = (io->n_cir < io->cir_cap && io->cir) ? io->cir + 2048 * (size_t)io->n_cir : NULL;   /* onNewImpulseResponse fires on every attempt :344 */
        io->n_cir++;
        startIndex = orc_find_index(ofdmBuffer, io->fft_placement, ir);
        if (frame < io->sidx_cap && io->start_index) { io->start_index[frame] = startIndex; io->frame_pos[frame] = buf_pos; }
        if (startIndex < 0) goto notSynced;
        memmove(ofdmBuffer, &ofdmBuffer[startIndex], sizeof(orc_cf32) * (size_t)(ORC_TU - startIndex));
        int ofdmBufferIndex = ORC_TU - startIndex;
        io->n_sync_true++;
        rx_get_samples(&R, &ofdmBuffer[ofdmBufferIndex], ORC_TU - ofdmBufferIndex, R.coarse + R.fine); if (R.failed) goto done;
        if (!io->disable_coarse && fic_ratio * 10 < 50) {
            int correction = orc_coarse_prs_method(ofdmBuffer, io->freqsync_sel == 1 ? 0 : io->freqsync_sel == 2 ? 1 : 2);
            if (correction != 100) {
                R.coarse += correction * 1000;
                if (abs(R.coarse) > 35000) R.coarse = 0;
            }
        }
        memcpy(syms, ofdmBuffer, sizeof(orc_cf32) * ORC_TU);
        orc_cf32 FreqCorr = {0, 0};
        for (int sym = 1; sym < ORC_L; sym++) {
            orc_cf32* buf = syms + ORC_TU + (size_t)(sym - 1) * ORC_TS;
            rx_get_samples(&R, buf, ORC_TS, R.coarse + R.fine); if (R.failed) goto done;
            for (int i = ORC_TU; i < ORC_TS; i++) {
                orc_cf32 p = cmul(buf[i], cconj(buf[i - ORC_TU]));
                FreqCorr.re += p.re; FreqCorr.im += p.im;
            }
        }
        /* ---- thread B, in lock step: ofdm-decoder.cpp:105-121 ---- */
        {
            float snr_v;
            if (orc_demod_prs(dem, syms, &snr_v)) { if (io->n_snr < io->snr_cap && io->snr) io->snr[io->n_snr] = snr_v; io->n_snr++; }
            orc_cf32* con = (frame < io->con_cap && io->con) ? io->con + 1200 * (size_t)frame : NULL;
            for (int sym = 1; sym < ORC_L; sym++) {
                orc_demod_symbol(dem, syms + ORC_TU + (size_t)(sym - 1) * ORC_TS, soft, con ? con + 16 * (sym - 1) : NULL);
                if (io->soft && frame < io->soft_cap) memcpy(io->soft + ((size_t)frame * 75 + (sym - 1)) * 3072, soft, 3072);
                if (sym < 4) {
                    memcpy(fic_soft + 3072 * (sym - 1), soft, 3072);
                    if (sym == 3) {
                        uint8_t bits[12 * 256], ok[12];
                        orc_fic_decode(fic_soft, bits, ok, &fic_ratio);
                        for (int f = 0; f < 12; f++) {
                            if (io->n_fib < io->fib_cap && io->fib) {
                                uint8_t* o = io->fib + 33 * (size_t)io->n_fib;
                                o[0] = ok[f];
                                for (int i = 0; i < 32; i++) { uint8_t b = 0; for (int j = 0; j < 8; j++) b = (uint8_t)((b << 1) | bits[256 * f + 8 * i + j]); o[1 + i] = b; }
                            }
                            io->n_fib++;
                        }
                    }
                } else {
                    int currentblk = (sym - 4) % 18;
                    memcpy(cif + currentblk * 3072, soft, 3072);
                    if (currentblk == 17)
                        for (int c = 0; c < io->n_subch; c++) {
                            uint8_t outb[3 * 384 + 8];
                            int nb = orc_subch_process(&subs[c], cif + io->subch[c].start_cu * 64, outb);
                            if (nb > 0 && io->msc_len[c] + nb <= io->msc_cap[c]) { memcpy(io->msc[c] + io->msc_len[c], outb, (size_t)nb); io->msc_len[c] += nb; }
                        }
                }
            }
        }
        /* ---- back in thread A: ofdm-processor.cpp:447-489 ---- */
        R.fine = (int16_t)(R.fine + 0.1 * carg_(FreqCorr) / M_PI * (1000 / 2));
        rx_get_samples(&R, nullSymbol, ORC_TNULL, R.coarse + R.fine); if (R.failed) { io->n_frames = ++frame; goto done; }
        if (frame < io->nul_cap && io->nul) memcpy(io->nul + 2656 * (size_t)frame, nullSymbol, sizeof nullSymbol);
        if (io->tii_state) {                                                       /* :464-466 pushSymbols(nullSymbol, prs) */
            orc_tii_event tev[16];
            const int ne = orc_tii_frame((orc_tii_state*)io->tii_state, nullSymbol, syms, io->tii_rank, tev, 16, NULL);
            for (int k = 0; k < ne && k < 16; k++) {
                if (io->n_tii < io->tii_cap) { tev[k].frame = frame; io->tii_ev[io->n_tii] = tev[k]; }
                io->n_tii++;
            }
        }
        if (frame < io->corr_cap && io->corr) { io->corr[2 * frame] = R.fine; io->corr[2 * frame + 1] = R.coarse; }
        if (R.fine > 1000 / 2) { R.coarse += 1000; R.fine -= 1000; }
        else if (R.fine < -1000 / 2) { R.coarse -= 1000; R.fine += 1000; }
        frame++;
        io->n_frames = frame;
        goto SyncOnPhase;
    }
done:
    for (int i = 0; i < io->n_subch; i++) orc_subch_free(&subs[i]);
    free(subs); free(dem); free(ofdmBuffer); free(syms); free(cif);
    return 0;
}
