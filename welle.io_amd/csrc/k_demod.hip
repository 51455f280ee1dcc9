// welle.io_amd/csrc/k_demod.hip -- OFDM demodulation kernel: NCO mix + 2048-pt FFT + differential QPSK
// demap + frequency de-interleave + float->int8 soft bits, fused, one pass over the IQ samples.
//
// Replaces (reference file:line, relative to src/backend):
//   OFDMProcessor::getSamples NCO loop        ofdm-processor.cpp:211-216   (when mix != 0)
//   OfdmDecoder::processPRS                   ofdm-decoder.cpp:144-166
//   OfdmDecoder::decodeDataSymbol             ofdm-decoder.cpp:175-230
//   FrequencyInterleaver::mapIn               freq-interleaver.cpp:88-91   (as the inverse table bin2soft)
//
// Work decomposition: one 128-thread work-group per (ensemble, frame, chunk of consecutive symbols).  A
// chunk first transforms the symbol preceding it (the PRS for chunk 0) to obtain the phase reference,
// keeps it in registers, then walks its symbols.  Every symbol is read from HBM exactly once per chunk
// (+1/chunk_len redundant reads for the reference symbol); the 3 KiB of soft bits are staged in LDS and
// leave as coalesced 8-byte stores.
#include "fft2048.h"
#include "dabphy_kernels.h"
#include <dabphy_wave_ops.h>
#include "osc_exact.h"

namespace dabphy {

__device__ __forceinline__ int32_t mod_rate(int64_t x)
{
    int64_t r = x % INPUT_RATE;
    if (r < 0) r += INPUT_RATE;
    return (int32_t)r;
}

// Position of a symbol's useful part as seen by thread t: ring index of its sample t and that sample's oscillator phase.
struct SymCursor { uint32_t a; int32_t ph; };
struct MixSteps { int32_t s128, sTS; uint32_t s256_bytes; int32_t f1; };   // (128 f, T_s f) mod RATE; 8 * ((256 f) mod RATE); f mod RATE

constexpr uint32_t NCO_BYTES = (uint32_t)INPUT_RATE * 8u;      // the oscillator table, one cf32 per phase step

// The oscillator of one thread: exp(j 2 pi phase / RATE) of its sample t of the current symbol, and the factors that
// advance it by 128 / 256 / 512 / 1024 samples (osc_exact.h).
struct OscChain { dc64 base, d128, d256, d512, d1024; };

__device__ __forceinline__ void osc_chain_steps(OscChain& k, int32_t f_hz)
{
    // the same in every lane (f_hz is per frame): kept in scalar registers, 24 VGPRs less
    const dc64 a = osc_step(128, f_hz), b = osc_step(256, f_hz), b2 = osc_step(512, f_hz), b4 = osc_step(1024, f_hz);
    k.d128.re = uniform_f64(a.re); k.d128.im = uniform_f64(a.im);
    k.d256.re = uniform_f64(b.re); k.d256.im = uniform_f64(b.im);
    k.d512.re = uniform_f64(b2.re); k.d512.im = uniform_f64(b2.im);
    k.d1024.re = uniform_f64(b4.re); k.d1024.im = uniform_f64(b4.im);
}

// Oscillator values of one half of a symbol in round-A order: o[j] = oscillatorTable[phase(t + 128h + 256j)]
// (ofdm-processor.cpp:211-214), computed, not gathered.  A sample whose double sits too close to a float rounding boundary
// (about 4 in 10^5) is read from the table; the walk over the integer phases only happens in a wave that has one.
// `checked` = the symbol reads a table entry next to a float rounding boundary (FrameDesc::osc_hazard, about one symbol in 25):
// only then does the conversion need osc_round's test; every other symbol converts its doubles directly (osc_exact.h).
__device__ __forceinline__ void osc_half(cf32 (&o)[8], const OscChain& k, const cf32* __restrict__ nco, const SymCursor& c,
                                         const MixSteps& st, int h, bool checked)
{
    if (st.f1 == 0) {
        // f = 0 (a carrier offset below the fine corrector's dead zone of 10 Hz leaves it there): the phase never moves, every sample
        // of the frame reads the same table entry -- read it (one uniform load) instead of computing 16 values that are all equal
        const cf32 v = nco[c.ph];
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = v;
        return;
    }
    // e[j] = base * exp(-j 2 pi (128 h + 256 j) f / RATE) as a tree of depth 3 (steps of 256, 512, 1024 samples) instead of a chain of
    // seven dependent double-precision complex multiplications: same operation count, no latency chain (and a shorter error chain)
#ifdef DEMOD_EXP_OSC_F32               // (timing experiment only, profiles/r06_demod_split.txt: the same tree in single precision -- what the
    {                                  //  double-precision arithmetic itself costs; the values are wrong in the last bits)
        auto fm = [](cf32 a, cf32 b) { cf32 r; r.re = a.re * b.re - a.im * b.im; r.im = a.re * b.im + a.im * b.re; return r; };
        auto f32 = [](dc64 a) { cf32 r; r.re = (float)a.re; r.im = (float)a.im; return r; };
        const cf32 b0 = f32(k.base), s128 = f32(k.d128), s256 = f32(k.d256), s512 = f32(k.d512), s1024 = f32(k.d1024);
        o[0] = h ? fm(b0, s128) : b0; o[1] = fm(o[0], s256); o[2] = fm(o[0], s512); o[3] = fm(o[1], s512);
        o[4] = fm(o[0], s1024); o[5] = fm(o[1], s1024); o[6] = fm(o[2], s1024); o[7] = fm(o[3], s1024);
        return;
    }
#endif
    dc64 e[8];
    e[0] = h ? osc_mul(k.base, k.d128) : k.base;
    e[1] = osc_mul(e[0], k.d256);
    e[2] = osc_mul(e[0], k.d512); e[3] = osc_mul(e[1], k.d512);
    e[4] = osc_mul(e[0], k.d1024); e[5] = osc_mul(e[1], k.d1024); e[6] = osc_mul(e[2], k.d1024); e[7] = osc_mul(e[3], k.d1024);
    if (!checked) {
#pragma unroll
        for (int j = 0; j < 8; j++) { o[j].re = (float)e[j].re; o[j].im = (float)e[j].im; }
        return;
    }
    uint32_t hard = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) hard |= osc_round(e[j], o[j]) << j;
    if (!wave_all(hard == 0)) {
        int32_t ph = c.ph; if (h) { ph -= st.s128; if (ph < 0) ph += INPUT_RATE; }
        uint32_t pb = (uint32_t)ph * 8u;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if ((hard >> j) & 1u) o[j] = *reinterpret_cast<const cf32*>(reinterpret_cast<const char*>(nco) + pb);
            pb -= st.s256_bytes; { const uint32_t w_ = pb + NCO_BYTES; pb = w_ < pb ? w_ : pb; }
        }
    }
}

// One half of a symbol straight from the sample ring in HBM, unmixed, and the mixing as a step of its own
__device__ __forceinline__ void load_raw_half(cf32 (&x)[8], const cf32* __restrict__ iq, uint32_t ring, const SymCursor& c, int h)
{
    uint32_t a = c.a + 128u * h; if (a >= ring) a -= ring;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        x[j] = iq[a];
        a += 256u; if (a >= ring) a -= ring;
    }
}
__device__ __forceinline__ void mix_half(cf32 (&x)[8], const SymCursor& c, const OscChain& k, const cf32* __restrict__ nco,
                                         const MixSteps& st, int h, int mix, bool checked)
{
    if (mix) {
        cf32 o[8]; osc_half(o, k, nco, c, st, h, checked);
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = pk_cmul(x[j], o[j]);
    }
}

// One half of a symbol straight from the sample ring in HBM: x[j] = sample (t + 128h + 256j), times its oscillator value
__device__ __forceinline__ void load_half(cf32 (&x)[8], const cf32* __restrict__ iq, uint32_t ring, const SymCursor& c,
                                          const OscChain& k, const cf32* __restrict__ nco, const MixSteps& st, int h, int mix, bool checked)
{
    uint32_t a = c.a + 128u * h; if (a >= ring) a -= ring;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        x[j] = iq[a];
        a += 256u; if (a >= ring) a -= ring;
    }
    if (mix) {
        cf32 o[8]; osc_half(o, k, nco, c, st, h, checked);
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = pk_cmul(x[j], o[j]);
    }
}

// Bins held by thread t after the transform: t + 128 j.  Carriers are k = +1..+768 (bins 1..768) and k = -768..-1 (bins
// 1280..2047): j = 0..5 and 10..15 for every thread but thread 0, whose j = 0 is the unused DC bin and which owns bin 768
// (j = 6) instead.  So every thread demaps exactly 12 carriers: slot q < 6 is j = q (thread 0, q = 0: j = 6), slot q >= 6
// is j = q + 4.
constexpr int N_SLOTS = 12;
#ifndef DIV127_VARIANT
#define DIV127_VARIANT 0
#endif
__device__ __forceinline__ void carrier_slots(cf32 (&c)[N_SLOTS], const cf32 (&v)[16], int t)
{
    // bitwise blend (the compiler turns a cf32 ?: into an indexed load from a stack copy)
    const uint32_t m = t == 0 ? 0xffffffffu : 0u;
    c[0].re = __uint_as_float((__float_as_uint(v[6].re) & m) | (__float_as_uint(v[0].re) & ~m));
    c[0].im = __uint_as_float((__float_as_uint(v[6].im) & m) | (__float_as_uint(v[0].im) & ~m));
#pragma unroll
    for (int q = 1; q < 6; q++) c[q] = v[q];
#pragma unroll
    for (int q = 6; q < N_SLOTS; q++) c[q] = v[q + 4];
}

template <bool CON>
__global__ void __launch_bounds__(FFT_THREADS, DEMOD_WAVES) k_demod(DemodArgs A)
{
    __shared__ __attribute__((aligned(16))) cf32 tile[FFT_INPLACE_TILE];
    __shared__ __attribute__((aligned(16))) cf32 twB[FFT_TWB_ENTRIES];
    // (the next symbol's raw samples land in the exchange tile itself: 24 KiB per work-group, 168 registers: three waves per SIMD)
    __shared__ __attribute__((aligned(16))) int8_t softbuf[SOFT_PER_SYM];
    __shared__ __attribute__((aligned(16))) dc64 s_symstep[L_SYM];   // exp(-j 2 pi k T_s f / RATE): symbol k of the chunk against its first
    const int t = threadIdx.x;
    const int chunk = blockIdx.x, f = A.frame_first + (int)blockIdx.y, b = blockIdx.z;
    const FrameDesc d = A.desc[(size_t)b * A.n_frames + f];
    if (d.valid != 1) return;
    const cf32* __restrict__ iq = A.iq + (size_t)b * A.iq_stride;
    const cf32* __restrict__ nco = A.tab.nco;
    const uint32_t ring = (uint32_t)A.ring;
    const int s_begin = 1 + chunk * A.chunk_len;                 // first data symbol of this chunk
    int s_end = s_begin + A.chunk_len; if (s_end > L_SYM) s_end = L_SYM;
    if (s_begin >= L_SYM) return;

    FftTwiddles w; fft_load_twiddles(w, A.tab.tw, twB, t);
    // soft-bit index (freq-interleaver.cpp:88-91 inverted) of each slot: a 3 KiB LDS table read back slot by slot (twelve loop-invariant
    // registers would not fit beside three waves per SIMD)
    __shared__ uint16_t sidx_lds[N_SLOTS * FFT_THREADS];
#pragma unroll
    for (int q = 0; q < N_SLOTS; q++) {
        const int j = q >= 6 ? q + 4 : (q == 0 && t == 0) ? 6 : q;
        sidx_lds[q * FFT_THREADS + t] = (uint16_t)A.tab.bin2soft[t + 128 * j];       // (read only by the thread that wrote it)
    }
    auto sidx_of = [&](int q) { return (int)sidx_lds[q * FFT_THREADS + t]; };

    // offset (from d.pos) of the useful part of symbol s: PRS at start_index; s >= 1 at J0 + (s-1) T_s + T_g.
    // Phase of the sample at offset j: j < J0: (L0 - (j+1) f_prs) mod RATE, else (L1 - (j-J0+1) f_sym) mod RATE.
    const int32_t J0 = d.start_index + T_U;
    auto cursor_at = [&](int32_t off) {
        SymCursor c;
        c.a = (uint32_t)((d.pos + off + t) % A.ring);
        if (off < J0) c.ph = mod_rate((int64_t)d.L0 - (int64_t)(off + t + 1) * d.f_prs);
        else c.ph = mod_rate((int64_t)d.L1 - (int64_t)(off - J0 + t + 1) * d.f_sym);
        return c;
    };
    auto steps_for = [&](int32_t fhz) { MixSteps m; m.s128 = mod_rate(128LL * fhz); m.s256_bytes = 8u * (uint32_t)mod_rate(256LL * fhz); m.sTS = mod_rate((int64_t)T_S * fhz); m.f1 = mod_rate((int64_t)fhz); return m; };

    // (selected, not indexed: a dynamically indexed member would move the whole descriptor to scratch memory)
    const uint32_t hz0 = d.osc_hazard[0], hz1 = d.osc_hazard[1], hz2 = d.osc_hazard[2];
#ifdef DEMOD_FORCE_CHECKED                      // (timing experiments only: one conversion compiled in)
    auto hazard_bit = [&](int) { return DEMOD_FORCE_CHECKED != 0; };
#else
    auto hazard_bit = [&](int sym) { return (((sym < 32 ? hz0 : sym < 64 ? hz1 : hz2) >> (sym & 31)) & 1u) != 0; };
#endif
    cf32 prev[N_SLOTS], v[16];
    OscChain osc{};
    {   // reference symbol of the chunk (the PRS for chunk 0)
        const int sref = s_begin - 1;
        const int32_t off = sref == 0 ? d.start_index : J0 + (sref - 1) * T_S + T_G;
        const SymCursor c = cursor_at(off);
        const MixSteps ms = steps_for(sref == 0 ? d.f_prs : d.f_sym);
        if (A.mix) { osc_chain_steps(osc, sref == 0 ? d.f_prs : d.f_sym); osc.base = osc_exp(c.ph); }
        __syncthreads();
        const bool checked = hazard_bit(sref);
#pragma unroll
        for (int h = 0; h < 2; h++) { cf32 x[8]; load_half(x, iq, ring, c, osc, nco, ms, h, A.mix, checked); fft_round_a_inplace<false>(x, h, tile, w, t); }
        fft_rounds_bc_split<false>(v, tile, w, t);
        if (sref == 0 && A.prs_mag) {
            // |bin| of the PRS for the SNR estimate (ofdm-decoder.cpp:240-266): stored in bin order, summed by k_snr_frames
            float* pm = A.prs_mag + ((size_t)b * A.n_frames + f) * T_U;
#pragma unroll
            for (int j = 0; j < 16; j++) pm[t + 128 * j] = hypotf_exact(v[j].re, v[j].im);
        }
        carrier_slots(prev, v, t);
    }
    const size_t slot = (size_t)((d.frame_no) % A.soft_ring);
    int8_t* soft_frame = A.soft + (size_t)b * (A.soft_ens_stride ? A.soft_ens_stride : (size_t)A.soft_ring * SOFT_PER_FRAME) + slot * SOFT_PER_FRAME;
    cf32* con_frame = CON ? A.con + ((size_t)b * A.n_frames + f) * 1200 : nullptr;

    SymCursor cur = cursor_at(J0 + (s_begin - 1) * T_S + T_G);
    const MixSteps ms = steps_for(d.f_sym);
    // Sample prefetch: while symbol s is transformed, the 16 KiB of symbol s+1 travel HBM -> LDS by LDS-DMA (no VGPRs, no
    // wave waiting on HBM).  Wave w copies samples 1024w .. 1024w+1023 as 8 x 1 KiB (lane l of KiB c: samples 128c + 2l,
    // 128c + 2l + 1), four KiB per address computation.  A symbol whose useful part wraps around the end of the sample ring
    // (once per ring revolution) takes the direct path instead.
    const int lane = t & 63, wv = t >> 6;
    uint32_t sym0 = (uint32_t)((d.pos + J0 + (int64_t)(s_begin - 1) * T_S + T_G) % A.ring);   // ring index of sample 0 (uniform)
    auto dma_ok = [&](uint32_t a0) { return a0 + (uint32_t)T_U <= ring; };
    // ... into the tile itself, rows of 256 samples 32 bytes apart (fft2048.h, "k_demod's variant"): wave w fills rows 4w .. 4w+3,
    // two requests per row; the row's address is scalar arithmetic, the lanes keep one 32-bit offset
    const int wvu = uniform_i32(wv);
    auto dma_issue = [&](uint32_t a0) {
        const char* g = reinterpret_cast<const char*>(iq + a0 + 1024 * wvu) + 16 * lane;
        cf32* l = tile + 4 * FFT_RAW_PITCH * wvu;
#pragma unroll
        for (int r = 0; r < 4; r++) { lds_dma16<0>(g + 2048 * r, l + FFT_RAW_PITCH * r); lds_dma16<1024>(g + 2048 * r, l + FFT_RAW_PITCH * r); }
    };
    bool staged = dma_ok(sym0);
    // Every symbol's base is ONE product of two osc_exp values (the thread's base at the chunk's first symbol and the wave-uniform step
    // to symbol k, from LDS), not a chain of k products: the unchecked conversion's error budget (osc_exact.h) counts on it.
    dc64 base0{};
    if (A.mix) {
        if (s_begin == 1 && d.f_prs != d.f_sym) osc_chain_steps(osc, d.f_sym);   // (the reference symbol's factors serve unless it was the PRS pulled at another frequency)
        base0 = osc_exp(cur.ph); osc.base = base0;
        if (t < s_end - s_begin) s_symstep[t] = osc_step((int64_t)t * T_S, d.f_sym);      // (published by the barrier that opens the loop)
    }
    if (staged) dma_issue(sym0);       // overlaps nothing yet (the reference symbol is done), but primes the pipeline
    // The 3 KiB of soft bits of symbol s leave at the START of iteration s+1: the barrier that opens an iteration is then also
    // the one that completes the soft-bit staging (4 barriers per symbol, not 5), and the wait for the LDS-DMA -- the wave's
    // only memory counter also counts stores -- finds stores that are a whole symbol old instead of ones just issued.
    auto store_soft = [&](int sym) {
        const uint2* src = reinterpret_cast<const uint2*>(softbuf);
        uint2* dst = reinterpret_cast<uint2*>(soft_frame + (size_t)(sym - 1) * SOFT_PER_SYM);
#pragma unroll
        for (int i = 0; i < 3; i++) dst[t + 128 * i] = src[t + 128 * i];
    };
    uint32_t n_checked = 0;
    for (int s = s_begin; s < s_end; s++) {
        const bool checked = hazard_bit(s);
        n_checked += checked ? 1u : 0u;
        if (staged) lds_dma_wait();
        __syncthreads();                                         // tile free again, stage complete, soft bits of s-1 complete
        if (s > s_begin) store_soft(s - 1);
        uint32_t next0 = sym0 + T_S; if (next0 >= ring) next0 -= ring;
        const bool next_staged = (s + 1 < s_end) && dma_ok(next0);
        // The samples lie in the tile (fft2048.h, "k_demod's variant"): a half is read, mixed, transformed and written back over
        // itself.  The next symbol's transfer starts behind this wave's round-C reads and is in flight during round C's arithmetic,
        // the demapper and the soft-bit stores -- and behind the other work-groups of the CU (three waves per SIMD).
#pragma unroll
        for (int h = 0; h < 2; h++) {
            cf32 x[8];
            if (staged) fft_raw_half(x, tile, h, t); else load_raw_half(x, iq, ring, cur, h);
            mix_half(x, cur, osc, nco, ms, h, A.mix, checked);
            fft_round_a_inplace<false>(x, h, tile, w, t);
        }
        fft_rounds_bc_split<false>(v, tile, w, t, [&]() { if (next_staged) dma_issue(next0); });
        sym0 = next0; staged = next_staged;
        cur.a += T_S; if (cur.a >= ring) cur.a -= ring;
        cur.ph -= ms.sTS; if (cur.ph < 0) cur.ph += INPUT_RATE;
        // the base is re-anchored every OSC_REANCHOR symbols: the unchecked conversion's error budget (osc_exact.h) counts on it
        if (A.mix && s + 1 < s_end) osc.base = osc_mul(base0, s_symstep[s + 1 - s_begin]);
#ifdef DEMOD_EXP_NODEMAP            // (timing experiment only, profiles/r06_demod_split.txt: no differential product, no 127 / |r| scaling, no scatter -- the
        {                              //  soft-bit stores stay: 24 bytes per thread and symbol straight from the transform's registers)
            uint32_t* sb = reinterpret_cast<uint32_t*>(softbuf);
#pragma unroll
            for (int q = 0; q < 6; q++) {           // (every output of the transform stays alive: 32 floats folded into 6 words)
                uint32_t acc = __float_as_uint(v[q].re) ^ __float_as_uint(v[q].im) ^ __float_as_uint(v[q + 6].re) ^ __float_as_uint(v[q + 6].im);
                if (q < 4) acc ^= __float_as_uint(v[q + 12].re) ^ __float_as_uint(v[q + 12].im);
                sb[t + 128 * q] = acc;
            }
            continue;
        }
#endif
        cf32 r1[N_SLOTS]; float l1[N_SLOTS];
        {
            cf32 cs[N_SLOTS]; carrier_slots(cs, v, t);
#pragma unroll
            for (int q = 0; q < N_SLOTS; q++) {
                r1[q] = pk_cmulc(cs[q], prev[q]);                             // ofdm-decoder.cpp:206
                l1[q] = l1norm(r1[q]);
                prev[q] = cs[q];                                              // :207
            }
        }
        float lo = l1[0], hi = l1[0];
#pragma unroll
        for (int q = 1; q < N_SLOTS; q++) { lo = fminf(lo, l1[q]); hi = fmaxf(hi, l1[q]); }
        if (wave_all(lo >= DIV127_LO && hi <= DIV127_HI)) {
            // every |r1| of the wave is an ordinary number: reciprocal-based 127/x (proven equal to the IEEE quotient on
            // this range by dabphy_selftest_div127), products bounded by 127, so the conversion needs no special cases.
            // (A NaN l1 hides from fminf/fmaxf; it makes vr, vi NaN, which v_cvt_i32_f32 turns into 0 like the reference.)
#pragma unroll
            for (int q = 0; q < N_SLOTS; q++) {
                const float ab1 = div127_fast<DIV127_VARIANT>(l1[q]);         // :208
                const float vr = (-r1[q].re) * ab1, vi = (-r1[q].im) * ab1;  // :211-212
                const int si = sidx_of(q);
                softbuf[si] = (int8_t)cvt_i32_trunc(vr);
                softbuf[K_CARR + si] = (int8_t)cvt_i32_trunc(vi);
            }
        } else {
#pragma unroll
            for (int q = 0; q < N_SLOTS; q++) {
                const float ab1 = 127.0f / l1[q];
                const float vr = (-r1[q].re) * ab1, vi = (-r1[q].im) * ab1;
                // float -> int8 as the reference's x86-64 build does it: cvttss2si yields 0x80000000 for NaN (r1 == 0 ->
                // inf * 0) and for anything out of int range, whose low byte is 0
                const int si = sidx_of(q);
                softbuf[si] = (fabsf(vr) < 2147483648.0f) ? (int8_t)(int)vr : (int8_t)0;
                softbuf[K_CARR + si] = (fabsf(vi) < 2147483648.0f) ? (int8_t)(int)vi : (int8_t)0;
            }
        }
        if (CON) {
#pragma unroll
            for (int q = 0; q < N_SLOTS; q++)
                if (sidx_of(q) % 96 == 0) con_frame[(s - 1) * 16 + sidx_of(q) / 96] = r1[q];   // :214-216
        }
    }
    __syncthreads();
    store_soft(s_end - 1);
    if (A.osc_stats && A.mix && t == 0) {
        atomicAdd(&A.osc_stats[0], (unsigned long long)((uint32_t)(s_end - s_begin) - n_checked));
        atomicAdd(&A.osc_stats[1], (unsigned long long)n_checked);
    }
}

// SNR estimate of OfdmDecoder::get_snr(method 1) (ofdm-decoder.cpp:240-266): 266 noise and 768 signal magnitudes summed in the
// reference's index order (one chain of float additions each, so that the int16 truncation of the dB difference sees the same value).
// One ROW of 16 lanes per (ensemble, frame), four frames per wave: the row loads its 1034 operands up front (coalesced 64-byte pieces),
// then lane 0 of the row folds them 16 at a time through row_shl DPP reads, one instruction per addition (chain16) -- round 3 gave
// every frame one thread with 1034 dependent strided loads: 4.0 ms at the head of the auxiliary stream for 8192 frames.
// Operand order: noise = bins (T_u/2 + k) % T_u for k = 70 .. low - 21, then k = high + 20 .. high + 119; signal = k = T_u/2 - K/4 ..
// T_u/2 + K/4 - 1 (which wraps from bin 2047 to bin 0).  Blocks are padded with +0.0f: x + 0 = x exactly for the non-negative sums.
__global__ void __launch_bounds__(64) k_snr_frames(SnrArgs A)
{
    const int lane = threadIdx.x, row = lane >> 4, l16 = lane & 15;
    const int n = A.n_ens * A.n_frames;
    int i = blockIdx.x * 4 + row;
    const bool live = i < n;
    if (!live) i = n - 1;                                         // (rows beyond the batch redo the last frame: every lane takes part in the DPP chains)
    const float* __restrict__ v = A.prs_mag + (size_t)i * T_U;
    constexpr int low = T_U / 2 - K_CARR / 2, high = low + K_CARR;
    constexpr int N1 = low - 20 - 70, N2 = 100, NS = K_CARR / 2;      // 166 + 100 noise operands, 768 signal operands
    constexpr int B1 = (N1 + 15) / 16, B2 = (N2 + 15) / 16, BS = NS / 16;
    static_assert(NS % 16 == 0, "signal operands come in whole blocks");
    float xn[B1 + B2], xs[BS];
#pragma unroll
    for (int k = 0; k < B1; k++) { const int j = 16 * k + l16; xn[k] = j < N1 ? v[(T_U / 2 + 70 + j) % T_U] : 0.0f; }
#pragma unroll
    for (int k = 0; k < B2; k++) { const int j = 16 * k + l16; xn[B1 + k] = j < N2 ? v[(T_U / 2 + high + 20 + j) % T_U] : 0.0f; }
#pragma unroll
    for (int k = 0; k < BS; k++) xs[k] = v[(T_U / 2 + T_U / 2 - K_CARR / 4 + 16 * k + l16) % T_U];
    float noise = 0.0f, signal = 0.0f;
#pragma unroll
    for (int k = 0; k < B1 + B2; k++) noise = chain16(noise, xn[k], 16);
#pragma unroll
    for (int k = 0; k < BS; k++) signal = chain16(signal, xs[k], 16);
    if (l16 == 0 && live) {
        noise /= (low - 90 + 100);
        const float qs = ((signal / (K_CARR / 2)) + 1.0f) / 256.0f, qn = (noise + 1.0f) / 256.0f;   // MathHelper.h:43-46
        const float dB_signal = (float)(20 * log10((double)qs));
        const float dB_noise = (float)(20 * log10((double)qn));
        A.snr_out[i] = (float)(int16_t)(dB_signal - dB_noise);     // get_snr returns int16_t
    }
}

// 0.7/0.3 IIR and the every-11th-frame report (ofdm-decoder.cpp:154-158): one thread per ensemble, frames in order
__global__ void k_snr(SnrArgs A)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= A.n_ens) return;
    DecState& st = A.state[b];
    float snr = st.snr; int cnt = st.snr_count;
    for (int f = 0; f < A.n_frames; f++) {
        float* out = A.snr_out + (size_t)b * A.n_frames + f;
        const float snr_new = *out;
        *out = __int_as_float(0x7fc00000);                     // NaN = "no report for this frame"
        if (A.desc[(size_t)b * A.n_frames + f].valid != 1) continue;
        snr = (float)(0.7 * snr + 0.3 * (double)(int16_t)snr_new);
        if (++cnt > 10) { *out = snr; cnt = 0; }
    }
    st.snr = snr; st.snr_count = cnt;
}

// Exhaustive check of div127_fast against the IEEE quotient: one thread per float bit pattern (2^32 of them)
__global__ void k_selftest_div127(unsigned long long* out)
{
    unsigned long long bad0 = 0, bad1 = 0, tried = 0;
    for (uint32_t i = 0; i < 256; i++) {
        const uint32_t bits = (blockIdx.x * 256u + i) * 256u + threadIdx.x;
        const float x = __uint_as_float(bits);
        if (!(x >= DIV127_LO && x <= DIV127_HI)) continue;
        const float want = 127.0f / x;
        bad0 += __float_as_uint(div127_fast<0>(x)) != __float_as_uint(want);
        bad1 += __float_as_uint(div127_fast<1>(x)) != __float_as_uint(want);
        tried++;
    }
    if (bad0) atomicAdd(&out[0], bad0);
    if (bad1) atomicAdd(&out[1], bad1);
    if (tried) atomicAdd(&out[2], tried);
}
// pk_cmul_unit against the two multiplications and the addition it stands for (pk_cmul), on 2^32 pairs of float bit patterns per
// twiddle sign: re = every exponent x sign x a sample of mantissas (zeros, denormals, infinities, NaNs included), im = a hash of it and,
// for one thread in eight, one of the special values.  A mismatch = different bits, unless both results are NaN (payloads are nobody's
// business: every later stage only asks whether a value is NaN).
__device__ __forceinline__ bool same_float(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }
__global__ void k_selftest_unit_twiddle(unsigned long long* out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;                        // 2^24 threads
    const uint32_t special[8] = {0x00000000u, 0x80000000u, 0x7f800000u, 0xff800000u, 0x7fc00000u, 0x00000001u, 0x807fffffu, 0x7f7fffffu};
    unsigned long long bad = 0, tried = 0;
    for (uint32_t k = 0; k < 256; k++) {
        const uint32_t n = i * 256u + k;
        const uint32_t rb = (n << 23) | (n >> 9);                              // all 512 sign/exponent combinations x 2^23 mantissas over the run
        uint32_t ib = rb * 2654435761u + 0x9e3779b9u; ib ^= ib >> 15; ib *= 2246822519u; ib ^= ib >> 13;
        if ((n & 7u) == 3u) ib = special[(n >> 3) & 7u];
        cf32 x; x.re = __uint_as_float(rb); x.im = __uint_as_float(ib);
        for (int sgn = 0; sgn < 2; sgn++) {
            cf32 w; w.re = __uint_as_float(opaque_vgpr(0x3f800000u)); w.im = __uint_as_float(opaque_vgpr(sgn ? 0x80000000u : 0u));   // (values the compiler cannot fold)
            const cf32 a = pk_cmul(x, w), b = pk_cmul_unit(x, w);
            bad += !(same_float(a.re, b.re) && same_float(a.im, b.im));
            cf32 y; y.re = x.im; y.im = x.re;                                    // (and with the parts exchanged)
            const cf32 c = pk_cmul(y, w), d = pk_cmul_unit(y, w);
            bad += !(same_float(c.re, d.re) && same_float(c.im, d.im));
            tried += 2;
        }
    }
    if (bad) atomicAdd(&out[0], bad);
    atomicAdd(&out[1], tried);
}
void launch_selftest_unit_twiddle(unsigned long long* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_selftest_unit_twiddle, dim3(1u << 16), dim3(256), 0, s, out);
}

void launch_selftest_div127(unsigned long long* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_selftest_div127, dim3(1u << 16), dim3(256), 0, s, out);
}

void launch_demod(const DemodArgs& a, int n_ens, hipStream_t s)
{
    int chunks = (75 + a.chunk_len - 1) / a.chunk_len;
    if (a.chunk_count > 0 && a.chunk_count < chunks) chunks = a.chunk_count;
    const int frames = a.frame_count > 0 ? a.frame_count : a.n_frames - a.frame_first;
    if (a.con) hipLaunchKernelGGL(k_demod<true>, dim3(chunks, frames, n_ens), dim3(FFT_THREADS), 0, s, a);
    else hipLaunchKernelGGL(k_demod<false>, dim3(chunks, frames, n_ens), dim3(FFT_THREADS), 0, s, a);
}

void launch_snr(const SnrArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(k_snr_frames, dim3((a.n_ens * a.n_frames + 3) / 4), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_snr, dim3((a.n_ens + 63) / 64), dim3(64), 0, s, a);
}

} // namespace dabphy
