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

namespace dabphy {

__device__ __forceinline__ int32_t mod_rate(int64_t x)
{
    int64_t r = x % INPUT_RATE;
    if (r < 0) r += INPUT_RATE;
    return (int32_t)r;
}

// Position of a symbol's useful part as seen by thread t: ring index of its sample t and that sample's oscillator phase.
struct SymCursor { uint32_t a; int32_t ph; };
struct MixSteps { int32_t s128, s256, sTS; };          // (128 f, 256 f, T_s f) mod RATE

// One half of a symbol in round-A order: x[j] = sample (t + 128h + 256j) * osc[phase]   (ofdm-processor.cpp:211-214)
__device__ __forceinline__ void load_half(cf32 (&x)[8], const cf32* __restrict__ iq, uint32_t ring, const cf32* __restrict__ nco,
                                          const SymCursor& c, const MixSteps& st, int h, int mix)
{
    uint32_t a = c.a + 128u * h; if (a >= ring) a -= ring;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        x[j] = iq[a];
        a += 256u; if (a >= ring) a -= ring;
    }
    if (!mix) return;
    int32_t ph = c.ph; if (h) { ph -= st.s128; if (ph < 0) ph += INPUT_RATE; }
    cf32 o[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        o[j] = nco[(uint32_t)ph];
        ph -= st.s256; if (ph < 0) ph += INPUT_RATE;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = cmul(x[j], o[j]);
}

// bins held by thread t after the transform: t + 128 j.  Carriers live in j = 0..5 (k = +1..+768; bin 0 itself is
// unused), j = 6 (bin 768, thread 0 only) and j = 10..15 (k = -768..-1).  PJ maps those 13 j to a compact index.
__device__ __forceinline__ constexpr int pj_of(int j) { return j < 7 ? j : j - 3; }

__global__ void __launch_bounds__(FFT_THREADS, DEMOD_WAVES) k_demod(DemodArgs A)
{
    __shared__ __attribute__((aligned(16))) cf32 tile[T_U];
    __shared__ __attribute__((aligned(16))) cf32 twB[FFT_TWB_ENTRIES];
    __shared__ __attribute__((aligned(16))) int8_t softbuf[SOFT_PER_SYM];
    const int t = threadIdx.x;
    const int chunk = blockIdx.x, f = blockIdx.y, b = blockIdx.z;
    const FrameDesc d = A.desc[(size_t)b * A.n_frames + f];
    if (d.valid != 1) return;
    const cf32* __restrict__ iq = A.iq + (size_t)b * A.iq_stride;
    const cf32* __restrict__ nco = A.tab.nco;
    const uint32_t ring = (uint32_t)A.ring;
    const int s_begin = 1 + chunk * A.chunk_len;                 // first data symbol of this chunk
    int s_end = s_begin + A.chunk_len; if (s_end > L_SYM) s_end = L_SYM;
    if (s_begin >= L_SYM) return;

    FftTwiddles w; fft_load_twiddles(w, A.tab.tw, twB, t);
    uint32_t sidx2[7];                                           // packed int16 pairs: soft-bit index of the 13 carrier bins
#pragma unroll
    for (int q = 0; q < 7; q++) {
        const int j0 = 2 * q, j1 = 2 * q + 1;                    // compact indices 2q, 2q+1 -> j
        const int ja = j0 < 7 ? j0 : j0 + 3, jb = j1 < 7 ? j1 : j1 + 3;
        const uint32_t lo = (uint16_t)A.tab.bin2soft[t + 128 * ja];
        const uint32_t hi = (j1 < 13) ? (uint16_t)A.tab.bin2soft[t + 128 * jb] : 0xffffu;
        sidx2[q] = lo | (hi << 16);
    }

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
    auto steps_for = [&](int32_t fhz) { MixSteps m; m.s128 = mod_rate(128LL * fhz); m.s256 = mod_rate(256LL * fhz); m.sTS = mod_rate((int64_t)T_S * fhz); return m; };

    cf32 prev[13], v[16];
    {   // reference symbol of the chunk (the PRS for chunk 0)
        const int sref = s_begin - 1;
        const int32_t off = sref == 0 ? d.start_index : J0 + (sref - 1) * T_S + T_G;
        const SymCursor c = cursor_at(off);
        const MixSteps ms = steps_for(sref == 0 ? d.f_prs : d.f_sym);
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; h++) { cf32 x[8]; load_half(x, iq, ring, nco, c, ms, h, A.mix); fft_round_a<false>(x, h, tile, w, t); }
        fft_rounds_bc<false>(v, tile, w, t);
        if (sref == 0 && A.prs_mag) {
            // |bin| of the PRS for the SNR estimate (ofdm-decoder.cpp:240-266): stored in bin order, summed by k_snr_frames
            float* pm = A.prs_mag + ((size_t)b * A.n_frames + f) * T_U;
#pragma unroll
            for (int j = 0; j < 16; j++) pm[t + 128 * j] = hypotf_exact(v[j].re, v[j].im);
        }
#pragma unroll
        for (int j = 0; j < 16; j++) if (j < 7 || j > 9) prev[pj_of(j)] = v[j];
    }
    const size_t slot = (size_t)((d.frame_no) % A.soft_ring);
    int8_t* soft_frame = A.soft + ((size_t)b * A.soft_ring + slot) * SOFT_PER_FRAME;
    cf32* con_frame = A.con ? A.con + ((size_t)b * A.n_frames + f) * 1200 : nullptr;

    SymCursor cur = cursor_at(J0 + (s_begin - 1) * T_S + T_G);
    const MixSteps ms = steps_for(d.f_sym);
    for (int s = s_begin; s < s_end; s++) {
        __syncthreads();                                         // tile + softbuf free again
#pragma unroll
        for (int h = 0; h < 2; h++) { cf32 x[8]; load_half(x, iq, ring, nco, cur, ms, h, A.mix); fft_round_a<false>(x, h, tile, w, t); }
        fft_rounds_bc<false>(v, tile, w, t);
        cur.a += T_S; if (cur.a >= ring) cur.a -= ring;
        cur.ph -= ms.sTS; if (cur.ph < 0) cur.ph += INPUT_RATE;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (j >= 7 && j <= 9) continue;
            const int p = pj_of(j);
            const int idx = (int)(int16_t)((sidx2[p >> 1] >> (16 * (p & 1))) & 0xffffu);
            if (idx >= 0) {
                const cf32 r1 = cmul(v[j], cconj(prev[p]));                   // ofdm-decoder.cpp:206
                const float ab1 = 127.0f / l1norm(r1);                        // :208
                const float vr = (-r1.re) * ab1, vi = (-r1.im) * ab1;        // :211-212
                // float -> int8: C truncation; NaN (r1 == 0 -> inf * 0) becomes 0 as with cvttss2si on the reference's x86-64 build
                softbuf[idx] = (vr != vr) ? (int8_t)0 : (int8_t)(int)vr;
                softbuf[K_CARR + idx] = (vi != vi) ? (int8_t)0 : (int8_t)(int)vi;
                if (con_frame && (idx % 96) == 0) con_frame[(s - 1) * 16 + idx / 96] = r1;   // :214-216
            }
            prev[p] = v[j];                                                   // :207
        }
        __syncthreads();
        {
            const uint2* src = reinterpret_cast<const uint2*>(softbuf);
            uint2* dst = reinterpret_cast<uint2*>(soft_frame + (size_t)(s - 1) * SOFT_PER_SYM);
#pragma unroll
            for (int i = 0; i < 3; i++) dst[t + 128 * i] = src[t + 128 * i];
        }
    }
}

// SNR estimate of OfdmDecoder::get_snr(method 1) (ofdm-decoder.cpp:240-266): one thread per (ensemble, frame) runs the
// float sums in the reference's order, so the int16 truncation of the dB difference sees the same value.
__global__ void k_snr_frames(SnrArgs A)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n_ens * A.n_frames) return;
    const float* v = A.prs_mag + (size_t)i * T_U;
    float noise = 0, signal = 0;
    const int low = T_U / 2 - K_CARR / 2, high = low + K_CARR;
    for (int k = 70; k < low - 20; k++) noise += v[(T_U / 2 + k) % T_U];
    for (int k = high + 20; k < high + 120; k++) noise += v[(T_U / 2 + k) % T_U];
    noise /= (low - 90 + 100);
    for (int k = T_U / 2 - K_CARR / 4; k < T_U / 2 + K_CARR / 4; k++) signal += v[(T_U / 2 + k) % T_U];
    const float qs = ((signal / (K_CARR / 2)) + 1.0f) / 256.0f, qn = (noise + 1.0f) / 256.0f;   // MathHelper.h:43-46
    const float dB_signal = (float)(20 * log10((double)qs));
    const float dB_noise = (float)(20 * log10((double)qn));
    A.snr_out[i] = (float)(int16_t)(dB_signal - dB_noise);     // get_snr returns int16_t
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

void launch_demod(const DemodArgs& a, int n_ens, hipStream_t s)
{
    const int chunks = (75 + a.chunk_len - 1) / a.chunk_len;
    hipLaunchKernelGGL(k_demod, dim3(chunks, a.n_frames, n_ens), dim3(FFT_THREADS), 0, s, a);
}

void launch_snr(const SnrArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(k_snr_frames, dim3((a.n_ens * a.n_frames + 63) / 64), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_snr, dim3((a.n_ens + 63) / 64), dim3(64), 0, s, a);
}

} // namespace dabphy
