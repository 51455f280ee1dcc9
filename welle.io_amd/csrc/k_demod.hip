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

// Load the 2048 samples of one symbol's useful part in round-A order and apply the oscillator.
//   off   : offset of the first wanted sample from the frame's sync-buffer start (d.pos)
//   v[8h + j] = x[t + 128h + 256j] * osc[phase]
__device__ __forceinline__ void load_mix(cf32 (&v)[16], const cf32* __restrict__ iq, int64_t ring, const FrameDesc& d,
                                         int32_t off, const cf32* __restrict__ nco, int mix, int t)
{
    // sample n of the symbol sits at ring index (d.pos + off + n) mod ring
    int64_t a0 = (d.pos + off + t) % ring;
#pragma unroll
    for (int i = 0; i < 16; i++) {                       // n = t + 128 i
        int64_t a = a0 + 128 * i;
        if (a >= ring) a -= ring;
        v[(i & 1) * 8 + (i >> 1)] = iq[a];               // n = t + 128h + 256j  <->  i = h + 2j
    }
    if (!mix) return;
    // phase of sample at offset j from d.pos (ofdm-processor.cpp:211-214, closed form of the running update):
    //   j <  J0: (L0 - (j+1) f_prs) mod RATE        J0 = start_index + T_u
    //   j >= J0: (L1 - (j-J0+1) f_sym) mod RATE
    const int32_t J0 = d.start_index + T_U;
    int32_t L, f; int64_t rel;
    if (off < J0) { L = d.L0; f = d.f_prs; rel = off; } else { L = d.L1; f = d.f_sym; rel = off - J0; }
    if (f == 0) {
        const cf32 o = nco[L];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = cmul(v[i], o);
        return;
    }
    int32_t ph = mod_rate((int64_t)L - (rel + t + 1) * (int64_t)f);
    const int32_t step = mod_rate(128 * (int64_t)f);
    cf32 o[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        o[i] = nco[ph];
        ph -= step; if (ph < 0) ph += INPUT_RATE;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) v[(i & 1) * 8 + (i >> 1)] = cmul(v[(i & 1) * 8 + (i >> 1)], o[i]);
}

__global__ void __launch_bounds__(FFT_THREADS) k_demod(DemodArgs A)
{
    __shared__ __attribute__((aligned(16))) cf32 tile[T_U];
    __shared__ __attribute__((aligned(16))) int8_t softbuf[SOFT_PER_SYM];
    const int t = threadIdx.x;
    const int chunk = blockIdx.x, f = blockIdx.y, b = blockIdx.z;
    const FrameDesc d = A.desc[(size_t)b * A.n_frames + f];
    if (!d.valid) return;
    const cf32* __restrict__ iq = A.iq + (size_t)b * A.iq_stride;
    const int s_begin = 1 + chunk * A.chunk_len;                 // first data symbol of this chunk
    int s_end = s_begin + A.chunk_len; if (s_end > L_SYM) s_end = L_SYM;
    if (s_begin >= L_SYM) return;

    FftTwiddles w; fft_load_twiddles(w, A.tab.tw, t);
    int16_t sidx[16];
#pragma unroll
    for (int j = 0; j < 16; j++) sidx[j] = A.tab.bin2soft[t + 128 * j];

    cf32 prev[16], v[16];
    // offset (from d.pos) of the useful part of symbol s: PRS at start_index; s >= 1 at J0 + (s-1) T_s + T_g
    const int32_t J0 = d.start_index + T_U;
    {
        const int sref = s_begin - 1;
        const int32_t off = sref == 0 ? d.start_index : J0 + (sref - 1) * T_S + T_G;
        load_mix(v, iq, A.ring, d, off, A.tab.nco, A.mix, t);
        fft2048_wg<false>(v, tile, w, t);
#pragma unroll
        for (int j = 0; j < 16; j++) prev[j] = v[j];
        if (sref == 0 && A.prs_mag) {
            // |bin| of the PRS for the SNR estimate (ofdm-decoder.cpp:240-266): stored in bin order, summed by k_snr
            float* pm = A.prs_mag + ((size_t)b * A.n_frames + f) * T_U;
#pragma unroll
            for (int j = 0; j < 16; j++) pm[t + 128 * j] = hypotf_exact(v[j].re, v[j].im);
        }
    }
    const size_t slot = (size_t)((d.frame_no) % A.soft_ring);
    int8_t* soft_frame = A.soft + ((size_t)b * A.soft_ring + slot) * SOFT_PER_FRAME;
    cf32* con_frame = A.con ? A.con + ((size_t)b * A.n_frames + f) * 1200 : nullptr;

    for (int s = s_begin; s < s_end; s++) {
        load_mix(v, iq, A.ring, d, J0 + (s - 1) * T_S + T_G, A.tab.nco, A.mix, t);
        fft2048_wg<false>(v, tile, w, t);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int idx = sidx[j];
            if (idx >= 0) {
                const cf32 r1 = cmul(v[j], cconj(prev[j]));                 // ofdm-decoder.cpp:206
                const float ab1 = 127.0f / l1norm(r1);                        // :208
                const float vr = (-r1.re) * ab1, vi = (-r1.im) * ab1;        // :211-212
                // float -> int8: C truncation; NaN (r1 == 0 -> inf * 0) becomes 0 as with cvttss2si on the reference's x86-64 build
                softbuf[idx] = (vr != vr) ? (int8_t)0 : (int8_t)(int)vr;
                softbuf[K_CARR + idx] = (vi != vi) ? (int8_t)0 : (int8_t)(int)vi;
                if (con_frame && (idx % 96) == 0) con_frame[(s - 1) * 16 + idx / 96] = r1;   // :214-216
            }
            prev[j] = v[j];                                                   // :207
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

// SNR estimate of OfdmDecoder::get_snr(method 1) + the 0.7/0.3 IIR and the every-11th-frame report
// (ofdm-decoder.cpp:154-158,240-266).  One thread per ensemble walks its frames in order; the float sums
// run in the reference's order so the int16 truncation of the dB difference sees the same value.
__global__ void k_snr(SnrArgs A)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= A.n_ens) return;
    RxState& st = A.state[b];
    float snr = st.snr; int cnt = st.snr_count;
    for (int f = 0; f < A.n_frames; f++) {
        const FrameDesc& d = A.desc[(size_t)b * A.n_frames + f];
        float* out = A.snr_out + (size_t)b * A.n_frames + f;
        *out = __int_as_float(0x7fc00000);                     // NaN = "no report for this frame"
        if (!d.valid) continue;
        const float* v = A.prs_mag + ((size_t)b * A.n_frames + f) * T_U;
        float noise = 0, signal = 0;
        const int low = T_U / 2 - K_CARR / 2, high = low + K_CARR;
        for (int i = 70; i < low - 20; i++) noise += v[(T_U / 2 + i) % T_U];
        for (int i = high + 20; i < high + 120; i++) noise += v[(T_U / 2 + i) % T_U];
        noise /= (low - 90 + 100);
        for (int i = T_U / 2 - K_CARR / 4; i < T_U / 2 + K_CARR / 4; i++) signal += v[(T_U / 2 + i) % T_U];
        const float qs = ((signal / (K_CARR / 2)) + 1.0f) / 256.0f, qn = (noise + 1.0f) / 256.0f;   // MathHelper.h:43-46
        const float dB_signal = (float)(20 * log10((double)qs));
        const float dB_noise = (float)(20 * log10((double)qn));
        const int16_t snr_new = (int16_t)(dB_signal - dB_noise);
        snr = (float)(0.7 * snr + 0.3 * snr_new);
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
    hipLaunchKernelGGL(k_snr, dim3((a.n_ens + 63) / 64), dim3(64), 0, s, a);
}

} // namespace dabphy
