// welle.io_amd/csrc/k_tii.hip -- transmitter identification (TII), the optional side path of the synchroniser.
//
// Replaces (reference file:line, relative to src/backend):
//   OFDMProcessor::run hand-over   ofdm-processor.cpp:381-386,462-466  prs = first T_u samples of the synchronised frame,
//                                                                      NULL = the T_null samples pulled after symbol 75
//   TIIDecoder::run                tii-decoder.cpp:189-334             two FFTs, pair products over the 4 carrier blocks,
//                                                                      threshold against the PRS power, comb/pattern vote
//   TIIDecoder::analyse_phase      tii-decoder.cpp:336-383             error of every candidate delay, 5-frame sums, winner
//
// The reference decoder has its own thread and drops (NULL, PRS) pairs while it is busy; here every demodulated frame is
// analysed.  k_tii_measure is parallel over (ensemble, frame): it ends with this frame's float error per candidate delay
// for each likely comb/pattern pair.  k_tii_accumulate walks the frames of an ensemble in order, because the reference's
// sums carry from frame to frame: uint64 sums updated through float (its map is unordered_map<float, uint64_t>), the
// winner after 5 measurements chosen like std::min_element walks that map (rank table from the host's own container).
// Float expressions keep the reference's operand order; rotators come from a host-built table (same libm as the
// reference), atan2f/hypotf are the restated glibc routines of dabphy_common.h.
#include "fft2048.h"
#include "dabphy_kernels.h"
#include <dabphy_wave_ops.h>
#include "mix2048.h"

namespace dabphy {

__device__ __forceinline__ int tii_bin(int k) { return k < 0 ? T_U + k : k; }           // k_to_ix, tii-decoder.cpp:343-347

__global__ void __launch_bounds__(FFT_THREADS) k_tii_measure(TiiArgs A)
{
    const int f = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    __shared__ __attribute__((aligned(16))) cf32 nfft[T_U];      // FFT work tile, then the NULL spectrum in bin order
    __shared__ __attribute__((aligned(16))) cf32 pfft[T_U];      // PRS spectrum in bin order
    __shared__ __attribute__((aligned(16))) cf32 twB[FFT_TWB_ENTRIES];
    __shared__ uint8_t det[192];
    __shared__ int s_cnt[FFT_THREADS];
    __shared__ int s_list[TII_MAX_LIKELY + 1];
    __shared__ float s_phase[32];
    __shared__ int s_k[32];

    const size_t fi = (size_t)b * A.n_frames + f;
    int32_t* const likely = A.likely + fi * (1 + TII_MAX_LIKELY);
    const FrameDesc d = A.desc[fi];
    if (d.valid != 1) { if (t == 0) likely[0] = 0; return; }
    const cf32* __restrict__ iq = A.iq + (size_t)b * A.iq_stride;
    FftTwiddles w;
    fft_load_twiddles(w, A.tab.tw, twB, t);
    cf32 v[16];
    // PRS: the first T_u samples of the synchronised frame (ofdm-processor.cpp:383-386; TIIDecoder::run :244-245)
    load_mix2048(v, iq, A.ring, d.pos, d.start_index, A.tab.nco, d.L0, d.f_prs, d.start_index, t);
    fft2048_wg<false>(v, nfft, w, t);
#pragma unroll
    for (int j = 0; j < 16; j++) pfft[t + 128 * j] = v[j];
    // NULL symbol without its cyclic prefix (:229-237): samples 608 .. 2655 of the T_null pulled after the last data symbol
    load_mix2048(v, iq, A.ring, d.pos, (int64_t)d.start_index + T_U + 75LL * T_S + (T_NULL - T_U), A.tab.nco, d.null_L, d.null_f, T_NULL - T_U, t);
    fft2048_wg<false>(v, nfft, w, t);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) nfft[t + 128 * j] = v[j];
    __syncthreads();

    // :272-306 pair products summed over the four carrier blocks, thresholded against the PRS power of the pair's odd carrier
    for (int i = t; i < 192; i += FFT_THREADS) {
        const cf32 p = pfft[1 + 2 * i];
        const float pw = p.re * p.re + p.im * p.im;                                   // std::norm
        cf32 bm; bm.re = 0.0f; bm.im = 0.0f;
        const int k_start[4] = {T_U - 768, T_U - 384, 1, 385};
#pragma unroll
        for (int g = 0; g < 4; g++) bm = cadd(bm, cmul(nfft[k_start[g] + 2 * i], cconj(nfft[k_start[g] + 2 * i + 1])));
        det[i] = hypotf_exact(bm.re, bm.im) > pw * 0.4f;
    }
    __syncthreads();
    // :308-323 a comb/pattern pair is likely when all four of its carriers 1 + 2c + 48b were detected
    constexpr int PER = (24 * 70 + FFT_THREADS - 1) / FFT_THREADS;                    // 14 candidates per thread, ascending
    uint32_t mine = 0;
    for (int q = 0; q < PER; q++) {
        const int cp = PER * t + q;
        if (cp >= 24 * 70) break;
        const int c = cp / 70; const uint32_t pat = A.pattern[cp % 70];
        int cnt = 0;
#pragma unroll
        for (int bb = 0; bb < 8; bb++) cnt += ((pat >> (7 - bb)) & 1u) && det[c + 24 * bb];
        if (cnt >= 4) mine |= 1u << q;
    }
    s_cnt[t] = __popc(mine);
    __syncthreads();
    int before = 0, total = 0;
    for (int u = 0; u < FFT_THREADS; u++) { const int n = s_cnt[u]; total += n; if (u < t) before += n; }
    if (total > TII_MAX_LIKELY) { if (t == 0) likely[0] = 0; return; }               // :327 "threshold is wrong", skip the frame
    for (int q = 0; q < PER; q++) if ((mine >> q) & 1u) s_list[before++] = PER * t + q;
    __syncthreads();
    if (t == 0) likely[0] = total;
    if (t < total) likely[1 + t] = s_list[t];

    for (int l = 0; l < total; l++) {
        const int cp = s_list[l], comb = cp / 70;
        const uint32_t pat = A.pattern[cp % 70];
        if (t < 16) {
            // CombPattern::generateCarriers (:106-129), sorted: block g, m-th set pattern bit, the pair (k, k + 1);
            // both carriers of a pair take the PRS phase of the first (:351-358)
            const int g = t >> 2, m = t & 3;
            int bb = 0, seen = 0;
            for (int x = 0; x < 8; x++) if ((pat >> (7 - x)) & 1u) { if (seen == m) bb = x; seen++; }
            const int off[4] = {-769, -385, 0, 384};
            const int k = 1 + 2 * comb + 48 * bb + off[g];
            const cf32 p = pfft[tii_bin(k)];
            const float ph = fdlibm_atan2f(p.im, p.re);
            s_k[2 * t] = k; s_k[2 * t + 1] = k + 1;
            s_phase[2 * t] = ph; s_phase[2 * t + 1] = ph;
        }
        __syncthreads();
        float* const out = A.abs_err + (fi * TII_MAX_LIKELY + l) * TII_NERR;
        for (int e = t; e < TII_NERR; e += FFT_THREADS) {
            float abs_err = 0.0f;
            for (int j = 0; j < 32; j++) {
                const int k = s_k[j];
                const cf32 r = cmul(nfft[tii_bin(k)], A.rot[(size_t)(k + 768) * TII_NERR + e]);
                const float delta = fdlibm_atan2f(r.im, r.re) - s_phase[j];
                abs_err += fabsf(delta);
            }
            out[e] = abs_err;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(512) k_tii_accumulate(TiiArgs A)
{
    const int b = blockIdx.x, t = threadIdx.x;
    __shared__ int s_slot;
    __shared__ unsigned long long red[512];
    TiiSlot* const slots = A.state + (size_t)b * TII_SLOTS;
    int n_ev = 0, dropped = 0;
    for (int f = 0; f < A.n_frames; f++) {
        const size_t fi = (size_t)b * A.n_frames + f;
        const int32_t* likely = A.likely + fi * (1 + TII_MAX_LIKELY);
        const int nl = likely[0];
        for (int l = 0; l < nl; l++) {
            const int cp = likely[1 + l];
            if (t == 0) {
                int s = -1;
                for (int i = 0; i < TII_SLOTS; i++) if (slots[i].cp1 == cp + 1) { s = i; break; }
                if (s < 0) for (int i = 0; i < TII_SLOTS; i++) if (slots[i].cp1 == 0) { s = i; slots[i].cp1 = cp + 1; break; }
                s_slot = s;
            }
            __syncthreads();
            const int s = s_slot;
            if (s < 0) { dropped++; __syncthreads(); continue; }
            TiiSlot* const slot = slots + s;
            const int num = slot->num + 1, cycle = slot->cycle;
            unsigned long long acc = 0;
            if (t < TII_NERR) {
                // meas.error_per_correction[err] += abs_err with a uint64 value and a float increment (tii-decoder.h:97, .cpp:362)
                acc = (unsigned long long)((float)slot->acc[t] + A.abs_err[(fi * TII_MAX_LIKELY + l) * TII_NERR + t]);
                slot->acc[t] = acc;
            }
            __syncthreads();                                                        // everyone has read num / cycle
            if (num >= 5) {
                // :368-386 min_element over the map: smallest sum, the container's iteration order among equals
                red[t] = t < TII_NERR ? (acc << 9) | (unsigned long long)A.rank[cycle * TII_NERR + t] : ~0ull;
                __syncthreads();
                for (int st = 256; st > 0; st >>= 1) {
                    if (t < st) { const unsigned long long o = red[t + st]; if (o < red[t]) red[t] = o; }
                    __syncthreads();
                }
                const unsigned long long best = red[0];
                if (t < TII_NERR && ((acc << 9) | (unsigned long long)A.rank[cycle * TII_NERR + t]) == best && n_ev < A.max_events) {
                    TiiEvent ev;
                    ev.frame = f; ev.comb = cp / 70; ev.pattern = cp % 70;
                    ev.delay_samples = (int)(float)(t - 4);                         // best->first: the float key
                    ev.error = (float)acc;                                          // best->second
                    A.events[(size_t)b * A.max_events + n_ev] = ev;
                }
                n_ev++;
                if (t < TII_NERR) slot->acc[t] = 0;                                 // clear()
                if (t == 0) { slot->num = 0; slot->cycle = 1; }
            } else if (t == 0) slot->num = num;
            __syncthreads();
        }
    }
    if (t == 0) { A.n_events[b] = n_ev; if (A.overflow) A.overflow[b] += dropped; }
}

void launch_tii(const TiiArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(k_tii_measure, dim3(a.n_frames, a.n_ens), dim3(FFT_THREADS), 0, s, a);
    hipLaunchKernelGGL(k_tii_accumulate, dim3(a.n_ens), dim3(512), 0, s, a);
}

} // namespace dabphy
