// welle.io_amd/csrc/k_sync.hip -- time/frequency synchronisation of OFDMProcessor::run, one work-group per ensemble.
//
// Replaces (reference file:line, relative to src/backend):
//   OFDMProcessor::run, notSynced .. SyncOnEndNull      ofdm-processor.cpp:249-319     -> k_acquire
//   OFDMProcessor::run, SyncOnPhase .. ReadyForNewFrame ofdm-processor.cpp:324-490     -> k_sync_frame
//   OFDMProcessor::getSample(s) oscillator              ofdm-processor.cpp:145-224     (closed-form phase, table gather)
//   PhaseReference::findIndex                           phasereference.cpp:73-256      (ThresholdBeforePeak, StrongestPeak)
//   OFDMProcessor::processPRS (PatternOfZeros)          ofdm-processor.cpp:537-616
//
// The chain frame -> frame is inherently serial per ensemble (the window position and the fine corrector of
// frame n+1 depend on frame n), so the batch dimension is the ensemble: B work-groups per launch, one launch
// per frame step.  Everything that decides an integer (window index, int16 corrector) is computed in the
// reference's operand order: the cyclic-prefix correlation is 37 800 float complex additions in sequence --
// the products are formed in parallel, the additions are done by one lane in order.
#include "fft2048.h"
#include "dabphy_kernels.h"
#include <dabphy_wave_ops.h>
#include "osc_exact.h"
#include "mix2048.h"

namespace dabphy {

__device__ __forceinline__ float block_max(float x, float* red, int t)
{
    __syncthreads();
    red[t] = x;
    __syncthreads();
    for (int s = FFT_THREADS / 2; s > 0; s >>= 1) {
        if (t < s) red[t] = fmaxf(red[t], red[t + s]);
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

__device__ __forceinline__ int block_min_int(int x, int* red, int t)
{
    __syncthreads();
    red[t] = x;
    __syncthreads();
    for (int s = FFT_THREADS / 2; s > 0; s >>= 1) {
        if (t < s) red[t] = red[t] < red[t + s] ? red[t] : red[t + s];
        __syncthreads();
    }
    const int r = red[0];
    __syncthreads();
    return r;
}

// The frame chain, two launches per frame on the sync stream:
//   k_sync_find(frame)      B work-groups: PRS window search (+ coarse corrector); leaves the descriptor "pending" (valid = 2)
//   k_sync_finish(frame)    B work-groups: forms the cyclic-prefix products of the pending frame and adds them in the reference's
//                           order (two chains of 37 800 dependent float additions), updates the fine/coarse correctors, consumes
//                           the null symbol, commits the state
// Both kernels are kept small (<= 19 KiB LDS) so that they find room next to the decode kernels of the previous batch that run
// concurrently on the main stream.
// ofdm-processor.cpp:447-490: correctors, null symbol, state commit (one thread)
// OFDMProcessor::sLevel is advanced by every sample getSample(s) hands out (ofdm-processor.cpp:174,216) but only ever read by the
// null-symbol search after a loss of lock (:284,:303).  While tracking, the synchroniser therefore just records what was pulled
// (one descriptor per window search); k_acquire replays those samples through the recurrence when it is next needed.
template <class State>
__device__ __forceinline__ void hist_append(const SyncArgs& A, int b, State& st, const FrameDesc& d)
{
    if (!A.hist) return;
    FrameDesc* h = A.hist + (size_t)b * A.hist_cap;
    if (st.hist_count == A.hist_cap) {                       // full: forget the oldest
        st.hist_head = (st.hist_head + 1) % A.hist_cap; st.hist_count--; st.hist_dropped = 1;
    }
    h[(st.hist_head + st.hist_count) % A.hist_cap] = d;
    st.hist_count++;
}

// ofdm-processor.cpp:450-451: fineCorrector (int16) += 0.1 * arg(FreqCorr) / M_PI * (carrierDiff / 2)
__device__ __forceinline__ int32_t fine_from_arg(int32_t fine_old, float a)
{
    return (int32_t)(int16_t)((double)fine_old + 0.1 * (double)a / M_PI * (1000 / 2));
}

// next float towards +inf (up) or -inf
__device__ __forceinline__ float f32_step(float x, bool up)
{
    union { float f; uint32_t u; } v; v.f = x;
    if ((v.u & 0x7fffffffu) == 0) { v.u = up ? 1u : 0x80000001u; return v.f; }
    const bool neg = (v.u >> 31) != 0;
    v.u += (neg != up) ? 1u : 0xffffffffu;                  // away from zero when the step and the sign agree
    return v.f;
}
__device__ __forceinline__ float f32_down(double x) { float f = (float)x; if ((double)f > x) f = f32_step(f, false); return f; }
__device__ __forceinline__ float f32_up(double x) { float f = (float)x; if ((double)f < x) f = f32_step(f, true); return f; }

// The only thing the reference takes from FreqCorr is the int16 it adds to the fine corrector.  The float sums it accumulates in
// index order differ from the exact sums by at most E = u Q / (1 - (n + 1) u), Q >= sum over all prefixes |S_k|, u = 2^-24 (each
// addition errs by at most u times its own result; Higham, "Accuracy and Stability of Numerical Algorithms", section 4.2, with the
// computed prefixes bounded by the exact ones plus E).  Q comes from block sums: a prefix that ends inside block b is at most
// |sum of the blocks before b| + sum of the magnitudes inside b.  atan2 is monotone along the edges of a box that avoids the
// origin and the branch cut, atan2f is within 2 ulp of it, and the corrector expression is monotone in the angle.  So evaluating
// it at the ends of the interval decides the int16 whenever both ends agree -- all but about one frame in 5000 -- and otherwise the
// caller falls back to the ordered float sums.  blk[b][0..3] = sum re, sum im, sum |re|, sum |im| of block b (double precision, any
// order), m[b] = products in block b.  Returns true when decided.
constexpr int FIN_BLOCK_ROWS = 8, FIN_BLOCKS = (75 + FIN_BLOCK_ROWS - 1) / FIN_BLOCK_ROWS;
__device__ __forceinline__ bool fine_decided(int32_t fine_old, const double (*blk)[4], int32_t& fine_new)
{
    constexpr double n = 75.0 * 504.0, u = 0x1p-24;
    double sre = 0.0, sim = 0.0, are = 0.0, aim = 0.0, qre = 0.0, qim = 0.0;
    for (int b = 0; b < FIN_BLOCKS; b++) {
        const double m = 504.0 * ((b + 1) * FIN_BLOCK_ROWS <= 75 ? FIN_BLOCK_ROWS : 75 - b * FIN_BLOCK_ROWS);
        qre += m * (fabs(sre) + blk[b][2]); qim += m * (fabs(sim) + blk[b][3]);
        sre += blk[b][0]; sim += blk[b][1]; are += blk[b][2]; aim += blk[b][3];
    }
    constexpr double k = u / (1.0 - (n + 1.0) * u) * (1.0 + 0x1p-30);
    constexpr double dsum = n * 0x1p-51;                              // this path's own (double precision) summation errors, generously
    const double ere = k * qre * (1.0 + dsum) + dsum * are + 1e-30, eim = k * qim * (1.0 + dsum) + dsum * aim + 1e-30;
    const double r_lo = sre - ere, r_hi = sre + ere, i_lo = sim - eim, i_hi = sim + eim;
    if (!(r_lo > -1e30) || !(r_hi < 1e30) || !(i_lo > -1e30) || !(i_hi < 1e30)) return false;    // (also NaN)
    const float xl = f32_down(r_lo), xh = f32_up(r_hi), yl = f32_down(i_lo), yh = f32_up(i_hi);
    // the box must avoid the origin and the branch cut: then atan2 is monotone along every edge and its extremes sit in the corners
    if (!(xl > 0.0f || yl > 0.0f || yh < 0.0f)) return false;
    const float c0 = fdlibm_atan2f(yl, xl), c1 = fdlibm_atan2f(yl, xh), c2 = fdlibm_atan2f(yh, xl), c3 = fdlibm_atan2f(yh, xh);
    float a_lo = fminf(fminf(c0, c1), fminf(c2, c3)), a_hi = fmaxf(fmaxf(c0, c1), fmaxf(c2, c3));
#pragma unroll
    for (int i = 0; i < 8; i++) { a_lo = f32_step(a_lo, false); a_hi = f32_step(a_hi, true); }    // atan2f's own error (< 2 ulp) at the corners and at the true point, with room for a binade change
    const int32_t n_lo = fine_from_arg(fine_old, a_lo), n_hi = fine_from_arg(fine_old, a_hi);
    fine_new = n_lo;
    return n_lo == n_hi;
}

// What a window search starts from: the members of RxState the tracking loop reads and writes.
struct SyncIn { int64_t pos, frame_no; int32_t local_phase, coarse, fine, synced; };
template <class State>
__device__ __forceinline__ SyncIn sync_in_of(const State& g)
{
    SyncIn s; s.pos = g.pos; s.frame_no = g.frame_no; s.local_phase = g.local_phase; s.coarse = g.coarse; s.fine = g.fine; s.synced = g.synced;
    return s;
}
// The WIDE synchroniser (k_sync_find_wide / k_sync_finish_wide / k_sync_validate) works on all frames of a batch at once, each
// from the state its predecessors leave behind IF they behave as a receiver in lock does: window index T_g (the search of frame n + 1
// starts T_null behind the end of frame n, whatever index frame n found), correctors unchanged (the fine corrector moves by
// (int16)(0.1 x the residual offset in Hz): zero once the residual is below 10 Hz, ofdm-processor.cpp:450-451).  Then every frame
// pulls exactly T_F samples at coarse + fine Hz.  k_sync_validate walks the frames in order and accepts a frame only if the state
// its predecessors REALLY left equals the one it was computed from; the rest of the batch goes through the serial chain.
__device__ __forceinline__ SyncIn sync_predict(const SyncIn& base, int n)
{
    SyncIn s = base;
    s.pos += (int64_t)n * T_F; s.frame_no += n;
    s.local_phase = mod_rate64((int64_t)base.local_phase - (int64_t)n * T_F * (int64_t)(base.coarse + base.fine));
    return s;
}

// descriptor of a frame slot before its window search (also what a slot without a frame keeps)
__device__ __forceinline__ FrameDesc desc_begin(const SyncIn& st)
{
    FrameDesc d;
    d.pos = st.pos; d.frame_no = st.frame_no; d.start_index = -1; d.L0 = st.local_phase; d.f_prs = st.coarse + st.fine;
    d.L1 = 0; d.f_sym = 0; d.valid = 0; d.fine_after = st.fine; d.coarse_after = st.coarse; d.null_L = 0; d.null_f = 0; d.coarse_ran = 0; d.exact_sums = 0;
    d.osc_hazard[0] = d.osc_hazard[1] = d.osc_hazard[2] = 0; d.coarse_step = 0;
    return d;
}
// samples a window search needs in the ring: a whole frame with the largest possible window index
constexpr int64_t SYNC_NEED = (int64_t)T_U + (T_U - 1) + 75 * (int64_t)T_S + T_NULL;

// ofdm-processor.cpp:447-490 in two halves.  finish_desc: the new fine corrector completes the frame's descriptor (the null symbol,
// :462-463, is pulled with it).  state_advance: what the frame leaves in the receiver state, from its finished descriptor alone.
__device__ __forceinline__ void finish_desc(FrameDesc& d, int32_t fine)
{
    const int32_t coarse = d.coarse_after;                            // after the coarse corrector of this frame
    d.null_L = mod_rate64((int64_t)d.L1 - (int64_t)75 * T_S * d.f_sym);
    d.null_f = coarse + fine;
    d.fine_after = fine; d.coarse_after = coarse;                     // as RadioControllerInterface sees them after the frame
    d.valid = 1;
}
template <class State>
__device__ __forceinline__ void state_advance(const SyncArgs& A, const int b, State& st, const FrameDesc& d)
{
    int32_t coarse = d.coarse_after, fine = d.fine_after;
    if (fine > 1000 / 2) { coarse += 1000; fine -= 1000; }            // :478-486
    else if (fine < -1000 / 2) { coarse -= 1000; fine += 1000; }
    hist_append(A, b, st, d);
    st.calm_frames = d.start_index == T_G ? (st.calm_frames < (1 << 20) ? st.calm_frames + 1 : st.calm_frames) : 0;
    st.pos = d.pos + (int64_t)d.start_index + T_U + 75 * (int64_t)T_S + T_NULL;
    st.local_phase = mod_rate64((int64_t)d.null_L - (int64_t)T_NULL * d.null_f);
    st.coarse = coarse; st.fine = fine; st.frame_no = d.frame_no + 1;
}

template <bool WIDE>
__device__ __forceinline__ void sync_finish_commit(const SyncArgs& A, const int b, FrameDesc& dfin, int32_t fine, int exact, const uint32_t (*hz)[3])
{
    FrameDesc d = dfin;
    finish_desc(d, fine);
    d.exact_sums = exact;
    for (int i = 0; i < A.tab.n_osc_unsafe; i++) { d.osc_hazard[0] |= hz[i][0]; d.osc_hazard[1] |= hz[i][1]; d.osc_hazard[2] |= hz[i][2]; }
    dfin = d;
    if (!WIDE) {
        RxState& st = A.state[b];          // updated field by field (the struct carries the 64-entry envelope history)
        if (exact) st.n_exact_sums += 1;
        state_advance(A, b, st, d);
    }
}

// ---- finish with its own product stage: the same work-group forms the cyclic-prefix products (waves 2-3, two rows ahead,
// through a shared 3-row LDS ring) while waves 0-1 fold them in the reference's order.  Replaces k_cp_products + the HBM round
// trip of its 75 x 512 products per ensemble and frame + one launch per frame of the chain.
struct CpRow { cf32 lo[4], hi[4]; };                 // raw samples buf[j], buf[T_u + j], j = tp + 128 k

// Per-thread walk over the symbols of a frame: ring index and oscillator of buf[tp] of the current row, advanced by `stride`
// symbols at a time.  The oscillator value is carried in double precision (osc_exact.h) and advanced by complex multiplication;
// it is re-anchored with osc_exp every 8 rows, so no chain is longer than 8 + 3 + 1 multiplications (error < 6e-15, a twentieth
// of osc_round's margin).  The step factors are the same for every thread: they live in SGPRs.
struct CpSteps { dc64 d128, dTU, dRow; int32_t step128, stepTU, stepRow; int64_t aRow; };
struct CpWalk { int64_t a; dc64 e; int32_t ph; int n; };

__device__ __forceinline__ CpSteps cp_steps(const SyncArgs& A, const FrameDesc& d, int stride)
{
    CpSteps c;
    const dc64 x = osc_step(128, d.f_sym), y = osc_step(T_U, d.f_sym), z = osc_step((int64_t)stride * T_S, d.f_sym);
    c.d128.re = uniform_f64(x.re); c.d128.im = uniform_f64(x.im); c.dTU.re = uniform_f64(y.re); c.dTU.im = uniform_f64(y.im);
    c.dRow.re = uniform_f64(z.re); c.dRow.im = uniform_f64(z.im);
    c.step128 = mod_rate64(128LL * d.f_sym); c.stepTU = mod_rate64((int64_t)T_U * d.f_sym); c.stepRow = mod_rate64((int64_t)stride * T_S * d.f_sym);
    c.aRow = ((int64_t)stride * T_S) % A.ring;
    return c;
}

// start at symbol sy0: buf[tp] sits J0 + sy0 T_s + tp samples behind the sync buffer start and carries phase L1 - (sy0 T_s + tp + 1) f
__device__ __forceinline__ CpWalk cp_walk_begin(const SyncArgs& A, const FrameDesc& d, int sy0, int tp)
{
    CpWalk w;
    w.a = (d.pos + d.start_index + T_U + (int64_t)sy0 * T_S + tp) % A.ring;
    w.ph = mod_rate64((int64_t)d.L1 - ((int64_t)sy0 * T_S + tp + 1) * (int64_t)d.f_sym);
    w.e = osc_exp(w.ph);
    w.n = 0;
    return w;
}

__device__ __forceinline__ void cp_row_fetch(const SyncArgs& A, const CpSteps& c, const cf32* __restrict__ iq, int64_t& a_row, int tp, CpRow& r)
{
    int64_t a = a_row;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (tp + 128 * k < T_G) {
            int64_t a_hi = a + T_U; if (a_hi >= A.ring) a_hi -= A.ring;
            r.lo[k] = iq[a]; r.hi[k] = iq[a_hi];
        }
        a += 128; if (a >= A.ring) a -= A.ring;
    }
    a_row += c.aRow; if (a_row >= A.ring) a_row -= A.ring;
}

// buf[i] * conj(buf[i - T_u]) of the walk's current symbol (ofdm-processor.cpp:436-441), oscillator applied as getSamples does
// (:211-214); p[k] = product j = tp + 128 k (zero beyond the 504 of the guard interval).  Advances the oscillator to the next row.
__device__ __forceinline__ void cp_row_products(const SyncArgs& A, const CpSteps& c, CpWalk& w, int tp, const CpRow& r, cf32 (&p)[4])
{
    const cf32* __restrict__ nco = A.tab.nco;
    dc64 e = w.e;
    int32_t ph = w.ph;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        cf32 o_lo, o_hi;
        const uint32_t hard_lo = osc_round(e, o_lo), hard_hi = osc_round(osc_mul(e, c.dTU), o_hi);
        if (hard_lo | hard_hi) {                                            // about one sample in 10^5: the table decides
            int32_t ph_hi = ph - c.stepTU; if (ph_hi < 0) ph_hi += INPUT_RATE;
            if (hard_lo) o_lo = nco[ph];
            if (hard_hi) o_hi = nco[ph_hi];
        }
        p[k].re = 0.0f; p[k].im = 0.0f;
        if (tp + 128 * k < T_G) p[k] = cmul(cmul(r.hi[k], o_hi), cconj(cmul(r.lo[k], o_lo)));
        if (k < 3) { e = osc_mul(e, c.d128); ph -= c.step128; if (ph < 0) ph += INPUT_RATE; }
    }
    w.ph -= c.stepRow; if (w.ph < 0) w.ph += INPUT_RATE;
    if (++w.n == 8) { w.e = osc_exp(w.ph); w.n = 0; }
    else w.e = osc_mul(w.e, c.dRow);
}

__device__ __forceinline__ void cp_row_emit(const SyncArgs& A, const CpSteps& c, CpWalk& w, int tp, const CpRow& r, cf32* row)
{
    cf32 p[4];
    cp_row_products(A, c, w, tp, r, p);
#pragma unroll
    for (int k = 0; k < 4; k++) if (tp + 128 * k < T_G) row[tp + 128 * k] = p[k];
}

constexpr int FINISH_THREADS = 256;
template <bool WIDE>
__device__ __forceinline__ void sync_finish_body(const SyncArgs& A, const int b, const int frame)
{
    __shared__ __attribute__((aligned(16))) cf32 ring[3 * 512];          // product rows sy, sy+1, sy+2 (504 used of 512)
    __shared__ float s_sum;
    const int t = threadIdx.x;
    FrameDesc& dfin = A.desc[(size_t)b * A.n_frames + frame];
    const int pending = dfin.valid;
    __syncthreads();
    if (pending != 2) return;
    const FrameDesc d = dfin;
    const cf32* __restrict__ iq = A.iq + (size_t)b * A.iq_stride;
    // which symbols of this frame read an oscillator table entry next to a float rounding boundary (osc_exact.h): one lane per entry
    // solves its congruence; the masks are OR-ed into the descriptor at the commit (there are barriers in between)
    __shared__ uint32_t s_hz[OSC_MAX_UNSAFE][3];
    if (t < A.tab.n_osc_unsafe) {
        uint32_t m[3] = {0u, 0u, 0u};
        osc_hazard_entry(m, A.tab.osc_unsafe[t], d.start_index, d.L0, d.f_prs, d.L1, d.f_sym);
        s_hz[t][0] = m[0]; s_hz[t][1] = m[1]; s_hz[t][2] = m[2];
    }
    {
        // ---- fast path: all four waves form the products, sums in double precision, the int16 decided by interval (fine_decided)
        __shared__ int s_decided, s_fine;
        const int tq = t & 127, grp = t >> 7;                                     // the two halves take alternate symbols
        __shared__ double s_blk[FIN_BLOCKS][4];
        if (t < FIN_BLOCKS * 4) (&s_blk[0][0])[t] = 0.0;
        __syncthreads();
        double sre = 0.0, sim = 0.0, are = 0.0, aim = 0.0;
        const CpSteps cs2 = cp_steps(A, d, 2);
        CpWalk w = cp_walk_begin(A, d, grp, tq);
        CpRow cur, nxt;
        cp_row_fetch(A, cs2, iq, w.a, tq, cur);
        int cur_blk = 0;
#pragma unroll 1
        for (int sy = grp; sy < 75; sy += 2) {
            if (sy + 2 < 75) cp_row_fetch(A, cs2, iq, w.a, tq, nxt);
            if ((sy / FIN_BLOCK_ROWS) != cur_blk) {
                lds_add_f64(&s_blk[cur_blk][0], sre); lds_add_f64(&s_blk[cur_blk][1], sim); lds_add_f64(&s_blk[cur_blk][2], are); lds_add_f64(&s_blk[cur_blk][3], aim);
                sre = sim = are = aim = 0.0; cur_blk = sy / FIN_BLOCK_ROWS;
            }
            cf32 p[4];
            cp_row_products(A, cs2, w, tq, cur, p);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                sre += (double)p[k].re; sim += (double)p[k].im; are += fabs((double)p[k].re); aim += fabs((double)p[k].im);
            }
            cur = nxt;
        }
        lds_add_f64(&s_blk[cur_blk][0], sre); lds_add_f64(&s_blk[cur_blk][1], sim); lds_add_f64(&s_blk[cur_blk][2], are); lds_add_f64(&s_blk[cur_blk][3], aim);
        __syncthreads();
        if (t == 0) {
            int32_t fine_new = 0;
            s_decided = fine_decided(A.state[b].fine, s_blk, fine_new) ? 1 : 0;
            s_fine = fine_new;
        }
        __syncthreads();
        if (s_decided) {
            if (t == 0) sync_finish_commit<WIDE>(A, b, dfin, s_fine, 0, s_hz);
            return;
        }
    }
    // ---- exact path: the reference's ordered float sums (the interval straddled a multiple of pi/50, or the signal is too weak to bound)
    const bool producer = t >= 128;
    const int tp = t - 128, wv = t >> 6;
    float acc = 0.0f;
    CpRow cur, nxt;
    const CpSteps cs1 = cp_steps(A, d, 1);
    CpWalk w1;
    if (producer) {
        w1 = cp_walk_begin(A, d, 0, tp);
        cp_row_fetch(A, cs1, iq, w1.a, tp, cur); cp_row_fetch(A, cs1, iq, w1.a, tp, nxt);
        cp_row_emit(A, cs1, w1, tp, cur, ring);
        cp_row_emit(A, cs1, w1, tp, nxt, ring + 512);
        cp_row_fetch(A, cs1, iq, w1.a, tp, nxt);
    }
    __syncthreads();
#pragma unroll 1
    for (int sy = 0; sy < 75; sy++) {
        if (!producer) {
            // wave 0 adds the real parts, wave 1 the imaginary parts (ofdm-processor.cpp:435-442 in index order).  Lanes 0..15
            // each hold one product of a block of 16; lane 0 folds them in order with row_shl DPP reads (one instruction per addition).
            const float* q = reinterpret_cast<const float*>(ring + (sy % 3) * 512) + wv;       // +0: re, +1: im
            const int l16 = t & 15;
            float xs[32];
#pragma unroll
            for (int i = 0; i < 32; i++) xs[i] = (16 * i + l16 < T_G) ? q[2 * (16 * i + l16)] : 0.0f;
#pragma unroll
            for (int i = 0; i < 32; i++) acc = chain16(acc, xs[i], (i < 31) ? 16 : 8);         // 504 = 31 * 16 + 8
        } else if (sy + 2 < 75) {
            cur = nxt;
            if (sy + 3 < 75) cp_row_fetch(A, cs1, iq, w1.a, tp, nxt);                          // row sy+3, in flight while row sy+2 is formed
            cp_row_emit(A, cs1, w1, tp, cur, ring + ((sy + 2) % 3) * 512);                     // its slot held row sy-1, consumed an iteration ago
        }
        __syncthreads();
    }
    if (t == 64) s_sum = acc;
    __syncthreads();
    // (A.state[b].fine is the fine corrector the frame was searched with in both modes: the wide pass predicts it unchanged)
    if (t == 0) sync_finish_commit<WIDE>(A, b, dfin, fine_from_arg(A.state[b].fine, fdlibm_atan2f(s_sum, acc)), 1, s_hz);
}

#ifndef SYNC_FINISH_OCC
#define SYNC_FINISH_OCC 1
#endif
__global__ void __launch_bounds__(FINISH_THREADS, SYNC_FINISH_OCC) k_sync_finish(SyncArgs A)
{
    __builtin_amdgcn_s_setprio(3);
    if (A.redo_from && A.frame < A.redo_from[blockIdx.x]) return;     // accepted from the wide pass
    sync_finish_body<false>(A, blockIdx.x, A.frame);
}
// the wide pass: work-group (frame, ensemble); throughput work, no priority
#ifndef SYNC_FINISH_WIDE_OCC
#define SYNC_FINISH_WIDE_OCC SYNC_FINISH_OCC
#endif
__global__ void __launch_bounds__(FINISH_THREADS, SYNC_FINISH_WIDE_OCC) k_sync_finish_wide(SyncArgs A)
{
    sync_finish_body<true>(A, blockIdx.y, blockIdx.x);
}

// ---- sLevel catches up with the samples that were pulled while tracking (ofdm-processor.cpp:216: once per sample, float result
// of a double expression).  Exact when the history reaches back to the last point at which the level was exact; otherwise two
// runs from the extremes of what the level can be bracket it (the update is monotone in the level): if they have met by the end,
// that is the level.  One work-group per ensemble; s_st = its state (LDS), l1 = a tile of |re| + |im| values (LDS).
template <int TILE, int NT>
__device__ __forceinline__ void slevel_replay(const SyncArgs& A, const int b, RxState& s_st, float* l1, const cf32* __restrict__ iq, const cf32* __restrict__ nco, const int t)
{
    const FrameDesc* hist = A.hist + (size_t)b * A.hist_cap;
    __shared__ float s_lo, s_hi;
    __shared__ int s_first;
    if (t == 0) {
        int first = 0; bool dropped = s_st.hist_dropped != 0;
        if (!A.loop)                                                         // samples that have left the ring cannot be replayed
            for (int i = s_st.hist_count - 1; i >= 0; i--) if (hist[(s_st.hist_head + i) % A.hist_cap].pos < A.n_valid - A.ring) { first = i + 1; dropped = true; break; }
        s_first = first;
        // (the upper end of the bracket: sLevel is a running convex combination of |re| + |im| of oscillator-corrected samples, so a bound
        // on those bounds it: 2 for streams that only ever held converted u8 / s8 / s16 samples -- then ~10 frames of history certify the
        // level, against ~50 from the 3e38 an unbounded cf32 stream has to assume)
        s_lo = dropped ? 0.0f : s_st.s_level; s_hi = dropped ? A.level_max : s_st.s_level;
    }
    __syncthreads();
    for (int e = s_first; e < s_st.hist_count; e++) {
        const FrameDesc d = hist[(s_st.hist_head + e) % A.hist_cap];
        // what one window search pulled: T_u + start_index samples at f_prs; then, if it succeeded, 75 symbols at f_sym and the null
        // symbol at null_f (ofdm-processor.cpp:337-344,371-374,432-434,462-463)
        const int nseg = d.valid == 1 ? 3 : 1;
        for (int sgm = 0; sgm < nseg; sgm++) {
            const int64_t off0 = sgm == 0 ? 0 : sgm == 1 ? (int64_t)T_U + d.start_index : (int64_t)T_U + d.start_index + 75LL * T_S;
            const int64_t n = sgm == 0 ? (int64_t)T_U + (d.valid == 1 ? d.start_index : 0) : sgm == 1 ? 75LL * T_S : T_NULL;
            const int32_t L = sgm == 0 ? d.L0 : sgm == 1 ? d.L1 : d.null_L, f = sgm == 0 ? d.f_prs : sgm == 1 ? d.f_sym : d.null_f;
            for (int64_t i0 = 0; i0 < n; i0 += TILE) {
                const int m = (int)((n - i0 < TILE) ? n - i0 : TILE);
                for (int i = t; i < m; i += NT) l1[i] = l1norm(mixed_sample(iq, A.ring, d.pos, off0 + i0 + i, nco, L, f, i0 + i));
                __syncthreads();
                if (t == 0) {
                    float lo = s_lo, hi = s_hi;
                    const bool one = lo == hi;                                   // exact start: a single chain
                    int i = 0;
                    for (; i + 16 <= m; i += 16) {                               // operands fetched 16 at a time, off the dependent chain
                        const float4 q0 = *reinterpret_cast<const float4*>(l1 + i), q1 = *reinterpret_cast<const float4*>(l1 + i + 4),
                                     q2 = *reinterpret_cast<const float4*>(l1 + i + 8), q3 = *reinterpret_cast<const float4*>(l1 + i + 12);
                        const float v[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
                        double a[16];
#pragma unroll
                        for (int k = 0; k < 16; k++) a[k] = 0.00001 * (double)v[k];
#pragma unroll
                        for (int k = 0; k < 16; k++) {
                            lo = (float)(a[k] + (1 - 0.00001) * (double)lo);
                            if (!one) hi = (float)(a[k] + (1 - 0.00001) * (double)hi);
                        }
                    }
                    for (; i < m; i++) {
                        const double a = 0.00001 * (double)l1[i];
                        lo = (float)(a + (1 - 0.00001) * (double)lo);
                        if (!one) hi = (float)(a + (1 - 0.00001) * (double)hi);
                    }
                    if (one) hi = lo;
                    s_lo = lo; s_hi = hi;
                }
                __syncthreads();
            }
        }
    }
    if (t == 0) {
        if (s_lo != s_hi) s_st.n_relock_inexact += 1;
        s_st.s_level = s_hi;
        s_st.hist_count = 0; s_st.hist_head = 0; s_st.hist_dropped = 0;
    }
    __syncthreads();
}

// ---- acquisition: OFDMProcessor::run from "Initing" / notSynced to SyncOnPhase (ofdm-processor.cpp:249-319).
// A strictly per-sample recurrence (sLevel IIR evaluated in double, 50-sample moving sum); the work-group stages |re|+|im| of the
// oscillator-corrected samples in LDS (l1: TILE floats, s_st: the ensemble's state), lane 0 walks the state machine.
// Runs at the head of k_sync_find for an ensemble that is not synchronised -- after the start of a stream and after every failed
// window search, in whatever slot of a batch that happens, like the reference falls back to notSynced (:347-350) -- so the frame
// chain stays at two launches per frame and a tracking ensemble pays one load for it.
constexpr int ACQ_TILE = 1024;
template <int NT>
__device__ __forceinline__ void acquire_body(const SyncArgs& A, const int b, float* l1, RxState& s_st, int& s_done, const int t)
{
    constexpr int TILE = ACQ_TILE;
    const cf32* __restrict__ iq = A.iq + (size_t)b * A.iq_stride;
    const cf32* __restrict__ nco = A.tab.nco;
    if (t == 0) { s_st = A.state[b]; s_done = 0; }
    __syncthreads();

    if (s_st.hist_count > 0 && A.hist) slevel_replay<TILE, NT>(A, b, s_st, l1, iq, nco, t);

    for (;;) {
        const int64_t pos = s_st.pos; const int32_t L = s_st.local_phase;
        // getSample(0) while priming / taking the first 50 samples, getSample(coarse+fine) while searching (:254,:269,:286,:305)
        const int32_t f = (s_st.acq_phase >= 2) ? s_st.coarse + s_st.fine : 0;
        int64_t avail = A.loop ? TILE : A.n_valid - pos;
        if (avail > TILE) avail = TILE;
        if (avail <= 0) break;                                       // starved: state is kept for the next call
        for (int i = t; i < (int)avail; i += NT) l1[i] = l1norm(mixed_sample(iq, A.ring, pos, i, nco, L, f, i));
        __syncthreads();
        if (t == 0) {
            RxState& st = s_st;
            float sLevel = st.s_level, cs = st.acq_cs;
            int ph = st.acq_phase, idx = st.acq_idx, counter = st.acq_counter, left = st.acq_left;
            int i = 0; bool done = false;
            for (;;) {
                // loop conditions are evaluated before a sample is pulled (:284, :303)
                if (ph == 2 && !((double)(cs / 50) > 0.50 * (double)sLevel)) { ph = 3; counter = 0; }
                if (ph == 3 && !((double)(cs / 50) < 0.75 * (double)sLevel)) { done = true; break; }
                if (i >= (int)avail) break;
                const float a = l1[i++];
                sLevel = (float)(0.00001 * (double)a + (1 - 0.00001) * (double)sLevel);         // :174
                if (ph == 0) {                                                                  // :252-255
                    if (--left <= 0) { ph = 1; idx = 0; cs = 0.0f; st.attempts += 1; }                  // falls through into notSynced (:256)
                } else if (ph == 1) {                                                           // :268-273
                    st.env[idx & 63] = a; cs += a; idx++;
                    if (idx == 50) { ph = 2; counter = 0; break; }                              // oscillator changes -> new tile
                } else {                                                                        // :285-297 / :304-316
                    st.env[idx & 63] = a;
                    cs += a - st.env[(idx - 50) & 63];
                    idx = (idx + 1) & 32767;
                    counter++;
                    if ((ph == 2 && counter > T_F) || (ph == 3 && counter > T_NULL + 50)) {     // hopeless -> notSynced
                        ph = 1; idx = 0; cs = 0.0f; st.attempts += 1; break;
                    }
                }
            }
            st.s_level = sLevel; st.acq_cs = cs; st.acq_phase = ph; st.acq_idx = idx; st.acq_counter = counter; st.acq_left = left;
            st.pos = pos + i;
            st.local_phase = (f == 0) ? L : mod_rate64((int64_t)L - (int64_t)i * f);
            if (done) { st.synced = 1; st.acq_phase = 1; st.acq_idx = 0; st.acq_cs = 0.0f; st.acq_counter = 0; }
            s_done = done ? 1 : 0;
        }
        __syncthreads();
        if (s_done) break;
    }
    if (t == 0) A.state[b] = s_st;
    __threadfence();
    __syncthreads();                                     // the caller re-reads the state from memory
}

// stand-alone acquisition launch (no longer part of the frame chain; kept for diagnostics)
__global__ void __launch_bounds__(256) k_acquire(SyncArgs A)
{
    __shared__ __attribute__((aligned(16))) float l1[ACQ_TILE];
    __shared__ RxState s_st;
    __shared__ int s_done;
    if (A.state[blockIdx.x].synced) return;
    acquire_body<256>(A, blockIdx.x, l1, s_st, s_done, threadIdx.x);
}

// MODE 0: the serial chain (the state in HBM is the truth: acquisition first where needed, a failed search falls back to notSynced);
// MODE 1: the wide pass (frame n from the state sync_predict gives it); MODE 2: the find chain (k_sync_find_chain: the caller hands in the
// state the frame starts from and, when the search succeeded, gets back the state the NEXT frame starts from if the fine corrector does not
// move).  Returns true when the descriptor is left pending (valid = 2) for k_sync_finish.
constexpr int SYNC_CALM_MIN = 8;
constexpr int SYNC_CHAIN_ROUNDS = 3;
#ifdef SYNC_CHAIN_TS
// (timing experiment: where a window search of the find chain spends its time beside the decoder -- 100 MHz stamps of ensemble 0's searches,
// printed when the handle is destroyed with DABPHY_CHAIN_TS set; build with EXTRA="-DDABPHY_EXPERIMENTS -DSYNC_CHAIN_TS")
__device__ unsigned long long g_chain_ts[64][8];
#define CHAIN_TS(k) do { if (MODE == 2 && b == 0 && threadIdx.x == 0) g_chain_ts[frame & 63][k] = wall_clock64(); } while (0)
#else
#define CHAIN_TS(k) do { } while (0)
#endif
template <int MODE>
__device__ __forceinline__ bool sync_find_body(const SyncArgs& A, const int b, const int frame, SyncIn& chain_st)
{
    constexpr bool WIDE = MODE != 0;
    // 17 KiB of LDS, reused phase by phase (FFT tile -> |IFFT| + window maxima)
    __shared__ __attribute__((aligned(16))) cf32 tile[T_U + 96];
    float* const lbuf = reinterpret_cast<float*>(tile);              // [T_U + 128], valid after the inverse transform
    float* const pa = lbuf + (T_U + 128);                            // [T_U]
    __shared__ float redf[FFT_THREADS];
    __shared__ int redi[FFT_THREADS];
    __shared__ float s_sum;
    __shared__ __attribute__((aligned(16))) cf32 twB[FFT_TWB_ENTRIES];
    const int t = threadIdx.x;
    const cf32* __restrict__ iq = A.iq + (size_t)b * A.iq_stride;
    const cf32* __restrict__ nco = A.tab.nco;
    if (!WIDE && !A.state[b].synced) {
        // notSynced: null-symbol search first (the FFT tile is free until the window search: it holds the sample tile and the state)
        static_assert(sizeof(tile) >= ACQ_TILE * sizeof(float) + sizeof(RxState) + 16, "acquisition scratch must fit the FFT tile");
        float* const l1 = reinterpret_cast<float*>(tile);
        RxState& s_st = *reinterpret_cast<RxState*>(l1 + ACQ_TILE);
        acquire_body<FFT_THREADS>(A, b, l1, s_st, redi[0], t);
    }
    const SyncIn st = MODE == 2 ? chain_st : MODE == 1 ? sync_predict(sync_in_of(A.state[b]), frame) : sync_in_of(A.state[b]);
    FrameDesc& dout = A.desc[(size_t)b * A.n_frames + frame];
    FrameDesc d = desc_begin(st);

    // a whole frame (with the largest possible window index) must be available
    if (!st.synced || (!A.loop && st.pos + SYNC_NEED > A.n_valid)) {
        if (t == 0) dout = d;
        return false;
    }
    // the wide pass leaves an ensemble whose window is moving to the find chain: its predicted positions would be wrong from the first
    // slip on (k_sync_validate then stops at this slot, which keeps valid = 0)
    if (MODE == 1 && A.state[b].calm_frames < SYNC_CALM_MIN) {
        if (t == 0) dout = d;
        return false;
    }

    CHAIN_TS(0);
    FftTwiddles w; fft_load_twiddles(w, A.tab.tw, twB, t);
    cf32 v[16], u[16];
    // ---- PhaseReference::findIndex (phasereference.cpp:73-92): FFT, multiply by conj(refTable), IFFT (scaled by 1/N)
    load_mix2048(v, iq, A.ring, st.pos, 0, nco, d.L0, d.f_prs, 0, t);
#ifdef SYNC_CHAIN_TS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    CHAIN_TS(1);
    fft2048_wg<false>(v, tile, w, t);
    CHAIN_TS(2);
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = cmul(v[j], cconj(A.tab.ref[t + 128 * j]));
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int j = 0; j < 8; j++) u[8 * h + j] = v[h + 2 * j];          // bin t + 128 (h + 2j) = input t + 128h + 256j
    fft2048_wg<true>(u, tile, w, t);
    CHAIN_TS(3);
    __syncthreads();                                                       // all round-C reads of the tile are done: it becomes lbuf / pa
    const float factor = 1.0f / (float)T_U;                                // fft.cpp:154
    float* cir = A.cir ? A.cir + ((size_t)b * A.n_frames + frame) * T_U : nullptr;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const float a = hypotf_exact(u[j].re * factor, u[j].im * factor);  // phasereference.cpp:214-215
        lbuf[t + 128 * j] = a;
        if (cir) cir[t + 128 * j] = a;
    }
    if (t < 128) lbuf[T_U + t] = 0.0f;
    __syncthreads();
    CHAIN_TS(4);

    int startIndex = -1;
    if (A.fft_placement == 0) {
        // StrongestPeak (phasereference.cpp:99-129): sum in order, first maximum
        if (t == 0) { float s = 0; for (int i = 0; i < T_U; i++) s += lbuf[i]; s_sum = s; }
        float mx = -10000.0f;
#pragma unroll
        for (int j = 0; j < 16; j++) mx = fmaxf(mx, lbuf[16 * t + j]);
        const float gmax = block_max(mx, redf, t);
        int cand = T_U;
        for (int j = 15; j >= 0; j--) if (lbuf[16 * t + j] == gmax) cand = 16 * t + j;
        const int first = block_min_int(cand, redi, t);
        const float sum = s_sum;
        if (sum == 0) startIndex = -1;
        else if (gmax < 3 * sum / T_U) startIndex = (int)(-fabsf(gmax * T_U / sum) - 1);
        else startIndex = first;
    } else if (A.fft_placement == 1) {
        // EarliestPeakWithBinning (phasereference.cpp:125-211): peaks over 102 bins of 20 samples (2040..2047 are never
        // looked at), the highest peak, the 4 highest bins within 500 samples of it, those above 3 * mean, the earliest.
        float* const bval = pa; int* const bidx = reinterpret_cast<int*>(pa + 128);
        if (t == 0) { float s = 0; for (int i = 0; i < 2040; i++) s += lbuf[i]; s_sum = s; }    // `mean += value` in index order
        if (t < 102) {
            float pv = 0.0f; int pi = -1;
            for (int j = 0; j < 20; j++) { const float v2 = lbuf[20 * t + j]; if (v2 > pv) { pv = v2; pi = 20 * t + j; } }
            bval[t] = pv; bidx[t] = pi;
        }
        if (cir && t < 8) cir[2040 + t] = 0.0f;                                                  // the reference's buffer keeps its zeros there
        __syncthreads();
        if (t == 0) {
            const float mean = s_sum / T_U;
            // std::sort by value (descending) is replaced by selection: the order among exactly equal peaks is the bin order
            int top = 0;
            for (int k = 1; k < 102; k++) if (bval[k] > bval[top]) top = k;
            const int peak_index = bidx[top];
            unsigned long long used_lo = 0, used_hi = 0;
            int found = 0, mn = 0;
            for (int pass = 0; pass < 4; pass++) {
                int best = -1;
                for (int k = 0; k < 102; k++) {
                    const bool used = k < 64 ? (used_lo >> k) & 1 : (used_hi >> (k - 64)) & 1;
                    const int dist = bidx[k] - peak_index;
                    if (used || (dist < 0 ? -dist : dist) > 500) continue;
                    if (best < 0 || bval[k] > bval[best]) best = k;
                }
                if (best < 0) break;
                if (best < 64) used_lo |= 1ull << best; else used_hi |= 1ull << (best - 64);
                if (bval[best] < 3 * mean) continue;
                if (!found || bidx[best] < mn) { mn = bidx[best]; found = 1; }
            }
            redi[0] = found ? mn : -1;
        }
        __syncthreads();
        startIndex = redi[0];
        __syncthreads();
    } else {
        // ThresholdBeforePeak (phasereference.cpp:212-252)
        if (t == 0) {                                                                          // :214-218, in order
            float s = 0; const float4* l4 = reinterpret_cast<const float4*>(lbuf);
            for (int i = 0; i < T_U / 4; i += 4) {
                const float4 q0 = l4[i], q1 = l4[i + 1], q2 = l4[i + 2], q3 = l4[i + 3];
                s += q0.x; s += q0.y; s += q0.z; s += q0.w; s += q1.x; s += q1.y; s += q1.z; s += q1.w;
                s += q2.x; s += q2.y; s += q2.z; s += q2.w; s += q3.x; s += q3.y; s += q3.z; s += q3.w;
            }
            s_sum = s;
        }
        // peak_averages[i] = max(lbuf[i .. i+99]) for i < 1948; thread t owns i = 16t .. 16t+15
        float mx = -10000.0f;
        if (16 * t < T_U - 100) {
            float common = -10000.0f;                                   // lbuf[16t+15 .. 16t+99]
            for (int k = 16 * t + 15; k <= 16 * t + 99; k++) common = fmaxf(common, lbuf[k]);
            float suf[16];                                              // suf[k] = max(lbuf[16t+k .. 16t+14])
            float run = -10000.0f;
#pragma unroll
            for (int k = 14; k >= 0; k--) { run = fmaxf(run, lbuf[16 * t + k]); suf[k] = run; }
            suf[15] = -10000.0f;
            run = -10000.0f;                                            // prefix over lbuf[16t+100 .. 16t+99+k]
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int i = 16 * t + k;
                if (k > 0) run = fmaxf(run, lbuf[16 * t + 99 + k]);
                float m = fmaxf(common, suf[k]);
                if (k > 0) m = fmaxf(m, run);
                if (i + 100 < T_U) { pa[i] = m; mx = fmaxf(mx, m); } else pa[i] = 0.0f;
            }
        } else {
            for (int k = 0; k < 16; k++) pa[16 * t + k] = 0.0f;
        }
        const float gmax = block_max(mx, redf, t);                       // contains the barriers that publish pa / s_sum
        const float sum = s_sum;
        int cand = T_U;
        if (gmax > 3 * sum / T_U) {                                      // :238-239
            const float thresh = gmax / 2;
            for (int k = 15; k >= 0; k--) {
                const int i = 16 * t + k;
                if (i + 100 < T_U && pa[i + 100] > thresh) cand = i;    // :241-245
            }
        }
        const int first = block_min_int(cand, redi, t);
        startIndex = first < T_U ? first : -1;
    }

    CHAIN_TS(5);
    if (startIndex < 0) {
        // ofdm-processor.cpp:347-350: SyncOnPhase failed -> notSynced (the 2048 samples are consumed)
        if (t == 0) {
            d.start_index = startIndex; d.valid = 3;   // window search failed: SyncOnPhase -> notSynced
            dout = d;
        }
        if (!WIDE && t == 0) {                         // (a wide slot that fails is left to the serial chain: k_sync_validate)
            RxState& g = A.state[b];
            g.pos = st.pos + T_U;
            g.local_phase = mod_rate64((int64_t)d.L0 - (int64_t)T_U * d.f_prs);
            g.synced = 0; g.lost = g.lost + 1; g.attempts = g.attempts + 1;                  // goto notSynced (:347-350 -> :256-262)
            hist_append(A, b, g, d);                   // the T_u samples of the failed attempt were pulled too
        }
        return false;
    }
    d.start_index = startIndex;
    if (!WIDE && t == 0 && A.state[b].first_lock_attempts < 0) A.state[b].first_lock_attempts = A.state[b].attempts;    // ofdm-processor.cpp:351-355
    const int32_t J0 = startIndex + T_U;
    d.L1 = mod_rate64((int64_t)d.L0 - (int64_t)J0 * d.f_prs);

    // ---- coarse frequency corrector (ofdm-processor.cpp:397-409, processPRS :537-616 PatternOfZeros)
    int32_t coarse = st.coarse;
    if (!A.disable_coarse && A.dec[b].fic_ratio * 10 < 50) {
        d.coarse_ran = 1;
        load_mix2048(v, iq, A.ring, st.pos, startIndex, nco, d.L0, d.f_prs, startIndex, t);
        fft2048_wg<false>(v, tile, w, t);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) tile[t + 128 * j] = v[j];          // natural bin order
        __syncthreads();
        if (A.freqsync != 2) {
            // FreqsyncMethod::GetMiddle (0, ofdm-processor.cpp:618-644) / CorrelatePRS (1, :547-581): rarely selected, only run
            // while the FIC does not decode -- one thread walks them exactly as written (ordered float sums, the int-abs quirk)
            if (t == 0) {
                int correction;
                if (A.freqsync == 0) {
                    float sum = 0, oldMax = 0; int maxIndex = 0;
                    for (int i = 40; i < K_CARR + 40; i++) { const cf32 z = tile[(T_U / 2 + i) % T_U]; sum += hypotf_exact(z.re, z.im); }
                    for (int i = 40; i < T_U - (K_CARR - 40); i++) {
                        const cf32 z0 = tile[(T_U / 2 + i) % T_U], z1 = tile[(T_U / 2 + i + K_CARR) % T_U];
                        sum -= hypotf_exact(z0.re, z0.im);
                        sum += hypotf_exact(z1.re, z1.im);
                        if (sum > oldMax) { sum = oldMax; maxIndex = i; }         // sic (the reference resets the running sum)
                    }
                    correction = maxIndex - (T_U - K_CARR) / 2;
                } else {
                    // (24 + 96 floats in the reduction scratch, not in a thread's own arrays: dynamically indexed locals are scratch memory)
                    static_assert(24 + 72 + 24 <= FFT_THREADS, "CorrelatePRS work arrays fit the reduction scratch");
                    float* const refArg = redf; float* const cv = redf + 24;
                    for (int i = 0; i < 24; i++) {
                        const cf32 z = cmul(A.tab.ref[(T_U + i) % T_U], cconj(A.tab.ref[(T_U + i + 1) % T_U]));
                        refArg[i] = fdlibm_atan2f(z.im, z.re);
                    }
                    for (int i = 0; i < 72 + 24; i++) {
                        const int base = T_U - 36 + i;
                        const cf32 z = cmul(tile[base % T_U], cconj(tile[(base + 1) % T_U]));
                        cv[i] = fdlibm_atan2f(z.im, z.re);
                    }
                    float MMax = 0; int index = 100;
                    for (int i = 0; i < 72; i++) {
                        float sum = 0;
                        for (int j = 0; j < 24; j++) {
                            sum += (float)abs((int)(refArg[j] * cv[i + j]));       // ::abs(int): the product is truncated first
                            if (sum > MMax) { MMax = sum; index = i; }
                        }
                    }
                    correction = T_U - 36 + index - T_U;
                }
                redi[0] = correction;
            }
            __syncthreads();
            const int correction = redi[0];
            __syncthreads();
            if (correction != 100) {
                coarse += correction * 1000;
                if (abs(coarse) > 35000) coarse = 0;
            }
        } else {
        float sum = 3.0e38f; int idx = 1 << 20;
        if (t < 72) {
            const int i = T_U - 36 + t;
#define FB(k) tile[(k) % T_U]
#define ARG(a, c) ([&] { const cf32 z_ = cmul(FB(a), cconj(FB(c))); return fdlibm_atan2f(z_.im, z_.re); }())
            // the reference's unqualified abs() binds to ::abs(int): arguments are truncated to int first
            // (disassembly of the -O2 build: cvttsd2si / cvttss2si); oracle/dabphy_oracle.c pins this.
            const float a1 = (float)abs(abs((int)((double)ARG(i + 1, i + 2) / M_PI)) - 1);
            const float a2 = (float)abs(abs((int)((double)ARG(i + 2, i + 3) / M_PI)) - 1);
            const float a3 = (float)abs((int)ARG(i + 3, i + 4));
            const float a4 = (float)abs((int)ARG(i + 4, i + 5));
            const float a5 = (float)abs((int)ARG(i + 5, i + 6));
            const float b1 = (float)abs(abs((int)((double)ARG(i + 17, i + 19) / M_PI)) - 1);
            const float b2 = (float)abs((int)ARG(i + 19, i + 20));
            const float b3 = (float)abs((int)ARG(i + 20, i + 21));
            const float b4 = (float)abs((int)ARG(i + 21, i + 22));
#undef ARG
#undef FB
            sum = a1 + a2 + a3 + a4 + a5 + b1 + b2 + b3 + b4;
            idx = i;
        }
        // first index with the smallest sum (strict '<' scan in ascending i, Mmin starts at 1000)
        const float neg_min = block_max(-sum, redf, t);
        const int first = block_min_int((-sum == neg_min) ? idx : (1 << 20), redi, t);
        {
            const int index = (-neg_min < 1000.0f) ? first : 100;       // "int16_t index = 100" when nothing beat Mmin = 1000
            const int correction = index - T_U;
            if (correction != 100) {                                     // :403 (always true, kept for fidelity)
                coarse += correction * 1000;
                if (abs(coarse) > 35000) coarse = 0;
            }
        }
        }
    }
    d.f_sym = coarse + st.fine;
    d.coarse_after = coarse;
    d.coarse_step = coarse - st.coarse;
    d.valid = 2;                                   // pending: k_sync_finish completes it
    if (t == 0) dout = d;
    if (MODE == 2) {
        // where the next frame starts if the fine corrector stays where it is (finish_desc + state_advance with fine unchanged;
        // k_sync_validate_chain checks exactly that against what k_sync_finish really decided)
        const int32_t null_L = mod_rate64((int64_t)d.L1 - (int64_t)75 * T_S * d.f_sym), null_f = coarse + st.fine;
        chain_st.pos = d.pos + (int64_t)startIndex + T_U + 75 * (int64_t)T_S + T_NULL;
        chain_st.frame_no = d.frame_no + 1;
        chain_st.local_phase = mod_rate64((int64_t)null_L - (int64_t)T_NULL * null_f);
        chain_st.coarse = coarse;
    }
    CHAIN_TS(6);
    return true;
}

#ifndef SYNC_FIND_OCC
#define SYNC_FIND_OCC 2
#endif
__global__ void __launch_bounds__(FFT_THREADS, SYNC_FIND_OCC) k_sync_find(SyncArgs A)
{
    __builtin_amdgcn_s_setprio(3);
    if (A.redo_from && A.frame < A.redo_from[blockIdx.x]) return;     // accepted from the wide pass
    SyncIn none{};
    sync_find_body<0>(A, blockIdx.x, A.frame, none);
}
__global__ void __launch_bounds__(FFT_THREADS, SYNC_FIND_OCC) k_sync_find_wide(SyncArgs A)
{
    SyncIn none{};
    sync_find_body<1>(A, blockIdx.y, blockIdx.x, none);
}
// The find chain: what the wide pass cannot predict is WHERE the window of frame n + 1 lies when the window index moves (a receiver
// whose sampling clock is off by 1 ppm sees it slip every fifth frame, ofdm-processor.cpp:337-350) -- but that needs only the window
// searches in order, not the cyclic-prefix sums: the fine corrector of a receiver in lock does not move (it steps by
// (int16)(0.1 x residual Hz), ofdm-processor.cpp:450-451).  One work-group per ensemble walks the frames the wide pass's judge did not
// accept (redo_out[b] ...), each search from the position the previous one really found, correctors as they are; the sums of all those
// frames then run at once (k_sync_finish_wide again: it takes the pending descriptors) and k_sync_validate_chain accepts the frames
// whose fine corrector indeed stayed.  Results are the serial chain's by construction; what is left (a corrector that moved, a failed
// search, an ensemble out of lock) still goes to the serial chain.
__global__ void __launch_bounds__(FFT_THREADS, 1) k_sync_find_chain(SyncArgs A)      // (one work-group per ensemble, a wave per SIMD: the whole register file, what does not fit the 256 architected registers spills to accumulation registers, not to memory)
{
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.x;
    int n = A.redo_out[b];
    if (n >= A.n_frames) return;
    SyncIn st = sync_in_of(A.state[b]);                 // the true state in front of frame n (k_sync_validate has advanced it)
    bool ok = true;
    for (; n < A.n_frames; n++) {
        if (ok) ok = sync_find_body<2>(A, b, n, st);
        else if (threadIdx.x == 0) A.desc[(size_t)b * A.n_frames + n] = desc_begin(st);      // behind a failed search: nothing to go by (valid = 0)
        __syncthreads();
    }
}

// The wide pass's judge: one thread per ensemble walks the batch in order.  A frame is accepted iff the state its predecessors left
// is the one it was computed from and it is an ordinary tracked frame; accepting it advances the state exactly as the serial chain
// does (state_advance).  redo_out[b] = first slot not settled (n_frames: none).
// CHAIN = false: behind the wide pass proper (frames computed from sync_predict's states);
// CHAIN = true: behind the find chain (frames from redo_out[b] on, each computed from the state its descriptor names: position, frame
// number, oscillator phase, coarse + fine -- and with the fine corrector this ensemble had in front of frame redo_out[b], which
// k_sync_finish_wide used for all of them); *any_redo |= some ensemble still has slots for the serial chain.
// (the judges walk 32 frames per lane: the members of RxState they touch live in registers meanwhile -- as members of the global struct
// every frame cost a chain of dependent memory round trips, 93 us per pass on the step's critical path -- and the next descriptor is
// fetched while the current one is judged)
struct RxHot {
    int64_t pos, frame_no; int32_t local_phase, coarse, fine, synced;
    int32_t n_exact_sums, hist_count, hist_head, hist_dropped, attempts, first_lock_attempts, n_wide_frames, n_chain_frames, calm_frames;
};
template <bool CHAIN>
__device__ __forceinline__ void sync_validate_body(const SyncArgs& A)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= A.n_ens) return;
    RxState& G = A.state[b];
    RxHot g;
    g.pos = G.pos; g.frame_no = G.frame_no; g.local_phase = G.local_phase; g.coarse = G.coarse; g.fine = G.fine; g.synced = G.synced;
    g.n_exact_sums = G.n_exact_sums; g.hist_count = G.hist_count; g.hist_head = G.hist_head; g.hist_dropped = G.hist_dropped; g.attempts = G.attempts;
    g.first_lock_attempts = G.first_lock_attempts; g.n_wide_frames = G.n_wide_frames; g.n_chain_frames = G.n_chain_frames; g.calm_frames = G.calm_frames;
    FrameDesc* const desc = A.desc + (size_t)b * A.n_frames;
    const SyncIn base = sync_in_of(g);
    int n = CHAIN ? A.redo_out[b] : 0;
    if (base.synced && n < A.n_frames) {
        FrameDesc next = desc[n];
        for (; n < A.n_frames; n++) {
            const FrameDesc d = next;
            if (n + 1 < A.n_frames) next = desc[n + 1];
            if (!CHAIN) {
                const SyncIn p = sync_predict(base, n);
                if (g.pos != p.pos || g.frame_no != p.frame_no || g.local_phase != p.local_phase || g.coarse != p.coarse || g.fine != p.fine) break;
            }
            if (!A.loop && g.pos + SYNC_NEED > A.n_valid) {
                // out of samples: this slot and the ones behind it stay empty, as the serial chain leaves them
                const FrameDesc e = desc_begin(sync_in_of(g));
                for (int m = n; m < A.n_frames; m++) desc[m] = e;
                n = A.n_frames;
                break;
            }
            if (CHAIN && (g.pos != d.pos || g.frame_no != d.frame_no || g.local_phase != d.L0 || g.coarse + g.fine != d.f_prs || g.fine != base.fine)) break;
            if (d.valid != 1) break;                                   // failed window search: the serial chain takes it from here
            if (d.exact_sums) g.n_exact_sums += 1;
            if (g.first_lock_attempts < 0) g.first_lock_attempts = g.attempts;                  // ofdm-processor.cpp:351-355
            state_advance(A, b, g, d);
            g.n_wide_frames += 1;
            if (CHAIN) g.n_chain_frames += 1;
        }
        G.pos = g.pos; G.frame_no = g.frame_no; G.local_phase = g.local_phase; G.coarse = g.coarse; G.fine = g.fine;
        G.n_exact_sums = g.n_exact_sums; G.hist_count = g.hist_count; G.hist_head = g.hist_head; G.hist_dropped = g.hist_dropped;
        G.first_lock_attempts = g.first_lock_attempts; G.n_wide_frames = g.n_wide_frames; G.n_chain_frames = g.n_chain_frames; G.calm_frames = g.calm_frames;
    }
    A.redo_out[b] = n;
    if (CHAIN && A.last_round && n < A.n_frames) *A.any_redo = 1;
    if (CHAIN && A.last_round && A.any_chain && base.synced && g.calm_frames < SYNC_CALM_MIN) *A.any_chain = 1;      // this ensemble's window is moving
}
__global__ void __launch_bounds__(64) k_sync_validate(SyncArgs A) { sync_validate_body<false>(A); }
__global__ void __launch_bounds__(64) k_sync_validate_chain(SyncArgs A) { sync_validate_body<true>(A); }

// Continuous mode (dabphy_set_track_slevel): the level follows the tracked frames one by one (3 ms per frame on one lane: meant for
// the single-ensemble real-time receiver, where it is 3 % of a frame's 96 ms) instead of catching up when lock is lost.
__global__ void __launch_bounds__(256) k_slevel_catchup(SyncArgs A)
{
    constexpr int TILE = 1024;
    __shared__ __attribute__((aligned(16))) float l1[TILE];
    __shared__ RxState s_st;
    const int t = threadIdx.x, b = blockIdx.x;
    if (t == 0) s_st = A.state[b];
    __syncthreads();
    if (s_st.hist_count == 0 || !A.hist) return;
    slevel_replay<TILE, 256>(A, b, s_st, l1, A.iq + (size_t)b * A.iq_stride, A.tab.nco, t);
    if (t == 0) {
        RxState& g = A.state[b];
        g.s_level = s_st.s_level; g.hist_count = 0; g.hist_head = 0; g.hist_dropped = 0; g.n_relock_inexact = s_st.n_relock_inexact;
    }
}
void launch_slevel_catchup(const SyncArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(k_slevel_catchup, dim3(a.n_ens), dim3(256), 0, s, a);
}

void launch_sync_wide(const SyncArgs& a, hipStream_t s, hipEvent_t front)
{
    if (a.skip_wide) {
        // the last pass met ensembles whose window moves: the wide searches would only be skipped one by one -- by 8192 work-groups of a
        // 196-register kernel that each have to find a wave slot beside the decoder first (4.5 ms of waiting in front of the find chain:
        // profiles/r06_drift_kernel_stats.csv) -- so the pass starts in the find chain for everybody
        hipError_t e = hipMemsetAsync(a.redo_out, 0, sizeof(int32_t) * (size_t)a.n_ens, s); (void)e;
    } else {
        hipLaunchKernelGGL(k_sync_find_wide, dim3(a.n_frames, a.n_ens), dim3(FFT_THREADS), 0, s, a);
        hipLaunchKernelGGL(k_sync_finish_wide, dim3(a.n_frames, a.n_ens), dim3(FINISH_THREADS), 0, s, a);
        hipLaunchKernelGGL(k_sync_validate, dim3((a.n_ens + 63) / 64), dim3(64), 0, s, a);
    }
    if (front) { hipError_t e = hipEventRecord(front, s); (void)e; }
    // what the judge did not accept: the find chain, the sums of its frames, its judge (work-groups with nothing to do return at once).
    // A fine corrector that moves ends an ensemble's round (its later searches were made with the old one): the next round starts from
    // there.  SYNC_CHAIN_ROUNDS rounds are queued whatever happens; what is left after the last one goes to the serial chain.
    SyncArgs c = a;
    for (int r = 0; r < SYNC_CHAIN_ROUNDS; r++) {
        c.last_round = r == SYNC_CHAIN_ROUNDS - 1;
        hipLaunchKernelGGL(k_sync_find_chain, dim3(c.n_ens), dim3(FFT_THREADS), 0, s, c);
        hipLaunchKernelGGL(k_sync_finish_wide, dim3(c.n_frames, c.n_ens), dim3(FINISH_THREADS), 0, s, c);
        hipLaunchKernelGGL(k_sync_validate_chain, dim3((c.n_ens + 63) / 64), dim3(64), 0, s, c);
    }
}
#ifdef SYNC_CHAIN_TS
void dump_chain_ts()
{
    unsigned long long ts[64][8];
    if (hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_chain_ts), sizeof(ts)) != hipSuccess) return;
    fprintf(stderr, "find chain, ensemble 0, last round that walked each frame [us]: frame: entry->samples  ->fft  ->ifft  ->cir  ->index  ->end | gap to the next frame\n");
    for (int f = 0; f < 32; f++) {
        fprintf(stderr, "chain_ts %2d:", f);
        for (int k = 1; k <= 6; k++) fprintf(stderr, " %7.2f", (double)(long long)(ts[f][k] - ts[f][k - 1]) / 100.0);
        if (f + 1 < 32) fprintf(stderr, " | %7.2f", (double)(long long)(ts[f + 1][0] - ts[f][6]) / 100.0);
        fprintf(stderr, "   total %7.2f\n", (double)(long long)(ts[f][6] - ts[f][0]) / 100.0);
    }
}
#endif
void launch_sync_find(const SyncArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(k_sync_find, dim3(a.n_ens), dim3(FFT_THREADS), 0, s, a);
}
void launch_sync_finish(const SyncArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(k_sync_finish, dim3(a.n_ens), dim3(FINISH_THREADS), 0, s, a);
}
void launch_acquire(const SyncArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(k_acquire, dim3(a.n_ens), dim3(256), 0, s, a);
}

} // namespace dabphy
