// welle.io_amd/csrc/dabphy_getters.hip -- what a caller reads back after a batch (frame info, FIBs, MSC bytes, taps, statistics), stage profiling, the TII side path.
// (split from dabphy_api.hip in round 3; dabphy_internal.h has the map of the translation units)
#include "dabphy_internal.h"

extern "C" {

int dabphy_get_frame_info(dabphy_handle* h, dabphy_frame_info* out)
{
    DeviceBind dev_(h);
    if (!h || !out || !h->last_frames) return DABPHY_ERR_INVALID;
    const size_t n = (size_t)h->cfg.n_ensembles * h->last_frames;
    for (size_t i = 0; i < n; i++) {
        const FrameDesc& d = h->h_desc[i];
        out[i].sample_pos = d.pos; out[i].frame_no = d.frame_no; out[i].start_index = d.start_index; out[i].valid = d.valid;
        out[i].fine_corrector = d.fine_after; out[i].coarse_corrector = d.coarse_after; out[i].snr = h->h_snr[i];
    }
    return DABPHY_OK;
}

int dabphy_get_fibs(dabphy_handle* h, uint8_t* fib, uint8_t* crc_ok)
{
    DeviceBind dev_(h);
    if (!h || !fib || !crc_ok || !h->last_frames) return DABPHY_ERR_INVALID;
    const size_t n = (size_t)h->cfg.n_ensembles * h->last_frames;
    memcpy(fib, h->h_fib, n * 384); memcpy(crc_ok, h->h_ok, n * 12);      // (they crossed PCIe inside dabphy_process)
    return DABPHY_OK;
}

int dabphy_get_fibs_host(dabphy_handle* h, const uint8_t** fib, const uint8_t** crc_ok)
{
    DeviceBind dev_(h);
    if (!h || !fib || !crc_ok || !h->last_frames) return DABPHY_ERR_INVALID;
    *fib = h->h_fib; *crc_ok = h->h_ok;
    return DABPHY_OK;
}

int dabphy_get_fibs_device(dabphy_handle* h, const uint8_t** d_fib, const uint8_t** d_crc_ok)
{
    DeviceBind dev_(h);
    if (!h || !d_fib || !d_crc_ok || !h->last_frames) return DABPHY_ERR_INVALID;
    *d_fib = h->s_fib.as<uint8_t>(); *d_crc_ok = h->s_ok.as<uint8_t>();
    return DABPHY_OK;
}

int dabphy_get_ratio_lag(dabphy_handle* h, int32_t* stale_frames, int64_t* first_stale_frame)
{
    DeviceBind dev_(h);
    if (!h) return DABPHY_ERR_INVALID;
    std::vector<DecState> st(h->cfg.n_ensembles);
    HIPCHK(h, hipMemcpyAsync(st.data(), h->d_dec, st.size() * sizeof(DecState), hipMemcpyDeviceToHost, h->stream));
    int r = sync(h); if (r) return r;
    for (size_t i = 0; i < st.size(); i++) {
        if (stale_frames) stale_frames[i] = st[i].stale_ratio_frames;
        if (first_stale_frame) first_stale_frame[i] = st[i].stale_ratio_frames ? st[i].first_stale_frame : -1;
    }
    return DABPHY_OK;
}

int dabphy_get_ratio_lag_effect(dabphy_handle* h, int32_t* effective_frames, int64_t* first_effective_frame)
{
    DeviceBind dev_(h);
    if (!h) return DABPHY_ERR_INVALID;
    std::vector<DecState> st(h->cfg.n_ensembles);
    HIPCHK(h, hipMemcpyAsync(st.data(), h->d_dec, st.size() * sizeof(DecState), hipMemcpyDeviceToHost, h->stream));
    int r = sync(h); if (r) return r;
    for (size_t i = 0; i < st.size(); i++) {
        if (effective_frames) effective_frames[i] = st[i].effective_stale_frames;
        if (first_effective_frame) first_effective_frame[i] = st[i].effective_stale_frames ? st[i].first_effective_frame : -1;
    }
    return DABPHY_OK;
}

int dabphy_get_scan_stats(dabphy_handle* h, int32_t* attempts, int32_t* attempts_at_first_lock)
{
    DeviceBind dev_(h);
    if (!h) return DABPHY_ERR_INVALID;
    std::vector<RxState> st(h->cfg.n_ensembles);
    if (h->s_desc2[0].p) { int r0 = resolve_all_chains(h); if (r0) return r0; }
    if (h->sync_stream) HIPCHK(h, hipStreamSynchronize(h->sync_stream));
    HIPCHK(h, hipMemcpyAsync(st.data(), h->d_state, st.size() * sizeof(RxState), hipMemcpyDeviceToHost, h->stream));
    int r = sync(h); if (r) return r;
    for (size_t i = 0; i < st.size(); i++) { if (attempts) attempts[i] = st[i].attempts; if (attempts_at_first_lock) attempts_at_first_lock[i] = st[i].first_lock_attempts; }
    return DABPHY_OK;
}

int dabphy_get_replayed_batches(dabphy_handle* h, uint64_t* batches)
{
    if (!h || !batches) return DABPHY_ERR_INVALID;
    *batches = h->n_replayed_batches;
    return DABPHY_OK;
}

int dabphy_get_osc_stats(dabphy_handle* h, uint64_t* unchecked_symbols, uint64_t* checked_symbols)
{
    DeviceBind dev_(h);
    if (!h) return DABPHY_ERR_INVALID;
    unsigned long long v[2] = {0, 0};
    HIPCHK(h, hipMemcpyAsync(v, h->d_osc_stats, sizeof v, hipMemcpyDeviceToHost, h->stream));
    int r = sync(h); if (r) return r;
    if (unchecked_symbols) *unchecked_symbols = v[0];
    if (checked_symbols) *checked_symbols = v[1];
    return DABPHY_OK;
}

int dabphy_get_wide_superframe_stats(dabphy_handle* h, uint64_t* settled, uint64_t* tried)
{
    DeviceBind dev_(h);
    if (!h) return DABPHY_ERR_INVALID;
    unsigned long long v[2] = {0, 0};
    if (h->sf_gf.p) {
        HIPCHK(h, hipMemcpyAsync(v, h->sf_gf.as<uint8_t>() + 512, sizeof v, hipMemcpyDeviceToHost, h->stream));
        int r = sync(h); if (r) return r;
    }
    if (settled) *settled = v[0];
    if (tried) *tried = v[1];
    return DABPHY_OK;
}

int dabphy_get_wide_sync_stats(dabphy_handle* h, int32_t* wide_frames, uint64_t* passes, uint64_t* fallbacks)
{
    DeviceBind dev_(h);
    if (!h) return DABPHY_ERR_INVALID;
    if (h->s_desc2[0].p) { int r0 = resolve_all_chains(h); if (r0) return r0; }
    if (passes) *passes = h->n_wide_passes;
    if (fallbacks) *fallbacks = h->n_wide_fallbacks;
    if (wide_frames) {
        std::vector<RxState> st(h->cfg.n_ensembles);
        if (h->sync_stream) HIPCHK(h, hipStreamSynchronize(h->sync_stream));
        HIPCHK(h, hipMemcpyAsync(st.data(), h->d_state, st.size() * sizeof(RxState), hipMemcpyDeviceToHost, h->stream));
        int r = sync(h); if (r) return r;
        for (size_t i = 0; i < st.size(); i++) wide_frames[i] = st[i].n_wide_frames;
    }
    return DABPHY_OK;
}

int dabphy_get_find_chain_stats(dabphy_handle* h, int32_t* chain_frames)
{
    DeviceBind dev_(h);
    if (!h || !chain_frames) return DABPHY_ERR_INVALID;
    if (h->s_desc2[0].p) { int r0 = resolve_all_chains(h); if (r0) return r0; }
    std::vector<RxState> st(h->cfg.n_ensembles);
    if (h->sync_stream) HIPCHK(h, hipStreamSynchronize(h->sync_stream));
    HIPCHK(h, hipMemcpyAsync(st.data(), h->d_state, st.size() * sizeof(RxState), hipMemcpyDeviceToHost, h->stream));
    int r = sync(h); if (r) return r;
    for (size_t i = 0; i < st.size(); i++) chain_frames[i] = st[i].n_chain_frames;
    return DABPHY_OK;
}

int dabphy_get_sync_stats(dabphy_handle* h, int32_t* lost, int32_t* exact_sums, int32_t* relock_inexact)
{
    DeviceBind dev_(h);
    if (!h) return DABPHY_ERR_INVALID;
    std::vector<RxState> st(h->cfg.n_ensembles);
    if (h->s_desc2[0].p) { int r0 = resolve_all_chains(h); if (r0) return r0; }
    if (h->sync_stream) HIPCHK(h, hipStreamSynchronize(h->sync_stream));
    HIPCHK(h, hipMemcpyAsync(st.data(), h->d_state, st.size() * sizeof(RxState), hipMemcpyDeviceToHost, h->stream));
    int r = sync(h); if (r) return r;
    for (size_t i = 0; i < st.size(); i++) { if (lost) lost[i] = st[i].lost; if (exact_sums) exact_sums[i] = st[i].n_exact_sums; if (relock_inexact) relock_inexact[i] = st[i].n_relock_inexact; }
    return DABPHY_OK;
}

int dabphy_get_fic_ratio(dabphy_handle* h, int32_t* ratio_percent)
{
    DeviceBind dev_(h);
    if (!h || !ratio_percent) return DABPHY_ERR_INVALID;
    std::vector<DecState> st(h->cfg.n_ensembles);
    HIPCHK(h, hipMemcpyAsync(st.data(), h->d_dec, st.size() * sizeof(DecState), hipMemcpyDeviceToHost, h->stream));
    int r = sync(h); if (r) return r;
    for (size_t i = 0; i < st.size(); i++) ratio_percent[i] = st[i].fic_ratio * 10;
    return DABPHY_OK;
}

namespace {
// first_valid / n_rows of one (ensemble, sub-channel) pair as documented in include/dabphy.h (host arithmetic on the batch's descriptors)
void msc_rows_info(const dabphy_handle* h, uint32_t b, dabphy_handle::PairRef w, int32_t* first_valid, int32_t* n_rows)
{
    const uint32_t F = h->last_frames;
    const auto& cls = h->classes[w.cls];
    if (first_valid) {
        // DabAudio emits its first logical frame on the 17th CIF it is fed (dab-audio.cpp:146-149): counted from the CIF at which this
        // sub-channel was selected (0 for one that was there from the start of the stream)
        const int64_t fed = 4 * h->h_desc[(size_t)b * F].frame_no - cls.pairs[w.pair].cif0;
        *first_valid = fed >= 16 ? 0 : (int32_t)(16 - fed);
    }
    if (n_rows) {
        int nv = 0;
        for (uint32_t f = 0; f < F; f++) nv += h->h_desc[(size_t)b * F + f].valid == 1 ? 1 : 0;
        *n_rows = 4 * nv;
    }
}
// rows of one (ensemble, sub-channel) pair from its class's output [pair][4F][nbits / 8]
int msc_rows_of(dabphy_handle* h, uint32_t b, dabphy_handle::PairRef w, uint8_t* out, int32_t* first_valid, int32_t* n_rows)
{
    const uint32_t F = h->last_frames;
    const auto& cls = h->classes[w.cls];
    const size_t bytes = cls.prot.nbits / 8, Rn = (size_t)4 * F;
    HIPCHK(h, hipMemcpyAsync(out, cls.out.as<uint8_t>() + (size_t)w.pair * Rn * bytes, Rn * bytes, hipMemcpyDeviceToHost, h->stream));
    msc_rows_info(h, b, w, first_valid, n_rows);
    return DABPHY_OK;
}
// the bulk drain's layout: class c's output [pairs][4F][bytes] at off[c] (256-byte aligned), one after the other
size_t drain_layout(const dabphy_handle* h, std::vector<size_t>& off)
{
    size_t at = 0;
    off.clear();
    for (const auto& cls : h->classes) {
        off.push_back(at);
        at += (cls.pairs.size() * (size_t)4 * h->last_frames * (cls.prot.nbits / 8) + 255) & ~(size_t)255;
    }
    return at;
}
}

int drain_wait(dabphy_handle* h)
{
    if (!h->drain_pending) return DABPHY_OK;
    h->drain_pending = false;
    HIPCHK(h, hipEventSynchronize(h->ev_drain_done));
    return DABPHY_OK;
}

int dabphy_msc_batch_size(dabphy_handle* h, size_t* buf_bytes, uint32_t* n_desc)
{
    if (!h || !h->last_frames) return DABPHY_ERR_INVALID;
    std::vector<size_t> off;
    const size_t total = drain_layout(h, off);
    if (buf_bytes) *buf_bytes = total;
    if (n_desc) { uint32_t n = 0; for (const auto& w : h->where) n += (uint32_t)w.size(); *n_desc = n; }
    return DABPHY_OK;
}

int dabphy_msc_drain_begin(dabphy_handle* h, dabphy_msc_desc* desc, uint32_t desc_capacity, uint32_t* n_desc, uint8_t* buf, size_t buf_capacity)
{
    DeviceBind dev_(h);
    if (!h || !h->last_frames || (!desc && desc_capacity) || !n_desc) return DABPHY_ERR_INVALID;
    const uint32_t B = h->cfg.n_ensembles, F = h->last_frames;
    std::vector<size_t> off;
    const size_t total = drain_layout(h, off);
    uint32_t n = 0; for (const auto& w : h->where) n += (uint32_t)w.size();
    if (n > desc_capacity || total > buf_capacity || (total && !buf)) { h->err = "dabphy_msc_drain_begin: buffer or index table too small (dabphy_msc_batch_size)"; return DABPHY_ERR_INVALID; }
    int r;
    if ((r = drain_wait(h))) return r;                      // one drain at a time
    if (!h->drain_stream) {
        // Which stream.  The runtime multiplexes a process's streams onto a few hardware queues, and a drain on a stream of its own -- the
        // handle's sixth -- landed on the main stream's queue on some boxes: 13.0-13.8 ms per step with the drain against 10.0 without
        // (tools/probe_streams.py, profiles/r06_step_variants.txt).  On the ingest stream, idle whenever the samples are resident in HBM,
        // it overlaps completely (10.07 ms).  A handle that is fed through dabphy_stream_write_raw_async keeps that stream for its
        // host-to-device transfers and drains on one of its own.
        if (h->s_enqueued == 0 && !(h->stream_layout & 8)) h->drain_stream = h->copy_stream;
        else HIPCHK(h, hipStreamCreateWithFlags(&h->drain_stream, hipStreamNonBlocking));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_drain_done, hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_drain_staged, hipEventDisableTiming));
    }
    // (dabphy_process has returned: the class outputs are final, nothing on the main stream is pending.)  The outputs first go to a
    // staging area in HBM -- a device copy on the main stream, tens of microseconds for a hundred MB, ordered in front of the next batch's
    // kernels like any other work there -- and cross PCIe from the staging area: the next batch's decoders never wait for the host link
    // (the class outputs are theirs again the moment the device copy is done), the drain has a whole step to arrive.
    { const void* before = h->drain_stage.p; if ((r = ensure(h, h->drain_stage, total))) return r; if (h->drain_stage.p != before) HIPCHK(h, hipMemsetAsync(h->drain_stage.p, 0, h->drain_stage.cap, h->stream)); }      // (the padding between two classes crosses to the host too: zeros, not stale HBM)
    int n_cu = 256; { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, h->cfg.device) == hipSuccess) n_cu = pr.multiProcessorCount; }
    for (size_t c = 0; c < h->classes.size(); c++) {
        const auto& cls = h->classes[c];
        const size_t bytes = (cls.pairs.size() * (size_t)4 * F * (cls.prot.nbits / 8) + 15) & ~(size_t)15;     // (whole 16-byte pieces: the regions are 256-byte aligned and padded)
        if (bytes) launch_copy_f4(cls.out.p, h->drain_stage.as<uint8_t>() + off[c], bytes / 16, 4 * n_cu, h->stream);
    }
    HIPCHK(h, hipEventRecord(h->ev_drain_staged, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->drain_stream, h->ev_drain_staged, 0));
    {
        // (pieces of a few MB: a copy engine works its queues off packet by packet, and one 100 MB packet would hold up the small
        // transfers of the next dabphy_process -- descriptors, FIBs -- that share the engine)
        constexpr size_t PIECE = (size_t)4 << 20;
        for (size_t at = 0; at < total; at += PIECE)
            HIPCHK(h, hipMemcpyAsync(buf + at, h->drain_stage.as<uint8_t>() + at, std::min(PIECE, total - at), hipMemcpyDeviceToHost, h->drain_stream));
    }
    HIPCHK(h, hipEventRecord(h->ev_drain_done, h->drain_stream));
    h->drain_pending = true;
    uint32_t k = 0;
    for (uint32_t b = 0; b < B; b++)
        for (size_t i = 0; i < h->where[b].size(); i++, k++) {
            const dabphy_handle::PairRef w = h->where[b][i];
            const auto& cls = h->classes[w.cls];
            dabphy_msc_desc& d = desc[k];
            d.ensemble = b; d.subch_index = (uint32_t)i; d.row_bytes = (uint32_t)(cls.prot.nbits / 8); d.subch_id = (uint32_t)cls.subch_id[w.pair];
            d.offset = off[w.cls] + (uint64_t)w.pair * 4 * F * d.row_bytes;
            msc_rows_info(h, b, w, &d.first_valid, &d.n_rows);
        }
    *n_desc = n;
    return DABPHY_OK;
}

int dabphy_msc_drain_wait(dabphy_handle* h)
{
    DeviceBind dev_(h);
    if (!h) return DABPHY_ERR_INVALID;
    return drain_wait(h);
}

int dabphy_get_msc_batch(dabphy_handle* h, dabphy_msc_desc* desc, uint32_t desc_capacity, uint32_t* n_desc, uint8_t* buf, size_t buf_capacity)
{
    int r = dabphy_msc_drain_begin(h, desc, desc_capacity, n_desc, buf, buf_capacity);
    return r ? r : dabphy_msc_drain_wait(h);
}

int dabphy_get_msc(dabphy_handle* h, uint32_t subch_index, uint8_t* out, size_t out_capacity, int32_t* first_valid, int32_t* n_rows)
{
    DeviceBind dev_(h);
    if (!h || !out || !h->last_frames) return DABPHY_ERR_INVALID;
    const uint32_t B = h->cfg.n_ensembles, F = h->last_frames;
    size_t bytes = 0;
    for (uint32_t b = 0; b < B; b++) {
        if (subch_index >= h->where[b].size()) { h->err = "ensemble " + std::to_string(b) + " has no sub-channel " + std::to_string(subch_index); return DABPHY_ERR_INVALID; }
        const size_t mine = h->classes[h->where[b][subch_index].cls].prot.nbits / 8;
        if (bytes && bytes != mine) { h->err = "the ensembles' sub-channels at this position differ in bit rate: use dabphy_get_msc_ensemble"; return DABPHY_ERR_INVALID; }
        bytes = mine;
    }
    if (out_capacity < (size_t)B * 4 * F * bytes) { h->err = "dabphy_get_msc: output buffer too small"; return DABPHY_ERR_INVALID; }
    // Runs of ensembles whose pairs lie in one class at a constant distance (the same list everywhere: pair = b * M + m) leave in ONE
    // strided copy: [run][4F * bytes] rows out of the class output [pair][4F][bytes].  A batch of different multiplexes falls apart into
    // shorter runs, down to one copy per ensemble.
    const size_t row = (size_t)4 * F * bytes;
    for (uint32_t b0 = 0; b0 < B;) {
        const dabphy_handle::PairRef w0 = h->where[b0][subch_index];
        uint32_t b1 = b0 + 1; int step = 0;
        if (b1 < B && h->where[b1][subch_index].cls == w0.cls && h->where[b1][subch_index].pair > w0.pair) {
            step = h->where[b1][subch_index].pair - w0.pair;
            for (b1++; b1 < B && h->where[b1][subch_index].cls == w0.cls && h->where[b1][subch_index].pair == w0.pair + (int)(b1 - b0) * step; b1++) {}
        }
        const auto& cls = h->classes[w0.cls];
        const uint8_t* src = cls.out.as<uint8_t>() + (size_t)w0.pair * row;
        if (b1 - b0 > 1) HIPCHK(h, hipMemcpy2DAsync(out + (size_t)b0 * row, row, src, (size_t)step * row, row, b1 - b0, hipMemcpyDeviceToHost, h->stream));
        else HIPCHK(h, hipMemcpyAsync(out + (size_t)b0 * row, src, row, hipMemcpyDeviceToHost, h->stream));
        for (uint32_t b = b0; b < b1; b++) msc_rows_info(h, b, h->where[b][subch_index], first_valid ? first_valid + b : nullptr, n_rows ? n_rows + b : nullptr);
        b0 = b1;
    }
    return sync(h);
}

int dabphy_get_msc_ensemble(dabphy_handle* h, uint32_t ensemble, uint32_t subch_index, uint8_t* out, size_t out_capacity, int32_t* first_valid, int32_t* n_rows)
{
    DeviceBind dev_(h);
    if (!h || !out || !h->last_frames || ensemble >= h->cfg.n_ensembles || subch_index >= h->where[ensemble].size()) return DABPHY_ERR_INVALID;
    const dabphy_handle::PairRef w = h->where[ensemble][subch_index];
    if (out_capacity < (size_t)4 * h->last_frames * (h->classes[w.cls].prot.nbits / 8)) { h->err = "dabphy_get_msc_ensemble: output buffer too small"; return DABPHY_ERR_INVALID; }
    int r = msc_rows_of(h, ensemble, w, out, first_valid, n_rows);
    return r ? r : sync(h);
}

int dabphy_get_impulse_response(dabphy_handle* h, float* out)
{
    DeviceBind dev_(h);
    if (!h || !out || !h->last_frames || !h->cfg.want_impulse_response) return DABPHY_ERR_INVALID;
    HIPCHK(h, hipMemcpyAsync(out, h->cur_cir, (size_t)h->cfg.n_ensembles * h->last_frames * T_U * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    return sync(h);
}

int dabphy_get_null_symbols(dabphy_handle* h, float* out)
{
    DeviceBind dev_(h);
    if (!h || !out || !h->last_frames || !h->last_desc || !h->s_iq) return DABPHY_ERR_INVALID;
    const uint32_t B = h->cfg.n_ensembles, F = h->last_frames;
    int r;
    if ((r = ensure(h, h->s_null, (size_t)B * F * T_NULL * sizeof(cf32)))) return r;
    NullArgs a{};
    a.tab = h->tab; a.iq = h->s_iq; a.iq_stride = h->s_stride; a.ring = (int64_t)h->s_ring; a.desc = h->last_desc; a.n_frames = (int)F;
    a.out = h->s_null.as<cf32>();
    launch_null_symbols(a, (int)B, h->stream);
    HIPCHK(h, hipMemcpyAsync(out, h->s_null.p, (size_t)B * F * T_NULL * sizeof(cf32), hipMemcpyDeviceToHost, h->stream));
    return sync(h);
}

int dabphy_get_constellation(dabphy_handle* h, float* out)
{
    DeviceBind dev_(h);
    if (!h || !out || !h->last_frames || !h->cfg.want_constellation) return DABPHY_ERR_INVALID;
    HIPCHK(h, hipMemcpyAsync(out, h->s_con.p, (size_t)h->cfg.n_ensembles * h->last_frames * 1200 * sizeof(cf32), hipMemcpyDeviceToHost, h->stream));
    return sync(h);
}

int dabphy_get_soft_bits(dabphy_handle* h, uint32_t ensemble, uint32_t frame, int8_t* out)
{
    DeviceBind dev_(h);
    if (!h || !out || ensemble >= h->cfg.n_ensembles || frame >= h->last_frames) return DABPHY_ERR_INVALID;
    const FrameDesc& d = h->h_desc[(size_t)ensemble * h->last_frames + frame];
    const size_t slot = (size_t)(d.frame_no % h->soft_ring);
    HIPCHK(h, hipMemcpyAsync(out, h->s_soft.as<int8_t>() + (size_t)ensemble * soft_ens_stride(h) + slot * SOFT_PER_FRAME, SOFT_PER_FRAME, hipMemcpyDeviceToHost, h->stream));
    return sync(h);
}

int dabphy_last_decode_plan(dabphy_handle* h, int32_t* shape, int32_t* fused_classes)
{
    if (!h) return DABPHY_ERR_INVALID;
    const auto& P = h->fplan;
    const bool ran = P.valid && P.launched && h->last_frames;
    if (shape) *shape = !ran ? 0 : P.use_sp ? (P.sp_two ? 2 : 3) : 1;         // (numbered like dabphy_config.decode_shape)
    if (fused_classes) *fused_classes = ran ? (int32_t)P.class_idx.size() : 0;
    return DABPHY_OK;
}

int dabphy_set_profiling(dabphy_handle* h, int32_t on)
{
    DeviceBind dev_(h);
    if (!h) return DABPHY_ERR_INVALID;
    h->profiling = on != 0;
    return DABPHY_OK;
}

int dabphy_get_stage_times(dabphy_handle* h, float* ms)
{
    DeviceBind dev_(h);
    if (!h || !ms) return DABPHY_ERR_INVALID;
    for (int i = 0; i < dabphy_handle::ST_COUNT; i++) {
        ms[i] = 0.0f;
        if (h->ev_used[i]) { float t = 0; if (hipEventElapsedTime(&t, h->ev_beg[i], h->ev_end[i]) == hipSuccess) ms[i] = t; }
    }
    ms[dabphy_handle::ST_SYNC] = h->chain_ms;    // measured on the sync stream (overlaps the previous batch's decode in pipelined mode)
    return DABPHY_OK;
}

int dabphy_set_track_slevel(dabphy_handle* h, int32_t on)
{
    DeviceBind dev_(h);
    if (!h) return DABPHY_ERR_INVALID;
    h->track_slevel = on != 0;
    return DABPHY_OK;
}


// RadioReceiverOptions::decodeTII (radio-receiver-options.h:75, consulted once per frame at ofdm-processor.cpp:376-386,464)
int dabphy_set_tii(dabphy_handle* h, int32_t on)
{
    DeviceBind dev_(h);
    if (!h) return DABPHY_ERR_INVALID;
    if (on && !h->tii_rot.p) {
        const TiiTables& T = tii_tables();
        const uint32_t B = h->cfg.n_ensembles;
        int r;
        if ((r = ensure(h, h->tii_rot, T.rot.size() * sizeof(cf32)))) return r;
        if ((r = ensure(h, h->tii_rank, sizeof T.rank))) return r;
        if ((r = ensure(h, h->tii_pat, sizeof T.pattern))) return r;
        if ((r = ensure(h, h->tii_state, (size_t)B * TII_SLOTS * sizeof(TiiSlot)))) return r;
        if ((r = ensure(h, h->tii_ovf, (size_t)B * sizeof(int32_t)))) return r;
        HIPCHK(h, hipMemcpyAsync(h->tii_rot.p, T.rot.data(), T.rot.size() * sizeof(cf32), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->tii_rank.p, T.rank, sizeof T.rank, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->tii_pat.p, T.pattern, sizeof T.pattern, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemsetAsync(h->tii_state.p, 0, h->tii_state.cap, h->stream));
        HIPCHK(h, hipMemsetAsync(h->tii_ovf.p, 0, h->tii_ovf.cap, h->stream));
        if ((r = sync(h))) return r;
    }
    h->tii_on = on != 0;
    return DABPHY_OK;
}

int dabphy_get_tii(dabphy_handle* h, dabphy_tii_measurement* out, int32_t* n, uint32_t max_per_ensemble)
{
    DeviceBind dev_(h);
    if (!h || !n || (!out && max_per_ensemble) || !h->last_frames) return DABPHY_ERR_INVALID;
    const uint32_t B = h->cfg.n_ensembles;
    if (!h->tii_ran) { for (uint32_t b = 0; b < B; b++) n[b] = 0; return DABPHY_OK; }
    static_assert(sizeof(dabphy_tii_measurement) == sizeof(TiiEvent), "dabphy_tii_measurement layout");
    std::vector<TiiEvent> ev((size_t)B * h->tii_max_events);
    HIPCHK(h, hipMemcpyAsync(n, h->tii_nev.p, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(ev.data(), h->tii_events.p, ev.size() * sizeof(TiiEvent), hipMemcpyDeviceToHost, h->stream));
    int r = sync(h); if (r) return r;
    for (uint32_t b = 0; b < B; b++) {
        const uint32_t k = std::min<uint32_t>((uint32_t)n[b], std::min(max_per_ensemble, h->tii_max_events));
        if (k) memcpy(out + (size_t)b * max_per_ensemble, ev.data() + (size_t)b * h->tii_max_events, k * sizeof(TiiEvent));
    }
    return DABPHY_OK;
}

} // extern "C"
