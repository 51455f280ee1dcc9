// welle.io_amd/csrc/dabphy_stream.hip -- streaming receiver: sample rings, the synchroniser's launches (serial chain and wide pass), reset.
// (split from dabphy_api.hip in round 3; dabphy_internal.h has the map of the translation units)
#include "dabphy_internal.h"

extern "C" {

// =================================================================================== streaming receiver

// ---- the synchroniser's launches.  It runs on its own stream; in pipelined mode (cfg.pipeline_sync) the frames of the NEXT batch are
// synchronised while this batch is decoded on the main stream (they need only the samples and the synchroniser's own state).
// Two forms:
//   serial chain   per frame: k_sync_find (PRS window search; acquisition first for an ensemble that is not synchronised -- start of a
//                  stream, or after a failed window search in whatever slot of a batch, as the reference falls back to notSynced,
//                  ofdm-processor.cpp:347-350) then k_sync_finish (cyclic-prefix sums -> correctors -> state).  2 F dependent launches.
//   wide pass      every frame of the batch at once, each from the state a receiver IN LOCK would be in (k_sync.hip: sync_predict), then
//                  k_sync_validate accepts the frames whose assumption held and says where the serial chain has to take over.  The
//                  verdict is read by the host the next time the batch is needed (resolve_chain): in pipelined mode that is a whole
//                  decode later, so nothing waits for it.
SyncArgs sync_args(dabphy_handle* h, int sel, uint32_t F, uint64_t n_valid)
{
    SyncArgs sa{};
    sa.tab = h->tab; sa.iq = h->s_iq; sa.iq_stride = h->s_stride; sa.ring = (int64_t)h->s_ring; sa.n_valid = (int64_t)n_valid;
    sa.loop = h->s_loop; sa.state = h->d_state; sa.dec = h->d_dec; sa.desc = h->s_desc2[sel].as<FrameDesc>(); sa.n_ens = (int)h->cfg.n_ensembles; sa.n_frames = (int)F;
    sa.fft_placement = h->cfg.fft_placement; sa.disable_coarse = h->cfg.disable_coarse; sa.freqsync = h->cfg.freqsync_method;
    sa.cir = h->cfg.want_impulse_response ? h->s_cir2[sel].as<float>() : nullptr;
    sa.hist = h->s_hist.as<FrameDesc>(); sa.hist_cap = HIST_CAP;
    // |re| + |im| of a sample with |re|, |im| <= 1 after the oscillator (|o| = 1 to 1e-7) is at most sqrt(2) * sqrt(2) = 2; a little room for rounding
    sa.level_max = h->s_bounded ? 2.125f : 3.0e38f;
    return sa;
}
void launch_serial_chain(dabphy_handle* h, SyncArgs sa)
{
    for (int f = 0; f < sa.n_frames; f++) {
        sa.frame = f;
        launch_sync_find(sa, h->sync_stream);
        launch_sync_finish(sa, h->sync_stream);
        if (h->track_slevel) launch_slevel_catchup(sa, h->sync_stream);
    }
}
constexpr int N_DESC_ = dabphy_handle::N_DESC;
int queue_chain(dabphy_handle* h, int sel, uint32_t F)
{
    SyncArgs sa = sync_args(h, sel, F, h->s_valid);
    h->chain_valid[sel] = h->s_valid; h->chain_frames[sel] = F;
    if (replay_armed(h, F) && h->snap_state[sel].p)              // (one frame per call on the serial schedule is exact by construction: nothing to put back)
    {
        HIPCHK(h, hipMemcpyAsync(h->snap_state[sel].p, h->d_state, sizeof(RxState) * h->cfg.n_ensembles, hipMemcpyDeviceToDevice, h->sync_stream));
        // ... and the history ring its hist_head / hist_count index: an acquisition inside the batch restarts the ring at entry 0, over the
        // window searches the second pass has to replay through the sLevel recurrence when it loses lock at the same frame
        if (h->snap_hist[sel].p && h->s_hist.p)
            HIPCHK(h, hipMemcpyAsync(h->snap_hist[sel].p, h->s_hist.p, (size_t)h->cfg.n_ensembles * HIST_CAP * sizeof(FrameDesc), hipMemcpyDeviceToDevice, h->sync_stream));
    }
    { hipError_t e = hipEventRecord(h->ev_chain_beg[sel], h->sync_stream); (void)e; }
    // one frame per call (the real-time facade) gains nothing from the wide pass; two batches ahead its verdict would come too late
    if (h->wide_sync && F >= 2 && h->cfg.pipeline_sync != 3 && !h->track_slevel) {
        // the verdict lands in page-locked host memory straight from the last judge kernel (no copy that could queue behind a bulk
        // transfer on the DMA engines); the host clears it here: the buffer's previous pass has been resolved
        h->h_any_redo[sel] = 0; h->h_any_redo[N_DESC_ + sel] = 0;
        sa.redo_out = h->s_redo[sel].as<int32_t>(); sa.any_redo = h->d_any_redo + sel; sa.any_chain = h->d_any_redo + N_DESC_ + sel; sa.skip_wide = h->drift_seen ? 1 : 0;
        if (!h->ev_wide_front) HIPCHK(h, hipEventCreateWithFlags(&h->ev_wide_front, hipEventDisableTiming));
        launch_sync_wide(sa, h->sync_stream, h->ev_wide_front); h->wide_front_recorded = true;
        HIPCHK(h, hipEventRecord(h->ev_wide_done[sel], h->sync_stream));
        h->wide_pending[sel] = true; h->n_wide_passes++;
    } else {
        launch_serial_chain(h, sa);
    }
    { hipError_t e = hipEventRecord(h->ev_chain_end[sel], h->sync_stream); (void)e; }
    return DABPHY_OK;
}
// reads the wide pass's verdict for descriptor buffer `sel` and queues the serial chain for what it did not settle
int resolve_chain(dabphy_handle* h, int sel)
{
    if (!h->wide_pending[sel]) return DABPHY_OK;
    HIPCHK(h, hipEventSynchronize(h->ev_wide_done[sel]));
    h->wide_pending[sel] = false;
    h->drift_seen = h->h_any_redo[N_DESC_ + sel] != 0;         // ensembles whose window moves: the next pass starts in the find chain for everybody
    if (h->h_any_redo[sel]) {
        SyncArgs sa = sync_args(h, sel, h->chain_frames[sel], h->chain_valid[sel]);
        sa.redo_from = h->s_redo[sel].as<int32_t>();
        launch_serial_chain(h, sa);
        { hipError_t e = hipEventRecord(h->ev_chain_end[sel], h->sync_stream); (void)e; }
        h->n_wide_fallbacks++;
    }
    return DABPHY_OK;
}
int resolve_all_chains(dabphy_handle* h)
{
    for (int i = 0; i < dabphy_handle::N_DESC; i++) { int r = resolve_chain(h, (h->desc_sel + i) % dabphy_handle::N_DESC); if (r) return r; }
    return DABPHY_OK;
}

// OFDMProcessor::restart (ofdm-processor.cpp:115-132) + the start of run(): correctors, phase and sync state zero, sLevel primed over
// the next T_F/2 samples (:252-255).  decoder_too (dabphy_reset: a freshly bound stream) also rewinds the stream to sample 0 and
// clears the frame counter; without it (setReceiverOptions on a running receiver) the stream goes on where the DECODED frames end:
// frames that pipelined mode had synchronised ahead are handed back, so the time de-interleavers see every CIF exactly once.
int reset_synchroniser(dabphy_handle* h, bool decoder_too)
{
    if (h->s_desc2[0].p) { int r = resolve_all_chains(h); if (r) return r; }
    if (h->sync_stream) HIPCHK(h, hipStreamSynchronize(h->sync_stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const uint32_t B = h->cfg.n_ensembles;
    std::vector<RxState> init(B);
    std::vector<FrameDesc> ahead;
    if (!decoder_too) {
        HIPCHK(h, hipMemcpy(init.data(), h->d_state, init.size() * sizeof(RxState), hipMemcpyDeviceToHost));
        if (h->presynced && h->ahead > 0 && h->s_desc2[h->desc_sel].p) {          // the earliest batch synchronised ahead starts where the decoded frames end
            ahead.resize((size_t)B * h->presynced);
            HIPCHK(h, hipMemcpy(ahead.data(), h->s_desc2[h->desc_sel].p, ahead.size() * sizeof(FrameDesc), hipMemcpyDeviceToHost));
        }
    }
    for (uint32_t b = 0; b < B; b++) {
        RxState& s = init[b];
        int64_t frame_no = decoder_too ? 0 : s.frame_no, pos = decoder_too ? 0 : s.pos;
        if (!ahead.empty()) { frame_no = ahead[(size_t)b * h->presynced].frame_no; pos = ahead[(size_t)b * h->presynced].pos; }
        // counters that outlive OFDMProcessor::restart (`attempts` is a member that only the end of a scan clears, ofdm-processor.h:111,
        // ofdm-processor.cpp:258-262,354) and this library's own statistics
        const RxState keep = s;
        memset(&s, 0, sizeof s);
        s.acq_phase = 0; s.acq_left = T_F / 2; s.first_lock_attempts = -1; s.frame_no = frame_no; s.pos = pos;
        if (!decoder_too) {
            s.attempts = keep.attempts; s.first_lock_attempts = keep.first_lock_attempts; s.lost = keep.lost;
            s.n_exact_sums = keep.n_exact_sums; s.n_relock_inexact = keep.n_relock_inexact; s.n_wide_frames = keep.n_wide_frames; s.n_chain_frames = keep.n_chain_frames;
        }
    }
    h->presynced = 0; h->ahead = 0;
    HIPCHK(h, hipMemcpyAsync(h->d_state, init.data(), init.size() * sizeof(RxState), hipMemcpyHostToDevice, h->stream));
    return sync(h);
}

int dabphy_reset(dabphy_handle* h)
{
    DeviceBind dev_(h);
    if (!h) return DABPHY_ERR_INVALID;
    int r = reset_synchroniser(h, true); if (r) return r;
    h->desc_sel = 0; h->n_wide_passes = h->n_wide_fallbacks = 0; h->n_replayed_batches = 0;
    HIPCHK(h, hipMemsetAsync(h->d_dec, 0, sizeof(DecState) * h->cfg.n_ensembles, h->stream));
    h->last_frames = 0; h->last_desc = nullptr;
    if (h->rs_stream) HIPCHK(h, hipStreamSynchronize(h->rs_stream));
    h->sf_def_pending = h->sf_def_unfetched = h->sf_def_inflight = false;      // (a deferred filter pass of the stream that ends here is dropped with it)
    for (auto& c : h->classes) if (c.sf_state.p) HIPCHK(h, hipMemsetAsync(c.sf_state.p, 0, c.sf_state.cap, h->stream));   // decoders restart too (RadioReceiver::restart_decoder)
    // ... and the frame count starts over: every selected sub-channel's time de-interleaver fills again from the first CIF decoded
    for (auto& c : h->classes) { for (MscPair& p : c.pairs) p.cif0 = -1; int r2 = upload_pairs(h, c); if (r2) return r2; }
    if (h->tii_state.p) HIPCHK(h, hipMemsetAsync(h->tii_state.p, 0, h->tii_state.cap, h->stream));      // a new OFDMProcessor owns a new TIIDecoder
    h->tii_ran = false;
    return sync(h);
}

int dabphy_stream_bind_device(dabphy_handle* h, const void* d_iq, uint64_t ring_samples, uint64_t stride_samples,
                              uint64_t n_valid, int32_t loop)
{
    DeviceBind dev_(h);
    if (!h || !d_iq || ring_samples < (uint64_t)T_F || stride_samples < ring_samples) return DABPHY_ERR_INVALID;
    h->s_iq = reinterpret_cast<const cf32*>(d_iq); h->s_ring = ring_samples; h->s_stride = stride_samples;
    h->s_valid = n_valid; h->s_enqueued = 0; h->commit_slot = -1; h->s_loop = loop;
    h->s_bounded = false;                                         // the caller's cf32 samples: no bound known
    return dabphy_reset(h);
}

int dabphy_stream_upload(dabphy_handle* h, const float* iq, uint64_t n_samples, int32_t loop)
{
    DeviceBind dev_(h);
    if (!h || !iq || n_samples < (uint64_t)T_F) return DABPHY_ERR_INVALID;
    const size_t bytes = (size_t)h->cfg.n_ensembles * n_samples * sizeof(cf32);
    int r;
    if ((r = ensure(h, h->s_iq_own, bytes))) return r;
    HIPCHK(h, hipMemcpyAsync(h->s_iq_own.p, iq, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return dabphy_stream_bind_device(h, h->s_iq_own.p, n_samples, n_samples, n_samples, loop);
}

int dabphy_stream_open(dabphy_handle* h, uint64_t ring_samples)
{
    DeviceBind dev_(h);
    if (!h || ring_samples < 4 * (uint64_t)T_F) return DABPHY_ERR_INVALID;
    const size_t bytes = (size_t)h->cfg.n_ensembles * ring_samples * sizeof(cf32);
    int r;
    if ((r = ensure(h, h->s_iq_own, bytes))) return r;
    HIPCHK(h, hipMemsetAsync(h->s_iq_own.p, 0, bytes, h->stream));
    r = dabphy_stream_bind_device(h, h->s_iq_own.p, ring_samples, ring_samples, 0, 0);
    h->s_bounded = true;                                          // an empty ring of zeros; a cf32 write (dabphy_stream_write) lifts the bound
    return r;
}

int dabphy_stream_write(dabphy_handle* h, const float* iq, uint64_t n_samples)
{
    DeviceBind dev_(h);
    if (!h || !iq || !h->s_iq_own.p || h->s_iq != h->s_iq_own.as<cf32>() || n_samples == 0 || n_samples > h->s_ring) return DABPHY_ERR_INVALID;
    h->s_bounded = false;                                         // cf32 from the caller: any magnitude
    // the chain that may be running ahead must not race with the copy
    HIPCHK(h, hipStreamSynchronize(h->sync_stream));
    const uint64_t w = h->s_valid % h->s_ring;
    const uint64_t first = std::min<uint64_t>(n_samples, h->s_ring - w);
    for (uint32_t b = 0; b < h->cfg.n_ensembles; b++) {
        cf32* dst = h->s_iq_own.as<cf32>() + (size_t)b * h->s_stride;
        const cf32* src = reinterpret_cast<const cf32*>(iq) + (size_t)b * n_samples;
        HIPCHK(h, hipMemcpyAsync(dst + w, src, first * sizeof(cf32), hipMemcpyHostToDevice, h->stream));
        if (first < n_samples) HIPCHK(h, hipMemcpyAsync(dst, src + first, (n_samples - first) * sizeof(cf32), hipMemcpyHostToDevice, h->stream));
    }
    h->s_valid += n_samples;
    return sync(h);
}

int dabphy_stream_write_raw(dabphy_handle* h, const void* data, uint64_t n_samples, int32_t format)
{
    DeviceBind dev_(h);
    if (format == DABPHY_FMT_CF32) return dabphy_stream_write(h, reinterpret_cast<const float*>(data), n_samples);
    if (!h || !data || !h->s_iq_own.p || h->s_iq != h->s_iq_own.as<cf32>() || n_samples == 0 || n_samples > h->s_ring ||
        format < DABPHY_FMT_U8 || format > DABPHY_FMT_S16BE) return DABPHY_ERR_INVALID;
    const size_t bps = (format == DABPHY_FMT_U8 || format == DABPHY_FMT_S8) ? 2 : 4;
    const uint32_t B = h->cfg.n_ensembles;
    int r;
    if ((r = ensure(h, h->s_raw, (size_t)B * n_samples * bps))) return r;
    HIPCHK(h, hipStreamSynchronize(h->sync_stream));           // the chain that may be running ahead must not race with the write
    HIPCHK(h, hipMemcpyAsync(h->s_raw.p, data, (size_t)B * n_samples * bps, hipMemcpyHostToDevice, h->stream));
    IngestArgs a{};
    a.raw = h->s_raw.as<uint8_t>(); a.raw_stride = n_samples * bps; a.iq = h->s_iq_own.as<cf32>(); a.iq_stride = h->s_stride;
    a.ring = h->s_ring; a.w = h->s_valid % h->s_ring; a.n = n_samples; a.format = format;
    launch_ingest(a, (int)B, h->stream);
    h->s_valid += n_samples;
    return sync(h);
}

int dabphy_stream_read(dabphy_handle* h, uint32_t ensemble, uint64_t pos, uint64_t n_samples, float* out)
{
    DeviceBind dev_(h);
    if (!h || !out || !h->s_iq || ensemble >= h->cfg.n_ensembles || n_samples == 0 || n_samples > h->s_ring) return DABPHY_ERR_INVALID;
    HIPCHK(h, hipStreamSynchronize(h->copy_stream));
    const cf32* src = h->s_iq + (size_t)ensemble * h->s_stride;
    const uint64_t w = pos % h->s_ring, first = std::min<uint64_t>(n_samples, h->s_ring - w);
    HIPCHK(h, hipMemcpyAsync(out, src + w, first * sizeof(cf32), hipMemcpyDeviceToHost, h->stream));
    if (first < n_samples) HIPCHK(h, hipMemcpyAsync(out + 2 * first, src, (n_samples - first) * sizeof(cf32), hipMemcpyDeviceToHost, h->stream));
    return sync(h);
}

int dabphy_stream_write_raw_async(dabphy_handle* h, const void* data, uint64_t n_samples, int32_t format)
{
    DeviceBind dev_(h);
    if (!h || !data || !h->s_iq_own.p || h->s_iq != h->s_iq_own.as<cf32>() || n_samples == 0 || n_samples > h->s_ring ||
        format < DABPHY_FMT_U8 || format > DABPHY_FMT_S16BE) return DABPHY_ERR_INVALID;
    const size_t bps = (format == DABPHY_FMT_U8 || format == DABPHY_FMT_S8) ? 2 : 4;
    const uint32_t B = h->cfg.n_ensembles;
    if (h->s_enqueued < h->s_valid) h->s_enqueued = h->s_valid;            // synchronous writes in between
    const int slot = h->raw_sel; h->raw_sel ^= 1;
    // the staging slot (and with it the host buffer of the call before last) is free once its previous conversion has run
    HIPCHK(h, hipEventSynchronize(h->ev_ingest[slot]));
    int r;
    if ((r = ensure(h, h->s_raw2[slot], (size_t)B * n_samples * bps))) return r;
    HIPCHK(h, hipMemcpyAsync(h->s_raw2[slot].p, data, (size_t)B * n_samples * bps, hipMemcpyHostToDevice, h->copy_stream));
    IngestArgs a{};
    a.raw = h->s_raw2[slot].as<uint8_t>(); a.raw_stride = n_samples * bps; a.iq = h->s_iq_own.as<cf32>(); a.iq_stride = h->s_stride;
    a.ring = h->s_ring; a.w = h->s_enqueued % h->s_ring; a.n = n_samples; a.format = format;
    launch_ingest(a, (int)B, h->copy_stream);
    HIPCHK(h, hipEventRecord(h->ev_ingest[slot], h->copy_stream));
    h->s_enqueued += n_samples;
    return DABPHY_OK;
}

int dabphy_stream_commit(dabphy_handle* h)
{
    DeviceBind dev_(h);
    if (!h) return DABPHY_ERR_INVALID;
    if (h->s_enqueued > h->s_valid) { h->s_valid = h->s_enqueued; h->commit_slot = h->raw_sel ^ 1; }
    return DABPHY_OK;
}

int dabphy_host_alloc(size_t bytes, void** out)
{
    if (!out || !bytes) return DABPHY_ERR_INVALID;
    return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? DABPHY_OK : DABPHY_ERR_NOMEM;
}

void dabphy_host_free(void* p) { if (p) { hipError_t e = hipHostFree(p); (void)e; } }

uint64_t dabphy_stream_consumed(dabphy_handle* h)
{
    DeviceBind dev_(h);
    if (!h) return 0;
    std::vector<RxState> st(h->cfg.n_ensembles);
    if (hipStreamSynchronize(h->sync_stream) != hipSuccess) return 0;
    if (hipMemcpy(st.data(), h->d_state, st.size() * sizeof(RxState), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    uint64_t m = ~0ull;
    for (auto& s : st) m = std::min<uint64_t>(m, (uint64_t)s.pos);
    return m;
}

} // extern "C"
