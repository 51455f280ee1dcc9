// welle.io_amd/csrc/dabphy_process.hip -- dabphy_process: one batch through the synchroniser, the demod kernel, the FIC and MSC decoders; pipelined schedules; exact batch mode.
// (split from dabphy_api.hip in round 3; dabphy_internal.h has the map of the translation units)
#include "dabphy_internal.h"

extern "C" {

// One batch: acquisition where needed, n_frames frame steps of the synchroniser, then the fully parallel stages.
// DABPHY_DEBUG_TIMING=1: host-side time line of dabphy_process (microseconds since entry, averaged, printed by dabphy_destroy)
struct HostTimeline { double acc[6] = {0, 0, 0, 0, 0, 0}; long n = 0; };
static HostTimeline g_tl; static int g_tl_on = -1;
static inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int dabphy_process(dabphy_handle* h, uint32_t n_frames)
{
    DeviceBind dev_(h);
    if (!h || n_frames == 0 || n_frames > h->cfg.max_frames) return DABPHY_ERR_INVALID;
    if (g_tl_on < 0) g_tl_on = debug_env("DABPHY_DEBUG_TIMING") ? 1 : 0;
    const double tl0 = g_tl_on ? now_us() : 0.0; double tl[6] = {0, 0, 0, 0, 0, 0};
    auto tick = [&](int i) { if (g_tl_on) tl[i] = now_us() - tl0; };
    if (!h->s_iq) { h->err = "no sample stream bound"; return DABPHY_ERR_STATE; }
    const uint32_t B = h->cfg.n_ensembles, F = n_frames;
    const int ring_frames = (int)h->cfg.max_frames + 5;
    int r;
    if (h->subch_dirty && (h->sf_def_pending || h->sf_def_inflight) && (r = flush_deferred_superframes(h))) return r;      // (the deferred filter pass of the last batch belongs to the classes that are about to be rebuilt)
    if ((r = apply_subchannels(h))) return r;                // per-ensemble sub-channel changes since the last batch (dabphy_set_subchannels_ensemble)
    for (int k = 0; k < dabphy_handle::N_DESC; k++) {
        if ((r = ensure(h, h->s_desc2[k], (size_t)B * h->cfg.max_frames * sizeof(FrameDesc)))) return r;
        if ((r = ensure(h, h->s_redo[k], (size_t)B * sizeof(int32_t)))) return r;
        if (h->exact_batch && (r = ensure(h, h->snap_state[k], (size_t)B * sizeof(RxState)))) return r;
        if (h->exact_batch && (r = ensure(h, h->snap_hist[k], (size_t)B * HIST_CAP * sizeof(FrameDesc)))) return r;
        if (h->cfg.want_impulse_response && (r = ensure(h, h->s_cir2[k], (size_t)B * h->cfg.max_frames * T_U * sizeof(float)))) return r;
    }
    const size_t ens_stride = soft_ens_stride(h);
    {   // [B][ring_frames frame slots + one frame of zeros]: the zeros are what the fused decode loads for CIFs that do not exist yet
        // (nothing ever writes them; one sub-channel's worth -- 864 CU x 64 bits -- is the most a row needs)
        const size_t ring_bytes = (size_t)B * ens_stride;
        if (h->s_soft.cap < ring_bytes) {
            if ((r = ensure(h, h->s_soft, ring_bytes))) return r;
            for (uint32_t b = 0; b < B; b++) HIPCHK(h, hipMemsetAsync(h->s_soft.as<int8_t>() + (size_t)b * ens_stride + (size_t)ring_frames * SOFT_PER_FRAME, 0, SOFT_PER_FRAME, h->stream));
        }
    }
    if ((r = ensure(h, h->s_hist, (size_t)B * HIST_CAP * sizeof(FrameDesc)))) return r;
    if ((r = ensure(h, h->s_mag, (size_t)B * F * T_U * sizeof(float)))) return r;
    if ((r = ensure(h, h->s_snr, (size_t)B * F * sizeof(float)))) return r;
    if ((r = ensure(h, h->s_fib, (size_t)B * F * 384))) return r;
    if ((r = ensure(h, h->s_ok, (size_t)B * F * 12))) return r;
    if (h->cfg.want_constellation && (r = ensure(h, h->s_con, (size_t)B * F * 1200 * sizeof(cf32)))) return r;
    // every allocation this call may need happens here, before any kernel is queued or any pipeline state advances: a failed
    // hipMalloc leaves the handle as it was
    VitClass fic_c{};
    {
        fic_c.nbits = 768; fic_c.nsteps = 774; fic_c.n_cw = (int)(B * F * 4); fic_c.n_groups = (fic_c.n_cw + 63) / 64; fic_c.dedisperse = 1; fic_c.g_begin = 0; fic_c.g_end = fic_c.n_groups;
        if ((r = ensure(h, h->s_fib, (size_t)fic_c.n_groups * 64 * 96))) return r;      // the class output holds whole groups of 64 codewords
        if (h->tii_on) {
            if ((r = ensure(h, h->tii_err, (size_t)B * F * TII_MAX_LIKELY * TII_NERR * sizeof(float)))) return r;
            if ((r = ensure(h, h->tii_likely, (size_t)B * F * (1 + TII_MAX_LIKELY) * sizeof(int32_t)))) return r;
            if ((r = ensure(h, h->tii_events, (size_t)B * TII_MAX_LIKELY * h->cfg.max_frames * sizeof(TiiEvent)))) return r;
            if ((r = ensure(h, h->tii_nev, (size_t)B * sizeof(int32_t)))) return r;
        }
        for (auto& cls : h->classes) {
            const size_t n_groups = ((size_t)4 * F * cls.pairs.size() + 63) / 64;
            if ((r = ensure(h, cls.out, n_groups * 64 * (cls.prot.nbits / 8)))) return r;
        }
        if (h->sf_auto && (r = prepare_superframes(h, F))) return r;
        // (the replay of exact batch mode decodes one frame's FIC at a time, state-parallel when 4 B code words are few: its buffers now)
        if (replay_armed(h, F) && sp_single_ok(h, (uint64_t)B * 4, fic_c.nsteps) && (r = sp_single_reserve(h, (uint64_t)B * 4, fic_c.nsteps))) return r;
        // the fused decode of this batch depth: which classes (and whether the FIC) ride in the one launch; its decision scratch
        if ((r = fused_plan(h, F, true))) return r;
        {   // what is left for the two-kernel path: Viterbi scratch of the largest such class; the FIC's own (the replay of exact batch
            // mode decodes one frame's 4 B code words at a time through it even when the batch's FIC is fused)
            size_t ci = 0;
            for (auto& cls : h->classes) {
                const bool fused = std::find(h->fplan.class_idx.begin(), h->fplan.class_idx.end(), (int)ci) != h->fplan.class_idx.end();
                ci++;
                if (fused) continue;
                VitClass c{};
                if ((r = prepare_class(h, c, cls.prot.nbits, (int)(4 * F * cls.pairs.size()), 1))) return r;
            }
            const size_t fic_groups = h->fplan.fic_in ? ((size_t)B * 4 + 63) / 64 : (size_t)fic_c.n_groups;
            if (!h->fplan.fic_in || h->exact_batch) {
                if ((r = ensure(h, h->fsym, fic_groups * fic_c.nsteps * 64 * sizeof(uint32_t)))) return r;
                if ((r = ensure(h, h->fdec, fic_groups * fic_c.nsteps * 64 * sizeof(uint2)))) return r;
            }
        }
        if (h->sf_auto && (r = ensure(h, h->sf_stats, sizeof(int32_t) * 4 * B))) return r;
        if (h->exact_batch) {
            if ((r = ensure(h, h->snap_dec, (size_t)B * sizeof(DecState)))) return r;
            if (h->tii_state.p && (r = ensure(h, h->snap_tii, h->tii_state.cap))) return r;
            if (!h->sf_deferred) for (auto& cls : h->classes) if (cls.sf_state.p && (r = ensure(h, cls.sf_snap, cls.sf_state.cap))) return r;      // (deferred filter: this batch's pass has not run when the batch is decoded again, nothing to put back)
        }
        // (an ensure() above may have moved a buffer the plan names: then it is stale -- plan again, nothing moves the second time)
        if (h->fplan.buf_gen != h->buf_gen && (r = fused_plan(h, F, true))) return r;
    }
    h->soft_ring = ring_frames;

    for (int i = 0; i < dabphy_handle::ST_COUNT; i++) h->ev_used[i] = false;
    auto mark = [&](int stage, bool end, hipStream_t st = nullptr) {
        if (!h->profiling) return;
        hipError_t e = hipEventRecord(end ? h->ev_end[stage] : h->ev_beg[stage], st ? st : h->stream); (void)e;
        h->ev_used[stage] = true;
    };
    if (h->presynced != 0 && h->presynced != F) { h->err = "pipelined mode needs a constant n_frames"; return DABPHY_ERR_STATE; }
    if (h->commit_slot >= 0) {
        // asynchronous ingest: everything committed must have landed before this call's kernels read the ring (the copy stream is
        // in order, the event of the last committed write covers the older ones); uncommitted writes keep flowing meanwhile
        HIPCHK(h, hipStreamWaitEvent(h->sync_stream, h->ev_ingest[h->commit_slot], 0));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_ingest[h->commit_slot], 0));
        h->commit_slot = -1;
    }
    const int ND = dabphy_handle::N_DESC;
    const int depth = h->cfg.pipeline_sync == 3 ? 2 : (h->cfg.pipeline_sync ? 1 : 0);     // batches the synchroniser runs ahead of the decoder
    const int cur = h->desc_sel;
    if (h->ahead == 0) {
        // the previous batch's decoder results (FIC ratio) must be final before the chain consults them
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if ((r = queue_chain(h, cur, F))) return r;
        h->ahead = 1;
    }
    tick(0);
    if ((r = resolve_chain(h, cur))) return r;
    tick(1);
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_chain_end[cur], 0));      // this batch's chain only: later ones may still be running
    // Pipelined modes: the chains of the NEXT batch(es) (40 launches each) are handed to the driver after this batch's decode kernels, so
    // that the main stream never waits for the host, and start on the device
    //   pipeline_sync = 1: when this batch's demod kernel has finished (event gate).  The FFT stage then runs at its own speed and the
    //                      chain shares the device with the Viterbi / RS kernels;
    //   pipeline_sync = 2: at once.  Chain and demod kernel share the device: the FFT stage is slower, the chain done earlier;
    //   pipeline_sync = 3: gated like 1, but TWO batches ahead: the chain of batch k + 2 is queued while batch k is decoded, so the one
    //                      placement stall it meets per step is off the decoder's critical path.
    // DESIGN.md section 4.3 has the numbers.
    h->presynced = depth ? F : 0;
    FrameDesc* const d_desc = h->s_desc2[cur].as<FrameDesc>();
    h->last_desc = d_desc;
    h->cur_cir = h->cfg.want_impulse_response ? h->s_cir2[cur].as<float>() : nullptr;

    // The decode of the batch whose descriptors are in d_desc.  `replay` = the second pass of exact batch mode (see below).
    auto decode = [&](const bool replay) -> int {
    DemodArgs da{};
    da.tab = h->tab; da.iq = h->s_iq; da.iq_stride = h->s_stride; da.ring = (int64_t)h->s_ring;
    da.desc = d_desc; da.n_frames = (int)F; da.chunk_len = h->cfg.demod_chunk; da.mix = 1;
    da.soft = h->s_soft.as<int8_t>(); da.soft_ring = ring_frames; da.soft_ens_stride = ens_stride;
    da.con = h->cfg.want_constellation ? h->s_con.as<cf32>() : nullptr; da.prs_mag = h->s_mag.as<float>();
    da.osc_stats = h->d_osc_stats;
    if (replay) {
        // Exact batch mode, second pass: the batch again, frame by frame, with the reference's own feedback -- the window search of
        // frame f consults the FIC ratio as it stands after frame f - 1 (ofdm-processor.cpp:397), which takes that frame's FIC: chain
        // step, the first chunk(s) of the frame's symbols (PRS + the three FIC symbols), FIC decode of the class, ratio of frame f.  Everything else
        // of the batch follows below as in the first pass (the demod kernel writes the same soft bits again where nothing changed).
        SyncArgs sa = sync_args(h, cur, F, h->chain_valid[cur]);
        // the FIC of ONE frame per step: a class of 4 B code words (frame_sel), decoded into the head of the FIB buffer -- the
        // full-batch FIC pass below writes every FIB again
        VitClass c = fic_c;
        c.n_cw = (int)(B * 4); c.n_groups = (c.n_cw + 63) / 64; c.g_begin = 0; c.g_end = c.n_groups;
        c.sym = h->fsym.as<uint32_t>(); c.dec = h->fdec.as<uint2>(); c.out = h->s_fib.as<uint8_t>();
        FicGatherArgs g{}; g.soft = da.soft; g.soft_ring = ring_frames; g.frame_stride = SOFT_PER_FRAME; g.soft_ens_stride = ens_stride; g.desc = d_desc;
        g.n_ens = (int)B; g.n_frames = (int)F; g.map = h->d_fic_map; g.c = c;
        VitArgs v{}; v.c = c; v.prbs_words = h->d_prbs_words;
        CrcArgs k{}; k.fib = c.out; k.ok = h->s_ok.as<uint8_t>(); k.state = h->d_dec; k.desc = d_desc; k.n_ens = (int)B; k.n_frames = (int)F; k.disable_coarse = h->cfg.disable_coarse;
        // ... state-parallel when the batch of 4 B code words is small enough (one wavefront per code word: a replayed frame then costs a
        // twentieth of a 774-step lane-per-code-word launch); the class is the same for every frame, only the frame selector moves
        const bool fic_sp = sp_single_ok(h, (uint64_t)B * 4, c.nsteps);
        FusedArgs spa{};
        if (fic_sp) {
            FusedClass fc{}; fc.map = h->d_fic_map; fc.out = c.out; fc.nsteps = c.nsteps; fc.nbits = 768; fc.n_cw = c.n_cw; fc.n_pairs = 1; fc.kind = 1; fc.dedisperse = 1;
            spa.soft = da.soft; spa.ens_stride = ens_stride; spa.soft_ring = ring_frames; spa.n_ens = (int)B; spa.n_frames = (int)F; spa.desc = d_desc;
            if ((r = sp_single_prepare(h, fc, spa, h->stream))) return r;
        }
        for (uint32_t f = 0; f < F; f++) {
            sa.frame = (int)f;
            launch_sync_find(sa, h->stream);
            launch_sync_finish(sa, h->stream);
            DemodArgs d1 = da; d1.frame_first = (int)f; d1.frame_count = 1; d1.con = nullptr; d1.osc_stats = nullptr;
            d1.chunk_count = (3 + da.chunk_len - 1) / da.chunk_len;          // the chunks that hold the FIC symbols 1..3 (demod_chunk may be 1 or 2)
            launch_demod(d1, (int)B, h->stream);
            g.frame_sel = (int)f + 1; k.frame_sel = (int)f + 1;
            if (fic_sp) { spa.fic_frame_sel = (int)f + 1; launch_sp(spa, h->sp1_two, sp_variant_for(c.nsteps), h->stream); }
            else { launch_fic_gather(g, h->stream); launch_viterbi(v, h->stream); }
            launch_fib_crc(k, h->stream);
            CrcArgs kf = k; kf.frame_sel = 0; kf.frame_first = (int)f; kf.frame_count = 1;
            launch_fic_ratio(kf, h->stream);
        }
    }
    mark(dabphy_handle::ST_DEMOD, false);
    launch_demod(da, (int)B, h->stream);
    tick(2);
    mark(dabphy_handle::ST_DEMOD, true);
    if (!replay && (h->cfg.pipeline_sync == 1 || h->cfg.pipeline_sync == 3)) HIPCHK(h, hipEventRecord(h->ev_chain_gate, h->stream));
    // (cfg.sync_early: in front of the decoder (0, the default: neutral on the headline, 0.2 ms on a batch of drifting ensembles, whose
    // window searches run one after the other in the find chain -- latency-bound work for one work-group per ensemble that belongs beside the
    // decoder); behind it (1); in front only while the last pass met ensembles whose window moves (2); 3: an experiment, see below)
    const bool early = h->chain_early || (h->cfg.pipeline_sync != 2 && (h->cfg.sync_early == 0 || h->cfg.sync_early == 3 || (h->cfg.sync_early == 2 && h->drift_seen)));
    if (!replay && depth && early) {
        // the next batch's synchroniser is handed to the device BEFORE this batch's decoder (whose persistent waves would otherwise hold
        // every wave slot until the end of the step: the synchroniser then runs in the step's tail)
        if (h->cfg.pipeline_sync != 2) HIPCHK(h, hipStreamWaitEvent(h->sync_stream, h->ev_chain_gate, 0));
        h->wide_front_recorded = false;
        for (; h->ahead < 1 + depth; h->ahead++) if ((r = queue_chain(h, (cur + h->ahead) % ND, F))) return r;
        // (sync_early 3: which of the two launches that become ready with the demod kernel's end gets the wave slots is the hardware's
        // choice -- the decoder's persistent waves, once resident, give none back --: the decoder's launch waits for the wide pass
        // proper, ~0.8 ms of throughput work that then has the device to itself; the find chain's rounds run beside the decoder)
        if (h->cfg.sync_early == 3 && h->wide_front_recorded) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_wide_front, 0));
    }
    // dabphy_set_auto_superframes(2): the PREVIOUS batch's superframe filter pass, beside this batch's FFT stage; this batch's decoders
    // wait for it on the device before they overwrite the class outputs it reads
    if (!replay && h->sf_auto && h->sf_deferred && (r = launch_deferred_superframes(h))) return r;
    if (h->sf_def_inflight) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_rs_done, 0));
    SnrArgs sn{}; sn.state = h->d_dec; sn.desc = d_desc; sn.n_ens = (int)B; sn.n_frames = (int)F; sn.prs_mag = da.prs_mag; sn.snr_out = h->s_snr.as<float>();

    // SNR + FIC + TII beside the MSC decode, on the auxiliary stream.  The FIC's 4 code words per frame ride in the fused launch
    // (dabphy_fused.hip) unless that is switched off; then -- B*F/16 wavefronts of 774 serial trellis steps -- they are decoded here,
    // by their own gather + Viterbi pair that fills execution slots beside the MSC classes.
    const bool fic_fused = h->fplan.fic_in;
    hipStream_t fs = h->aux_stream;
    VitClass ficc = fic_c;
    ficc.sym = h->fsym.as<uint32_t>(); ficc.dec = h->fdec.as<uint2>(); ficc.out = h->s_fib.as<uint8_t>();
    {
        const bool snr_main = (h->stream_layout & 16) != 0;
        if (snr_main) { mark(dabphy_handle::ST_SNR, false); launch_snr(sn, h->stream); mark(dabphy_handle::ST_SNR, true); }
        HIPCHK(h, hipEventRecord(h->ev_demod_done, h->stream));
        HIPCHK(h, hipStreamWaitEvent(fs, h->ev_demod_done, 0));
        // the SNR estimate feeds nothing on the device: off the main stream, so that the MSC decode starts the moment the demod kernel ends
        if (!snr_main) {
            mark(dabphy_handle::ST_SNR, false, fs);
            launch_snr(sn, fs);
            mark(dabphy_handle::ST_SNR, true, fs);
        }
        if (!fic_fused) {
            FicGatherArgs g{}; g.soft = da.soft; g.soft_ring = ring_frames; g.frame_stride = SOFT_PER_FRAME; g.soft_ens_stride = ens_stride; g.desc = d_desc;
            g.n_ens = (int)B; g.n_frames = (int)F; g.map = h->d_fic_map; g.c = ficc;
            mark(dabphy_handle::ST_FIC, false, fs);
            launch_fic_gather(g, fs);
            VitArgs v{}; v.c = ficc; v.prbs_words = h->d_prbs_words;
            launch_viterbi(v, fs);
        }
        h->tii_ran = false;
        if (h->tii_on) {
            // TII side path (ofdm-processor.cpp:462-466 -> TIIDecoder): needs only the samples and the frame descriptors
            h->tii_max_events = TII_MAX_LIKELY * h->cfg.max_frames;
            TiiArgs ta{};
            ta.tab = h->tab; ta.iq = h->s_iq; ta.iq_stride = h->s_stride; ta.ring = (int64_t)h->s_ring; ta.desc = d_desc; ta.n_ens = (int)B; ta.n_frames = (int)F;
            ta.rot = h->tii_rot.as<cf32>(); ta.rank = h->tii_rank.as<int32_t>(); ta.pattern = h->tii_pat.as<uint8_t>();
            ta.abs_err = h->tii_err.as<float>(); ta.likely = h->tii_likely.as<int32_t>(); ta.state = h->tii_state.as<TiiSlot>();
            ta.events = h->tii_events.as<TiiEvent>(); ta.n_events = h->tii_nev.as<int32_t>(); ta.max_events = (int)h->tii_max_events;
            ta.overflow = h->tii_ovf.as<int32_t>();
            launch_tii(ta, fs);
            h->tii_ran = true;
        }
        // the host's copies of the descriptors and SNR reports leave here, beside the decoder, instead of behind the step's last kernel
        launch_copy_out(d_desc, h->h_desc, (size_t)B * F * sizeof(FrameDesc), fs);      // (kernel stores, not copy-engine packets: k_ingest.hip, launch_copy_out)
        launch_copy_out(h->s_snr.p, h->h_snr, (size_t)B * F * sizeof(float), fs);
        HIPCHK(h, hipEventRecord(h->ev_aux_done, fs));
    }
    // pairs selected since the last batch learn the CIF count they start at (their time de-interleaver fills from here, dab-audio.cpp:146-149)
    for (auto& cls : h->classes) if (cls.cif0_pending) launch_pair_cif0(cls.pair_tab.as<MscPair>(), (int)cls.pairs.size(), d_desc, (int)F, h->stream);
    // MSC (+ FIC): every class the plan holds in ONE launch; the stage events bracket all of it
    h->last_frames = F;
    h->sf_stats_ready = false; h->h_sf_stats_valid = false;
    if (h->fplan.args.n_work > 0) {
        FusedArgs fa = h->fplan.args; fa.desc = d_desc;
        h->fplan.args = fa; h->fplan.launched = true;
        mark(dabphy_handle::ST_MSC_VITERBI, false);
        if (h->fplan.use_sp) launch_sp(fa, h->fplan.sp_two, h->fplan.sp_variant, h->stream);
        else { const FusedSplit sp{h->tb_no_walkers ? nullptr : h->tb_stream, h->ev_tb_fork, h->ev_tb_join}; launch_viterbi_fused(fa, h->fplan.variant, h->fplan.n_slots, h->stream, fa.done ? &sp : nullptr); }
        mark(dabphy_handle::ST_MSC_VITERBI, true);
        if (fa.done) {
            if (!h->h_tb_gave_up) { void* p = nullptr; HIPCHK(h, hipHostMalloc(&p, sizeof(uint32_t), hipHostMallocDefault)); h->h_tb_gave_up = reinterpret_cast<uint32_t*>(p); }
            launch_copy_out(fa.done + fa.n_work + 1, h->h_tb_gave_up, sizeof(uint32_t), h->stream);
        }
        if (fic_fused) HIPCHK(h, hipEventRecord(h->ev_fused_done, h->stream));
    }
    {
        // FIB CRCs, the FIC success ratio (and with it the verdict of exact batch mode), the host's copies of both.  With the FIC in the
        // fused launch they wait for nothing but that launch, on a stream of their own: the SNR sums (2048 short waves that feed nothing
        // on the device and find no slot while the persistent decoder waves hold them all) must not stand in front of the FIC verdict
        if (fic_fused) { fs = h->fic_stream; HIPCHK(h, hipStreamWaitEvent(fs, h->ev_fused_done, 0)); mark(dabphy_handle::ST_FIC, false, fs); }
        CrcArgs k{}; k.fib = ficc.out; k.ok = h->s_ok.as<uint8_t>(); k.state = h->d_dec; k.desc = d_desc; k.n_ens = (int)B; k.n_frames = (int)F; k.disable_coarse = h->cfg.disable_coarse;
        launch_fib_crc(k, fs);
        k.any_effective = h->d_any_eff;
        if (!replay) launch_fic_ratio(k, fs);                    // (the second pass of exact batch mode has advanced the ratio frame by frame)
        launch_copy_out(h->d_any_eff, h->h_any_eff, sizeof(int32_t), fs);
        mark(dabphy_handle::ST_FIC, true, fs);
        launch_copy_out(h->s_fib.p, h->h_fib, (size_t)B * F * 384, fs);
        launch_copy_out(h->s_ok.p, h->h_ok, (size_t)B * F * 12, fs);
        HIPCHK(h, hipEventRecord(h->ev_fic_done, fs));
    }
    // classes the fused launch does not take (DABPHY_FUSED_MSC=0, a window schedule the kernel cannot follow, a span beyond 4 GiB): two
    // kernels each, one class after the other (they share the Viterbi scratch)
    {
        bool first_two = true; size_t ci = 0;
        for (auto& cls : h->classes) {
            const bool fused = std::find(h->fplan.class_idx.begin(), h->fplan.class_idx.end(), (int)ci) != h->fplan.class_idx.end();
            ci++;
            if (fused) continue;
            if (debug_env("DABPHY_DEBUG")) fprintf(stderr, "dabphy: class %zu (%d bits) through k_msc_gather + k_viterbi\n", ci - 1, cls.prot.nbits);
            VitClass c{};
            const int P = (int)cls.pairs.size();
            if ((r = prepare_class(h, c, cls.prot.nbits, (int)(4 * F * (uint32_t)P), 1))) return r;
            c.out = cls.out.as<uint8_t>();
            MscGatherArgs g{}; g.soft = da.soft; g.soft_ring = ring_frames; g.soft_ens_stride = ens_stride; g.state = h->d_state; g.n_ens = (int)B; g.n_frames = (int)F;
            g.map = cls.map.as<int16_t>(); g.pairs = cls.pair_tab.as<MscPair>(); g.tiles = cls.tiles.as<int32_t>(); g.n_pairs = P; g.desc = d_desc; g.c = c;
            if (first_two) mark(dabphy_handle::ST_MSC_GATHER, false);
            launch_msc_gather(g, h->stream);
            VitArgs v{}; v.c = c; v.prbs_words = h->d_prbs_words;
            launch_viterbi(v, h->stream);
            first_two = false;
        }
        if (!first_two) mark(dabphy_handle::ST_MSC_GATHER, true);       // (gather + decode pairs of all such classes)
    }
    if (h->sf_auto && !h->sf_deferred) {
        if ((r = launch_superframe_stats(h))) return r;
        h->sf_stats_ready = true;
        launch_copy_out(h->sf_stats.p, h->h_sf_stats, sizeof(int32_t) * 4 * B, h->stream);
        h->h_sf_stats_valid = true;
    }
    return DABPHY_OK;
    };
    if (replay_armed(h, F)) {
        // what the decoders carry from batch to batch, as it is in front of this one (the synchroniser's share was saved when this
        // batch's chain was queued: queue_chain)
        HIPCHK(h, hipMemcpyAsync(h->snap_dec.p, h->d_dec, sizeof(DecState) * B, hipMemcpyDeviceToDevice, h->stream));
        if (h->tii_state.p && h->snap_tii.p) HIPCHK(h, hipMemcpyAsync(h->snap_tii.p, h->tii_state.p, h->tii_state.cap, hipMemcpyDeviceToDevice, h->stream));
        if (!h->sf_deferred) for (auto& cls : h->classes) if (cls.sf_state.p && cls.sf_snap.p) HIPCHK(h, hipMemcpyAsync(cls.sf_snap.p, cls.sf_state.p, cls.sf_state.cap, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemsetAsync(h->d_any_eff, 0, sizeof(int32_t), h->stream));
    }
    if ((r = decode(false))) return r;
    if (depth) {
        if (h->cfg.pipeline_sync != 2) HIPCHK(h, hipStreamWaitEvent(h->sync_stream, h->ev_chain_gate, 0));
        for (; h->ahead < 1 + depth; h->ahead++) if ((r = queue_chain(h, (cur + h->ahead) % ND, F))) return r;
    }
    h->desc_sel = (cur + 1) % ND; h->ahead--;
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_fic_done, 0));
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_aux_done, 0));
    h->last_frames = F;
    tick(3);
    if ((r = sync(h))) return r;
    tick(4);
    if (replay_armed(h, F) && *h->h_any_eff) {
        // Exact batch mode: a coarse-corrector decision of this batch was taken with a stale FIC ratio and can have mattered.  Everything
        // the batch changed is put back -- synchroniser state (as saved when its chain was queued), decoder state, superframe windows,
        // TII sums; the soft-bit ring and the outputs are simply written again -- and the batch is decoded a second time with the
        // feedback the reference has; the chains that ran ahead on the wrong state are queued again behind it.
        HIPCHK(h, hipStreamSynchronize(h->sync_stream));
        HIPCHK(h, hipStreamSynchronize(h->aux_stream));
        HIPCHK(h, hipStreamSynchronize(h->fic_stream));
        for (int i = 0; i < ND; i++) h->wide_pending[i] = false;
        HIPCHK(h, hipMemcpyAsync(h->d_state, h->snap_state[cur].p, sizeof(RxState) * B, hipMemcpyDeviceToDevice, h->stream));
        if (h->snap_hist[cur].p && h->s_hist.p) HIPCHK(h, hipMemcpyAsync(h->s_hist.p, h->snap_hist[cur].p, (size_t)B * HIST_CAP * sizeof(FrameDesc), hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_dec, h->snap_dec.p, sizeof(DecState) * B, hipMemcpyDeviceToDevice, h->stream));
        if (h->tii_state.p && h->snap_tii.p) HIPCHK(h, hipMemcpyAsync(h->tii_state.p, h->snap_tii.p, h->tii_state.cap, hipMemcpyDeviceToDevice, h->stream));
        if (!h->sf_deferred) for (auto& cls : h->classes) if (cls.sf_state.p && cls.sf_snap.p) HIPCHK(h, hipMemcpyAsync(cls.sf_state.p, cls.sf_snap.p, cls.sf_state.cap, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemsetAsync(h->d_any_eff, 0, sizeof(int32_t), h->stream));
        if ((r = decode(true))) return r;
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_fic_done, 0));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_aux_done, 0));
        if ((r = sync(h))) return r;
        // the batches synchronised ahead started from the state the first pass left: again, from the right one.  (The chain reads the
        // FIC ratio: the main stream has just been drained.)
        for (int i = 1; i <= depth; i++) if ((r = queue_chain(h, (cur + i) % ND, F))) return r;
        h->n_replayed_batches++;
    }
    if (h->sf_auto && h->sf_deferred) {
        // (the main stream has waited for the previous batch's pass: its totals are in host memory; this batch's pass is the next call's)
        h->sf_def_inflight = false;
        h->sf_def_pending = true; h->sf_def_desc = d_desc; h->sf_def_frames = F;
    }
    // the host's mirror of the pair tables follows what k_pair_cif0 wrote (same rule, from the host's copy of the descriptors)
    for (auto& cls : h->classes) if (cls.cif0_pending) {
        for (MscPair& p : cls.pairs) if (p.cif0 < 0) p.cif0 = 4 * h->h_desc[(size_t)p.ens * F].frame_no;
        cls.cif0_pending = false;
    }
    if (h->fplan.args.done && h->fplan.launched && !h->fplan.use_sp) {
        // split traceback: walkers that gave up on a group's flag (k_viterbi.hip: tb_consume) would have decoded garbage -- never silently
        // (the counter came back with the batch: a copy queued behind the launch, in front of the call's final synchronisation)
        if (h->h_tb_gave_up && *h->h_tb_gave_up) { h->err = "split traceback: " + std::to_string(*h->h_tb_gave_up) + " groups were walked back without their decisions having been published"; return DABPHY_ERR_HIP; }
    }
    if (g_tl_on) { for (int i = 0; i < 5; i++) g_tl.acc[i] += tl[i]; g_tl.n++; if (g_tl.n % 8 == 0) fprintf(stderr, "dabphy timing [us]: before resolve %.1f, resolved %.1f, demod launched %.1f, all launched %.1f, synced %.1f (n=%ld)\n", g_tl.acc[0] / g_tl.n, g_tl.acc[1] / g_tl.n, g_tl.acc[2] / g_tl.n, g_tl.acc[3] / g_tl.n, g_tl.acc[4] / g_tl.n, g_tl.n); }
    { float t = 0; h->chain_ms = (hipEventElapsedTime(&t, h->ev_chain_beg[cur], h->ev_chain_end[cur]) == hipSuccess) ? t : 0.0f; }
    return DABPHY_OK;
}

} // extern "C"
