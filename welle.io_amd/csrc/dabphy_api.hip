// welle.io_amd/csrc/dabphy_api.hip -- C ABI of libdabphy_hip.so (include/dabphy.h): handle, device tables,
// buffer management and kernel sequencing.  No arithmetic of the hot path happens on the host.
#include "dabphy_internal.h"
#include <cstddef>

extern "C" {

uint32_t dabphy_abi_version(void) { return DABPHY_ABI_VERSION; }
size_t dabphy_struct_size(int32_t which)
{
    switch (which) {
        case DABPHY_STRUCT_CONFIG: return sizeof(dabphy_config);
        case DABPHY_STRUCT_FRAME_INFO: return sizeof(dabphy_frame_info);
        case DABPHY_STRUCT_SF_EVENT: return sizeof(dabphy_sf_event);
        case DABPHY_STRUCT_SUBCHANNEL: return sizeof(dabphy_subchannel);
        case DABPHY_STRUCT_PROTECTION: return sizeof(dabphy_protection);
        case DABPHY_STRUCT_TII_MEASUREMENT: return sizeof(dabphy_tii_measurement);
        case DABPHY_STRUCT_MSC_DESC: return sizeof(dabphy_msc_desc);
    }
    return 0;
}

// The configuration as the callers of rounds 1-3 were compiled with it: no size member, twelve 32-bit fields.  Their objects call the
// exported symbol `dabphy_create`, which stays and means exactly this layout; everything added since takes its default.
struct dabphy_config_r3 {
    uint32_t n_ensembles, max_frames; int32_t device, fft_placement, disable_coarse, want_constellation, want_impulse_response, demod_chunk,
             freqsync_method, pipeline_sync, serial_sync, no_batch_replay;
};
static_assert(sizeof(dabphy_config_r3) == 48 && offsetof(dabphy_config, n_ensembles) == 4 && sizeof(dabphy_config) == 4 + sizeof(dabphy_config_r3) + 8, "dabphy_config: fields are appended only");
int dabphy_create(const dabphy_config_r3* old, dabphy_handle** out)
{
    if (!old || !out) return DABPHY_ERR_INVALID;
    dabphy_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = (uint32_t)(sizeof(uint32_t) + sizeof *old);                  // the sized form of what that caller knew
    memcpy(&cfg.n_ensembles, old, sizeof *old);
    return dabphy_create_v2(&cfg, out);
}

int dabphy_create_v2(const dabphy_config* cfg_in, dabphy_handle** out)
{
    // the caller's structure may be older (shorter: the missing tail = defaults) but not newer than this library's
    if (!cfg_in || !out || cfg_in->struct_size < offsetof(dabphy_config, fft_placement) || cfg_in->struct_size > sizeof(dabphy_config) || cfg_in->struct_size % 4) return DABPHY_ERR_INVALID;
    dabphy_config cfg_full;
    memset(&cfg_full, 0, sizeof cfg_full);
    memcpy(&cfg_full, cfg_in, cfg_in->struct_size);
    if (cfg_in->struct_size <= offsetof(dabphy_config, fft_placement)) cfg_full.fft_placement = 2;      // (a caller that knows no synchroniser options gets the reference's defaults,
    if (cfg_in->struct_size <= offsetof(dabphy_config, freqsync_method)) cfg_full.freqsync_method = 2;  //  radio-receiver-options.h:66-84: ThresholdBeforePeak, PatternOfZeros)
    cfg_full.struct_size = (uint32_t)sizeof(dabphy_config);
    const dabphy_config* const cfg = &cfg_full;
    if (cfg->n_ensembles < 1 || cfg->max_frames < 1) return DABPHY_ERR_INVALID;
    // include/dabphy.h "limits": frame and code word counts are 32-bit quantities in the kernels' argument blocks
    if (cfg->max_frames > DABPHY_MAX_FRAMES || (uint64_t)cfg->n_ensembles * cfg->max_frames > DABPHY_MAX_ENSEMBLE_FRAMES) return DABPHY_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) return DABPHY_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return DABPHY_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return DABPHY_ERR_NO_DEVICE;     // kernels are built for gfx950 only
    if (hipSetDevice(cfg->device) != hipSuccess) return DABPHY_ERR_NO_DEVICE;
    dabphy_handle* h = new dabphy_handle();
    h->cfg = *cfg;
    h->subch_e.resize(cfg->n_ensembles); h->subch_next.resize(cfg->n_ensembles); h->where.resize(cfg->n_ensembles);
    // big batches: fewer reference-symbol transforms; small ones: more work-groups.  (Longer chunks -- 38, 75 symbols -- are 2-3 %
    // faster when the kernel runs alone, dabphy_time_demod, and 1-4 % slower inside the pipelined step: measured, round 2.)
    if (h->cfg.demod_chunk <= 0) h->cfg.demod_chunk = ((int64_t)h->cfg.n_ensembles * h->cfg.max_frames >= 1024) ? 25 : 15;
    if (h->cfg.demod_chunk > 75) h->cfg.demod_chunk = 75;
    snprintf(h->devname, sizeof h->devname, "%s (%s)", prop.name, prop.gcnArchName);
    int r = 0;
    auto fail = [&](int code) { dabphy_destroy(h); return code; };
    if (h->cfg.fft_placement < 0 || h->cfg.fft_placement > 2 || h->cfg.freqsync_method < 0 || h->cfg.freqsync_method > 2 || h->cfg.pipeline_sync < 0 || h->cfg.pipeline_sync > 3 || h->cfg.decode_shape < 0 || h->cfg.decode_shape > 3 || h->cfg.sync_early < 0 || h->cfg.sync_early > 3) return fail(DABPHY_ERR_INVALID);
    const HostTables& T = host_tables();
    if ((r = upload_const(h, &h->d_tw, T.tw))) return fail(r);
    if ((r = upload_const(h, &h->d_ref, T.ref))) return fail(r);
    if ((r = upload_const(h, &h->d_nco, T.nco))) return fail(r);
    if ((r = upload_const(h, &h->d_bin2soft, T.bin2soft))) return fail(r);
    if ((r = upload_const(h, &h->d_prbs_words, T.prbs_words))) return fail(r);
    dabphy_protection pf; protection_fic(&pf);
    if ((r = upload_const(h, &h->d_fic_map, depuncture_map(&pf)))) return fail(r);
    if ((r = upload_const(h, &h->d_osc_unsafe, T.osc_unsafe))) return fail(r);
    h->tab.tw = h->d_tw; h->tab.ref = h->d_ref; h->tab.nco = h->d_nco; h->tab.bin2soft = h->d_bin2soft; h->tab.prbs_bytes = nullptr;
    h->tab.osc_unsafe = h->d_osc_unsafe; h->tab.n_osc_unsafe = T.n_osc_unsafe;
    {
        void* p = nullptr;
        if (hipMalloc(&p, 2 * sizeof(unsigned long long)) != hipSuccess) return fail(DABPHY_ERR_NOMEM);
        h->owned.push_back(p); h->d_osc_stats = reinterpret_cast<unsigned long long*>(p);
        if (hipMemset(p, 0, 2 * sizeof(unsigned long long)) != hipSuccess) return fail(DABPHY_ERR_HIP);
    }
    void* st = nullptr;
    if (hipMalloc(&st, sizeof(RxState) * cfg->n_ensembles) != hipSuccess) return fail(DABPHY_ERR_NOMEM);
    h->owned.push_back(st); h->d_state = reinterpret_cast<RxState*>(st);
    if (hipMemset(st, 0, sizeof(RxState) * cfg->n_ensembles) != hipSuccess) return fail(DABPHY_ERR_HIP);
    void* ds = nullptr;
    if (hipMalloc(&ds, sizeof(DecState) * cfg->n_ensembles) != hipSuccess) return fail(DABPHY_ERR_NOMEM);
    h->owned.push_back(ds); h->d_dec = reinterpret_cast<DecState*>(ds);
    if (hipMemset(ds, 0, sizeof(DecState) * cfg->n_ensembles) != hipSuccess) return fail(DABPHY_ERR_HIP);
    {
        // Stream placement.  Measured in round 6 (tools/probe_handle_order.py, profiles/r06_step_variants.txt): the SECOND handle a process
        // opens decodes the benchmark batch 3 % faster than the first (decoder 6.0 against 6.6-6.7 ms, step 10.1 against 10.4) -- also
        // behind a one-ensemble handle that never decodes anything, not behind a 12 GB dummy allocation or five torch streams: what
        // matters is which hardware queues the runtime gives the handle's streams (it multiplexes streams onto a few queues in creation
        // order), i.e. who wins the wave slots when the decoder and the next batch's synchroniser become ready together.  A handle
        // therefore first creates the five streams such a predecessor would have created -- same order, same priorities, never used --
        // and keeps them until it is destroyed.
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = 0; hi = 0; }
        int n_placeholders = 5;
#ifdef DABPHY_EXPERIMENTS
        if (const char* e = getenv("DABPHY_STREAM_LAYOUT")) { h->stream_layout = atoi(e); n_placeholders = (h->stream_layout & 1) ? 5 : 0; }      // (tools/probe_streams.py)
#endif
        for (int i = 0; i < n_placeholders; i++) {
            hipStream_t ps = nullptr;
            if ((i == 1 ? hipStreamCreateWithPriority(&ps, hipStreamNonBlocking, hi) : hipStreamCreateWithFlags(&ps, hipStreamNonBlocking)) != hipSuccess) return fail(DABPHY_ERR_HIP);
            h->placeholder_streams.push_back(ps);
        }
    }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return fail(DABPHY_ERR_HIP);
    {   // the frame chain is short serial work: give its queue the highest dispatch priority
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = 0; hi = 0; }
        if (hipStreamCreateWithPriority(&h->sync_stream, hipStreamNonBlocking, hi) != hipSuccess) return fail(DABPHY_ERR_HIP);
    }
    if (hipEventCreate(&h->ev_sync_done) != hipSuccess) return fail(DABPHY_ERR_HIP);
    if (hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking) != hipSuccess) return fail(DABPHY_ERR_HIP);
    if (hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) != hipSuccess) return fail(DABPHY_ERR_HIP);
    if (h->stream_layout & 2) h->fic_stream = h->aux_stream;      // (experiment: fewer streams -- measured slower, profiles/r06_step_variants.txt)
    else if (hipStreamCreateWithFlags(&h->fic_stream, hipStreamNonBlocking) != hipSuccess) return fail(DABPHY_ERR_HIP);
    if (hipEventCreateWithFlags(&h->ev_aux_done, hipEventDisableTiming) != hipSuccess) return fail(DABPHY_ERR_HIP);
    for (int i = 0; i < 2; i++) if (hipEventCreateWithFlags(&h->ev_ingest[i], hipEventDisableTiming) != hipSuccess) return fail(DABPHY_ERR_HIP);
    if (hipEventCreateWithFlags(&h->ev_chain_gate, hipEventDisableTiming) != hipSuccess) return fail(DABPHY_ERR_HIP);
    if (hipEventCreateWithFlags(&h->ev_demod_done, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->ev_fic_done, hipEventDisableTiming) != hipSuccess) return fail(DABPHY_ERR_HIP);
    if (hipEventCreateWithFlags(&h->ev_fused_done, hipEventDisableTiming) != hipSuccess) return fail(DABPHY_ERR_HIP);
    if ((r = fused_class_tables(h, pf, true, h->fic_steps, h->fic_windows))) return fail(r);
    for (int i = 0; i < dabphy_handle::N_DESC; i++) if (hipEventCreate(&h->ev_chain_beg[i]) != hipSuccess || hipEventCreate(&h->ev_chain_end[i]) != hipSuccess) return fail(DABPHY_ERR_HIP);
    for (int i = 0; i < dabphy_handle::N_DESC; i++) if (hipEventCreateWithFlags(&h->ev_wide_done[i], hipEventDisableTiming) != hipSuccess) return fail(DABPHY_ERR_HIP);
    {
        // the wide synchroniser pass's verdict flags: page-locked HOST memory the last judge kernel writes directly (d_any_redo = the
        // device's address of the same words)
        void* p = nullptr; void* dp = nullptr;
        // ([N_DESC] "the serial chain has slots left" + [N_DESC] "the find chain settled frames": ensembles whose window moves)
        if (hipHostMalloc(&p, sizeof(int32_t) * 2 * dabphy_handle::N_DESC, hipHostMallocDefault) != hipSuccess) return fail(DABPHY_ERR_NOMEM);
        h->h_any_redo = reinterpret_cast<int32_t*>(p);
        for (int i = 0; i < 2 * dabphy_handle::N_DESC; i++) h->h_any_redo[i] = 0;
        if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess) return fail(DABPHY_ERR_HIP);
        h->d_any_redo = reinterpret_cast<int32_t*>(dp);
    }
    {
        void* p = nullptr; const size_t n = (size_t)cfg->n_ensembles * cfg->max_frames;
        if (hipHostMalloc(&p, n * sizeof(FrameDesc), hipHostMallocDefault) != hipSuccess) return fail(DABPHY_ERR_NOMEM);
        h->h_desc = reinterpret_cast<FrameDesc*>(p);
        if (hipHostMalloc(&p, n * sizeof(float), hipHostMallocDefault) != hipSuccess) return fail(DABPHY_ERR_NOMEM);
        h->h_snr = reinterpret_cast<float*>(p);
        if (hipHostMalloc(&p, n * 384, hipHostMallocDefault) != hipSuccess) return fail(DABPHY_ERR_NOMEM);
        h->h_fib = reinterpret_cast<uint8_t*>(p);
        if (hipHostMalloc(&p, n * 12, hipHostMallocDefault) != hipSuccess) return fail(DABPHY_ERR_NOMEM);
        h->h_ok = reinterpret_cast<uint8_t*>(p);
        if (hipHostMalloc(&p, (size_t)cfg->n_ensembles * 4 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) return fail(DABPHY_ERR_NOMEM);
        h->h_sf_stats = reinterpret_cast<int32_t*>(p);
    }
    h->exact_batch = cfg->no_batch_replay == 0;
#ifdef DABPHY_EXPERIMENTS
    if (const char* e = getenv("DABPHY_EXACT_BATCH")) h->exact_batch = atoi(e) != 0;   // (experiments: overrides the configuration)
#endif
    {
        void* p = nullptr;
        if (hipMalloc(&p, sizeof(int32_t)) != hipSuccess) return fail(DABPHY_ERR_NOMEM);
        h->owned.push_back(p); h->d_any_eff = reinterpret_cast<int32_t*>(p);
        if (hipMemset(p, 0, sizeof(int32_t)) != hipSuccess) return fail(DABPHY_ERR_HIP);
        if (hipHostMalloc(&p, sizeof(int32_t), hipHostMallocDefault) != hipSuccess) return fail(DABPHY_ERR_NOMEM);
        h->h_any_eff = reinterpret_cast<int32_t*>(p); *h->h_any_eff = 0;
    }
    h->wide_sync = cfg->serial_sync == 0;
#ifdef DABPHY_EXPERIMENTS
    if (const char* e = getenv("DABPHY_SYNC_WIDE")) h->wide_sync = atoi(e) != 0;    // (experiments: overrides the configuration)
#endif
#ifdef DABPHY_EXPERIMENTS
    if (const char* e = getenv("DABPHY_FUSED_MSC")) h->fused_msc = atoi(e) != 0;
    if (const char* e = getenv("DABPHY_FUSED_FIC")) h->fused_fic = atoi(e) != 0;
    if (const char* e = getenv("DABPHY_SP_MAX_CW")) h->sp_max_codewords = (uint32_t)atoll(e);
    if (const char* e = getenv("DABPHY_SP2_MIN_CW")) h->sp2_min_codewords = (uint32_t)atoll(e);
    if (const char* e = getenv("DABPHY_SP2_TB_WARM")) h->sp2_tb_warm = (uint32_t)atoll(e);
    if (const char* e = getenv("DABPHY_SP2_TB_RESIDENT")) h->sp2_tb_resident = (uint32_t)atoll(e);
    if (const char* e = getenv("DABPHY_CHAIN_EARLY")) h->chain_early = atoi(e) != 0;
    if (const char* e = getenv("DABPHY_TB_SPLIT")) h->tb_split = atoi(e) != 0;      // (the fused decode's traceback as a pass of its own: dabphy_test_traceback_split)
#endif
    for (int i = 0; i < dabphy_handle::ST_COUNT; i++)
        if (hipEventCreate(&h->ev_beg[i]) != hipSuccess || hipEventCreate(&h->ev_end[i]) != hipSuccess) return fail(DABPHY_ERR_HIP);
#ifdef DABPHY_WRONG_RESULTS_BUILD
    fprintf(stderr, "dabphy: THIS LIBRARY WAS BUILT WITH AN FM_EXP_* TIMING SWITCH: its decoder output is wrong by construction\n");
    strncat(h->devname, " [timing-experiment build: wrong results]", sizeof h->devname - strlen(h->devname) - 1);
#endif
    *out = h;
    return DABPHY_OK;
}

// every device buffer of one protection class (apply_subchannels replaces the classes, dabphy_destroy ends them)
void free_class(dabphy_handle::MscClass& c)
{
    hipError_t e = hipSuccess;
    DevBuf* bufs[] = {&c.map, &c.pair_tab, &c.tiles, &c.out, &c.steps[0], &c.steps[1], &c.steps[2], &c.sf_state, &c.sf_snap};
    for (DevBuf* b : bufs) if (b->p) { e = hipFree(b->p); b->p = nullptr; b->cap = 0; }
    (void)e;
}

void dabphy_destroy(dabphy_handle* h)
{
    DeviceBind dev_(h);
    if (!h) return;
    hipError_t e;
    if (h->sync_stream) { e = hipStreamSynchronize(h->sync_stream); e = hipStreamDestroy(h->sync_stream); }
#ifdef SYNC_CHAIN_TS
    if (getenv("DABPHY_CHAIN_TS")) dump_chain_ts();
#endif
    if (h->ev_sync_done) e = hipEventDestroy(h->ev_sync_done);
    if (h->aux_stream) { e = hipStreamSynchronize(h->aux_stream); e = hipStreamDestroy(h->aux_stream); }
    if (h->copy_stream) { e = hipStreamSynchronize(h->copy_stream); e = hipStreamDestroy(h->copy_stream); }
    if (h->fic_stream && h->fic_stream != h->aux_stream) { e = hipStreamSynchronize(h->fic_stream); e = hipStreamDestroy(h->fic_stream); }
    for (hipStream_t ps : h->placeholder_streams) if (ps) e = hipStreamDestroy(ps);
    if (h->drain_stream && h->drain_stream != h->copy_stream) { e = hipStreamSynchronize(h->drain_stream); e = hipStreamDestroy(h->drain_stream); }
    if (h->tb_stream) { e = hipStreamSynchronize(h->tb_stream); e = hipStreamDestroy(h->tb_stream); }
    if (h->ev_rs_done) e = hipEventDestroy(h->ev_rs_done);
    if (h->ev_wide_front) e = hipEventDestroy(h->ev_wide_front);
    if (h->h_tb_gave_up) e = hipHostFree(h->h_tb_gave_up);
    if (h->ev_tb_fork) e = hipEventDestroy(h->ev_tb_fork);
    if (h->ev_tb_join) e = hipEventDestroy(h->ev_tb_join);
    if (h->ev_drain_done) e = hipEventDestroy(h->ev_drain_done);
    if (h->ev_drain_staged) e = hipEventDestroy(h->ev_drain_staged);
    if (h->drain_stage.p) e = hipFree(h->drain_stage.p);
    if (h->ev_aux_done) e = hipEventDestroy(h->ev_aux_done);
    for (int i = 0; i < 2; i++) if (h->ev_ingest[i]) e = hipEventDestroy(h->ev_ingest[i]);
    if (h->ev_demod_done) e = hipEventDestroy(h->ev_demod_done);
    if (h->ev_chain_gate) e = hipEventDestroy(h->ev_chain_gate);
    for (int i = 0; i < dabphy_handle::N_DESC; i++) { if (h->ev_wide_done[i]) e = hipEventDestroy(h->ev_wide_done[i]); if (h->s_redo[i].p) e = hipFree(h->s_redo[i].p); }
    if (h->h_any_redo) e = hipHostFree(h->h_any_redo);
    if (h->h_desc) e = hipHostFree(h->h_desc);
    if (h->h_snr) e = hipHostFree(h->h_snr);
    if (h->h_fib) e = hipHostFree(h->h_fib);
    if (h->h_ok) e = hipHostFree(h->h_ok);
    if (h->h_sf_stats) e = hipHostFree(h->h_sf_stats);
    if (h->h_any_eff) e = hipHostFree(h->h_any_eff);
    for (int i = 0; i < dabphy_handle::N_DESC; i++) if (h->snap_state[i].p) e = hipFree(h->snap_state[i].p);
    for (int i = 0; i < dabphy_handle::N_DESC; i++) if (h->snap_hist[i].p) e = hipFree(h->snap_hist[i].p);
    if (h->snap_dec.p) e = hipFree(h->snap_dec.p);
    if (h->snap_tii.p) e = hipFree(h->snap_tii.p);
    if (h->ev_fic_done) e = hipEventDestroy(h->ev_fic_done);
    if (h->ev_fused_done) e = hipEventDestroy(h->ev_fused_done);
    { DevBuf* fb[] = {&h->fused_cls, &h->fused_work, &h->fused_dec_off, &h->fused_done, &h->fic_steps[0], &h->fic_steps[1], &h->fic_steps[2], &h->sp1_cls, &h->sp1_work}; for (DevBuf* b : fb) if (b->p) e = hipFree(b->p); }
    if (h->h_sp1) e = hipHostFree(h->h_sp1);
    if (h->h_sf_batch) e = hipHostFree(h->h_sf_batch);
    for (int i = 0; i < dabphy_handle::N_DESC; i++) { if (h->ev_chain_beg[i]) e = hipEventDestroy(h->ev_chain_beg[i]); if (h->ev_chain_end[i]) e = hipEventDestroy(h->ev_chain_end[i]); }
    if (h->stream) { e = hipStreamSynchronize(h->stream); e = hipStreamDestroy(h->stream); }
    for (void* p : h->owned) e = hipFree(p);
    for (int i = 0; i < dabphy_handle::ST_COUNT; i++) { if (h->ev_beg[i]) e = hipEventDestroy(h->ev_beg[i]); if (h->ev_end[i]) e = hipEventDestroy(h->ev_end[i]); }
    DevBuf* more[] = {&h->s_raw, &h->s_raw2[0], &h->s_raw2[1], &h->s_null, &h->s_iq_own, &h->s_desc2[0], &h->s_desc2[1], &h->s_desc2[2], &h->s_soft, &h->s_cir2[0], &h->s_cir2[1], &h->s_cir2[2], &h->s_con, &h->s_mag, &h->s_snr, &h->s_fib, &h->s_ok, &h->rs_first, &h->rs_result};
    for (DevBuf* b : more) if (b->p) e = hipFree(b->p);
    { DevBuf* sfb[] = {&h->sf_events, &h->sf_count, &h->sf_bytes, &h->sf_stats, &h->sf_gf, &h->sf_accept, &h->sf_run, &h->sf_batch}; for (DevBuf* b : sfb) if (b->p) e = hipFree(b->p); }
    { DevBuf* tb[] = {&h->s_hist, &h->tii_rot, &h->tii_rank, &h->tii_pat, &h->tii_err, &h->tii_likely, &h->tii_state, &h->tii_events, &h->tii_nev, &h->tii_ovf}; for (DevBuf* b : tb) if (b->p) e = hipFree(b->p); }
    for (auto& c : h->classes) free_class(c);
    DevBuf* bufs[] = {&h->iq, &h->soft, &h->con, &h->prs_mag, &h->snr, &h->desc, &h->in8, &h->map, &h->vsym, &h->vdec, &h->vout, &h->ok, &h->fsym, &h->fdec};
    for (DevBuf* b : bufs) if (b->p) e = hipFree(b->p);
    (void)e;
    delete h;
}

const char* dabphy_last_error(const dabphy_handle* h) { return h ? h->err.c_str() : "null handle"; }
const char* dabphy_device_name(const dabphy_handle* h) { return h ? h->devname : ""; }

int dabphy_get_config(const dabphy_handle* h, dabphy_config_r3* out)       // (objects of rounds 1-3: the unsized layout)
{
    if (!h || !out) return DABPHY_ERR_INVALID;
    memcpy(out, &h->cfg.n_ensembles, sizeof *out);
    return DABPHY_OK;
}

int dabphy_get_config_v2(const dabphy_handle* h, dabphy_config* out)
{
    DeviceBind dev_(h);
    if (!h || !out || out->struct_size < sizeof(uint32_t) || out->struct_size % 4) return DABPHY_ERR_INVALID;
    const uint32_t n = std::min<uint32_t>(out->struct_size, (uint32_t)sizeof(dabphy_config));
    memcpy(out, &h->cfg, n);
    out->struct_size = n;
    return DABPHY_OK;
}


int dabphy_set_options(dabphy_handle* h, int32_t fft_placement, int32_t freqsync_method, int32_t disable_coarse, int32_t* restarted)
{
    DeviceBind dev_(h);
    if (!h || fft_placement < 0 || fft_placement > 2 || freqsync_method < 0 || freqsync_method > 2) return DABPHY_ERR_INVALID;
    const bool need_reset = (h->cfg.disable_coarse != 0) != (disable_coarse != 0);      // ofdm-processor.cpp:521
    if (h->s_desc2[0].p) { int r = resolve_all_chains(h); if (r) return r; }            // frames synchronised ahead keep the options they were queued with
    h->cfg.fft_placement = fft_placement; h->cfg.freqsync_method = freqsync_method; h->cfg.disable_coarse = disable_coarse != 0;
    if (restarted) *restarted = need_reset ? 1 : 0;
    // :523-528 -> OFDMProcessor::restart (:115-132): correctors, phase, sLevel and the sync state start over; the decoders (FIC
    // counter, SNR filter, time de-interleaver, superframe windows) are not touched by this path
    return (need_reset && h->s_iq) ? reset_synchroniser(h, false) : DABPHY_OK;
}

int dabphy_protection_fic(dabphy_protection* p) { return p ? protection_fic(p) : DABPHY_ERR_INVALID; }
int dabphy_protection_eep(dabphy_protection* p, int bitrate, int profile_b, int level)
{
    if (!p) return DABPHY_ERR_INVALID;
    return protection_eep(p, bitrate, profile_b, level) ? DABPHY_ERR_INVALID : DABPHY_OK;
}
int dabphy_protection_uep(dabphy_protection* p, int bitrate, int level) { return p ? protection_uep(p, bitrate, level) : DABPHY_ERR_INVALID; }
int dabphy_uep_table_entry(int table_index, int* size_cu, int* level, int* bitrate)
{
    if (!size_cu || !level || !bitrate) return DABPHY_ERR_INVALID;
    return uep_table_entry(table_index, size_cu, level, bitrate) ? DABPHY_ERR_INVALID : DABPHY_OK;
}
int dabphy_protection_input_bits(const dabphy_protection* p) { return p ? protection_input_bits(p) : DABPHY_ERR_INVALID; }

int dabphy_demod_frames(dabphy_handle* h, const float* frames, uint32_t n_frames, int8_t* soft, float* constellation, float* snr)
{
    DeviceBind dev_(h);
    if (!h || !frames || !soft || n_frames == 0) return DABPHY_ERR_INVALID;
    const size_t per = (size_t)T_U + 75 * (size_t)T_S;
    int r;
    if ((r = ensure(h, h->iq, per * n_frames * sizeof(cf32)))) return r;
    if ((r = ensure(h, h->soft, (size_t)n_frames * SOFT_PER_FRAME))) return r;
    if ((r = ensure(h, h->desc, n_frames * sizeof(FrameDesc)))) return r;
    if ((r = ensure(h, h->prs_mag, (size_t)n_frames * T_U * sizeof(float)))) return r;
    if ((r = ensure(h, h->snr, n_frames * sizeof(float)))) return r;
    if (constellation && (r = ensure(h, h->con, (size_t)n_frames * 1200 * sizeof(cf32)))) return r;
    std::vector<FrameDesc> d(n_frames);
    for (uint32_t f = 0; f < n_frames; f++) {
        memset(&d[f], 0, sizeof(FrameDesc));
        d[f].pos = (int64_t)(per * f); d[f].frame_no = f; d[f].valid = 1;
    }
    HIPCHK(h, hipMemcpyAsync(h->iq.p, frames, per * n_frames * sizeof(cf32), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->desc.p, d.data(), n_frames * sizeof(FrameDesc), hipMemcpyHostToDevice, h->stream));
    DemodArgs a{};
    a.tab = h->tab; a.iq = h->iq.as<cf32>(); a.iq_stride = 0; a.ring = (int64_t)(per * n_frames);
    a.desc = h->desc.as<FrameDesc>(); a.n_frames = (int)n_frames; a.chunk_len = h->cfg.demod_chunk; a.mix = 0;
    a.soft = h->soft.as<int8_t>(); a.soft_ring = (int)n_frames;
    a.con = constellation ? h->con.as<cf32>() : nullptr; a.prs_mag = h->prs_mag.as<float>();
    launch_demod(a, 1, h->stream);
    SnrArgs s{}; s.state = h->d_dec; s.desc = a.desc; s.n_ens = 1; s.n_frames = (int)n_frames; s.prs_mag = a.prs_mag; s.snr_out = h->snr.as<float>();
    launch_snr(s, h->stream);
    HIPCHK(h, hipMemcpyAsync(soft, h->soft.p, (size_t)n_frames * SOFT_PER_FRAME, hipMemcpyDeviceToHost, h->stream));
    if (constellation) HIPCHK(h, hipMemcpyAsync(constellation, h->con.p, (size_t)n_frames * 1200 * sizeof(cf32), hipMemcpyDeviceToHost, h->stream));
    if (snr) HIPCHK(h, hipMemcpyAsync(snr, h->snr.p, n_frames * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    return sync(h);
}

static int run_lin_decode(dabphy_handle* h, const int8_t* in, size_t in_stride, const int16_t* d_map, int nbits, uint32_t n_cw,
                          int dedisperse, uint8_t* out)
{
    int r;
    if ((r = ensure(h, h->in8, in_stride * n_cw))) return r;
    VitClass c{};
    if ((r = prepare_class(h, c, nbits, (int)n_cw, dedisperse))) return r;
    HIPCHK(h, hipMemcpyAsync(h->in8.p, in, in_stride * n_cw, hipMemcpyHostToDevice, h->stream));
    if (sp_single_ok(h, n_cw, c.nsteps)) {
        // a small call (the per-frame seams of INTEGRATION.md level 2: a few code words): one wavefront per code word
        FusedClass fc{}; fc.map = d_map; fc.out = c.out; fc.nsteps = c.nsteps; fc.nbits = nbits; fc.n_cw = (int32_t)n_cw; fc.n_pairs = 1; fc.kind = 2; fc.dedisperse = dedisperse;
        FusedArgs a{}; a.n_ens = 1; a.n_frames = 1; a.lin_in = h->in8.as<int8_t>(); a.lin_stride = in_stride;
        if ((r = sp_single_prepare(h, fc, a, h->stream))) return r;
        launch_sp(a, h->sp1_two, sp_variant_for(c.nsteps), h->stream);
        HIPCHK(h, hipMemcpyAsync(out, c.out, (size_t)n_cw * (nbits / 8), hipMemcpyDeviceToHost, h->stream));
        return sync(h);
    }
    LinGatherArgs g{}; g.in = h->in8.as<int8_t>(); g.in_stride = in_stride; g.map = d_map; g.c = c;
    launch_lin_gather(g, h->stream);
    VitArgs v{}; v.c = c; v.prbs_words = h->d_prbs_words;
    launch_viterbi(v, h->stream);
    HIPCHK(h, hipMemcpyAsync(out, c.out, (size_t)n_cw * (nbits / 8), hipMemcpyDeviceToHost, h->stream));
    return sync(h);
}

int dabphy_viterbi_batch(dabphy_handle* h, const int8_t* in, uint32_t nbits, uint32_t n_codewords, uint8_t* out)
{
    DeviceBind dev_(h);
    if (!h || !in || !out || n_codewords == 0 || nbits == 0 || nbits % 32 || nbits > PRBS_MAX_BITS) return DABPHY_ERR_INVALID;
    return run_lin_decode(h, in, (size_t)4 * (nbits + 6), nullptr, (int)nbits, n_codewords, 0, out);
}

int dabphy_msc_deconvolve(dabphy_handle* h, const dabphy_protection* prot, const int8_t* in, uint32_t n_codewords, uint8_t* out)
{
    DeviceBind dev_(h);
    if (!h || !prot || !in || !out || n_codewords == 0 || !protection_valid(prot) || prot->nbits > PRBS_MAX_BITS) return DABPHY_ERR_INVALID;
    const std::vector<int16_t> m = depuncture_map(prot);
    int r;
    if ((r = ensure(h, h->map, m.size() * sizeof(int16_t)))) return r;
    HIPCHK(h, hipMemcpyAsync(h->map.p, m.data(), m.size() * sizeof(int16_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));     // m goes out of scope on return paths below only after the copy
    return run_lin_decode(h, in, (size_t)protection_input_bits(prot), h->map.as<int16_t>(), prot->nbits, n_codewords, 1, out);
}

int dabphy_fic_decode(dabphy_handle* h, const int8_t* soft, uint32_t n_frames, uint8_t* fib, uint8_t* crc_ok, int32_t* ratio_percent)
{
    DeviceBind dev_(h);
    if (!h || !soft || !fib || !crc_ok || n_frames == 0) return DABPHY_ERR_INVALID;
    int r;
    if ((r = ensure(h, h->in8, (size_t)n_frames * 9216))) return r;
    if ((r = ensure(h, h->desc, n_frames * sizeof(FrameDesc)))) return r;
    if ((r = ensure(h, h->ok, (size_t)n_frames * 12))) return r;
    VitClass c{};
    if ((r = prepare_class(h, c, 768, (int)n_frames * 4, 1))) return r;
    std::vector<FrameDesc> d(n_frames);
    for (uint32_t f = 0; f < n_frames; f++) { memset(&d[f], 0, sizeof(FrameDesc)); d[f].frame_no = f; d[f].valid = 1; }
    HIPCHK(h, hipMemcpyAsync(h->in8.p, soft, (size_t)n_frames * 9216, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->desc.p, d.data(), n_frames * sizeof(FrameDesc), hipMemcpyHostToDevice, h->stream));
    FicGatherArgs g{}; g.soft = h->in8.as<int8_t>(); g.soft_ring = (int)n_frames; g.frame_stride = 9216; g.desc = h->desc.as<FrameDesc>();
    g.n_ens = 1; g.n_frames = (int)n_frames; g.map = h->d_fic_map; g.c = c;
    if (sp_single_ok(h, (uint64_t)n_frames * 4, c.nsteps)) {
        // (FicHandler::processFicBlock bound per frame: four code words a call)
        FusedClass fc{}; fc.map = h->d_fic_map; fc.out = c.out; fc.nsteps = c.nsteps; fc.nbits = 768; fc.n_cw = (int32_t)(n_frames * 4); fc.n_pairs = 1; fc.kind = 1; fc.dedisperse = 1;
        FusedArgs a{}; a.soft = g.soft; a.ens_stride = (size_t)n_frames * 9216; a.soft_ring = (int)n_frames; a.n_ens = 1; a.n_frames = (int)n_frames; a.desc = g.desc; a.fic_frame_stride = 9216;
        if ((r = sp_single_prepare(h, fc, a, h->stream))) return r;
        launch_sp(a, h->sp1_two, sp_variant_for(c.nsteps), h->stream);
    } else {
        launch_fic_gather(g, h->stream);
        VitArgs v{}; v.c = c; v.prbs_words = h->d_prbs_words;
        launch_viterbi(v, h->stream);
    }
    CrcArgs k{}; k.fib = c.out; k.ok = h->ok.as<uint8_t>(); k.state = h->d_dec; k.desc = g.desc; k.n_ens = 1; k.n_frames = (int)n_frames; k.disable_coarse = 1;   // (no synchroniser behind this seam)
    launch_fib_crc(k, h->stream);
    launch_fic_ratio(k, h->stream);
    HIPCHK(h, hipMemcpyAsync(fib, c.out, (size_t)n_frames * 384, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(crc_ok, h->ok.p, (size_t)n_frames * 12, hipMemcpyDeviceToHost, h->stream));
    DecState st;
    HIPCHK(h, hipMemcpyAsync(&st, h->d_dec, sizeof st, hipMemcpyDeviceToHost, h->stream));
    if ((r = sync(h))) return r;
    if (ratio_percent) *ratio_percent = st.fic_ratio * 10;
    return DABPHY_OK;
}

namespace {
int check_subchannels(dabphy_handle* h, const dabphy_subchannel* list, uint32_t n)
{
    if (n > 64) { h->err = "more than 64 sub-channels in one ensemble"; return DABPHY_ERR_INVALID; }      // SubChId is a 6-bit field (fib-processor.cpp:331)
    for (uint32_t i = 0; i < n; i++) {
        const dabphy_subchannel& s = list[i];
        if (!protection_valid(&s.prot) || s.prot.nbits > PRBS_MAX_BITS || s.start_cu < 0 || s.size_cu <= 0 || s.start_cu + s.size_cu > 864 ||
            protection_input_bits(&s.prot) > s.size_cu * 64) { h->err = "invalid sub-channel " + std::to_string(i); return DABPHY_ERR_INVALID; }
    }
    return DABPHY_OK;
}
}

// MscHandler::addSubchannel / removeSubchannel / stopProcessing (msc-handler.cpp:61-127) for EVERY ensemble of the handle at once
int dabphy_set_subchannels(dabphy_handle* h, const dabphy_subchannel* list, uint32_t n)
{
    DeviceBind dev_(h);
    if (!h || (n && !list)) return DABPHY_ERR_INVALID;
    int r;
    if ((r = check_subchannels(h, list, n))) return r;
    for (auto& l : h->subch_next) l.assign(list, list + n);
    h->subch_dirty = true;
    return apply_subchannels(h);
}

// ... and for ONE ensemble of the batch, as each receiver of the reference does for itself (radio-receiver.cpp:120-137: playSingleProgramme /
// addServiceToDecode / removeServiceToDecode drive its own MscHandler).  Takes effect with the next dabphy_process, which rebuilds the
// protection classes once for all the ensembles that changed.
int dabphy_set_subchannels_ensemble(dabphy_handle* h, uint32_t ensemble, const dabphy_subchannel* list, uint32_t n)
{
    DeviceBind dev_(h);
    if (!h || (n && !list) || ensemble >= h->cfg.n_ensembles) return DABPHY_ERR_INVALID;
    int r;
    if ((r = check_subchannels(h, list, n))) return r;
    h->subch_next[ensemble].assign(list, list + n);
    h->subch_dirty = true;
    return DABPHY_OK;
}

int dabphy_get_subchannel_count(dabphy_handle* h, uint32_t ensemble, uint32_t* n)
{
    if (!h || !n || ensemble >= h->cfg.n_ensembles) return DABPHY_ERR_INVALID;
    *n = (uint32_t)h->subch_next[ensemble].size();
    return DABPHY_OK;
}

int upload_pairs(dabphy_handle* h, dabphy_handle::MscClass& c)
{
    int r;
    if ((r = ensure(h, c.pair_tab, c.pairs.size() * sizeof(MscPair)))) return r;
    HIPCHK(h, hipMemcpy(c.pair_tab.p, c.pairs.data(), c.pairs.size() * sizeof(MscPair), hipMemcpyHostToDevice));
    c.cif0_pending = false;
    for (const MscPair& p : c.pairs) if (p.cif0 < 0) c.cif0_pending = true;
    return DABPHY_OK;
}

// The classes of the batch from the per-ensemble lists.  A service that stays -- same ensemble, same SubChId, same place and protection
// -- keeps what it carries from batch to batch: the CIF count its time de-interleaver started at and its SuperframeFilter window
// (the reference's addSubchannel / removeSubchannel touch no other stream, msc-handler.cpp:61-127); a new one starts empty.
int apply_subchannels(dabphy_handle* h)
{
    if (!h->subch_dirty) return DABPHY_OK;
    const uint32_t B = h->cfg.n_ensembles;
    int r;
    HIPCHK(h, hipStreamSynchronize(h->stream));              // nothing queued still reads the classes that are about to go
    if ((r = drain_wait(h))) return r;                       // (nor a bulk MSC drain in flight)
    HIPCHK(h, hipStreamSynchronize(h->aux_stream));
    std::vector<dabphy_handle::MscClass> old;
    old.swap(h->classes);
    const std::vector<std::vector<dabphy_handle::PairRef>> old_where = h->where;
    const std::vector<std::vector<dabphy_subchannel>> old_lists = h->subch_e;
    h->fplan.valid = false; h->fplan.launched = false; h->buf_gen++;          // the plan names the classes' buffers
    h->last_frames = 0; h->last_desc = nullptr; h->sf_stats_ready = false; h->h_sf_stats_valid = false;   // the class outputs of the last batch go with the classes
    h->subch_e = h->subch_next;
    h->subch_dirty = false;
    auto fail = [&](int code) {                             // nothing half-built stays: the handle then decodes no sub-channel at all
        for (auto& c : old) free_class(c);
        for (auto& c : h->classes) free_class(c);
        h->classes.clear();
        // (the rejected request goes too: dabphy_get_subchannel_count then reports what is in effect -- nothing -- instead of sub-channels
        // that are never decoded, and the next dabphy_process has nothing stale to apply)
        for (uint32_t b = 0; b < B; b++) { h->subch_e[b].clear(); h->where[b].clear(); h->subch_next[b].clear(); }
        return code;
    };
    struct Carry { int old_cls, old_pair; };
    std::vector<std::vector<Carry>> carry;                   // per new class, per pair: where its state lies now (-1: nowhere)
    for (uint32_t b = 0; b < B; b++) {
        h->where[b].assign(h->subch_e[b].size(), dabphy_handle::PairRef{});
        std::vector<char> taken(old_lists[b].size(), 0);
        for (size_t i = 0; i < h->subch_e[b].size(); i++) {
            const dabphy_subchannel& sc = h->subch_e[b][i];
            int ci = -1;
            for (size_t k = 0; k < h->classes.size(); k++) if (!memcmp(&h->classes[k].prot, &sc.prot, sizeof(dabphy_protection))) { ci = (int)k; break; }
            if (ci < 0) {
                if (h->classes.size() >= 255) { h->err = "more than 255 protection classes"; return fail(DABPHY_ERR_INVALID); }
                ci = (int)h->classes.size(); h->classes.emplace_back(); h->classes.back().prot = sc.prot; carry.emplace_back();
            }
            auto& c = h->classes[ci];
            MscPair p{}; p.ens = (int32_t)b; p.start_bit = sc.start_cu * 64; p.cif0 = -1; p.idx = (int32_t)i;
            Carry from{-1, -1};
            for (size_t k = 0; k < old_lists[b].size(); k++) {
                const dabphy_subchannel& o = old_lists[b][k];
                if (taken[k] || o.subch_id != sc.subch_id || o.start_cu != sc.start_cu || o.size_cu != sc.size_cu || memcmp(&o.prot, &sc.prot, sizeof(dabphy_protection))) continue;
                taken[k] = 1;
                const dabphy_handle::PairRef w = old_where[b][k];
                from = Carry{w.cls, w.pair};
                p.cif0 = old[w.cls].pairs[w.pair].cif0;
                break;
            }
            h->where[b][i] = dabphy_handle::PairRef{ci, (int)c.pairs.size()};
            c.pairs.push_back(p); c.subch_id.push_back(sc.subch_id); c.start_cu.push_back(sc.start_cu);
            carry[ci].push_back(from);
        }
    }
    for (size_t ci = 0; ci < h->classes.size(); ci++) {
        auto& c = h->classes[ci];
        // the per-profile tables: taken over from the old class of the same profile when there is one
        dabphy_handle::MscClass* prev = nullptr;
        for (auto& o : old) if (o.map.p && !memcmp(&o.prot, &c.prot, sizeof(dabphy_protection))) { prev = &o; break; }
        if (prev) {
            std::swap(c.map, prev->map); std::swap(c.tiles, prev->tiles);
            for (int v = 0; v < FUSED_VARIANTS; v++) { std::swap(c.steps[v], prev->steps[v]); c.n_windows[v] = prev->n_windows[v]; }
        } else {
            const std::vector<int16_t> m = depuncture_map(&c.prot);
            if ((r = ensure(h, c.map, m.size() * sizeof(int16_t)))) return fail(r);
            if (hipMemcpy(c.map.p, m.data(), m.size() * sizeof(int16_t), hipMemcpyHostToDevice) != hipSuccess) { h->err = "hipMemcpy(depuncturing map) failed"; return fail(DABPHY_ERR_HIP); }
            // step tiles of the MSC gather kernel (56 trellis steps each): source byte range of every tile
            std::vector<int32_t> tl;
            const int nsteps = c.prot.nbits + 6;
            for (int s0 = 0; s0 < nsteps; s0 += 56) {
                const int s1 = std::min(nsteps, s0 + 56);
                int lo = -1, hi = -1;
                for (int v = 4 * s0; v < 4 * s1; v++) if (m[v] >= 0) { if (lo < 0) lo = m[v]; hi = m[v]; }
                if (lo < 0) { tl.push_back(0); tl.push_back(0); continue; }
                const int lo_al = lo & ~3;
                tl.push_back(lo_al); tl.push_back((hi - lo_al) / 4 + 1);
            }
            if ((r = ensure(h, c.tiles, tl.size() * sizeof(int32_t)))) return fail(r);
            if (hipMemcpy(c.tiles.p, tl.data(), tl.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) { h->err = "hipMemcpy(gather tiles) failed"; return fail(DABPHY_ERR_HIP); }
            // fused decode: what every trellis step reads, in terms of the wave's window ring, for each build of the kernel (dabphy_fused.hip)
            if ((r = fused_class_tables(h, c.prot, false, c.steps, c.n_windows))) return fail(r);
        }
        if ((r = upload_pairs(h, c))) return fail(r);
        if (c.dabplus_rate()) {
            // SuperframeFilter windows: empty for new pairs, moved for those that stay (runs of neighbours in one copy)
            const size_t stride = c.sf_stride(), n = c.pairs.size();
            if ((r = ensure(h, c.sf_state, stride * n))) return fail(r);
            if (hipMemsetAsync(c.sf_state.p, 0, c.sf_state.cap, h->stream) != hipSuccess) { h->err = "hipMemsetAsync(superframe windows) failed"; return fail(DABPHY_ERR_HIP); }
            for (size_t p = 0; p < n;) {
                const Carry f = carry[ci][p];
                if (f.old_cls < 0 || !old[f.old_cls].sf_state.p) { p++; continue; }
                size_t e = p + 1;
                while (e < n && carry[ci][e].old_cls == f.old_cls && carry[ci][e].old_pair == f.old_pair + (int)(e - p)) e++;
                if (hipMemcpyAsync(c.sf_state.as<uint8_t>() + p * stride, old[f.old_cls].sf_state.as<uint8_t>() + (size_t)f.old_pair * stride, (e - p) * stride,
                                   hipMemcpyDeviceToDevice, h->stream) != hipSuccess) { h->err = "hipMemcpyAsync(superframe windows) failed"; return fail(DABPHY_ERR_HIP); }
                p = e;
            }
        }
    }
    if (hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "hipStreamSynchronize failed (sub-channel change)"; return fail(DABPHY_ERR_HIP); }
    for (auto& c : old) free_class(c);
    return DABPHY_OK;
}

int dabphy_selftest_unit_twiddle(dabphy_handle* h, uint64_t* counts)
{
    DeviceBind dev_(h);
    if (!h || !counts) return DABPHY_ERR_INVALID;
    unsigned long long* d = nullptr;
    HIPCHK(h, hipMalloc((void**)&d, 2 * sizeof *d));
    HIPCHK(h, hipMemsetAsync(d, 0, 2 * sizeof *d, h->stream));
    launch_selftest_unit_twiddle(d, h->stream);
    unsigned long long host[2];
    HIPCHK(h, hipMemcpyAsync(host, d, sizeof host, hipMemcpyDeviceToHost, h->stream));
    const int r = sync(h);
    (void)hipFree(d);
    for (int i = 0; i < 2; i++) counts[i] = host[i];
    return r;
}

int dabphy_selftest_pair_exchange(dabphy_handle* h, uint64_t* counts)
{
    DeviceBind dev_(h);
    if (!h || !counts) return DABPHY_ERR_INVALID;
    unsigned* d = nullptr;
    HIPCHK(h, hipMalloc((void**)&d, 2 * sizeof *d));
    HIPCHK(h, hipMemsetAsync(d, 0, 2 * sizeof *d, h->stream));
    launch_selftest_pair_exchange(d, h->stream);
    launch_selftest_half_exchange(d, h->stream);                 // ... and the forms of the two-code-words-per-wavefront kernel (same counters)
    unsigned host[2];
    HIPCHK(h, hipMemcpyAsync(host, d, sizeof host, hipMemcpyDeviceToHost, h->stream));
    const int r = sync(h);
    (void)hipFree(d);
    for (int i = 0; i < 2; i++) counts[i] = host[i];
    return r;
}

int dabphy_selftest_div127(dabphy_handle* h, uint64_t* counts)
{
    DeviceBind dev_(h);
    if (!h || !counts) return DABPHY_ERR_INVALID;
    unsigned long long* d = nullptr;
    HIPCHK(h, hipMalloc((void**)&d, 3 * sizeof *d));
    HIPCHK(h, hipMemsetAsync(d, 0, 3 * sizeof *d, h->stream));
    launch_selftest_div127(d, h->stream);
    unsigned long long host[3];
    HIPCHK(h, hipMemcpyAsync(host, d, sizeof host, hipMemcpyDeviceToHost, h->stream));
    const int r = sync(h);
    (void)hipFree(d);
    for (int i = 0; i < 3; i++) counts[i] = host[i];
    return r;
}

int dabphy_time_demod(dabphy_handle* h, const float* frames, uint32_t n_src, uint32_t n_ens, uint32_t n_frames,
                      int32_t mix, int32_t f_hz, uint32_t iters, float* ms)
{
    DeviceBind dev_(h);
    if (!h || !frames || !ms || n_src == 0 || n_ens == 0 || n_frames == 0 || iters == 0) return DABPHY_ERR_INVALID;
    const size_t per = (size_t)T_U + 75 * (size_t)T_S;
    const size_t total = (size_t)n_ens * n_frames;
    int r;
    if ((r = ensure(h, h->iq, per * total * sizeof(cf32)))) return r;
    if ((r = ensure(h, h->soft, total * SOFT_PER_FRAME))) return r;
    if ((r = ensure(h, h->desc, total * sizeof(FrameDesc)))) return r;
    for (size_t i = 0; i < total; i++)
        HIPCHK(h, hipMemcpyAsync(h->iq.as<cf32>() + per * i, frames + 2 * per * (i % n_src), per * sizeof(cf32), hipMemcpyHostToDevice, h->stream));
    std::vector<FrameDesc> d(total);
    for (uint32_t b = 0; b < n_ens; b++)
        for (uint32_t f = 0; f < n_frames; f++) {
            FrameDesc& x = d[(size_t)b * n_frames + f];
            memset(&x, 0, sizeof x);
            x.pos = (int64_t)(per * f); x.frame_no = f; x.valid = 1; x.f_prs = x.f_sym = f_hz; x.L0 = 12345; x.L1 = 54321;
            const HostTables& T = host_tables();
            for (int i = 0; i < T.n_osc_unsafe; i++) osc_hazard_entry(x.osc_hazard, T.osc_unsafe[i], x.start_index, x.L0, x.f_prs, x.L1, x.f_sym);
        }
    HIPCHK(h, hipMemcpyAsync(h->desc.p, d.data(), total * sizeof(FrameDesc), hipMemcpyHostToDevice, h->stream));
    DemodArgs a{};
    a.tab = h->tab; a.iq = h->iq.as<cf32>(); a.iq_stride = per * n_frames; a.ring = (int64_t)(per * n_frames);
    a.desc = h->desc.as<FrameDesc>(); a.n_frames = (int)n_frames; a.chunk_len = h->cfg.demod_chunk; a.mix = mix;
    a.soft = h->soft.as<int8_t>(); a.soft_ring = (int)n_frames; a.con = nullptr; a.prs_mag = nullptr;
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
    launch_demod(a, (int)n_ens, h->stream);                       // warm-up
    HIPCHK(h, hipEventRecord(e0, h->stream));
    for (uint32_t i = 0; i < iters; i++) launch_demod(a, (int)n_ens, h->stream);
    HIPCHK(h, hipEventRecord(e1, h->stream));
    HIPCHK(h, hipEventSynchronize(e1));
    float t = 0; HIPCHK(h, hipEventElapsedTime(&t, e0, e1));
    *ms = t / iters;
    HIPCHK(h, hipEventDestroy(e0)); HIPCHK(h, hipEventDestroy(e1));
    return sync(h);
}

int dabphy_time_copy(dabphy_handle* h, uint64_t bytes, uint32_t blocks_per_cu, uint32_t iters, float* gbytes_per_s)
{
    DeviceBind dev_(h);
    if (!h || !gbytes_per_s || bytes < 4096 || iters == 0) return DABPHY_ERR_INVALID;
    hipDeviceProp_t p;
    HIPCHK(h, hipGetDeviceProperties(&p, h->cfg.device));
    const int blocks = p.multiProcessorCount * (int)(blocks_per_cu ? blocks_per_cu : 4);
    void *src = nullptr, *dst = nullptr;
    if (hipMalloc(&src, bytes) != hipSuccess) { h->err = "hipMalloc failed (copy source)"; return DABPHY_ERR_NOMEM; }
    if (hipMalloc(&dst, bytes) != hipSuccess) { (void)hipFree(src); h->err = "hipMalloc failed (copy destination)"; return DABPHY_ERR_NOMEM; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMemsetAsync(src, 1, bytes, h->stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float t = 0;
    if (e == hipSuccess) {
        for (int w = 0; w < 3; w++) launch_copy_f4(src, dst, bytes / 16, blocks, h->stream);      // clocks up, pages touched
        e = hipEventRecord(e0, h->stream);
        for (uint32_t i = 0; i < iters; i++) launch_copy_f4(src, dst, bytes / 16, blocks, h->stream);
        if (e == hipSuccess) e = hipEventRecord(e1, h->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&t, e0, e1);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(src); (void)hipFree(dst);
    if (e != hipSuccess) { h->err = std::string("dabphy_time_copy: ") + hipGetErrorString(e); return DABPHY_ERR_HIP; }
    *gbytes_per_s = (float)(2.0 * (double)(bytes / 16 * 16) * iters / ((double)t * 1e-3) / 1e9);
    return sync(h);
}

int dabphy_time_viterbi(dabphy_handle* h, uint32_t nbits, uint32_t n_codewords, uint32_t iters, float* ms_gather, float* ms_decode)
{
    DeviceBind dev_(h);
    if (!h || !ms_gather || !ms_decode || nbits == 0 || nbits % 32 || nbits > PRBS_MAX_BITS || n_codewords == 0 || iters == 0) return DABPHY_ERR_INVALID;
    const size_t stride = (size_t)4 * (nbits + 6);
    int r;
    if ((r = ensure(h, h->in8, stride * n_codewords))) return r;
    VitClass c{};
    if ((r = prepare_class(h, c, (int)nbits, (int)n_codewords, 1))) return r;
    {   // pseudo-random soft bits (content only drives the data-dependent clock, not the instruction count)
        std::vector<int8_t> host(stride * n_codewords);
        uint32_t x = 12345u;
        for (auto& v : host) { x = x * 1664525u + 1013904223u; v = (int8_t)(x >> 24); }
        HIPCHK(h, hipMemcpyAsync(h->in8.p, host.data(), host.size(), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    LinGatherArgs g{}; g.in = h->in8.as<int8_t>(); g.in_stride = stride; g.map = nullptr; g.c = c;
    VitArgs v{}; v.c = c; v.prbs_words = h->d_prbs_words;
    hipEvent_t e0, e1, e2;
    HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1)); HIPCHK(h, hipEventCreate(&e2));
    launch_lin_gather(g, h->stream); launch_viterbi(v, h->stream);
    float tg = 0, tv = 0;
    for (uint32_t i = 0; i < iters; i++) {
        HIPCHK(h, hipEventRecord(e0, h->stream));
        launch_lin_gather(g, h->stream);
        HIPCHK(h, hipEventRecord(e1, h->stream));
        launch_viterbi(v, h->stream);
        HIPCHK(h, hipEventRecord(e2, h->stream));
        HIPCHK(h, hipEventSynchronize(e2));
        float a = 0, b = 0; HIPCHK(h, hipEventElapsedTime(&a, e0, e1)); HIPCHK(h, hipEventElapsedTime(&b, e1, e2));
        tg += a; tv += b;
    }
    *ms_gather = tg / iters; *ms_decode = tv / iters;
    HIPCHK(h, hipEventDestroy(e0)); HIPCHK(h, hipEventDestroy(e1)); HIPCHK(h, hipEventDestroy(e2));
    return sync(h);
}

int dabphy_test_traceback_split(dabphy_handle* h, int32_t on)
{
    if (!h) return DABPHY_ERR_INVALID;
    h->tb_split = (on & 1) != 0;                             // (takes effect with the next batch's launch plan)
    // (bit 1: also launch the 24-register walker waves, k_traceback_fused.  As measured in round 6 they read STALE decisions from their
    // XCD's L2 on the second batch -- the scratch addresses repeat from batch to batch and an agent-scope acquire does not evict L2 --:
    // wrong bytes.  Kept for the experiment's record only; without the bit the forward waves walk everything back themselves)
    h->tb_no_walkers = (on & 2) == 0;
    return DABPHY_OK;
}

int dabphy_time_fused_msc(dabphy_handle* h, uint32_t iters, float* ms)
{
    DeviceBind dev_(h);
    if (!h || !ms || iters == 0) return DABPHY_ERR_INVALID;
    // the launch as the last batch queued it -- only while every buffer it names is still the one it named (sub-channel changes,
    // resets and reallocations bump buf_gen)
    const auto& P = h->fplan;
    if (!P.valid || !P.launched || P.buf_gen != h->buf_gen || P.args.n_work == 0) { h->err = "no batch has been decoded by the fused kernel with the present buffers"; return DABPHY_ERR_STATE; }
    HIPCHK(h, hipDeviceSynchronize());                       // alone on the device: nothing of the pipeline beside it
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHK(h, hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); h->err = "hipEventCreate failed"; return DABPHY_ERR_HIP; }
    const FusedSplit sp{h->tb_no_walkers ? nullptr : h->tb_stream, h->ev_tb_fork, h->ev_tb_join};
    auto again = [&]() { if (P.use_sp) launch_sp(P.args, P.sp_two, P.sp_variant, h->stream); else launch_viterbi_fused(P.args, P.variant, P.n_slots, h->stream, P.args.done ? &sp : nullptr); };
    again();                                                                  // (same inputs, same outputs: the launch is idempotent)
    hipError_t e = hipEventRecord(e0, h->stream);
    // (split traceback: the launch forks a second stream off and joins it through two events the NEXT launch records again -- the stream
    // is drained between launches so that no wait is pending on an event when it is re-recorded; ~20 us per launch in the figure)
    for (uint32_t i = 0; i < iters; i++) { again(); if (P.args.done && e == hipSuccess) e = hipStreamSynchronize(h->stream); }
    if (e == hipSuccess) e = hipEventRecord(e1, h->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float t = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (e != hipSuccess) { h->err = std::string("dabphy_time_fused_msc: ") + hipGetErrorString(e); return DABPHY_ERR_HIP; }
    *ms = t / iters;
    return sync(h);
}

} // extern "C"
