// welle.io_amd/csrc/dabphy_api.hip -- C ABI of libdabphy_hip.so (include/dabphy.h): handle, device tables,
// buffer management and kernel sequencing.  No arithmetic of the hot path happens on the host.
#include "dabphy_kernels.h"
#include "dabphy_host.h"
#include "osc_exact.h"
#include <string>
#include <vector>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <algorithm>
#include <chrono>

using namespace dabphy;

namespace {

// Environment variables are read only by builds made with -DDABPHY_EXPERIMENTS (timing / debugging builds and the GPU-less test
// build): the product library is configured through dabphy_config alone.
inline bool debug_env(const char* name)
{
#ifdef DABPHY_EXPERIMENTS
    return getenv(name) != nullptr;
#else
    (void)name; return false;
#endif
}

struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

} // namespace

struct dabphy_handle {
    dabphy_config cfg{};
    hipStream_t stream = nullptr;
    std::string err;
    char devname[256] = {0};
    // constant tables in HBM
    cf32 *d_tw = nullptr, *d_ref = nullptr, *d_nco = nullptr;
    int16_t* d_bin2soft = nullptr; uint32_t* d_prbs_words = nullptr; int16_t* d_fic_map = nullptr;
    int32_t* d_osc_unsafe = nullptr; unsigned long long* d_osc_stats = nullptr;   // osc_exact.h: unsafe table entries; symbols mixed unchecked / checked
    Tables tab{};
    // grow-only scratch
    DevBuf iq, soft, con, prs_mag, snr, desc, in8, map, vsym, vdec, vout, ok;
    DevBuf fsym, fdec;                      // Viterbi scratch of the FIC class (it decodes beside the MSC classes on aux_stream)
    RxState* d_state = nullptr;       // [n_ensembles] synchroniser state
    DecState* d_dec = nullptr;        // [n_ensembles] decoder state
    std::vector<void*> owned;

    // ---- streaming receiver (dabphy_stream_* / dabphy_process)
    struct MscClass {
        dabphy_protection prot{};
        std::vector<int> members;     // indices into subch
        DevBuf map, start_bits, tiles, out;  // depuncture map, startAddr*64 per member, gather tiles, decoded bytes [B][members][4F][nbits/8]
        DevBuf steps; int n_windows = 0;     // fused decode (k_viterbi_msc): per-step window-ring descriptors, 16-byte windows of the punctured stream
        DevBuf sf_state;                     // SuperframeFilter window of every (ensemble, member)
        DevBuf sf_snap;                      // ... as it was in front of the current batch (exact batch mode)
    };
    const cf32* s_iq = nullptr;       // DEVICE pointer to [B][stride] samples (caller's or s_iq_own)
    DevBuf s_iq_own;
    uint64_t s_stride = 0, s_ring = 0, s_valid = 0; int s_loop = 0;
    bool s_bounded = false;           // every sample the stream ever held came through k_ingest from u8 / s8 / s16: |re|, |im| <= 1
    std::vector<dabphy_subchannel> subch;
    std::vector<MscClass> classes;
    DevBuf s_raw;                           // staging of raw-format samples (dabphy_stream_write_raw)
    DevBuf s_raw2[2]; hipStream_t copy_stream = nullptr; hipEvent_t ev_ingest[2] = {nullptr, nullptr}; int raw_sel = 0;   // dabphy_stream_write_raw_async
    uint64_t s_enqueued = 0; int commit_slot = -1;    // samples handed to the copy stream so far; slot whose event covers the committed ones
    DevBuf s_null;                          // null symbols on request (dabphy_get_null_symbols)
    DevBuf sf_events, sf_count, sf_bytes, sf_stats; const FrameDesc* last_desc = nullptr;
    static constexpr int N_DESC = 3;    // descriptor buffers: the batch being decoded + up to two synchronised ahead
    DevBuf s_desc2[N_DESC], s_cir2[N_DESC], s_soft, s_con, s_mag, s_snr, s_fib, s_ok;
    hipStream_t sync_stream = nullptr; hipEvent_t ev_sync_done = nullptr;
    hipStream_t aux_stream = nullptr; hipEvent_t ev_demod_done = nullptr, ev_fic_done = nullptr, ev_chain_gate = nullptr;
    FusedMscArgs last_fused{}; bool have_last_fused = false;   // the fused decode launch of the last batch (dabphy_time_fused_msc re-runs it alone)
    bool fused_msc = true;                               // MSC classes with >= 64 CIFs per batch: gather inside the Viterbi kernel (DABPHY_FUSED_MSC=0: two kernels)
    hipEvent_t ev_chain_beg[N_DESC]{}, ev_chain_end[N_DESC]{}; float chain_ms = 0.0f;   // duration of the sync chain that produced the current batch
    // wide synchroniser pass (all frames of a batch at once, k_sync_find_wide/_finish_wide/_validate) and its serial fall-back
    bool wide_sync = true;            // cfg.serial_sync == 0 (DABPHY_SYNC_WIDE overrides)
    DevBuf s_redo[N_DESC];            // [B] first frame slot the wide pass did not settle
    int32_t* d_any_redo = nullptr;    // [N_DESC] device flags; h_any_redo: their page-locked host copies
    int32_t* h_any_redo = nullptr;
    hipEvent_t ev_wide_done[N_DESC]{};
    bool wide_pending[N_DESC]{};      // the wide pass of this descriptor buffer has been queued, its verdict not yet read
    uint64_t chain_valid[N_DESC]{}; uint32_t chain_frames[N_DESC]{};   // n_valid and n_frames the chain of this buffer was queued with
    uint64_t n_wide_passes = 0, n_wide_fallbacks = 0;
    // exact batch mode (cfg.no_batch_replay == 0): state as it was in front of a batch, to replay the batch frame by frame when one of its
    // coarse-corrector decisions was taken with a stale FIC ratio and can have mattered (k_fic_ratio's verdict)
    bool exact_batch = false;
    DevBuf snap_state[N_DESC], snap_dec, snap_tii;
    int32_t* d_any_eff = nullptr; int32_t* h_any_eff = nullptr;
    uint64_t n_replayed_batches = 0;
    int desc_sel = 0;                 // which of s_desc2/s_cir2 holds the batch that dabphy_process decodes next
    int ahead = 0;                    // batches whose chain has been queued but which have not been decoded yet (pipelined modes: 1 or 2 between calls)
    uint32_t presynced = 0;           // frames already synchronised ahead into s_desc2[desc_sel] (pipelined mode)
    int soft_ring = 0;
    uint32_t last_frames = 0;         // n_frames of the last dabphy_process
    float* cur_cir = nullptr;
    FrameDesc* h_desc = nullptr;      // host copy of the last batch's frame descriptors (page-locked, [B][max_frames])
    float* h_snr = nullptr;
    uint8_t *h_fib = nullptr, *h_ok = nullptr;   // ... of its FIBs [B][F][12][32] and CRC flags [B][F][12]: they cross PCIe inside the step, beside the decoder
    int32_t* h_sf_stats = nullptr; bool h_sf_stats_valid = false;   // ... of the superframe totals when the filter rode in dabphy_process
    // stage timing (HIP events on the handle's stream, recorded when profiling is on)
    enum { ST_SYNC = 0, ST_DEMOD, ST_SNR, ST_FIC, ST_MSC_GATHER, ST_MSC_VITERBI, ST_RS, ST_COUNT };
    bool profiling = false;
    hipEvent_t ev_beg[ST_COUNT]{}, ev_end[ST_COUNT]{};
    bool ev_used[ST_COUNT]{};
    DevBuf rs_first, rs_result;
    DevBuf s_hist;                          // [B][HIST_CAP] window searches since the last acquisition (sLevel replay in k_acquire)
    // TII (RadioReceiverOptions::decodeTII): constants, per-batch scratch, per-ensemble sums that live across batches
    bool tii_on = false; bool tii_ran = false;
    bool track_slevel = false;        // dabphy_set_track_slevel: sLevel follows every tracked frame instead of catching up at a loss of lock
    bool sf_auto = false, sf_stats_ready = false;   // dabphy_set_auto_superframes: the all-sub-channel filter rides in dabphy_process's submission
    DevBuf tii_rot, tii_rank, tii_pat, tii_err, tii_likely, tii_state, tii_events, tii_nev, tii_ovf;
    uint32_t tii_max_events = 0;
};

namespace {

// A handle lives on one device; HIP's current device is a property of the calling THREAD.  Every entry point makes the handle's device
// current for its duration, so that one process can own several handles on several devices (welle.io_amd/host/gpu_node_receiver.h:
// one host thread per device) and a caller's own device selection survives the call.
struct DeviceBind {
    int prev = -1; bool switched = false;
    explicit DeviceBind(const dabphy_handle* h)
    {
        if (h && hipGetDevice(&prev) == hipSuccess && prev != h->cfg.device) switched = hipSetDevice(h->cfg.device) == hipSuccess;
    }
    ~DeviceBind() { if (switched) { hipError_t e = hipSetDevice(prev); (void)e; } }
    DeviceBind(const DeviceBind&) = delete; DeviceBind& operator=(const DeviceBind&) = delete;
};

#define HIPCHK(h, call)                                                                                   \
    do { hipError_t e_ = (call); if (e_ != hipSuccess) { (h)->err = std::string(#call) + ": " + hipGetErrorString(e_); return DABPHY_ERR_HIP; } } while (0)

int ensure(dabphy_handle* h, DevBuf& b, size_t bytes)
{
    if (bytes <= b.cap) return 0;
    if (b.p) { hipError_t e = hipFree(b.p); (void)e; b.p = nullptr; b.cap = 0; }
    bytes = (bytes + 4095) & ~(size_t)4095;
    if (hipMalloc(&b.p, bytes) != hipSuccess) { h->err = "hipMalloc failed (" + std::to_string(bytes) + " bytes)"; b.p = nullptr; return DABPHY_ERR_NOMEM; }
    b.cap = bytes;
    return 0;
}

template <typename T> int upload_const(dabphy_handle* h, T** dst, const std::vector<T>& src)
{
    void* p = nullptr;
    if (hipMalloc(&p, src.size() * sizeof(T)) != hipSuccess) { h->err = "hipMalloc(table) failed"; return DABPHY_ERR_NOMEM; }
    h->owned.push_back(p);
    HIPCHK(h, hipMemcpy(p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    *dst = reinterpret_cast<T*>(p);
    return 0;
}

int sync(dabphy_handle* h)
{
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    return 0;
}

// Fill a VitClass for n_cw codewords of nbits and make sure its device buffers exist.
int prepare_class(dabphy_handle* h, VitClass& c, int nbits, int n_cw, int dedisperse)
{
    c.nbits = nbits; c.nsteps = nbits + 6; c.n_cw = n_cw; c.n_groups = (n_cw + 63) / 64; c.dedisperse = dedisperse; c.g_begin = 0; c.g_end = c.n_groups;
    const size_t cells = (size_t)c.n_groups * c.nsteps * 64;
    int r;
    if ((r = ensure(h, h->vsym, cells * sizeof(uint32_t)))) return r;
    if ((r = ensure(h, h->vdec, cells * sizeof(uint2)))) return r;
    if ((r = ensure(h, h->vout, (size_t)c.n_groups * 64 * (nbits / 8)))) return r;
    c.sym = h->vsym.as<uint32_t>(); c.dec = h->vdec.as<uint2>(); c.out = h->vout.as<uint8_t>();
    return 0;
}

} // namespace

extern "C" {

int dabphy_create(const dabphy_config* cfg, dabphy_handle** out)
{
    if (!cfg || !out || cfg->n_ensembles < 1 || cfg->max_frames < 1) return DABPHY_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) return DABPHY_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return DABPHY_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return DABPHY_ERR_NO_DEVICE;     // kernels are built for gfx950 only
    if (hipSetDevice(cfg->device) != hipSuccess) return DABPHY_ERR_NO_DEVICE;
    dabphy_handle* h = new dabphy_handle();
    h->cfg = *cfg;
    // big batches: fewer reference-symbol transforms; small ones: more work-groups.  (Longer chunks -- 38, 75 symbols -- are 2-3 %
    // faster when the kernel runs alone, dabphy_time_demod, and 1-4 % slower inside the pipelined step: measured, round 2.)
    if (h->cfg.demod_chunk <= 0) h->cfg.demod_chunk = ((int64_t)h->cfg.n_ensembles * h->cfg.max_frames >= 1024) ? 25 : 15;
    if (h->cfg.demod_chunk > 75) h->cfg.demod_chunk = 75;
    snprintf(h->devname, sizeof h->devname, "%s (%s)", prop.name, prop.gcnArchName);
    int r = 0;
    auto fail = [&](int code) { dabphy_destroy(h); return code; };
    if (h->cfg.fft_placement < 0 || h->cfg.fft_placement > 2 || h->cfg.freqsync_method < 0 || h->cfg.freqsync_method > 2 || h->cfg.pipeline_sync < 0 || h->cfg.pipeline_sync > 3) return fail(DABPHY_ERR_INVALID);
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return fail(DABPHY_ERR_HIP);
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
    {   // the frame chain is short serial work: give its queue the highest dispatch priority
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = 0; hi = 0; }
        if (hipStreamCreateWithPriority(&h->sync_stream, hipStreamNonBlocking, hi) != hipSuccess) return fail(DABPHY_ERR_HIP);
    }
    if (hipEventCreate(&h->ev_sync_done) != hipSuccess) return fail(DABPHY_ERR_HIP);
    if (hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking) != hipSuccess) return fail(DABPHY_ERR_HIP);
    if (hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) != hipSuccess) return fail(DABPHY_ERR_HIP);
    for (int i = 0; i < 2; i++) if (hipEventCreateWithFlags(&h->ev_ingest[i], hipEventDisableTiming) != hipSuccess) return fail(DABPHY_ERR_HIP);
    if (hipEventCreateWithFlags(&h->ev_chain_gate, hipEventDisableTiming) != hipSuccess) return fail(DABPHY_ERR_HIP);
    if (hipEventCreateWithFlags(&h->ev_demod_done, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->ev_fic_done, hipEventDisableTiming) != hipSuccess) return fail(DABPHY_ERR_HIP);
    for (int i = 0; i < dabphy_handle::N_DESC; i++) if (hipEventCreate(&h->ev_chain_beg[i]) != hipSuccess || hipEventCreate(&h->ev_chain_end[i]) != hipSuccess) return fail(DABPHY_ERR_HIP);
    for (int i = 0; i < dabphy_handle::N_DESC; i++) if (hipEventCreateWithFlags(&h->ev_wide_done[i], hipEventDisableTiming) != hipSuccess) return fail(DABPHY_ERR_HIP);
    {
        void* p = nullptr;
        if (hipMalloc(&p, sizeof(int32_t) * dabphy_handle::N_DESC) != hipSuccess) return fail(DABPHY_ERR_NOMEM);
        h->owned.push_back(p); h->d_any_redo = reinterpret_cast<int32_t*>(p);
        if (hipHostMalloc(&p, sizeof(int32_t) * dabphy_handle::N_DESC, hipHostMallocDefault) != hipSuccess) return fail(DABPHY_ERR_NOMEM);
        h->h_any_redo = reinterpret_cast<int32_t*>(p);
        for (int i = 0; i < dabphy_handle::N_DESC; i++) h->h_any_redo[i] = 0;
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
#endif
    for (int i = 0; i < dabphy_handle::ST_COUNT; i++)
        if (hipEventCreate(&h->ev_beg[i]) != hipSuccess || hipEventCreate(&h->ev_end[i]) != hipSuccess) return fail(DABPHY_ERR_HIP);
    *out = h;
    return DABPHY_OK;
}

namespace {
// every device buffer of one protection class (dabphy_set_subchannels replaces the classes, dabphy_destroy ends them)
void free_class(dabphy_handle::MscClass& c)
{
    hipError_t e = hipSuccess;
    DevBuf* bufs[] = {&c.map, &c.start_bits, &c.tiles, &c.out, &c.steps, &c.sf_state, &c.sf_snap};
    for (DevBuf* b : bufs) if (b->p) { e = hipFree(b->p); b->p = nullptr; b->cap = 0; }
    (void)e;
}
}

void dabphy_destroy(dabphy_handle* h)
{
    DeviceBind dev_(h);
    if (!h) return;
    hipError_t e;
    if (h->sync_stream) { e = hipStreamSynchronize(h->sync_stream); e = hipStreamDestroy(h->sync_stream); }
    if (h->ev_sync_done) e = hipEventDestroy(h->ev_sync_done);
    if (h->aux_stream) { e = hipStreamSynchronize(h->aux_stream); e = hipStreamDestroy(h->aux_stream); }
    if (h->copy_stream) { e = hipStreamSynchronize(h->copy_stream); e = hipStreamDestroy(h->copy_stream); }
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
    if (h->snap_dec.p) e = hipFree(h->snap_dec.p);
    if (h->snap_tii.p) e = hipFree(h->snap_tii.p);
    if (h->ev_fic_done) e = hipEventDestroy(h->ev_fic_done);
    for (int i = 0; i < dabphy_handle::N_DESC; i++) { if (h->ev_chain_beg[i]) e = hipEventDestroy(h->ev_chain_beg[i]); if (h->ev_chain_end[i]) e = hipEventDestroy(h->ev_chain_end[i]); }
    if (h->stream) { e = hipStreamSynchronize(h->stream); e = hipStreamDestroy(h->stream); }
    for (void* p : h->owned) e = hipFree(p);
    for (int i = 0; i < dabphy_handle::ST_COUNT; i++) { if (h->ev_beg[i]) e = hipEventDestroy(h->ev_beg[i]); if (h->ev_end[i]) e = hipEventDestroy(h->ev_end[i]); }
    DevBuf* more[] = {&h->s_raw, &h->s_raw2[0], &h->s_raw2[1], &h->s_null, &h->s_iq_own, &h->s_desc2[0], &h->s_desc2[1], &h->s_desc2[2], &h->s_soft, &h->s_cir2[0], &h->s_cir2[1], &h->s_cir2[2], &h->s_con, &h->s_mag, &h->s_snr, &h->s_fib, &h->s_ok, &h->rs_first, &h->rs_result};
    for (DevBuf* b : more) if (b->p) e = hipFree(b->p);
    { DevBuf* sfb[] = {&h->sf_events, &h->sf_count, &h->sf_bytes, &h->sf_stats}; for (DevBuf* b : sfb) if (b->p) e = hipFree(b->p); }
    { DevBuf* tb[] = {&h->s_hist, &h->tii_rot, &h->tii_rank, &h->tii_pat, &h->tii_err, &h->tii_likely, &h->tii_state, &h->tii_events, &h->tii_nev, &h->tii_ovf}; for (DevBuf* b : tb) if (b->p) e = hipFree(b->p); }
    for (auto& c : h->classes) free_class(c);
    DevBuf* bufs[] = {&h->iq, &h->soft, &h->con, &h->prs_mag, &h->snr, &h->desc, &h->in8, &h->map, &h->vsym, &h->vdec, &h->vout, &h->ok, &h->fsym, &h->fdec};
    for (DevBuf* b : bufs) if (b->p) e = hipFree(b->p);
    (void)e;
    delete h;
}

const char* dabphy_last_error(const dabphy_handle* h) { return h ? h->err.c_str() : "null handle"; }
const char* dabphy_device_name(const dabphy_handle* h) { return h ? h->devname : ""; }

int dabphy_get_config(const dabphy_handle* h, dabphy_config* out)
{
    DeviceBind dev_(h);
    if (!h || !out) return DABPHY_ERR_INVALID;
    *out = h->cfg;
    return DABPHY_OK;
}

static int reset_synchroniser(dabphy_handle* h, bool decoder_too);
static int resolve_all_chains(dabphy_handle* h);
constexpr int HIST_CAP = 64;     // window searches remembered per ensemble for the sLevel replay

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
    launch_fic_gather(g, h->stream);
    VitArgs v{}; v.c = c; v.prbs_words = h->d_prbs_words;
    launch_viterbi(v, h->stream);
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
static SyncArgs sync_args(dabphy_handle* h, int sel, uint32_t F, uint64_t n_valid)
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
static void launch_serial_chain(dabphy_handle* h, SyncArgs sa)
{
    for (int f = 0; f < sa.n_frames; f++) {
        sa.frame = f;
        launch_sync_find(sa, h->sync_stream);
        launch_sync_finish(sa, h->sync_stream);
        if (h->track_slevel) launch_slevel_catchup(sa, h->sync_stream);
    }
}
static int queue_chain(dabphy_handle* h, int sel, uint32_t F)
{
    SyncArgs sa = sync_args(h, sel, F, h->s_valid);
    h->chain_valid[sel] = h->s_valid; h->chain_frames[sel] = F;
    if (h->exact_batch && F > 1 && h->snap_state[sel].p)              // (one frame per call is exact by construction: nothing to put back)
        HIPCHK(h, hipMemcpyAsync(h->snap_state[sel].p, h->d_state, sizeof(RxState) * h->cfg.n_ensembles, hipMemcpyDeviceToDevice, h->sync_stream));
    { hipError_t e = hipEventRecord(h->ev_chain_beg[sel], h->sync_stream); (void)e; }
    // one frame per call (the real-time facade) gains nothing from the wide pass; two batches ahead its verdict would come too late
    if (h->wide_sync && F >= 2 && h->cfg.pipeline_sync != 3 && !h->track_slevel) {
        HIPCHK(h, hipMemsetAsync(h->d_any_redo + sel, 0, sizeof(int32_t), h->sync_stream));
        sa.redo_out = h->s_redo[sel].as<int32_t>(); sa.any_redo = h->d_any_redo + sel;
        launch_sync_wide(sa, h->sync_stream);
        HIPCHK(h, hipMemcpyAsync(h->h_any_redo + sel, h->d_any_redo + sel, sizeof(int32_t), hipMemcpyDeviceToHost, h->sync_stream));
        HIPCHK(h, hipEventRecord(h->ev_wide_done[sel], h->sync_stream));
        h->wide_pending[sel] = true; h->n_wide_passes++;
    } else {
        launch_serial_chain(h, sa);
    }
    { hipError_t e = hipEventRecord(h->ev_chain_end[sel], h->sync_stream); (void)e; }
    return DABPHY_OK;
}
// reads the wide pass's verdict for descriptor buffer `sel` and queues the serial chain for what it did not settle
static int resolve_chain(dabphy_handle* h, int sel)
{
    if (!h->wide_pending[sel]) return DABPHY_OK;
    HIPCHK(h, hipEventSynchronize(h->ev_wide_done[sel]));
    h->wide_pending[sel] = false;
    if (h->h_any_redo[sel]) {
        SyncArgs sa = sync_args(h, sel, h->chain_frames[sel], h->chain_valid[sel]);
        sa.redo_from = h->s_redo[sel].as<int32_t>();
        launch_serial_chain(h, sa);
        { hipError_t e = hipEventRecord(h->ev_chain_end[sel], h->sync_stream); (void)e; }
        h->n_wide_fallbacks++;
    }
    return DABPHY_OK;
}
static int resolve_all_chains(dabphy_handle* h)
{
    for (int i = 0; i < dabphy_handle::N_DESC; i++) { int r = resolve_chain(h, (h->desc_sel + i) % dabphy_handle::N_DESC); if (r) return r; }
    return DABPHY_OK;
}

// OFDMProcessor::restart (ofdm-processor.cpp:115-132) + the start of run(): correctors, phase and sync state zero, sLevel primed over
// the next T_F/2 samples (:252-255).  decoder_too (dabphy_reset: a freshly bound stream) also rewinds the stream to sample 0 and
// clears the frame counter; without it (setReceiverOptions on a running receiver) the stream goes on where the DECODED frames end:
// frames that pipelined mode had synchronised ahead are handed back, so the time de-interleavers see every CIF exactly once.
static int reset_synchroniser(dabphy_handle* h, bool decoder_too)
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
            s.n_exact_sums = keep.n_exact_sums; s.n_relock_inexact = keep.n_relock_inexact; s.n_wide_frames = keep.n_wide_frames;
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
    for (auto& c : h->classes) if (c.sf_state.p) HIPCHK(h, hipMemsetAsync(c.sf_state.p, 0, c.sf_state.cap, h->stream));   // decoders restart too (RadioReceiver::restart_decoder)
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

int dabphy_set_subchannels(dabphy_handle* h, const dabphy_subchannel* list, uint32_t n)
{
    DeviceBind dev_(h);
    if (!h || (n && !list)) return DABPHY_ERR_INVALID;
    for (uint32_t i = 0; i < n; i++) {
        const dabphy_subchannel& s = list[i];
        if (!protection_valid(&s.prot) || s.prot.nbits > PRBS_MAX_BITS || s.start_cu < 0 || s.size_cu <= 0 || s.start_cu + s.size_cu > 864 ||
            protection_input_bits(&s.prot) > s.size_cu * 64) { h->err = "invalid sub-channel " + std::to_string(i); return DABPHY_ERR_INVALID; }
    }
    for (auto& c : h->classes) free_class(c);
    h->classes.clear();
    h->last_frames = 0; h->last_desc = nullptr; h->sf_stats_ready = false;     // the class outputs of the last batch are gone with the classes
    h->subch.assign(list, list + n);
    for (uint32_t i = 0; i < n; i++) {
        dabphy_handle::MscClass* cls = nullptr;
        for (auto& c : h->classes) if (!memcmp(&c.prot, &list[i].prot, sizeof(dabphy_protection))) { cls = &c; break; }
        if (!cls) { h->classes.emplace_back(); cls = &h->classes.back(); cls->prot = list[i].prot; }
        cls->members.push_back((int)i);
    }
    for (auto& c : h->classes) {
        const std::vector<int16_t> m = depuncture_map(&c.prot);
        std::vector<int32_t> sb;
        for (int i : c.members) sb.push_back(h->subch[i].start_cu * 64);
        int r;
        if ((r = ensure(h, c.map, m.size() * sizeof(int16_t)))) return r;
        if ((r = ensure(h, c.start_bits, sb.size() * sizeof(int32_t)))) return r;
        HIPCHK(h, hipMemcpy(c.map.p, m.data(), m.size() * sizeof(int16_t), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(c.start_bits.p, sb.data(), sb.size() * sizeof(int32_t), hipMemcpyHostToDevice));
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
        if ((r = ensure(h, c.tiles, tl.size() * sizeof(int32_t)))) return r;
        HIPCHK(h, hipMemcpy(c.tiles.p, tl.data(), tl.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        // fused decode: what every trellis step reads, in terms of the wave's window ring (k_viterbi_msc).  Source byte u sits in
        // window u >> 4 (slot (u >> 4) & 1), column u & 15, map16[u & 15] rows below the lane's row base.
        {
            static const int map16[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
            constexpr int ROWS = 96, SLOT = ROWS * 16, ZERO = 2 * SLOT;
            constexpr int PADDING = 6;                        // the kernel requests descriptors one block of six steps ahead
            std::vector<MscStep> st((size_t)nsteps + PADDING);
            std::vector<int> wlo((size_t)nsteps, -1), whi((size_t)nsteps, -1);
            for (int q = 0; q < nsteps; q++) {
                uint32_t off[4];
                for (int j = 0; j < 4; j++) {
                    const int u = m[4 * q + j];
                    if (u < 0) { off[j] = ZERO; continue; }
                    const int w = u >> 4, col = u & 15;
                    off[j] = (uint32_t)((w & 1) * SLOT + map16[col] * 16 + col);
                    if (wlo[q] < 0) wlo[q] = w;
                    whi[q] = w;
                }
                st[q].off01 = off[0] | (off[1] << 16); st[q].off23 = off[2] | (off[3] << 16);
            }
            for (int q = nsteps; q < nsteps + PADDING; q++) st[(size_t)q].off01 = st[(size_t)q].off23 = ZERO | (ZERO << 16);
            const int n_in = protection_input_bits(&c.prot);
            c.n_windows = (n_in + 15) / 16;
            // lowest window any LATER step reads: when it moves up, the window below it has died and its slot takes the window after the next
            std::vector<int> low_after((size_t)nsteps, c.n_windows);
            for (int q = nsteps - 2, low = c.n_windows; q >= 0; q--) { if (wlo[q + 1] >= 0) low = wlo[q + 1]; low_after[q] = low; }
            int seen = 1, prev_low = 0;
            std::vector<int> load_step((size_t)c.n_windows + 2, -10);
            bool ok = nsteps % 6 == 0; int why = ok ? 0 : 8;    // (the kernel walks the trellis in blocks of six steps: true for every 24 * bitrate + 6)
            for (int q = 0; q < nsteps; q++) {
                if (whi[q] > seen) {
                    st[q].off01 |= MSC_FIRST_USE; seen = whi[q];
                    // the step BEFORE q waits for the window with s_waitcnt vmcnt(2): its load must be older than two decision stores
                    if (q - 1 - load_step[seen] < 2) { ok = false; why |= 1; }
                }
                if (whi[q] >= 0 && whi[q] - wlo[q] > 1) { ok = false; why |= 2; }
                if (low_after[q] > prev_low) {
                    if (low_after[q] != prev_low + 1 && low_after[q] < c.n_windows) { ok = false; why |= 4; }
                    prev_low = low_after[q];
                    if (prev_low + 1 < c.n_windows) { st[q].off01 |= MSC_LOAD_NEXT; load_step[prev_low + 1] = q; }
                }
            }
            if (!ok && debug_env("DABPHY_DEBUG")) fprintf(stderr, "dabphy: class nbits %d: no fused decode (window schedule, reason %d)\n", c.prot.nbits, why);
            if (!ok) c.n_windows = 0;             // (never for the profiles of EN 300 401; the two-kernel path decodes such a class)
            if ((r = ensure(h, c.steps, st.size() * sizeof(MscStep)))) return r;
            HIPCHK(h, hipMemcpy(c.steps.p, st.data(), st.size() * sizeof(MscStep), hipMemcpyHostToDevice));

        }
    }
    return DABPHY_OK;
}

namespace {
int launch_superframe_stats(dabphy_handle* h);
int run_superframes(dabphy_handle* h, dabphy_handle::MscClass& cls, int member, int32_t* stats, hipStream_t st = nullptr, int ens0 = 0, int ens_count = 0);
// device buffers of the superframe filter for one class and F frames per batch (the window state is zeroed when it is (re)allocated)
int prepare_superframes(dabphy_handle* h, dabphy_handle::MscClass& cls, uint32_t F)
{
    const uint32_t B = h->cfg.n_ensembles;
    const int fb = cls.prot.nbits / 8, M = (int)cls.members.size();
    if ((cls.prot.nbits / 24) % 8 || fb < 10) return 0;                  // not a DAB+ rate: the filter never runs on this class
    const int n_cif = (int)(4 * F), n_slots = n_cif / 5 + 1;
    const size_t stride = ((size_t)16 + 5 * fb + 15) & ~(size_t)15;
    int r;
    if (cls.sf_state.cap < stride * B * M) {
        if ((r = ensure(h, cls.sf_state, stride * B * M))) return r;
        HIPCHK(h, hipMemsetAsync(cls.sf_state.p, 0, cls.sf_state.cap, h->stream));      // frame_count = 0: nothing collected yet
    }
    if ((r = ensure(h, h->sf_events, sizeof(SfEvent) * B * M * n_cif))) return r;
    if ((r = ensure(h, h->sf_count, sizeof(int32_t) * B * M))) return r;
    if ((r = ensure(h, h->sf_bytes, (size_t)B * M * n_slots * 5 * fb))) return r;
    return 0;
}
}

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
    for (int k = 0; k < dabphy_handle::N_DESC; k++) {
        if ((r = ensure(h, h->s_desc2[k], (size_t)B * h->cfg.max_frames * sizeof(FrameDesc)))) return r;
        if ((r = ensure(h, h->s_redo[k], (size_t)B * sizeof(int32_t)))) return r;
        if (h->exact_batch && (r = ensure(h, h->snap_state[k], (size_t)B * sizeof(RxState)))) return r;
        if (h->cfg.want_impulse_response && (r = ensure(h, h->s_cir2[k], (size_t)B * h->cfg.max_frames * T_U * sizeof(float)))) return r;
    }
    {   // + a tail of zeros (one sub-channel's worth: 864 CU x 64 bits) that the fused MSC decode loads for CIFs that do not exist yet
        const size_t ring_bytes = (size_t)B * ring_frames * SOFT_PER_FRAME, tail = 864 * 64 + 64;
        if (h->s_soft.cap < ring_bytes + tail) {
            if ((r = ensure(h, h->s_soft, ring_bytes + tail))) return r;
            HIPCHK(h, hipMemsetAsync(h->s_soft.as<int8_t>() + ring_bytes, 0, tail, h->stream));
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
        const size_t cells = (size_t)fic_c.n_groups * fic_c.nsteps * 64;
        if ((r = ensure(h, h->fsym, cells * sizeof(uint32_t)))) return r;
        if ((r = ensure(h, h->fdec, cells * sizeof(uint2)))) return r;
        if ((r = ensure(h, h->s_fib, (size_t)fic_c.n_groups * 64 * 96))) return r;      // the class output holds whole groups of 64 codewords
        if (h->tii_on) {
            if ((r = ensure(h, h->tii_err, (size_t)B * F * TII_MAX_LIKELY * TII_NERR * sizeof(float)))) return r;
            if ((r = ensure(h, h->tii_likely, (size_t)B * F * (1 + TII_MAX_LIKELY) * sizeof(int32_t)))) return r;
            if ((r = ensure(h, h->tii_events, (size_t)B * TII_MAX_LIKELY * h->cfg.max_frames * sizeof(TiiEvent)))) return r;
            if ((r = ensure(h, h->tii_nev, (size_t)B * sizeof(int32_t)))) return r;
        }
        for (auto& cls : h->classes) {
            VitClass c{};
            if ((r = prepare_class(h, c, cls.prot.nbits, (int)(B * 4 * F * cls.members.size()), 1))) return r;
            if ((r = ensure(h, cls.out, (size_t)c.n_groups * 64 * (cls.prot.nbits / 8)))) return r;
            if (h->sf_auto && (r = prepare_superframes(h, cls, F))) return r;
        }
        if (h->sf_auto && (r = ensure(h, h->sf_stats, sizeof(int32_t) * 4 * B))) return r;
        if (h->exact_batch) {
            if ((r = ensure(h, h->snap_dec, (size_t)B * sizeof(DecState)))) return r;
            if (h->tii_state.p && (r = ensure(h, h->snap_tii, h->tii_state.cap))) return r;
            for (auto& cls : h->classes) if (cls.sf_state.p && (r = ensure(h, cls.sf_snap, cls.sf_state.cap))) return r;
        }
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
    da.soft = h->s_soft.as<int8_t>(); da.soft_ring = ring_frames;
    da.con = h->cfg.want_constellation ? h->s_con.as<cf32>() : nullptr; da.prs_mag = h->s_mag.as<float>();
    da.osc_stats = h->d_osc_stats;
    if (replay) {
        // Exact batch mode, second pass: the batch again, frame by frame, with the reference's own feedback -- the window search of
        // frame f consults the FIC ratio as it stands after frame f - 1 (ofdm-processor.cpp:397), which takes that frame's FIC: chain
        // step, the first chunk(s) of the frame's symbols (PRS + the three FIC symbols), FIC decode of the class, ratio of frame f.  Everything else
        // of the batch follows below as in the first pass (the demod kernel writes the same soft bits again where nothing changed).
        SyncArgs sa = sync_args(h, cur, F, h->chain_valid[cur]);
        VitClass c = fic_c;
        c.sym = h->fsym.as<uint32_t>(); c.dec = h->fdec.as<uint2>(); c.out = h->s_fib.as<uint8_t>();
        FicGatherArgs g{}; g.soft = da.soft; g.soft_ring = ring_frames; g.frame_stride = SOFT_PER_FRAME; g.desc = d_desc;
        g.n_ens = (int)B; g.n_frames = (int)F; g.map = h->d_fic_map; g.c = c;
        VitArgs v{}; v.c = c; v.prbs_words = h->d_prbs_words;
        CrcArgs k{}; k.fib = c.out; k.ok = h->s_ok.as<uint8_t>(); k.state = h->d_dec; k.desc = d_desc; k.n_ens = (int)B; k.n_frames = (int)F; k.disable_coarse = h->cfg.disable_coarse;
        for (uint32_t f = 0; f < F; f++) {
            sa.frame = (int)f;
            launch_sync_find(sa, h->stream);
            launch_sync_finish(sa, h->stream);
            DemodArgs d1 = da; d1.frame_first = (int)f; d1.frame_count = 1; d1.con = nullptr; d1.osc_stats = nullptr;
            d1.chunk_count = (3 + da.chunk_len - 1) / da.chunk_len;          // the chunks that hold the FIC symbols 1..3 (demod_chunk may be 1 or 2)
            launch_demod(d1, (int)B, h->stream);
            launch_fic_gather(g, h->stream);
            launch_viterbi(v, h->stream);
            launch_fib_crc(k, h->stream);
            CrcArgs kf = k; kf.frame_first = (int)f; kf.frame_count = 1;
            launch_fic_ratio(kf, h->stream);
        }
    }
    mark(dabphy_handle::ST_DEMOD, false);
    launch_demod(da, (int)B, h->stream);
    tick(2);
    mark(dabphy_handle::ST_DEMOD, true);
    if (!replay && (h->cfg.pipeline_sync == 1 || h->cfg.pipeline_sync == 3)) HIPCHK(h, hipEventRecord(h->ev_chain_gate, h->stream));
    SnrArgs sn{}; sn.state = h->d_dec; sn.desc = d_desc; sn.n_ens = (int)B; sn.n_frames = (int)F; sn.prs_mag = da.prs_mag; sn.snr_out = h->s_snr.as<float>();
    mark(dabphy_handle::ST_SNR, false);
    launch_snr(sn, h->stream);
    mark(dabphy_handle::ST_SNR, true);

    // FIC: 4 codewords per frame.  Only B*F/16 wavefronts of 774 serial trellis steps: it runs on its own stream beside the
    // MSC classes (own Viterbi scratch), filling execution slots instead of holding the whole device for a latency-bound tail.
    {
        VitClass c = fic_c;
        c.sym = h->fsym.as<uint32_t>(); c.dec = h->fdec.as<uint2>(); c.out = h->s_fib.as<uint8_t>();
        FicGatherArgs g{}; g.soft = da.soft; g.soft_ring = ring_frames; g.frame_stride = SOFT_PER_FRAME; g.desc = d_desc;
        g.n_ens = (int)B; g.n_frames = (int)F; g.map = h->d_fic_map; g.c = c;
        hipStream_t fs = h->aux_stream;
        HIPCHK(h, hipEventRecord(h->ev_demod_done, h->stream));
        HIPCHK(h, hipStreamWaitEvent(fs, h->ev_demod_done, 0));
        mark(dabphy_handle::ST_FIC, false, fs);
        launch_fic_gather(g, fs);
        VitArgs v{}; v.c = c; v.prbs_words = h->d_prbs_words;
        launch_viterbi(v, fs);
        CrcArgs k{}; k.fib = c.out; k.ok = h->s_ok.as<uint8_t>(); k.state = h->d_dec; k.desc = d_desc; k.n_ens = (int)B; k.n_frames = (int)F; k.disable_coarse = h->cfg.disable_coarse;
        launch_fib_crc(k, fs);
        k.any_effective = h->d_any_eff;
        if (!replay) launch_fic_ratio(k, fs);                    // (the second pass of exact batch mode has advanced the ratio frame by frame)
        HIPCHK(h, hipMemcpyAsync(h->h_any_eff, h->d_any_eff, sizeof(int32_t), hipMemcpyDeviceToHost, fs));
        mark(dabphy_handle::ST_FIC, true, fs);
        h->tii_ran = false;
        if (h->tii_on) {
            // TII side path (ofdm-processor.cpp:462-466 -> TIIDecoder): needs only the samples and the frame descriptors, rides behind
            // the FIC on the auxiliary stream
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
        HIPCHK(h, hipMemcpyAsync(h->h_desc, d_desc, (size_t)B * F * sizeof(FrameDesc), hipMemcpyDeviceToHost, fs));
        HIPCHK(h, hipMemcpyAsync(h->h_snr, h->s_snr.p, (size_t)B * F * sizeof(float), hipMemcpyDeviceToHost, fs));
        HIPCHK(h, hipMemcpyAsync(h->h_fib, h->s_fib.p, (size_t)B * F * 384, hipMemcpyDeviceToHost, fs));
        HIPCHK(h, hipMemcpyAsync(h->h_ok, h->s_ok.p, (size_t)B * F * 12, hipMemcpyDeviceToHost, fs));
        HIPCHK(h, hipEventRecord(h->ev_fic_done, fs));
    }
    // MSC: one decode per protection class (stage events bracket the first class only: one class in the canonical ensemble).
    h->last_frames = F;
    h->sf_stats_ready = false; h->h_sf_stats_valid = false;
    for (auto& cls : h->classes) {
        VitClass c{};
        const int M = (int)cls.members.size();
        const int n_cw = (int)(B * 4 * F * M);
        if ((r = prepare_class(h, c, cls.prot.nbits, n_cw, 1))) return r;
        if ((r = ensure(h, cls.out, (size_t)c.n_groups * 64 * (cls.prot.nbits / 8)))) return r;
        c.out = cls.out.as<uint8_t>();
        const bool first_cls = (&cls == &h->classes.front());
        if (h->fused_msc && cls.n_windows > 0 && 4 * F >= 64) {
            // fused: the gather happens inside the Viterbi kernel (needs >= 64 CIFs per sub-channel and batch: a wave then spans at
            // most two (ensemble, sub-channel) pairs)
            FusedMscArgs fa{}; fa.soft = da.soft; fa.soft_ring = ring_frames; fa.n_ens = (int)B; fa.n_frames = (int)F;
            fa.steps = cls.steps.as<MscStep>(); fa.n_windows = cls.n_windows; fa.start_bit = cls.start_bits.as<int32_t>(); fa.n_members = M; fa.desc = d_desc; fa.zero_off16 = (uint32_t)(((size_t)B * ring_frames * SOFT_PER_FRAME) >> 4);
            fa.c = c; fa.prbs_words = h->d_prbs_words;
            if (first_cls) { h->last_fused = fa; h->have_last_fused = true; }
            if (first_cls) mark(dabphy_handle::ST_MSC_VITERBI, false);
            launch_viterbi_msc(fa, h->stream);
            if (first_cls) mark(dabphy_handle::ST_MSC_VITERBI, true);
            continue;
        }
        // batches of fewer than 64 CIFs per sub-channel (and classes whose window schedule the fused kernel cannot follow): two kernels
        MscGatherArgs g{}; g.soft = da.soft; g.soft_ring = ring_frames; g.state = h->d_state; g.n_ens = (int)B; g.n_frames = (int)F;
        g.map = cls.map.as<int16_t>(); g.start_bit = cls.start_bits.as<int32_t>(); g.tiles = cls.tiles.as<int32_t>(); g.n_members = M; g.desc = d_desc; g.c = c;
        if (first_cls) mark(dabphy_handle::ST_MSC_GATHER, false);
        launch_msc_gather(g, h->stream);
        if (first_cls) { mark(dabphy_handle::ST_MSC_GATHER, true); mark(dabphy_handle::ST_MSC_VITERBI, false); }
        VitArgs v{}; v.c = c; v.prbs_words = h->d_prbs_words;
        launch_viterbi(v, h->stream);
        if (first_cls) mark(dabphy_handle::ST_MSC_VITERBI, true);
    }
    if (h->sf_auto) {
        if ((r = launch_superframe_stats(h))) return r;
        h->sf_stats_ready = true;
        HIPCHK(h, hipMemcpyAsync(h->h_sf_stats, h->sf_stats.p, sizeof(int32_t) * 4 * B, hipMemcpyDeviceToHost, h->stream));
        h->h_sf_stats_valid = true;
    }
    return DABPHY_OK;
    };
    if (h->exact_batch && F > 1) {
        // what the decoders carry from batch to batch, as it is in front of this one (the synchroniser's share was saved when this
        // batch's chain was queued: queue_chain)
        HIPCHK(h, hipMemcpyAsync(h->snap_dec.p, h->d_dec, sizeof(DecState) * B, hipMemcpyDeviceToDevice, h->stream));
        if (h->tii_state.p && h->snap_tii.p) HIPCHK(h, hipMemcpyAsync(h->snap_tii.p, h->tii_state.p, h->tii_state.cap, hipMemcpyDeviceToDevice, h->stream));
        for (auto& cls : h->classes) if (cls.sf_state.p && cls.sf_snap.p) HIPCHK(h, hipMemcpyAsync(cls.sf_snap.p, cls.sf_state.p, cls.sf_state.cap, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemsetAsync(h->d_any_eff, 0, sizeof(int32_t), h->stream));
    }
    if ((r = decode(false))) return r;
    if (depth) {
        if (h->cfg.pipeline_sync != 2) HIPCHK(h, hipStreamWaitEvent(h->sync_stream, h->ev_chain_gate, 0));
        for (; h->ahead < 1 + depth; h->ahead++) if ((r = queue_chain(h, (cur + h->ahead) % ND, F))) return r;
    }
    h->desc_sel = (cur + 1) % ND; h->ahead--;
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_fic_done, 0));
    h->last_frames = F;
    tick(3);
    if ((r = sync(h))) return r;
    tick(4);
    if (h->exact_batch && F > 1 && *h->h_any_eff) {
        // Exact batch mode: a coarse-corrector decision of this batch was taken with a stale FIC ratio and can have mattered.  Everything
        // the batch changed is put back -- synchroniser state (as saved when its chain was queued), decoder state, superframe windows,
        // TII sums; the soft-bit ring and the outputs are simply written again -- and the batch is decoded a second time with the
        // feedback the reference has; the chains that ran ahead on the wrong state are queued again behind it.
        HIPCHK(h, hipStreamSynchronize(h->sync_stream));
        HIPCHK(h, hipStreamSynchronize(h->aux_stream));
        for (int i = 0; i < ND; i++) h->wide_pending[i] = false;
        HIPCHK(h, hipMemcpyAsync(h->d_state, h->snap_state[cur].p, sizeof(RxState) * B, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_dec, h->snap_dec.p, sizeof(DecState) * B, hipMemcpyDeviceToDevice, h->stream));
        if (h->tii_state.p && h->snap_tii.p) HIPCHK(h, hipMemcpyAsync(h->tii_state.p, h->snap_tii.p, h->tii_state.cap, hipMemcpyDeviceToDevice, h->stream));
        for (auto& cls : h->classes) if (cls.sf_state.p && cls.sf_snap.p) HIPCHK(h, hipMemcpyAsync(cls.sf_state.p, cls.sf_snap.p, cls.sf_state.cap, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemsetAsync(h->d_any_eff, 0, sizeof(int32_t), h->stream));
        if ((r = decode(true))) return r;
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_fic_done, 0));
        if ((r = sync(h))) return r;
        // the batches synchronised ahead started from the state the first pass left: again, from the right one.  (The chain reads the
        // FIC ratio: the main stream has just been drained.)
        for (int i = 1; i <= depth; i++) if ((r = queue_chain(h, (cur + i) % ND, F))) return r;
        h->n_replayed_batches++;
    }
    if (g_tl_on) { for (int i = 0; i < 5; i++) g_tl.acc[i] += tl[i]; g_tl.n++; if (g_tl.n % 8 == 0) fprintf(stderr, "dabphy timing [us]: before resolve %.1f, resolved %.1f, demod launched %.1f, all launched %.1f, synced %.1f (n=%ld)\n", g_tl.acc[0] / g_tl.n, g_tl.acc[1] / g_tl.n, g_tl.acc[2] / g_tl.n, g_tl.acc[3] / g_tl.n, g_tl.acc[4] / g_tl.n, g_tl.n); }
    { float t = 0; h->chain_ms = (hipEventElapsedTime(&t, h->ev_chain_beg[cur], h->ev_chain_end[cur]) == hipSuccess) ? t : 0.0f; }
    return DABPHY_OK;
}

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

int dabphy_get_msc(dabphy_handle* h, uint32_t subch_index, uint8_t* out, size_t out_capacity, int32_t* first_valid, int32_t* n_rows)
{
    DeviceBind dev_(h);
    if (!h || !out || subch_index >= h->subch.size() || !h->last_frames) return DABPHY_ERR_INVALID;
    if (out_capacity < (size_t)h->cfg.n_ensembles * 4 * h->last_frames * (h->subch[subch_index].prot.nbits / 8)) { h->err = "dabphy_get_msc: output buffer too small"; return DABPHY_ERR_INVALID; }
    const uint32_t B = h->cfg.n_ensembles, F = h->last_frames;
    for (auto& cls : h->classes) {
        for (size_t m = 0; m < cls.members.size(); m++) {
            if (cls.members[m] != (int)subch_index) continue;
            const size_t bytes = cls.prot.nbits / 8, nm = cls.members.size();
            std::vector<uint8_t> all((size_t)B * 4 * F * nm * bytes);
            HIPCHK(h, hipMemcpyAsync(all.data(), cls.out.p, all.size(), hipMemcpyDeviceToHost, h->stream));
            int r = sync(h); if (r) return r;
            const size_t Rn = (size_t)4 * F;
            for (size_t b = 0; b < B; b++) memcpy(out + b * Rn * bytes, all.data() + ((b * nm + m) * Rn) * bytes, Rn * bytes);
            if (first_valid)
                for (uint32_t b = 0; b < B; b++) {
                    // DabAudio emits its first logical frame on the 17th CIF it is fed (dab-audio.cpp:146-149)
                    const int64_t c0 = 4 * h->h_desc[(size_t)b * F].frame_no;
                    first_valid[b] = c0 >= 16 ? 0 : (int32_t)(16 - c0);
                }
            if (n_rows)
                for (uint32_t b = 0; b < B; b++) {
                    int nv = 0;
                    for (uint32_t f = 0; f < F; f++) nv += h->h_desc[(size_t)b * F + f].valid == 1 ? 1 : 0;
                    n_rows[b] = 4 * nv;
                }
            return DABPHY_OK;
        }
    }
    return DABPHY_ERR_INVALID;
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
    HIPCHK(h, hipMemcpyAsync(out, h->s_soft.as<int8_t>() + ((size_t)ensemble * h->soft_ring + slot) * SOFT_PER_FRAME, SOFT_PER_FRAME, hipMemcpyDeviceToHost, h->stream));
    return sync(h);
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

int dabphy_rs_superframes(dabphy_handle* h, uint8_t* sf, uint32_t s_per_sf, uint32_t n_sf, int32_t* corrected, int32_t* uncorrectable)
{
    DeviceBind dev_(h);
    if (!h || !sf || !corrected || !uncorrectable || s_per_sf == 0 || n_sf == 0) return DABPHY_ERR_INVALID;
    const size_t bytes = (size_t)120 * s_per_sf * n_sf;
    int r;
    if ((r = ensure(h, h->in8, bytes))) return r;
    if ((r = ensure(h, h->rs_result, 2 * sizeof(int) * n_sf))) return r;
    HIPCHK(h, hipMemcpyAsync(h->in8.p, sf, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(h->rs_result.p, 0, 2 * sizeof(int) * n_sf, h->stream));
    RsArgs a{}; a.data = h->in8.as<uint8_t>(); a.sf_stride = (size_t)120 * s_per_sf; a.n_sf = (int)n_sf; a.s = (int)s_per_sf;
    a.corr = h->rs_result.as<int>(); a.uncorr = h->rs_result.as<int>() + n_sf;
    launch_rs_superframes(a, h->stream);
    HIPCHK(h, hipMemcpyAsync(sf, h->in8.p, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(corrected, a.corr, sizeof(int) * n_sf, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(uncorrectable, a.uncorr, sizeof(int) * n_sf, hipMemcpyDeviceToHost, h->stream));
    return sync(h);
}

int dabphy_rs_decode_msc(dabphy_handle* h, int32_t subch_index, const int32_t* first_cif, int32_t* corrected, int32_t* uncorrectable)
{
    DeviceBind dev_(h);
    if (!h || !first_cif || !h->last_frames || subch_index >= (int32_t)h->subch.size()) return DABPHY_ERR_INVALID;
    const uint32_t B = h->cfg.n_ensembles, F = h->last_frames;
    const int n_cif = (int)(4 * F), n_sf = n_cif / 5 + 1;
    int r;
    if ((r = ensure(h, h->rs_first, sizeof(int) * B))) return r;
    HIPCHK(h, hipMemcpyAsync(h->rs_first.p, first_cif, sizeof(int) * B, hipMemcpyHostToDevice, h->stream));
    if (corrected) memset(corrected, 0, sizeof(int32_t) * B);
    if (uncorrectable) memset(uncorrectable, 0, sizeof(int32_t) * B);
    bool first_launch = true;
    for (auto& cls : h->classes) {
        int member = -1;
        if (subch_index >= 0) {
            for (size_t m = 0; m < cls.members.size(); m++) if (cls.members[m] == subch_index) member = (int)m;
            if (member < 0) continue;
        }
        const int bitrate = cls.prot.nbits / 24;
        if (bitrate % 8) continue;
        const size_t nres = (size_t)B * n_sf * cls.members.size() * 2;
        if ((r = ensure(h, h->rs_result, nres * sizeof(int)))) return r;
        HIPCHK(h, hipMemsetAsync(h->rs_result.p, 0, nres * sizeof(int), h->stream));
        RsMscArgs a{}; a.out = cls.out.as<uint8_t>(); a.n_ens = (int)B; a.n_cif = n_cif; a.n_members = (int)cls.members.size();
        a.frame_bytes = cls.prot.nbits / 8; a.s = bitrate / 8; a.n_sf_per_ens = n_sf; a.member_only = member;
        a.first_cif = h->rs_first.as<int>(); a.result = h->rs_result.as<int>();
        if (h->profiling && first_launch) { hipError_t e = hipEventRecord(h->ev_beg[dabphy_handle::ST_RS], h->stream); (void)e; }
        launch_rs_msc(a, h->stream);
        if (h->profiling && first_launch) { hipError_t e = hipEventRecord(h->ev_end[dabphy_handle::ST_RS], h->stream); (void)e; h->ev_used[dabphy_handle::ST_RS] = true; }
        first_launch = false;
        if (corrected || uncorrectable) {
            std::vector<int> res(nres);
            HIPCHK(h, hipMemcpyAsync(res.data(), h->rs_result.p, nres * sizeof(int), hipMemcpyDeviceToHost, h->stream));
            if ((r = sync(h))) return r;
            for (uint32_t b = 0; b < B; b++)
                for (size_t k = 0; k < (size_t)n_sf * cls.members.size(); k++) {
                    const size_t o = ((size_t)b * n_sf * cls.members.size() + k) * 2;   // [b][superframe][member]
                    if (corrected) corrected[b] += res[o];
                    if (uncorrectable) uncorrectable[b] += res[o + 1];
                }
        }
    }
    return sync(h);
}

namespace {
// launches k_superframe for one class: member >= 0 -> that member only, -1 -> all members
int run_superframes(dabphy_handle* h, dabphy_handle::MscClass& cls, int member, int32_t* stats, hipStream_t st, int ens0, int ens_count)
{
    const uint32_t B = h->cfg.n_ensembles, F = h->last_frames;
    const int bitrate = cls.prot.nbits / 24, fb = cls.prot.nbits / 8, M = (int)cls.members.size();
    const int n_cif = (int)(4 * F), n_slots = n_cif / 5 + 1;
    const size_t stride = ((size_t)16 + 5 * fb + 15) & ~(size_t)15;
    int r;
    if ((r = prepare_superframes(h, cls, F))) return r;
    SfArgs a{};
    a.out = cls.out.as<uint8_t>(); a.n_ens = (int)B; a.n_cif = n_cif; a.n_members = M; a.frame_bytes = fb;
    a.s = bitrate / 8; a.member = member; a.desc = h->last_desc; a.n_frames = (int)F;
    a.state = cls.sf_state.as<uint8_t>(); a.state_stride = stride; a.events = h->sf_events.as<SfEvent>(); a.n_events = h->sf_count.as<int32_t>();
    a.sf = h->sf_bytes.as<uint8_t>(); a.n_slots = n_slots; a.stats = stats; a.ens0 = ens0; a.ens_count = ens_count;
    launch_superframe(a, st ? st : h->stream);
    return 0;
}
}

int dabphy_superframes(dabphy_handle* h, uint32_t subch_index, dabphy_sf_event* events, int32_t* n_events, uint8_t* sf)
{
    DeviceBind dev_(h);
    static_assert(sizeof(dabphy_sf_event) == sizeof(SfEvent), "event layouts must match");
    if (!h || !events || !n_events || subch_index >= h->subch.size() || !h->last_frames || !h->last_desc) return DABPHY_ERR_INVALID;
    const uint32_t B = h->cfg.n_ensembles, F = h->last_frames;
    for (auto& cls : h->classes)
        for (size_t m = 0; m < cls.members.size(); m++) {
            if (cls.members[m] != (int)subch_index) continue;
            const int bitrate = cls.prot.nbits / 24, fb = cls.prot.nbits / 8, M = (int)cls.members.size();
            if (bitrate % 8 || fb < 10) { h->err = "sub-channel bit rate is not a DAB+ rate"; return DABPHY_ERR_INVALID; }
            const int n_cif = (int)(4 * F), n_slots = n_cif / 5 + 1;
            int r;
            if ((r = run_superframes(h, cls, (int)m, nullptr))) return r;
            for (uint32_t b = 0; b < B; b++) {          // rows of member m
                const size_t bm = (size_t)b * M + m;
                HIPCHK(h, hipMemcpyAsync(events + (size_t)b * n_cif, h->sf_events.as<SfEvent>() + bm * n_cif, sizeof(SfEvent) * n_cif, hipMemcpyDeviceToHost, h->stream));
                HIPCHK(h, hipMemcpyAsync(n_events + b, h->sf_count.as<int32_t>() + bm, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
                if (sf) HIPCHK(h, hipMemcpyAsync(sf + (size_t)b * n_slots * 5 * fb, h->sf_bytes.as<uint8_t>() + bm * n_slots * 5 * fb, (size_t)n_slots * 5 * fb, hipMemcpyDeviceToHost, h->stream));
            }
            return sync(h);
        }
    return DABPHY_ERR_INVALID;
}

namespace {
// SuperframeFilter over every DAB+ sub-channel of every ensemble: one launch per protection class on the main stream, totals into sf_stats
int launch_superframe_stats(dabphy_handle* h)
{
    const uint32_t B = h->cfg.n_ensembles;
    int r;
    if ((r = ensure(h, h->sf_stats, sizeof(int32_t) * 4 * B))) return r;
    HIPCHK(h, hipMemsetAsync(h->sf_stats.p, 0, sizeof(int32_t) * 4 * B, h->stream));
    bool first_launch = true;
    for (auto& cls : h->classes) {
        const int bitrate = cls.prot.nbits / 24, fb = cls.prot.nbits / 8;
        if (bitrate % 8 || fb < 10) continue;
        if (h->profiling && first_launch) { hipError_t e = hipEventRecord(h->ev_beg[dabphy_handle::ST_RS], h->stream); (void)e; }
        if ((r = run_superframes(h, cls, -1, h->sf_stats.as<int32_t>()))) return r;
        if (h->profiling && first_launch) { hipError_t e = hipEventRecord(h->ev_end[dabphy_handle::ST_RS], h->stream); (void)e; h->ev_used[dabphy_handle::ST_RS] = true; }
        first_launch = false;
    }
    return 0;
}
}

int dabphy_set_track_slevel(dabphy_handle* h, int32_t on)
{
    DeviceBind dev_(h);
    if (!h) return DABPHY_ERR_INVALID;
    h->track_slevel = on != 0;
    return DABPHY_OK;
}

int dabphy_set_auto_superframes(dabphy_handle* h, int32_t on)
{
    DeviceBind dev_(h);
    if (!h) return DABPHY_ERR_INVALID;
    h->sf_auto = on != 0;
    return DABPHY_OK;
}

int dabphy_superframes_stats(dabphy_handle* h, int32_t* stats)
{
    DeviceBind dev_(h);
    if (!h || !stats || !h->last_frames || !h->last_desc) return DABPHY_ERR_INVALID;
    int r;
    if (h->sf_stats_ready && h->h_sf_stats_valid) {      // the filter rode in dabphy_process and its totals came back with the batch
        h->sf_stats_ready = false; h->h_sf_stats_valid = false;
        memcpy(stats, h->h_sf_stats, sizeof(int32_t) * 4 * h->cfg.n_ensembles);
        return DABPHY_OK;
    }
    if (!h->sf_stats_ready) { if ((r = launch_superframe_stats(h))) return r; }
    h->sf_stats_ready = false;                   // one filter pass per batch: a second call would feed the same frames again
    HIPCHK(h, hipMemcpyAsync(stats, h->sf_stats.p, sizeof(int32_t) * 4 * h->cfg.n_ensembles, hipMemcpyDeviceToHost, h->stream));
    return sync(h);
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

int dabphy_time_fused_msc(dabphy_handle* h, uint32_t iters, float* ms)
{
    DeviceBind dev_(h);
    if (!h || !ms || iters == 0) return DABPHY_ERR_INVALID;
    if (!h->have_last_fused) { h->err = "no batch has been decoded by the fused MSC kernel yet"; return DABPHY_ERR_STATE; }
    HIPCHK(h, hipDeviceSynchronize());                       // alone on the device: nothing of the pipeline beside it
    hipEvent_t e0, e1; HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
    launch_viterbi_msc(h->last_fused, h->stream);            // (same inputs, same outputs: the launch is idempotent)
    HIPCHK(h, hipEventRecord(e0, h->stream));
    for (uint32_t i = 0; i < iters; i++) launch_viterbi_msc(h->last_fused, h->stream);
    HIPCHK(h, hipEventRecord(e1, h->stream));
    HIPCHK(h, hipEventSynchronize(e1));
    float t = 0; HIPCHK(h, hipEventElapsedTime(&t, e0, e1));
    *ms = t / iters;
    HIPCHK(h, hipEventDestroy(e0)); HIPCHK(h, hipEventDestroy(e1));
    return sync(h);
}

} // extern "C"
