// welle.io_amd/csrc/dabphy_internal.h -- what the translation units of the C ABI share: the handle (streams, events, device buffers of one
// receiver batch), small host helpers, and the internal functions that cross files.  Not installed, not part of include/dabphy.h.
//   dabphy_api.hip          create / destroy / options / sub-channel classes, the stateless seams (demod, Viterbi, FIC, RS) and the timing drivers
//   dabphy_stream.hip       sample rings (bind / upload / write / raw formats), the synchroniser's chain and wide pass, reset
//   dabphy_process.hip      dabphy_process: the pipelined schedules, exact batch mode (replay), the decode of one batch
//   dabphy_fused.hip        the fused decode's host side: per-class step tables, the launch plan (build, classes, work list)
//   dabphy_superframes.hip  Reed-Solomon seams and the DAB+ superframe filter
//   dabphy_getters.hip      everything a caller reads back after a batch, profiling, TII
#pragma once
#define DABPHY_BUILDING_LIBRARY          // (the exported symbol `dabphy_create` is the frozen round-3 entry point here, not the header's inline)
#include "../../include/dabphy.h"
#include "../../include/dabphy_test.h"
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

#define DABPHY_INTERNAL __attribute__((visibility("hidden")))

namespace dabphy {

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


constexpr int HIST_CAP = 64;     // window searches remembered per ensemble for the sLevel replay
// superframe filter launches (dabphy_superframes.hip): a class to walk and which of its pairs (DEVICE list; nullptr: all of them)
struct SfSel { int cls; const int32_t* d_run; int n_run; };
constexpr int SP_SINGLE_MAX_GROUPS = 4096;   // groups of 64 code words a one-class state-parallel launch takes (sp_single_*: the seams, the replay's FIC)
constexpr int SF_BATCH_CLASSES = 256;                                                        // classes per bucket (a handle has at most 255)
constexpr size_t SF_BATCH_BYTES = 3 * (SF_BATCH_CLASSES * sizeof(SfArgs) + (SF_BATCH_CLASSES + 1) * sizeof(int32_t) + 12);
} // namespace dabphy
using namespace dabphy;          // (internal header: the handle below names the kernels' argument blocks)

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
    // One protection class of the batch: the (ensemble, sub-channel) pairs of ALL ensembles that share a protection profile -- every
    // ensemble selects its own sub-channels (msc-handler.cpp:61-103) --, ordered by ensemble, then by position in the ensemble's list.
    struct MscClass {
        dabphy_protection prot{};
        std::vector<MscPair> pairs;          // host mirror of pair_tab
        std::vector<int32_t> subch_id, start_cu;   // per pair: Subchannel::subChId / startAddr (what identifies a running service when the lists change)
        bool cif0_pending = false;           // some pair's cif0 is still -1 on the device: the next decode resolves it (k_pair_cif0), then the mirror follows
        DevBuf map, pair_tab, tiles, out;    // depuncture map, the pair table, gather tiles, decoded bytes [pair][4F][nbits/8]
        DevBuf steps[FUSED_VARIANTS]; int n_windows[FUSED_VARIANTS] = {0, 0, 0};   // fused decode (k_viterbi_fused): per-step window-ring descriptors for each row-count build (0 windows: not decodable that way)
        DevBuf sf_state;                     // SuperframeFilter window of every pair
        size_t sf_pair0 = 0, sf_bytes0 = 0;  // this class's region of the filter's shared event / count / verdict buffers (in pairs) and of its superframe buffer (in bytes): prepare_superframes
        DevBuf sf_snap;                      // ... as it was in front of the current batch (exact batch mode)
        bool dabplus_rate() const { return (prot.nbits / 24) % 8 == 0 && prot.nbits / 8 >= 10; }
        size_t sf_stride() const { return ((size_t)16 + 5 * (size_t)(prot.nbits / 8) + 15) & ~(size_t)15; }
    };
    struct PairRef { int cls = -1, pair = -1; };
    const cf32* s_iq = nullptr;       // DEVICE pointer to [B][stride] samples (caller's or s_iq_own)
    DevBuf s_iq_own;
    uint64_t s_stride = 0, s_ring = 0, s_valid = 0; int s_loop = 0;
    bool s_bounded = false;           // every sample the stream ever held came through k_ingest from u8 / s8 / s16: |re|, |im| <= 1
    // sub-channel selection: per ensemble.  `subch_e` + `classes` + `where` = what the kernels decode with; `subch_next` = what the caller
    // has asked for (dabphy_set_subchannels / _ensemble); apply_subchannels rebuilds the former from the latter (carrying the state of
    // every service that stays) before the next batch is decoded.
    std::vector<std::vector<dabphy_subchannel>> subch_e, subch_next;     // [n_ensembles]
    std::vector<std::vector<PairRef>> where;                             // [n_ensembles][position in the list] -> class, pair
    bool subch_dirty = false;
    std::vector<MscClass> classes;
    DevBuf sf_batch; void* h_sf_batch = nullptr;                         // argument blocks of the filter's per-bucket launches (device, page-locked staging)
    DevBuf sf_run;                                                       // pair selections of one-sub-channel superframe filter launches
    DevBuf s_raw;                           // staging of raw-format samples (dabphy_stream_write_raw)
    DevBuf s_raw2[2]; hipStream_t copy_stream = nullptr; hipEvent_t ev_ingest[2] = {nullptr, nullptr}; int raw_sel = 0;   // dabphy_stream_write_raw_async
    uint64_t s_enqueued = 0; int commit_slot = -1;    // samples handed to the copy stream so far; slot whose event covers the committed ones
    DevBuf s_null;                          // null symbols on request (dabphy_get_null_symbols)
    // bulk MSC drain (dabphy_msc_drain_begin / _wait): the class outputs leave on a stream of their own while the next batch starts
    hipStream_t drain_stream = nullptr; hipEvent_t ev_drain_done = nullptr, ev_drain_staged = nullptr; bool drain_pending = false; DevBuf drain_stage;
    DevBuf sf_events, sf_count, sf_bytes, sf_stats, sf_gf, sf_accept; const FrameDesc* last_desc = nullptr;
    static constexpr int N_DESC = 3;    // descriptor buffers: the batch being decoded + up to two synchronised ahead
    DevBuf s_desc2[N_DESC], s_cir2[N_DESC], s_soft, s_con, s_mag, s_snr, s_fib, s_ok;
    int stream_layout = 1;                          // experiments: bit 0 placeholder streams, bit 1 FIC work on the auxiliary stream, bit 3 the bulk drain on a stream of its own even when nothing is ingested asynchronously, bit 4 the SNR kernels on the main stream in front of the decoder (no gain: profiles/r06_step_variants.txt)
    std::vector<hipStream_t> placeholder_streams;   // created in front of the handle's own, never used (dabphy_create_v2: stream placement)
    hipStream_t sync_stream = nullptr; hipEvent_t ev_sync_done = nullptr;
    hipStream_t aux_stream = nullptr; hipEvent_t ev_demod_done = nullptr, ev_fic_done = nullptr, ev_chain_gate = nullptr;
    hipStream_t fic_stream = nullptr; hipEvent_t ev_aux_done = nullptr;     // FIB CRC + FIC ratio behind a fused launch; end of the auxiliary stream's work of a batch
    // fused decode (k_viterbi_fused): every class of the batch (and the FIC) in one launch.  The plan = which build, which classes, the
    // work list; rebuilt when the batch depth, the class set or a buffer address changes (dabphy_fused.hip)
    struct FusedPlan {
        bool valid = false; uint32_t F = 0; bool want_fic = false; bool fic_in = false;
        int variant = 0, n_slots = 0; size_t dec_slot_cells = 0;
        bool use_sp = false; int sp_variant = 0; bool sp_two = false;   // ... two code words per wavefront (k_viterbi_sp2)         // the batch is small: one wavefront per code word (k_viterbi_sp) instead of 64 code words per wavefront
        std::vector<int> class_idx;                      // classes decoded by the fused launch (the others take k_msc_gather + k_viterbi)
        std::vector<FusedClass> host_cls; std::vector<uint32_t> host_work, host_dec_off;
        bool tb_split = false;                           // the launch publishes its groups' decisions for k_traceback_fused (decision scratch then always per group)
        bool dec_by_item = false;                        // decision scratch per group of 64 code words instead of per work-group (dabphy_fused.hip)
        FusedArgs args{}; uint64_t buf_gen = 0;          // the launch as it was last queued (dabphy_time_fused_msc re-runs it alone while buf_gen is current)
        bool launched = false;
    } fplan;
    DevBuf fused_cls, fused_work, fused_dec_off; uint32_t* d_fused_next = nullptr;
    // the traceback of the lane-per-code-word kernel as a pass of its own beside the forward pass (k_traceback_fused): per-item flags + cursor, its stream
    uint32_t* h_tb_gave_up = nullptr;                    // page-locked: walkers that gave up on a flag in the last launch (must be 0)
    bool tb_split = false; bool tb_no_walkers = true; DevBuf fused_done; hipStream_t tb_stream = nullptr; hipEvent_t ev_tb_fork = nullptr, ev_tb_join = nullptr;
    bool sp1_two = false;                                // the last one-class launch prepared goes to k_viterbi_sp2
    DevBuf sp1_cls, sp1_work; void* h_sp1 = nullptr;     // one-class state-parallel launches (the seams, the replay's one-frame FIC): descriptor + work list, page-locked staging
    DevBuf fic_steps[FUSED_VARIANTS]; int fic_windows[FUSED_VARIANTS] = {0, 0, 0};
    hipEvent_t ev_fused_done = nullptr;
    uint64_t buf_gen = 1;                                // bumped whenever a device buffer is reallocated or a class is rebuilt
    bool fused_msc = true;                               // MSC classes: gather inside the Viterbi kernel (DABPHY_FUSED_MSC=0: two kernels)
    uint32_t sp_max_codewords = 40960;                   // batches with at most this many code words (all classes + FIC) are decoded state-parallel: measured crossover of the whole call, profiles/r05_viterbi_sp2.txt (DABPHY_SP_MAX_CW; 0: never)
    uint32_t sp2_min_codewords = 1024;                   // ... of which those above this many take two code words per wavefront and the traceback as a pass of its own (k_viterbi_sp2 + k_traceback_sp2) (DABPHY_SP2_MIN_CW)
    uint32_t sp2_tb_resident = 512;                      // k_traceback_sp2: work-groups of four waves the device holds at once (two per compute unit: LDS); a launch of more takes three waves each (DABPHY_SP2_TB_RESIDENT)
    uint32_t sp2_tb_warm = 4;                            // k_traceback_sp2: blocks of 30 steps a stretch's walk runs in over (120 steps ~ 17 constraint lengths; DABPHY_SP2_TB_WARM)
    bool chain_early = false;                            // pipelined schedules: queue the next batch's synchroniser in front of this batch's decoder instead of behind it (DABPHY_CHAIN_EARLY)
    bool fused_fic = true;                               // the FIC rides in the same launch (DABPHY_FUSED_FIC=0: k_fic_gather + k_viterbi on the auxiliary stream)
    hipEvent_t ev_chain_beg[N_DESC]{}, ev_chain_end[N_DESC]{}; float chain_ms = 0.0f;   // duration of the sync chain that produced the current batch
    // wide synchroniser pass (all frames of a batch at once, k_sync_find_wide/_finish_wide/_validate) and its serial fall-back
    bool wide_sync = true;            // cfg.serial_sync == 0 (DABPHY_SYNC_WIDE overrides)
    DevBuf s_redo[N_DESC];            // [B] first frame slot the wide pass did not settle
    int32_t* d_any_redo = nullptr;    // [N_DESC] verdict flags in page-locked host memory: h_any_redo = the host's address, d_any_redo = the device's
    int32_t* h_any_redo = nullptr;
    bool drift_seen = false;          // the last resolved pass settled frames through the find chain (ensembles whose PRS window moves): cfg.sync_early == 0 then queues the next batch's synchroniser in FRONT of the decoder
    hipEvent_t ev_wide_done[N_DESC]{};
    hipEvent_t ev_wide_front = nullptr; bool wide_front_recorded = false;   // behind the wide pass proper of the chain queued last (cfg.sync_early == 3: the decoder's launch waits for it)
    bool wide_pending[N_DESC]{};      // the wide pass of this descriptor buffer has been queued, its verdict not yet read
    uint64_t chain_valid[N_DESC]{}; uint32_t chain_frames[N_DESC]{};   // n_valid and n_frames the chain of this buffer was queued with
    uint64_t n_wide_passes = 0, n_wide_fallbacks = 0;
    // exact batch mode (cfg.no_batch_replay == 0): state as it was in front of a batch, to replay the batch frame by frame when one of its
    // coarse-corrector decisions was taken with a stale FIC ratio and can have mattered (k_fic_ratio's verdict)
    bool exact_batch = false;
    DevBuf snap_state[N_DESC], snap_hist[N_DESC], snap_dec, snap_tii;     // (snap_hist: the synchroniser's history ring goes with its state -- an acquisition inside the first pass restarts the ring over the entries the second pass must replay)
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
    // dabphy_set_auto_superframes(2): the filter pass of batch k runs beside batch k + 1's FFT stage, on a stream of its own
    bool sf_deferred = false, sf_def_pending = false, sf_def_unfetched = false, sf_def_inflight = false;
    const FrameDesc* sf_def_desc = nullptr; uint32_t sf_def_frames = 0;
    hipStream_t rs_stream = nullptr; hipEvent_t ev_rs_done = nullptr;
    bool sf_auto = false, sf_stats_ready = false;   // dabphy_set_auto_superframes: the all-sub-channel filter rides in dabphy_process's submission
    DevBuf tii_rot, tii_rank, tii_pat, tii_err, tii_likely, tii_state, tii_events, tii_nev, tii_ovf;
    uint32_t tii_max_events = 0;
};

namespace dabphy {

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

inline int ensure(dabphy_handle* h, DevBuf& b, size_t bytes)
{
    if (bytes <= b.cap) return 0;
    if (b.p) { hipError_t e = hipFree(b.p); (void)e; b.p = nullptr; b.cap = 0; }
    h->buf_gen++;
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

// Exact batch mode's second pass is armed whenever a coarse-corrector decision can have seen a FIC ratio older than the reference's
// (ofdm-processor.cpp:397-409: the ratio of the PREVIOUS frame): batches of several frames, and ONE frame per call too when the
// synchroniser runs ahead of the decoder (pipeline_sync 1-3).  One frame per call on the serial schedule is exact by construction.
inline bool replay_armed(const dabphy_handle* h, uint32_t F) { return h->exact_batch && (F > 1 || h->cfg.pipeline_sync != 0); }

inline int sync(dabphy_handle* h)
{
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    return 0;
}

// Fill a VitClass for n_cw codewords of nbits and make sure its device buffers exist.  (h->vdec is also the decision scratch of the
// fused decode: grow-only, so a buffer that is large enough for one of the two users is never shrunk under the other.)
inline int prepare_class(dabphy_handle* h, VitClass& c, int nbits, int n_cw, int dedisperse)
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

} // namespace dabphy

// ---- internal functions that cross translation units (C linkage like their callers, hidden from the library's export table)
extern "C" {
DABPHY_INTERNAL int reset_synchroniser(dabphy_handle* h, bool decoder_too);                       // dabphy_stream.hip
DABPHY_INTERNAL int resolve_all_chains(dabphy_handle* h);
DABPHY_INTERNAL SyncArgs sync_args(dabphy_handle* h, int sel, uint32_t F, uint64_t n_valid);
DABPHY_INTERNAL void launch_serial_chain(dabphy_handle* h, SyncArgs sa);
DABPHY_INTERNAL int queue_chain(dabphy_handle* h, int sel, uint32_t F);
DABPHY_INTERNAL int resolve_chain(dabphy_handle* h, int sel);
DABPHY_INTERNAL int prepare_superframes(dabphy_handle* h, uint32_t F);       // dabphy_superframes.hip
DABPHY_INTERNAL int run_superframes(dabphy_handle* h, const std::vector<dabphy::SfSel>& sel, int32_t* stats, hipStream_t st = nullptr, const FrameDesc* desc = nullptr, uint32_t n_frames = 0);
DABPHY_INTERNAL int apply_subchannels(dabphy_handle* h);                                          // dabphy_api.hip
DABPHY_INTERNAL int upload_pairs(dabphy_handle* h, dabphy_handle::MscClass& cls);
DABPHY_INTERNAL void free_class(dabphy_handle::MscClass& c);
DABPHY_INTERNAL int launch_superframe_stats(dabphy_handle* h, hipStream_t st = nullptr, const FrameDesc* desc = nullptr, uint32_t n_frames = 0);
DABPHY_INTERNAL int launch_deferred_superframes(dabphy_handle* h);       // dabphy_set_auto_superframes(2): the pending pass of the previous batch, on rs_stream
DABPHY_INTERNAL int flush_deferred_superframes(dabphy_handle* h);        // ... now, and wait for it
DABPHY_INTERNAL int fused_class_tables(dabphy_handle* h, const dabphy_protection& prot, bool fic, DevBuf (&steps)[FUSED_VARIANTS], int (&n_windows)[FUSED_VARIANTS]);   // dabphy_fused.hip
DABPHY_INTERNAL int fused_plan(dabphy_handle* h, uint32_t F, bool want_fic);
DABPHY_INTERNAL bool sp_single_ok(const dabphy_handle* h, uint64_t n_cw, int nsteps);
DABPHY_INTERNAL int sp_single_reserve(dabphy_handle* h, uint64_t n_cw, int nsteps);
DABPHY_INTERNAL int sp_single_prepare(dabphy_handle* h, const FusedClass& fc, FusedArgs& a, hipStream_t st);
DABPHY_INTERNAL int sp_variant_for(int nsteps);
// the state-parallel launch of `a`: two code words per wavefront (k_viterbi_sp2) or one (k_viterbi_sp), as sp_two_for decided when the
// decision scratch was laid out
DABPHY_INTERNAL bool sp_two_for(const dabphy_handle* h, uint64_t n_cw);
DABPHY_INTERNAL void launch_sp(const FusedArgs& a, bool two, int lds_variant, hipStream_t s);
DABPHY_INTERNAL int drain_wait(dabphy_handle* h);                                                // dabphy_getters.hip: host waits for a bulk MSC drain in flight
DABPHY_INTERNAL size_t soft_ens_stride(const dabphy_handle* h);                                   // bytes between the soft-bit ring slices of two ensembles
}
