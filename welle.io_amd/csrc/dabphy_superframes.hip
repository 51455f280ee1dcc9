// welle.io_amd/csrc/dabphy_superframes.hip -- Reed-Solomon seams (RSDecoder::DecodeSuperframe) and the DAB+ superframe filter (SuperframeFilter::Feed) of a batch.
// (split from dabphy_api.hip in round 3; dabphy_internal.h has the map of the translation units)
#include "dabphy_internal.h"

extern "C" {

// Device buffers of the superframe filter for F frames per batch: every DAB+-rate class gets its own region of the event / count /
// superframe / verdict buffers (all classes of a bucket run in one launch); the window state of a class is created with the class
// (apply_subchannels, which also carries the windows of the services that stay).
int prepare_superframes(dabphy_handle* h, uint32_t F)
{
    const int n_cif = (int)(4 * F), n_slots = n_cif / 5 + 1;
    size_t pairs = 0, bytes = 0;
    int r;
    for (auto& cls : h->classes) {
        if (!cls.dabplus_rate()) continue;
        const size_t P = cls.pairs.size();
        cls.sf_pair0 = pairs; cls.sf_bytes0 = bytes;
        pairs += P; bytes += P * n_slots * 5 * (size_t)(cls.prot.nbits / 8);
        if (cls.sf_state.cap < cls.sf_stride() * P) {
            if ((r = ensure(h, cls.sf_state, cls.sf_stride() * P))) return r;
            HIPCHK(h, hipMemsetAsync(cls.sf_state.p, 0, cls.sf_state.cap, h->stream));      // frame_count = 0: nothing collected yet
        }
    }
    if (!pairs) return 0;
    if ((r = ensure(h, h->sf_events, sizeof(SfEvent) * pairs * n_cif))) return r;
    if ((r = ensure(h, h->sf_count, sizeof(int32_t) * pairs))) return r;
    if ((r = ensure(h, h->sf_bytes, bytes))) return r;
    if ((r = ensure(h, h->sf_accept, sizeof(int32_t) * pairs))) return r;
    if ((r = ensure(h, h->sf_batch, SF_BATCH_BYTES))) return r;
    if (!h->h_sf_batch) {
        void* p = nullptr;
        if (hipHostMalloc(&p, SF_BATCH_BYTES, hipHostMallocDefault) != hipSuccess) { h->err = "hipHostMalloc failed (superframe filter staging)"; return DABPHY_ERR_NOMEM; }
        h->h_sf_batch = p;
    }
    if (!h->sf_gf.p) {
        // GF(256) of RS(120,110), generator polynomial 0x11D (init_rs.h:48-60): alpha_to[256], index_of[256]  (built once, thread-safely:
        // the node receiver creates its handles from several host threads)
        struct Gf { uint8_t b[512]; Gf() { int sr = 1; memset(b, 0, sizeof b); b[256 + 0] = 255; b[255] = 0; for (int i = 0; i < 255; i++) { b[256 + sr] = (uint8_t)i; b[i] = (uint8_t)sr; sr <<= 1; if (sr & 256) sr ^= 0x11D; sr &= 255; } } };
        static const Gf gf_tab;
        const uint8_t (&gf)[512] = gf_tab.b;
        if ((r = ensure(h, h->sf_gf, sizeof gf + 2 * sizeof(unsigned long long)))) return r;      // + the wide pass' two counters
        HIPCHK(h, hipMemsetAsync(h->sf_gf.p, 0, sizeof gf + 2 * sizeof(unsigned long long), h->stream));
        HIPCHK(h, hipMemcpyAsync(h->sf_gf.p, gf, sizeof gf, hipMemcpyHostToDevice, h->stream));
    }
    return 0;
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
    if (!h || !first_cif || !h->last_frames) return DABPHY_ERR_INVALID;
    const uint32_t B = h->cfg.n_ensembles, F = h->last_frames;
    const int n_cif = (int)(4 * F), n_sf = n_cif / 5 + 1;
    int r;
    if ((r = ensure(h, h->rs_first, sizeof(int) * B))) return r;
    HIPCHK(h, hipMemcpyAsync(h->rs_first.p, first_cif, sizeof(int) * B, hipMemcpyHostToDevice, h->stream));
    if (corrected) memset(corrected, 0, sizeof(int32_t) * B);
    if (uncorrectable) memset(uncorrectable, 0, sizeof(int32_t) * B);
    bool first_launch = true, any = subch_index < 0;
    for (auto& cls : h->classes) {
        if (subch_index >= 0) {
            bool here = false;
            for (const MscPair& p : cls.pairs) if (p.idx == subch_index) { here = true; break; }
            if (!here) continue;
            any = true;
        }
        const int bitrate = cls.prot.nbits / 24;
        if (bitrate % 8) continue;
        const size_t P = cls.pairs.size(), nres = P * n_sf * 2;
        if ((r = ensure(h, h->rs_result, nres * sizeof(int)))) return r;
        HIPCHK(h, hipMemsetAsync(h->rs_result.p, 0, nres * sizeof(int), h->stream));
        RsMscArgs a{}; a.out = cls.out.as<uint8_t>(); a.n_cif = n_cif; a.n_pairs = (int)P; a.pairs = cls.pair_tab.as<MscPair>();
        a.frame_bytes = cls.prot.nbits / 8; a.s = bitrate / 8; a.n_sf_per_pair = n_sf; a.idx_only = subch_index;
        a.first_cif = h->rs_first.as<int>(); a.result = h->rs_result.as<int>();
        if (h->profiling && first_launch) { hipError_t e = hipEventRecord(h->ev_beg[dabphy_handle::ST_RS], h->stream); (void)e; }
        launch_rs_msc(a, h->stream);
        if (h->profiling && first_launch) { hipError_t e = hipEventRecord(h->ev_end[dabphy_handle::ST_RS], h->stream); (void)e; h->ev_used[dabphy_handle::ST_RS] = true; }
        first_launch = false;
        if (corrected || uncorrectable) {
            std::vector<int> res(nres);
            HIPCHK(h, hipMemcpyAsync(res.data(), h->rs_result.p, nres * sizeof(int), hipMemcpyDeviceToHost, h->stream));
            if ((r = sync(h))) return r;
            for (size_t p = 0; p < P; p++)
                for (int q = 0; q < n_sf; q++) {
                    const size_t o = (p * n_sf + q) * 2;                            // [pair][superframe]
                    if (corrected) corrected[cls.pairs[p].ens] += res[o];
                    if (uncorrectable) uncorrectable[cls.pairs[p].ens] += res[o + 1];
                }
        }
    }
    if (!any) return DABPHY_ERR_INVALID;                             // no ensemble has a sub-channel at that position
    return sync(h);
}

// The filter over a selection of classes: sel[i] = (class, DEVICE list of its pairs to walk or nullptr for all of them, how many).  The
// classes of a bucket (kernel LDS size) go in ONE launch each of the wide pass, the verdict and the serial walk; their argument blocks
// travel through a page-locked staging area that the caller must not reuse before the stream has been synchronised (every caller
// synchronises before it returns, dabphy_process at its end).
int run_superframes(dabphy_handle* h, const std::vector<SfSel>& sel, int32_t* stats, hipStream_t st, const FrameDesc* desc, uint32_t n_frames)
{
    const uint32_t F = n_frames ? n_frames : h->last_frames;        // (a deferred pass names the batch it belongs to: the handle has moved on)
    if (!desc) desc = h->last_desc;
    const int n_cif = (int)(4 * F), n_slots = n_cif / 5 + 1;
    int r;
    if ((r = prepare_superframes(h, F))) return r;
    if (!h->sf_batch.p) return 0;                                        // no DAB+-rate class at all
    if (!st) st = h->stream;
    // staging layout per bucket: [SF_BATCH_CLASSES argument blocks][SF_BATCH_CLASSES + 1 first blocks]
    uint8_t* const hs = reinterpret_cast<uint8_t*>(h->h_sf_batch);
    uint8_t* const ds = h->sf_batch.as<uint8_t>();
    constexpr size_t BUCKET = SF_BATCH_BYTES / 3, FIRST0 = SF_BATCH_CLASSES * sizeof(SfArgs);
    int n_in[3] = {0, 0, 0}, blocks[3] = {0, 0, 0};
    for (const SfSel& e : sel) {
        auto& cls = h->classes[e.cls];
        if (!cls.dabplus_rate() || cls.pairs.empty()) continue;
        const int bitrate = cls.prot.nbits / 24, fb = cls.prot.nbits / 8, bk = sf_bucket(fb);
        SfArgs a{};
        a.out = cls.out.as<uint8_t>(); a.n_cif = n_cif; a.n_pairs = (int)cls.pairs.size(); a.pairs = cls.pair_tab.as<MscPair>(); a.frame_bytes = fb;
        a.run = e.d_run; a.n_run = e.d_run ? e.n_run : a.n_pairs;
        if (a.n_run <= 0) continue;
        a.s = bitrate / 8; a.desc = desc; a.n_frames = (int)F;
        a.state = cls.sf_state.as<uint8_t>(); a.state_stride = cls.sf_stride();
        a.events = h->sf_events.as<SfEvent>() + cls.sf_pair0 * n_cif; a.n_events = h->sf_count.as<int32_t>() + cls.sf_pair0;
        a.sf = h->sf_bytes.as<uint8_t>() + cls.sf_bytes0; a.n_slots = n_slots; a.stats = stats;
        a.gf = h->sf_gf.as<uint8_t>(); a.accepted = h->sf_accept.as<int32_t>() + cls.sf_pair0;
        a.wide_stats = reinterpret_cast<unsigned long long*>(h->sf_gf.as<uint8_t>() + 512);
        reinterpret_cast<SfArgs*>(hs + bk * BUCKET)[n_in[bk]] = a;
        reinterpret_cast<int32_t*>(hs + bk * BUCKET + FIRST0)[n_in[bk]] = blocks[bk];
        n_in[bk]++; blocks[bk] += a.n_run;
    }
    for (int bk = 0; bk < 3; bk++) {
        if (!n_in[bk]) continue;
        reinterpret_cast<int32_t*>(hs + bk * BUCKET + FIRST0)[n_in[bk]] = blocks[bk];
        HIPCHK(h, hipMemcpyAsync(ds + bk * BUCKET, hs + bk * BUCKET, n_in[bk] * sizeof(SfArgs), hipMemcpyHostToDevice, st));
        HIPCHK(h, hipMemcpyAsync(ds + bk * BUCKET + FIRST0, hs + bk * BUCKET + FIRST0, (n_in[bk] + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
    }
    for (int bk = 0; bk < 3; bk++) {
        if (!n_in[bk]) continue;
        SfBatch bt{}; bt.cls = reinterpret_cast<const SfArgs*>(ds + bk * BUCKET); bt.first = reinterpret_cast<const int32_t*>(ds + bk * BUCKET + FIRST0); bt.n_cls = n_in[bk];
        launch_superframe_bucket(bt, bk, blocks[bk], n_cif, true, st);
    }
    return 0;
}

namespace {
// The filter over a selection of pairs (class, pair) given per output row; events / counts / superframes of row i land at
// events + i * n_cif, n_events + i, sf + i * n_slots * 5 * fb.  Every selected pair must have frames of fb bytes.
int superframes_of(dabphy_handle* h, const std::vector<dabphy_handle::PairRef>& rows, int fb, dabphy_sf_event* events, int32_t* n_events, uint8_t* sf)
{
    const uint32_t F = h->last_frames;
    const int n_cif = (int)(4 * F), n_slots = n_cif / 5 + 1;
    int r;
    if ((r = prepare_superframes(h, F))) return r;
    std::vector<int32_t> run(rows.size());
    if ((r = ensure(h, h->sf_run, rows.size() * sizeof(int32_t)))) return r;
    std::vector<SfSel> sel; std::vector<std::vector<size_t>> mine_of;
    size_t used = 0;
    for (size_t ci = 0; ci < h->classes.size(); ci++) {
        std::vector<size_t> mine;
        for (size_t i = 0; i < rows.size(); i++) if (rows[i].cls == (int)ci) mine.push_back(i);
        if (mine.empty()) continue;
        for (size_t k = 0; k < mine.size(); k++) run[used + k] = rows[mine[k]].pair;
        sel.push_back(SfSel{(int)ci, h->sf_run.as<int32_t>() + used, (int)mine.size()});
        used += mine.size(); mine_of.push_back(std::move(mine));
    }
    HIPCHK(h, hipMemcpyAsync(h->sf_run.p, run.data(), used * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    if ((r = run_superframes(h, sel, nullptr, h->stream))) return r;
    for (size_t s = 0; s < sel.size(); s++) {
        const auto& cls = h->classes[sel[s].cls];
        for (size_t i : mine_of[s]) {
            const size_t bm = (size_t)rows[i].pair;
            HIPCHK(h, hipMemcpyAsync(events + i * n_cif, h->sf_events.as<SfEvent>() + (cls.sf_pair0 + bm) * n_cif, sizeof(SfEvent) * n_cif, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipMemcpyAsync(n_events + i, h->sf_count.as<int32_t>() + cls.sf_pair0 + bm, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
            if (sf) HIPCHK(h, hipMemcpyAsync(sf + i * n_slots * 5 * fb, h->sf_bytes.as<uint8_t>() + cls.sf_bytes0 + bm * n_slots * 5 * fb, (size_t)n_slots * 5 * fb, hipMemcpyDeviceToHost, h->stream));
        }
    }
    return sync(h);                  // (`run` and the staging area are free again)
}
}

int dabphy_superframes(dabphy_handle* h, uint32_t subch_index, dabphy_sf_event* events, int32_t* n_events, uint8_t* sf)
{
    DeviceBind dev_(h);
    static_assert(sizeof(dabphy_sf_event) == sizeof(SfEvent), "event layouts must match");
    if (!h || !events || !n_events || !h->last_frames || !h->last_desc) return DABPHY_ERR_INVALID;
    const uint32_t B = h->cfg.n_ensembles;
    std::vector<dabphy_handle::PairRef> rows(B);
    int fb = 0;
    for (uint32_t b = 0; b < B; b++) {
        if (subch_index >= h->where[b].size()) { h->err = "ensemble " + std::to_string(b) + " has no sub-channel " + std::to_string(subch_index); return DABPHY_ERR_INVALID; }
        rows[b] = h->where[b][subch_index];
        const auto& cls = h->classes[rows[b].cls];
        if (!cls.dabplus_rate()) { h->err = "sub-channel bit rate is not a DAB+ rate"; return DABPHY_ERR_INVALID; }
        if (fb && fb != cls.prot.nbits / 8) { h->err = "the ensembles' sub-channels at this position differ in bit rate: use dabphy_superframes_ensemble"; return DABPHY_ERR_INVALID; }
        fb = cls.prot.nbits / 8;
    }
    return superframes_of(h, rows, fb, events, n_events, sf);
}

int dabphy_superframes_ensemble(dabphy_handle* h, uint32_t ensemble, uint32_t subch_index, dabphy_sf_event* events, int32_t* n_events, uint8_t* sf)
{
    DeviceBind dev_(h);
    if (!h || !events || !n_events || !h->last_frames || !h->last_desc || ensemble >= h->cfg.n_ensembles || subch_index >= h->where[ensemble].size()) return DABPHY_ERR_INVALID;
    const std::vector<dabphy_handle::PairRef> rows(1, h->where[ensemble][subch_index]);
    const auto& cls = h->classes[rows[0].cls];
    if (!cls.dabplus_rate()) { h->err = "sub-channel bit rate is not a DAB+ rate"; return DABPHY_ERR_INVALID; }
    return superframes_of(h, rows, cls.prot.nbits / 8, events, n_events, sf);
}

// SuperframeFilter over every DAB+ sub-channel of every ensemble: the classes of a bucket in one launch each on the main stream, totals into sf_stats
int launch_superframe_stats(dabphy_handle* h, hipStream_t st, const FrameDesc* desc, uint32_t n_frames)
{
    const uint32_t B = h->cfg.n_ensembles;
    if (!st) st = h->stream;
    int r;
    if ((r = ensure(h, h->sf_stats, sizeof(int32_t) * 4 * B))) return r;
    HIPCHK(h, hipMemsetAsync(h->sf_stats.p, 0, sizeof(int32_t) * 4 * B, st));
    std::vector<SfSel> sel;
    for (size_t ci = 0; ci < h->classes.size(); ci++) if (h->classes[ci].dabplus_rate()) sel.push_back(SfSel{(int)ci, nullptr, 0});
    if (sel.empty()) return 0;
    if (h->profiling) { hipError_t e = hipEventRecord(h->ev_beg[dabphy_handle::ST_RS], st); (void)e; }
    if ((r = run_superframes(h, sel, h->sf_stats.as<int32_t>(), st, desc, n_frames))) return r;
    if (h->profiling) { hipError_t e = hipEventRecord(h->ev_end[dabphy_handle::ST_RS], st); (void)e; h->ev_used[dabphy_handle::ST_RS] = true; }
    return 0;
}

// dabphy_set_auto_superframes(2).  The filter pass of a batch needs nothing but the batch's class outputs and descriptors and the
// windows it carries; nothing of the NEXT batch needs its results before that batch's decoders overwrite the class outputs.  So the pass of
// batch k is queued by dabphy_process(k + 1), behind that batch's demod launch, on a stream of its own: it runs beside the FFT stage
// instead of in the step's tail, and the decoders of batch k + 1 wait for it on the device (it is long done by then).
int launch_deferred_superframes(dabphy_handle* h)
{
    if (!h->sf_def_pending) return DABPHY_OK;
    const uint32_t B = h->cfg.n_ensembles;
    // (on the AUXILIARY stream, which is idle until this batch's demod kernel has finished: a stream of its own would be the handle's
    // eighth, and the runtime multiplexes streams onto four hardware queues -- the first version shared one with the main stream and ran
    // behind the demod kernel instead of beside it: profiles/r06_step_variants.txt)
    if (!h->rs_stream) {
        h->rs_stream = h->aux_stream;
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_rs_done, hipEventDisableTiming));
    }
    int r;
    if ((r = launch_superframe_stats(h, h->rs_stream, h->sf_def_desc, h->sf_def_frames))) return r;
    launch_copy_out(h->sf_stats.p, h->h_sf_stats, sizeof(int32_t) * 4 * B, h->rs_stream);
    HIPCHK(h, hipEventRecord(h->ev_rs_done, h->rs_stream));
    h->sf_def_pending = false; h->sf_def_inflight = true; h->sf_def_unfetched = true;
    return DABPHY_OK;
}
int flush_deferred_superframes(dabphy_handle* h)
{
    int r;
    if ((r = launch_deferred_superframes(h))) return r;
    if (h->sf_def_inflight) { HIPCHK(h, hipEventSynchronize(h->ev_rs_done)); h->sf_def_inflight = false; }
    return DABPHY_OK;
}

int dabphy_set_auto_superframes(dabphy_handle* h, int32_t on)
{
    DeviceBind dev_(h);
    if (!h) return DABPHY_ERR_INVALID;
    if (h->sf_deferred && on != 2) { int r = flush_deferred_superframes(h); if (r) return r; }      // (leaving the deferred mode: nothing stays pending)
    h->sf_auto = on != 0;
    h->sf_deferred = on == 2;
    return DABPHY_OK;
}

int dabphy_superframes_stats(dabphy_handle* h, int32_t* stats)
{
    DeviceBind dev_(h);
    if (!h || !stats || !h->last_frames || !h->last_desc) return DABPHY_ERR_INVALID;
    int r;
    if (h->sf_auto && h->sf_deferred) {
        // the totals of the pass that ran last and has not been fetched (the batch BEFORE the last dabphy_process); if there is none,
        // the pending pass of the last batch runs now (the end of a stream: one more call fetches the last batch's totals)
        if (!h->sf_def_unfetched && h->sf_def_pending) { if ((r = launch_deferred_superframes(h))) return r; }
        if (h->sf_def_inflight) { HIPCHK(h, hipEventSynchronize(h->ev_rs_done)); h->sf_def_inflight = false; }
        if (h->sf_def_unfetched) memcpy(stats, h->h_sf_stats, sizeof(int32_t) * 4 * h->cfg.n_ensembles);
        else memset(stats, 0, sizeof(int32_t) * 4 * h->cfg.n_ensembles);
        h->sf_def_unfetched = false;
        return DABPHY_OK;
    }
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

} // extern "C"
