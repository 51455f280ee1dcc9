// welle.io_amd/csrc/dabphy_fused.hip -- host side of the fused decode (k_viterbi_fused, k_viterbi.hip): the per-class step tables and the
// launch plan.  No arithmetic of the hot path happens here: the tables restate the depuncturing maps (eep-protection.cpp:32-152,
// uep-protection.cpp:27-239, fic-handler.cpp:158-191) in terms of the kernel's LDS window ring.
#include "dabphy_internal.h"

namespace {

// What every trellis step reads, in terms of a wave's window ring of `rows` rows.  Source byte u of the punctured stream sits in window
// u >> 4 (slot (u >> 4) & 1), column u & 15 -- and, for an MSC class, map16[u & 15] rows below the lane's row base (the time
// de-interleaver, dab-audio.cpp:138-143; the FIC has none).  Returns false when the kernel's window schedule cannot follow the map.
bool step_table(const std::vector<int16_t>& m, int nsteps, int n_in, int rows, bool skew, std::vector<MscStep>& st, int& n_windows, int& why)
{
    static const int map16[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
    const int PITCH = MSC_ROW_PITCH, SLOT = rows * MSC_ROW_PITCH, ZERO = 2 * SLOT;
    constexpr int PADDING = 6;                        // the kernel requests descriptors one block of six steps ahead
    st.assign((size_t)nsteps + PADDING, MscStep{0, 0});
    std::vector<int> wlo((size_t)nsteps, -1), whi((size_t)nsteps, -1);
    for (int q = 0; q < nsteps; q++) {
        uint32_t off[4];
        for (int j = 0; j < 4; j++) {
            const int u = m[4 * q + j];
            if (u < 0) { off[j] = (uint32_t)ZERO; continue; }
            const int w = u >> 4, col = u & 15;
            off[j] = (uint32_t)((w & 1) * SLOT + (skew ? map16[col] : 0) * PITCH + col);
            if (wlo[q] < 0) wlo[q] = w;
            whi[q] = w;
        }
        st[q].off01 = off[0] | (off[1] << 16); st[q].off23 = off[2] | (off[3] << 16);
    }
    for (int q = nsteps; q < nsteps + PADDING; q++) st[(size_t)q].off01 = st[(size_t)q].off23 = (uint32_t)ZERO | ((uint32_t)ZERO << 16);
    n_windows = (n_in + 15) / 16;
    // lowest window any LATER step reads: when it moves up, the window below it has died and its slot takes the window after the next
    std::vector<int> low_after((size_t)nsteps, n_windows);
    for (int q = nsteps - 2, low = n_windows; q >= 0; q--) { if (wlo[q + 1] >= 0) low = wlo[q + 1]; low_after[q] = low; }
    int seen = 1, prev_low = 0;
    std::vector<int> load_step((size_t)n_windows + 2, -10);
    bool ok = nsteps % 6 == 0; why = ok ? 0 : 8;    // (the kernel walks the trellis in blocks of six steps: true for every 24 * bitrate + 6 and for the FIC's 774)
    for (int q = 0; q < nsteps; q++) {
        if (whi[q] > seen) {
            st[q].off01 |= MSC_FIRST_USE; seen = whi[q];
            // the step BEFORE q waits for the window with s_waitcnt vmcnt(2): its load must be older than two decision stores
            if (q - 1 - load_step[seen] < 2) { ok = false; why |= 1; }
        }
        if (whi[q] >= 0 && whi[q] - wlo[q] > 1) { ok = false; why |= 2; }
        if (low_after[q] > prev_low) {
            if (low_after[q] != prev_low + 1 && low_after[q] < n_windows) { ok = false; why |= 4; }
            prev_low = low_after[q];
            if (prev_low + 1 < n_windows) { st[q].off01 |= MSC_LOAD_NEXT; load_step[prev_low + 1] = q; }
        }
    }
    return ok;
}

}  // namespace

extern "C" {

size_t soft_ens_stride(const dabphy_handle* h)
{
    // [max_frames + 5 frame slots][one frame of zeros]: what the fused decode loads for CIFs that do not exist yet sits behind every
    // ensemble's own slice, inside the reach of a buffer resource based at that slice
    return ((size_t)h->cfg.max_frames + 5 + 1) * SOFT_PER_FRAME;
}

int fused_class_tables(dabphy_handle* h, const dabphy_protection& prot, bool fic, DevBuf (&steps)[FUSED_VARIANTS], int (&n_windows)[FUSED_VARIANTS])
{
    const std::vector<int16_t> m = depuncture_map(&prot);
    const int nsteps = prot.nbits + 6, n_in = protection_input_bits(&prot);
    for (int v = 0; v < FUSED_VARIANTS; v++) {
        std::vector<MscStep> st; int nw = 0, why = 0;
        const bool ok = step_table(m, nsteps, n_in, FUSED_ROWS[v], !fic, st, nw, why);
        if (!ok && debug_env("DABPHY_DEBUG")) fprintf(stderr, "dabphy: class nbits %d: no fused decode (window schedule, reason %d)\n", prot.nbits, why);
        n_windows[v] = ok ? nw : 0;                   // (never 0 for the profiles of EN 300 401; the two-kernel path decodes such a class)
        int r;
        if ((r = ensure(h, steps[v], st.size() * sizeof(MscStep)))) return r;
        HIPCHK(h, hipMemcpy(steps[v].p, st.data(), st.size() * sizeof(MscStep), hipMemcpyHostToDevice));
    }
    return DABPHY_OK;
}

// The launch plan of one batch depth: which build of the kernel (how many (ensemble, sub-channel) pairs the 64 code words of a wave can
// span), which classes it decodes, their descriptors and the work list.  Called from dabphy_process's allocation phase (every buffer
// it names exists by then); uploads happen on the handle's stream, in order with the launches that read them, and only when
// something changed.
int fused_plan(dabphy_handle* h, uint32_t F, bool want_fic)
{
    auto& P = h->fplan;
    // (everything below is a function of the batch depth, the class set and the buffers' addresses: buf_gen moves with the last two)
    if (P.valid && P.F == F && P.want_fic == want_fic && P.buf_gen == h->buf_gen && P.tb_split == h->tb_split) return DABPHY_OK;
    const uint32_t B = h->cfg.n_ensembles;
    const int R = 4 * (int)F;
    const int v = R >= FUSED_MIN_CIFS[0] ? 0 : R >= FUSED_MIN_CIFS[1] ? 1 : 2;
    const size_t ens_stride = soft_ens_stride(h);
    // A wave's sources are addressed with 32-bit offsets from the ring slice of its first ensemble.  Its code words span at most
    // nseg consecutive (ensemble, sub-channel) pairs of the class's table (64 + 15 nseg <= rows) -- from the first pair's ensemble to the
    // last one's, however many ensembles without a sub-channel of this class lie between; the FIC's 64 code words = 16 frames,
    // ceil(16 / F) + 1 ensembles.  A class whose span leaves the 4 GiB takes the two-kernel path (64-bit addresses).
    const int nseg = (FUSED_ROWS[v] - 64) / 15;
    auto reach_ok = [&](int n_ens_spanned) { return (uint64_t)n_ens_spanned * ens_stride <= 0xffffffffull; };
    auto ens_span = [&](const std::vector<MscPair>& pr) {
        int span = 1;
        for (size_t p = 0; p < pr.size(); p++) { const size_t q = std::min(pr.size() - 1, p + (size_t)nseg - 1); span = std::max(span, pr[q].ens - pr[p].ens + 1); }
        return span;
    };
    std::vector<FusedClass> cls; std::vector<int> idx;
    struct Item { int nsteps, ci, n_groups; };
    std::vector<Item> items;
    size_t max_steps = 0;
    // A small batch -- fewer code words than the device has lanes to give them -- is decoded STATE-PARALLEL: one wavefront per code word
    // (k_viterbi_sp.hip), every class and the FIC, whatever their window schedules and spans (it addresses with 64-bit pointers).
    uint64_t total_cw = (uint64_t)B * F * 4 * (want_fic && h->fused_fic ? 1 : 0);
    bool sp_ok = h->fused_msc && h->cfg.decode_shape != 1 && (h->sp_max_codewords > 0 || h->cfg.decode_shape >= 2);
    for (auto& c : h->classes) { total_cw += (uint64_t)4 * F * c.pairs.size(); sp_ok = sp_ok && (c.prot.nbits + 6) % 6 == 0 && c.prot.nbits + 6 <= SP_MAXSTEPS[SP_VARIANTS - 1]; }
    // ... and its decision scratch is sized per code word from the LONGEST code word of the launch (a 256-byte history row per 30 steps):
    // one 384 kbit/s class among small ones makes that 79 KB per code word, 3.2 GB at the code word limit.  The automatic choice therefore
    // also asks for the scratch to stay below a GiB (the lane-per-code-word kernel sizes its scratch per group: dec_by_item)
    size_t longest = want_fic && h->fused_fic ? 774 : 0;
    for (auto& c : h->classes) longest = std::max(longest, (size_t)c.prot.nbits + 6);
    const uint64_t sp_scratch = total_cw * (uint64_t)(longest / 30 + 1) * 32 * sizeof(uint2);
    const bool use_sp = sp_ok && (h->cfg.decode_shape >= 2 || (total_cw <= h->sp_max_codewords && sp_scratch <= ((uint64_t)1 << 30)));
    for (size_t i = 0; i < h->classes.size(); i++) {
        auto& c = h->classes[i];
        const int P = (int)c.pairs.size();
        if (!h->fused_msc || (!use_sp && (c.n_windows[v] <= 0 || !reach_ok(ens_span(c.pairs))))) continue;
        FusedClass fc{};
        fc.steps = c.steps[v].as<MscStep>(); fc.pairs = c.pair_tab.as<MscPair>(); fc.map = c.map.as<int16_t>(); fc.out = c.out.as<uint8_t>();
        fc.nbits = c.prot.nbits; fc.nsteps = fc.nbits + 6; fc.n_windows = c.n_windows[v]; fc.n_cw = (int32_t)(4 * F * (uint32_t)P);
        fc.n_pairs = P; fc.kind = 0; fc.dedisperse = 1;
        items.push_back({fc.nsteps, (int)cls.size(), (fc.n_cw + 63) / 64});
        cls.push_back(fc); idx.push_back((int)i);
        max_steps = std::max(max_steps, (size_t)fc.nsteps);
    }
    bool fic_in = false;
    if (want_fic && h->fused_fic && (use_sp || (h->fic_windows[v] > 0 && reach_ok((16 + (int)F - 1) / (int)F + 1)))) {
        FusedClass fc{};
        fc.steps = h->fic_steps[v].as<MscStep>(); fc.pairs = nullptr; fc.map = h->d_fic_map; fc.out = h->s_fib.as<uint8_t>();
        fc.nbits = 768; fc.nsteps = 774; fc.n_windows = h->fic_windows[v]; fc.n_cw = (int32_t)(B * F * 4);
        fc.n_pairs = 1; fc.kind = 1; fc.dedisperse = 1;
        items.push_back({fc.nsteps, (int)cls.size(), (fc.n_cw + 63) / 64});
        cls.push_back(fc);
        max_steps = std::max(max_steps, (size_t)fc.nsteps);
        fic_in = true;
    }
    if (cls.size() > 255) { h->err = "more than 255 protection classes"; return DABPHY_ERR_INVALID; }
    // longest code words first: the waves that pull the long groups start them while every slot is still busy, the short ones fill the end
    std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.nsteps > b.nsteps; });
    std::vector<uint32_t> work, item_off;
    uint64_t item_rows = 0;                          // decision rows ([64 lanes] cells) of all groups together
    for (const Item& it : items) {
        if (it.n_groups > 0xffffff) { h->err = "class too large for the fused decode's work list"; return DABPHY_ERR_INVALID; }
        for (int g = 0; g < it.n_groups; g++) { work.push_back(((uint32_t)it.ci << 24) | (uint32_t)g); item_off.push_back((uint32_t)item_rows); item_rows += (uint64_t)it.nsteps; }
    }
    const int sp_variant = sp_variant_for((int)max_steps);
    const bool sp_two = use_sp && sp_two_for(h, total_cw);
    const int slots = fused_wave_slots(v);
    // (state-parallel: one work-group per code word slot -- or pair of slots -- of every listed group, each with its own decision scratch)
    const int n_slots = use_sp ? (int)work.size() * (sp_two ? 32 : 64) : (int)std::min<size_t>(work.size(), (size_t)slots);
    const size_t sp_cells = (max_steps / 30 + 1) * (sp_two ? 64 : 32);                 // one 256-byte row of history words per 30 steps and code word
    // Decision scratch of the lane-per-code-word kernel: one region per WORK-GROUP sized for the longest code word of the launch (reused by
    // every group the wave pulls: the headline's shape) -- unless a few very long code words ride among many short ones (one 384 kbit/s
    // service in a multiplex of small ones: 9222 steps x 5120 waves = 24 GB): then one region per GROUP, each of its own length.
    // (the traceback as a pass of its own reads a group's decisions after the wave that wrote them has gone on to the next group: per group)
    const bool tb_split = h->tb_split && !use_sp && item_rows < 0xffffffffull && !work.empty();
    const bool dec_by_item = !use_sp && (tb_split || debug_env("DABPHY_FORCE_DEC_BY_ITEM") || item_rows < (uint64_t)n_slots * max_steps) && item_rows < 0xffffffffull;
    int r;
    if (!work.empty()) {
        const size_t dec_cells = use_sp ? (size_t)n_slots * sp_cells : dec_by_item ? (size_t)item_rows * 64 : (size_t)n_slots * max_steps * 64;
        if ((r = ensure(h, h->vdec, dec_cells * sizeof(uint2)))) return r;
        if (dec_by_item && (r = ensure(h, h->fused_dec_off, item_off.size() * sizeof(uint32_t)))) return r;
        if (tb_split) {
            if ((r = ensure(h, h->fused_done, (work.size() + 2) * sizeof(uint32_t)))) return r;
            if (!h->tb_stream) {
                HIPCHK(h, hipStreamCreateWithFlags(&h->tb_stream, hipStreamNonBlocking));
                HIPCHK(h, hipEventCreateWithFlags(&h->ev_tb_fork, hipEventDisableTiming));
                HIPCHK(h, hipEventCreateWithFlags(&h->ev_tb_join, hipEventDisableTiming));
            }
        }
        if ((r = ensure(h, h->fused_cls, cls.size() * sizeof(FusedClass)))) return r;
        if ((r = ensure(h, h->fused_work, work.size() * sizeof(uint32_t)))) return r;
        if (!h->d_fused_next) {
            void* p = nullptr;
            if (hipMalloc(&p, sizeof(uint32_t)) != hipSuccess) { h->err = "hipMalloc failed (work cursor)"; return DABPHY_ERR_NOMEM; }
            h->owned.push_back(p); h->d_fused_next = reinterpret_cast<uint32_t*>(p);
        }
    }
    const bool same_cls = P.valid && P.buf_gen == h->buf_gen && P.host_cls.size() == cls.size() &&
                          (cls.empty() || !memcmp(P.host_cls.data(), cls.data(), cls.size() * sizeof(FusedClass)));
    const bool same_work = P.valid && P.buf_gen == h->buf_gen && P.host_work == work && P.dec_by_item == dec_by_item;
    // (an earlier upload may still be reading the plan's host vectors -- a pageable source is not always staged before the call returns --:
    // nothing is replaced under it.  The stream is idle here except inside a call that plans twice.)
    if (P.valid && (!same_cls || !same_work)) HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!same_cls) {
        P.host_cls = cls;                              // (the plan's own storage: it outlives the copy, which the stream orders before the launch)
        if (!cls.empty()) HIPCHK(h, hipMemcpyAsync(h->fused_cls.p, P.host_cls.data(), cls.size() * sizeof(FusedClass), hipMemcpyHostToDevice, h->stream));
    }
    if (!same_work) {
        P.host_work = work; P.host_dec_off = item_off;
        if (!work.empty()) HIPCHK(h, hipMemcpyAsync(h->fused_work.p, P.host_work.data(), work.size() * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
        if (!work.empty() && dec_by_item) HIPCHK(h, hipMemcpyAsync(h->fused_dec_off.p, P.host_dec_off.data(), item_off.size() * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    }
    if (debug_env("DABPHY_DEBUG")) fprintf(stderr, "dabphy: decode plan for %u frames per call: %s, %zu of %zu classes%s, %zu groups, %d work-groups, decision scratch %s\n", F, use_sp ? (sp_two ? "state-parallel, two code words per wavefront" : "state-parallel") : "lane-per-code-word", idx.size(), h->classes.size(), fic_in ? " + FIC" : "", work.size(), n_slots, dec_by_item ? "per group" : "per work-group");
    P.valid = true; P.F = F; P.want_fic = want_fic; P.fic_in = fic_in; P.variant = v; P.n_slots = n_slots;
    P.use_sp = use_sp; P.sp_variant = sp_variant; P.sp_two = sp_two; P.dec_by_item = dec_by_item; P.tb_split = h->tb_split;
    P.dec_slot_cells = use_sp ? sp_cells : max_steps * 64;
    P.class_idx = idx; P.buf_gen = h->buf_gen;
    FusedArgs a{};
    a.soft = h->s_soft.as<int8_t>(); a.ens_stride = ens_stride; a.soft_ring = (int)h->cfg.max_frames + 5; a.n_ens = (int)B; a.n_frames = (int)F;
    a.desc = nullptr;                                  // (set per batch: the descriptor buffers rotate)
    a.cls = h->fused_cls.as<FusedClass>(); a.work = h->fused_work.as<uint32_t>(); a.n_work = (uint32_t)work.size(); a.next = h->d_fused_next;
    a.dec = h->vdec.as<uint2>(); a.dec_slot_cells = P.dec_slot_cells; a.prbs_words = h->d_prbs_words;
    a.dec_off = dec_by_item ? h->fused_dec_off.as<uint32_t>() : nullptr;
    if (tb_split) { a.done = h->fused_done.as<uint32_t>(); a.next_tb = a.done + work.size(); }
    a.sp2_warm = (int)h->sp2_tb_warm; a.sp2_resident = (int)h->sp2_tb_resident;
    P.args = a;
    return DABPHY_OK;
}

// Whether a stand-alone class of n_cw code words (a seam's call, the replay's one-frame FIC) is decoded state-parallel: the same rule
// as for a batch (dabphy_config.decode_shape, the code word limit), and the kernel's own conditions (blocks of six steps, its LDS sizes).
bool sp_single_ok(const dabphy_handle* h, uint64_t n_cw, int nsteps)
{
    if (!h->fused_msc || h->cfg.decode_shape == 1 || nsteps % 6 != 0 || nsteps > SP_MAXSTEPS[SP_VARIANTS - 1]) return false;
    if (n_cw > (uint64_t)SP_SINGLE_MAX_GROUPS * 64) return false;        // beyond the one-class launch's work list: the two-kernel path decodes it
    return h->cfg.decode_shape >= 2 || (h->sp_max_codewords > 0 && n_cw <= h->sp_max_codewords);
}

// One class through k_viterbi_sp on stream st: descriptor and work list go up through a small page-locked staging area (the copies are
// ordered with the launch by the stream; the caller does not reuse the staging before it has synchronised or queued its last launch of
// this class), decision scratch = the handle's (the stream orders it with every other user).  `a` carries what the class kind reads
// (soft / desc / strides, or lin_in / lin_stride); cls, work, dec are filled in here and returned in `a` for further launches.
// every buffer such a launch needs, so that a caller in the middle of a submission (the replay inside dabphy_process) allocates nothing
int sp_single_reserve(dabphy_handle* h, uint64_t n_cw, int nsteps)
{
    const uint64_t n_groups = (n_cw + 63) / 64;
    int r;
    if (n_groups > (uint64_t)SP_SINGLE_MAX_GROUPS) { h->err = "one-class state-parallel launch: too many groups"; return DABPHY_ERR_INVALID; }
    if (!h->h_sp1) {
        void* p = nullptr;
        if (hipHostMalloc(&p, sizeof(FusedClass) + SP_SINGLE_MAX_GROUPS * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) { h->err = "hipHostMalloc failed (one-class staging)"; return DABPHY_ERR_NOMEM; }
        h->h_sp1 = p;
    }
    const size_t cells = ((size_t)nsteps / 30 + 1) * 32;                 // (per code word: the same for either kernel)
    if ((r = ensure(h, h->sp1_cls, sizeof(FusedClass)))) return r;
    if ((r = ensure(h, h->sp1_work, SP_SINGLE_MAX_GROUPS * sizeof(uint32_t)))) return r;
    return ensure(h, h->vdec, (size_t)n_groups * 64 * cells * sizeof(uint2));
}
int sp_single_prepare(dabphy_handle* h, const FusedClass& fc, FusedArgs& a, hipStream_t st)
{
    const uint32_t n_groups = (uint32_t)((fc.n_cw + 63) / 64);
    int r;
    if ((r = sp_single_reserve(h, (uint64_t)fc.n_cw, fc.nsteps))) return r;
    h->sp1_two = sp_two_for(h, (uint64_t)fc.n_cw);
    const size_t cells = ((size_t)fc.nsteps / 30 + 1) * (h->sp1_two ? 64 : 32);
    FusedClass* hc = reinterpret_cast<FusedClass*>(h->h_sp1);
    uint32_t* hw = reinterpret_cast<uint32_t*>(hc + 1);
    *hc = fc;
    for (uint32_t g = 0; g < n_groups; g++) hw[g] = g;                 // class 0, group g
    HIPCHK(h, hipMemcpyAsync(h->sp1_cls.p, hc, sizeof(FusedClass), hipMemcpyHostToDevice, st));
    HIPCHK(h, hipMemcpyAsync(h->sp1_work.p, hw, n_groups * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    a.cls = h->sp1_cls.as<FusedClass>(); a.work = h->sp1_work.as<uint32_t>(); a.n_work = n_groups; a.next = nullptr;
    a.dec = h->vdec.as<uint2>(); a.dec_slot_cells = cells; a.prbs_words = h->d_prbs_words;
    a.sp2_warm = (int)h->sp2_tb_warm; a.sp2_resident = (int)h->sp2_tb_resident;
    return DABPHY_OK;
}
// Two code words per wavefront (k_viterbi_sp2: half the vector instructions per code word) pays once the code words outnumber the SIMDs
// a few times over; below that the launch runs at the latency of ONE wave, and a wave that gathers and walks back two code words takes
// longer than one that does it for one (profiles/r05_viterbi_sp2.txt).  decode_shape 2 / 3 force either kernel.
bool sp_two_for(const dabphy_handle* h, uint64_t n_cw)
{
    if (h->cfg.decode_shape == 3) return false;
    if (h->cfg.decode_shape == 2) return true;
    return n_cw > h->sp2_min_codewords;
}
void launch_sp(const FusedArgs& a, bool two, int lds_variant, hipStream_t s)
{
    if (two) launch_viterbi_sp2(a, lds_variant < SP2_VARIANTS ? lds_variant : SP2_VARIANTS - 1, s); else launch_viterbi_sp(a, lds_variant, s);
}
int sp_variant_for(int nsteps)
{
    int v = 0;
    while (v + 1 < SP_VARIANTS && SP_MAXSTEPS[v] < nsteps) v++;
    return v;
}

} // extern "C"
