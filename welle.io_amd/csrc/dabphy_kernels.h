// welle.io_amd/csrc/dabphy_kernels.h -- device-side data structures and kernel argument blocks.
#pragma once
#include "dabphy_common.h"

namespace dabphy {

// Receiver state of one ensemble (the members of OFDMProcessor / OfdmDecoder / FicHandler that survive from
// frame to frame: ofdm-processor.h:95-110, ofdm-decoder.h:75-77, fic-handler.h:64).  Lives in HBM.
struct RxState {
    int64_t pos;            // absolute index of the next unread sample (ring index = pos % ring)
    int64_t frame_no;       // frames demodulated so far (selects the soft-bit ring slot; CIF count = 4*frame_no)
    int32_t local_phase;    // OFDMProcessor::localPhase, 0..INPUT_RATE-1
    int32_t coarse;         // coarseCorrector [Hz]
    int32_t fine;           // fineCorrector [Hz], int16 range
    int32_t synced;         // 0: acquisition needed (notSynced), 1: tracking (SyncOnPhase loop)
    float s_level;          // OFDMProcessor::sLevel: advanced by the acquisition kernel sample by sample; the samples pulled while
                            // tracking are replayed from `hist` when lock is lost (the level is only read by the acquisition)
    int32_t lost;           // number of findIndex failures seen
    int32_t n_exact_sums;   // frames whose fine corrector needed the ordered float sums (k_sync_finish's interval test was undecided)
    int32_t hist_count, hist_head; // window searches since the last acquisition whose samples sLevel has not seen yet: ring entries head .. head+count-1 of `hist`
    int32_t hist_dropped;   // 1: older entries were overwritten (or their samples have left the ring): the replay cannot start from an exact level
    int32_t n_relock_inexact; // re-acquisitions that started from a level the replay could not certify
    int32_t attempts;       // entries into the notSynced state since reset (ofdm-processor.cpp:256-262: what scan mode counts)
    int32_t first_lock_attempts; // `attempts` when the first window search succeeded (:351-355 onSignalPresence(true)); -1 = not yet
    int32_t n_wide_frames;  // frames accepted from the wide (all frames of a batch at once) synchroniser pass
    int32_t n_chain_frames; // ... of which the window search ran in the find chain (k_sync_find_chain: the true window positions, correctors predicted)
    int32_t calm_frames;    // consecutive tracked frames whose window index was T_g (saturating): below SYNC_CALM_MIN the ensemble's window is
                            // moving (a sampling-clock offset, ofdm-processor.cpp:337-350) and the wide pass leaves its searches to the find chain
    // acquisition state machine (survives a call that ran out of samples mid-search)
    int32_t acq_phase;      // 0 priming sLevel, 1 first 50 samples, 2 looking for the dip, 3 looking for the end of the null
    int32_t acq_counter, acq_idx, acq_left;
    float acq_cs;           // currentStrength
    float env[64];          // last 64 entries of envBuffer (only the 50 most recent are ever read)
};

// Decoder-side state (written by the decode kernels, which may run concurrently with the NEXT batch's
// synchroniser on another stream -- hence not part of RxState): OfdmDecoder::snr/snrCount (ofdm-decoder.h:75-77)
// and FicHandler::fic_decode_success_ratio (fic-handler.h:64).
struct DecState {
    int32_t fic_ratio;      // saturating 0..10 FIB CRC success counter
    int32_t snr_count;      // OfdmDecoder::snrCount
    float snr;              // OfdmDecoder::snr
    int32_t stale_ratio_frames;   // frames (since reset) whose coarse-corrector decision -- made by the synchroniser ahead of the decoder from
                                  // the ratio of an earlier batch -- differs from the one the reference makes with the ratio after the
                                  // previous frame (ofdm-processor.cpp:397): from the first such frame on this ensemble may deviate
    int64_t first_stale_frame;    // its frame number, -1: none
    // ... of which those that can have changed anything: the corrector was consulted although the reference would not have AND it moved
    // coarseCorrector (consulting it has no other effect, ofdm-processor.cpp:397-409), or it was NOT consulted although the reference
    // would have (what it would have returned is not known).  Zero = this ensemble's output is the reference's, frame for frame.
    int32_t effective_stale_frames; int32_t pad_;
    int64_t first_effective_frame;
};

// Where one transmission frame sits in the sample stream and which oscillator settings were in force while
// its samples were pulled (written by the sync kernel, read by the demod kernel).
struct FrameDesc {
    int64_t pos;            // absolute sample index of the T_u-sample sync buffer (ofdm-processor.cpp:337)
    int64_t frame_no;
    int32_t start_index;    // PhaseReference::findIndex result; the PRS useful part starts at pos + start_index
    int32_t L0;             // localPhase before the first sample at pos
    int32_t f_prs;          // coarse+fine while the PRS was pulled
    int32_t L1;             // localPhase after the PRS (before the first sample of symbol 1)
    int32_t f_sym;          // coarse+fine while symbols 1..75 were pulled
    int32_t valid;          // 0: no frame (not synchronised / not enough samples), 1: demodulated, 2: pending (inside the chain), 3: window search failed
    int32_t coarse_ran;     // 1: the coarse corrector was consulted for this frame (FIC ratio as the synchroniser knew it was < 50)
    int32_t fine_after, coarse_after;   // correctors after the frame (reported like onFrequencyCorrectorChange)
    int32_t null_L, null_f;             // oscillator state while the trailing null symbol was pulled (onNewNullSymbol)
    int32_t exact_sums;                 // 1: the fine corrector of this frame needed the ordered float sums
    uint32_t osc_hazard[3];             // bit s: the useful part of symbol s (0 = PRS) reads an oscillator table entry for which k_demod
                                        // must take the checked conversion (osc_exact.h: osc_hazard_entry); written with the descriptor
    int32_t coarse_step;                // what the coarse corrector added to coarseCorrector in this frame [Hz] (0: not consulted, or no correction)
};
static_assert(sizeof(FrameDesc) == 80, "FrameDesc layout");

// Argument block of the synchronisation kernels (k_sync.hip)
struct SyncArgs {
    Tables tab;
    const cf32* iq; size_t iq_stride; int64_t ring;     // sample ring of each ensemble
    int64_t n_valid;                                     // samples written so far (absolute); ignored when loop != 0
    int loop;                                            // the ring is a looping recording (CRAWFile with rewind, raw_file.cpp:284-286)
    RxState* state; const DecState* dec; FrameDesc* desc; int n_ens, n_frames, frame;
    int fft_placement, disable_coarse, freqsync;        // FFTPlacementMethod, disableCoarseCorrector, FreqsyncMethod (reference numbering)
    float* cir;                                          // optional [B][n_frames][2048] impulse responses
    FrameDesc* hist; int hist_cap;                       // [B][hist_cap] ring of the window searches since the last acquisition (sLevel replay)
    float level_max;                                     // upper bound of OFDMProcessor::sLevel for this stream (3e38: unknown): where the bracketing replay starts from above
    const int32_t* redo_from;                            // serial chain: [B] first frame slot the wide pass did not settle (nullptr: no wide pass ran)
    int32_t* redo_out; int32_t* any_redo;                // k_sync_validate: [B] and a flag (page-locked HOST memory, written by the last judge of the pass: no copy stands between the verdict and the host)
    int32_t* any_chain;                                  // k_sync_validate_chain: raised (page-locked host memory) when some ensemble's window index moved within its last SYNC_CALM_MIN frames
    int skip_wide;                                       // launch_sync_wide: no wide searches / sums / judge (every ensemble starts in the find chain: redo_out = 0)
    int last_round;                                      // k_sync_validate_chain: this is the pass's last judge (it raises any_redo for what is left)
};

struct DemodArgs {
    Tables tab;
    const cf32* iq; size_t iq_stride; int64_t ring;     // per-ensemble sample ring: iq + b*iq_stride, `ring` samples long
    const FrameDesc* desc; int n_frames;
    int chunk_len;                                        // data symbols per work-group
    int mix;                                              // 1: apply the NCO (streaming path); 0: input already mixed
    int8_t* soft; int soft_ring;                          // [B][soft_ring (+ 1 zero frame in the streaming receiver)][75][3072]
    size_t soft_ens_stride;                               // bytes between the ring slices of two ensembles (0: soft_ring * SOFT_PER_FRAME)
    cf32* con;                                            // optional [B][n_frames][1200]
    float* prs_mag;                                       // optional [B][n_frames][2048]
    unsigned long long* osc_stats;                        // optional [2]: symbols mixed with the unchecked / the checked oscillator conversion
    int frame_first, frame_count, chunk_count;            // launch only frames [frame_first, frame_first + frame_count) and the first chunk_count chunks of a frame; 0 = all
};

struct SnrArgs {
    DecState* state; const FrameDesc* desc; int n_ens, n_frames;
    const float* prs_mag; float* snr_out;                 // [B][n_frames], NaN = no report
};

// ---------------------------------------------------------------------------------------------- Viterbi
// A "class" is a set of codewords with identical length and puncturing, decoded 64 per wavefront
// (lane = codeword).  Symbols are stored step-major so that each trellis step is one coalesced dword
// load per lane: sym[(group*nsteps + step)*64 + lane] = 4 x uint8 (viterbi.cpp:233-238 mapping applied).
struct VitClass {
    int nbits;              // decoded bits per codeword (768 FIC, 24*bitrate MSC)
    int nsteps;             // nbits + 6
    int n_cw;               // codewords in the class
    int n_groups;           // ceil(n_cw / 64)
    int g_begin, g_end;     // groups [g_begin, g_end) are handled by this launch (a class can be decoded in several parts on different streams)
    uint32_t* sym;          // [n_groups][nsteps][64]
    uint2* dec;             // [n_groups][nsteps][64] decision words (scratch)
    uint8_t* out;           // [n_cw][nbits/8] decoded bytes, energy dispersal removed when prbs != 0
    int dedisperse;
};

struct VitArgs { VitClass c; const uint32_t* prbs_words; };

// Gather for the FIC: soft bits of symbols 1..3 of frame (b,f) -> 4 codewords, depunctured (fic-handler.cpp:158-191)
struct FicGatherArgs {
    const int8_t* soft; int soft_ring; size_t frame_stride;   // bytes between frame slots (SOFT_PER_FRAME in the ring)
    size_t soft_ens_stride;                                   // bytes between ensembles (0: soft_ring * frame_stride)
    const FrameDesc* desc; int n_ens, n_frames;
    const int16_t* map;     // [3096] mother-code index -> index into the 2304 punctured bits, -1 = erasure
    VitClass c;
    int frame_sel;          // 0: every frame of the batch, codeword (b F + f) 4 + q; f + 1: frame f only, codeword 4 b + q (c.n_cw = 4 B)
};

// One (ensemble, sub-channel) PAIR of an MSC protection class.  Every ensemble of a batch selects its own sub-channels
// (MscHandler::addSubchannel, msc-handler.cpp:61-103, is per receiver): a class is the list of the pairs of the whole batch that share
// one protection profile, ordered by ensemble, then by position in that ensemble's list.  Code word of a class: cw = pair * R + r
// (CIF r of this batch, R = 4 * n_frames).
struct MscPair {
    int32_t ens;            // ensemble of the batch
    int32_t start_bit;      // Subchannel::startAddr * 64: first soft bit of the sub-channel inside a CIF
    int64_t cif0;           // CIF count of the ensemble (4 * frame_no) from which this sub-channel's time de-interleaver has been fed: a
                            // sub-channel added in mid-stream emits its first logical frame 16 CIFs later (dab-audio.cpp:146-149).
                            // -1: not known yet -- k_pair_cif0 sets it from the first batch decoded with the pair
    int32_t idx;            // position in its ensemble's sub-channel list (the subch_index of the getters)
    int32_t pad_;
};
static_assert(sizeof(MscPair) == 24, "MscPair layout");
void launch_pair_cif0(MscPair* pairs, int n_pairs, const FrameDesc* desc, int n_frames, hipStream_t s);

// Gather for one MSC sub-channel class: time de-interleave (dab-audio.cpp:138-143) + depuncture
// (eep-protection.cpp:127-148) fused into one indexed read of the soft-bit ring.
struct MscGatherArgs {
    const int8_t* soft; int soft_ring; const RxState* state; int n_ens, n_frames;
    size_t soft_ens_stride;   // bytes between ensembles (0: soft_ring * SOFT_PER_FRAME)
    const int16_t* map;     // [4*nbits+24] -> index into the sub-channel's length*64 soft bits, -1 = erasure
    const MscPair* pairs;     // [n_pairs] the (ensemble, sub-channel) pairs of the class
    const int32_t* tiles;     // [ceil(nsteps/56)][2]: first source byte (4-aligned) and dwords per row of each step tile
    int n_pairs;
    const FrameDesc* desc;    // [B][n_frames]; desc[b][0].frame_no is the first frame of this batch
    VitClass c;
};

// Fused decode (k_viterbi_fused): the MSC gather (time de-interleave + depuncture) and the FIC gather inside the Viterbi kernel.
// MscStep = what one trellis step needs, identical for all code words of a class: byte offsets of its four soft bits in the wave's
// LDS window ring (relative to the lane's row base; an erasure points into the zero slot), whether it is the first step to read a
// window, and which window to load once it has read.
// (8 bytes per step: off0 | off1 << 16, off2 | off3 << 16 with two flags in the spare top bits -- bit 15 of off01: first step to read
// a new window, wait for its load; bit 31 of off01: once this step has read, load the next window (they are loaded in order 2, 3, ...))
struct MscStep { uint32_t off01, off23; };
// geometry of the window ring (k_viterbi.hip: FM_*): ROWS rows of 20 bytes per slot (16 window bytes + 4 bytes of padding: a pitch of
// five dwords keeps the byte reads of consecutive rows on different LDS banks), slots 0 / 1 = windows, slot 2 = zeros (erasures).
// ROWS depends on how many (ensemble, sub-channel) pairs the 64 code words of a wave can span, i.e. on the batch depth: the kernel is
// built for three row counts and the per-step tables (whose offsets contain the slot size) once per variant.
constexpr int MSC_ROW_PITCH = 20;
constexpr int FUSED_VARIANTS = 3;
constexpr int FUSED_ROWS[FUSED_VARIANTS] = {96, 144, 324};      // >= 64 + 15 * segments: 2 segments (>= 64 CIFs per batch), 5 (>= 16), 17 (>= 4)
constexpr int FUSED_MIN_CIFS[FUSED_VARIANTS] = {64, 16, 4};
constexpr uint32_t MSC_FIRST_USE = 1u << 15, MSC_LOAD_NEXT = 1u << 31, MSC_OFF_MASK = 0x7fffu;
// One class of a fused launch (read through the constant address space).  kind 0 = an MSC protection class: code word
// cw = pair * R + r ((ensemble, sub-channel) pair of the class's table, CIF r of this batch, R = 4 * n_frames); kind 1 = the FIC:
// code word (b * n_frames + f) * 4 + q -- or, k_viterbi_sp only, 4 b + q of frame FusedArgs::fic_frame_sel - 1 alone (the replay of exact
// batch mode); kind 2 (k_viterbi_sp only) = code words that lie one after the other in a plain array (the Viterbi::deconvolve /
// Protection::deconvolve seams): FusedArgs::lin_in + cw * lin_stride, depunctured through `map` when there is one.
struct FusedClass {
    const MscStep* steps;     // [nsteps + 6] for the launch's row-count variant
    const MscPair* pairs;     // MSC: [n_pairs] (ensemble, start bit) of every pair, ensembles ascending
    const int16_t* map;       // [4 * nsteps] mother-code index -> index into the class's punctured bit stream, -1 = erasure (k_viterbi_sp gathers by it)
    uint8_t* out;             // [n_cw][nbits / 8]
    int32_t nsteps, nbits, n_windows, n_cw, n_pairs, kind, dedisperse, reserved_;
};
static_assert(sizeof(FusedClass) == 64, "FusedClass layout");
struct FusedArgs {
    const int8_t* soft; size_t ens_stride; int soft_ring; int n_ens, n_frames;
    const FrameDesc* desc;
    const FusedClass* cls;
    const uint32_t* work; uint32_t n_work;     // work list: class << 24 | group of 64 code words, longest code words first
    uint32_t* next;                            // its dynamic cursor (zeroed by the launcher)
    uint2* dec; size_t dec_slot_cells;         // decision scratch of work-group i: dec + i * dec_slot_cells ([step][64 lanes] cells)
    const uint32_t* dec_off;                   // k_viterbi_fused: nullptr, or the scratch goes by work ITEM: item i at dec + dec_off[i] * 64 cells
    const uint32_t* prbs_words;
    // k_viterbi_fused with the traceback as a pass of its own BESIDE the forward pass (nullptr: the wave that ran a group's trellis walks
    // it back itself): done[item] is raised when a group's decisions are complete, next_tb is the cursor of the waves that walk back
    uint32_t* done; uint32_t* next_tb;
    // k_viterbi_sp's one-class launches (dabphy_fused.hip: sp_single): the replay's one-frame FIC, the linear seams
    int fic_frame_sel;                         // kind 1: 0 = every frame of the batch; f + 1 = frame f only (n_cw = 4 B)
    size_t fic_frame_stride;                   // kind 1: bytes between frame slots of `soft` (0: SOFT_PER_FRAME, the ring)
    const int8_t* lin_in; size_t lin_stride;   // kind 2
    int sp2_warm;                              // k_traceback_sp2: history blocks (30 steps each) a stretch's walk runs in over before its own blocks
    int sp2_resident;                          // k_traceback_sp2: groups of 64 code words up to which a launch cuts the code word into four stretches (above: three)
};
// variant = index into FUSED_ROWS; n_slots = work-groups (one wave each) to launch
// split: the stream the traceback waves are launched on and the two events that fork it off `s` and join it back (a.done != nullptr)
struct FusedSplit { hipStream_t tb_stream; hipEvent_t fork, join; };
void launch_viterbi_fused(const FusedArgs& a, int variant, int n_slots, hipStream_t s, const FusedSplit* split = nullptr);
int fused_wave_slots(int variant);             // resident waves of that variant on the current device
// State-parallel decode of the same work (k_viterbi_sp.hip): one wavefront per CODE WORD, lanes = the 64 trellis states -- the shape for
// batches too small to fill the device with 64-code-word waves.  lds_variant = index into SP_MAXSTEPS (the longest code word of the launch);
// one work-group per code word slot: a.n_work * 64 of them; a.dec holds a.dec_slot_cells decision words per work-group.
constexpr int SP_VARIANTS = 3;
constexpr int SP_MAXSTEPS[SP_VARIANTS] = {1542, 3078, 9222};     // <= 64, <= 128, <= 384 kbit/s (24 * bitrate + 6 trellis steps)
void launch_viterbi_sp(const FusedArgs& a, int lds_variant, hipStream_t s);
void launch_selftest_pair_exchange(unsigned* out, hipStream_t s);
// ... two code words per wavefront (k_viterbi_sp2.hip): half the instructions per code word; one work-group per PAIR of code word slots:
// a.n_work * 32 of them, a.dec_slot_cells = (steps / 30 + 1) * 64 per work-group.  Its LDS table holds four 16-bit sums per trellis
// step and code word for 480 steps at a time: code words of any length pass through it chunk by chunk (lds_variant is ignored)
constexpr int SP2_VARIANTS = 2;
void launch_viterbi_sp2(const FusedArgs& a, int lds_variant, hipStream_t s);
void launch_selftest_half_exchange(unsigned* out, hipStream_t s);

struct CrcArgs {
    const uint8_t* fib;     // [B][F][12][32]
    uint8_t* ok;            // [B][F][12]
    DecState* state; const FrameDesc* desc; int n_ens, n_frames;
    int disable_coarse;     // RadioReceiverOptions::disableCoarseCorrector as the synchroniser used it (k_fic_ratio checks the ratio it saw)
    int frame_first, frame_count;   // k_fic_ratio: walk only these frame slots (0 = all)
    int32_t* any_effective; // k_fic_ratio: set to 1 when a stale decision with a possible effect is found (optional)
    int frame_sel;          // k_fib_crc: 0: fib = [B][F][12][32]; f + 1: fib = [B][12][32] holds frame f only (flags still land in ok[b][f][.])
};

// Gather from a plain [n_cw][in_stride] array of soft bits (the Viterbi::deconvolve / Protection::deconvolve seams)
struct LinGatherArgs {
    const int8_t* in; size_t in_stride;
    const int16_t* map;     // nullptr: input is already depunctured (index = 4*step + j)
    VitClass c;
};

// Null symbols of the demodulated frames as OFDMProcessor hands them to onNewNullSymbol (k_ingest.hip: k_null_symbols)
struct NullArgs {
    Tables tab; const cf32* iq; size_t iq_stride; int64_t ring;
    const FrameDesc* desc; int n_frames;
    cf32* out;                                            // [B][n_frames][T_NULL], zeros for frames that were not demodulated
};
void launch_null_symbols(const NullArgs& a, int n_ens, hipStream_t s);

// TII (k_tii.hip): TIIDecoder::run + analyse_phase (tii-decoder.cpp:189-383) over the (NULL, PRS) pair of every frame
constexpr int TII_MAX_LIKELY = 9;        // the reference skips frames with 10 or more likely comb/pattern pairs
constexpr int TII_SLOTS = 32;            // comb/pattern pairs tracked per ensemble
struct TiiSlot { int32_t cp1, num, cycle, pad; unsigned long long acc[TII_NERR]; };   // cp1 = comb * 70 + pattern + 1, 0 = free
struct TiiEvent { int32_t frame, comb, pattern, delay_samples; float error; };       // = dabphy_tii_measurement
struct TiiArgs {
    Tables tab; const cf32* iq; size_t iq_stride; int64_t ring;
    const FrameDesc* desc; int n_ens, n_frames;
    const cf32* rot; const int32_t* rank; const uint8_t* pattern;
    float* abs_err;                      // [B][F][TII_MAX_LIKELY][TII_NERR] this frame's error per candidate delay
    int32_t* likely;                     // [B][F][1 + TII_MAX_LIKELY] count, comb * 70 + pattern ascending
    TiiSlot* state;                      // [B][TII_SLOTS]
    TiiEvent* events; int32_t* n_events; int max_events;   // [B][max_events], [B] (counts every measurement, stored or not)
    int32_t* overflow;                   // [B] measurements dropped because all slots were taken
};
void launch_tii(const TiiArgs& a, hipStream_t s);

// Sample ingest (k_ingest.hip): n samples per ensemble of raw format `format` -> ring positions w, w+1, ... (mod ring)
struct IngestArgs {
    const uint8_t* raw; size_t raw_stride;               // bytes between ensembles
    cf32* iq; size_t iq_stride; uint64_t ring, w, n;
    int format;                                           // dabphy_sample_format
};
void launch_ingest(const IngestArgs& a, int n_ens, hipStream_t s);
void launch_copy_f4(const void* src, void* dst, size_t n16, int blocks, hipStream_t s);
void launch_copy_out(const void* src_device, void* dst_host, size_t bytes, hipStream_t s);      // a kernel's stores into page-locked host memory (hipHostMalloc: the same address on the device)

// Reed-Solomon (k_rs.hip)
struct RsArgs {            // contiguous superframes [n_sf][sf_stride], s = bitrate/8 codewords each
    uint8_t* data; size_t sf_stride; int n_sf, s;
    int* corr; int* uncorr;                       // [n_sf]
};
struct RsMscArgs {         // superframes inside a class's MSC output [n_pairs][n_cif][frame_bytes]
    uint8_t* out; int n_cif, n_pairs, frame_bytes, s, n_sf_per_pair;
    const MscPair* pairs;
    int idx_only;                                 // -1: every pair, else only the pairs at this position of their ensemble's list
    const int* first_cif;                         // [B] logical-frame slot (in this batch) where the ensemble's first superframe starts
    int* result;                                  // [n_pairs][n_sf_per_pair][2] = corrected symbols, uncorrectable flag
};
// DAB+ superframe filter (k_rs.hip: k_superframe): SuperframeFilter::Feed over the logical frames of one batch
struct SfEvent {           // = dabphy_sf_event (include/dabphy.h)
    int32_t cif, corrected, uncorrectable, sync, format, num_aus, au_start[7], au_crc_ok, sf_slot;
};
struct SfArgs {
    const uint8_t* out; int n_cif, n_pairs, frame_bytes, s;    // class output [n_pairs][n_cif][frame_bytes]
    const MscPair* pairs;                         // [n_pairs]
    const int32_t* run; int n_run;                // the pairs this launch walks: run[0 .. n_run) (nullptr: every pair of the class, n_run = n_pairs)
    const FrameDesc* desc; int n_frames;
    uint8_t* state; size_t state_stride;          // [n_pairs] records: int32 frame_count (+12 pad), raw[5 * frame_bytes]
    SfEvent* events; int32_t* n_events;           // [n_pairs][n_cif], [n_pairs]
    uint8_t* sf; int n_slots;                     // [n_pairs][n_slots][5 * frame_bytes] corrected superframes of the synced attempts
    int32_t* stats;                               // optional [B][4]: synchronised superframes, corrected symbols, uncorrectable attempts, AUs failing their CRC
    const uint8_t* gf;                            // alpha_to[256], index_of[256] of GF(256) / 0x11D (init_rs.h:48-60)
    int32_t* accepted;                            // [n_pairs]: set by the wide pass for what it settled (nullptr: serial walk only)
    unsigned long long* wide_stats;               // [2]: pair batches the wide pass settled / was tried on
};
// the classes of one launch (k_rs.hip): cls = DEVICE array of n_cls argument blocks, first = DEVICE array [n_cls + 1] of first blocks
struct SfBatch { const SfArgs* cls; const int32_t* first; int n_cls; };
inline int sf_bucket(int frame_bytes) { const int sf_len = 5 * frame_bytes; return sf_len <= 960 ? 0 : sf_len <= 2880 ? 1 : 2; }
void launch_superframe_bucket(const SfBatch& Bt, int bucket, int total_blocks, int n_cif, bool wide_pass, hipStream_t s);
void launch_rs_superframes(const RsArgs& a, hipStream_t s);
void launch_rs_msc(const RsMscArgs& a, hipStream_t s);

// host-callable launchers (defined next to their kernels)
void launch_demod(const DemodArgs& a, int n_ens, hipStream_t s);
void launch_snr(const SnrArgs& a, hipStream_t s);
void launch_selftest_div127(unsigned long long* out, hipStream_t s);
void launch_selftest_unit_twiddle(unsigned long long* out, hipStream_t s);
void launch_viterbi(const VitArgs& a, hipStream_t s);
void launch_fic_gather(const FicGatherArgs& a, hipStream_t s);
void launch_msc_gather(const MscGatherArgs& a, hipStream_t s);
void launch_lin_gather(const LinGatherArgs& a, hipStream_t s);
void launch_fib_crc(const CrcArgs& a, hipStream_t s);
void launch_fic_ratio(const CrcArgs& a, hipStream_t s);
#ifdef SYNC_CHAIN_TS
void dump_chain_ts();
#endif
void launch_sync_find(const SyncArgs& a, hipStream_t s);
void launch_sync_wide(const SyncArgs& a, hipStream_t s, hipEvent_t front = nullptr);      // front: recorded behind the wide pass proper (searches, sums, judge), in front of the find chain's rounds
void launch_sync_finish(const SyncArgs& a, hipStream_t s);
void launch_acquire(const SyncArgs& a, hipStream_t s);
void launch_slevel_catchup(const SyncArgs& a, hipStream_t s);

} // namespace dabphy
