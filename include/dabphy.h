/* include/dabphy.h -- C ABI of libdabphy_hip.so, the MI355X (gfx950) DAB Mode-I PHY backend for welle.io.
 *
 * The reference (welle.io v2.7) has no FFI: its backend boundary is the C++ facade RadioReceiver
 * (src/backend/radio-receiver.h:52-116) over InputInterface / RadioControllerInterface
 * (src/backend/radio-controller.h:83-215).  This header is the thin C boundary placed UNDER that facade:
 * each entry point replaces one internal seam of the reference's hot path (cited per function) and is
 * what a maintainer binds from the existing C++ host code (see INTEGRATION.md for the 1:1 call sites).
 *
 * Conventions: extern "C"; opaque handle; int status (0 = ok, negative = dabphy_status); no exceptions and
 * no C++/torch types cross the boundary; the caller owns every host buffer, the library owns all device
 * memory; one handle = one HIP device + one stream; a handle is not thread-safe, distinct handles are.
 * Unless a parameter is documented as a DEVICE pointer it is a HOST pointer.
 * The library has no CPU fallback: dabphy_create fails with DABPHY_ERR_NO_DEVICE when no gfx950 GPU is visible.
 */
#ifndef DABPHY_H
#define DABPHY_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dabphy_handle dabphy_handle;

typedef enum {
    DABPHY_OK = 0,
    DABPHY_ERR_NO_DEVICE = -1,      /* no HIP device / not gfx950 */
    DABPHY_ERR_INVALID = -2,        /* bad argument */
    DABPHY_ERR_NOMEM = -3,
    DABPHY_ERR_HIP = -4,            /* a HIP call failed; see dabphy_last_error */
    DABPHY_ERR_STATE = -5           /* call sequence error (e.g. process before bind) */
} dabphy_status;

/* Limits of the batch geometry (dabphy_create returns DABPHY_ERR_INVALID beyond them).  The reference has none -- any number of
 * receivers decode correctly, msc-handler.cpp:129-158 -- and neither has the batch below them: no offset on the decode path is
 * narrower than the range it has to cover.  (Round 3 addressed the soft-bit ring of the fused MSC decode with 32-bit offsets from
 * the START of the ring: n_ensembles * (max_frames + 5) > 18 641 wrapped silently.  The offsets are now relative to the ring slice
 * of the first ensemble a wavefront decodes, a span of a few ensembles that dabphy_process checks against 4 GiB per class -- a
 * class that would exceed it is decoded through the two-kernel path with 64-bit addresses -- see DESIGN.md 4.2.) */
#define DABPHY_MAX_FRAMES 4096u                   /* max_frames: frames per call and ensemble */
#define DABPHY_MAX_ENSEMBLE_FRAMES (1u << 22)     /* n_ensembles * max_frames: code word counts stay inside 31 bits with 64 sub-channels */

/* ---- ABI versioning.  Structures cross this boundary by layout, and the library and its callers are built separately:
 *   dabphy_config            is SIZED: its first member is the sizeof() the caller was compiled with.  Fields are only ever appended;
 *                            dabphy_create reads the fields the caller's header knew and takes the default for the rest (0, except
 *                            fft_placement and freqsync_method: 2 = the reference's own defaults, radio-receiver-options.h:66-84),
 *                            and refuses a structure LARGER than its own (fields it does not know: DABPHY_ERR_INVALID).
 *   every other structure    (dabphy_frame_info, dabphy_sf_event, dabphy_subchannel, dabphy_protection, dabphy_tii_measurement, dabphy_msc_desc) is pinned
 *                            by DABPHY_ABI_VERSION: a change of any of them bumps it.  A caller checks once, e.g. in its constructor:
 *                                dabphy_abi_version() == DABPHY_ABI_VERSION   (dabphy_struct_size(which) tells the library's sizeof for a
 *                                finer diagnosis).
 *   objects built against the headers of rounds 1-3 call the exported symbols `dabphy_create` / `dabphy_get_config` with the UNSIZED
 *   48-byte configuration of those rounds: those entry points stay, frozen to that layout (new fields at their defaults); this header's
 *   dabphy_create / dabphy_get_config are the _v2 symbols (the sized form).  Objects built against ROUND 4's header (an unsized 52-byte
 *   structure that ended in decode_shape) bind the same exported symbols and cannot be told from round 3's: their decode_shape is not
 *   read (the automatic choice applies) and dabphy_get_config does not write it -- rebuild them against this header. */
#define DABPHY_ABI_VERSION 6u
uint32_t dabphy_abi_version(void);
typedef enum { DABPHY_STRUCT_CONFIG = 0, DABPHY_STRUCT_FRAME_INFO = 1, DABPHY_STRUCT_SF_EVENT = 2, DABPHY_STRUCT_SUBCHANNEL = 3,
               DABPHY_STRUCT_PROTECTION = 4, DABPHY_STRUCT_TII_MEASUREMENT = 5, DABPHY_STRUCT_MSC_DESC = 6 } dabphy_struct_id;
size_t dabphy_struct_size(int32_t which);      /* 0 for an unknown id */

/* RadioReceiverOptions (src/backend/radio-receiver-options.h:66-84) + batch geometry */
typedef struct {
    uint32_t struct_size;           /* sizeof(dabphy_config) as the CALLER was compiled: DABPHY_CONFIG_INIT sets it */
    uint32_t n_ensembles;           /* independent ensembles (streams) decoded in lock step; >= 1, n_ensembles * max_frames <= DABPHY_MAX_ENSEMBLE_FRAMES */
    uint32_t max_frames;            /* transmission frames per dabphy_process call and ensemble; 1 .. DABPHY_MAX_FRAMES */
    int32_t device;                 /* HIP device ordinal */
    int32_t fft_placement;          /* FFTPlacementMethod: 2 = ThresholdBeforePeak (default), 1 = EarliestPeakWithBinning, 0 = StrongestPeak */
    int32_t disable_coarse;         /* RadioReceiverOptions::disableCoarseCorrector */
    int32_t want_constellation;     /* keep the 1200 constellation points per frame (onConstellationPoints) */
    int32_t want_impulse_response;  /* keep the 2048-float CIR per frame (onNewImpulseResponse) */
    int32_t demod_chunk;            /* data symbols per work-group of the demod kernel; 0 = default */
    int32_t freqsync_method;        /* FreqsyncMethod of the coarse corrector: 2 = PatternOfZeros (default), 1 = CorrelatePRS, 0 = GetMiddle */
    int32_t pipeline_sync;          /* 1, 2 or 3: synchronise batch k+1 on a second stream while batch k is decoded (throughput mode:
                                       constant n_frames, samples of the next batch already in the ring; the coarse-corrector
                                       feedback then lags one more batch).  1: the next batch's synchroniser is queued behind this
                                       batch's demod kernel; 2: at once (it then competes with the demod kernel: more frames per
                                       second in total, a slower FFT stage); 3: queued like 1 but TWO batches ahead (the samples of
                                       the next two batches must be in the ring; the coarse-corrector feedback lags one batch more) */
    int32_t serial_sync;            /* 0 (default): with two or more frames per call the synchroniser first tries all frames of a batch at
                                       once, each from the state a receiver in lock would be in (window index T_g, correctors unchanged),
                                       accepts those whose assumption held and runs the frame-by-frame chain (2 dependent launches per
                                       frame, OFDMProcessor::run's loop) for the rest: same results, a fraction of the latency while
                                       tracking.  1: always the frame-by-frame chain */
    int32_t no_batch_replay;        /* 0 (default): batch mode (n_frames > 1) follows the reference's state machine EXACTLY also where the FIC
                                       decodes badly: when a coarse-corrector decision of a batch was taken with a stale FIC ratio and can
                                       have mattered (dabphy_get_ratio_lag_effect), dabphy_process puts back everything the batch changed
                                       and decodes it a second time frame by frame, with the FIC of frame n feeding the decision of frame
                                       n + 1 as in OFDMProcessor::run (ofdm-processor.cpp:397).  Costs a copy of the carried state per batch
                                       (a few MB, device to device) and, in such a batch, about three times its normal time.
                                       1: report only (round 1's behaviour) */
    int32_t decode_shape;           /* which Viterbi kernel decodes the FIC and the sub-channels.  0 (default): by batch size -- small batches (at
                                       most 40 960 code words per call: e.g. one ensemble of 18 sub-channels, any batch depth; 32 ensembles x 16
                                       frames) STATE-PARALLEL, the 64 trellis states of a code word in the lanes of a wavefront: one code word
                                       per wavefront up to 1 024 code words (k_viterbi_sp: four times the instructions per code word of the
                                       throughput kernel, a hundredth of its latency), two per wavefront, 32 lanes and two states per lane each,
                                       above that (k_viterbi_sp2: half the vector instructions; its traceback is a pass of its own,
                                       k_traceback_sp2: a lane per code word, a wave per stretch of it); larger batches one LANE per code word
                                       (k_viterbi_fused: the throughput shape).  1: always lane-per-code-word; 2: always k_viterbi_sp2;
                                       3: always k_viterbi_sp.  Same bytes whichever runs. */
    int32_t sync_early;             /* pipelined schedules 1 and 3: where the NEXT batch's synchroniser is queued relative to this batch's decoder.
                                       0 (default): in front of it -- neutral on a batch of receivers in lock, 2 % on one whose ensembles'
                                       PRS windows move (a sampling-clock offset: their window searches run one after the other in the find
                                       chain, latency-bound work that belongs BESIDE the decoder, not behind it).  1: behind it (rounds 1-5).
                                       2: in front only while the last pass met such ensembles.  3 (experiment, a measured loss): in front,
                                       and the decoder's launch waits for the wide pass proper.  Same bytes. */
} dabphy_config;
#define DABPHY_CONFIG_INIT { (uint32_t)sizeof(dabphy_config) }      /* dabphy_config cfg = DABPHY_CONFIG_INIT;  -- sized, every option at its default */

/* Depuncturing description of one convolutional codeword class: up to four (L_i blocks of 128 bits, PI_i)
 * tuples followed by the 24-bit tail (PI_X).  Replaces the constructor tables of EEPProtection
 * (eep-protection.cpp:32-113), UEPProtection (uep-protection.cpp:27-167) and the fixed FIC scheme
 * (fic-handler.cpp:158-191). */
typedef struct {
    int32_t nbits;                  /* decoded bits: 768 (FIC) or 24 * bitrate */
    int32_t L[4];
    int32_t PI[4];                  /* 1..24, 0 = tuple unused */
} dabphy_protection;

/* One MSC sub-channel to decode: Subchannel (src/backend/dab-constants.h:164-198) after bitrate()/protection lookup */
typedef struct {
    int32_t subch_id;
    int32_t start_cu;               /* Subchannel::startAddr */
    int32_t size_cu;                /* Subchannel::length */
    dabphy_protection prot;
} dabphy_subchannel;

int dabphy_create_v2(const dabphy_config* cfg, dabphy_handle** out);   /* RadioReceiver::RadioReceiver, radio-receiver.cpp:66-80 */
#ifndef DABPHY_BUILDING_LIBRARY
static inline int dabphy_create(const dabphy_config* cfg, dabphy_handle** out) { return dabphy_create_v2(cfg, out); }
#endif
void dabphy_destroy(dabphy_handle* h);
const char* dabphy_last_error(const dabphy_handle* h);
const char* dabphy_device_name(const dabphy_handle* h);
/* the configuration in effect (defaults resolved: e.g. demod_chunk = 25 when n_ensembles * max_frames >= 1024, else 15; the
 * synchroniser options as last set by dabphy_set_options) */
int dabphy_get_config_v2(const dabphy_handle* h, dabphy_config* out);  /* out->struct_size = the caller's sizeof(dabphy_config) on entry (at most that much is written) */
#ifndef DABPHY_BUILDING_LIBRARY
static inline int dabphy_get_config(const dabphy_handle* h, dabphy_config* out) { return dabphy_get_config_v2(h, out); }
#endif
/* RadioReceiver::setReceiverOptions -> OFDMProcessor::setReceiverOptions (ofdm-processor.cpp:518-529) at run time: the FFT placement
 * and frequency-sync methods apply from the next frame on (the reference calls phaseRef.selectFFTWindowPlacement and reads
 * freqsyncMethod per frame, :399); a CHANGE of disable_coarse restarts the receiver exactly like the reference does (:523-528
 * resetCoarseCorrector + restart: dabphy_reset semantics, the stream position is kept).  In pipelined mode frames that were already
 * synchronised ahead keep the old options.  *restarted (may be NULL) tells whether the restart happened. */
int dabphy_set_options(dabphy_handle* h, int32_t fft_placement, int32_t freqsync_method, int32_t disable_coarse, int32_t* restarted);

/* protection helpers (pure host) */
int dabphy_protection_fic(dabphy_protection* p);
int dabphy_protection_eep(dabphy_protection* p, int bitrate, int profile_b, int level);     /* eep-protection.cpp:32-113 */
int dabphy_protection_uep(dabphy_protection* p, int bitrate, int level);                    /* uep-protection.cpp:120-167 */
int dabphy_protection_input_bits(const dabphy_protection* p);                               /* punctured soft bits consumed */
/* row `table_index` (0 .. 63, the 6-bit index a short-form FIG 0/1 carries) of the UEP table: sub-channel size in CUs, protection level
 * 1 .. 5, bit rate (fib-processor.cpp:496-503 reads the same triple from ProtLevel, dab-constants.cpp:75-142) */
int dabphy_uep_table_entry(int table_index, int* size_cu, int* level, int* bitrate);

/* ---- seam 1: OfdmDecoder::pushAllSymbols (ofdm-decoder.cpp:132-139) -> processPRS + 75 x decodeDataSymbol --------
 * frames: n_frames x (2048 + 75*2552) complex floats, [PRS useful part][75 symbols incl. cyclic prefix], already
 *         frequency-corrected (as OFDMProcessor hands them over).
 * soft:   n_frames x 75 x 3072 int8 (layout of ofdm-decoder.cpp:211-212: [0,1536) real bits, [1536,3072) imaginary)
 * constellation (may be NULL): n_frames x 1200 complex floats (every 96th carrier, :214-216)
 * snr (may be NULL): n_frames floats; the value OfdmDecoder would pass to onSNR after that frame, NaN when it
 *         would not report (it reports every 11th frame, :155-158).  The IIR state persists in the handle. */
int dabphy_demod_frames(dabphy_handle* h, const float* frames, uint32_t n_frames,
                        int8_t* soft, float* constellation, float* snr);

/* ---- seam 2: Viterbi::deconvolve (viterbi.cpp:227-245), batched -------------------------------------------
 * in:  n_codewords x 4*(nbits+6) soft values (depunctured, erasures = 0)
 * out: n_codewords x nbits/8 bytes, bit i of a codeword at byte i/8, MSB first (the packing of
 *      decoder_adapter.cpp:61-67; the reference returns one bit per byte) */
int dabphy_viterbi_batch(dabphy_handle* h, const int8_t* in, uint32_t nbits, uint32_t n_codewords, uint8_t* out);

/* ---- FicHandler::processFicBlock x3 (fic-handler.cpp:111-230), batched over frames -------------------------
 * soft: n_frames x 9216 soft bits (symbols 1..3).  fib: n_frames x 12 x 32 bytes (energy dispersal removed).
 * crc_ok: n_frames x 12 flags.  Returns the FIC success ratio in percent (getFicDecodeRatioPercent, :234-237)
 * through *ratio_percent when not NULL; the counter persists in the handle (ensemble 0). */
int dabphy_fic_decode(dabphy_handle* h, const int8_t* soft, uint32_t n_frames, uint8_t* fib, uint8_t* crc_ok,
                      int32_t* ratio_percent);

/* ---- Protection::deconvolve (protection.h:32-37) + EnergyDispersal + bit packing, batched ------------------
 * in:  n_codewords x dabphy_protection_input_bits(prot) punctured soft bits (time de-interleaved)
 * out: n_codewords x nbits/8 bytes (as written to the reference's .msc dump, decoder_adapter.cpp:71-73) */
int dabphy_msc_deconvolve(dabphy_handle* h, const dabphy_protection* prot, const int8_t* in, uint32_t n_codewords,
                          uint8_t* out);

/* ======================================================================================================
 * Streaming receiver: OFDMProcessor::run (ofdm-processor.cpp:235-501) for n_ensembles independent sample
 * streams at once, followed by the whole decode chain.  This is what RadioReceiver::restart() starts in the
 * reference; the per-frame callbacks of RadioControllerInterface become the getters below.
 * ====================================================================================================== */

/* InputInterface::getSamples (radio-controller.h:193-199) replaced by a sample ring in HBM.
 * dabphy_stream_bind_device: d_iq is a DEVICE pointer to n_ensembles rows of `stride_samples` complex floats, of
 *   which the first `ring_samples` are the ring; `n_valid` samples have been written (ignored when loop != 0:
 *   the ring then repeats for ever like CRAWFile with rewind, raw_file.cpp:284-286).  Resets the receiver state.
 * dabphy_stream_upload: convenience for host data: n_ensembles x n_samples complex floats are copied to HBM. */
int dabphy_stream_bind_device(dabphy_handle* h, const void* d_iq, uint64_t ring_samples, uint64_t stride_samples,
                              uint64_t n_valid, int32_t loop);
int dabphy_stream_upload(dabphy_handle* h, const float* iq, uint64_t n_samples, int32_t loop);

/* Live input: a library-owned ring of ring_samples per ensemble (>= 4 frames) that the host appends to, the way
 * CVirtualInput implementations fill their ring buffers (input/raw_file.cpp:244-290).  dabphy_stream_write copies
 * n_ensembles x n_samples complex floats ([ensemble][sample]) behind the samples written so far; the caller must not
 * overwrite unread samples: at most ring_samples - (written - dabphy_stream_consumed()) may be appended.
 * dabphy_stream_consumed = samples the synchroniser no longer needs (minimum over ensembles). */
int dabphy_stream_open(dabphy_handle* h, uint64_t ring_samples);
int dabphy_stream_write(dabphy_handle* h, const float* iq, uint64_t n_samples);
uint64_t dabphy_stream_consumed(dabphy_handle* h);

/* The same, from the sample formats the reference's file / SDR front ends deliver (CRAWFileFormat, input/raw_file.h;
 * conversion = CRAWFile::convertSamples, input/raw_file.cpp:324-366, done on the device: u8 moves 2 bytes per sample
 * over PCIe instead of 8).  data = [ensemble][n_samples] interleaved I/Q in `format`.  DABPHY_FMT_S16LE / _S16BE keep
 * the reference's byte order for those names ((byte0 << 8) | byte1 and (byte1 << 8) | byte0) and its missing scaling. */
typedef enum { DABPHY_FMT_CF32 = 0, DABPHY_FMT_U8 = 1, DABPHY_FMT_S8 = 2, DABPHY_FMT_S16LE = 3, DABPHY_FMT_S16BE = 4 } dabphy_sample_format;
int dabphy_stream_write_raw(dabphy_handle* h, const void* data, uint64_t n_samples, int32_t format);

/* The same without waiting: the copy and the conversion run on the library's copy stream while earlier batches are decoded.
 * The samples become visible to dabphy_process only after dabphy_stream_commit (which costs nothing; dabphy_process then
 * orders its kernels behind the committed copies).  Double buffering:
 *     commit();  write_raw_async(batch k+1);  process(batch k);     -- k+1 crosses PCIe while k is decoded
 * `data` (page-locked memory, see below) must stay untouched until the next-but-one call of this function returns, i.e.
 * alternate between two host buffers. */
int dabphy_stream_write_raw_async(dabphy_handle* h, const void* data, uint64_t n_samples, int32_t format);
int dabphy_stream_commit(dabphy_handle* h);

/* diagnostics: n samples of ensemble `ensemble` starting at absolute sample index `pos`, as they lie in the ring (cf32), e.g. to
 * compare the device's format conversion with CRAWFile::convertSamples */
int dabphy_stream_read(dabphy_handle* h, uint32_t ensemble, uint64_t pos, uint64_t n_samples, float* out);

/* Page-locked host memory for the buffers handed to dabphy_stream_write / _write_raw (DMA at PCIe rate instead of a
 * staged copy); plain malloc'ed buffers work too. */
int dabphy_host_alloc(size_t bytes, void** out);
void dabphy_host_free(void* p);

/* OFDMProcessor::restart (ofdm-processor.cpp:115-132): correctors, phase, sync state, FIC counter, SNR filter cleared */
int dabphy_reset(dabphy_handle* h);

/* Sub-channel selection.  Every receiver of the reference selects its own services (radio-receiver.cpp:120-137 -> MscHandler::
 * addSubchannel / removeSubchannel / stopProcessing, msc-handler.cpp:61-127), so every ensemble of a batch has its OWN list:
 *   dabphy_set_subchannels_ensemble   the list of ONE ensemble (at most 64 entries; n = 0: none).  Takes effect with the next
 *                                     dabphy_process; the results of the last batch stay readable until then.
 *   dabphy_set_subchannels            the same list for every ensemble of the handle (applied at once).
 * A list REPLACES the ensemble's previous one.  A sub-channel that is in both (same subch_id, start_cu, size_cu and protection) is a
 * service that keeps playing: its time de-interleaver and its DAB+ superframe window (dabphy_superframes*) are not disturbed, whatever
 * else is added or removed, in its own or in any other ensemble -- as in the reference, where addSubchannel touches no other stream.
 * A sub-channel that is new starts with an empty time de-interleaver: like DabAudio (dab-audio.cpp:146-149) it emits its first
 * logical frame on the 17th CIF after it was selected (first_valid of dabphy_get_msc).
 * subch_index in the getters below = position in the ensemble's list. */
int dabphy_set_subchannels(dabphy_handle* h, const dabphy_subchannel* list, uint32_t n);
int dabphy_set_subchannels_ensemble(dabphy_handle* h, uint32_t ensemble, const dabphy_subchannel* list, uint32_t n);
/* entries of the ensemble's list as last set */
int dabphy_get_subchannel_count(dabphy_handle* h, uint32_t ensemble, uint32_t* n);

/* Decode the next n_frames (<= max_frames) transmission frames of every ensemble: acquisition (null-symbol
 * search) where an ensemble is not synchronised, PRS time sync, coarse/fine frequency tracking, demodulation,
 * FIC and MSC channel decoding.  Results stay in HBM until fetched with the getters.
 * Note on the coarse corrector: the reference consults the FIB CRC success ratio of the PREVIOUS frame
 * (ofdm-processor.cpp:397); that feedback is exact when n_frames == 1.  With n_frames > 1 the ratio of the
 * last finished batch is used for the whole batch -- identical once the ratio is >= 50 (normal tracking) or
 * with disable_coarse; where that was not so AND can have mattered the batch is decoded a second time frame by
 * frame (dabphy_config.no_batch_replay), so the results are the reference's in every case;
 * dabphy_get_ratio_lag / dabphy_get_ratio_lag_effect report such decisions when the replay is turned off.
 * The FIBs, CRC flags, frame information and (with dabphy_set_auto_superframes) the superframe totals of the batch
 * are back in page-locked host memory when the call returns; everything else is copied on request. */
int dabphy_process(dabphy_handle* h, uint32_t n_frames);
/* ... and the library tells exactly when that matters: stale_frames[b] = frames of ensemble b (since dabphy_reset) for which the
 * synchroniser, running ahead of the decoder, consulted the coarse corrector although the reference -- which knows the ratio after the
 * previous frame -- would not have, or the other way round; first_stale_frame[b] = frame number of the first one (-1: none).  Up to
 * that frame the ensemble's output is the reference's bit for bit; from it on it is unless the corrector's step differed.  Always 0
 * with one frame per call, with disable_coarse, and while the ratio stays on one side of 50 %.  [n_ensembles] each, may be NULL. */
int dabphy_get_ratio_lag(dabphy_handle* h, int32_t* stale_frames, int64_t* first_stale_frame);
/* ... and those of them that can have changed anything, [n_ensembles] each: the coarse corrector was consulted although the reference
 * would not have AND it moved coarseCorrector (a consultation that returns no correction has no other effect,
 * ofdm-processor.cpp:397-409), or it was not consulted although the reference would have.  While this count is zero the ensemble's
 * output is the reference's frame for frame; first_effective_frame is where it may part from it (-1: nowhere) */
int dabphy_get_ratio_lag_effect(dabphy_handle* h, int32_t* effective_frames, int64_t* first_effective_frame);

typedef struct {
    int64_t sample_pos;             /* absolute index of the sync buffer start (ofdm-processor.cpp:337) */
    int64_t frame_no;               /* running frame counter of the ensemble */
    int32_t start_index;            /* PhaseReference::findIndex result (< 0: sync lost, onSyncChange(false)) */
    int32_t valid;                  /* 1: frame was demodulated; 0: no frame this slot (acquiring / not enough samples);
                                       3: the PRS window search failed on this slot (SyncOnPhase failed, onSyncChange(false)) */
    int32_t fine_corrector;         /* as passed to onFrequencyCorrectorChange after the frame */
    int32_t coarse_corrector;
    float snr;                      /* onSNR value, NaN when the reference would not report on this frame */
} dabphy_frame_info;

int dabphy_get_frame_info(dabphy_handle* h, dabphy_frame_info* out /* [n_ensembles][n_frames] */);
/* onFIBDecodeSuccess: fib [n_ensembles][n_frames][12][32], crc_ok [n_ensembles][n_frames][12] */
int dabphy_get_fibs(dabphy_handle* h, uint8_t* fib, uint8_t* crc_ok);
/* the same buffers where they lie in HBM (DEVICE pointers, valid until the next dabphy_process / dabphy_destroy; complete when
 * dabphy_process has returned): for consumers on the device, e.g. the multi-GPU gather of the FIC over RCCL */
int dabphy_get_fibs_device(dabphy_handle* h, const uint8_t** d_fib, const uint8_t** d_crc_ok);
/* ... and the page-locked HOST copies dabphy_process brings back with every batch (same layout, valid until the next
 * dabphy_process / dabphy_destroy): what dabphy_get_fibs copies from; a consumer that only reads can skip that copy */
int dabphy_get_fibs_host(dabphy_handle* h, const uint8_t** fib, const uint8_t** crc_ok);
int dabphy_get_fic_ratio(dabphy_handle* h, int32_t* ratio_percent /* [n_ensembles] */);
/* synchroniser counters since dabphy_reset (any pointer may be NULL), [n_ensembles] each: failed window searches
 * (PhaseReference::findIndex < 0, ofdm-processor.cpp:347); frames whose fine corrector had to be settled by the ordered float sums
 * of ofdm-processor.cpp:435-442 because the interval test on the exact sums was undecided; re-acquisitions whose sLevel
 * (ofdm-processor.cpp:216) could not be certified because the samples pulled since the last acquisition were no longer all
 * available (more than 64 frames ago or out of the ring) and the two bracketing replays had not met (DESIGN.md sections 4.3, 7) */
int dabphy_get_sync_stats(dabphy_handle* h, int32_t* lost, int32_t* exact_sums, int32_t* relock_inexact);
/* the wide synchroniser pass (dabphy_config.serial_sync) since dabphy_reset / dabphy_create (any pointer may be NULL): frames per
 * ensemble that were accepted from it [n_ensembles]; passes queued; passes after which the frame-by-frame chain had to take over for
 * at least one ensemble */
int dabphy_get_wide_sync_stats(dabphy_handle* h, int32_t* wide_frames, uint64_t* passes, uint64_t* fallbacks);
/* ... of the frames accepted from the wide pass, those whose window search ran in the FIND CHAIN [n_ensembles]: an ensemble whose window
 * index moves (a sampling-clock offset: ofdm-processor.cpp:337-350) cannot have its window positions predicted; its searches then run
 * one after the other on the device, each from the position the previous one really found, and only the cyclic-prefix sums of all
 * those frames run at once */
int dabphy_get_find_chain_stats(dabphy_handle* h, int32_t* chain_frames);
/* the wide pass of the DAB+ superframe filter since dabphy_create (either pointer may be NULL): (ensemble, sub-channel) batches whose
 * superframe attempts were all made at once and accepted -- the rest were walked frame by frame as SuperframeFilter::Feed does
 * (dabplus_decoder.cpp:50-158); results are identical either way -- and batches it was tried on (those with at least one full
 * five-frame window).  Counts passes, not batches: the second pass of a batch that exact batch mode decodes twice counts again. */
int dabphy_get_wide_superframe_stats(dabphy_handle* h, uint64_t* settled, uint64_t* tried);
/* batches decoded a second time (see dabphy_config.no_batch_replay) since dabphy_reset / dabphy_create */
int dabphy_get_replayed_batches(dabphy_handle* h, uint64_t* batches);
/* OFDM symbols (since dabphy_create) whose samples dabphy_process mixed with oscillator values converted without / with the
 * per-sample rounding test (csrc/osc_exact.h: the synchroniser marks the symbols that read one of the 36 table entries next to a
 * float rounding boundary; only those take the test and, where it cannot decide, the table) */
int dabphy_get_osc_stats(dabphy_handle* h, uint64_t* unchecked_symbols, uint64_t* checked_symbols);
/* Scan mode (RadioReceiver::restart(doScan = true) -> OFDMProcessor::set_scanMode, ofdm-processor.cpp:256-262,351-355), [n_ensembles]
 * each, since dabphy_reset: attempts = entries into the notSynced state (after the sLevel priming, after every hopeless null search,
 * after every failed window search) -- the reference reports onSignalPresence(false) when this exceeds 5 before a lock;
 * attempts_at_first_lock = that count when the first window search succeeded (-1: none yet) -- onSignalPresence(true) if <= 5. */
int dabphy_get_scan_stats(dabphy_handle* h, int32_t* attempts, int32_t* attempts_at_first_lock);
/* OFDMProcessor::sLevel (ofdm-processor.cpp:216) is only read when lock has been lost; by default the library advances it then, over
 * the samples pulled since the last acquisition (up to 64 frames back, as far as they are still in the ring).  on = 1: advance it
 * after every frame instead (about 3 ms of one GPU lane per frame and ensemble): always exact, meant for the real-time
 * single-ensemble receiver whose ring holds only a few frames. */
int dabphy_set_track_slevel(dabphy_handle* h, int32_t on);
/* decoded logical frames of sub-channel `subch_index` (position in the list) of EVERY ensemble -- all of them must have a sub-channel
 * of the same bit rate there (DABPHY_ERR_INVALID otherwise: read such batches per ensemble, dabphy_get_msc_ensemble):
 * out [n_ensembles][4*n_frames][nbits/8] = the bytes DecoderAdapter::addtoFrame writes to its dump file (out_capacity = size of
 * `out` in bytes; DABPHY_ERR_INVALID when it is too small).  Row layout: the logical frames of an ensemble are packed in CIF order
 * from row 0 on, whatever slots of the batch its demodulated frames occupied (a slot that failed its window search leaves no gap):
 * rows [first_valid[b], n_rows[b]) are this batch's frames, n_rows[b] = 4 x (frames of ensemble b with valid == 1), rows beyond it
 * are undefined.  first_valid[b] = number of leading rows that carry no frame yet (the de-interleaver emits its first frame on the
 * 17th CIF after the sub-channel was selected, dab-audio.cpp:146-149).  first_valid / n_rows may be NULL. */
int dabphy_get_msc(dabphy_handle* h, uint32_t subch_index, uint8_t* out, size_t out_capacity, int32_t* first_valid, int32_t* n_rows);
/* ... of ONE ensemble: out [4*n_frames][nbits/8], first_valid / n_rows single values */
int dabphy_get_msc_ensemble(dabphy_handle* h, uint32_t ensemble, uint32_t subch_index, uint8_t* out, size_t out_capacity, int32_t* first_valid, int32_t* n_rows);
/* ... of EVERY selected sub-channel of EVERY ensemble in one pass: the bulk drain (MscHandler hands every selected sub-channel its
 * logical frames as the CIF completes, msc-handler.cpp:129-158 -> dab-audio.cpp:151-160; a host that drains thousands of services per
 * batch cannot afford one copy and one stream synchronisation per service).  A protection class's output is ONE contiguous array in
 * HBM ([pair][4*n_frames][nbits/8], pairs = the (ensemble, sub-channel) pairs of the batch that share the profile), so the drain is one
 * device-to-host copy per class into `buf` -- at PCIe rate when `buf` is page-locked (dabphy_host_alloc) -- and an index table:
 *   desc [n]  one record per (ensemble, list position), ordered by ensemble, then by position: where that service's rows lie in `buf`
 *             (row r of the service = buf + offset + r * row_bytes, 4*n_frames rows reserved), and first_valid / n_rows as dabphy_get_msc
 *             defines them.
 *   dabphy_msc_batch_size        bytes `buf` must hold and records `desc` must hold for the last batch (either pointer may be NULL)
 *   dabphy_get_msc_batch         queue the copies, wait for them, return (n_desc = records written)
 *   dabphy_msc_drain_begin       queue the copies on a stream of their own and return at once: the NEXT dabphy_process may be called while
 *                                they are in flight (the class outputs are first copied to a staging area in HBM -- tens of microseconds on
 *                                the device -- so that no decoder ever waits for the host link); `desc` is complete on return, `buf` when
 *   dabphy_msc_drain_wait        returns (also called by dabphy_destroy / dabphy_reset and before the sub-channel lists are re-applied).
 * DABPHY_ERR_INVALID when a capacity is too small (nothing is queued then). */
typedef struct {
    uint32_t ensemble, subch_index;      /* position in that ensemble's list (dabphy_set_subchannels[_ensemble]) */
    uint32_t row_bytes;                  /* one logical frame: nbits / 8 = 3 * bit rate in kbit/s */
    int32_t first_valid, n_rows;         /* rows [first_valid, n_rows) are this batch's logical frames (dabphy_get_msc) */
    uint32_t subch_id;                   /* SubChId as given in the list */
    uint64_t offset;                     /* byte offset of row 0 in buf */
} dabphy_msc_desc;
int dabphy_msc_batch_size(dabphy_handle* h, size_t* buf_bytes, uint32_t* n_desc);
int dabphy_get_msc_batch(dabphy_handle* h, dabphy_msc_desc* desc, uint32_t desc_capacity, uint32_t* n_desc, uint8_t* buf, size_t buf_capacity);
int dabphy_msc_drain_begin(dabphy_handle* h, dabphy_msc_desc* desc, uint32_t desc_capacity, uint32_t* n_desc, uint8_t* buf, size_t buf_capacity);
int dabphy_msc_drain_wait(dabphy_handle* h);
int dabphy_get_impulse_response(dabphy_handle* h, float* out /* [n_ensembles][n_frames][2048] */);      /* onNewImpulseResponse */
/* onNewNullSymbol (ofdm-processor.cpp:462-469): the 2656 oscillator-corrected samples of the null symbol that follows each
 * demodulated frame of the last batch, out[n_ensembles][n_frames][2656][2] (zeros where valid != 1).  Computed on request. */
int dabphy_get_null_symbols(dabphy_handle* h, float* out);
int dabphy_get_constellation(dabphy_handle* h, float* out /* [n_ensembles][n_frames][1200] cf32 */);   /* onConstellationPoints */
int dabphy_get_soft_bits(dabphy_handle* h, uint32_t ensemble, uint32_t frame, int8_t* out /* 75*3072 */);

/* ---- RSDecoder::DecodeSuperframe (dabplus_decoder.cpp:326-359), batched --------------------------------------
 * sf: n_sf superframes of 120*s_per_sf bytes (s_per_sf = bitrate/8), corrected in place.  corrected[i] = number of
 * corrected symbols in superframe i (total_corr_count), uncorrectable[i] != 0 when any of its codewords failed. */
int dabphy_rs_superframes(dabphy_handle* h, uint8_t* sf, uint32_t s_per_sf, uint32_t n_sf, int32_t* corrected,
                          int32_t* uncorrectable);
/* The same on the MSC output of the last dabphy_process, in HBM.  subch_index < 0: every sub-channel; >= 0: the sub-channel at that
 * position of every ensemble that has one.
 * first_cif[b]: CIF slot of this batch (0..4*n_frames-1, may be negative) where ensemble b's first superframe
 * starts -- the alignment SuperframeFilter::CheckSync (dabplus_decoder.cpp:171-215) finds on the host.  Only
 * superframes lying entirely inside the batch are decoded.  corrected / uncorrectable (may be NULL): per ensemble sums. */
int dabphy_rs_decode_msc(dabphy_handle* h, int32_t subch_index, const int32_t* first_cif, int32_t* corrected,
                         int32_t* uncorrectable);

/* ---- per-stage device time of the last dabphy_process (HIP events on the stream each stage is queued on; dabphy_get_stage_times) ----
 * ms[0] sync chain   the synchroniser chain that produced this batch (on its own stream: in pipelined modes it ran beside the previous batch's decode)
 * ms[1] demod        the demod kernel
 * ms[2] SNR          the SNR kernels (auxiliary stream; they may wait for wave slots beside the decoder: wall time, not work)
 * ms[3] FIC          since round 4 the FIC's code words ride in the fused decode launch: this is then ONLY the FIB CRC + FIC-ratio kernels
 *                    behind it; with the FIC on its own kernels (experiments build, spans beyond 4 GiB) gather + Viterbi + CRC as before
 * ms[4] MSC gather   classes that take the two-kernel path (0 when every class rides in the fused launch)
 * ms[5] MSC Viterbi  the fused decode launch: EVERY protection class of every ensemble AND, when fused, the FIC
 * ms[6] Reed-Solomon the superframe filter pass over all classes (dabphy_set_auto_superframes / dabphy_superframes_stats), or dabphy_rs_decode_msc */
/* DAB+ superframe filter on the device (SuperframeFilter::Feed / CheckSync, dabplus_decoder.cpp:50-213) for sub-channel
 * subch_index of every ensemble, over the logical frames of the last dabphy_process() batch: 5-frame sliding window,
 * Reed-Solomon, Fire-code synchronisation, access-unit table and AU CRCs.  The window state is carried from batch to
 * batch, so call it exactly once per dabphy_process() for every sub-channel it is used on.
 *   events   [n_ensembles][4 * n_frames]: one record per decode attempt (= per FECInfo() call of the reference), in order
 *   n_events [n_ensembles]
 *   sf       [n_ensembles][n_slots][120 * bitrate/8], n_slots = 4 * n_frames / 5 + 1: the corrected superframes of the
 *            synchronised attempts (event.sf_slot); may be NULL
 * The access units of a synchronised superframe are sf[au_start[i] .. au_start[i+1]) with the last two bytes = CRC. */
typedef struct {
    int32_t cif;                         /* logical frame (0 .. 4*n_frames-1 of the batch) that completed the window */
    int32_t corrected, uncorrectable;    /* RSDecoder::DecodeSuperframe totals (what onRsErrors / FECInfo report) */
    int32_t sync;                        /* CheckSync() */
    int32_t format;                      /* sf[2]: dac_rate 0x40, sbr 0x20, aac_channel_mode 0x10, ps 0x08, mpeg_surround 0x07 */
    int32_t num_aus, au_start[7];
    int32_t au_crc_ok;                   /* bit i = access unit i passed its CRC-16 */
    int32_t sf_slot;                     /* index into sf, -1 when not synchronised */
} dabphy_sf_event;
int dabphy_superframes(dabphy_handle* h, uint32_t subch_index, dabphy_sf_event* events, int32_t* n_events, uint8_t* sf);
/* ... for sub-channel subch_index of ONE ensemble (its own list position): events [4 * n_frames], n_events [1], sf [n_slots][120 * bitrate/8].
 * dabphy_superframes needs the same bit rate at that position in every ensemble. */
int dabphy_superframes_ensemble(dabphy_handle* h, uint32_t ensemble, uint32_t subch_index, dabphy_sf_event* events, int32_t* n_events, uint8_t* sf);
/* The same filter over EVERY DAB+ sub-channel of every ensemble in one launch per protection class (instead of, not in
 * addition to, dabphy_superframes for this batch); nothing but totals leaves the device:
 * stats [n_ensembles][4] = synchronised superframes, corrected symbols, uncorrectable attempts, access units failing their CRC */
int dabphy_superframes_stats(dabphy_handle* h, int32_t* stats);
/* on = 1: every following dabphy_process queues that all-sub-channel filter pass itself, behind the MSC Viterbi kernels of the same
 * submission (no host round trip between decode and filter); dabphy_superframes_stats then only fetches the totals of the batch.
 * on = 2: the pass of a batch is DEFERRED to the next dabphy_process, which queues it beside its own FFT stage on a stream of its own
 * (nothing of the next batch needs the filter's results before its decoders overwrite the class outputs, and those wait for the pass
 * on the device): the filter leaves the step's tail.  dabphy_superframes_stats then returns the totals of the batch BEFORE the last
 * dabphy_process (zeros after the first one); one more call without a dabphy_process in between runs the last batch's pass at once and
 * returns its totals (the end of a stream).  Every batch is filtered exactly once either way, with the same results.  A change of
 * the sub-channel lists, dabphy_reset and switching the mode run or drop what is pending first.
 * on = 0: dabphy_superframes_stats runs the pass when it is called. */
int dabphy_set_auto_superframes(dabphy_handle* h, int32_t on);

/* ---- TIIDecoder (tii-decoder.cpp:189-383), fed by OFDMProcessor::run with the PRS and the trailing NULL symbol of every
 * frame (ofdm-processor.cpp:381-386,462-466) when RadioReceiverOptions::decodeTII is set (radio-receiver-options.h:75; welle-cli
 * sets it by default, welle-cli.cpp:409).  dabphy_set_tii switches the side path on or off for the following dabphy_process calls
 * (setReceiverOptions semantics).  Every demodulated frame is analysed -- the reference decoder drops pairs while its thread is
 * busy -- and its arithmetic is kept literally: uint64 error sums updated through float, 5 measurements per report, the winner
 * picked in the iteration order of the C++ library's unordered_map<float, uint64_t> (taken from the host's own container).
 * dabphy_get_tii returns what onTIIMeasurement (radio-controller.h:128) would have been called with during the last batch:
 * out [n_ensembles][max_per_ensemble], n[b] = measurements of ensemble b (may exceed what was stored), ordered by frame, then by
 * (comb, pattern) -- the reference's order inside one frame is unspecified.  Up to 32 comb/pattern pairs are tracked per
 * ensemble (the reference's map is unbounded). */
typedef struct {
    int32_t frame;                       /* frame of the batch whose NULL symbol completed the 5th measurement */
    int32_t comb, pattern;               /* tii_measurement_t (radio-controller.h:56-63) */
    int32_t delay_samples;
    float error;
} dabphy_tii_measurement;
int dabphy_set_tii(dabphy_handle* h, int32_t on);
int dabphy_get_tii(dabphy_handle* h, dabphy_tii_measurement* out, int32_t* n, uint32_t max_per_ensemble);

int dabphy_set_profiling(dabphy_handle* h, int32_t on);
int dabphy_get_stage_times(dabphy_handle* h, float* ms /* [7] */);

/* (Timing drivers and device self-tests used by tests/ and tools/ are declared in include/dabphy_test.h: not part of the receiver API.) */

#ifdef __cplusplus
}
#endif
#endif
