/* oracle/dabphy_oracle.h -- TEST INFRASTRUCTURE (CPU restatement of the reference hot path).
 *
 * This library is the parity checker for the HIP product path.  Only tests/, __graft_entry__.smoke()
 * and the cpu_baseline leg of bench.py may load it; the product (welle.io_amd/csrc, libdabphy_hip.so)
 * never links, loads or calls it.
 *
 * Pinning: the reference ships no golden vectors for this path.  The restatement
 * is therefore pinned against the reference ITSELF, compiled unmodified into oracle/_ref/ (see
 * oracle/Makefile, oracle/ref_harness.cpp) -- tests/test_oracle_vs_ref.py compares every function below
 * bit-for-bit with the real classes, and tests/golden/ holds vectors generated from oracle/_ref by
 * tests/golden/make_golden.py for machines where /root/reference is absent.
 */
#ifndef DABPHY_ORACLE_H
#define DABPHY_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } orc_cf32;

#define ORC_TU 2048
#define ORC_TS 2552
#define ORC_TG 504
#define ORC_TNULL 2656
#define ORC_TF 196608
#define ORC_L 76
#define ORC_K 1536
#define ORC_INPUT_RATE 2048000

/* tables (built once, thread-safe after orc_init) */
void orc_init(void);
const orc_cf32* orc_twiddles_fwd(void);          /* kiss_fft.c:353-364, 2048 entries */
const orc_cf32* orc_prs_reftable(void);          /* phasereference.cpp:45-51 */
const int16_t* orc_freq_perm(void);              /* freq-interleaver.cpp:35-59, 1536 entries (-768..768) */
const int8_t* orc_pcodes(int idx);               /* protTables.cpp:25-56, idx 0..23 -> 32 entries */
const orc_cf32* orc_nco_table(void);             /* ofdm-processor.cpp:92-94, 2 048 000 entries */
const uint8_t* orc_prbs(int n);                  /* fic-handler.cpp:62-71 / energy_dispersal.h:39-49, n <= 9216 */

/* kiss_fft.c:21-90,232-300 for nfft=2048 (factors 4,4,4,4,4,2); inverse adds the 1/N scaling of fft.cpp:152-164 */
void orc_fft2048(const orc_cf32* in, orc_cf32* out, int inverse);

/* phasereference.cpp:73-256, method ThresholdBeforePeak (2) and StrongestPeak (0) */
int orc_find_index(const orc_cf32* v, int method, float* impulse2048);

/* ofdm-processor.cpp:537-616, PatternOfZeros; returns carrier offset (100 = none) */
int orc_coarse_prs(const orc_cf32* prs2048);
int orc_coarse_prs_method(const orc_cf32* prs2048, int freqsync_method /* 0 GetMiddle, 1 CorrelatePRS, 2 PatternOfZeros */);

/* ofdm-decoder.cpp:144-230.  state: phase reference (2048 cf32), snr (float), snrCount. */
typedef struct {
    orc_cf32 phase_ref[ORC_TU];
    float snr;
    int snr_count;
} orc_demod_state;
void orc_demod_reset(orc_demod_state* st);
/* returns 1 when the reference would call onSNR (every 11th frame) and writes *snr_out */
int orc_demod_prs(orc_demod_state* st, const orc_cf32* prs2048, float* snr_out);
void orc_demod_symbol(orc_demod_state* st, const orc_cf32* sym2552, int8_t* soft3072, orc_cf32* constellation16);

/* viterbi.cpp:227-339.  in: 4*(nbits+6) soft values, out: nbits bytes of 0/1 */
void orc_viterbi(const int8_t* in, int nbits, uint8_t* out);

/* depuncturing (fic-handler.cpp:158-191, eep-protection.cpp:115-148, uep-protection.cpp:169-233) */
typedef struct {
    int nbits;            /* decoded bits of the codeword (768 FIC, 24*bitrate MSC) */
    int L[4];             /* blocks of 128 mother-code bits */
    int PI[4];            /* puncturing index 1..24 (0 = unused) */
    int n_in;             /* punctured input length (derived) */
} orc_prot;
int orc_prot_fic(orc_prot* p);
int orc_prot_eep(orc_prot* p, int bitrate, int profile_b, int level);     /* eep-protection.cpp:32-113 */
int orc_uep_table(int idx, int* bitrate, int* level, int* size_cu);         /* dab-constants.cpp:45-109 */
int orc_prot_uep(orc_prot* p, int bitrate, int level);                    /* uep-protection.cpp:27-167 */
void orc_depuncture(const orc_prot* p, const int8_t* in, int8_t* out /* 4*nbits+24 */);

/* fic-handler.cpp:111-230: 3 symbols (9216 soft bits) -> 12 x 256 bit-bytes, 12 ok flags; updates the
 * saturating 0..10 counter in *ratio */
void orc_fic_decode(const int8_t* soft9216, uint8_t* bits12x256, uint8_t* ok12, int* ratio);
int orc_crc16_bits(const uint8_t* bits, int n);                            /* MathHelper.h:53-80 */

/* dab-audio.cpp:113-164 + decoder_adapter.cpp:55-73: one sub-channel */
typedef struct {
    orc_prot prot;
    int frag;                 /* length*64 */
    int8_t* hist;             /* 16 x frag */
    int idx, count;
    int8_t* tmp; int8_t* vit; uint8_t* bits;
} orc_subch;
int orc_subch_init(orc_subch* s, const orc_prot* prot, int length_cu);
void orc_subch_free(orc_subch* s);
/* feed one CIF slice (frag soft bits); returns number of bytes written to out (0 during the first 16 CIFs, else nbits/8) */
int orc_subch_process(orc_subch* s, const int8_t* cif_slice, uint8_t* out_bytes);

/* dabplus_decoder.cpp:326-359 + libs/fec decode_rs.h (8,0x11D,fcr 0,prim 1,10 roots,pad 135) */
void orc_rs_superframe(uint8_t* sf, int len, int* corrected, int* uncorrectable);

/* DAB+ superframe filter (SuperframeFilter::Feed / CheckSync, dabplus_decoder.cpp:50-213): one record per decode attempt */
typedef struct {
    int32_t cif;             /* index of the logical frame that completed the 5-frame window */
    int32_t corrected, uncorrectable;   /* RSDecoder::DecodeSuperframe totals = FECInfo() arguments */
    int32_t sync;            /* CheckSync() result */
    int32_t format;          /* sf[2] (valid when sync) */
    int32_t num_aus, au_start[7];
    int32_t au_crc_ok;       /* bit i: access unit i passed its CRC */
    int32_t sf_slot;         /* device API: where the corrected superframe was stored, -1 = not kept */
} orc_sf_event;
uint16_t orc_crc16(const uint8_t* data, int len, int initial_invert, int final_invert, uint16_t poly);
int orc_sf_state_bytes(int frame_len);
int orc_superframe_feed(uint8_t* st, const uint8_t* frame, int len, int frame_index, orc_sf_event* ev, uint8_t* sf_out);
void orc_rs_encode120(const uint8_t* data110, uint8_t* parity10);          /* encode_rs.h (test input generation) */

/* TII (tii-decoder.cpp:189-383): one (NULL symbol, PRS) pair per call, every frame looked at.  rank[2][504] = position of
 * key float(err), err = -4..499, in the iteration order of the reference's unordered_map<float, uint64_t> when it is
 * filled for the first time ([0]) and refilled after clear() ([1]) -- see oracle/tii_order.cpp.  Returns the number of
 * onTIIMeasurement calls, in ascending (comb, pattern) order (the reference's order within a frame is unspecified). */
#define ORC_TII_NERR 504
typedef struct { int32_t frame, comb, pattern, delay_samples; float error; } orc_tii_event;
typedef struct orc_tii_state orc_tii_state;
size_t orc_tii_state_bytes(void);
void orc_tii_reset(orc_tii_state* st);
int orc_tii_frame(orc_tii_state* st, const orc_cf32* null2656, const orc_cf32* prs2048, const int32_t* rank,
                  orc_tii_event* ev, int max_ev, uint8_t* detect192);
void orc_tii_iteration_rank(int32_t* rank2x504);                            /* oracle/tii_order.cpp */

/* Full receiver: ofdm-processor.cpp:235-501 driving all of the above in lock step. */
typedef struct {
    int subch_id, start_cu, length_cu;
    orc_prot prot;
} orc_subch_cfg;

typedef struct {
    /* in */
    const orc_cf32* iq; int64_t n_samples;
    int disable_coarse, fft_placement;
    int n_subch; const orc_subch_cfg* subch;
    /* out, caller allocated, capacities in frames / bytes */
    uint8_t* fib; int fib_cap;              /* 33 B per FIB: ok + 32 bytes */
    float* cir; int cir_cap;                /* 2048 per frame */
    orc_cf32* con; int con_cap;             /* 1200 per frame */
    orc_cf32* nul; int nul_cap;             /* 2656 per frame */
    float* snr; int snr_cap;
    int32_t* corr; int corr_cap;            /* fine, coarse after each frame */
    int32_t* start_index; int64_t* frame_pos; int sidx_cap; /* findIndex result and absolute sample index of the sync buffer start */
    int8_t* soft; int soft_cap;             /* 75*3072 per frame (optional, may be NULL) */
    uint8_t** msc; int64_t* msc_cap; int64_t* msc_len;   /* per sub-channel byte streams */
    /* counts */
    int n_fib, n_frames, n_snr, n_sync_true, n_sync_false, n_cir;
    int freqsync_sel;          /* 0 = PatternOfZeros (default), 1 = GetMiddle, 2 = CorrelatePRS */
    /* decodeTII: state (orc_tii_state_bytes(), zeroed) or NULL; rank as for orc_tii_frame; events out */
    void* tii_state; const int32_t* tii_rank; orc_tii_event* tii_ev; int tii_cap; int n_tii;
} orc_run_io;
int orc_receiver_run(orc_run_io* io);

#ifdef __cplusplus
}
#endif
#endif
