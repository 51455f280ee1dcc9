/* include/dabphy_test.h -- test and measurement entry points of libdabphy_hip.so: timing drivers that run ONE stage alone on
 * device-resident data, and exhaustive device self-tests of two arithmetic shortcuts of the demod kernel.  tests/, tools/ and
 * bench.py's side measurements call them; a receiver needs only include/dabphy.h. */
#ifndef DABPHY_TEST_H
#define DABPHY_TEST_H

#include "dabphy.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- diagnostics: time one stage on device-resident data (HIP events on the handle's stream) ------------------
 * dabphy_time_demod: tiles `n_src` host frames (layout of dabphy_demod_frames) over n_ens x n_frames frame slots in
 *   HBM and runs the demod kernel `iters` times; *ms = mean kernel time.  mix/f_hz exercise the NCO path.
 * dabphy_time_viterbi: decodes n_codewords random-content codewords of nbits `iters` times; *ms_gather / *ms_decode.
 * dabphy_time_fused_msc: re-runs the fused decode launch of the last dabphy_process batch (every class it held, the FIC included)
 *   `iters` times with nothing else on the device; *ms = mean kernel time.  DABPHY_ERR_STATE before the first such batch and after
 *   anything that replaced a buffer the launch names (dabphy_set_subchannels, a larger batch, another seam's scratch). */
int dabphy_time_demod(dabphy_handle* h, const float* frames, uint32_t n_src, uint32_t n_ens, uint32_t n_frames,
                      int32_t mix, int32_t f_hz, uint32_t iters, float* ms);
int dabphy_time_viterbi(dabphy_handle* h, uint32_t nbits, uint32_t n_codewords, uint32_t iters, float* ms_gather,
                        float* ms_decode);
int dabphy_time_fused_msc(dabphy_handle* h, uint32_t iters, float* ms);
/* EXPERIMENT of round 6, off by default (profiles/r06_viterbi_split.txt: a measured loss).  on = 1: the forward waves of the lane-per-code-word
 * Viterbi kernel publish every group's decisions (per-group scratch, agent-scope release, a flag) and walk back only when no trellis is left
 * to run -- the walks of the whole launch then overlap the forward passes of other waves instead of following each group's own.  Same bytes,
 * 6.46 against 5.74-5.87 ms for the launch alone.  on = 3 adds the 24-register walker waves (k_traceback_fused, a sixth wave per SIMD): they
 * read stale decisions across XCDs -- WRONG bytes -- and are there for the record only.  From the next batch on. */
int dabphy_test_traceback_split(dabphy_handle* h, int32_t on);
/* dabphy_time_copy: a plain device-to-device copy of `bytes` bytes (16 bytes per lane and request, grid-stride, blocks_per_cu work-groups
 *   of 256 threads per compute unit, 0 = 4), `iters` passes after three warm-up passes; *gbytes_per_s = bytes read + bytes written per
 *   second.  What the device's HBM delivers to the simplest possible kernel on this box: the measured denominator bench.py prints
 *   next to the 8 TB/s specification (tools/ubench/copy_f4.hip is the stand-alone sweep of the same kernel). */
int dabphy_time_copy(dabphy_handle* h, uint64_t bytes, uint32_t blocks_per_cu, uint32_t iters, float* gbytes_per_s);

/* Device self-test of the reciprocal-based 127/x the demapper uses in place of the IEEE division sequence
 * (ofdm-decoder.cpp:208 computes 127.0f / l1_norm): every float x in [2^-100, 2^100] is divided both ways on the device.
 * counts[0] = mismatches of the 4-instruction variant, counts[1] = of the 6-instruction variant, counts[2] = values tried. */
int dabphy_selftest_div127(dabphy_handle* h, uint64_t* counts);
/* Device self-test of the one-instruction product by the unit twiddle tw[0] = (1, +-0) in the first two passes of the demod kernel's
 * FFT (kiss_fft.c:21-90 multiplies by it like by any other twiddle): 2^33 operand pairs -- every exponent, zeros, denormals, infinities
 * and NaNs included -- through both forms.  counts[0] = results that differ in a bit (two NaNs count as equal), counts[1] = pairs tried. */
int dabphy_selftest_unit_twiddle(dabphy_handle* h, uint64_t* counts);
/* Device self-test of the lane exchanges of the state-parallel Viterbi kernel (k_viterbi_sp.hip: v_permlane32_swap, v_permlane16_swap,
 * bank-masked row DPP moves, quad_perm DPP reads, v_readlane) against plain shuffles: the forms the
 * GPU-less execution model of tests/hipemu stands in for.  counts[0] = mismatches, counts[1] = values checked. */
int dabphy_selftest_pair_exchange(dabphy_handle* h, uint64_t* counts);   /* (also swap16 / partner of k_viterbi_sp2.hip) */

/* Which Viterbi kernel decoded the last dabphy_process batch (dabphy_config.decode_shape = 0 leaves the choice to the library):
 * *shape = numbered like dabphy_config.decode_shape: 1 lane per code word (k_viterbi_fused), 2 state-parallel with two code words per wavefront
 * (k_viterbi_sp2 + k_traceback_sp2), 3 state-parallel with one (k_viterbi_sp), 0 nothing decoded yet; *fused_classes = protection
 * classes (not counting the FIC) that rode in that launch -- the others took the two-kernel path.  Either pointer may be NULL. */
int dabphy_last_decode_plan(dabphy_handle* h, int32_t* shape, int32_t* fused_classes);

#ifdef __cplusplus
}
#endif

#endif /* DABPHY_TEST_H */
