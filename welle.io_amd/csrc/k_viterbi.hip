// welle.io_amd/csrc/k_viterbi.hip -- K=7 rate-1/4 Viterbi decoder, depuncturing gathers, FIB CRC.
//
// Replaces (reference file:line, relative to src/backend):
//   Viterbi::deconvolve / BFLY / chainback_viterbi    viterbi.cpp:227-339
//   FicHandler::processFicInput (depuncture, PRBS)    fic-handler.cpp:144-204
//   EEPProtection / UEPProtection::deconvolve         eep-protection.cpp:115-152, uep-protection.cpp:169-239
//   DabAudio::run time de-interleaver                 dab-audio.cpp:113-149
//   EnergyDispersal::dedisperse                       energy_dispersal.h:35-54
//   DecoderAdapter::addtoFrame bit packing            decoder_adapter.cpp:55-67
//   check_CRC_bits                                    various/MathHelper.h:53-80
//
// MI355X mapping.  The reference decodes one codeword at a time with 64 scalar states.  Here ONE LANE
// decodes ONE CODEWORD: a wavefront carries 64 independent codewords, the 64 path metrics of each live in
// 32 VGPRs as pairs of uint16 and a trellis step is 32 x (2 plain 32-bit additions, v_pk_min_u16, v_pk_sub_i16) in a
// pairing of states that rotates through six layouts (viterbi_acs.h), with no cross-lane traffic and no LDS.
// Decisions (64 bit per step and codeword) stream to HBM as one coalesced 8-byte store per lane and are read
// back by the same lane during traceback.  Exact-integer equivalence with the reference: metrics are uint16
// without wrap-around, decisions are "m0 > m1" evaluated on the true integers, ties keep the m0/m2 branch --
// the reference's own renormalisation schedule (viterbi.cpp:104-120) changes no decision, so none of it is mimicked.
#include "dabphy_kernels.h"
#include <cstdlib>
#include <dabphy_wave_ops.h>
#include "viterbi_acs.h"

namespace dabphy {

// Symbol word of one trellis step as the gathers store it.  s0..s3 = the four soft symbols 0..255 of the step (output j of
// the mother code, viterbi.cpp:233-238 mapping applied; 255 cannot occur: an int8 + 127 is at most 254).  Outputs 0 and 3 share a
// generator, so the engine takes x0 = (s0 - 127) + (s3 - 127), v1 = s1 - 127, v2 = s2 - 127 (viterbi_acs.h): packed
// int16 | int8 << 16 | int8 << 24, so the decoder spends three instructions on unpacking.
__device__ __forceinline__ uint32_t pack_step(uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3)
{
    return ((s0 + s3 - 254u) & 0xffffu) | (((s1 - 127u) & 0xffu) << 16) | ((s2 - 127u) << 24);
}
__device__ __forceinline__ uint32_t pack_step_word(uint32_t w) { return pack_step(w & 0xff, (w >> 8) & 0xff, (w >> 16) & 0xff, w >> 24); }
template <int F>
__device__ __forceinline__ uint2 step_from_word(uint32_t (&M)[32], uint32_t w, uint32_t ones)
{
    return acs::step<F>(M, (int)(int16_t)(w & 0xffffu), (int)(int8_t)((w >> 16) & 0xffu), (int)w >> 24, ones);
}

// Traceback of one codeword from state 0, skipping the 6 tail steps (chainback_viterbi, viterbi.cpp:313-339); bits are packed MSB
// first (decoder_adapter.cpp:61-67) into little-endian 32-bit words and XORed with the energy-dispersal sequence when asked to
// (fic-handler.cpp:206-208, energy_dispersal.h:51-53).  The decision words do not depend on the path, so 8 steps are fetched at a
// time and resolved from registers (acs::back: four instructions per step).  load_dec(step) = this lane's decision word of that trellis step.
template <typename LoadDec>
__device__ __forceinline__ void traceback(LoadDec&& load_dec, int nbits, uint32_t* __restrict__ out, bool live,
                                          int dedisperse, const uint32_t* __restrict__ prbs_words)
{
    uint32_t J = 0, outw = 0;
    uint32_t rho = (uint32_t)acs::dec_rot((nbits - 1) % 6);       // step t = n + 6 ran in layout t % 6 = n % 6
    uint2 dq[8], dn[8];                                           // this iteration's decision words and the next one's (already in flight)
#pragma unroll
    for (int k = 0; k < 8; k++) dq[k] = load_dec(nbits - 1 - k + 6);
    for (int n = nbits - 1; n >= 0; n -= 8) {
        {
            const int m = n >= 8 ? n - 8 : n;                     // the last iteration re-reads its own words
#pragma unroll
            for (int k = 0; k < 8; k++) dn[k] = load_dec(m - k + 6);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            acs::back(dq[k], J, outw, rho);
            rho = rho == 5 ? 0 : rho + 1;
        }
        if (((n - 7) & 31) == 0) {
            const int wi = (n - 7) >> 5;
            const uint32_t word = acs::back_word(outw);
            if (live) out[wi] = dedisperse ? word ^ prbs_words[wi] : word;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) dq[k] = dn[k];
    }
}

// One wavefront per work-group; a work-group walks groups g, g + gridDim.x, ... of 64 codewords (launch_viterbi sizes the
// grid to the device's wave slots, so very large batches do not queue tens of thousands of millisecond-long waves).
#ifndef VIT_OCC
#define VIT_OCC 1
#endif
__global__ void __launch_bounds__(64, VIT_OCC) k_viterbi(VitArgs A)
{
  const int lane = threadIdx.x;
#pragma unroll 1
  for (int g = A.c.g_begin + blockIdx.x; g < A.c.g_end; g += gridDim.x) {
    const int nsteps = A.c.nsteps, nbits = A.c.nbits;
    const uint32_t* __restrict__ sym = A.c.sym + (size_t)g * nsteps * 64 + lane;
    uint2* __restrict__ dec = A.c.dec + (size_t)g * nsteps * 64 + lane;

    const uint32_t ones = opaque_sgpr(0x01010101u);
    uint32_t M[32];
    acs::init(M);

    // Six trellis steps per iteration (one turn through the layouts); the symbols of the NEXT iteration are requested before the
    // current ones are consumed, so a wave waits neither on its own loads nor on the completion of its decision stores
    // (one vmcnt counter covers both on gfx9: a wait for a load also waits for every older store).
    const int nblk = nsteps / 6;
    uint32_t y[6];
#pragma unroll
    for (int k = 0; k < 6; k++) y[k] = sym[(size_t)(k < nsteps ? k : nsteps - 1) * 64];
    int s = 0, since_renorm = 0;
    for (int blk = 0; blk < nblk; blk++, s += 6) {
        uint32_t n[6];
        {
            const int i0 = (s + 12 <= nsteps) ? s + 6 : s;        // the last iteration re-reads valid memory
            const uint32_t* __restrict__ q = sym + (size_t)i0 * 64;
#pragma unroll
            for (int k = 0; k < 6; k++) n[k] = q[k * 64];
        }
        dec[(size_t)(s + 0) * 64] = step_from_word<0>(M, y[0], ones);
        dec[(size_t)(s + 1) * 64] = step_from_word<1>(M, y[1], ones);
        dec[(size_t)(s + 2) * 64] = step_from_word<2>(M, y[2], ones);
        dec[(size_t)(s + 3) * 64] = step_from_word<3>(M, y[3], ones);
        dec[(size_t)(s + 4) * 64] = step_from_word<4>(M, y[4], ones);
        dec[(size_t)(s + 5) * 64] = step_from_word<5>(M, y[5], ones);
        if (++since_renorm == acs::RENORM_BLOCKS) { acs::renorm(M); since_renorm = 0; }
#pragma unroll
        for (int k = 0; k < 6; k++) y[k] = n[k];
    }
    // fewer than six steps left (a length that is not a multiple of six: only through the dabphy_viterbi_batch seam)
    if (s < nsteps) { dec[(size_t)s * 64] = step_from_word<0>(M, sym[(size_t)s * 64], ones); s++; }
    if (s < nsteps) { dec[(size_t)s * 64] = step_from_word<1>(M, sym[(size_t)s * 64], ones); s++; }
    if (s < nsteps) { dec[(size_t)s * 64] = step_from_word<2>(M, sym[(size_t)s * 64], ones); s++; }
    if (s < nsteps) { dec[(size_t)s * 64] = step_from_word<3>(M, sym[(size_t)s * 64], ones); s++; }
    if (s < nsteps) { dec[(size_t)s * 64] = step_from_word<4>(M, sym[(size_t)s * 64], ones); s++; }

    const int cw = g * 64 + lane;
    traceback([&](int st) { return dec[(uint32_t)(st * 64)]; }, nbits, reinterpret_cast<uint32_t*>(A.c.out) + (size_t)cw * (nbits / 32),
              cw < A.c.n_cw, A.c.dedisperse, A.prbs_words);
  }
}

// ------------------------------------------------------------------------------------------ FIC gather
// soft bits of symbols 1..3 (9216 = 4 x 2304) -> depunctured, offset (viterbi.cpp:233-238) symbols of the
// 4 FIC codewords of each frame, written in the step-major layout k_viterbi reads.
__global__ void __launch_bounds__(256) k_fic_gather(FicGatherArgs A)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x;
    const int cw = g * 64 + lane;
    const int nsteps = A.c.nsteps;
    const bool live = cw < A.c.n_cw;
    const int q = cw & 3;
    const int bf = A.frame_sel ? (cw >> 2) * A.n_frames + (A.frame_sel - 1) : cw >> 2;     // bf = b * n_frames + f
    const int b = live ? bf / A.n_frames : 0;
    const FrameDesc d = A.desc[live ? bf : 0];
    const size_t ens_stride = A.soft_ens_stride ? A.soft_ens_stride : (size_t)A.soft_ring * A.frame_stride;
    const int8_t* __restrict__ src = A.soft + (size_t)b * ens_stride + (size_t)(d.frame_no % A.soft_ring) * A.frame_stride + 2304 * q;
    uint32_t* __restrict__ dst = A.c.sym + (size_t)g * nsteps * 64 + lane;
    for (int s = blockIdx.y * 4 + wave; s < nsteps; s += gridDim.y * 4) {
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int u = A.map[4 * s + j];
            int v = (u >= 0 && live && d.valid == 1) ? (int)src[u] : 0;
            v += 127; v = v < 0 ? 0 : v; v = v > 255 ? 255 : v;
            word |= (uint32_t)v << (8 * j);
        }
        dst[(size_t)s * 64] = pack_step_word(word);
    }
}

// ------------------------------------------------------------------------------------------ MSC gather
// Codeword order of an MSC class: cw = pair * R + r  ((ensemble, sub-channel) pair of the class's table -- every ensemble
// selects its own sub-channels, msc-handler.cpp:61-103 --, CIF r of this batch, R = 4 * n_frames), so the 64 lanes of a
// Viterbi wave are consecutive CIFs of one sub-channel (at most a few pairs per group).  Soft bit u of the logical frame emitted at CIF c comes from CIF
// c - 16 + map16[u & 15] (dab-audio.cpp:113,138-143: tempX[i] = hist[(idx + map[i & 15]) & 15][i], read BEFORE the
// current CIF is stored): the time de-interleaver is an address computation on the soft-bit ring, never a copy.
//
// Data movement: a work-group owns one group of 64 codewords and walks the trellis in tiles of GT_STEPS steps.
// For a tile it stages the needed byte columns [u_lo, u_hi) of every source CIF row (<= 64 + 15 rows per (b, m)
// segment) in LDS with coalesced dword loads, then every lane picks its 4 symbols per step from LDS and the wave
// stores one coalesced 256-byte row of the step-major symbol array.  HBM traffic is ~1.25x the soft bits read
// once plus the symbols written once; the per-byte global gather this replaces moved ~100x more through L2.
constexpr int GT_STEPS = 56;                    // 224 mother-code bits -> at most 224 + 3 source bytes = 57 dwords per row
constexpr int GT_PITCHW = 57;                   // odd number of dwords: consecutive rows start on consecutive banks
constexpr int GT_MAXROWS = 112;                 // two (b, m) segments: 64 + 2 * 15 = 94 rows

__global__ void __launch_bounds__(256) k_msc_gather(MscGatherArgs A)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile[GT_MAXROWS * GT_PITCHW];
    __shared__ long long s_rowsrc[GT_MAXROWS];
    __shared__ int s_pair[64], s_rowbase[64];
    __shared__ long long s_c[64];
    __shared__ int s_nrows;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int g = A.c.g_begin + blockIdx.x;
    const int cw = g * 64 + lane;
    const int nsteps = A.c.nsteps, R = 4 * A.n_frames;
    const bool live = cw < A.c.n_cw;
    const int pair = live ? cw / R : -1, r = live ? cw % R : 0;
    const MscPair mine = A.pairs[live ? pair : 0];
    const int b = mine.ens;
    const long long c_glob = 4 * A.desc[(size_t)b * A.n_frames].frame_no + r;   // CIF whose arrival emits this logical frame
    const size_t ens_stride = A.soft_ens_stride ? A.soft_ens_stride : (size_t)A.soft_ring * SOFT_PER_FRAME;
    const size_t ens_base = (size_t)b * ens_stride + (size_t)mine.start_bit;
    const int map16[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
    uint32_t* __restrict__ dst = A.c.sym + (size_t)g * nsteps * 64 + lane;

    if (wave == 0) { s_pair[lane] = pair; s_c[lane] = c_glob; }
    __syncthreads();
    if (t == 0) {
        // segments = runs of lanes with the same pair and consecutive CIFs; a segment [c_a .. c_b] needs the rows
        // of CIFs c_a - 16 .. c_b - 1
        int nrows = 0;
        for (int l = 0; l < 64;) {
            int e = l;
            while (e + 1 < 64 && s_pair[e + 1] == s_pair[l] && s_c[e + 1] == s_c[e] + 1) e++;
            const int need = (e - l) + 16;
            if (s_pair[l] < 0) { for (int k = l; k <= e; k++) s_rowbase[k] = -1; l = e + 1; continue; }
            if (nrows + need > GT_MAXROWS) { nrows = -1; break; }
            const MscPair pp = A.pairs[s_pair[l]];
            const long long pbase = (long long)pp.ens * (long long)ens_stride + pp.start_bit;
            for (int k = 0; k < need; k++) {
                const long long c_src = s_c[l] - 16 + k;
                long long src = -1;
                if (c_src >= 0) src = pbase + ((long long)((c_src >> 2) % A.soft_ring) * 75 + 3 + 18 * (int)(c_src & 3)) * SOFT_PER_SYM;
                s_rowsrc[nrows + k] = src;
            }
            for (int k = l; k <= e; k++) s_rowbase[k] = nrows + (k - l);
            nrows += need;
            l = e + 1;
        }
        s_nrows = nrows;
    }
    __syncthreads();
    const int nrows = s_nrows;
    const int rowbase = s_rowbase[lane];

    if (nrows < 0) {
        // many short segments (tiny batches): plain per-byte gather
        for (int s = wave; s < nsteps; s += 4) {
            uint32_t word = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int u = A.map[4 * s + j];
                int v = 0;
                if (u >= 0 && live) {
                    const long long c_src = c_glob - 16 + map16[u & 15];
                    if (c_src >= 0) v = A.soft[ens_base + ((size_t)((c_src >> 2) % A.soft_ring) * 75 + 3 + 18 * (int)(c_src & 3)) * SOFT_PER_SYM + u];
                }
                v += 127; v = v < 0 ? 0 : v; v = v > 255 ? 255 : v;
                word |= (uint32_t)v << (8 * j);
            }
            dst[(size_t)s * 64] = pack_step_word(word);
        }
        return;
    }

    const int8_t* tile8 = reinterpret_cast<const int8_t*>(tile);
    __shared__ __attribute__((aligned(16))) int s_map[4 * GT_STEPS];          // this tile's slice of the depuncturing map, pre-resolved to tile offsets
    for (int s0 = 0, ti = 0; s0 < nsteps; s0 += GT_STEPS, ti++) {
        const int u_lo = A.tiles[2 * ti], ndw = A.tiles[2 * ti + 1];       // first source byte (4-aligned), dwords per row
        const int s1 = (s0 + GT_STEPS < nsteps) ? s0 + GT_STEPS : nsteps;
        // map entry -> byte offset inside a tile row plus the row shift of the time de-interleaver; -1 = erasure
        if (t < 4 * (s1 - s0)) {
            const int u = A.map[4 * s0 + t];
            s_map[t] = (u >= 0) ? (map16[u & 15] * (GT_PITCHW * 4) + (u - u_lo)) : -1;
        }
        // rows travel HBM -> LDS by LDS-DMA: nothing waits between the (up to 28) row requests of a wave
        for (int row = wave; row < nrows; row += 4) {
            const long long src = s_rowsrc[row];
            if (lane < ndw) {
                if (src >= 0) lds_dma4(A.soft + src + u_lo + 4 * lane, &tile[row * GT_PITCHW]);
                else tile[row * GT_PITCHW + lane] = 0u;
            }
        }
        lds_dma_wait();
        __syncthreads();
        const int rb = rowbase * (GT_PITCHW * 4);
        auto pick = [&](int off) { int v = (off >= 0 && rowbase >= 0) ? (int)tile8[rb + off] : 0; v += 127; return v < 0 ? 0 : v; };   // viterbi.cpp:233-236 (an int8 + 127 never exceeds 254)
        int s = s0 + wave;
        for (; s + 4 < s1; s += 8) {                     // two steps per iteration: their LDS reads overlap
            const int4 ma = *reinterpret_cast<const int4*>(&s_map[4 * (s - s0)]);
            const int4 mb = *reinterpret_cast<const int4*>(&s_map[4 * (s + 4 - s0)]);
            const int a0 = pick(ma.x), a1 = pick(ma.y), a2 = pick(ma.z), a3 = pick(ma.w);
            const int b0 = pick(mb.x), b1 = pick(mb.y), b2 = pick(mb.z), b3 = pick(mb.w);
            dst[(size_t)s * 64] = pack_step(a0, a1, a2, a3);
            dst[(size_t)(s + 4) * 64] = pack_step(b0, b1, b2, b3);
        }
        if (s < s1) {
            const int4 ma = *reinterpret_cast<const int4*>(&s_map[4 * (s - s0)]);
            const int a0 = pick(ma.x), a1 = pick(ma.y), a2 = pick(ma.z), a3 = pick(ma.w);
            dst[(size_t)s * 64] = pack_step(a0, a1, a2, a3);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ fused decode
// k_viterbi_fused = k_msc_gather / k_fic_gather + k_viterbi in one kernel, for EVERY class of a batch in one launch: the step-word
// array (4 bytes per trellis step and code word written by a gather and read back by the decoder) never exists, and the classes of a
// heterogeneous multiplex share one grid instead of queueing one partly filled launch each.
//
// Work: a list of (class, group of 64 code words), longest code words first, handed out through an atomic cursor -- every
// work-group is ONE wave that lives for the whole launch (one per resident wave slot) and pulls the next group when it has finished
// one, so SIMDs stay evenly loaded whatever the mix of code word lengths.  Its decision scratch is per work-group, not per group.
//
// A wave (64 code words) keeps in LDS a sliding WINDOW of the soft-bit rows it decodes from:
//   rows    MSC class: the source CIFs of its code words.  Code words are consecutive CIFs of consecutive (ensemble, sub-channel)
//           pairs; byte u of the frame emitted at CIF c comes from CIF c - 16 + map16[u & 15] (time de-interleaver,
//           dab-audio.cpp:113,138-143), so a run of n lanes of one pair (a segment) needs n + 15 rows: 64 + 15 * segments in all.
//           Batches of >= 64 CIFs per sub-channel give <= 2 segments (96 rows), >= 16 CIFs <= 5 (144 rows), >= 4 CIFs <= 17 (324
//           rows): three builds of the kernel.  FIC: one row per code word (its 2304 punctured bits, fic-handler.cpp:158-191), no skew.
//   columns 16-byte windows of the punctured bit stream: window w = bytes [16 w, 16 w + 16).  Two windows are resident (slot w & 1),
//           the next one travels HBM -> LDS by LDS-DMA (requests of 12 rows x 4 dwords, no VGPRs) while the current ones are consumed;
//           u & 15 is the column inside a window, so the de-interleaver's row skew is a function of the column.
// Sources are addressed through a buffer resource that starts at the ring slice of the wave's FIRST ensemble (a wave spans a handful
// of consecutive ensembles; the host checks that span against the 4 GiB a 32-bit offset reaches -- whole-ring offsets wrapped for
// B x (F + 5) > 18 641 frame slots in round 3); rows without a source CIF point at that ensemble's zero frame.
// Everything that depends only on the step -- which of the 4 mother-code bits are punctured, where the others lie in the window ring,
// when a window dies -- is the same for all lanes and comes from a per-class table (MscStep, built on the host from the depuncturing
// map) through scalar loads.  Per step a lane adds its row base to four uniform offsets, reads four bytes from LDS (an erasure
// reads a zero from a third, never written slot) and forms the branch metrics; the trellis and the traceback are k_viterbi's.
constexpr int FM_PITCH = MSC_ROW_PITCH;           // bytes per row of a window slot: 16 window bytes + 4 of padding.  FIVE dwords per row, so the
                                                  // byte reads of 32 consecutive rows (lanes) at one column fall into 32 different banks; with
                                                  // the 16-byte pitch of round 2 they were 4-way conflicted (SQ_LDS_BANK_CONFLICT / IDX_ACTIVE 0.72)
constexpr int FM_REQ_ROWS = 12;                   // rows per LDS-DMA request: 12 x 5 = 60 lanes, lane l moves dword (l % 5) of row l / 5 (dword 4 = the padding: idle)
template <int ROWS> struct FmGeom {
    static constexpr int SLOT = ROWS * FM_PITCH;  // bytes per window slot: [row][20]
    static constexpr int ZERO = 2 * SLOT;         // third slot: zeros (erasures, viterbi.cpp:233-238 maps soft value 0 to symbol 127)
    static constexpr int ROWPTR = 3 * SLOT;       // then the rows' sources: byte offset / 16 from the wave's base
    static constexpr int LDS = ROWPTR + ROWS * 4;
    static constexpr int NREQ = ROWS / FM_REQ_ROWS;
    static_assert(ROWS % FM_REQ_ROWS == 0 && 3 * SLOT + 64 * FM_PITCH <= (int)MSC_OFF_MASK, "window ring geometry");
};
// cache policy of the decision traffic (8 bytes per trellis step and code word, written once, read once ~a millisecond later by the
// same wave), measured in round 4 (profiles/r04_viterbi_cache_policy.txt): `nt` on the traceback's loads -2.4 % on the launch alone and
// -1 ... -2.5 % on the step; `nt` on the stores +1.4 % alone (and no gain in the step), both together = neither.  2 = nt, 16 = sc1.
#ifndef FM_DEC_STORE_AUX
#define FM_DEC_STORE_AUX 0
#endif
#ifndef FM_DEC_LOAD_AUX
#define FM_DEC_LOAD_AUX 2
#endif

#ifndef VITM_OCC
#define VITM_OCC 5
#endif
// ---- the traceback as a pass of its own (FusedArgs::done != nullptr).  A group's decisions go to a scratch region of its own; the wave
// that ran its trellis RELEASES them (agent scope: the per-XCD L2s are not coherent with each other) and raises done[item]; whoever
// pulls the item from the next_tb cursor -- the waves of k_traceback_fused, which need 32 registers and ride as a SIXTH wave per SIMD beside
// five forward waves of 96, and at the end of the launch the forward waves themselves -- waits for the flag, acquires and walks back.
// The forward pass is bound by vector issue, the walk by its reads: side by side instead of one after the other in every wave.
__device__ __forceinline__ void tb_publish(const FusedArgs& A, uint32_t item, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    lds_dma_wait();                                        // s_waitcnt vmcnt(0) of our own (the compiler may drop the fence's when it knows the counter empty: MI355X_MICROARCH.md, inter-workgroup visibility)
    // (the flag is polled with read-modify-write atomics: they execute where no per-XCD L2 can hold a stale copy.  The store is issued by
    // EVERY lane, same word, same value: with `if (lane == 0)` around it hipcc 7.2 structurised the work loop into a lane-masked loop that
    // never terminated on the device -- profiles/r06_viterbi_split.txt)
    (void)lane; __hip_atomic_store(A.done + item, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (every lane the same word: no divergent region around a store whose address is wave-uniform)
}
// LEAN = the 32-register form (eight decision words in flight, no double buffering); otherwise `traceback` as the forward waves have it
template <bool LEAN>
__device__ __forceinline__ void tb_consume(const FusedArgs& A, int lane)
{
    const DABPHY_CONST_AS uint32_t* const dec_off = as_constant(A.dec_off);
    const DABPHY_CONST_AS FusedClass* const classes = as_constant(A.cls);
    const DABPHY_CONST_AS uint32_t* const work = as_constant(A.work);
    const uint32_t lane8 = (uint32_t)lane * 8u;
#pragma unroll 1
    for (;;) {
        uint32_t item = 0;
        if (lane == 0) item = atomicAdd(A.next_tb, 1u);
        item = (uint32_t)uniform_i32(__shfl((int)item, 0));
        if (item >= A.n_work) break;
        // (bounded: a flag that never comes -- a forward wave that died, a protocol error -- must not hang the device.  After 2^19 polls,
        // a few seconds, the wave gives up, counts the item in done[n_work + 1] and walks back whatever the scratch holds: the host checks the
        // counter after the launch and fails the call)
        uint32_t polls = 0;
        for (;;) {
            uint32_t f = 0;
            if (lane == 0) f = atomicAdd(A.done + item, 0u);
            f = (uint32_t)uniform_i32(__shfl((int)f, 0));
            if (f) break;
            __builtin_amdgcn_s_sleep(127);
            if (++polls > (1u << 19)) { if (lane == 0) atomicAdd(A.done + A.n_work + 1, 1u); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const uint32_t wk = work[item];
        const DABPHY_CONST_AS FusedClass& C = classes[wk >> 24];
        const int g = (int)(wk & 0xffffffu);
        const int nbits = C.nbits, n_cw = C.n_cw;
        const BufRsrc dec_rs = buf_rsrc(A.dec + (size_t)dec_off[item] * 64);
        const int cw_out = g * 64 + lane;
        uint32_t* const out = reinterpret_cast<uint32_t*>(C.out) + (size_t)cw_out * (nbits / 32);
        const bool live = cw_out < n_cw;
        if constexpr (!LEAN) {
            traceback([&](int st) { return buf_load_b64<FM_DEC_LOAD_AUX>(dec_rs, lane8, (uint32_t)st * 512u); }, nbits, out, live, C.dedisperse, A.prbs_words);
        } else {
            uint32_t J = 0, outw = 0;
            uint32_t rho = (uint32_t)acs::dec_rot((nbits - 1) % 6);
#pragma unroll 1
            for (int n = nbits - 1; n >= 0; n -= 8) {                // (nbits is a multiple of 8: 768, 24 * bit rate)
                uint2 dq[8];
#pragma unroll
                for (int k = 0; k < 8; k++) dq[k] = buf_load_b64<FM_DEC_LOAD_AUX>(dec_rs, lane8, (uint32_t)(n - k + 6) * 512u);
#pragma unroll
                for (int k = 0; k < 8; k++) { acs::back(dq[k], J, outw, rho); rho = rho == 5 ? 0 : rho + 1; }
                if (((n - 7) & 31) == 0) {
                    const int wi = (n - 7) >> 5;
                    const uint32_t word = acs::back_word(outw);
                    if (live) out[wi] = C.dedisperse ? word ^ A.prbs_words[wi] : word;
                }
            }
        }
    }
}

template <int ROWS, int OCC, bool SPLIT>
__global__ void __launch_bounds__(64, OCC) k_viterbi_fused(FusedArgs A)
{
  using G = FmGeom<ROWS>;
  __shared__ __attribute__((aligned(16))) uint8_t lds[G::LDS];
  const int lane = threadIdx.x;
  const int F = A.n_frames, R = 4 * F;
  const uint32_t lane8 = (uint32_t)lane * 8u;                       // byte offset of this lane in a decision row
  const DABPHY_CONST_AS uint32_t* const dec_off = as_constant(A.dec_off);
  const DABPHY_CONST_AS FusedClass* const classes = as_constant(A.cls);
  const DABPHY_CONST_AS uint32_t* const work = as_constant(A.work);
  uint32_t* const rowptr = reinterpret_cast<uint32_t*>(lds + G::ROWPTR);
#pragma unroll 1
  for (;;) {
    uint32_t item = 0;
    if (lane == 0) item = atomicAdd(A.next, 1u);
    item = (uint32_t)uniform_i32(__shfl((int)item, 0));
    if (item >= A.n_work) break;
    const uint32_t wk = work[item];
    // decision scratch: this work-group's own (every group it pulls reuses it), or -- when the launch holds a few very long code words
    // among many short ones, so that a scratch of the longest per work-group would be many times what all groups need -- this group's own
    // (wave-uniform base + a 32-bit lane/step offset: the stores take the scalar-base addressing mode, one VGPR instead of a 64-bit pointer)
    const BufRsrc dec_rs = buf_rsrc(A.dec + (A.dec_off ? (size_t)dec_off[item] * 64 : (size_t)blockIdx.x * A.dec_slot_cells));
    const DABPHY_CONST_AS FusedClass& C = classes[wk >> 24];
    const int g = (int)(wk & 0xffffffu);
    const int nsteps = C.nsteps, nbits = C.nbits, n_cw = C.n_cw, n_windows = C.n_windows;
    const long long cw0 = (long long)g * 64;
    const int cw = (int)cw0 + lane;
    const uint32_t zero16 = (uint32_t)(((size_t)A.soft_ring * SOFT_PER_FRAME) >> 4);     // the zero frame behind the wave's first ensemble's ring slice
    __syncthreads();                                                                     // (one wave per work-group: orders the LDS reuse between groups)
    // zero the window slots and the erasure slot
    for (int i = lane; i < G::ROWPTR / 16; i += 64) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);
    // ---- which rows this wave needs, and where they lie
    int rb, nrows, pb0;
    if (C.kind == 0) {
        // segments = runs of lanes with the same (ensemble, sub-channel) pair: pair0, pair0 + 1, ... of the class's table; the first one
        // starts at CIF r0 of its pair, the others at CIF 0.  Segment j >= 1 begins at lane n0 + (j - 1) R and at row n0 + 15 + (j - 1) (R + 15).
        const MscPair* __restrict__ pairs = C.pairs;
        const int pair0 = (int)(cw0 / R), r0 = (int)(cw0 - (long long)pair0 * R);
        const int n0 = R - r0;
        const long long cw_last = cw0 + 63 < n_cw ? cw0 + 63 : (long long)n_cw - 1;
        const int nseg = (int)(cw_last / R) - pair0 + 1;
        nrows = 64 + 15 * nseg;
        pb0 = pairs[pair0].ens;                                                          // (ensembles ascend along the table: the wave's lowest)
        int seg = cw / R - pair0; if (seg > nseg - 1) seg = nseg - 1;                    // (dead lanes ride in the last segment: their output is dropped)
        rb = lane + 15 * seg;                                                            // row of CIF c_glob - 16
        for (int row = lane; row < ROWS; row += 64) {
            uint32_t src = zero16;                     // no such CIF yet (start of a stream) / row not used by this wave
            if (row < nrows) {
                int j = 0, idx = row;
                if (row >= n0 + 15) { const int q = row - (n0 + 15); j = 1 + q / (R + 15); idx = q - (j - 1) * (R + 15); }
                const int pr = pair0 + j;
                if (j == 0 || (long long)pr * R < n_cw) {
                    const MscPair pp = pairs[pr];
                    const int pb = pp.ens;
                    const long long c_src = 4 * A.desc[(size_t)pb * F].frame_no + (j == 0 ? r0 : 0) - 16 + idx;
                    if (c_src >= 0)
                        src = (uint32_t)(((size_t)(pb - pb0) * A.ens_stride + (size_t)pp.start_bit +
                                          ((size_t)((c_src >> 2) % A.soft_ring) * 75 + 3 + 18 * (int)(c_src & 3)) * SOFT_PER_SYM) >> 4);     // (start_bit is a multiple of 64)
                }
            }
            rowptr[row] = src;
        }
    } else {
        // FIC: code word -> (ensemble, frame slot, quarter); one row per code word: the 2304 soft bits of that quarter of symbols 1..3
        auto bf_of = [&](int c) { return c >> 2; };
        nrows = 64; rb = lane;
        pb0 = bf_of((int)cw0) / F;
        for (int row = lane; row < ROWS; row += 64) {
            uint32_t src = zero16;
            const int cwr = (int)cw0 + row;
            if (row < 64 && cwr < n_cw) {
                const int bf = bf_of(cwr), b = bf / F;
                const FrameDesc& d = A.desc[bf];
                if (d.valid == 1)
                    src = (uint32_t)(((size_t)(b - pb0) * A.ens_stride + (size_t)(d.frame_no % A.soft_ring) * SOFT_PER_FRAME + (size_t)2304 * (cwr & 3)) >> 4);
            }
            rowptr[row] = src;
        }
    }
    __syncthreads();
    // window w -> slot w & 1: requests of 12 rows x 5 dwords (60 lanes; the fifth dword of a row is padding and its lane stays idle).
    // The row addresses are read first, then the requests go out back to back: no LDS round trip between them.  Sources are
    // addressed through a buffer resource on the first ensemble's ring slice: row offset (one VGPR) + the window's column (one SGPR).
    const BufRsrc soft_rs = buf_rsrc_4g(A.soft + (size_t)uniform_i32(pb0) * A.ens_stride);
    const int nreq = uniform_i32((nrows + FM_REQ_ROWS - 1) / FM_REQ_ROWS);
    auto load_window = [&](int w) {
        uint8_t* slot = lds + (w & 1) * G::SLOT;
        const uint32_t l = opaque_vgpr((uint32_t)lane);             // (nothing of this may be hoisted out of the step loop)
        const uint32_t r5 = (l * 205u) >> 10, q5 = l - 5u * r5;     // l / 5, l % 5 for l < 64
        if (q5 < 4u && l < 5u * FM_REQ_ROWS) {
            if constexpr (G::NREQ == 8) {
                uint32_t src[8];
#pragma unroll
                for (int k = 0; k < 8; k++) src[k] = rowptr[FM_REQ_ROWS * k + r5];
#pragma unroll
                for (int k = 0; k < 8; k++) buf_dma4(soft_rs, (src[k] << 4) + 4u * q5, 16u * (uint32_t)w, slot + FM_REQ_ROWS * FM_PITCH * k);
            } else {
#pragma unroll 1
                for (int k0 = 0; k0 < nreq; k0 += 3) {              // (G::NREQ is a multiple of three)
                    uint32_t src[3];
#pragma unroll
                    for (int k = 0; k < 3; k++) src[k] = rowptr[FM_REQ_ROWS * (k0 + k) + r5];
#pragma unroll
                    for (int k = 0; k < 3; k++) if (k0 + k < nreq) buf_dma4(soft_rs, (src[k] << 4) + 4u * q5, 16u * (uint32_t)w, slot + FM_REQ_ROWS * FM_PITCH * (k0 + k));
                }
            }
        }
    };
    load_window(0);
    if (n_windows > 1) load_window(1);

    const uint32_t ones = opaque_sgpr(0x01010101u);
    uint32_t M[32];
    acs::init(M);
#ifdef FM_EXP_BROADCAST          // (timing experiment only: every lane reads the same bytes -- no bank conflicts, wrong results)
    const uint32_t lane_base = 0u; (void)rb;
#else
    const uint32_t lane_base = (uint32_t)rb * (uint32_t)FM_PITCH;
#endif
    const int8_t* lds_c = reinterpret_cast<const int8_t*>(lds);        // (sign-extending byte reads)
    auto fetch = [&](const MscStep& d, int (&y)[4]) {
#ifdef FM_EXP_NOLDS              // (timing experiment only)
        y[0] = (int)(d.off01 & 127) - 60; y[1] = (int)(d.off23 & 127) - 61; y[2] = lane - 30; y[3] = 5; return;
#endif
        y[0] = lds_c[lane_base + (d.off01 & MSC_OFF_MASK)]; y[1] = lds_c[lane_base + ((d.off01 >> 16) & MSC_OFF_MASK)];
        y[2] = lds_c[lane_base + (d.off23 & 0xffffu)]; y[3] = lds_c[lane_base + (d.off23 >> 16)];
    };
    lds_dma_wait();                                                  // windows 0 and 1 have landed
    int cur[4];
    int next_window = 2;
    // Per step: the soft values of this step (signed; symbol = value + 127, viterbi.cpp:233-236 -- the clamp at 0 only matters for
    // -128, which the demapper never produces: |v| <= 127 by construction, ofdm-decoder.cpp:208-212, asserted by the tests) are
    // taken from the registers the previous step requested them into; if this step was the last to read a window, the window after
    // the next is requested into its slot; then the 32 butterflies and the decision store; then the bytes of step s + 1 are
    // requested into the same registers (after waiting for its window, if it is the first step to read a new one).
    // The wait for a window is s_waitcnt vmcnt(2): memory operations complete in order, the load is older than the last two decision
    // stores (the host checks that when it builds the table), so the newest stores stay in flight.
    // Descriptors come through the constant address space (scalar loads), six steps ahead (six entries of padding end the table).  Scalar loads share the LDS counter and return out of order, so any wait for LDS bytes also waits for them: they are
    // issued right AFTER the first such wait of a block and have a whole step to arrive.
    const DABPHY_CONST_AS MscStep* steps = as_constant(C.steps);
    auto desc_at = [&](int i) { MscStep d; d.off01 = steps[i].off01; d.off23 = steps[i].off23; return d; };
    auto one_step = [&](auto fc, int s, const MscStep& d_cur, const MscStep& d_next, auto&& after_wait) {
        constexpr int FL = decltype(fc)::value;
        int x0 = cur[0] + cur[3], v1 = cur[1], v2 = cur[2];
        asm volatile("" : "+v"(x0), "+v"(v1), "+v"(v2));                  // this step's LDS reads have been consumed here ...
        after_wait();
#ifndef FM_EXP_NODMA             // (timing experiment only)
        if (d_cur.off01 & MSC_LOAD_NEXT) { wave_converge(); load_window(next_window); next_window++; }     // ... by every lane, before the dying window's slot is refilled
#endif
        const uint2 dd = acs::step<FL>(M, x0, v1, v2, ones);
#ifdef FM_EXP_NOSTORE            // (timing experiment only)
        if (dd.x == 0x12345u && dd.y == 0x777u) buf_store_b64<FM_DEC_STORE_AUX>(dec_rs, lane8, (uint32_t)s * 512u, dd);
#else
        buf_store_b64<FM_DEC_STORE_AUX>(dec_rs, lane8, (uint32_t)s * 512u, dd);
#endif
#ifndef FM_EXP_NOWAIT            // (timing experiment only)
        if (d_next.off01 & MSC_FIRST_USE) lds_dma_wait_but<2>();
#endif
        fetch(d_next, cur);
    };
    auto nothing = []() {};
    MscStep d0 = desc_at(0), d1 = desc_at(1), d2 = desc_at(2), d3 = desc_at(3), d4 = desc_at(4), d5 = desc_at(5);
    fetch(d0, cur);
    int since_renorm = 0;
    for (int s = 0; s < nsteps; s += 6) {                            // (nsteps is a multiple of six for every sub-channel size and the FIC: 24 * bitrate + 6, 774)
        // the descriptors of the next block arrive pair by pair (four scalar registers in flight, not twelve: the loop is short of them)
        MscStep ea, eb;
        one_step(std::integral_constant<int, 0>{}, s, d0, d1, [&]() { ea = desc_at(s + 6); eb = desc_at(s + 7); });
        one_step(std::integral_constant<int, 1>{}, s + 1, d1, d2, nothing);
        d0 = ea; d1 = eb;
        one_step(std::integral_constant<int, 2>{}, s + 2, d2, d3, [&]() { ea = desc_at(s + 8); eb = desc_at(s + 9); });
        one_step(std::integral_constant<int, 3>{}, s + 3, d3, d4, nothing);
        d2 = ea; d3 = eb;
        one_step(std::integral_constant<int, 4>{}, s + 4, d4, d5, [&]() { ea = desc_at(s + 10); eb = desc_at(s + 11); });
        one_step(std::integral_constant<int, 5>{}, s + 5, d5, d0, nothing);
        d4 = ea; d5 = eb;
        if (++since_renorm == acs::RENORM_BLOCKS) { acs::renorm(M); since_renorm = 0; }
    }
    lds_dma_wait();                                                  // (no load may still be in flight when the next group reuses the slots)

#ifndef FM_EXP_NOTRACE            // (timing experiment only)
    if constexpr (SPLIT) {
        tb_publish(A, item, lane);                                   // this group's decisions are complete: somebody else walks them back
    } else {
        // traceback: as in k_viterbi, the decision words through the same buffer resource the stores took
        const int cw_out = g * 64 + lane;                            // (recomputed: nothing but the trellis lives across the step loop)
        traceback([&](int st) { return buf_load_b64<FM_DEC_LOAD_AUX>(dec_rs, lane8, (uint32_t)st * 512u); }, nbits,
                  reinterpret_cast<uint32_t*>(C.out) + (size_t)cw_out * (nbits / 32), cw_out < n_cw, C.dedisperse, A.prbs_words);
    }
#endif
  }
#ifndef FM_EXP_NOTRACE
  if constexpr (SPLIT) tb_consume<false>(A, lane);                   // no trellis left to run: this wave walks back too (the last groups' walks get every wave of the device)
#endif
}

// the traceback waves: 32 registers, no LDS -- one per SIMD rides beside the five forward waves for the whole launch
__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(32))) k_traceback_fused(FusedArgs A)
{
    tb_consume<true>(A, threadIdx.x);
}

static int device_simds()
{
    static int n_simd = 0;
    if (!n_simd) {
        int dev = 0; hipDeviceProp_t p;
        n_simd = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? 4 * p.multiProcessorCount : 1024;
    }
    return n_simd;
}
// waves per SIMD of the three builds: 96 rows -- 6.1 KiB of LDS, 96 VGPRs: five; 144 rows -- 9 KiB: four; 324 rows -- 20.3 KiB: seven per CU
constexpr int FUSED_OCC[FUSED_VARIANTS] = {VITM_OCC, 4, 1};
int fused_wave_slots(int variant)
{
    int occ = FUSED_OCC[variant];
#ifdef DABPHY_EXPERIMENTS
    { const char* e = getenv("DABPHY_VITM_SLOTS"); if (e && atoi(e) > 0) occ = atoi(e); }
    // (round 5 experiment: leave a few per cent of the wave slots to the next batch's synchroniser, profiles/r05_sync_tail.txt)
    { const char* e = getenv("DABPHY_VITM_SLOTS_PCT"); if (e && atoi(e) > 0 && atoi(e) <= 100) return (int)((long long)occ * device_simds() * atoi(e) / 100); }
#endif
    return occ * device_simds();
}
void launch_viterbi_fused(const FusedArgs& a, int variant, int n_slots, hipStream_t s, const FusedSplit* split)
{
    if (a.n_work == 0 || n_slots <= 0) return;
    // One work-group (= one wave) per resident wave slot, or fewer when there is less work; the groups are pulled from the list.
    hipError_t e = hipMemsetAsync(a.next, 0, sizeof(uint32_t), s); (void)e;
    if (a.done && split) {
        // the traceback beside the forward pass: flags and the second cursor cleared, the traceback waves forked off onto their own stream
        // (forward launch first: the GPU-less execution model runs launches to completion in order), joined back behind the launch
        e = hipMemsetAsync(a.done, 0, ((size_t)a.n_work + 2) * sizeof(uint32_t), s);      // flags, the walkers' cursor, the count of flags given up on
        if (split->tb_stream) { e = hipEventRecord(split->fork, s); e = hipStreamWaitEvent(split->tb_stream, split->fork, 0); }
        if (variant == 0) hipLaunchKernelGGL((k_viterbi_fused<FUSED_ROWS[0], FUSED_OCC[0], true>), dim3(n_slots), dim3(64), 0, s, a);
        else if (variant == 1) hipLaunchKernelGGL((k_viterbi_fused<FUSED_ROWS[1], FUSED_OCC[1], true>), dim3(n_slots), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((k_viterbi_fused<FUSED_ROWS[2], FUSED_OCC[2], true>), dim3(n_slots), dim3(64), 0, s, a);
        if (split->tb_stream) {                         // (nullptr = diagnosis: the forward waves walk everything back at the end of theirs)
            const int n_tb = std::min<int>((int)a.n_work, device_simds());
            hipLaunchKernelGGL(k_traceback_fused, dim3(n_tb), dim3(64), 0, split->tb_stream, a);
            e = hipEventRecord(split->join, split->tb_stream);
            e = hipStreamWaitEvent(s, split->join, 0);
        }
        return;
    }
    if (variant == 0) hipLaunchKernelGGL((k_viterbi_fused<FUSED_ROWS[0], FUSED_OCC[0], false>), dim3(n_slots), dim3(64), 0, s, a);
    else if (variant == 1) hipLaunchKernelGGL((k_viterbi_fused<FUSED_ROWS[1], FUSED_OCC[1], false>), dim3(n_slots), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((k_viterbi_fused<FUSED_ROWS[2], FUSED_OCC[2], false>), dim3(n_slots), dim3(64), 0, s, a);
}

// ------------------------------------------------------------------------------------------ new pairs
// A sub-channel selected in mid-stream (MscHandler::addSubchannel while the receiver runs) starts with an empty time de-interleaver:
// its first logical frame leaves 16 CIFs later (dab-audio.cpp:146-149).  The CIF count at which a new pair joined is only known on the
// device (the synchroniser runs ahead of the host): the first batch decoded with the pair writes it into the table.
__global__ void k_pair_cif0(MscPair* pairs, int n_pairs, const FrameDesc* desc, int n_frames)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_pairs && pairs[p].cif0 < 0) pairs[p].cif0 = 4 * desc[(size_t)pairs[p].ens * n_frames].frame_no;
}
void launch_pair_cif0(MscPair* pairs, int n_pairs, const FrameDesc* desc, int n_frames, hipStream_t s)
{
    if (n_pairs > 0) hipLaunchKernelGGL(k_pair_cif0, dim3((n_pairs + 255) / 256), dim3(256), 0, s, pairs, n_pairs, desc, n_frames);
}

// ------------------------------------------------------------------------------------------ linear gather
// Input as the reference's seams hand it over: in[cw][...] int8, either already depunctured
// (Viterbi::deconvolve, viterbi.cpp:227) or punctured with a depuncturing map (Protection::deconvolve).
__global__ void __launch_bounds__(256) k_lin_gather(LinGatherArgs A)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x, cw = g * 64 + lane;
    const bool live = cw < A.c.n_cw;
    const int8_t* __restrict__ src = A.in + (size_t)(live ? cw : 0) * A.in_stride;
    uint32_t* __restrict__ dst = A.c.sym + (size_t)g * A.c.nsteps * 64 + lane;
    for (int s = blockIdx.y * 4 + wave; s < A.c.nsteps; s += gridDim.y * 4) {
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int u = A.map ? (int)A.map[4 * s + j] : 4 * s + j;
            int v = (live && u >= 0) ? (int)src[u] : 0;
            v += 127; v = v < 0 ? 0 : v; v = v > 255 ? 255 : v;
            word |= (uint32_t)v << (8 * j);
        }
        dst[(size_t)s * 64] = pack_step_word(word);
    }
}

// ------------------------------------------------------------------------------------------ FIB CRC
// CRC-16-CCITT (x^16+x^12+x^5+1, register preset to ones, transmitted inverted) over 30 data bytes,
// compared with the 2 stored bytes: the byte-wise form of check_CRC_bits (MathHelper.h:53-80).
__global__ void k_fib_crc(CrcArgs A)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = A.n_ens * (A.frame_sel ? 1 : A.n_frames) * 12;
    if (i >= n) return;
    const int o = A.frame_sel ? ((i / 12) * A.n_frames + (A.frame_sel - 1)) * 12 + i % 12 : i;      // FIB index in [B][F][12]
    // the 32 bytes of a FIB as two 16-byte loads; bytes are consumed in transmission order (little-endian words)
    const uint4* p4 = reinterpret_cast<const uint4*>(A.fib + (size_t)i * 32);
    const uint4 qa = p4[0], qb = p4[1];
    const uint32_t w[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
    uint8_t p[32];
#pragma unroll
    for (int k = 0; k < 32; k++) p[k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
    uint32_t crc = 0xFFFF;
#pragma unroll
    for (int k = 0; k < 30; k++) {
        crc ^= (uint32_t)p[k] << 8;
#pragma unroll
        for (int bit = 0; bit < 8; bit++) crc = (crc & 0x8000) ? ((crc << 1) ^ 0x1021) & 0xFFFF : (crc << 1) & 0xFFFF;
    }
    crc ^= 0xFFFF;
    const bool valid = A.desc[o / 12].valid == 1;
    A.ok[o] = (valid && crc == (((uint32_t)p[30] << 8) | p[31])) ? 1 : 0;
}

// saturating success counter (fic-handler.cpp:219-229), one thread per ensemble walks its FIBs in order
__global__ void k_fic_ratio(CrcArgs A)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= A.n_ens) return;
    int r = A.state[b].fic_ratio;
    const int f_end = A.frame_count > 0 ? A.frame_first + A.frame_count : A.n_frames;
    for (int f = A.frame_first; f < f_end; f++) {
        const FrameDesc& d = A.desc[(size_t)b * A.n_frames + f];
        if (d.valid != 1) continue;
        // what the reference's synchroniser sees before this frame (ofdm-processor.cpp:397) against what ours saw when it ran ahead
        if ((int)(!A.disable_coarse && r * 10 < 50) != d.coarse_ran) {
            if (A.state[b].stale_ratio_frames == 0) A.state[b].first_stale_frame = d.frame_no;
            A.state[b].stale_ratio_frames += 1;
            if (!d.coarse_ran || d.coarse_step != 0) {          // a needless consultation that moved nothing changes nothing
                if (A.state[b].effective_stale_frames == 0) A.state[b].first_effective_frame = d.frame_no;
                A.state[b].effective_stale_frames += 1;
                if (A.any_effective) *A.any_effective = 1;
            }
        }
        for (int k = 0; k < 12; k++) {
            if (A.ok[((size_t)b * A.n_frames + f) * 12 + k]) { if (r < 10) r++; }
            else if (r > 0) r--;
        }
    }
    A.state[b].fic_ratio = r;
}

static inline int gather_rows(int nsteps) { int y = (nsteps + 63) / 64; return y < 1 ? 1 : (y > 32 ? 32 : y); }

void launch_viterbi(const VitArgs& a, hipStream_t s)
{
    static int n_simd = 0;
    if (!n_simd) {
        int dev = 0; hipDeviceProp_t p;
        n_simd = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? 4 * p.multiProcessorCount : 1024;
    }
    // equal work per work-group: `per` groups each, chosen so that the grid never exceeds the 8 wave slots per SIMD (the
    // decoder needs ~6 resident waves per SIMD to keep the VALU busy: 3 per SIMD measured 30 % slower)
    const int n = a.c.g_end - a.c.g_begin;
    if (n <= 0) return;
    const int per = (n + 8 * n_simd - 1) / (8 * n_simd);
    const int grid = (n + per - 1) / per;
    hipLaunchKernelGGL(k_viterbi, dim3(grid), dim3(64), 0, s, a);
}
void launch_fic_gather(const FicGatherArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(k_fic_gather, dim3(a.c.n_groups, gather_rows(a.c.nsteps)), dim3(256), 0, s, a);
}
void launch_msc_gather(const MscGatherArgs& a, hipStream_t s)
{
    if (a.c.g_end > a.c.g_begin) hipLaunchKernelGGL(k_msc_gather, dim3(a.c.g_end - a.c.g_begin), dim3(256), 0, s, a);
}
void launch_lin_gather(const LinGatherArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(k_lin_gather, dim3(a.c.n_groups, gather_rows(a.c.nsteps)), dim3(256), 0, s, a);
}
void launch_fib_crc(const CrcArgs& a, hipStream_t s)
{
    const int n = a.n_ens * (a.frame_sel ? 1 : a.n_frames) * 12;
    hipLaunchKernelGGL(k_fib_crc, dim3((n + 255) / 256), dim3(256), 0, s, a);
}
void launch_fic_ratio(const CrcArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(k_fic_ratio, dim3((a.n_ens + 63) / 64), dim3(64), 0, s, a);
}

} // namespace dabphy
