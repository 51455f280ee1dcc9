// welle.io_amd/csrc/k_viterbi_sp2.hip -- K = 7 Viterbi decoder, state-parallel, TWO code words per wavefront.
//
// Replaces (reference file:line, relative to src/backend) what k_viterbi_sp replaces -- Viterbi::deconvolve / BFLY / chainback_viterbi
// (viterbi.cpp:227-339), the depuncturing of EEPProtection / UEPProtection::deconvolve (eep-protection.cpp:115-152,
// uep-protection.cpp:169-239) and FicHandler::processFicInput (fic-handler.cpp:144-204), DabAudio's time de-interleaver
// (dab-audio.cpp:113-149), energy dispersal and bit packing -- for the batches the lane-per-code-word kernel is the wrong shape for,
// at half the instructions per code word of round 4's k_viterbi_sp (which stays for the smallest batches and as
// dabphy_config.decode_shape = 3).  Code words of any length: the LDS table of branch-metric sums holds 480 trellis steps at a time.
//
// Layout.  A code word owns HALF a wavefront: 32 lanes, two path metrics per lane.  In round 4's kernel a lane held one state, both
// lanes of a pair fetched both inputs of their butterfly and each computed one output: twelve instructions per step, half of them the
// exchange and the branch metric.  Here a lane holds BOTH inputs k and k + 32 of butterfly k and computes BOTH outputs 2k and 2k + 1 --
// the butterfly is local -- and what crosses lanes is one register per step:
//   step t runs in layout f = t % 5.  Lane j (5 bits) of a half works butterfly k = rotl5(j, f); its registers (P, Q) hold the metrics
//   of states k and k + 32, in this order or swapped (s_in, below).  The outputs 2k + y have their new top bit = old state bit 4, which
//   is lane bit p = 4 - f: the next butterfly pairs the lane with its partner along that bit, same y.  So each lane keeps the output
//   its own bit selects and takes the partner's other one:
//     f = 0 (lane bit 4):   (P, Q)' = swap16(out0, out1): ONE v_permlane16_swap_b32 delivers both registers in (k, k + 32) order;
//     f = 1 .. 4 (bits 3 .. 0):  a lane whose bit is set writes (out1, out0) instead of (out0, out1) -- a sign folded into its
//                           constants, no instruction -- then (P, Q)' = (reg0, partner's reg1): one DPP move (row_ror:8 / quad_perm;
//                           two bank-masked row moves for bit 2).  Lanes with the bit set now hold (k + 32, k): s_in = that bit, again
//                           only a sign and, in the traceback, one XOR on the scalar unit.
//   After five steps the layout is back where it started.
// Branch metrics.  b = bm(pattern) - 510 of butterfly k is +-a0 +-a1 +-a2 in the three soft inputs of the step (x0 = v0 + v3, v1, v2:
// viterbi_acs.h): four values up to sign.  The gather leaves all four per step in LDS (int16 x 4); a lane reads THE ONE its pattern
// selects with one ds_read_i16 at a per-layout address -- no multiply-add chain -- and its sign rides in the per-lane multiplier of the
// four v_mad_i32_i24 that form P + b, Q - b, P - b, Q + b.  Per step and wavefront: 4 mad, 2 sub + 2 v_alignbit (decisions into two
// history words), 2 min, 1-2 exchange = 11-12 vector instructions for TWO code words (round 4: 12 for one).
// Metrics are int32, doubled, never renormalised (|b| <= 1020 per step, 9222 steps); decisions compare true integers, ties keep the
// k branch (viterbi.cpp:263-268) -- for a swapped lane "Q < P" is that comparison with the roles exchanged, and the traceback undoes it.
//
// Traceback: a second launch, k_traceback_sp2 (below) -- lane = code word, wave = a stretch of the code word.  The forward kernel leaves
// its decisions as history rows: per block of 30 steps and code word 64 words, register 0 of lanes 0 .. 31 and register 1 in words 32 .. 63.
// A walk carries PHYSICAL coordinates c = (register r, lane j of the half) = the word of the row that holds its decision.  One step back
// at step t (layout f): dec = history bit of c; the decoded bit is dec ^ s_in_f(j); the survivor came in through input register q = dec,
// which the exchange after step t - 1 (layout f - 1, lane bit p') filled from: swap16 -- register = bit p' of j, lane = j with bit p' := q;
// kept / given -- register q of lane j (q = 0) or of its partner (q = 1).  State 0 ends in lane 0, register 0.
// (Round 5's first versions walked back inside this kernel, on the scalar unit -- one per compute unit, shared by its four SIMDs: ~10
// scalar instructions per code word and step bounded the kernel at scale, and one wave's walk bounded it for small batches:
// profiles/r05_viterbi_sp2.txt has the numbers; the separate pass is faster at every size.)
#include "dabphy_kernels.h"
#include <dabphy_wave_ops.h>
#include "viterbi_acs.h"

namespace dabphy {

constexpr int SP2_HIST = 30;                      // trellis steps per decision history word: a multiple of the five layouts and of the six-step code word granule
namespace sp2 {
__device__ __forceinline__ int rotl5(int x, int r) { r %= 5; return r == 0 ? x : (((x << r) | (x >> (5 - r))) & 31); }
__device__ __forceinline__ int brev4(int i) { return ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3); }   // = map16[i] of dab-audio.cpp:113
__host__ __device__ constexpr int xbit(int f) { return 4 - f; }                  // lane bit of the exchange after a step in layout f
__host__ __device__ constexpr int inbit(int f) { return f == 0 ? 0 : f == 1 ? -1 : 5 - f; }   // lane bit that tells whether a lane's inputs are swapped in layout f (-1: never)
}

// CHUNK = trellis steps the LDS table holds (a multiple of 30): the code word passes through it chunk by chunk.  The table is what bounds
// the waves per SIMD (8 bytes per step and code word): 480 steps = 7.7 KB per work-group, five waves per SIMD -- with the whole 1542-step
// code word resident (25 KB: 1.5 waves per SIMD) the kernel ran at the latency of its dependent chain, twice as slow as round 4's.
template <int CHUNK, int OCC>
__global__ void __launch_bounds__(64, OCC) k_viterbi_sp2(FusedArgs A)
{
    static_assert(CHUNK % SP2_HIST == 0, "whole history blocks per chunk");
    __shared__ __attribute__((aligned(8))) int16_t tab[2][CHUNK * 4 + 4];                   // per code word and step: +a0+a1+a2, -a0+a1+a2, +a0-a1+a2, -a0-a1+a2 (+ 8 bytes: the halves' reads fall on different banks)
    __shared__ long long s_rowoff[2][16];
    const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
    const int F = A.n_frames, R = 4 * F;
    const uint32_t wk = as_constant(A.work)[blockIdx.x >> 5];
    const DABPHY_CONST_AS FusedClass& C = as_constant(A.cls)[wk >> 24];
    const int cw_a = (int)(wk & 0xffffffu) * 64 + 2 * (int)(blockIdx.x & 31u);
    const int nsteps = C.nsteps;
    if (cw_a >= C.n_cw) return;
    const bool second = cw_a + 1 < C.n_cw;                                  // (an odd class: the last wave's upper half decodes the same code word again, its output is dropped)
    const int cw = cw_a + (second ? half : 0);

    // ---- where this half's code word lies: 16 row offsets (one per column u & 15 of the time de-interleaver), -1 = no such CIF
    const int8_t* base;
    if (C.kind == 0) {
        const int pair = cw / R, r = cw - pair * R;
        const MscPair pp = C.pairs[pair];                                   // every ensemble selects its own sub-channels (msc-handler.cpp:61-103)
        const int b = pp.ens;
        base = A.soft + (size_t)b * A.ens_stride + (size_t)pp.start_bit;
        if (j < 16) {
            const long long c_src = 4 * A.desc[(size_t)b * F].frame_no + r - 16 + sp2::brev4(j);      // dab-audio.cpp:113,138-143
            s_rowoff[half][j] = c_src >= 0 ? ((long long)((c_src >> 2) % A.soft_ring) * 75 + 3 + 18 * (int)(c_src & 3)) * SOFT_PER_SYM : -1;
        }
    } else if (C.kind == 1) {
        const int fsel = A.fic_frame_sel;
        const int bf = fsel ? (cw >> 2) * F + (fsel - 1) : cw >> 2, b = bf / F;
        const FrameDesc& d = A.desc[bf];
        const size_t fstride = A.fic_frame_stride ? A.fic_frame_stride : (size_t)SOFT_PER_FRAME;
        base = A.soft + (size_t)b * A.ens_stride + (size_t)(d.frame_no % A.soft_ring) * fstride + (size_t)2304 * (cw & 3);
        if (j < 16) s_rowoff[half][j] = d.valid == 1 ? 0 : -1;
    } else {
        base = A.lin_in + (size_t)cw * A.lin_stride;                        // a code word of the linear seams: no de-interleaver
        if (j < 16) s_rowoff[half][j] = 0;
    }
    __syncthreads();
    // the table of steps [c0, c0 + n): entry (s - c0)
    auto fill = [&](int c0, int n) {
        const int16_t* __restrict__ map = C.map;
        for (int s = c0 + j; s < c0 + n; s += 32) {
            uint2 mm = make_uint2(0, 0);
            if (map) mm = *reinterpret_cast<const uint2*>(map + 4 * s);                    // four map entries
            int v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int u = map ? (int)(int16_t)(((q < 2 ? mm.x : mm.y) >> (16 * (q & 1))) & 0xffffu) : 4 * s + q;
                long long off = -1;
                if (u >= 0) off = s_rowoff[half][u & 15];
                v[q] = off >= 0 ? (int)base[off + u] : 0;
                if (v[q] < -127) v[q] = -127;                               // -128 maps to symbol 0 like -127 (viterbi.cpp:233-236)
            }
            // the three branch-metric inputs, doubled and biased as the trellis takes them (viterbi.cpp:233-238 puts the symbol levels at
            // v + 127: bm(p) = 510 + e0 (x0 - 1) + e1 (v1 - 1/2) + e2 (v2 - 1/2), x0 = v0 + v3), and their four sign combinations
            const int a0 = 2 * (v[0] + v[3]) - 2, a1 = 2 * v[1] - 1, a2 = 2 * v[2] - 1;
            const int t0 = a0 + a1 + a2, t1 = -a0 + a1 + a2, t2 = a0 - a1 + a2, t3 = -a0 - a1 + a2;
            int16_t* e = &tab[half][4 * (s - c0)];
            e[0] = (int16_t)t0; e[1] = (int16_t)t1; e[2] = (int16_t)t2; e[3] = (int16_t)t3;
        }
    };

    // ---- per-lane constants of the five layouts: which of the four sums the lane's butterfly takes (an LDS address), and the sign
    // it enters P + b with: pattern sign x (the lane writes its outputs swapped) x (the lane's inputs are swapped)
    const int16_t* tp[5]; int M[5];
#pragma unroll
    for (int f = 0; f < 5; f++) {
        const int k = sp2::rotl5(j, f), p = acs::pat(k);
        const int idx = p < 4 ? p : 7 - p;
        int sg = p < 4 ? 1 : -1;
        if (f != 0 && ((j >> sp2::xbit(f)) & 1)) sg = -sg;                  // o_out
        if (sp2::inbit(f) >= 0 && ((j >> sp2::inbit(f)) & 1)) sg = -sg;     // s_in
        M[f] = sg; tp[f] = &tab[half][idx];
    }
    int P = j == 0 ? 0 : 126, Q = 126;                                      // init_viterbi (viterbi.cpp:342-354): all 63, start state 0 at 0; doubled.  (Lane 0's inputs are never swapped.)
    uint32_t* __restrict__ const dec_g = reinterpret_cast<uint32_t*>(A.dec + (size_t)blockIdx.x * A.dec_slot_cells);
    uint32_t acc0 = 0, acc1 = 0;
    auto one_step = [&](auto fc, int T) {
        constexpr int FL = decltype(fc)::value;
        const int x0 = mad_i24_vv(M[FL], T, P), y0 = mad_i24_vv(-M[FL], T, Q);     // reg0: P + b against Q - b
        const int x1 = mad_i24_vv(-M[FL], T, P), y1 = mad_i24_vv(M[FL], T, Q);     // reg1: P - b against Q + b
        // decision = "the Q side wins": strictly smaller where Q is state k + 32 -- ties keep the k branch, viterbi.cpp:263-268 --, smaller
        // OR EQUAL where the lane's inputs are swapped and Q is state k.  All metrics are even (doubled), so "y <= x" is "y - x - 1 < 0": the
        // lane's swap bit enters the subtraction as its borrow, and the sign goes into the history word
        if constexpr (sp2::inbit(FL) >= 0) {
            constexpr unsigned long long SW = sp2::inbit(FL) == 0 ? 0xAAAAAAAAAAAAAAAAull : sp2::inbit(FL) == 1 ? 0xCCCCCCCCCCCCCCCCull : sp2::inbit(FL) == 2 ? 0xF0F0F0F0F0F0F0F0ull : 0xFF00FF00FF00FF00ull;
            acc0 = funnel_shr(acc0, sub_borrow((uint32_t)y0, (uint32_t)x0, SW), 31);
            acc1 = funnel_shr(acc1, sub_borrow((uint32_t)y1, (uint32_t)x1, SW), 31);
        } else {
            acc0 = funnel_shr(acc0, (uint32_t)(y0 - x0), 31);
            acc1 = funnel_shr(acc1, (uint32_t)(y1 - x1), 31);
        }
        const uint32_t r0 = (uint32_t)(x0 < y0 ? x0 : y0), r1 = (uint32_t)(x1 < y1 ? x1 : y1);
        if constexpr (FL == 0) { uint32_t a, b; swap16(r0, r1, a, b); P = (int)a; Q = (int)b; }
        else { P = (int)r0; Q = (int)partner<sp2::xbit(FL)>(r1); }
    };
    // six steps from step index K of a block (K a multiple of six; the block starts at a multiple of 30, so the layout of step K + i is (K + i) % 5)
    auto six_steps = [&](int e0, auto kc) {                               // e0 = table entry of the block's first step
        constexpr int K = decltype(kc)::value;
        int T[6];
#pragma unroll
        for (int i = 0; i < 6; i++) T[i] = tp[(K + i) % 5][4 * (e0 + K + i)];
        one_step(std::integral_constant<int, (K + 0) % 5>{}, T[0]); one_step(std::integral_constant<int, (K + 1) % 5>{}, T[1]);
        one_step(std::integral_constant<int, (K + 2) % 5>{}, T[2]); one_step(std::integral_constant<int, (K + 3) % 5>{}, T[3]);
        one_step(std::integral_constant<int, (K + 4) % 5>{}, T[4]); one_step(std::integral_constant<int, (K + 5) % 5>{}, T[5]);
    };
    // the history words of a block leave as TWO rows of 64: one per code word, the words of its register 0 in lanes 0 .. 31 and those of
    // its register 1 in lanes 32 .. 63 (one v_permlane32_swap_b32), so that the traceback reads coordinate (register, lane) with one v_readlane
    auto store_hist = [&](int blk, uint32_t h0, uint32_t h1) {
        uint32_t ca, cb; swap32(h0, h1, ca, cb);
        dec_g[blk * 128 + lane] = ca; dec_g[blk * 128 + 64 + lane] = cb;
    };
    const int nfull = nsteps / SP2_HIST;
    const int rem = nsteps - nfull * SP2_HIST;                               // steps in the last, partial block (a multiple of six)
    for (int c0 = 0; c0 < nsteps; c0 += CHUNK) {
        const int n = nsteps - c0 < CHUNK ? nsteps - c0 : CHUNK;
        if (c0) __syncthreads();                                            // (the previous chunk has been read)
        fill(c0, n);
        __syncthreads();
        for (int blk = c0 / SP2_HIST; blk < (c0 + n) / SP2_HIST; blk++) {
            const int e0 = blk * SP2_HIST - c0;
            six_steps(e0, std::integral_constant<int, 0>{}); six_steps(e0, std::integral_constant<int, 6>{}); six_steps(e0, std::integral_constant<int, 12>{});
            six_steps(e0, std::integral_constant<int, 18>{}); six_steps(e0, std::integral_constant<int, 24>{});
            store_hist(blk, acc0, acc1);
        }
    }
    {
        const int t0 = nfull * SP2_HIST - (nsteps - 1) / CHUNK * CHUNK;     // (the partial block lies in the last chunk)
        if (rem >= 6) six_steps(t0, std::integral_constant<int, 0>{});
        if (rem >= 12) six_steps(t0, std::integral_constant<int, 6>{});
        if (rem >= 18) six_steps(t0, std::integral_constant<int, 12>{});
        if (rem >= 24) six_steps(t0, std::integral_constant<int, 18>{});
        if (rem) store_hist(nfull, acc0 << (SP2_HIST - rem), acc1 << (SP2_HIST - rem));        // its first step in bit SP2_HIST - 1 like the others
    }
    // the traceback is a pass of its own (k_traceback_sp2, below)
}

// ---- The traceback as a pass of its own: LANE = code word, WAVE = a stretch of the code word.  At scale the in-kernel walk above is what
// bounds a state-parallel kernel -- it runs on the scalar unit, one per compute unit, ~10 instructions per code word and step --; here 64
// walks run side by side on the vector unit, one work-group per group of 64 code words (= one item of the work list = 32 work-groups of
// the forward launch).  Per block of 30 steps the 64 history rows of the group (256 bytes each, one per code word: register 0 in words
// 0 .. 31, register 1 in 32 .. 63) come into LDS with coalesced 16-byte loads -- the next block's are in flight while this one is walked --;
// a lane then reads word c of ITS row per step -- c = its walk's coordinate, exactly the lane the scalar walk hands to v_readlane -- and
// applies the same step (back() above, on vector registers).
// A walk is a chain of 1500+ dependent LDS reads, and a medium batch has too few code words to hide it (152 work-groups for 16 ensembles x 8
// frames): the pass ran at the latency of ONE walk.  So the code word is cut into TB_SEG stretches of whole blocks, one per wave of the
// work-group, walked AT THE SAME TIME (the block-parallel traceback of the north star):
//   * the last stretch starts where the trellis ends, in state 0 (viterbi.cpp:313): exact;
//   * every other stretch does not know the state its walk enters with -- the exit state of the stretch above.  It starts `warm` blocks
//     higher from an arbitrary state (coordinate 0), walks them without output -- survivor paths merge within a few constraint lengths, so
//     it almost always arrives in the right state --, then walks its own blocks and writes their output words;
//   * then the guess is CHECKED, nothing is assumed: each wave compares the state it entered its stretch with against the exit state of
//     the wave above (LDS); a wave with any lane that differs walks its stretch again from the true state, which can change ITS exit, so
//     the check repeats until no wave has walked again -- at most TB_SEG - 1 rounds, because the top stretch is exact, after one round
//     the one below it is, and so on.  The result is the serial walk's, bit for bit, whatever the data (tests force warm = 0, where every
//     guess is wrong and the rounds cascade);
//   * stretches are whole blocks, output words are 32 bits: the word that straddles a boundary is put together at the end from the upper
//     stretch's last bits (LDS) and the lower one's first.
// 8.5 bytes of HBM per code word and step (+ warm / stretch length), eleven vector instructions per step for 64 code words.
constexpr int TB_PITCH = 65;                       // words per row in LDS: consecutive rows start on consecutive banks
// TB_SEG = stretches per code word = waves per work-group.  4 x 16.6 KB of rows: two work-groups per compute unit, 512 on the device --
// a launch of more groups than that (up to 640 + the classes' remainders within the state-parallel limit) takes 3: three work-groups per
// compute unit, so that every walk of the launch is resident at once instead of a second round of work-groups waiting for the first
template <int TB_SEG>
__global__ void __launch_bounds__(64 * TB_SEG) DABPHY_WAVES_PER_SIMD(TB_SEG == 3 ? 3 : 2) k_traceback_sp2(FusedArgs A)
{
    __shared__ uint32_t s_rows[TB_SEG][64 * TB_PITCH];
    __shared__ uint32_t s_exit[TB_SEG][64], s_part[TB_SEG][64];             // per stretch and code word: exit coordinate; the bits it holds of the word that straddles its lower boundary
    __shared__ int s_again[TB_SEG];
    const int lane = threadIdx.x & 63, seg = uniform_i32((int)(threadIdx.x >> 6));
    const uint32_t wk = as_constant(A.work)[blockIdx.x];
    const DABPHY_CONST_AS FusedClass& C = as_constant(A.cls)[wk >> 24];
    const int cw = (int)(wk & 0xffffffu) * 64 + lane;
    const int nsteps = C.nsteps, nbits = C.nbits;
    const bool live = cw < C.n_cw;
    if (threadIdx.x < TB_SEG) s_again[threadIdx.x] = 0;
    // the forward launch's work-group (32 * item + lane / 2) decoded this code word; its scratch holds two rows of 64 words per block
    const uint32_t* __restrict__ const dec_w = reinterpret_cast<const uint32_t*>(A.dec) + (size_t)blockIdx.x * 32 * A.dec_slot_cells * 2;
    const size_t wg_words = A.dec_slot_cells * 2;
    uint32_t* const rows = s_rows[seg];
    // block `blk` of the group's 64 rows: 16 x 16 bytes per row, 16 loads per lane
    auto load = [&](int blk, uint4 (&nx)[16]) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int i = lane + 64 * k, r = i >> 4, q = i & 15;
            nx[k] = *reinterpret_cast<const uint4*>(dec_w + (size_t)(r >> 1) * wg_words + (size_t)blk * 128 + (r & 1) * 64 + 4 * q);
        }
    };
    auto put = [&](const uint4 (&nx)[16]) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int i = lane + 64 * k, r = i >> 4, q = i & 15;
            uint32_t* d = &rows[r * TB_PITCH + 4 * q];
            d[0] = nx[k].x; d[1] = nx[k].y; d[2] = nx[k].z; d[3] = nx[k].w;
        }
    };
    // ---- the stretches: whole blocks [lo_b, hi_b); the partial block at the end of the code word (index nfull, `rem` steps) rides with the
    // last one, which has no warm-up: it gets `warm` blocks more than the others
    const int nfull = nsteps / SP2_HIST, rem = nsteps - nfull * SP2_HIST;
    const int warm = A.sp2_warm;
    const int spread = nfull - warm;                                        // blocks shared out evenly
    const int n_seg = spread >= 2 * 6 ? (spread / 6 < TB_SEG ? spread / 6 : TB_SEG) : 1;       // (six blocks or more per stretch: a word has at most one boundary in it)
    const bool active = seg < n_seg, last = seg == n_seg - 1;
    const int lo_b = active ? spread * seg / n_seg : 0, hi_b = last ? nfull : spread * (seg + 1) / n_seg;
    const int own_top = last ? (rem ? nfull : nfull - 1) : hi_b - 1;       // the first block of the stretch that is walked
    const int hi_bit = last ? nbits : SP2_HIST * hi_b - 6;                  // step t decides data bit t - 6: the stretch's bits are [.., hi_bit)
    const int hpart = hi_bit & 31;                                          // of them in the word that straddles the upper boundary

    uint32_t c = 0, blkbits = 0, used = 0, headw = 0; unsigned long long bits = 0;
    int cnt = 0, wi = 0; bool head_pending = false;
    uint32_t* __restrict__ const out = reinterpret_cast<uint32_t*>(C.out) + (size_t)cw * (nbits / 32);
    const uint32_t* __restrict__ prbs = A.prbs_words;
    const int dedisperse = C.dedisperse;
    const uint32_t* const my = &rows[lane * TB_PITCH];
    // bytes packed MSB first (decoder_adapter.cpp:61-67), energy dispersal (a permutation of bit positions, then an XOR: partial words may be OR-ed before it)
    auto finish_word = [&](uint32_t w, int idx) { const uint32_t v = acs::back_word(bit_reverse32(w)); return dedisperse ? v ^ prbs[idx] : v; };
    auto back = [&](auto fc, int pos) {
        constexpr int FL = decltype(fc)::value;
        const uint32_t dec = (my[c] >> pos) & 1u;
        uint32_t d = dec;
        if constexpr (sp2::inbit(FL) >= 0) d ^= (c >> sp2::inbit(FL)) & 1u;
        blkbits = (blkbits << 1) | d;
        constexpr int FP = (FL + 4) % 5, pb = sp2::xbit(FP);
        if constexpr (FP == 0) c = (c & 15u) | (dec << 4) | ((c & 16u) << 1);
        else c = (c & 31u) ^ (dec * ((1u << pb) | 32u));
    };
    auto back_n = [&](auto hi, auto lo) {                                   // steps hi - 1 down to lo of the staged block, straight-line
        constexpr int HI = decltype(hi)::value, LO = decltype(lo)::value;
        auto go = [&](auto self, auto kc) -> void {
            constexpr int k = decltype(kc)::value;
            back(std::integral_constant<int, k % 5>{}, SP2_HIST - 1 - k);
            if constexpr (k > LO) self(self, std::integral_constant<int, k - 1>{});
        };
        if constexpr (HI > LO) go(go, std::integral_constant<int, HI - 1>{});
    };
    // n more bits: whenever 32 have gathered the oldest 32 leave as one output word.  A stretch that does not start on a word boundary
    // starts as if the bits above it in that word (the upper stretch's) had been gathered as zeros; the first word to leave is then the
    // straddling one: held back (headw) until the upper stretch's share is known
    auto gathered = [&](int n) {
        bits = (bits << n) | blkbits; blkbits = 0; cnt += n;
        if (cnt >= 32) {
            wi--; cnt -= 32;
            const uint32_t w = (uint32_t)(bits >> cnt);
            if (head_pending) { headw = w; head_pending = false; }
            else if (live) out[wi] = finish_word(w, wi);
        }
    };
    // blocks top .. lo_b from coordinate c0; blocks above own_top are warm-up: walked, nothing kept but the coordinate
    auto run = [&](int top, uint32_t c0) {
        c = c0; blkbits = 0; bits = 0;
        cnt = hpart ? 32 - hpart : 0; wi = (hi_bit + 31) >> 5; head_pending = hpart != 0; headw = 0;
        uint4 nx[16];
        lds_reads_done(); wave_converge();                                  // (whatever this wave read of its rows before)
        load(top, nx); put(nx);
        for (int blk = top; blk >= lo_b; blk--) {
            lds_reads_done(); wave_converge();                              // the rows are in LDS for every lane of the wave
            if (blk > lo_b) load(blk - 1, nx);                              // the block below, in flight while this one is walked
            if (blk == own_top) used = c;
            int n;
            if (blk == nfull) {                                             // the partial block (a multiple of six steps)
                if (rem == 6) back_n(std::integral_constant<int, 6>{}, std::integral_constant<int, 0>{});
                else if (rem == 12) back_n(std::integral_constant<int, 12>{}, std::integral_constant<int, 0>{});
                else if (rem == 18) back_n(std::integral_constant<int, 18>{}, std::integral_constant<int, 0>{});
                else back_n(std::integral_constant<int, 24>{}, std::integral_constant<int, 0>{});
                n = rem;
            } else if (blk == 0) {                                          // block 0: its first six steps decide nothing that is kept
                back_n(std::integral_constant<int, SP2_HIST>{}, std::integral_constant<int, 6>{});
                n = SP2_HIST - 6;
            } else {
                back_n(std::integral_constant<int, SP2_HIST>{}, std::integral_constant<int, 0>{});
                n = SP2_HIST;
            }
            if (blk <= own_top) gathered(n); else blkbits = 0;
            if (blk > lo_b) { lds_reads_done(); wave_converge(); put(nx); }
        }
        s_exit[seg][lane] = c;
        s_part[seg][lane] = cnt ? (uint32_t)bits << (32 - cnt) : 0u;        // what is left: the top bits of the word that straddles the lower boundary
    };
    if (active) {
        int top = own_top;
        if (!last) { top = hi_b + warm - 1; if (top > nfull - 1) top = nfull - 1; }
        run(top, 0u);                                                       // state 0 ends in lane 0 of its half, register 0: coordinate 0
    }
    __syncthreads();
    for (int round = 1; round < n_seg; round++) {
        uint32_t truth = 0; bool redo = false;
        if (active && !last) {
            truth = s_exit[seg + 1][lane];
            redo = __ballot(live && truth != used) != 0;
        }
        __syncthreads();                                                    // (every wave has read the exit it checks against before any is written again)
        if (redo) {
            run(own_top, truth);                                            // lanes that had guessed right walk the same way again
            if (lane == 0) s_again[round] = 1;
        }
        __syncthreads();
        if (!s_again[round]) break;
    }
    // the words that straddle the boundaries
    if (active && !last && hpart && live) { const int idx = hi_bit >> 5; out[idx] = finish_word(headw | s_part[seg + 1][lane], idx); }
}

// swap16 / swap32 / partner against plain shuffles, all five lane bits (device self-test of the instruction forms the execution model of
// tests/hipemu stands in for): out[0] += mismatching lanes, out[1] += lanes checked
__global__ void __launch_bounds__(64) k_selftest_half_exchange(unsigned* out)
{
    const int lane = threadIdx.x;
    unsigned bad = 0, n = 0;
    for (unsigned round = 0; round < 16; round++) {
        const uint32_t r0 = (uint32_t)lane * 2654435761u + round * 40503u + (blockIdx.x << 20), r1 = ~r0 * 2246822519u + round;
        {
            uint32_t a, b; swap16(r0, r1, a, b);
            const bool set = (lane >> 4) & 1;
            const uint32_t p0 = (uint32_t)__shfl((int)r0, lane ^ 16), p1 = (uint32_t)__shfl((int)r1, lane ^ 16);
            bad += (a != (set ? p1 : r0)) + (b != (set ? r1 : p0)); n += 2;
        }
        {
            uint32_t a, b; swap32(r0, r1, a, b);
            const bool up = lane >= 32;
            const uint32_t p0 = (uint32_t)__shfl((int)r0, lane ^ 32), p1 = (uint32_t)__shfl((int)r1, lane ^ 32);
            bad += (a != (up ? p1 : r0)) + (b != (up ? r1 : p0)); n += 2;
        }
        auto chk = [&](auto bc) {
            constexpr int B = decltype(bc)::value;
            bad += partner<B>(r1) != (uint32_t)__shfl((int)r1, lane ^ (1 << B)); n += 1;
        };
        chk(std::integral_constant<int, 0>{}); chk(std::integral_constant<int, 1>{}); chk(std::integral_constant<int, 2>{}); chk(std::integral_constant<int, 3>{});
    }
    if (bad) atomicAdd(&out[0], bad);
    atomicAdd(&out[1], n);
}
void launch_selftest_half_exchange(unsigned* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_selftest_half_exchange, dim3(8), dim3(64), 0, s, out);
}

void launch_viterbi_sp2(const FusedArgs& a, int lds_variant, hipStream_t s)
{
    if (a.n_work == 0) return;
    const dim3 grid(a.n_work * 32u);
    // LDS: four 16-bit sums per trellis step, for two code words, 480 steps at a time: 7.7 KiB, 20 work-groups per compute unit
    (void)lds_variant;
    hipLaunchKernelGGL((k_viterbi_sp2<480, 4>), grid, dim3(64), 0, s, a);
    if ((int)a.n_work <= a.sp2_resident) hipLaunchKernelGGL(k_traceback_sp2<4>, dim3(a.n_work), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_traceback_sp2<3>, dim3(a.n_work), dim3(192), 0, s, a);
}

} // namespace dabphy
