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
// Traceback on the scalar unit -- one per compute unit, shared by its four SIMDs: at scale it is what bounds a state-parallel kernel, so
// a step is eight scalar instructions and one v_readlane per code word --, both code words of the wave interleaved (two independent
// chains).  The walk carries PHYSICAL coordinates c = (register r, lane j of the half); the history words of a block are stored so that
// a code word's row holds register 0 in lanes 0 .. 31 and register 1 in lanes 32 .. 63: c is the lane to read.  One step back at step t
// (layout f): dec = history bit of c; the decoded bit is dec ^ s_in_f(j); the survivor came in through input register q = dec, which
// the exchange after step t - 1 (layout f - 1, lane bit p') filled from: swap16 -- register = bit p' of j, lane = j with bit p' := q;
// kept / given -- register q of lane j (q = 0) or of its partner (q = 1).  State 0 ends in lane 0, register 0.
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
    const int nsteps = C.nsteps, nbits = C.nbits;
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
    if (A.sp2_split) return;                                                // the traceback is a pass of its own (k_traceback_sp2, below)
    __syncthreads();                                                        // (one wave: the wait it implies orders the stores above before the loads below)

    // ---- traceback from state 0 (chainback_viterbi, viterbi.cpp:313-339), both code words of the wave side by side on the scalar unit.
    // The decoded bits are shifted into the top of a 64-bit register, newest first; whenever 32 of them have gathered the oldest 32
    // leave as one output word (bytes packed MSB first, decoder_adapter.cpp:61-67; the first bit read is data bit nbits - 1).
    // A walk's coordinate c = register << 5 | lane of the half: the lane of its code word's history row that holds the decision.
    struct Walk { uint32_t c, blk; unsigned long long bits; };              // blk: the bits of the block being walked, newest in bit 0
    Walk W[2] = {{0u, 0u, 0ull}, {0u, 0u, 0ull}};                           // state 0 ends in lane 0 of its half, register 0
    int cnt = 0, wi = nbits / 32;
    uint32_t* __restrict__ const out_a = reinterpret_cast<uint32_t*>(C.out) + (size_t)cw_a * (nbits / 32);
    const uint32_t* __restrict__ prbs = A.prbs_words;
    const int dedisperse = C.dedisperse;
    // one step back in layout FL at bit position `pos` of the history words (row = the code word's history row of this block, one word per lane)
    auto back = [&](auto fc, Walk& w, uint32_t row, int pos) {
        constexpr int FL = decltype(fc)::value;
        const uint32_t dec = ubfe(lane_get(row, w.c), pos, 1);
        uint32_t d = dec;
        if constexpr (sp2::inbit(FL) >= 0) d ^= ubfe(w.c, sp2::inbit(FL), 1);           // the lane's inputs were swapped: "Q side" was state k
        w.blk = (w.blk << 1) | d;                                                         // (newest in bit 0: reversed when a word leaves)
        // the survivor came in through input register q = dec; the exchange after the previous step (layout FP, lane bit pb) filled it
        constexpr int FP = (FL + 4) % 5, pb = sp2::xbit(FP);
        if constexpr (FP == 0)                           // swap16: outputs in (2k, 2k + 1) order; register = the lane's bit, the lane = this one with the bit set to q
            w.c = (w.c & 15u) | (dec << 4) | ((w.c & 16u) << 1);
        else                                             // kept / given: input 0 = the lane's own register 0, input 1 = its partner's register 1
            w.c = (w.c & 31u) ^ (dec ? ((1u << pb) | 32u) : 0u);
    };
    auto emit = [&]() {
        if (cnt >= 32) {
            wi--; cnt -= 32;
            // the 32 oldest of the bits gathered, the first one read in bit 0 (acs::back_word's order): bits [cnt, cnt + 32), reversed
            const uint32_t wa = acs::back_word(bit_reverse32((uint32_t)(W[0].bits >> cnt))), wb = acs::back_word(bit_reverse32((uint32_t)(W[1].bits >> cnt)));
            const uint32_t x = dedisperse ? prbs[wi] : 0u;
            if (lane == 0) out_a[wi] = wa ^ x;
            if (lane == 32 && second) out_a[(nbits / 32) + wi] = wb ^ x;
        }
    };
    auto back_n = [&](uint32_t ca, uint32_t cb, auto hi, auto lo) {          // steps hi - 1 down to lo of a block, straight-line, the two walks side by side
        constexpr int HI = decltype(hi)::value, LO = decltype(lo)::value;
        auto go = [&](auto self, auto kc) -> void {
            constexpr int k = decltype(kc)::value;
            back(std::integral_constant<int, k % 5>{}, W[0], ca, SP2_HIST - 1 - k);
            back(std::integral_constant<int, k % 5>{}, W[1], cb, SP2_HIST - 1 - k);
            if constexpr (k > LO) self(self, std::integral_constant<int, k - 1>{});
        };
        if constexpr (HI > LO) go(go, std::integral_constant<int, HI - 1>{});
    };
    if (rem) {
        const uint32_t ca = dec_g[nfull * 128 + lane], cb = dec_g[nfull * 128 + 64 + lane];
        if (rem == 6) back_n(ca, cb, std::integral_constant<int, 6>{}, std::integral_constant<int, 0>{});
        else if (rem == 12) back_n(ca, cb, std::integral_constant<int, 12>{}, std::integral_constant<int, 0>{});
        else if (rem == 18) back_n(ca, cb, std::integral_constant<int, 18>{}, std::integral_constant<int, 0>{});
        else back_n(ca, cb, std::integral_constant<int, 24>{}, std::integral_constant<int, 0>{});
        W[0].bits = (W[0].bits << rem) | W[0].blk; W[1].bits = (W[1].bits << rem) | W[1].blk; W[0].blk = W[1].blk = 0;
        cnt += rem; emit();
    }
    uint32_t ca = nfull ? dec_g[(nfull - 1) * 128 + lane] : 0u, cb = nfull ? dec_g[(nfull - 1) * 128 + 64 + lane] : 0u;
    for (int blk = nfull - 1; blk >= 1; blk--) {                            // whole blocks above the first
        const uint32_t na = dec_g[(blk - 1) * 128 + lane], nb = dec_g[(blk - 1) * 128 + 64 + lane];   // the block below, in flight while this one is walked
        back_n(ca, cb, std::integral_constant<int, SP2_HIST>{}, std::integral_constant<int, 0>{});
        W[0].bits = (W[0].bits << SP2_HIST) | W[0].blk; W[1].bits = (W[1].bits << SP2_HIST) | W[1].blk; W[0].blk = W[1].blk = 0;
        cnt += SP2_HIST; emit();
        ca = na; cb = nb;
    }
    if (nfull) {                                                            // block 0: its first six steps decide nothing that is kept
        back_n(ca, cb, std::integral_constant<int, SP2_HIST>{}, std::integral_constant<int, 6>{});
        W[0].bits = (W[0].bits << (SP2_HIST - 6)) | W[0].blk; W[1].bits = (W[1].bits << (SP2_HIST - 6)) | W[1].blk;
        cnt += SP2_HIST - 6; emit();
    }
}

// ---- The traceback as a pass of its own: LANE = code word.  At scale the in-kernel walk above is what bounds a state-parallel kernel --
// it runs on the scalar unit, one per compute unit, ~10 instructions per code word and step --; here 64 walks run side by side on the
// vector unit, one work-group per group of 64 code words (= one item of the work list = 32 work-groups of the forward launch).  Per block
// of 30 steps the 64 history rows of the group (256 bytes each, one per code word: register 0 in words 0 .. 31, register 1 in 32 .. 63)
// come into LDS with coalesced 16-byte loads; a lane then reads word c of ITS row per step -- c = its walk's coordinate, exactly the lane
// the scalar walk hands to v_readlane -- and applies the same step (back() above, on vector registers).  8.5 bytes of HBM per code word
// and step, eleven vector instructions per step for 64 code words.
constexpr int TB_PITCH = 65;                       // words per row in LDS: consecutive rows start on consecutive banks
__global__ void __launch_bounds__(64) k_traceback_sp2(FusedArgs A)
{
    __shared__ uint32_t rows[64 * TB_PITCH];
    const int lane = threadIdx.x;
    const uint32_t wk = as_constant(A.work)[blockIdx.x];
    const DABPHY_CONST_AS FusedClass& C = as_constant(A.cls)[wk >> 24];
    const int cw = (int)(wk & 0xffffffu) * 64 + lane;
    const int nsteps = C.nsteps, nbits = C.nbits;
    const bool live = cw < C.n_cw;
    // the forward launch's work-group (32 * item + lane / 2) decoded this code word; its scratch holds two rows of 64 words per block
    const uint32_t* __restrict__ const dec_w = reinterpret_cast<const uint32_t*>(A.dec) + (size_t)blockIdx.x * 32 * A.dec_slot_cells * 2;
    const size_t wg_words = A.dec_slot_cells * 2;
    auto stage = [&](int blk) {
        __syncthreads();                                                    // (the previous block's rows have been read)
        for (int i = lane; i < 64 * 16; i += 64) {                          // 16 x 16 bytes per row
            const int r = i >> 4, q = i & 15;
            const uint4 v = *reinterpret_cast<const uint4*>(dec_w + (size_t)(r >> 1) * wg_words + (size_t)blk * 128 + (r & 1) * 64 + 4 * q);
            uint32_t* d = &rows[r * TB_PITCH + 4 * q];
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
    };
    uint32_t c = 0, blkbits = 0; unsigned long long bits = 0;              // state 0 ends in lane 0 of its half, register 0: coordinate 0
    int cnt = 0, wi = nbits / 32;
    uint32_t* __restrict__ const out = reinterpret_cast<uint32_t*>(C.out) + (size_t)cw * (nbits / 32);
    const uint32_t* __restrict__ prbs = A.prbs_words;
    const int dedisperse = C.dedisperse;
    const uint32_t* const my = &rows[lane * TB_PITCH];
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
    auto gathered = [&](int n) {                                            // n more bits: whenever 32 have gathered the oldest 32 leave as one output word
        bits = (bits << n) | blkbits; blkbits = 0; cnt += n;
        if (cnt >= 32) {
            wi--; cnt -= 32;
            const uint32_t w = acs::back_word(bit_reverse32((uint32_t)(bits >> cnt)));
            if (live) out[wi] = dedisperse ? w ^ prbs[wi] : w;
        }
    };
    const int nfull = nsteps / SP2_HIST, rem = nsteps - nfull * SP2_HIST;
    if (rem) {
        stage(nfull);
        if (rem == 6) back_n(std::integral_constant<int, 6>{}, std::integral_constant<int, 0>{});
        else if (rem == 12) back_n(std::integral_constant<int, 12>{}, std::integral_constant<int, 0>{});
        else if (rem == 18) back_n(std::integral_constant<int, 18>{}, std::integral_constant<int, 0>{});
        else back_n(std::integral_constant<int, 24>{}, std::integral_constant<int, 0>{});
        gathered(rem);
    }
    for (int blk = nfull - 1; blk >= 1; blk--) {
        stage(blk);
        back_n(std::integral_constant<int, SP2_HIST>{}, std::integral_constant<int, 0>{});
        gathered(SP2_HIST);
    }
    if (nfull) {                                                            // block 0: its first six steps decide nothing that is kept
        stage(0);
        back_n(std::integral_constant<int, SP2_HIST>{}, std::integral_constant<int, 6>{});
        gathered(SP2_HIST - 6);
    }
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
    if (a.sp2_split) hipLaunchKernelGGL(k_traceback_sp2, dim3(a.n_work), dim3(64), 0, s, a);
}

} // namespace dabphy
