// welle.io_amd/csrc/k_viterbi_sp.hip -- K = 7 Viterbi decoder, STATE-PARALLEL: one wavefront per code word, lanes = the 64 trellis states.
//
// Replaces (reference file:line, relative to src/backend) exactly what k_viterbi_fused replaces -- Viterbi::deconvolve / BFLY /
// chainback_viterbi (viterbi.cpp:227-339), the depuncturing of EEPProtection / UEPProtection::deconvolve (eep-protection.cpp:115-152,
// uep-protection.cpp:169-239) and FicHandler::processFicInput (fic-handler.cpp:144-204), DabAudio's time de-interleaver
// (dab-audio.cpp:113-149), energy dispersal and bit packing -- for the batches k_viterbi_fused is the wrong shape for.
//
// Why a second decoder.  k_viterbi_fused gives every LANE a code word: three VALU instructions per code word and trellis step, nothing
// cheaper exists on this machine -- but a wavefront then needs 64 code words and ~1 000 cycles per step, so one transmission frame of one
// ensemble (72 MSC + 4 FIC code words) is two waves that walk 1542 dependent steps for half a millisecond on an otherwise idle device:
// the latency regime (the live receiver behind GpuRadioReceiver, BASELINE configs 2-3).  Here the 64 path metrics of ONE code word are
// the 64 lanes of a wavefront: ~12 VALU instructions per step, four times the work per code word -- and a hundredth of the latency,
// because 76 code words are 76 wavefronts on 76 SIMDs.  dabphy_fused.hip picks this kernel while the batch has fewer code words than the
// device has lanes to give them (profiles/r04_viterbi_state_parallel.txt has the crossover).
//
// The trellis in place.  Lane l holds the metric of state rotl6(l, f) at a step in layout f = t % 6, so the two inputs k and k + 32 of a
// butterfly are the lanes of a pair that differs in lane bit 5 - f, and its outputs 2k and 2k + 1 stay in those two lanes (the new
// layout is f + 1).  Both lanes of a pair fetch X = M[k] and Y = M[k + 32] (pair_values: v_permlane32_swap, v_permlane16_swap, masked
// row DPP moves, quad_perm DPP reads, as the bit asks) and each computes its own output min(X + b, Y - b), where b = +-(bm(p_k) - 510)
// -- the sign flips for the lane that computes 2k + 1, and bm(p) + bm(~p) = 1020 (viterbi.cpp:170-177) lets every step drop the common
// 510.  Branch metrics: the soft bits of a step are the same for all lanes (scalar registers), b is three multiply-adds with per-lane
// constants +-1.  Metrics are int32, doubled (the +-1/2 of the mapping viterbi.cpp:233-238 become integers), never renormalised:
// |b| <= 1020 per step, 9222 steps.  Decisions: the sign of (Y - b) - (X + b) is "m0 > m1" with the reference's tie-break for BOTH lanes of
// a pair (both compare the X path against the Y path); one v_alignbit_b32 shifts it into the lane's own history, and 30 steps (five
// turns through the layouts: straight-line code) leave as one coalesced 256-byte store: 8.5 bytes per step and code word, no cross-lane
// packing at all.
//
// Traceback on the scalar unit: the walk from state 0 carries the LANE index of the current state; one step back replaces one bit of it
// -- position (5 - f) mod 6 -- by the decision it has just read (the same algebra as viterbi_acs.h, with the identity as decision
// index).  Histories come back 30 steps per load, one word per lane; a step reads the word of lane l with v_readlane.
//
// Gather.  The soft bits of a code word are fetched once, up front, by all 64 lanes (lane j: steps j, j + 64, ...) straight from the
// soft-bit ring -- time de-interleaver as an address computation, depuncturing by the class's map -- and parked in LDS as one packed
// word per step (the three branch-metric inputs, doubled and biased: x0 = v0 + v3 because outputs 0 and 3 share a generator).
//
// Who launches it (dabphy_fused.hip).  A batch of at most 16 384 code words: every MSC class and the FIC of the batch, kinds 0 and 1 of
// FusedClass, through the same work list as k_viterbi_fused.  And stand-alone classes (sp_single_*): the one-frame FIC of exact batch
// mode's replay (kind 1 with FusedArgs::fic_frame_sel) and small calls of the level-2 seams -- dabphy_viterbi_batch,
// dabphy_msc_deconvolve (kind 2: code words one after the other in a plain array, arbitrary int8 with the clamp of viterbi.cpp:233-236)
// and dabphy_fic_decode (kind 1 with its own frame stride).
#include "dabphy_kernels.h"
#include <dabphy_wave_ops.h>
#include "viterbi_acs.h"

namespace dabphy {

constexpr int SP_HIST = 30;                       // trellis steps per decision history word (k_viterbi_sp)
namespace sp {
__device__ __forceinline__ int rotl6(int x, int r) { r %= 6; return r == 0 ? x : (((x << r) | (x >> (6 - r))) & 63); }
__device__ __forceinline__ int brev4(int i) { return ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3); }   // = map16[i] of dab-audio.cpp:113
}

template <int MAXSTEPS, int OCC>
__global__ void __launch_bounds__(64, OCC) k_viterbi_sp(FusedArgs A)
{
    __shared__ uint32_t sym[MAXSTEPS + 2 * SP_HIST];
    __shared__ long long s_rowoff[16];
    const int lane = threadIdx.x;
    const int F = A.n_frames, R = 4 * F;
    const uint32_t wk = as_constant(A.work)[blockIdx.x >> 6];
    const DABPHY_CONST_AS FusedClass& C = as_constant(A.cls)[wk >> 24];
    const int cw = (int)(wk & 0xffffffu) * 64 + (int)(blockIdx.x & 63u);
    const int nsteps = C.nsteps, nbits = C.nbits;
    if (cw >= C.n_cw || nsteps > MAXSTEPS) return;

    // ---- where this code word's soft bits lie: 16 row offsets (one per column u & 15 of the time de-interleaver), -1 = no such CIF
    const int8_t* base;
    if (C.kind == 0) {
        const int pair = cw / R, r = cw - pair * R;
        const MscPair pp = C.pairs[pair];                                   // every ensemble selects its own sub-channels (msc-handler.cpp:61-103)
        const int b = pp.ens;
        base = A.soft + (size_t)b * A.ens_stride + (size_t)pp.start_bit;
        if (lane < 16) {
            const long long c_src = 4 * A.desc[(size_t)b * F].frame_no + r - 16 + sp::brev4(lane);      // dab-audio.cpp:113,138-143
            s_rowoff[lane] = c_src >= 0 ? ((long long)((c_src >> 2) % A.soft_ring) * 75 + 3 + 18 * (int)(c_src & 3)) * SOFT_PER_SYM : -1;
        }
    } else if (C.kind == 1) {
        const int fsel = A.fic_frame_sel;
        const int bf = fsel ? (cw >> 2) * F + (fsel - 1) : cw >> 2, b = bf / F;
        const FrameDesc& d = A.desc[bf];
        const size_t fstride = A.fic_frame_stride ? A.fic_frame_stride : (size_t)SOFT_PER_FRAME;
        base = A.soft + (size_t)b * A.ens_stride + (size_t)(d.frame_no % A.soft_ring) * fstride + (size_t)2304 * (cw & 3);
        if (lane < 16) s_rowoff[lane] = d.valid == 1 ? 0 : -1;
    } else {
        base = A.lin_in + (size_t)cw * A.lin_stride;                     // a code word of the linear seams: no de-interleaver
        if (lane < 16) s_rowoff[lane] = 0;
    }
    __syncthreads();
    {
        const int16_t* __restrict__ map = C.map;
        for (int s = lane; s < nsteps; s += 64) {
            uint2 mm = make_uint2(0, 0);
            if (map) mm = *reinterpret_cast<const uint2*>(map + 4 * s);                    // four map entries
            int v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int u = map ? (int)(int16_t)(((j < 2 ? mm.x : mm.y) >> (16 * (j & 1))) & 0xffffu) : 4 * s + j;
                long long off = -1;
                if (u >= 0) off = s_rowoff[u & 15];
                v[j] = off >= 0 ? (int)base[off + u] : 0;
                if (v[j] < -127) v[j] = -127;                               // -128 maps to symbol 0 like -127 (viterbi.cpp:233-236): the demapper never produces it, a seam's caller may
            }
            // the three branch-metric inputs of the step, doubled and biased as the trellis takes them (viterbi.cpp:233-238 puts the symbol
            // levels at v + 127: bm(p) = 510 + e0 (x0 - 1) + e1 (v1 - 1/2) + e2 (v2 - 1/2), x0 = v0 + v3): 12 + 10 + 10 signed bits
            sym[s] = ((uint32_t)(2 * (v[0] + v[3]) - 2) & 0xfffu) | (((uint32_t)(2 * v[1] - 1) & 0x3ffu) << 12) | ((uint32_t)(2 * v[2] - 1) << 22);
        }
    }
    __syncthreads();

    // ---- per-lane constants: for each layout f the signs of the three branch-metric terms of this lane's butterfly output
    int E[6][3];
#pragma unroll
    for (int f = 0; f < 6; f++) {
        const int st = sp::rotl6(lane, f), p = acs::pat(st & 31), sg = (st >> 5) ? -1 : 1;
#pragma unroll
        for (int j = 0; j < 3; j++) E[f][j] = ((p >> j) & 1) ? -sg : sg;
    }
    int Mx = lane == 0 ? 0 : 126;                                        // init_viterbi (viterbi.cpp:342-354): all 63, start state 0 at 0; doubled
    // decisions: lane l keeps the decisions of ITS new states, newest in bit 0; SP_HIST steps leave as one coalesced 256-byte store.
    // (30, not 32: a multiple of the six layouts, so a block of steps is straight-line code with one store at its end.)
    uint32_t* __restrict__ const dec_g = reinterpret_cast<uint32_t*>(A.dec + (size_t)blockIdx.x * A.dec_slot_cells);
    uint32_t acc = 0;
    // the packed soft bits of a block, one step per lane (lane k: step t0 + k), fetched from LDS a block ahead; a step takes its
    // word with v_readlane (constant lane): no LDS round trip inside the dependent chain of the trellis
    auto one_step = [&](auto fc, uint32_t w) {
        constexpr int FL = decltype(fc)::value;
        const int a0 = (int)(w << 20) >> 20, a1 = (int)(w << 10) >> 22, a2 = (int)w >> 22;
        const int beta = mad_i24(E[FL][0], a0, mad_i24(E[FL][1], a1, mul_i24(E[FL][2], a2)));
        uint32_t X, Y;
        pair_values<5 - FL>((uint32_t)Mx, X, Y);
        const int cX = (int)X + beta, cY = (int)Y - beta;
        acc = funnel_shr(acc, (uint32_t)(cY - cX), 31);                 // acc << 1 | (m0 > m1): ties keep the m0 branch (viterbi.cpp:263-268)
        Mx = cX < cY ? cX : cY;
    };
    auto six_steps = [&](uint32_t wv, auto kc) {
        constexpr int K = decltype(kc)::value;
        one_step(std::integral_constant<int, 0>{}, lane_get(wv, K + 0)); one_step(std::integral_constant<int, 1>{}, lane_get(wv, K + 1));
        one_step(std::integral_constant<int, 2>{}, lane_get(wv, K + 2)); one_step(std::integral_constant<int, 3>{}, lane_get(wv, K + 3));
        one_step(std::integral_constant<int, 4>{}, lane_get(wv, K + 4)); one_step(std::integral_constant<int, 5>{}, lane_get(wv, K + 5));
    };
    const int nfull = nsteps / SP_HIST;
    uint32_t wv = sym[lane < SP_HIST ? lane : 0];                        // (sym has SP_HIST words of slack behind nsteps)
    for (int blk = 0; blk < nfull; blk++) {
        const uint32_t wn = sym[(blk + 1) * SP_HIST + (lane < SP_HIST ? lane : 0)];
        six_steps(wv, std::integral_constant<int, 0>{}); six_steps(wv, std::integral_constant<int, 6>{}); six_steps(wv, std::integral_constant<int, 12>{});
        six_steps(wv, std::integral_constant<int, 18>{}); six_steps(wv, std::integral_constant<int, 24>{});
        dec_g[blk * 64 + lane] = acc;
        wv = wn;
    }
    {   // the last, partial block (nsteps is a multiple of six): its first step in bit SP_HIST - 1 like the others
        const int rem = nsteps - nfull * SP_HIST;
        if (rem >= 6) six_steps(wv, std::integral_constant<int, 0>{});
        if (rem >= 12) six_steps(wv, std::integral_constant<int, 6>{});
        if (rem >= 18) six_steps(wv, std::integral_constant<int, 12>{});
        if (rem >= 24) six_steps(wv, std::integral_constant<int, 18>{});
        if (rem) dec_g[nfull * 64 + lane] = acc << (SP_HIST - rem);
    }
    __syncthreads();                                                    // (one wave: the wait it implies orders the stores above before the loads below)

    // ---- traceback from state 0 (chainback_viterbi, viterbi.cpp:313-339): l = lane that computed the current state when its step ran.
    // All of it is wave-uniform (scalar unit + one v_readlane per step).  The decoded bits -- the decisions themselves -- are shifted
    // into the top of a 64-bit register, newest first; whenever 32 of them have gathered the oldest 32 leave as one output word
    // (bytes packed MSB first, decoder_adapter.cpp:61-67; the first bit read is data bit nbits - 1, and nbits is a multiple of 32).
    uint32_t* __restrict__ const out = reinterpret_cast<uint32_t*>(C.out) + (size_t)cw * (nbits / 32);
    const uint32_t* __restrict__ prbs = A.prbs_words;
    const int dedisperse = C.dedisperse;
    uint32_t l = 0;
    unsigned long long bits = 0; int cnt = 0, wi = nbits / 32;
    auto take = [&](uint32_t d, uint32_t rho) {
        bits = (bits >> 1) | ((unsigned long long)d << 63);
        l = (l & ~(1u << rho)) | (d << rho);
    };
    auto emit = [&]() {
        if (cnt >= 32) {
            const uint32_t word = acs::back_word((uint32_t)(bits >> (64 - cnt)));
            wi--; cnt -= 32;
            if (lane == 0) out[wi] = dedisperse ? word ^ prbs[wi] : word;
        }
    };
    const int rem = nsteps - nfull * SP_HIST;                           // steps in the top, partial block (a multiple of six)
    if (rem) {
        const uint32_t cur = dec_g[nfull * 64 + lane];
        uint32_t rho = 0;                                               // step nsteps - 1 ran in layout 5: bit (5 - 5) is replaced first
        for (int j = rem - 1; j >= 0; j--) {
            take((lane_get(cur, l) >> (SP_HIST - 1 - j)) & 1u, rho);
            rho = rho == 5 ? 0 : rho + 1;
        }
        cnt += rem; emit();
    }
    uint32_t cur = nfull ? dec_g[(nfull - 1) * 64 + lane] : 0u;
    for (int blk = nfull - 1; blk >= 1; blk--) {                        // whole blocks above the first: 30 steps of straight-line code
        const uint32_t nxt = dec_g[(blk - 1) * 64 + lane];              // the block below, in flight while this one is walked
#pragma unroll
        for (int j = SP_HIST - 1; j >= 0; j--) take((lane_get(cur, l) >> (SP_HIST - 1 - j)) & 1u, (uint32_t)((5 - j % 6) % 6));
        cnt += SP_HIST; emit();
        cur = nxt;
    }
    if (nfull) {                                                        // block 0: its first six steps decide nothing that is kept
#pragma unroll
        for (int j = SP_HIST - 1; j >= 6; j--) take((lane_get(cur, l) >> (SP_HIST - 1 - j)) & 1u, (uint32_t)((5 - j % 6) % 6));
        cnt += SP_HIST - 6; emit();
    }
}

// pair_values against plain shuffles, all six lane bits: out[0] = mismatching lanes, out[1] = lanes checked (device self-test of the
// instruction forms the execution model of tests/hipemu stands in for)
__global__ void __launch_bounds__(64) k_selftest_pair_exchange(unsigned* out)
{
    const int lane = threadIdx.x;
    unsigned bad = 0, n = 0;
    for (unsigned round = 0; round < 16; round++) {
        const uint32_t m = (uint32_t)lane * 2654435761u + round * 40503u + (blockIdx.x << 20);
        auto chk = [&](auto bc) {
            constexpr int B = decltype(bc)::value;
            uint32_t x, y; pair_values<B>(m, x, y);
            const uint32_t wx = (uint32_t)__shfl((int)m, lane & ~(1 << B)), wy = (uint32_t)__shfl((int)m, lane | (1 << B));
            bad += (x != wx) + (y != wy); n += 2;
        };
        chk(std::integral_constant<int, 0>{}); chk(std::integral_constant<int, 1>{}); chk(std::integral_constant<int, 2>{});
        chk(std::integral_constant<int, 3>{}); chk(std::integral_constant<int, 4>{}); chk(std::integral_constant<int, 5>{});
        const uint32_t own = lane_get(m, (round * 5u + 7u) & 63u);
        bad += (own != (uint32_t)__shfl((int)m, (int)((round * 5u + 7u) & 63u))); n += 1;
    }
    if (bad) atomicAdd(&out[0], bad);
    atomicAdd(&out[1], n);
}
void launch_selftest_pair_exchange(unsigned* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_selftest_pair_exchange, dim3(8), dim3(64), 0, s, out);
}

void launch_viterbi_sp(const FusedArgs& a, int lds_variant, hipStream_t s)
{
    if (a.n_work == 0) return;
    const dim3 grid(a.n_work * 64u);
    // LDS: one packed word per trellis step of the longest code word: 6 / 12 / 36 KiB -- 8 / 8 / 4 work-groups per compute unit
    if (lds_variant == 0) hipLaunchKernelGGL((k_viterbi_sp<SP_MAXSTEPS[0], 2>), grid, dim3(64), 0, s, a);
    else if (lds_variant == 1) hipLaunchKernelGGL((k_viterbi_sp<SP_MAXSTEPS[1], 2>), grid, dim3(64), 0, s, a);
    else hipLaunchKernelGGL((k_viterbi_sp<SP_MAXSTEPS[2], 1>), grid, dim3(64), 0, s, a);
}

} // namespace dabphy
