// welle.io_amd/csrc/viterbi_acs.h -- the add-compare-select engine of the K = 7 Viterbi kernels (k_viterbi, k_viterbi_msc).
//
// Replaces Viterbi::BFLY / update_viterbi_blk / chainback_viterbi (src/backend/viterbi.cpp:227-339); one LANE decodes one code word.
//
// What the VALU of gfx950 charges (tools/ubench/valu_rate2.hip, profiles/r03_ubench_valu_rate2.txt): v_add_u32 / v_sub_u32 / and /
// or / xor / v_lshrrev_b32 issue a wave64 instruction every ~2.3-2.5 cycles, EVERY packed 16-bit operation (v_pk_add/min/sub_*16),
// v_perm_b32, v_and_or_b32, v_bfi_b32, min/max of any width but 16 every ~4.2.  Round 2's butterfly was four packed operations
// (2 x v_pk_add_u16 with op_sel broadcasts, v_pk_min_u16, v_pk_sub_i16).  Here the two additions are PLAIN 32-bit additions of a
// register holding two metrics and a register holding two branch metrics (no carry can cross: metrics stay below 2^16), which
// needs the two metrics of a register to belong to two DIFFERENT butterflies, i.e. a pairing of states that the trellis maps
// onto itself:
//
//   layout f (f = 0..5):  register reg_of(a, f) = (metric of state a | metric of state a + 2^f << 16)   for every a with bit f clear.
//
//   A step in layout f <= 4 takes the registers (k, k') and (k + 32, k' + 32), k' = k + 2^f, of the butterflies k and k':
//       X = (k, k') + (bm(p), bm(p'))          Z = (k+32, k'+32) + (bm(~p), bm(~p'))        -> new (2k,   2k')   = pk_min(X, Z)
//       Y = (k, k') + (bm(~p), bm(~p'))        W = (k+32, k'+32) + (bm(p), bm(p'))          -> new (2k+1, 2k'+1) = pk_min(Y, W)
//   and 2k' = 2k + 2^(f+1): the new registers are in layout f + 1.  The branch pattern is linear in the state, p' = p ^ pat(2^f),
//   so a step needs the eight pairs (bm(p), bm(p ^ pi_f)), p = 0..7; they are linear in the soft bits and come from three
//   24-bit multiplications and a dozen additions (bm_pairs).  Layout 5 pairs k with k + 32 -- the two inputs of ONE butterfly -- and is worked the old way
//   (op_sel broadcasts); its results (2k, 2k+1) are layout 0 again.  Steps are numbered from layout 0: step t runs in layout t % 6.
//
// Decisions.  D = Z - X (v_pk_sub_i16): the sign bits are "m0 > m1" of viterbi.cpp:263-268 (ties keep the m0 / m2 branch); v_perm_b32
// expands the four signs of two D registers into bytes and v_and_or_b32 drops them onto bit i of each byte of a 64-bit word.  WHICH
// bit is free to choose, and the choice that makes the traceback cheap is  index(n) = rotl6(n, 3 - f)  for new state n of a step in
// layout f: the traceback then carries J = rotl6(state, .) instead of the state, and walking back one step REPLACES ONE BIT of J
// (position (3 - f) mod 6) by the decision it has just read -- v_lshrrev_b64, v_alignbit, v_lshlrev, v_bfi per step instead of the
// 14 instructions the state -> word/byte/bit arithmetic took.
//
// Exactness: metrics are true integers (minimum subtracted every 30 steps: spread <= 6 * 1020, growth <= 30 * 1020, so every half
// stays below 2^16 and every difference inside int16), decisions are comparisons of those integers: the reference's own
// renormalisation schedule (viterbi.cpp:104-120) changes no decision and is not mimicked.
#pragma once
#include <utility>
#include "dabphy_common.h"
#include <dabphy_wave_ops.h>

namespace dabphy {
namespace acs {

__device__ __forceinline__ uint32_t asu(u16x2 a) { uint32_t r; __builtin_memcpy(&r, &a, 4); return r; }
__device__ __forceinline__ u16x2 asv(uint32_t a) { u16x2 r; __builtin_memcpy(&r, &a, 4); return r; }
__device__ __forceinline__ u16x2 pkmin(u16x2 a, u16x2 b) { return (a < b) ? a : b; }

// Branch pattern of butterfly k (0..31): bit j = parity((2k) & poly_j) for polys {0155, 0117, 0123} (viterbi.cpp:36,170-177; the 4th
// output repeats poly 0155).  bm(p) = sum over the four outputs of (bit ? 255 - s : s).
__host__ __device__ constexpr int par8(int x) { x ^= x >> 4; x ^= x >> 2; x ^= x >> 1; return x & 1; }
__host__ __device__ constexpr int pat(int k) { return par8((2 * k) & 0155) | (par8((2 * k) & 0117) << 1) | (par8((2 * k) & 0123) << 2); }
__host__ __device__ constexpr int reg_of(int a, int f) { return ((a >> (f + 1)) << f) | (a & ((1 << f) - 1)); }
__host__ __device__ constexpr int rotl6(int x, int r) { return r % 6 == 0 ? x : (((x << (r % 6)) | (x >> (6 - r % 6))) & 63); }
__host__ __device__ constexpr int pi_of(int f) { return f < 5 ? pat(1 << f) : 7; }
__host__ __device__ constexpr int dec_rot(int f) { return (9 - f) % 6; }                 // (3 - f) mod 6
__host__ __device__ constexpr int spread(int j, int pos) { return ((j >> pos) << (pos + 1)) | (j & ((1 << pos) - 1)); }   // insert a zero bit at `pos`

// The soft bits of one trellis step as the engine takes them: x0 = v0 + v3 (outputs 0 and 3 share a generator), v1, v2 with
// v = symbol - 127 (viterbi.cpp:233-238; v in [-127, 127]).
//   bm(p) = 510 + e0 (x0 - 1) + e1 (v1 - 1/2) + e2 (v2 - 1/2),   e_j = +1 / -1 for pattern bit j clear / set.
// A pair (bm(p) | bm(p ^ pi) << 16) is therefore K + e0 T0 + e1 T1 + e2 T2 with T_j = t_j * (65537 or -65535 when pi flips output j);
// T1 + T2 and T1 - T2 are integers, all arithmetic is modulo 2^32 and the results are exact.
template <int PI>
__device__ __forceinline__ void bm_pairs(uint32_t (&BP)[8], int x0, int v1, int v2)
{
    constexpr int m0 = (PI & 1) ? -65535 : 65537, m1 = (PI & 2) ? -65535 : 65537, m2 = (PI & 4) ? -65535 : 65537;
    constexpr int K = 510 * 65537;
    const uint32_t T0 = (uint32_t)mul_i24(x0, m0), T1 = (uint32_t)mul_i24(v1, m1), T2 = (uint32_t)mul_i24(v2, m2);
    const uint32_t A0 = T0 + (uint32_t)(K - m0), A1 = (uint32_t)(K + m0) - T0;                       // K + (x0 - 1) m0, K - (x0 - 1) m0
    const uint32_t U1 = T1 - (uint32_t)((m1 + m2) / 2), V1 = T1 - (uint32_t)((m1 - m2) / 2);
    const uint32_t Gp = U1 + T2, Gm = V1 - T2;                                                        // (v1 - 1/2) m1 +- (v2 - 1/2) m2
    BP[0] = A0 + Gp; BP[1] = A1 + Gp;          // e1 = e2 = +
    BP[6] = A0 - Gp; BP[7] = A1 - Gp;          // e1 = e2 = -
    BP[4] = A0 + Gm; BP[5] = A1 + Gm;          // e1 = +, e2 = -
    BP[2] = A0 - Gm; BP[3] = A1 - Gm;          // e1 = -, e2 = +
}

// the four decisions of two difference registers -> bit (index & 7) of bytes 0..3 of word (index >> 5); NA = the new state in the
// low half of Da (Db: NA + 2^gamma, high halves: + 2^beta; see the file header)
template <int F, int NA>
__device__ __forceinline__ void place(u16x2 Da, u16x2 Db, uint32_t& accA, uint32_t& accB, uint32_t ones)
{
    constexpr int idx = rotl6(NA, dec_rot(F));
    static_assert((idx & 24) == 0, "decision layout");
    static_assert(rotl6(NA | (1 << (F % 6)), dec_rot(F)) == (idx | 8), "decision layout: partner register");
    static_assert(rotl6(NA | (1 << ((F + 1) % 6)), dec_rot(F)) == (idx | 16), "decision layout: high halves");
    const uint32_t x = pk_sign_bytes(Da, Db);                    // bytes [Da.lo, Db.lo, Da.hi, Db.hi]
    if (idx & 32) accB = and_or(x, ones << (idx & 7), accB);
    else          accA = and_or(x, ones << (idx & 7), accA);
}

// butterflies K and K + 2^F (layout F <= 4)
template <int F, int K>
__device__ __forceinline__ void dbfly(const uint32_t (&M)[32], uint32_t (&N)[32], const uint32_t (&BP)[8], u16x2& D0, u16x2& D1)
{
    constexpr int p = pat(K);
    static_assert(pat(K | (1 << F)) == (p ^ pi_of(F)), "pattern linearity");
    const uint32_t in1 = M[reg_of(K, F)], in2 = M[reg_of(K + 32, F)];
    const uint32_t X = in1 + BP[p], Z = in2 + BP[p ^ 7], Y = in1 + BP[p ^ 7], W = in2 + BP[p];
    N[reg_of(2 * K, F + 1)] = asu(pkmin(asv(X), asv(Z)));     D0 = asv(Z) - asv(X);
    N[reg_of(2 * K + 1, F + 1)] = asu(pkmin(asv(Y), asv(W))); D1 = asv(W) - asv(Y);
}

template <int F, int J>
__device__ __forceinline__ void group(const uint32_t (&M)[32], uint32_t (&N)[32], const uint32_t (&BP)[8], uint32_t& accA, uint32_t& accB, uint32_t ones)
{
    if constexpr (F == 0) {
        // partner of new state 2K is 2K + 1: the two results of one double butterfly; two of them per group for the scheduler
        constexpr int K0 = 4 * J, K1 = K0 | 2;                               // bit 0 clear (layout), bit 1: the two of this group
        u16x2 D0, D1, E0, E1;
        dbfly<0, K0>(M, N, BP, D0, D1);
        dbfly<0, K1>(M, N, BP, E0, E1);
        place<0, 2 * K0>(D0, D1, accA, accB, ones);
        place<0, 2 * K1>(E0, E1, accA, accB, ones);
    } else if constexpr (F <= 4) {
        constexpr int K0 = spread(spread(J, F - 1), F), K1 = K0 | (1 << (F - 1));      // bits F and F - 1 clear
        u16x2 D0, D1, E0, E1;
        dbfly<F, K0>(M, N, BP, D0, D1);
        dbfly<F, K1>(M, N, BP, E0, E1);
        place<F, 2 * K0>(D0, E0, accA, accB, ones);
        place<F, 2 * K0 + 1>(D1, E1, accA, accB, ones);
    } else {
        // layout 5: register k = (state k | state k + 32 << 16), the two inputs of butterfly k; BP[p] = (bm(p), bm(p ^ 7)) for p < 4
        auto bfly = [&](auto kc, u16x2& D) {
            constexpr int k = decltype(kc)::value, p = pat(k);
            const u16x2 in = asv(M[k]);
            const u16x2 bm = (p < 4) ? asv(BP[p]) : pk_swap(asv(BP[p ^ 7]));
            const u16x2 P = pk_dup_lo(in) + bm, Q = pk_dup_hi(in) + pk_swap(bm);
            N[k] = asu(pkmin(P, Q)); D = Q - P;                                  // new (2k, 2k + 1) = layout 0 register k
        };
        constexpr int K0 = 2 * J, K1 = 2 * J + 1;
        u16x2 D0, D1, E0, E1;
        bfly(std::integral_constant<int, K0>{}, D0); bfly(std::integral_constant<int, K0 + 16>{}, D1);
        bfly(std::integral_constant<int, K1>{}, E0); bfly(std::integral_constant<int, K1 + 16>{}, E1);
        place<5, 2 * K0>(D0, D1, accA, accB, ones);
        place<5, 2 * K1>(E0, E1, accA, accB, ones);
    }
}

template <int F, int... J>
__device__ __forceinline__ void groups(const uint32_t (&M)[32], uint32_t (&N)[32], const uint32_t (&BP)[8], uint32_t& accA, uint32_t& accB, uint32_t ones,
                                       std::integer_sequence<int, J...>)
{
    (group<F, J>(M, N, BP, accA, accB, ones), ...);
}

// One trellis step in layout F: M (layout F) -> M (layout F + 1), decision words returned.
template <int F>
__device__ __forceinline__ uint2 step(uint32_t (&M)[32], int x0, int v1, int v2, uint32_t ones)
{
    uint32_t BP[8];
    bm_pairs<pi_of(F)>(BP, x0, v1, v2);
    uint32_t N[32];
    uint32_t accA = 0, accB = 0;
    groups<F>(M, N, BP, accA, accB, ones, std::make_integer_sequence<int, 8>{});
#pragma unroll
    for (int j = 0; j < 32; j++) M[j] = N[j];
    return make_uint2(accA, accB);
}

// init_viterbi (viterbi.cpp:342-354): all 63, start state 0 biased to 0; layout 0
__device__ __forceinline__ void init(uint32_t (&M)[32])
{
#pragma unroll
    for (int j = 0; j < 32; j++) M[j] = 63u | (63u << 16);
    M[0] = 63u << 16;
}

// subtract the smallest metric from all (any layout)
__device__ __forceinline__ void renorm(uint32_t (&M)[32])
{
    u16x2 t[16];
#pragma unroll
    for (int j = 0; j < 16; j++) t[j] = pkmin(asv(M[j]), asv(M[j + 16]));
#pragma unroll
    for (int w = 8; w > 0; w >>= 1)
#pragma unroll
        for (int j = 0; j < w; j++) t[j] = pkmin(t[j], t[j + w]);
    const uint32_t mu = asu(t[0]);
    const uint32_t lo = mu & 0xffffu, hi = mu >> 16;
    const uint32_t mn = lo < hi ? lo : hi;
    uint32_t sub = mn | (mn << 16);
    sub = opaque_vgpr(sub);                                         // (one v_sub_u32 per register, not a multiply-add by 65537)
#pragma unroll
    for (int j = 0; j < 32; j++) M[j] -= sub;                       // (no borrow: every half >= mn)
}
constexpr int RENORM_BLOCKS = 5;                                    // every 5 blocks of 6 steps: 6120 + 30 * 1020 + 63 < 65536

// ---- traceback (chainback_viterbi, viterbi.cpp:313-339, from state 0).  J = rotl6(state at step t, dec_rot(t % 6)); one step back:
// read decision bit J of step t's words, replace bit dec_rot(t % 6) of J by it.  The decoded bit IS the decision.  `out` collects the
// bits of 32 steps, first one read in bit 0 ... : a byte swap away from the MSB-first packing of decoder_adapter.cpp:61-67.
__device__ __forceinline__ void back(uint2 d, uint32_t& J, uint32_t& out, uint32_t rho)
{
    const uint32_t w = (uint32_t)(((((unsigned long long)d.y) << 32) | d.x) >> (J & 63u));   // bit 0 = the decision
    out = funnel_shr(w, out, 1);                                                                // out = out >> 1 | w << 31
    const uint32_t m = 1u << rho;
    J = (J & ~m) | ((w << rho) & m);
}
// the 32 bits collected for steps n = 32 wi + 31 down to 32 wi -> the little-endian word of bytes packed MSB first
__device__ __forceinline__ uint32_t back_word(uint32_t out) { return __builtin_bswap32(out); }

}  // namespace acs
}  // namespace dabphy
