// welle.io_amd/csrc/fft2048.h -- 2048-point complex FFT for one 128-thread work-group (two wave64).
//
// Replaces fft::Forward / fft::Backward (src/various/fft.cpp:98-164) for T_u = 2048.  The result is
// bit-identical to the reference's KISS FFT build: same mixed-radix decomposition 2048 = 4.4.4.4.4.2
// (kiss_fft.c:309-330), same butterfly arithmetic and operation order (kf_bfly2 :21-42, kf_bfly4 :44-90),
// same float twiddles (double cos/sin rounded to float, :353-364), no FMA contraction.  What is NOT
// taken from KISS is the schedule: the six passes run as three register-resident rounds of 16 points
// per thread, with two exchanges through a 16 KiB LDS tile:
//
//   round A  (leaf copy + radix-2 m=1 + radix-4 m=2)   two blocks of 8 consecutive positions per thread,
//            inputs x[b + 256 j] (b = t, t+128): 64 lanes read 512 contiguous bytes per load
//   round B  (radix-4 m=8, m=32)                        positions 128c + k + 8a + 32b,  t = 8c + k
//   round C  (radix-4 m=128, m=512)                     positions k + 128a + 512b,      t = k
//
// After round C thread t holds bins t + 128 j (j = 0..15) -- exactly the inputs x[b + 256 j'] that round A
// of a following transform needs, so FFT -> multiply -> IFFT (phasereference.cpp:81-90) chains in
// registers.  LDS addressing is XOR-swizzled so that every ds_write_b128 / ds_read_b64 / ds_write_b64 of
// the two exchanges is bank-conflict free on gfx950 (MI355X_MICROARCH.md, LDS table):
//   exchange 1: E1(p) = p ^ (((p>>7)&3)<<3) ^ (((p>>9)&3)<<1)
//   exchange 2: E2(p) = p ^ (((p>>7)&1)<<3)
// Register budget: the 15 round-C twiddles are per-thread constants kept in VGPRs; the 15 round-B twiddles
// come from a 1 KiB LDS copy of tw[16 i] (ds_read_b64, 60 LDS cycles per transform); round-A twiddles are
// wave-uniform (SGPRs).
#pragma once
#include "dabphy_common.h"
#include <dabphy_wave_ops.h>

namespace dabphy {

constexpr int FFT_TWB_ENTRIES = 128;      // tw[16 i], i < 128

struct FftTwiddles {           // forward values; the inverse conjugates them on use
    cf32 t0, a1, a2, a3;       // round A: tw[0], tw[256], tw[512], tw[768]   (uniform)
    cf32 c128[3];              // m=128: tw[4t], tw[8t], tw[12t]
    cf32 c512[4][3];           // m=512: k' = t + 128a: tw[k'], tw[2k'], tw[3k']
    const cf32* twB;           // LDS: tw[16 i]
};

// fills the LDS sub-table (needs a __syncthreads() before first use: fft2048_wg starts with one) and the registers
__device__ __forceinline__ void fft_load_twiddles(FftTwiddles& w, const cf32* __restrict__ tw, cf32* twB_lds, int t)
{
    if (t < FFT_TWB_ENTRIES) twB_lds[t] = tw[16 * t];
    w.twB = twB_lds;
    w.t0 = tw[0]; w.a1 = tw[256]; w.a2 = tw[512]; w.a3 = tw[768];
#pragma unroll
    for (int i = 0; i < 3; i++) w.c128[i] = tw[4 * t * (i + 1)];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int i = 0; i < 3; i++) w.c512[a][i] = tw[(t + 128 * a) * (i + 1)];
}

// x * w (forward) or x * conj(w) (inverse): three packed instructions either way
template <bool INV> __device__ __forceinline__ cf32 twmul(cf32 x, cf32 w) { return INV ? pk_cmulc(x, w) : pk_cmul(x, w); }

// kf_bfly2 (kiss_fft.c:21-42), one butterfly
template <bool INV>
__device__ __forceinline__ void bfly2(cf32& f0, cf32& f1, cf32 tw)
{
    const cf32 t = twmul<INV>(f1, tw);
    f1 = pk_sub(f0, t);
    f0 = pk_add(f0, t);
}

// kf_bfly4 (kiss_fft.c:44-90), one butterfly: 17 packed instructions
template <bool INV>
__device__ __forceinline__ void bfly4(cf32& f0, cf32& f1, cf32& f2, cf32& f3, cf32 tw1, cf32 tw2, cf32 tw3)
{
    const cf32 s0 = twmul<INV>(f1, tw1);
    const cf32 s1 = twmul<INV>(f2, tw2);
    const cf32 s2 = twmul<INV>(f3, tw3);
    const cf32 s5 = pk_sub(f0, s1);
    f0 = pk_add(f0, s1);
    const cf32 s3 = pk_add(s0, s2);
    const cf32 s4 = pk_sub(s0, s2);
    f2 = pk_sub(f0, s3);
    f0 = pk_add(f0, s3);
    if (INV) { f1 = pk_add_ib(s5, s4); f3 = pk_sub_ib(s5, s4); }     // f1 = s5 + i s4, f3 = s5 - i s4
    else     { f1 = pk_sub_ib(s5, s4); f3 = pk_add_ib(s5, s4); }
}

// Round A for one half (h = 0: inputs x[t + 256 j], h = 1: inputs x[t + 128 + 256 j], j = 0..7 in x[]) and the
// swizzled store of its 8 consecutive positions.  The caller must have passed a barrier since the tile was last read.
template <bool INV>
__device__ __forceinline__ void fft_round_a(const cf32 (&x)[8], int h, cf32* lds, const FftTwiddles& w, int t)
{
    cf32 a[8];
#pragma unroll
    for (int j5 = 0; j5 < 4; j5++) { a[2 * j5] = x[j5]; a[2 * j5 + 1] = x[j5 + 4]; }
#pragma unroll
    for (int j5 = 0; j5 < 4; j5++) bfly2<INV>(a[2 * j5], a[2 * j5 + 1], w.t0);
    bfly4<INV>(a[0], a[2], a[4], a[6], w.t0, w.t0, w.t0);
    bfly4<INV>(a[1], a[3], a[5], a[7], w.a1, w.a2, w.a3);
    const int b = t + 128 * h;
    const int j1 = b & 3, j2 = (b >> 2) & 3, j3 = (b >> 4) & 3, j4 = (b >> 6) & 3;
    const int q = 64 * j1 + 16 * j2 + 4 * j3 + (j4 ^ j2);           // chunk index with the E1 row swizzle
    float4* dst = reinterpret_cast<float4*>(lds + 8 * q);
#pragma unroll
    for (int s = 0; s < 4; s++)                                      // 16-byte slot s holds positions 2s, 2s+1
        dst[s ^ j1] = make_float4(a[2 * s].re, a[2 * s].im, a[2 * s + 1].re, a[2 * s + 1].im);
}

// Rounds B and C.  Out: v[j] = X[t + 128 j].  Starts with the barrier that publishes round A's stores; `after_a` runs
// right behind that barrier (every thread has consumed its round-A inputs: the place to start fetching the next symbol).
struct FftNoHook { __device__ __forceinline__ void operator()() const {} };
template <bool INV, typename Hook = FftNoHook>
__device__ __forceinline__ void fft_rounds_bc(cf32 (&v)[16], cf32* lds, const FftTwiddles& w, int t, Hook after_a = Hook())
{
    __syncthreads();
    after_a();
    {
        const int c = t >> 3, k = t & 7;
        const int sw = (8 * (c & 3)) ^ (2 * (c >> 2));
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int a = 0; a < 4; a++) v[4 * b + a] = lds[(128 * c + k + 8 * a + 32 * b) ^ sw];
        {
            const cf32 w1 = w.twB[4 * k], w2 = w.twB[8 * k], w3 = w.twB[12 * k];   // tw[64k(i+1)]
#pragma unroll
            for (int b = 0; b < 4; b++) bfly4<INV>(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3], w1, w2, w3);
        }
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const int kp = k + 8 * a;                                                                              // tw[16 k'(i+1)]
            bfly4<INV>(v[a], v[4 + a], v[8 + a], v[12 + a], w.twB[kp], w.twB[2 * kp], w.twB[3 * kp]);
        }
        __syncthreads();        // everyone has read exchange 1
        const int sw2 = 8 * (c & 1);
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int a = 0; a < 4; a++) lds[(128 * c + k + 8 * a + 32 * b) ^ sw2] = v[4 * b + a];
    }
    __syncthreads();
    {
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int a = 0; a < 4; a++) v[4 * b + a] = lds[(t + 128 * a + 512 * b) ^ (8 * (a & 1))];
#pragma unroll
        for (int b = 0; b < 4; b++)
            bfly4<INV>(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3], w.c128[0], w.c128[1], w.c128[2]);
#pragma unroll
        for (int a = 0; a < 4; a++)
            bfly4<INV>(v[a], v[4 + a], v[8 + a], v[12 + a], w.c512[a][0], w.c512[a][1], w.c512[a][2]);
    }
    // v[4b + a] = X[t + 128a + 512b] = X[t + 128 (a + 4b)]  -> already in j = a + 4b order
}

// ---- k_demod's variant of the two exchanges (DEMOD_STAGE 2): the raw samples of a symbol arrive IN the tile (LDS-DMA, 16-byte units
// = sample pairs), round A overwrites each sample it read with one of its results (no barrier between reading and writing: a thread
// only touches its own 16 elements), round B picks the positions up from there, and the second exchange is laid out so that each wave
// reads only ITS half of the tile in round C -- the wave may then start the next symbol's transfer into that half without waiting for
// the other one (4 barriers per symbol as before, 16 KiB of LDS less).  tools/layout/demod_inplace_layout.py checks the algebra:
//   raw sample n, and result r of thread-half (t, h) written over sample t + 128h + 256r:  element (n & 255) + 260 (n >> 8) -- rows
//       of 256 samples 32 bytes apart: every address below is one register + an immediate
//   position 8q + r, q = 64 j1 + 16 j2 + 4 j3 + j4, is result r of thread-half j1 + 4 j2 + 16 j3 + 64 j4
//   round B thread t = k + 8 j1 + 32 j2 works c = 4 j1 + j2: the 32 lanes of a read group differ in k and j1, elements j1 + 4k mod 32
//   second exchange: E2(p) = 1040 (p >> 6 & 1) + (q ^ (q >> 8 & 1) << 3), q = (p & 63) | (p >> 7) << 6
// every ds_read_b64 / ds_write_b64 is conflict-free by the rules of MI355X_MICROARCH.md (read groups of 32 lanes, write groups of 16).
constexpr int FFT_RAW_PITCH = 260;                       // elements per row of 256 raw samples
constexpr int FFT_INPLACE_TILE = 8 * FFT_RAW_PITCH;      // elements of the tile
__device__ __forceinline__ int fft_raw_index(int t, int h, int j) { return t + 128 * h + FFT_RAW_PITCH * j; }

// the 8 samples x[t + 128h + 256j] of one half out of the tile
__device__ __forceinline__ void fft_raw_half(cf32 (&x)[8], const cf32* lds, int h, int t)
{
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = lds[fft_raw_index(t, h, j)];
}

// kf_bfly2 / kf_bfly4 with the unit twiddle tw[0] (the first two passes of the forward transform): same operations, the
// multiplications by (1, -0) as one instruction each (pk_cmul_unit)
__device__ __forceinline__ void bfly2_unit(cf32& f0, cf32& f1, cf32 tw)
{
    const cf32 t = pk_cmul_unit(f1, tw);
    f1 = pk_sub(f0, t);
    f0 = pk_add(f0, t);
}
__device__ __forceinline__ void bfly4_unit(cf32& f0, cf32& f1, cf32& f2, cf32& f3, cf32 tw)
{
    const cf32 s0 = pk_cmul_unit(f1, tw);
    const cf32 s1 = pk_cmul_unit(f2, tw);
    const cf32 s2 = pk_cmul_unit(f3, tw);
    const cf32 s5 = pk_sub(f0, s1);
    f0 = pk_add(f0, s1);
    const cf32 s3 = pk_add(s0, s2);
    const cf32 s4 = pk_sub(s0, s2);
    f2 = pk_sub(f0, s3);
    f0 = pk_add(f0, s3);
    f1 = pk_sub_ib(s5, s4); f3 = pk_add_ib(s5, s4);
}

template <bool INV>
__device__ __forceinline__ void fft_round_a_inplace(const cf32 (&x)[8], int h, cf32* lds, const FftTwiddles& w, int t)
{
    static_assert(!INV, "forward transform only (k_demod)");
    cf32 a[8];
#pragma unroll
    for (int j5 = 0; j5 < 4; j5++) { a[2 * j5] = x[j5]; a[2 * j5 + 1] = x[j5 + 4]; }
#pragma unroll
    for (int j5 = 0; j5 < 4; j5++) bfly2_unit(a[2 * j5], a[2 * j5 + 1], w.t0);
    bfly4_unit(a[0], a[2], a[4], a[6], w.t0);
    bfly4<INV>(a[1], a[3], a[5], a[7], w.a1, w.a2, w.a3);
#pragma unroll
    for (int r = 0; r < 8; r++) lds[fft_raw_index(t, h, r)] = a[r];
}

// Rounds B and C over a tile round A wrote in place.  Out: v[j] = X[t + 128 j].  `after_c_reads` runs when this WAVE's reads of
// round C have returned: from then on nobody reads the wave's half of the tile.
template <bool INV, typename Hook = FftNoHook>
__device__ __forceinline__ void fft_rounds_bc_split(cf32 (&v)[16], cf32* lds, const FftTwiddles& w, int t, Hook after_c_reads = Hook())
{
    __syncthreads();
    {
        const int k = t & 7, j1 = (t >> 3) & 3, j2 = t >> 5, c = 4 * j1 + j2;
        {
            const cf32* rb = lds + (j1 + 4 * j2 + FFT_RAW_PITCH * k);
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int a = 0; a < 4; a++) v[4 * b + a] = rb[16 * b + 64 * a];
        }
        {
            const cf32 w1 = w.twB[4 * k], w2 = w.twB[8 * k], w3 = w.twB[12 * k];   // tw[64k(i+1)]
#pragma unroll
            for (int b = 0; b < 4; b++) bfly4<INV>(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3], w1, w2, w3);
        }
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const int kp = k + 8 * a;                                                                              // tw[16 k'(i+1)]
            bfly4<INV>(v[a], v[4 + a], v[8 + a], v[12 + a], w.twB[kp], w.twB[2 * kp], w.twB[3 * kp]);
        }
        __syncthreads();        // everyone has read exchange 1
        cf32* wb0 = lds + (k + 64 * c + 8 * (j1 & 1)), * wb1 = lds + (k + 64 * c + 8 * ((j1 & 1) ^ 1));
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int a = 0; a < 4; a++) ((a & 1) ? wb1 : wb0)[16 * (a >> 1) + 32 * (b & 1) + 4 * FFT_RAW_PITCH * (b >> 1)] = v[4 * b + a];
    }
    __syncthreads();
    {
        const cf32* r0 = lds + (4 * FFT_RAW_PITCH * (t >> 6) + (t & 63)), * r1 = lds + (4 * FFT_RAW_PITCH * (t >> 6) + ((t & 63) ^ 8));
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int a = 0; a < 4; a++) v[4 * b + a] = ((b & 1) ? r1 : r0)[64 * a + 256 * b];
        lds_reads_done();
        after_c_reads();
#pragma unroll
        for (int b = 0; b < 4; b++)
            bfly4<INV>(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3], w.c128[0], w.c128[1], w.c128[2]);
#pragma unroll
        for (int a = 0; a < 4; a++)
            bfly4<INV>(v[a], v[4 + a], v[8 + a], v[12 + a], w.c512[a][0], w.c512[a][1], w.c512[a][2]);
    }
}

// Whole transform.  In: v[8h + j] = x[t + 128h + 256j].  Out: v[j] = X[t + 128 j].  4 barriers.
template <bool INV>
__device__ __forceinline__ void fft2048_wg(cf32 (&v)[16], cf32* lds, const FftTwiddles& w, int t)
{
    __syncthreads();            // previous users of the tile are done
#pragma unroll
    for (int h = 0; h < 2; h++) {
        cf32 x[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = v[8 * h + j];
        fft_round_a<INV>(x, h, lds, w, t);
    }
    fft_rounds_bc<INV>(v, lds, w, t);
}

} // namespace dabphy
