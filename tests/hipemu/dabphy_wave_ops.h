// tests/hipemu/dabphy_wave_ops.h -- TEST INFRASTRUCTURE: functional model of welle.io_amd/csrc/dabphy_wave_ops.h for the
// CPU execution model (same results, lane exchange through the emulated wave instead of DPP instructions).
#pragma once
#include <hip/hip_runtime.h>
#include "dabphy_common.h"

namespace dabphy {

static inline float chain16(float acc, float x, int nk)
{
    // lane i adds x of lanes i, i+1, ... of its row of 16 (0 beyond the row), exactly like row_shl with bound_ctrl
    unsigned xb; memcpy(&xb, &x, 4);
    const int lane = hipemu_lane();
    for (int k = 0; k < nk; k++) {
        const bool in_row = (lane & 15) + k < 16;
        const unsigned v = hipemu_wave_exchange(xb, in_row ? lane + k : lane, true);
        float f; memcpy(&f, &v, 4);
        acc += in_row ? f : 0.0f;
    }
    return acc;
}

// packed complex arithmetic: the same IEEE operations, spelled out
static inline cf32 pk_add(cf32 a, cf32 b) { return cadd(a, b); }
static inline cf32 pk_sub(cf32 a, cf32 b) { return csub(a, b); }
static inline cf32 pk_cmul(cf32 a, cf32 b) { return cmul(a, b); }
static inline cf32 pk_cmulc(cf32 a, cf32 b) { return cmul(a, cconj(b)); }
static inline cf32 pk_cmul_unit(cf32 a, cf32 w) { return cmul(a, w); }
static inline cf32 pk_sub_ib(cf32 a, cf32 b) { cf32 r; r.re = a.re + b.im; r.im = a.im - b.re; return r; }
static inline cf32 pk_add_ib(cf32 a, cf32 b) { cf32 r; r.re = a.re - b.im; r.im = a.im + b.re; return r; }
static inline int cvt_i32_trunc(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 0x7fffffff;
    if (x <= -2147483648.0f) return (int)0x80000000;
    return (int)x;
}

typedef unsigned short u16x2 __attribute__((vector_size(4)));
static inline u16x2 pk_dup_lo(u16x2 v) { u16x2 r = {v[0], v[0]}; return r; }
static inline u16x2 pk_dup_hi(u16x2 v) { u16x2 r = {v[1], v[1]}; return r; }
static inline u16x2 pk_swap(u16x2 v) { u16x2 r = {v[1], v[0]}; return r; }
static inline uint32_t opaque_sgpr(uint32_t x) { return x; }
static inline double uniform_f64(double x) { return x; }
static inline uint32_t and_or(uint32_t x, uint32_t mask, uint32_t acc) { return (x & mask) | acc; }
static inline uint32_t pk_sign_bytes(u16x2 a, u16x2 b)
{
    return ((a[0] & 0x8000) ? 0xffu : 0u) | ((b[0] & 0x8000) ? 0xff00u : 0u) | ((a[1] & 0x8000) ? 0xff0000u : 0u) | ((b[1] & 0x8000) ? 0xff000000u : 0u);
}

static inline int mul_i24(int x, int m) { return (int)((uint32_t)x * (uint32_t)m); }
static inline int mad_i24(int x, int m, int acc) { return (int)((uint32_t)x * (uint32_t)m + (uint32_t)acc); }
// lanes = trellis states: pairs differ in lane bit B; x = the value of the pair's lane with bit B clear, y = of the lane with it set
template <int B> static inline void pair_values(uint32_t m, uint32_t& x, uint32_t& y)
{
    const int lane = hipemu_lane();
    x = hipemu_wave_exchange(m, lane & ~(1 << B), true);
    y = hipemu_wave_exchange(m, lane | (1 << B), true);
}
static inline int mad_i24_vv(int x, int m, int acc) { return (int)((uint32_t)x * (uint32_t)m + (uint32_t)acc); }
static inline uint32_t sub_borrow(uint32_t y, uint32_t x, unsigned long long mask) { return y - x - (uint32_t)((mask >> hipemu_lane()) & 1ull); }
// two code words per wavefront: pairs along lane bit B inside each half of 32 lanes (see the product header)
static inline void swap16(uint32_t r0, uint32_t r1, uint32_t& a, uint32_t& b)
{
    const int lane = hipemu_lane(); const bool set = (lane >> 4) & 1;
    const uint32_t p0 = hipemu_wave_exchange(r0, lane ^ 16, true), p1 = hipemu_wave_exchange(r1, lane ^ 16, true);
    a = set ? p1 : r0; b = set ? r1 : p0;
}
static inline uint32_t ubfe(uint32_t x, uint32_t off, uint32_t width) { return (x >> off) & ((width >= 32 ? 0u : (1u << width)) - 1u); }
static inline uint32_t bit_reverse32(uint32_t x) { uint32_t r = 0; for (int i = 0; i < 32; i++) r |= ((x >> i) & 1u) << (31 - i); return r; }
static inline void swap32(uint32_t r0, uint32_t r1, uint32_t& a, uint32_t& b)
{
    const int lane = hipemu_lane(); const bool up = lane >= 32;
    const uint32_t p0 = hipemu_wave_exchange(r0, lane ^ 32, true), p1 = hipemu_wave_exchange(r1, lane ^ 32, true);
    a = up ? p1 : r0; b = up ? r1 : p0;
}
template <int B> static inline uint32_t partner(uint32_t g) { return hipemu_wave_exchange(g, hipemu_lane() ^ (1 << B), true); }
static inline uint32_t lane_get(uint32_t v, uint32_t idx) { return hipemu_wave_exchange(v, (int)(idx & 63), true); }
static inline uint32_t funnel_shr(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((((unsigned long long)hi) << 32) | lo) >> (sh & 31)); }

// buffer addressing model: the resource is the base pointer
typedef char* BufRsrc;
static inline BufRsrc buf_rsrc(const void* base) { return (char*)base; }
template <int AUX = 0> static inline uint2 buf_load_b64(BufRsrc r, uint32_t lane_bytes, uint32_t uniform_bytes) { uint2 v; memcpy(&v, r + lane_bytes + uniform_bytes, 8); return v; }
template <int AUX = 0> static inline void buf_store_b64(BufRsrc r, uint32_t lane_bytes, uint32_t uniform_bytes, uint2 v) { memcpy(r + lane_bytes + uniform_bytes, &v, 8); }
static inline void buf_store_b32(BufRsrc r, uint32_t lane_bytes, uint32_t uniform_bytes, uint32_t v) { memcpy(r + lane_bytes + uniform_bytes, &v, 4); }
static inline BufRsrc buf_rsrc_4g(const void* base) { return (char*)base; }
static inline void buf_dma4(BufRsrc r, uint32_t lane_bytes, uint32_t uniform_bytes, void* lds_wave_base) { memcpy((char*)lds_wave_base + 4 * hipemu_lane(), r + lane_bytes + uniform_bytes, 4); }

template <int VARIANT> static inline float div127_fast(float x) { return 127.0f / x; }
constexpr float DIV127_LO = 0x1p-100f, DIV127_HI = 0x1p100f;
// the model decides per lane: both sides of a wave_all() branch must compute the same result wherever the fast side is legal
static inline bool wave_all(bool p) { return p; }

// LDS-DMA model: the copy happens at issue time
template <int OFF> static inline void lds_dma16(const void* gptr, void* lds_wave_base) { memcpy((char*)lds_wave_base + OFF + 16 * hipemu_lane(), (const char*)gptr + OFF, 16); }
static inline void lds_dma4(const void* gptr, void* lds_wave_base) { memcpy((char*)lds_wave_base + 4 * hipemu_lane(), gptr, 4); }
// the hardware executes a wave's DMA requests for all lanes at once; the fibers of the model do not run in lock-step, so
// "the data has landed" must also mean "every lane of the wave has issued its part": a wave-wide rendezvous
static inline uint32_t opaque_vgpr(uint32_t x) { return x; }
#define DABPHY_CONST_AS
template <typename T> static inline const T* as_constant(const T* p) { return p; }
static inline uint32_t u32_max(uint32_t a, uint32_t b) { return a > b ? a : b; }
#define DABPHY_WAVES_PER_SIMD(n)
static inline void wave_converge() { (void)hipemu_wave_exchange(0u, hipemu_lane(), true); }
static inline int uniform_i32(int x) { return x; }
static inline void lds_reads_done() { (void)hipemu_wave_exchange(0u, hipemu_lane(), true); }     // every lane of the wave has read
static inline void lds_dma_wait() { (void)hipemu_wave_exchange(0u, hipemu_lane(), true); }
template <int N> static inline void lds_dma_wait_but() { (void)hipemu_wave_exchange(0u, hipemu_lane(), true); }

// accumulate into a double in LDS from many threads (order irrelevant to its users)
static inline void lds_add_f64(double* p, double v) { *p += v; }      // the fibers of a block take turns: no race in the model

} // namespace dabphy
