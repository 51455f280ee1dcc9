// tools/ubench/acs_rate.hip -- what the add-compare-select engine (welle.io_amd/csrc/viterbi_acs.h) costs per trellis step when nothing
// but the VALU is in its way, and what decision stores / traceback reads add (not part of the product).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iwelle.io_amd/csrc tools/ubench/acs_rate.hip -o tools/ubench/acs_rate
//   MODE 0: trellis only (decisions folded into a checksum)      MODE 1: + decision stores (8 B per lane and step, as the kernels do)
//   MODE 2: + traceback over the stored decisions
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "viterbi_acs.h"
using namespace dabphy;

template <int MODE>
__global__ void __launch_bounds__(64, 5) k(uint2* dec, unsigned* out, int nsteps, int groups_per_wave, unsigned long long* clocks)
{
    const int lane = threadIdx.x;
    const uint32_t ones = opaque_sgpr(0x01010101u);
    unsigned chk = 0;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int gi = 0; gi < groups_per_wave; gi++) {
        const size_t g = (size_t)blockIdx.x + (size_t)gi * gridDim.x;
        uint2* __restrict__ dec_g = dec + g * (size_t)nsteps * 64;
        uint32_t M[32];
        acs::init(M);
        unsigned h = (unsigned)g * 2654435761u + lane * 40503u;
        int rn = 0;
        for (int s = 0; s < nsteps; s += 6) {
            uint2 d[6];
            h = h * 1664525u + 1013904223u;
            const int a = (int)(h >> 24) - 127, b = (int)((h >> 16) & 255) - 127, c = (int)((h >> 8) & 255) - 127;
            d[0] = acs::step<0>(M, a + b, b, c, ones);
            d[1] = acs::step<1>(M, b + c, c, a, ones);
            d[2] = acs::step<2>(M, c, a, b, ones);
            d[3] = acs::step<3>(M, a, c, b, ones);
            d[4] = acs::step<4>(M, b - c, a, c, ones);
            d[5] = acs::step<5>(M, c + a, b, a, ones);
            if (MODE >= 1) {
#pragma unroll
                for (int k2 = 0; k2 < 6; k2++) dec_g[(uint32_t)((s + k2) * 64 + lane)] = d[k2];
            } else {
#pragma unroll
                for (int k2 = 0; k2 < 6; k2++) chk ^= d[k2].x + d[k2].y;
            }
            if (++rn == acs::RENORM_BLOCKS) { acs::renorm(M); rn = 0; }
        }
        chk ^= M[0] ^ M[31];
        if (MODE >= 2) {
            uint32_t J = 0, outw = 0, rho = (uint32_t)acs::dec_rot((nsteps - 7) % 6);
            const int nbits = nsteps - 6;
            uint2 dq[8], dn[8];
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) dq[k2] = dec_g[(uint32_t)((nbits - 1 - k2 + 6) * 64 + lane)];
            for (int n = nbits - 1; n >= 0; n -= 8) {
                const int m = n >= 8 ? n - 8 : n;
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) dn[k2] = dec_g[(uint32_t)((m - k2 + 6) * 64 + lane)];
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) { acs::back(dq[k2], J, outw, rho); rho = rho == 5 ? 0 : rho + 1; }
                if (((n - 7) & 31) == 0) chk ^= outw;
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) dq[k2] = dn[k2];
            }
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (chk == 0x12345678u) out[0] = chk;
    if (blockIdx.x == 0 && lane == 0) { clocks[0] = c1 - c0; clocks[1] = w1 - w0; }
}

template <int MODE> void run(const char* name, uint2* dec, unsigned* out, unsigned long long* clk, int waves_per_simd, int groups_per_wave)
{
    const int nsteps = 1542, grid = 1024 * waves_per_simd;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, dec, out, nsteps, 1, clk);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, dec, out, nsteps, groups_per_wave, clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; (void)hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    const double steps_per_simd = (double)nsteps * groups_per_wave * waves_per_simd;
    const double mhz = (double)c[0] / ((double)c[1] / 100.0);       // wall_clock64 ticks at 100 MHz
    printf("%-28s waves/SIMD %d, %d groups/wave: %.3f ms; %.0f cycles per trellis step and SIMD at 2.4 GHz, %.0f at the measured shader clock %.0f MHz\n",
           name, waves_per_simd, groups_per_wave, ms, ms * 1e-3 * 2.4e9 / steps_per_simd, ms * 1e-3 * mhz * 1e6 / steps_per_simd, mhz);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const int gpw = argc > 1 ? atoi(argv[1]) : 2;
    uint2* dec; unsigned* out; unsigned long long* clk;
    const size_t bytes = (size_t)1024 * 8 * gpw * 1542 * 64 * 8;
    if (hipMalloc((void**)&dec, bytes) != hipSuccess) { printf("alloc of %zu failed\n", bytes); return 1; }
    (void)hipMalloc((void**)&out, 64); (void)hipMalloc((void**)&clk, 64);
    for (int w : {1, 2, 4, 5}) {
        run<0>("trellis only", dec, out, clk, w, gpw);
        run<1>("+ decision stores", dec, out, clk, w, gpw);
        run<2>("+ stores + traceback", dec, out, clk, w, gpw);
    }
    return 0;
}
