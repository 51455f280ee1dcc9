// welle.io_amd/csrc/k_ingest.hip -- sample ingest: raw device-format IQ -> cf32 in the per-ensemble sample ring.
//
// Replaces CRAWFile::convertSamples (input/raw_file.cpp:324-366), which every file/SDR front end of the reference
// funnels through before OFDMProcessor sees a sample.  Converting on the GPU means 2 bytes per sample cross PCIe for
// u8/s8 (4 for s16) instead of 8 for cf32.  The arithmetic is the reference's: u8 (b - 128) / 128.0, s8 b / 128.0 (both
// exact in float), s16 the raw integer value unscaled -- and its byte order: the reference's "S16LE" assembles
// (byte0 << 8) | byte1 and "S16BE" (byte1 << 8) | byte0; the names are kept, the behaviour too.
#include "dabphy_kernels.h"

namespace dabphy {

__global__ void __launch_bounds__(256) k_ingest(IngestArgs A)
{
    const int b = blockIdx.y;
    const uint8_t* __restrict__ src = A.raw + (size_t)b * A.raw_stride;
    cf32* __restrict__ dst = A.iq + (size_t)b * A.iq_stride;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += (uint64_t)gridDim.x * blockDim.x) {
        cf32 v;
        if (A.format == 1) {                                  // U8
            const uint16_t w = reinterpret_cast<const uint16_t*>(src)[i];
            v.re = (float)((int)(w & 0xff) - 128) * 0.0078125f; v.im = (float)((int)(w >> 8) - 128) * 0.0078125f;
        } else if (A.format == 2) {                           // S8
            const uint16_t w = reinterpret_cast<const uint16_t*>(src)[i];
            v.re = (float)(int8_t)(w & 0xff) * 0.0078125f; v.im = (float)(int8_t)(w >> 8) * 0.0078125f;
        } else {                                              // S16 "LE" (3) / "BE" (4)
            const uint32_t w = reinterpret_cast<const uint32_t*>(src)[i];
            const uint32_t b0 = w & 0xff, b1 = (w >> 8) & 0xff, b2 = (w >> 16) & 0xff, b3 = w >> 24;
            const int16_t I = A.format == 3 ? (int16_t)((b0 << 8) | b1) : (int16_t)((b1 << 8) | b0);
            const int16_t Q = A.format == 3 ? (int16_t)((b2 << 8) | b3) : (int16_t)((b3 << 8) | b2);
            v.re = (float)I; v.im = (float)Q;
        }
        uint64_t p = A.w + i; if (p >= A.ring) p -= A.ring;
        dst[p] = v;
    }
}

// The null symbol that follows frame (b, f), pulled like the reference pulls it: getSamples(nullSymbol, coarse + fine) with
// the correctors as updated by this frame (ofdm-processor.cpp:462-469, NCO :211-214).  Sample j sits T_u + 75 T_s after
// the PRS start and carries phase (null_L - (j + 1) null_f) mod RATE.  Diagnostic tap (spectrum / TII consumers): plain
// table look-ups.
__global__ void __launch_bounds__(256) k_null_symbols(NullArgs A)
{
    const int f = blockIdx.y, b = blockIdx.z;
    const FrameDesc d = A.desc[(size_t)b * A.n_frames + f];
    cf32* out = A.out + ((size_t)b * A.n_frames + f) * T_NULL;
    const cf32* __restrict__ iq = A.iq + (size_t)b * A.iq_stride;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < T_NULL; j += gridDim.x * blockDim.x) {
        cf32 v; v.re = 0.0f; v.im = 0.0f;
        if (d.valid == 1) {
            const int64_t p = (d.pos + d.start_index + T_U + 75LL * T_S + j) % A.ring;
            int64_t ph = ((int64_t)d.null_L - (int64_t)(j + 1) * d.null_f) % INPUT_RATE; if (ph < 0) ph += INPUT_RATE;
            v = cmul(iq[p], A.tab.nco[ph]);
        }
        out[j] = v;
    }
}

void launch_null_symbols(const NullArgs& a, int n_ens, hipStream_t s)
{
    hipLaunchKernelGGL(k_null_symbols, dim3(4, a.n_frames, n_ens), dim3(256), 0, s, a);
}

// Plain device copy, 16 bytes per lane and request, grid-stride, one request per lane and iteration: the measured denominator of the
// FFT stage's roofline figures (dabphy_time_copy, include/dabphy_test.h).  tools/ubench/copy_f4.hip sweeps grid size and unroll depth:
// on the boxes of this pool FEWER work-groups and no unrolling are fastest (profiles/r04_copy_f4.txt).
__global__ void __launch_bounds__(256) k_copy_f4(const float4* __restrict__ src, float4* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
void launch_copy_f4(const void* src, void* dst, size_t n16, int blocks, hipStream_t s)
{
    hipLaunchKernelGGL(k_copy_f4, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), n16);
}

// A batch's results -- descriptors, SNR reports, FIBs, CRC flags, verdict words: a few MB -- into the handle's page-locked host buffers by a
// KERNEL (stores over the fabric), not by a copy-engine packet: a packet queues behind whatever the engine has in hand, e.g. the 100 MB of a
// bulk MSC drain (dabphy_msc_drain_begin), and dabphy_process then waits for that.  Any size and alignment (dwords where both ends allow).
__global__ void __launch_bounds__(256) k_copy_out(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t bytes)
{
    const size_t stride = (size_t)gridDim.x * 256, i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const size_t n16 = bytes / 16;
        const float4* s4 = reinterpret_cast<const float4*>(src); float4* d4 = reinterpret_cast<float4*>(dst);
        for (size_t i = i0; i < n16; i += stride) d4[i] = s4[i];
        for (size_t i = n16 * 16 + i0; i < bytes; i += stride) dst[i] = src[i];
    } else {
        for (size_t i = i0; i < bytes; i += stride) dst[i] = src[i];
    }
}
void launch_copy_out(const void* src_device, void* dst_host, size_t bytes, hipStream_t s)
{
    if (!bytes) return;
    const unsigned blocks = (unsigned)std::min<size_t>((bytes / 16 + 255) / 256 + 1, 256);
    hipLaunchKernelGGL(k_copy_out, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const uint8_t*>(src_device), reinterpret_cast<uint8_t*>(dst_host), bytes);
}

void launch_ingest(const IngestArgs& a, int n_ens, hipStream_t s)
{
    const unsigned blocks = (unsigned)std::min<uint64_t>((a.n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_ingest, dim3(blocks, n_ens), dim3(256), 0, s, a);
}

} // namespace dabphy
