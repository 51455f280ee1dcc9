// welle.io_amd/csrc/k_rs.hip -- DAB+ Reed-Solomon RS(120,110) over GF(2^8), one thread per codeword.
//
// Replaces (reference file:line):
//   RSDecoder::DecodeSuperframe          src/backend/dabplus_decoder.cpp:326-359
//   decode_rs_char / decode_rs.h         src/libs/fec/decode_rs.h:71-298  (KA9Q libfec, no erasures)
//   init_rs_char(8, 0x11D, 0, 1, 10, 135)  src/libs/fec/init_rs.h:6-103   (log/antilog tables)
//
// A superframe of a sub-channel with s = bitrate/8 holds s column-interleaved codewords: codeword i consists of
// bytes sf[pos*s + i], pos = 0..119.  Consecutive threads take consecutive i, so every syndrome step reads s
// consecutive bytes.  The decoder is the reference's algorithm statement by statement (syndromes ->
// Berlekamp-Massey -> Chien -> Forney) with its uint8 index arithmetic, so corrected bytes, the corrected-symbol
// count and the "uncorrectable" verdict match it also for words beyond the correction capacity (miscorrections
// included).  GF tables live in LDS; the rare non-zero-syndrome path works on per-thread arrays.
#include "dabphy_kernels.h"

namespace dabphy {

constexpr int RS_NN = 255, RS_NROOTS = 10, RS_PAD = 135, RS_A0 = 255, RS_LEN = 120;

__device__ __forceinline__ int rs_modnn(int x)
{
    while (x >= RS_NN) { x -= RS_NN; x = (x >> 8) + (x & RS_NN); }
    return x;
}

struct RsIo {                         // byte `pos` of codeword i
    uint8_t* base; size_t pos_stride;
    __device__ __forceinline__ uint8_t get(int pos) const { return base[(size_t)pos * pos_stride]; }
    __device__ __forceinline__ void put(int pos, uint8_t v) const { base[(size_t)pos * pos_stride] = v; }
};

// returns the number of corrected symbols, -1 when uncorrectable (decode_rs.h:71-298, no_eras = 0)
template <typename IO>
__device__ int rs_decode120(const IO& io, const uint8_t* __restrict__ alpha_to, const uint8_t* __restrict__ index_of)
{
    uint8_t s[RS_NROOTS];
    {
        const uint8_t d0 = io.get(0);
#pragma unroll
        for (int i = 0; i < RS_NROOTS; i++) s[i] = d0;
    }
    for (int j = 1; j < RS_LEN; j++) {
        const uint8_t dj = io.get(j);
#pragma unroll
        for (int i = 0; i < RS_NROOTS; i++)
            s[i] = (s[i] == 0) ? dj : (uint8_t)(dj ^ alpha_to[rs_modnn(index_of[s[i]] + i)]);     // FCR = 0, PRIM = 1
    }
    int syn_error = 0;
#pragma unroll
    for (int i = 0; i < RS_NROOTS; i++) { syn_error |= s[i]; s[i] = index_of[s[i]]; }
    if (!syn_error) return 0;

    uint8_t lambda[RS_NROOTS + 1], b[RS_NROOTS + 1], t[RS_NROOTS + 1], omega[RS_NROOTS + 1], root[RS_NROOTS], reg[RS_NROOTS + 1], loc[RS_NROOTS];
    for (int i = 1; i <= RS_NROOTS; i++) lambda[i] = 0;
    lambda[0] = 1;
    for (int i = 0; i <= RS_NROOTS; i++) b[i] = index_of[lambda[i]];
    int r = 0, el = 0;
    while (++r <= RS_NROOTS) {                                   // Berlekamp-Massey
        uint8_t discr = 0;
        for (int i = 0; i < r; i++)
            if (lambda[i] != 0 && s[r - i - 1] != RS_A0) discr ^= alpha_to[rs_modnn(index_of[lambda[i]] + s[r - i - 1])];
        discr = index_of[discr];
        if (discr == RS_A0) {
            for (int i = RS_NROOTS; i > 0; i--) b[i] = b[i - 1];
            b[0] = RS_A0;
        } else {
            t[0] = lambda[0];
            for (int i = 0; i < RS_NROOTS; i++)
                t[i + 1] = (b[i] != RS_A0) ? (uint8_t)(lambda[i + 1] ^ alpha_to[rs_modnn(discr + b[i])]) : lambda[i + 1];
            if (2 * el <= r - 1) {
                el = r - el;
                for (int i = 0; i <= RS_NROOTS; i++) b[i] = (lambda[i] == 0) ? (uint8_t)RS_A0 : (uint8_t)rs_modnn(index_of[lambda[i]] - discr + RS_NN);
            } else {
                for (int i = RS_NROOTS; i > 0; i--) b[i] = b[i - 1];
                b[0] = RS_A0;
            }
            for (int i = 0; i <= RS_NROOTS; i++) lambda[i] = t[i];
        }
    }
    int deg_lambda = 0;
    for (int i = 0; i <= RS_NROOTS; i++) { lambda[i] = index_of[lambda[i]]; if (lambda[i] != RS_A0) deg_lambda = i; }
    for (int i = 1; i <= RS_NROOTS; i++) reg[i] = lambda[i];
    int count = 0;
    for (int i = 1, k = 0; i <= RS_NN; i++, k = rs_modnn(k + 1)) {       // Chien search, IPRIM = 1
        uint8_t q = 1;
        for (int j = deg_lambda; j > 0; j--)
            if (reg[j] != RS_A0) { reg[j] = (uint8_t)rs_modnn(reg[j] + j); q ^= alpha_to[reg[j]]; }
        if (q != 0) continue;
        root[count] = (uint8_t)i; loc[count] = (uint8_t)k;
        if (++count == deg_lambda) break;
    }
    if (deg_lambda != count) return -1;
    const int deg_omega = deg_lambda - 1;
    for (int i = 0; i <= deg_omega; i++) {
        uint8_t tmp = 0;
        for (int j = i; j >= 0; j--)
            if (s[i - j] != RS_A0 && lambda[j] != RS_A0) tmp ^= alpha_to[rs_modnn(s[i - j] + lambda[j])];
        omega[i] = index_of[tmp];
    }
    for (int j = count - 1; j >= 0; j--) {                                 // Forney
        uint8_t num1 = 0;
        for (int i = deg_omega; i >= 0; i--)
            if (omega[i] != RS_A0) num1 ^= alpha_to[rs_modnn(omega[i] + i * root[j])];
        const uint8_t num2 = alpha_to[rs_modnn(root[j] * (0 - 1) + RS_NN)];
        uint8_t den = 0;
        for (int i = (deg_lambda < RS_NROOTS - 1 ? deg_lambda : RS_NROOTS - 1) & ~1; i >= 0; i -= 2)
            if (lambda[i + 1] != RS_A0) den ^= alpha_to[rs_modnn(lambda[i + 1] + i * root[j])];
        if (num1 != 0 && loc[j] >= RS_PAD) {
            const int p = loc[j] - RS_PAD;
            io.put(p, (uint8_t)(io.get(p) ^ alpha_to[rs_modnn(index_of[num1] + index_of[num2] + RS_NN - index_of[den])]));
        }
    }
    return count;
}

__device__ __forceinline__ void rs_tables(uint8_t* alpha_to, uint8_t* index_of, int t, int nthreads)
{
    if (t == 0) {                                              // init_rs.h:48-60
        int sr = 1;
        index_of[0] = RS_A0; alpha_to[RS_A0] = 0;
        for (int i = 0; i < RS_NN; i++) {
            index_of[sr] = (uint8_t)i; alpha_to[i] = (uint8_t)sr;
            sr <<= 1; if (sr & 256) sr ^= 0x11D; sr &= RS_NN;
        }
    }
    (void)nthreads;
    __syncthreads();
}

// Contiguous superframes: sf[n_sf][120*s]; result per superframe: corrected-symbol total and uncorrectable flag.
__global__ void __launch_bounds__(256) k_rs_superframes(RsArgs A)
{
    __shared__ uint8_t alpha_to[256], index_of[256];
    rs_tables(alpha_to, index_of, threadIdx.x, blockDim.x);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = A.n_sf * A.s;
    if (idx >= total) return;
    const int sf = idx / A.s, i = idx % A.s;
    RsIo io; io.base = A.data + (size_t)sf * A.sf_stride + i; io.pos_stride = (size_t)A.s;
    const int c = rs_decode120(io, alpha_to, index_of);
    if (c < 0) atomicOr(A.uncorr + sf, 1);
    else if (c > 0) atomicAdd(A.corr + sf, c);
}

// Superframes inside the MSC output of one protection class: byte k of the superframe that starts at logical
// frame r0 of (ensemble b, member m) lives in frame r0 + k / frame_bytes at offset k % frame_bytes.
struct RsMscIo {
    uint8_t* frames; size_t frame_stride; int frame_bytes, s, i;
    __device__ __forceinline__ uint8_t* at(int pos) const { const int k = pos * s + i; return frames + (size_t)(k / frame_bytes) * frame_stride + (k % frame_bytes); }
    __device__ __forceinline__ uint8_t get(int pos) const { return *at(pos); }
    __device__ __forceinline__ void put(int pos, uint8_t v) const { *at(pos) = v; }
};

__global__ void __launch_bounds__(256) k_rs_msc(RsMscArgs A)
{
    __shared__ uint8_t alpha_to[256], index_of[256];
    rs_tables(alpha_to, index_of, threadIdx.x, blockDim.x);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int per_ens = A.n_sf_per_ens * A.n_members * A.s;
    if (idx >= A.n_ens * per_ens) return;
    const int b = idx / per_ens; int rem = idx % per_ens;
    const int q = rem / (A.n_members * A.s); rem %= (A.n_members * A.s);
    const int m = rem / A.s, i = rem % A.s;
    if (A.member_only >= 0 && m != A.member_only) return;
    const int r0 = A.first_cif[b] + 5 * q;                       // first logical frame of superframe q
    if (r0 < 0 || r0 + 5 > A.n_cif) return;
    RsMscIo io;
    io.frame_bytes = A.frame_bytes; io.s = A.s; io.i = i; io.frame_stride = (size_t)A.frame_bytes;
    io.frames = A.out + (((size_t)b * A.n_members + m) * A.n_cif + r0) * A.frame_bytes;
    const int c = rs_decode120(io, alpha_to, index_of);
    int* cnt = A.result + 2 * (((size_t)b * A.n_sf_per_ens + q) * A.n_members + m);
    if (c < 0) atomicOr(cnt + 1, 1);
    else if (c > 0) atomicAdd(cnt, c);
}

void launch_rs_superframes(const RsArgs& a, hipStream_t s)
{
    const int total = a.n_sf * a.s;
    hipLaunchKernelGGL(k_rs_superframes, dim3((total + 255) / 256), dim3(256), 0, s, a);
}
void launch_rs_msc(const RsMscArgs& a, hipStream_t s)
{
    const int total = a.n_ens * a.n_sf_per_ens * a.n_members * a.s;
    hipLaunchKernelGGL(k_rs_msc, dim3((total + 255) / 256), dim3(256), 0, s, a);
}

} // namespace dabphy
