// welle.io_amd/csrc/k_rs.hip -- DAB+ Reed-Solomon RS(120,110) over GF(2^8), one thread per codeword.
//
// Replaces (reference file:line):
//   RSDecoder::DecodeSuperframe          src/backend/dabplus_decoder.cpp:326-359
//   decode_rs_char / decode_rs.h         src/libs/fec/decode_rs.h:71-298  (KA9Q libfec, no erasures)
//   init_rs_char(8, 0x11D, 0, 1, 10, 135)  src/libs/fec/init_rs.h:6-103   (log/antilog tables)
//
// A superframe of a sub-channel with s = bitrate/8 holds s column-interleaved codewords: codeword i consists of
// bytes sf[pos*s + i], pos = 0..119.  Consecutive threads take consecutive i, so every syndrome step reads s
// consecutive bytes.  Syndromes -> locator (Massey) -> roots -> values, register-resident (rs_correct120); corrected
// bytes, the corrected-symbol count and the "uncorrectable" verdict equal the reference's also for words beyond the
// correction capacity (miscorrections included: tests/golden rs vectors, random error patterns of weight 0 .. 12 against
// the oracle, which is pinned to decode_rs_char itself).  GF tables live in LDS.
//
// Second half of the file: the DAB+ superframe filter (SuperframeFilter::Feed / CheckSync, dabplus_decoder.cpp:50-213) on the
// class output of a batch -- k_superframe_wide + k_superframe_settle (every attempt a receiver in lock makes in the batch at once,
// accepted per sub-channel iff all of them synchronise) and k_superframe (the reference's walk frame by frame, for what is left).
#include "dabphy_kernels.h"
#include <dabphy_wave_ops.h>

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
__device__ __forceinline__ int rs_correct120(const IO& io, const uint32_t (&syn)[RS_NROOTS], const uint8_t* __restrict__ alpha_to, const uint8_t* __restrict__ index_of, uint8_t* ws);

template <typename IO>
__device__ __forceinline__ int rs_decode120(const IO& io, const uint8_t* __restrict__ alpha_to, const uint8_t* __restrict__ index_of, uint8_t* ws)
{
    // the syndromes of the common case live in registers: `syn` is only ever indexed by unrolled constants (the array the
    // Berlekamp-Massey iteration indexes dynamically is a copy made on the error path -- sharing one array sent every
    // Horner step through scratch memory, 10x slower)
    uint32_t syn[RS_NROOTS];
    {
        const uint32_t d0 = io.get(0);
#pragma unroll
        for (int i = 0; i < RS_NROOTS; i++) syn[i] = d0;
    }
#pragma unroll 2
    for (int j = 1; j < RS_LEN; j++) {
        const uint32_t dj = io.get(j);
#pragma unroll
        for (int i = 0; i < RS_NROOTS; i++)
        {
            // FCR = 0, PRIM = 1.  index_of[..] + i <= 254 + 9, where modnn() is one conditional subtraction: written
            // branch-free so that the ten Horner chains overlap their two table look-ups instead of queueing behind ten loops
            // Both look-ups are unconditional (index_of[0] = 255 is harmless, its result is discarded by the select): a
            // look-up under `if (syn != 0)` is a branch per syndrome, ten serial round trips to LDS per byte.
            const uint32_t e = index_of[syn[i]] + i;
            const uint32_t a = alpha_to[e >= RS_NN ? e - RS_NN : e];
            syn[i] = (syn[i] == 0) ? dj : (dj ^ a);
        }
    }
    return rs_correct120(io, syn, alpha_to, index_of, ws);
}

// Errors from the ten syndromes on.  The polynomials are field VALUES (not logarithms) in a small per-thread LDS workspace -- the
// path is rare and serial, what matters is that it costs the callers' hot loops neither registers nor scratch memory (a fully
// unrolled register-resident version spilled the superframe filter's syndrome loop; the first version indexed per-thread arrays in
// scratch) -- and every data-dependent choice is a select.
// What has to equal the reference (decode_rs.h:117-298, no erasures) is the mathematics, including what it does beyond the
// correction capacity -- a decoder that "detects" more or fewer uncorrectable words, or applies different wrong corrections,
// would not give the reference's bytes:
//   * Massey's iteration in the form that multiplies the auxiliary polynomial by x every round: the locator after ten rounds
//     is the same whatever representation is used;
//   * the word is given up (-1) exactly when the locator's degree differs from the number of its roots among alpha^1 .. alpha^255;
//   * roots at positions inside the 135 bytes of padding are counted but not applied; a zero evaluator value is not applied;
//   * a zero derivative is NOT treated as a failure (the reference checks that only in DEBUG builds): its logarithm reads 255
//     from the table and the division degenerates into a multiplication by one.
constexpr int RS_WS_BYTES = 64;                      // per-thread workspace: lam[11] aux[11] term[11] omega[10] root[10]
__device__ __forceinline__ uint32_t gf_mul(uint32_t a, uint32_t b, const uint8_t* __restrict__ alpha_to, const uint8_t* __restrict__ index_of)
{
    uint32_t e = (uint32_t)index_of[a] + index_of[b];            // <= 254 + 254 when both are non-zero
    e = e >= RS_NN ? e - RS_NN : e;
    const uint32_t v = alpha_to[e >= RS_NN ? e - RS_NN : e];      // (index_of[0] = 255 can push e to 510: folded twice, result discarded)
    return (a != 0 && b != 0) ? v : 0u;
}
// a * alpha^k, k in 0 .. 254
__device__ __forceinline__ uint32_t gf_scale(uint32_t a, uint32_t k, const uint8_t* __restrict__ alpha_to, const uint8_t* __restrict__ index_of)
{
    uint32_t e = (uint32_t)index_of[a] + k;
    e = e >= RS_NN ? e - RS_NN : e;
    return a != 0 ? (uint32_t)alpha_to[e >= RS_NN ? e - RS_NN : e] : 0u;
}

template <typename IO>
__device__ __forceinline__ int rs_correct120(const IO& io, const uint32_t (&syn_regs)[RS_NROOTS], const uint8_t* __restrict__ alpha_to, const uint8_t* __restrict__ index_of,
                                             uint8_t* ws /* RS_WS_BYTES of LDS owned by this thread */)
{
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < RS_NROOTS; i++) any |= syn_regs[i];
    if (!any) return 0;
    constexpr int T2 = RS_NROOTS;                    // 2t = 10 syndromes, locator degree <= 10
    uint8_t* const lam = ws; uint8_t* const aux = ws + 11; uint8_t* const term = ws + 22; uint8_t* const omega = ws + 33; uint8_t* const root = ws + 43;
    uint8_t* const syn = ws + 53;                    // (a dynamically indexed copy: the register array is only indexed by unrolled constants)
#pragma unroll
    for (int i = 0; i < T2; i++) syn[i] = (uint8_t)syn_regs[i];
    // ---- locator polynomial: Massey's iteration, aux <- x aux every round
    for (int i = 0; i <= T2; i++) { lam[i] = i == 0; aux[i] = i == 0; }
    int L = 0;
#pragma unroll 1
    for (int r = 0; r < T2; r++) {
        uint32_t d = 0;                              // discrepancy: sum lam[i] S[r - i]
        for (int i = 0; i <= r; i++) d ^= gf_mul(lam[i], syn[r - i], alpha_to, index_of);
        for (int i = T2; i > 0; i--) aux[i] = aux[i - 1];
        aux[0] = 0;
        if (d != 0) {
            const bool grow = 2 * L <= r;
            const uint32_t ld = index_of[d];
            const uint32_t dinv = alpha_to[ld == 0 ? 0 : RS_NN - ld];            // 1 / d
            for (int i = 0; i <= T2; i++) {
                const uint32_t li = lam[i];
                const uint32_t next = li ^ gf_mul(d, aux[i], alpha_to, index_of);
                if (grow) aux[i] = (uint8_t)gf_mul(li, dinv, alpha_to, index_of);
                lam[i] = (uint8_t)next;
            }
            if (grow) L = r + 1 - L;
        }
    }
    int deg = 0;
    for (int i = 1; i <= T2; i++) if (lam[i] != 0) deg = i;
    // ---- roots: Lambda(alpha^i) for i = 1 .. 255, terms advanced by alpha^j per step; position of a root: i - 1 (of the padded word)
    for (int j = 1; j <= T2; j++) term[j] = lam[j];
    int count = 0;
#pragma unroll 1
    for (int i = 1; i <= RS_NN && count < deg; i++) {
        uint32_t q = 1;
        for (int j = 1; j <= deg; j++) { const uint32_t t = gf_scale(term[j], (uint32_t)j, alpha_to, index_of); term[j] = (uint8_t)t; q ^= t; }
        if (q == 0) root[count++] = (uint8_t)i;
    }
    if (count != deg) return -1;
    // ---- evaluator: omega = S Lambda mod x^deg
    for (int k = 0; k < deg; k++) {
        uint32_t v = 0;
        for (int j = 0; j <= k; j++) v ^= gf_mul(syn[k - j], lam[j], alpha_to, index_of);
        omega[k] = (uint8_t)v;
    }
    // ---- values: omega(alpha^i) alpha^(-i) / Lambda'(alpha^i), applied where the position lies in the 120 transmitted bytes
#pragma unroll 1
    for (int n = 0; n < count; n++) {
        const uint32_t i = root[n];
        uint32_t num = 0, den = 0, p = 0;            // p = (k i) mod 255
        for (int k = 0; k < T2; k++) {
            if (k < deg) num ^= gf_scale(omega[k], p, alpha_to, index_of);
            if ((k & 1) == 0) den ^= gf_scale(lam[k + 1], p, alpha_to, index_of);    // formal derivative: the odd coefficients
            p += i; p = p >= RS_NN ? p - RS_NN : p;
        }
        const int pos = (int)i - 1 - RS_PAD;
        if (num != 0 && pos >= 0) {
            const uint32_t xinv = i == RS_NN ? 0u : RS_NN - i;                       // log of alpha^(-i)
            const uint32_t lden = index_of[den];                                      // 255 for den = 0 (see above)
            uint32_t e = (uint32_t)index_of[num] + xinv + RS_NN - lden;               // <= 254 + 254 + 255
            e = e >= 2 * RS_NN ? e - 2 * RS_NN : e; e = e >= RS_NN ? e - RS_NN : e;
            io.put(pos, (uint8_t)(io.get(pos) ^ alpha_to[e]));
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
    __shared__ uint8_t ws[256 * RS_WS_BYTES];
    rs_tables(alpha_to, index_of, threadIdx.x, blockDim.x);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = A.n_sf * A.s;
    if (idx >= total) return;
    const int sf = idx / A.s, i = idx % A.s;
    RsIo io; io.base = A.data + (size_t)sf * A.sf_stride + i; io.pos_stride = (size_t)A.s;
    const int c = rs_decode120(io, alpha_to, index_of, ws + threadIdx.x * RS_WS_BYTES);
    if (c < 0) atomicOr(A.uncorr + sf, 1);
    else if (c > 0) atomicAdd(A.corr + sf, c);
}

// Superframes inside the MSC output of one protection class: byte k of the superframe that starts at logical
// frame r0 of a pair lives in frame r0 + k / frame_bytes at offset k % frame_bytes.
struct RsMscIo {
    uint8_t* frames; size_t frame_stride; int frame_bytes, s, i;
    __device__ __forceinline__ uint8_t* at(int pos) const { const int k = pos * s + i; return frames + (size_t)(k / frame_bytes) * frame_stride + (k % frame_bytes); }
    __device__ __forceinline__ uint8_t get(int pos) const { return *at(pos); }
    __device__ __forceinline__ void put(int pos, uint8_t v) const { *at(pos) = v; }
};

__global__ void __launch_bounds__(256) k_rs_msc(RsMscArgs A)
{
    __shared__ uint8_t alpha_to[256], index_of[256];
    __shared__ uint8_t ws[256 * RS_WS_BYTES];
    rs_tables(alpha_to, index_of, threadIdx.x, blockDim.x);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int per_pair = A.n_sf_per_pair * A.s;
    if (idx >= A.n_pairs * per_pair) return;
    const int pair = idx / per_pair; const int rem = idx % per_pair;
    const int q = rem / A.s, i = rem % A.s;
    const MscPair pp = A.pairs[pair];
    if (A.idx_only >= 0 && pp.idx != A.idx_only) return;
    const int r0 = A.first_cif[pp.ens] + 5 * q;                  // first logical frame of superframe q
    if (r0 < 0 || r0 + 5 > A.n_cif) return;
    RsMscIo io;
    io.frame_bytes = A.frame_bytes; io.s = A.s; io.i = i; io.frame_stride = (size_t)A.frame_bytes;
    io.frames = A.out + ((size_t)pair * A.n_cif + r0) * A.frame_bytes;
    const int c = rs_decode120(io, alpha_to, index_of, ws + threadIdx.x * RS_WS_BYTES);
    int* cnt = A.result + 2 * ((size_t)pair * A.n_sf_per_pair + q);
    if (c < 0) atomicOr(cnt + 1, 1);
    else if (c > 0) atomicAdd(cnt, c);
}

// ------------------------------------------------------------------------------------------ DAB+ superframe filter
// SuperframeFilter::Feed / CheckSync (dabplus_decoder.cpp:50-213) for one sub-channel of every ensemble: one 64-thread
// work-group per ensemble walks the logical frames of the batch in order with the reference's state machine -- 5-frame
// sliding window, Reed-Solomon on a copy (thread i = codeword i), Fire-code / AU-table check, AU CRCs (thread i = access
// unit i), and after a hit a fresh window -- and carries frame_count + the raw window to the next batch.  Only events and
// corrected, synchronised superframes leave the device.
__device__ __forceinline__ uint16_t crc16_msb(const uint8_t* data, int len, bool initial_invert, bool final_invert, uint16_t poly)
{
    uint16_t crc = initial_invert ? 0xFFFF : 0x0000;                           // CalcCRC, tools.cpp:41-72
    for (int o = 0; o < len; o++) {
        crc ^= (uint16_t)(data[o] << 8);
        for (int i = 0; i < 8; i++) crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ poly) : (uint16_t)(crc << 1);
    }
    return final_invert ? (uint16_t)~crc : crc;
}

// the same CRC a byte at a time through a 256-entry table of the polynomial (LDS): crc' = (crc << 8) ^ tab[(crc >> 8) ^ byte]
__device__ __forceinline__ uint16_t crc16_msb_tab(const uint8_t* data, int len, bool initial_invert, bool final_invert, const uint16_t* tab)
{
    uint32_t crc = initial_invert ? 0xFFFFu : 0x0000u;
    for (int o = 0; o < len; o++) crc = ((crc << 8) & 0xFFFFu) ^ tab[(crc >> 8) ^ data[o]];
    return (uint16_t)(final_invert ? ~crc : crc);
}

constexpr int SF_PREFETCH = 6;      // rows of the class output in flight per work-group (serial walk)

// GF(256) tables of the code (init_rs.h:48-60) from the copy the host uploaded, and the byte-wise table of CRC-16-CCITT (0x1021)
__device__ __forceinline__ void sf_tables(const SfArgs& A, uint8_t* alpha_to, uint8_t* index_of, uint16_t* crctab, int t)
{
    reinterpret_cast<uint32_t*>(alpha_to)[t] = reinterpret_cast<const uint32_t*>(A.gf)[t];
    reinterpret_cast<uint32_t*>(index_of)[t] = reinterpret_cast<const uint32_t*>(A.gf + 256)[t];
    if (crctab)
        for (int v = t; v < 256; v += 64) {
            uint16_t c = (uint16_t)(v << 8);
            for (int i = 0; i < 8; i++) c = (c & 0x8000) ? (uint16_t)((c << 1) ^ 0x1021) : (uint16_t)(c << 1);
            crctab[v] = c;
        }
}

// What the batch holds for one (ensemble, sub-channel) pair: fc frames carried in, of which the window machine still uses the last cu (a full
// window that failed drops its oldest frame with the next one, dabplus_decoder.cpp:78-81); rows [r_first, n_rows) of the class output
// are the frames it will be fed (rows are packed: k_msc_gather lays the logical frames of an ensemble out in CIF order from the first
// frame number of the batch, whatever slots the demodulated frames occupied; frames before the 16-CIF fill of the time de-interleaver,
// dab-audio.cpp:146-149, counted from the CIF at which the pair was selected (MscPair::cif0), are never emitted).  If every attempt synchronises, attempts happen at frames 5q + 4 of that sequence: nq of them.
struct SfPlan { int fc, cu, n_rows, r_first, avail, nq; };
__device__ __forceinline__ SfPlan sf_plan(const SfArgs& A, const MscPair& pp, const uint8_t* st)
{
    const int b = pp.ens;
    SfPlan p;
    p.fc = *reinterpret_cast<const int32_t*>(st);
    p.cu = p.fc == 5 ? 4 : p.fc;
    int nv = 0;
    for (int f = 0; f < A.n_frames; f++) nv += A.desc[(size_t)b * A.n_frames + f].valid == 1 ? 1 : 0;
    const long long c0 = 4 * A.desc[(size_t)b * A.n_frames].frame_no;
    p.n_rows = 4 * nv;
    p.r_first = 0;
    while (p.r_first < p.n_rows && c0 + p.r_first - pp.cif0 < 16) p.r_first++;
    p.avail = p.n_rows - p.r_first;
    p.nq = p.avail > 0 ? (p.cu + p.avail) / 5 : 0;
    return p;
}

// One attempt on the 5-frame copy in s_sf: Reed-Solomon over its s code words, then CheckSync (dabplus_decoder.cpp:97-213).
// Thread 0 gets the attempt's event (cif and sf_slot are the caller's); sh.sync / corr / unc are valid for all threads on return.
struct SfShared { int corr, unc, sync, au_start[8]; };
__device__ __forceinline__ void sf_attempt(uint8_t* s_sf, int fb, int s, const uint8_t* alpha_to, const uint8_t* index_of, uint8_t* s_ws,
                                           SfShared& sh, SfEvent& e, int t)
{
    const int sf_len = 5 * fb;
    if (t == 0) { sh.corr = 0; sh.unc = 0; }
    __syncthreads();
    // Syndromes, eight codewords at a time with the whole wave: S_i = XOR_j d_j alpha^(i (119 - j)) is a sum, so lane
    // (codeword c = l & 7, byte group l >> 3 = 15 positions) adds its terms without any chain of dependent look-ups and
    // three butterfly exchanges fold the eight groups (Horner's 119 dependent steps were the whole cost of this kernel).
    for (int c0 = 0; c0 < s; c0 += 8) {
        const int c = c0 + (t & 7), jg = t >> 3;
        uint32_t syn[RS_NROOTS];
#pragma unroll
        for (int i = 0; i < RS_NROOTS; i++) syn[i] = 0;
        if (c < s) {
#pragma unroll 3
            for (int jj = 0; jj < 15; jj++) {
                const int j = jg * 15 + jj;
                const uint32_t dj = s_sf[j * s + c];
                const uint32_t k = (uint32_t)(RS_LEN - 1 - j);               // exponent step: x^(119 - j) at x = alpha^i
                uint32_t ex = index_of[dj];                                  // 255 for dj = 0: the terms are masked below
                syn[0] ^= dj;
#pragma unroll
                for (int i = 1; i < RS_NROOTS; i++) {
                    ex += k; ex = ex >= RS_NN ? ex - RS_NN : ex;
                    const uint32_t av = alpha_to[ex >= RS_NN ? ex - RS_NN : ex];
                    syn[i] ^= dj ? av : 0u;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < RS_NROOTS; i++) {
            syn[i] ^= __shfl_xor(syn[i], 8); syn[i] ^= __shfl_xor(syn[i], 16); syn[i] ^= __shfl_xor(syn[i], 32);
        }
        if (jg == 0 && c < s) {
            RsIo io; io.base = s_sf + c; io.pos_stride = (size_t)s;
            const int n = rs_correct120(io, syn, alpha_to, index_of, s_ws + (t & 7) * RS_WS_BYTES);
            if (n < 0) atomicOr(&sh.unc, 1); else if (n > 0) atomicAdd(&sh.corr, n);
        }
    }
    __syncthreads();
    if (t == 0) {                                                          // CheckSync, :160-213
        const uint8_t* sf = s_sf;
        int sync = 0, num_aus = 0;
        if (!(sf[3] == 0x00 && sf[4] == 0x00) && (uint16_t)(sf[0] << 8 | sf[1]) == crc16_msb(sf + 2, 9, false, false, 0x782F)) {
            const int dac_rate = sf[2] & 0x40, sbr_flag = sf[2] & 0x20;
            num_aus = dac_rate ? (sbr_flag ? 3 : 6) : (sbr_flag ? 2 : 4);
            sh.au_start[0] = dac_rate ? (sbr_flag ? 6 : 11) : (sbr_flag ? 5 : 8);
            sh.au_start[num_aus] = sf_len / 120 * 110;
            sh.au_start[1] = sf[3] << 4 | sf[4] >> 4;
            if (num_aus >= 3) sh.au_start[2] = (sf[4] & 0x0F) << 8 | sf[5];
            if (num_aus >= 4) sh.au_start[3] = sf[6] << 4 | sf[7] >> 4;
            if (num_aus == 6) { sh.au_start[4] = (sf[7] & 0x0F) << 8 | sf[8]; sh.au_start[5] = sf[9] << 4 | sf[10] >> 4; }
            sync = 1;
            for (int i = 0; i < num_aus; i++) if (sh.au_start[i] >= sh.au_start[i + 1]) sync = 0;
        }
        sh.sync = sync;
        e = SfEvent{};
        e.corrected = sh.corr; e.uncorrectable = sh.unc; e.sync = sync; e.sf_slot = -1;
        if (sync) {
            e.format = sf[2]; e.num_aus = num_aus;
            for (int i = 0; i <= num_aus; i++) e.au_start[i] = sh.au_start[i];
        }
    }
    __syncthreads();
}

// :122-131 AU CRC-16-CCITT of the first ne events of one (ensemble, sub-channel) pair, one lane per access unit.  The verdicts do not steer the
// state machine.  (Events and superframes were written by earlier kernels or, behind a barrier, by this work-group.)  Returns nothing:
// failures are counted into *aubad (LDS).
// (Round 5 tried bringing the superframes into LDS with coalesced loads first and walking LDS bytes: 0.45-0.60 ms for the verdict kernel
// against 0.40 ms as it is -- the staging's barriers and its 12-24 KB of LDS per work-group cost more occupancy than the byte-wise HBM
// reads cost latency; with ~16 work-groups per compute unit in flight those round trips overlap.  profiles/r05_summary.md.)
__device__ __forceinline__ void sf_au_crcs(const SfArgs& A, SfEvent* ev, size_t bm, int ne, int sf_len, const uint16_t* crctab, int* aubad, int t)
{
    for (int base = 0; base < ne * 6; base += 64) {
        const int k = base + t, e_i = k / 6, au_i = k % 6;
        if (e_i < ne && ev[e_i].sync && au_i < ev[e_i].num_aus) {
            const uint8_t* au = A.sf + (bm * A.n_slots + ev[e_i].sf_slot) * sf_len + ev[e_i].au_start[au_i];
            const int au_len = ev[e_i].au_start[au_i + 1] - ev[e_i].au_start[au_i];
            if (au_len >= 2 && (uint16_t)(au[au_len - 2] << 8 | au[au_len - 1]) == crc16_msb_tab(au, au_len - 2, true, true, crctab)) atomicOr(&ev[e_i].au_crc_ok, 1 << au_i);
            else atomicAdd(aubad, 1);
        }
    }
}

// ---- The wide pass.  A receiver in lock finds a superframe every five frames: all attempts of the batch are made at once, one
// work-group per (pair, attempt q), each on the window the serial machine WOULD see if every earlier attempt of the batch
// synchronises (sf_plan).  k_superframe_settle then accepts an (ensemble, sub-channel) pair iff all its attempts did -- in that case the serial
// walk makes exactly these attempts on exactly these windows -- and does what the walk does at its end; every other (ensemble, sub-channel) pair
// is walked by k_superframe from the untouched state, as before.  The state machine's 26 dependent attempts per batch were this
// stage's whole time: 0.8 ms of a mostly idle device per step.
// Every class of a bucket (same LDS size) in ONE launch: block -> (class, its block) through the bucket's table of first blocks; the class's
// argument block comes from HBM.  A real multiplex has a handful of protection classes, a batch of independent ensembles a few dozen:
// one launch per class and kernel was 60 dependent launches of a few work-groups each at the end of every step.
__device__ __forceinline__ SfArgs sf_args_of(const SfBatch& Bt, uint32_t& bx)
{
    // (everything here is wave-uniform and read-only: through the constant address space these are scalar loads, the argument block
    // lives in scalar registers as a kernel argument would)
    const DABPHY_CONST_AS int32_t* first = as_constant(Bt.first);
    int c = 0;
    while (c + 1 < Bt.n_cls && bx >= (uint32_t)first[c + 1]) c++;
    bx -= (uint32_t)first[c];
    const DABPHY_CONST_AS SfArgs* cls = as_constant(Bt.cls);
    SfArgs a;
    __builtin_memcpy(&a, (const void DABPHY_CONST_AS*)&cls[c], sizeof a);
    return a;
}

template <int SF_MAX>
__global__ void __launch_bounds__(64) k_superframe_wide(SfBatch Bt)
{
    uint32_t bx = blockIdx.x;
    const SfArgs A = sf_args_of(Bt, bx);
    __shared__ __attribute__((aligned(16))) uint8_t s_sf[SF_MAX];
    __shared__ __attribute__((aligned(16))) uint8_t alpha_to[256], index_of[256];
    __shared__ SfShared sh;
    __shared__ uint8_t s_ws[8 * RS_WS_BYTES];
    const int t = threadIdx.x, q = (int)blockIdx.y;
    const int fb = A.frame_bytes, sf_len = 5 * fb, fw = fb >> 3;
    const size_t bm = A.run ? (size_t)A.run[bx] : (size_t)bx;      // the pair
    const uint8_t* st = A.state + bm * A.state_stride;
    const SfPlan p = sf_plan(A, A.pairs[bm], st);
    if (q >= p.nq) return;
    sf_tables(A, alpha_to, index_of, nullptr, t);
    for (int k = 0; k < 5; k++) {
        const int g = 5 * q + k;                                               // frame g of (carried frames, then rows)
        const uint8_t* src = g < p.cu ? st + 16 + (size_t)(p.fc - p.cu + g) * fb : A.out + (bm * A.n_cif + (p.r_first + g - p.cu)) * fb;
        for (int i = t; i < fw; i += 64) reinterpret_cast<uint2*>(s_sf + k * fb)[i] = reinterpret_cast<const uint2*>(src)[i];
    }
    SfEvent e;
    sf_attempt(s_sf, fb, A.s, alpha_to, index_of, s_ws, sh, e, t);
    if (t == 0) {
        e.cif = p.r_first + 5 * q + 4 - p.cu;                                  // the row whose arrival triggers the attempt
        if (sh.sync) e.sf_slot = q;
        A.events[bm * A.n_cif + q] = e;
    }
    if (sh.sync) {
        uint8_t* gsf = A.sf + (bm * A.n_slots + q) * sf_len;
        for (int i = t; i < 5 * fw; i += 64) reinterpret_cast<uint2*>(gsf)[i] = reinterpret_cast<const uint2*>(s_sf)[i];
    }
}

__global__ void __launch_bounds__(64) k_superframe_settle(SfBatch Bt)
{
    uint32_t bx = blockIdx.x;
    const SfArgs A = sf_args_of(Bt, bx);
    __shared__ uint16_t s_crctab[256];
    __shared__ int s_ok, s_corr, s_unc, s_aubad;
    const int t = threadIdx.x;
    const int fb = A.frame_bytes, sf_len = 5 * fb, fw = fb >> 3;
    const size_t bm = A.run ? (size_t)A.run[bx] : (size_t)bx;      // the pair
    const int b = A.pairs[bm].ens;
    uint8_t* st = A.state + bm * A.state_stride;
    const SfPlan p = sf_plan(A, A.pairs[bm], st);
    SfEvent* ev = A.events + bm * A.n_cif;
    for (int v = t; v < 256; v += 64) {
        uint16_t c = (uint16_t)(v << 8);
        for (int i = 0; i < 8; i++) c = (c & 0x8000) ? (uint16_t)((c << 1) ^ 0x1021) : (uint16_t)(c << 1);
        s_crctab[v] = c;
    }
    if (t == 0) { s_ok = p.nq >= 1; s_corr = 0; s_unc = 0; s_aubad = 0; }
    __syncthreads();
    for (int q = t; q < p.nq; q += 64) {
        if (!ev[q].sync) s_ok = 0;                                  // (every writer stores the same value)
        else { if (ev[q].corrected) atomicAdd(&s_corr, ev[q].corrected); if (ev[q].uncorrectable) atomicAdd(&s_unc, ev[q].uncorrectable); }
    }
    __syncthreads();
    const int ok = s_ok;
    if (t == 0) { A.accepted[bm] = ok; if (A.wide_stats && p.nq >= 1) { atomicAdd(A.wide_stats + 1, 1ull); if (ok) atomicAdd(A.wide_stats, 1ull); } }   // (a batch without a full window has nothing the wide pass could settle: not counted as tried)
    if (!ok) return;
    sf_au_crcs(A, ev, bm, p.nq, sf_len, s_crctab, &s_aubad, t);
    // the frames behind the last attempt are the next batch's carried window (a hit empties it, dabplus_decoder.cpp:156)
    const int n_left = p.cu + p.avail - 5 * p.nq;
    for (int k = 0; k < n_left; k++) {
        const uint8_t* src = A.out + (bm * A.n_cif + (p.n_rows - n_left + k)) * fb;
        for (int i = t; i < fw; i += 64) reinterpret_cast<uint2*>(st + 16 + (size_t)k * fb)[i] = reinterpret_cast<const uint2*>(src)[i];
    }
    __syncthreads();
    if (t == 0) {
        *reinterpret_cast<int32_t*>(st) = n_left; A.n_events[bm] = p.nq;
        if (A.stats) { atomicAdd(A.stats + 4 * b, p.nq); atomicAdd(A.stats + 4 * b + 1, s_corr); atomicAdd(A.stats + 4 * b + 2, s_unc); atomicAdd(A.stats + 4 * b + 3, s_aubad); }
    }
}

// ---- The serial walk: SuperframeFilter::Feed / CheckSync (dabplus_decoder.cpp:50-213) for one sub-channel of one ensemble, frame by
// frame with the reference's state machine -- 5-frame sliding window, Reed-Solomon on a copy, Fire-code / AU-table check, AU CRCs, and
// after a hit a fresh window -- carrying frame_count + the raw window to the next batch.  Runs for what the wide pass did not settle.
template <int SF_MAX>       // superframe bytes the instance can hold (120 * bitrate / 8)
__global__ void __launch_bounds__(64, SF_MAX <= 2880 ? 4 : 2) k_superframe(SfBatch Bt)      // (four waves per SIMD: at five the 960-byte build spilled ten registers)
{
    uint32_t bx = blockIdx.x;
    const SfArgs A = sf_args_of(Bt, bx);
    // LDS: the raw 5-frame window (a ring: `head` = oldest frame, nothing is ever shifted) and the working copy
    __shared__ __attribute__((aligned(16))) uint8_t s_dyn[2 * SF_MAX];
    __shared__ __attribute__((aligned(16))) uint8_t alpha_to[256], index_of[256];
    __shared__ SfShared sh;
    __shared__ int s_aubad;
    __shared__ uint8_t s_ws[8 * RS_WS_BYTES];                   // error-path workspaces of the eight code words decoded at a time
    __shared__ uint16_t s_crctab[256];                          // CRC-16-CCITT (0x1021), one byte per step: the AU checks
    const int t = threadIdx.x;
    const size_t bm = A.run ? (size_t)A.run[bx] : (size_t)bx;      // the pair
    const int b = A.pairs[bm].ens;
    if (A.accepted && A.accepted[bm]) return;                  // settled by the wide pass
    sf_tables(A, alpha_to, index_of, s_crctab, t);
    if (t == 0) s_aubad = 0;
    const int fb = A.frame_bytes, sf_len = 5 * fb;
    uint8_t* const s_raw = s_dyn;
    uint8_t* const s_sf = s_dyn + SF_MAX;
    uint8_t* st = A.state + bm * A.state_stride;
    const SfPlan p = sf_plan(A, A.pairs[bm], st);
    int frame_count = p.fc;
    __syncthreads();
    for (int i = t; i < frame_count * fb; i += 64) s_raw[i] = st[16 + i];     // carried frames, oldest first
    int head = 0;                                                              // ring slot of the oldest frame
    int ne = 0, slot = 0;
    SfEvent* ev = A.events + bm * A.n_cif;
    int tot_sync = 0, tot_corr = 0, tot_unc = 0;
    // A logical frame is fb = 24 * (bitrate / 8) bytes.  Rows travel HBM -> LDS by LDS-DMA, SF_PREFETCH rows ahead, into a ring of
    // their own (no registers, nothing waits until the row is needed).  Every row costs exactly ROW_DMA requests (lanes beyond the row
    // re-fetch its last dword into the slot's padding), so "row r has landed" is a constant vmcnt.
    constexpr int FB_MAX = SF_MAX / 5, ROW_DMA = (FB_MAX / 4 + 63) / 64, PRE_PITCH = ROW_DMA * 256;
    __shared__ __attribute__((aligned(16))) uint8_t s_pre[SF_PREFETCH * PRE_PITCH];
    const int fdw = fb >> 2;                                                   // dwords per row
    const int n_rows = p.n_rows, r_first = p.r_first;
    auto row_issue = [&](int r) {
        const uint8_t* src = A.out + (bm * A.n_cif + r) * fb;
        uint8_t* dst = s_pre + (r % SF_PREFETCH) * PRE_PITCH;
#pragma unroll
        for (int i = 0; i < ROW_DMA; i++) { int w = t + 64 * i; w = w < fdw ? w : fdw - 1; lds_dma4(src + 4 * w, dst + 256 * i); }
    };
    for (int i = 0; i < SF_PREFETCH; i++) if (r_first + i < n_rows) row_issue(r_first + i);
    for (int r = r_first; r < n_rows; r++) {
        if (r + SF_PREFETCH - 1 < n_rows) lds_dma_wait_but<(SF_PREFETCH - 1) * ROW_DMA>(); else lds_dma_wait();
        __syncthreads();
        int dst_slot;
        if (frame_count == 5) { dst_slot = head; head = head == 4 ? 0 : head + 1; }   // :78-81 "shift the previous frames": drop the oldest
        else { dst_slot = head + frame_count; if (dst_slot >= 5) dst_slot -= 5; frame_count++; }
        {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(s_pre + (r % SF_PREFETCH) * PRE_PITCH);
            uint32_t* dst = reinterpret_cast<uint32_t*>(s_raw + dst_slot * fb);
#pragma unroll
            for (int i = 0; i < ROW_DMA; i++) if (t + 64 * i < fdw) dst[t + 64 * i] = src[t + 64 * i];
        }
        lds_reads_done();
        if (r + SF_PREFETCH < n_rows) row_issue(r + SF_PREFETCH);              // into the slot just emptied
        __syncthreads();
        if (frame_count < 5) continue;
        for (int k = 0; k < 5; k++) {                                          // :97 decode on a copy, frames in age order
            int sl = head + k; if (sl >= 5) sl -= 5;
            for (int i = t; i < fdw; i += 64) reinterpret_cast<uint32_t*>(s_sf + k * fb)[i] = reinterpret_cast<const uint32_t*>(s_raw + sl * fb)[i];
        }
        SfEvent e;
        sf_attempt(s_sf, fb, A.s, alpha_to, index_of, s_ws, sh, e, t);
        if (t == 0) {
            e.cif = r;
            if (sh.sync) e.sf_slot = slot;
            ev[ne] = e;
        }
        ne++;
        tot_corr += sh.corr; tot_unc += sh.unc;
        if (sh.sync) {
            uint8_t* gsf = A.sf + (bm * A.n_slots + slot) * sf_len;             // only synchronised superframes are kept
            for (int i = t; i < 5 * fdw; i += 64) reinterpret_cast<uint32_t*>(gsf)[i] = reinterpret_cast<const uint32_t*>(s_sf)[i];
            tot_sync++;
            frame_count = 0; head = 0; if (slot + 1 < A.n_slots) slot++;       // :156 wait for a complete new superframe
        }
    }
    __syncthreads();
    sf_au_crcs(A, ev, bm, ne, sf_len, s_crctab, &s_aubad, t);
    __syncthreads();
    for (int i = t; i < frame_count * fb; i += 64) {                           // carry the window, oldest frame first
        int sl = head + i / fb; if (sl >= 5) sl -= 5;
        st[16 + i] = s_raw[sl * fb + i % fb];
    }
    if (t == 0) {
        *reinterpret_cast<int32_t*>(st) = frame_count; A.n_events[bm] = ne;
        if (A.stats) { atomicAdd(A.stats + 4 * b, tot_sync); atomicAdd(A.stats + 4 * b + 1, tot_corr); atomicAdd(A.stats + 4 * b + 2, tot_unc); atomicAdd(A.stats + 4 * b + 3, s_aubad); }
    }
}

// bucket = index of the kernels' LDS size (superframes of <= 960 / <= 2880 / <= 5760 bytes: <= 64 / 192 / 384 kbit/s); Bt.cls are the
// classes of that bucket (DEVICE array), total_blocks = Bt.first[n_cls] (known to the host), n_cif = CIFs per batch, wide = run the wide pass
void launch_superframe_bucket(const SfBatch& Bt, int bucket, int total_blocks, int n_cif, bool wide_pass, hipStream_t s)
{
    if (total_blocks <= 0) return;
    const dim3 grid(total_blocks);
    if (wide_pass) {
        // the wide pass: every attempt a locked receiver makes in this batch at once, then the verdict per (ensemble, sub-channel) pair
        const dim3 wide(grid.x, (n_cif + 4) / 5);
        if (bucket == 0) hipLaunchKernelGGL(k_superframe_wide<960>, wide, dim3(64), 0, s, Bt);
        else if (bucket == 1) hipLaunchKernelGGL(k_superframe_wide<2880>, wide, dim3(64), 0, s, Bt);
        else hipLaunchKernelGGL(k_superframe_wide<5760>, wide, dim3(64), 0, s, Bt);
        hipLaunchKernelGGL(k_superframe_settle, grid, dim3(64), 0, s, Bt);
    }
    if (bucket == 0) hipLaunchKernelGGL(k_superframe<960>, grid, dim3(64), 0, s, Bt);            // <= 64 kbit/s
    else if (bucket == 1) hipLaunchKernelGGL(k_superframe<2880>, grid, dim3(64), 0, s, Bt);     // <= 192 kbit/s
    else hipLaunchKernelGGL(k_superframe<5760>, grid, dim3(64), 0, s, Bt);                       // <= 384 kbit/s
}

void launch_rs_superframes(const RsArgs& a, hipStream_t s)
{
    const int total = a.n_sf * a.s;
    hipLaunchKernelGGL(k_rs_superframes, dim3((total + 255) / 256), dim3(256), 0, s, a);
}
void launch_rs_msc(const RsMscArgs& a, hipStream_t s)
{
    const int total = a.n_pairs * a.n_sf_per_pair * a.s;
    if (total <= 0) return;
    hipLaunchKernelGGL(k_rs_msc, dim3((total + 255) / 256), dim3(256), 0, s, a);
}

} // namespace dabphy
