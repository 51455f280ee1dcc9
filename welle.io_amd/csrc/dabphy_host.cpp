// welle.io_amd/csrc/dabphy_host.cpp -- host-side constant tables and protection profiles.
//
// Everything here is computed once per process with the HOST C library, in the reference's own expression
// forms, so that the uploaded constants are the bits the reference's constructors produce:
//   KISS twiddles          libs/kiss_fft/kiss_fft.c:353-364   (double cos/sin of -2*pi*i/N, rounded to float)
//   oscillator table       backend/ofdm-processor.cpp:92-94   (double cos/sin of 2*pi*i/2048000, rounded to float)
//   PRS reference table    backend/phasereference.cpp:45-51, phasetable.cpp:24-75,140-183 (float cos/sin of a float phase)
//   frequency interleaver  backend/freq-interleaver.cpp:35-59
//   puncturing vectors     backend/protTables.cpp:25-51        (EN 300 401 table 29)
//   EEP / UEP profiles     backend/eep-protection.cpp:32-113, uep-protection.cpp:27-118, dab-constants.cpp:45-109
//   PRBS                   backend/fic-handler.cpp:62-71, energy_dispersal.h:39-49
#include "dabphy_host.h"
#include "osc_exact.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <complex>
#include <unordered_map>

namespace dabphy {

// EN 300 401 clause 14.3.2, Mode I: 48 blocks of 32 carriers from k = -768 upward; (row i of h, offset n)
static const int8_t kPrsRow[48] = {0,1,2,3,0,1,2,3,0,1,2,3,0,1,2,3,0,1,2,3,0,1,2,3, 0,3,2,1,0,3,2,1,0,3,2,1,0,3,2,1,0,3,2,1,0,3,2,1};
static const int8_t kPrsOff[48] = {1,2,0,1,3,2,2,3,2,1,2,3,1,2,3,3,2,2,2,1,1,3,1,2, 3,1,1,1,2,2,1,0,2,2,3,3,0,2,1,3,3,3,3,0,3,0,1,1};
static const int8_t kPrsH[4][16] = {
    {0,2,0,0,0,0,1,1,2,0,0,0,2,2,1,1}, {0,3,2,3,0,1,3,0,2,1,2,3,2,3,3,0},
    {0,0,0,2,0,2,1,3,2,2,0,2,2,0,1,3}, {0,1,2,1,0,3,3,2,2,3,2,1,2,1,3,2}};

static float prs_phase(int k)   // PhaseTable::get_Phi: double product M_PI / 2.0f * (h + n), returned as float
{
    const int blk = k < 0 ? (k + 768) / 32 : 24 + (k - 1) / 32;
    const int kmin = k < 0 ? -768 + 32 * blk : 1 + 32 * (blk - 24);
    const int hv = kPrsH[kPrsRow[blk]][(k - kmin) & 15];
    return (float)(M_PI / 2.0f * (hv + kPrsOff[blk]));
}

static void build(HostTables& T)
{
    T.tw.resize(T_U);
    for (int i = 0; i < T_U; i++) {
        const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
        const double phase = -2 * pi * i / T_U;
        T.tw[i].re = (float)cos(phase);
        T.tw[i].im = (float)sin(phase);
    }
    // pk_cmul_unit (k_demod's first two passes) counts on tw[0] = (1, +-0): cos(-0.0) and sin(-0.0) of any libm
    if (T.tw[0].re != 1.0f || T.tw[0].im != 0.0f) { fprintf(stderr, "dabphy: this libm's cos(-0.0) / sin(-0.0) are not 1 / -0: the twiddle table cannot be used\n"); abort(); }
    T.nco.resize(INPUT_RATE);
    for (int i = 0; i < INPUT_RATE; i++) {
        T.nco[i].re = (float)cos(2.0 * M_PI * i / INPUT_RATE);
        T.nco[i].im = (float)sin(2.0 * M_PI * i / INPUT_RATE);
    }
    T.osc_unsafe.assign(OSC_MAX_UNSAFE, -1);
    T.n_osc_unsafe = osc_unsafe_list(T.osc_unsafe.data());
    if (T.n_osc_unsafe < 0) { fprintf(stderr, "dabphy: this libm's oscillator table has more than %d entries next to a float rounding boundary\n", OSC_MAX_UNSAFE); abort(); }
    T.ref.assign(T_U, cf32{0.f, 0.f});
    for (int i = 1; i <= K_CARR / 2; i++) {
        float phi = prs_phase(i);
        T.ref[i] = cf32{cosf(phi), sinf(phi)};
        phi = prs_phase(-i);
        T.ref[T_U - i] = cf32{cosf(phi), sinf(phi)};
    }
    // frequency interleaver: LCG over 0..2047 keeping 256 <= v <= 1792, v != 1024, in generation order
    T.perm.clear();
    {
        int v = 0;
        for (int i = 0; i < T_U; i++) {
            if (i > 0) v = (13 * v + 511) % T_U;
            if (v == T_U / 2 || v < 256 || v > 256 + K_CARR) continue;
            T.perm.push_back((int16_t)(v - T_U / 2));
        }
    }
    T.bin2soft.assign(T_U, -1);
    for (int i = 0; i < K_CARR; i++) {
        int bin = T.perm[i]; if (bin < 0) bin += T_U;
        T.bin2soft[bin] = (int16_t)i;
    }
    // puncturing vectors: 8+p ones; column 0 of every group always set, then columns 1..3 filled group by
    // group in the order 0,4,2,6,1,5,3,7
    static const int order[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    for (int p = 1; p <= 24; p++) {
        int8_t* v = T.pcodes[p - 1];
        memset(v, 0, 32);
        for (int g = 0; g < 8; g++) v[4 * g] = 1;
        for (int r = 0; r < p; r++) v[4 * order[r % 8] + 1 + r / 8] = 1;
    }
    // PRBS x^9 + x^5 + 1, all-ones preset
    T.prbs_bits.resize(PRBS_MAX_BITS);
    {
        uint8_t sr[9]; memset(sr, 1, 9);
        for (int i = 0; i < PRBS_MAX_BITS; i++) {
            const uint8_t b = sr[8] ^ sr[4];
            for (int j = 8; j > 0; j--) sr[j] = sr[j - 1];
            sr[0] = b; T.prbs_bits[i] = b;
        }
    }
    T.prbs_words.assign(PRBS_MAX_BITS / 32, 0);
    for (int i = 0; i < PRBS_MAX_BITS; i++)
        if (T.prbs_bits[i]) T.prbs_words[i >> 5] |= 1u << (8 * ((i >> 3) & 3) + 7 - (i & 7));
}

const HostTables& host_tables()
{
    static HostTables T;
    static std::once_flag once;
    std::call_once(once, [] { build(T); });
    return T;
}

// ------------------------------------------------------------------------------------------ TII
// Constants of TIIDecoder (tii-decoder.cpp): the 70 patterns, the rotators analyse_phase recomputes for every carrier and
// candidate delay, and the order in which its unordered_map<float, uint64_t> presents the candidates to std::min_element.
const TiiTables& tii_tables()
{
    static TiiTables T;
    static std::once_flag once;
    std::call_once(once, [] {
        int n = 0;
        for (int v = 0; v < 256; v++) if (__builtin_popcount(v) == 4) T.pattern[n++] = (uint8_t)v;   // tii-decoder.cpp:29-99, b = 0 is the MSB
        // tii-decoder.cpp:355-356: polar(1.0f, 2.0f * pi * err * carriers[j] / 2048.0f), same operand types, same libm
        T.rot.resize((size_t)TII_CARRIER_ROWS * TII_NERR);
        constexpr float pi = M_PI;
        for (int k = -768; k <= 768; k++)
            for (int err = -4; err < 500; err++) {
                const std::complex<float> r = std::polar(1.0f, 2.0f * pi * err * k / 2048.0f);
                cf32 c; c.re = r.real(); c.im = r.imag();
                T.rot[(size_t)(k + 768) * TII_NERR + (err + 4)] = c;
            }
        // tii-decoder.cpp:360-366: min_element keeps the first minimum in the container's iteration order, a property of the
        // C++ library the receiver is built with -- so ask the same container: filled for the first time, and refilled after clear()
        std::unordered_map<float, uint64_t> m;
        for (int cycle = 0; cycle < 2; cycle++) {
            for (int err = -4; err < 500; err++) m[err] += 0.0f;
            int pos = 0;
            for (const auto& kv : m) T.rank[cycle][(int)kv.first + 4] = pos++;
            m.clear();
        }
    });
    return T;
}

// ------------------------------------------------------------------------------------------ protection

int protection_fic(dabphy_protection* p)
{
    memset(p, 0, sizeof *p);
    p->nbits = 768; p->L[0] = 21; p->PI[0] = 16; p->L[1] = 3; p->PI[1] = 15;
    return 0;
}

int protection_eep(dabphy_protection* p, int bitrate, int profile_b, int level)
{
    memset(p, 0, sizeof *p);
    if (bitrate <= 0 || level < 1 || level > 4) return -1;
    p->nbits = 24 * bitrate;
    const int n = bitrate / 8;
    if (!profile_b) {
        switch (level) {
        case 1: p->L[0] = 6 * n - 3; p->L[1] = 3; p->PI[0] = 24; p->PI[1] = 23; break;
        case 2:
            if (bitrate == 8) { p->L[0] = 5; p->L[1] = 1; p->PI[0] = 13; p->PI[1] = 12; }
            else { p->L[0] = 2 * n - 3; p->L[1] = 4 * n + 3; p->PI[0] = 14; p->PI[1] = 13; }
            break;
        case 3: p->L[0] = 6 * n - 3; p->L[1] = 3; p->PI[0] = 8; p->PI[1] = 7; break;
        case 4: p->L[0] = 4 * n - 3; p->L[1] = 2 * n + 3; p->PI[0] = 3; p->PI[1] = 2; break;
        }
    } else {
        static const int pi_b[5][2] = {{0, 0}, {10, 9}, {6, 5}, {4, 3}, {2, 1}};
        p->L[0] = 24 * bitrate / 32 - 3; p->L[1] = 3;
        p->PI[0] = pi_b[level][0]; p->PI[1] = pi_b[level][1];
    }
    return 0;
}

// Short-form (UEP) profiles: bitrate, protection level, sub-channel size in CU, L1..L4, PI1..PI4.  Values are
// the reference's (uep-protection.cpp:38-117 merged with dab-constants.cpp:45-109), including its entry for
// 80 kbit/s level 1 (PI2 = 7), because parity is defined against the reference, not against the standard.
static const int16_t kUep[64][11] = {
    {32,5,16,3,4,17,0,5,3,2,0}, {32,4,21,3,3,18,0,11,6,5,0}, {32,3,24,3,4,14,3,15,9,6,8}, {32,2,29,3,4,14,3,22,13,8,13},
    {32,1,35,3,5,13,3,24,17,12,17}, {48,5,24,4,3,26,3,5,4,2,3}, {48,4,29,3,4,26,3,9,6,4,6}, {48,3,35,3,4,26,3,15,10,6,9},
    {48,2,42,3,4,26,3,24,14,8,15}, {48,1,52,3,5,25,3,24,18,13,18}, {56,5,29,6,10,23,3,5,4,2,3}, {56,4,35,6,10,23,3,9,6,4,5},
    {56,3,42,6,12,21,3,16,7,6,9}, {56,2,52,6,10,23,3,23,13,8,13}, {64,5,32,6,9,31,2,5,3,2,3}, {64,4,42,6,9,33,0,11,6,5,0},
    {64,3,48,6,12,27,3,16,8,6,9}, {64,2,58,6,10,29,3,23,13,8,13}, {64,1,70,6,11,28,3,24,18,12,18}, {80,5,40,6,10,41,3,6,3,2,3},
    {80,4,52,6,10,41,3,11,6,5,6}, {80,3,58,6,11,40,3,16,8,6,7}, {80,2,70,6,10,41,3,23,13,8,13}, {80,1,84,6,10,41,3,24,7,12,18},
    {96,5,48,7,9,53,3,5,4,2,4}, {96,4,58,7,10,52,3,9,6,4,6}, {96,3,70,6,12,51,3,16,9,6,10}, {96,2,84,6,10,53,3,22,12,9,12},
    {96,1,104,6,13,50,3,24,18,13,19}, {112,5,58,14,17,50,3,5,4,2,5}, {112,4,70,11,21,49,3,9,6,4,8}, {112,3,84,11,23,47,3,16,8,6,9},
    {112,2,104,11,21,49,3,23,12,9,14}, {128,5,64,12,19,62,3,5,3,2,4}, {128,4,84,11,21,61,3,11,6,5,7}, {128,3,96,11,22,60,3,16,9,6,10},
    {128,2,116,11,21,61,3,22,12,9,14}, {128,1,140,11,20,62,3,24,17,13,19}, {160,5,80,11,19,87,3,5,4,2,4}, {160,4,104,11,23,83,3,11,6,5,9},
    {160,3,116,11,24,82,3,16,8,6,11}, {160,2,140,11,21,85,3,22,11,9,13}, {160,1,168,11,22,84,3,24,18,12,19}, {192,5,96,11,20,110,3,6,4,2,5},
    {192,4,116,11,22,108,3,10,6,4,9}, {192,3,140,11,24,106,3,16,10,6,11}, {192,2,168,11,20,110,3,22,13,9,13}, {192,1,208,11,21,109,3,24,20,13,24},
    {224,5,116,12,22,131,3,8,6,2,6}, {224,4,140,12,26,127,3,12,8,4,11}, {224,3,168,11,20,134,3,16,10,7,9}, {224,2,208,11,22,132,3,24,16,10,15},
    {224,1,232,11,24,130,3,24,20,12,20}, {256,5,128,11,24,154,3,6,5,2,5}, {256,4,168,11,24,154,3,12,9,5,10}, {256,3,192,11,27,151,3,16,10,7,10},
    {256,2,232,11,22,156,3,24,14,10,13}, {256,1,280,11,26,152,3,24,19,14,18}, {320,5,160,11,26,200,3,8,5,2,6}, {320,4,208,11,25,201,3,13,9,5,10},
    {320,2,280,11,26,200,3,24,17,9,17}, {384,5,192,11,27,247,3,8,6,2,7}, {384,3,280,11,24,250,3,16,9,7,10}, {384,1,416,12,28,245,3,24,20,14,23}};

int protection_uep(dabphy_protection* p, int bitrate, int level)
{
    memset(p, 0, sizeof *p);
    p->nbits = 24 * bitrate;
    int idx = -1;
    for (int i = 0; i < 64; i++) if (kUep[i][0] == bitrate && kUep[i][1] == level) { idx = i; break; }
    if (idx < 0) idx = 1;     // the reference falls back to row 1 (uep-protection.cpp:142-146)
    for (int s = 0; s < 4; s++) { p->L[s] = kUep[idx][3 + s]; p->PI[s] = kUep[idx][7 + s]; }
    return 0;
}

int uep_table_entry(int table_index, int* size_cu, int* level, int* bitrate)
{
    if (table_index < 0 || table_index >= 64) return -1;
    *bitrate = kUep[table_index][0]; *level = kUep[table_index][1]; *size_cu = kUep[table_index][2];
    return 0;
}

int protection_input_bits(const dabphy_protection* p)
{
    int n = 12;
    for (int s = 0; s < 4; s++) if (p->L[s] > 0 && p->PI[s] > 0) n += p->L[s] * 4 * (8 + p->PI[s]);
    return n;
}

bool protection_valid(const dabphy_protection* p)
{
    if (p->nbits <= 0 || p->nbits % 32) return false;
    int blocks = 0;
    for (int s = 0; s < 4; s++) {
        if (p->L[s] < 0 || p->PI[s] < 0 || p->PI[s] > 24) return false;
        if (p->L[s] > 0 && p->PI[s] == 0) return false;
        blocks += p->L[s];
    }
    return blocks * 128 == 4 * p->nbits;
}

std::vector<int16_t> depuncture_map(const dabphy_protection* p)
{
    const HostTables& T = host_tables();
    std::vector<int16_t> m((size_t)4 * p->nbits + 24, -1);
    int in = 0, v = 0;
    for (int s = 0; s < 4; s++)
        for (int i = 0; i < p->L[s]; i++)
            for (int j = 0; j < 128; j++) { if (T.pcodes[p->PI[s] - 1][j % 32]) m[v] = (int16_t)in++; v++; }
    for (int i = 0; i < 24; i++) { if ((i & 3) < 2) m[v] = (int16_t)in++; v++; }     // PI_X = 1100 x 6 (fic-handler.cpp:39-42)
    return m;
}

} // namespace dabphy
