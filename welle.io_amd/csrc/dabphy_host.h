// welle.io_amd/csrc/dabphy_host.h -- host-side constant tables and protection profiles (see dabphy_host.cpp).
#pragma once
#include "dabphy_common.h"
#include "../../include/dabphy.h"
#include <vector>

namespace dabphy {

constexpr int PRBS_MAX_BITS = 9216;      // 24 * 384 kbit/s

struct HostTables {
    std::vector<cf32> tw, ref, nco;
    std::vector<int16_t> perm, bin2soft;
    std::vector<int32_t> osc_unsafe;     // OSC_MAX_UNSAFE entries, -1 padded; n_osc_unsafe used (osc_exact.h)
    int n_osc_unsafe = 0;
    std::vector<uint8_t> prbs_bits;
    std::vector<uint32_t> prbs_words;    // 32 PRBS bits per word in the byte order of the decoded output
    int8_t pcodes[24][32];
};
const HostTables& host_tables();

struct TiiTables {
    uint8_t pattern[70];                 // bit (7 - b) = tii_pattern[p][b]
    std::vector<cf32> rot;               // [TII_CARRIER_ROWS][TII_NERR]
    int32_t rank[2][TII_NERR];
};
const TiiTables& tii_tables();           // built on first use (decodeTII)

int protection_fic(dabphy_protection* p);
int protection_eep(dabphy_protection* p, int bitrate, int profile_b, int level);
int protection_uep(dabphy_protection* p, int bitrate, int level);
int uep_table_entry(int table_index, int* size_cu, int* level, int* bitrate);
int protection_input_bits(const dabphy_protection* p);
bool protection_valid(const dabphy_protection* p);
std::vector<int16_t> depuncture_map(const dabphy_protection* p);   // mother-code index -> punctured index, -1 = erasure

} // namespace dabphy
