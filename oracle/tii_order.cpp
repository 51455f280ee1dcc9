// oracle/tii_order.cpp -- TEST INFRASTRUCTURE.  The reference keeps the TII error sums in a std::unordered_map<float, uint64_t>
// (tii-decoder.h:97) and picks the winner with std::min_element over it (tii-decoder.cpp:360-366): among equal sums the one the
// container iterates first wins.  That order belongs to the C++ library the reference is built with, so the restatement asks the
// same container instead of imitating its hash and rehash policy.
#include <cstdint>
#include <unordered_map>

extern "C" void orc_tii_iteration_rank(int32_t* rank /* [2][504] */)
{
    std::unordered_map<float, uint64_t> m;
    for (int cycle = 0; cycle < 2; cycle++) {
        for (int err = -4; err < 500; err++) m[err] += 0.0f;     // same key conversion and insertion order as analyse_phase
        int pos = 0;
        for (const auto& kv : m) rank[504 * cycle + (int)kv.first + 4] = pos++;
        m.clear();
    }
}
