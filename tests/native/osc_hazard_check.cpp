// TEST INFRASTRUCTURE: osc_hazard_entry (welle.io_amd/csrc/osc_exact.h) against brute force: for random and adversarial frames, the
// symbols it marks must be exactly those whose useful part reads an unsafe oscillator table entry.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <set>
#include "osc_exact.h"
using namespace dabphy;
int main() {
    int32_t U[OSC_MAX_UNSAFE]; int n = osc_unsafe_list(U);
    std::set<int> us(U, U + n);
    unsigned long long x = 88172645463325252ull; auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    long miss = 0, extra = 0, flagged = 0, total = 0;
    for (int it = 0; it < 3000; it++) {
        int32_t f_prs, f_sym;
        int kind = it % 6;
        if (kind == 0) { f_prs = f_sym = 0; }
        else if (kind == 1) { f_prs = (int)(rnd() % 201) - 100; f_sym = f_prs + (int)(rnd() % 5) - 2; }
        else if (kind == 2) { f_prs = 1000 * ((int)(rnd() % 71) - 35); f_sym = f_prs; }
        else if (kind == 3) { f_prs = 500 * ((int)(rnd() % 141) - 70); f_sym = f_prs + 1000; }
        else { f_prs = (int)(rnd() % 70001) - 35000; f_sym = (int)(rnd() % 70001) - 35000; }
        int32_t L0 = (int32_t)(rnd() % INPUT_RATE), start = (int32_t)(rnd() % 2048);
        if (kind == 0 || kind == 2 || kind == 3) { if (rnd() & 1) L0 = U[rnd() % n]; if (rnd() & 1) L0 = (L0 / 500) * 500; }
        const int32_t J0 = start + T_U;
        auto modr = [](int64_t v) { int64_t r = v % INPUT_RATE; if (r < 0) r += INPUT_RATE; return (int32_t)r; };
        const int32_t L1 = modr((int64_t)L0 - (int64_t)J0 * f_prs);
        uint32_t mask[3] = {0, 0, 0};
        for (int i = 0; i < n; i++) osc_hazard_entry(mask, U[i], start, L0, f_prs, L1, f_sym);
        for (int s = 0; s < 76; s++) {
            bool hit = false;
            for (int k = 0; k < T_U && !hit; k++) {
                int32_t ph;
                if (s == 0) ph = modr((int64_t)L0 - (int64_t)(start + k + 1) * f_prs);
                else ph = modr((int64_t)L1 - ((int64_t)(s - 1) * T_S + T_G + k + 1) * (int64_t)f_sym);
                if (us.count(ph)) hit = true;
            }
            const bool fl = (mask[s >> 5] >> (s & 31)) & 1;
            total++; flagged += fl;
            if (hit && !fl) miss++;
            if (!hit && fl) extra++;
        }
    }
    printf("{\"unsafe\": %d, \"symbols\": %ld, \"flagged\": %ld, \"missed\": %ld, \"extra\": %ld}\n", n, total, flagged, miss, extra);
    return miss != 0;
}
