// welle.io_amd/host/seams/viterbi_seam.cpp -- INTEGRATION.md level 2: this file REPLACES src/backend/viterbi.cpp in a build of the
// reference; every other source, and the class's own header viterbi.h, stay as they are.  The seam is Viterbi::deconvolve
// (viterbi.cpp:227-245), called from FicHandler (fic-handler.cpp:197), EEPProtection (eep-protection.cpp:150) and UEPProtection
// (uep-protection.cpp:237): one depunctured code word in, one bit per byte out; dabphy_viterbi_batch decodes it on the device.
// Every Viterbi object has its own handle (the reference's objects live on different threads -- the FIC on the OFDM decoder's, one
// per sub-channel on DabAudio's -- and a handle is not thread-safe).  oracle/Makefile builds it (welle-cli-l2b-*, libwelle_l2b_*);
// tests/test_level2_seams.py compares every callback and dump with the reference build's.
#include "viterbi.h"
#include "../../../include/dabphy.h"
#include <map>
#include <mutex>
#include <stdexcept>
#include <vector>

namespace {   // the class declaration is the reference's own: the device handle of each decoder lives beside it
std::mutex g_m; std::map<const Viterbi*, dabphy_handle*> g_h;
}

Viterbi::Viterbi(int16_t wordlength) : data(nullptr), symbols(nullptr), frameBits(wordlength)
{
    vp.decisions = nullptr;
    dabphy_config cfg = DABPHY_CONFIG_INIT;
    cfg.n_ensembles = 1; cfg.max_frames = 1; cfg.fft_placement = 2; cfg.freqsync_method = 2;
    dabphy_handle* h = nullptr;
    if (dabphy_create(&cfg, &h) != DABPHY_OK) throw std::runtime_error("Viterbi (GPU seam): no gfx950 device");
    std::lock_guard<std::mutex> l(g_m); g_h[this] = h;
}

Viterbi::~Viterbi()
{
    dabphy_handle* h;
    { std::lock_guard<std::mutex> l(g_m); h = g_h.at(this); g_h.erase(this); }
    dabphy_destroy(h);
}

void Viterbi::deconvolve(softbit_t* input, uint8_t* output)
{
    dabphy_handle* h;
    { std::lock_guard<std::mutex> l(g_m); h = g_h.at(this); }
    std::vector<uint8_t> packed(frameBits / 8);
    if (dabphy_viterbi_batch(h, input, (uint32_t)frameBits, 1, packed.data()) != DABPHY_OK) throw std::runtime_error(dabphy_last_error(h));
    for (int i = 0; i < frameBits; i++) output[i] = (packed[i >> 3] >> (7 - (i & 7))) & 1;
}
