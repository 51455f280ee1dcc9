// welle.io_amd/host/seams/viterbi_seam.cpp -- INTEGRATION.md level 2: this file REPLACES src/backend/viterbi.cpp in a build of the
// reference; every other source, and the class's own header viterbi.h, stay as they are.  The seam is Viterbi::deconvolve
// (viterbi.cpp:227-245), called from FicHandler (fic-handler.cpp:197), EEPProtection (eep-protection.cpp:150) and UEPProtection
// (uep-protection.cpp:237): one depunctured code word in, one bit per byte out.
//
// The reference's decoders live on different threads -- the FIC's on the OFDM decoder's, one per selected sub-channel on DabAudio's --
// and MscHandler hands a CIF to all sub-channels at once (msc-handler.cpp:129-158), so their deconvolve calls arrive together.  All
// Viterbi objects therefore share ONE device handle (created with the first object, destroyed with the last) behind a COMBINER: a
// caller queues its code word and waits; the first caller that finds the device idle becomes the leader, takes everything queued,
// groups it by code word length and decodes each group with ONE dabphy_viterbi_batch call (SURVEY 8(b) seam 2: "batch ... all
// sub-channels of a CIF per launch"), hands the results out and wakes the others.  (Round 5 gave every object its own handle -- a
// stream, a 16 MB oscillator table and one device round trip per code word each.)
// oracle/Makefile builds it (welle-cli-l2b-*, libwelle_l2b_*); tests/test_level2_seams.py compares every callback and dump with the
// reference build's.
#include "viterbi.h"
#include "../../../include/dabphy.h"
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace {   // the class declaration is the reference's own: what the seam needs lives beside it
struct Request { const softbit_t* in; uint8_t* out; int nbits; bool done = false; std::string error; };

struct Combiner {
    std::mutex m; std::condition_variable cv;
    dabphy_handle* h = nullptr; int users = 0;
    std::vector<Request*> queue; bool busy = false;
    // statistics (dabphy_seam_viterbi_stats below): device calls, code words
    unsigned long long calls = 0, codewords = 0;

    void attach()
    {
        std::lock_guard<std::mutex> l(m);
        if (users++ == 0) {
            dabphy_config cfg = DABPHY_CONFIG_INIT;
            cfg.n_ensembles = 1; cfg.max_frames = 1; cfg.fft_placement = 2; cfg.freqsync_method = 2;
            if (dabphy_create(&cfg, &h) != DABPHY_OK) { users--; h = nullptr; throw std::runtime_error("Viterbi (GPU seam): no gfx950 device"); }
        }
    }
    void detach()
    {
        dabphy_handle* gone = nullptr;
        {
            std::unique_lock<std::mutex> l(m);
            if (--users == 0) {
                cv.wait(l, [&] { return !busy; });
                gone = h; h = nullptr;
            }
        }
        if (gone) dabphy_destroy(gone);
    }
    // decode everything in `batch` (taken off the queue by the leader; the mutex is NOT held)
    void run(std::vector<Request*>& batch)
    {
        std::map<int, std::vector<Request*>> by_len;
        for (Request* r : batch) by_len[r->nbits].push_back(r);
        for (auto& kv : by_len) {
            const int nbits = kv.first; auto& rs = kv.second;
            const size_t n_in = (size_t)4 * (nbits + 6);                 // viterbi.cpp:229: (frameBits + 6) * 4 soft bits per code word
            std::vector<softbit_t> in(n_in * rs.size());
            std::vector<uint8_t> packed((size_t)(nbits / 8) * rs.size());
            for (size_t i = 0; i < rs.size(); i++) memcpy(in.data() + n_in * i, rs[i]->in, n_in * sizeof(softbit_t));
            const int rc = dabphy_viterbi_batch(h, in.data(), (uint32_t)nbits, (uint32_t)rs.size(), packed.data());
            calls++; codewords += rs.size();
            for (size_t i = 0; i < rs.size(); i++) {
                if (rc != DABPHY_OK) { rs[i]->error = dabphy_last_error(h); continue; }
                const uint8_t* p = packed.data() + (size_t)(nbits / 8) * i;
                for (int b = 0; b < nbits; b++) rs[i]->out[b] = (p[b >> 3] >> (7 - (b & 7))) & 1;
            }
        }
    }
    void decode(Request& rq)
    {
        std::unique_lock<std::mutex> l(m);
        queue.push_back(&rq);
        while (!rq.done) {
            if (!busy) {
                // leader: everything queued so far (its own request included) in one pass over the device
                busy = true;
                std::vector<Request*> batch; batch.swap(queue);
                l.unlock();
                run(batch);
                l.lock();
                for (Request* r : batch) r->done = true;
                busy = false;
                cv.notify_all();
            } else {
                cv.wait(l);
            }
        }
    }
};
Combiner g_c;
}

// (for the bench: device calls and code words of the shared decoder since the process started)
extern "C" void dabphy_seam_viterbi_stats(unsigned long long* calls, unsigned long long* codewords)
{
    std::lock_guard<std::mutex> l(g_c.m);
    if (calls) *calls = g_c.calls;
    if (codewords) *codewords = g_c.codewords;
}

Viterbi::Viterbi(int16_t wordlength) : data(nullptr), symbols(nullptr), frameBits(wordlength)
{
    vp.decisions = nullptr;
    g_c.attach();
}

Viterbi::~Viterbi()
{
    g_c.detach();
}

void Viterbi::deconvolve(softbit_t* input, uint8_t* output)
{
    Request rq; rq.in = input; rq.out = output; rq.nbits = frameBits;
    g_c.decode(rq);
    if (!rq.error.empty()) throw std::runtime_error(rq.error);
}
