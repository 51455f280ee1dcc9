// oracle/ref_ofdm_harness.cpp -- TEST INFRASTRUCTURE ONLY.
// Drives the reference's own OfdmDecoder (src/backend/ofdm-decoder.cpp:93-230, unmodified) with the
// recording FicHandler/MscHandler stand-ins of oracle/shim/, to tap every soft bit, the 1200
// constellation points and the SNR callback of one or more frames.  Built into its own shared object
// (oracle/_ref/libwelle_ref_ofdm.so, -Bsymbolic) so the stand-in classes never meet the real ones.
#include <vector>
#include <atomic>
#include <thread>
#include <chrono>
#include <cstring>
#include "ofdm-decoder.h"

namespace {
struct Rec : RadioControllerInterface {
    float* con = nullptr; std::atomic<int> n_con{0}; float* snr = nullptr; int snr_cap = 0; std::atomic<int> n_snr{0};
    void onSNR(float s) override { int k = n_snr++; if (k < snr_cap) snr[k] = s; }
    void onFrequencyCorrectorChange(int, int) override {}
    void onSyncChange(char) override {}
    void onSignalPresence(bool) override {}
    void onServiceDetected(uint32_t) override {}
    void onNewEnsemble(uint16_t) override {}
    void onSetEnsembleLabel(DabLabel&) override {}
    void onDateTimeUpdate(const dab_date_time_t&) override {}
    void onFIBDecodeSuccess(bool, const uint8_t*) override {}
    void onNewImpulseResponse(std::vector<float>&&) override {}
    void onConstellationPoints(std::vector<DSPCOMPLEX>&& d) override {
        if (con && d.size() == 1200) memcpy(con + 2400 * (size_t)n_con.load(), d.data(), 2400 * 4);
        n_con++;
    }
    void onNewNullSymbol(std::vector<DSPCOMPLEX>&&) override {}
    void onTIIMeasurement(tii_measurement_t&&) override {}
    void onMessage(message_level_t, const std::string&, const std::string&) override {}
};
}

extern "C" {
// frames: n_frames x (2048 + 75*2552) cf32 laid out as [PRS useful part (T_u)][75 symbols of T_s incl. guard]
// soft: n_frames x 75 x 3072 int8; con: n_frames x 1200 cf32; snr: up to snr_cap values (one per 11 frames)
int ref_ofdm_decode_frames(const float* frames, int n_frames, int8_t* soft, float* con, float* snr, int snr_cap)
{
    DABParams p(1);
    Rec rec; rec.con = con; rec.snr = snr; rec.snr_cap = snr_cap;
    FicHandler fic; MscHandler msc; SoftbitTap tap; fic.tap = &tap; msc.tap = &tap;
    {
        OfdmDecoder dec(p, rec, fic, msc);
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        const size_t per = 2048 + 75 * 2552;
        for (int f = 0; f < n_frames; f++) {
            tap.dst = soft + (size_t)f * 75 * 3072;
            const DSPCOMPLEX* src = (const DSPCOMPLEX*)frames + per * f;
            std::vector<std::vector<DSPCOMPLEX>> syms(76);
            syms[0].assign(src, src + 2048);
            for (int s = 1; s < 76; s++) syms[s].assign(src + 2048 + 2552 * (s - 1), src + 2048 + 2552 * s);
            dec.pushAllSymbols(std::move(syms));
            while (rec.n_con.load() <= f) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    return rec.n_snr;
}
}
