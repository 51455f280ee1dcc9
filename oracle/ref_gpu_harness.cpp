// oracle/ref_gpu_harness.cpp -- TEST INFRASTRUCTURE: drives welle.io_amd/host/GpuRadioReceiver (our drop-in for the
// reference's RadioReceiver facade) through the SAME public surface and the same recording callbacks as
// oracle/ref_harness.cpp drives the reference's RadioReceiver, so the two runs can be compared callback by callback.
// Linked against the reference's unmodified FIBProcessor / DecoderAdapter objects and either the CPU execution model
// of the kernels (libdabphy_emu.so) or the real libdabphy_hip.so.
#include <cstring>
#include <atomic>
#include <thread>
#include <chrono>
#include <vector>
#include "gpu_radio_receiver.h"
#include "gpu_batch_receiver.h"
#include "gpu_node_receiver.h"
#include "../include/dabphy.h"

namespace {
struct NullProgrammeHandler : ProgrammeHandlerInterface {
    std::atomic<int> rs_calls{0}, rs_uncorr{0}, rs_corr{0};
    void onFrameErrors(int) override {}
    void onNewAudio(std::vector<int16_t>&&, int, const std::string&) override {}
    void onRsErrors(bool u, int n) override { rs_calls++; if (u) rs_uncorr++; rs_corr += n; }
    void onAacErrors(int) override {}
    void onNewDynamicLabel(const std::string&) override {}
    void onMOT(const mot_file_t&) override {}
    void onPADLengthError(size_t, size_t) override {}
};

struct Rec : RadioControllerInterface {
    uint8_t* fib = nullptr; int fib_cap = 0; std::atomic<int> n_fib{0};
    float* cir = nullptr; int cir_cap = 0; std::atomic<int> n_cir{0};
    float* con = nullptr; int con_cap = 0; std::atomic<int> n_con{0};
    float* snr = nullptr; int snr_cap = 0; std::atomic<int> n_snr{0};
    int32_t* corr = nullptr; int corr_cap = 0; std::atomic<int> n_corr{0};
    float* nul = nullptr; int nul_cap = 0; std::atomic<int> n_nul{0};
    std::atomic<int> n_sync_true{0}, n_sync_false{0}, n_services{0};
    std::atomic<bool> failed{false};
    void onSNR(float s) override { int k = n_snr++; if (k < snr_cap) snr[k] = s; }
    void onFrequencyCorrectorChange(int f, int c) override { int k = n_corr++; if (k < corr_cap) { corr[2 * k] = f; corr[2 * k + 1] = c; } }
    void onSyncChange(char s) override { if (s) n_sync_true++; else n_sync_false++; }
    void onSignalPresence(bool) override {}
    void onServiceDetected(uint32_t) override { n_services++; }
    void onNewEnsemble(uint16_t) override {}
    void onSetEnsembleLabel(DabLabel&) override {}
    void onDateTimeUpdate(const dab_date_time_t&) override {}
    void onFIBDecodeSuccess(bool ok, const uint8_t* bits) override {
        int k = n_fib++;
        if (k < fib_cap) {
            uint8_t* o = fib + 33 * (size_t)k; o[0] = ok;
            for (int i = 0; i < 32; i++) { uint8_t b = 0; for (int j = 0; j < 8; j++) b = (b << 1) | (bits[8 * i + j] & 1); o[1 + i] = b; }
        }
    }
    void onNewImpulseResponse(std::vector<float>&& d) override { int k = n_cir++; if (k < cir_cap && d.size() == 2048) memcpy(cir + 2048 * (size_t)k, d.data(), 8192); }
    void onConstellationPoints(std::vector<DSPCOMPLEX>&& d) override { int k = n_con++; if (k < con_cap && d.size() == 1200) memcpy(con + 2400 * (size_t)k, d.data(), 9600); }
    void onNewNullSymbol(std::vector<DSPCOMPLEX>&& d) override { int k = n_nul++; if (k < nul_cap && d.size() == 2656) memcpy(nul + 5312 * (size_t)k, d.data(), 5312 * 4); }
    struct TiiEv { int32_t frame, comb, pattern, delay_samples; float error; };
    TiiEv* tii = nullptr; int tii_cap = 0; std::atomic<int> n_tii{0};
    void onTIIMeasurement(tii_measurement_t&& m) override {
        int k = n_tii++;
        if (k < tii_cap) { tii[k].frame = n_nul.load() - 1; tii[k].comb = m.comb; tii[k].pattern = m.pattern; tii[k].delay_samples = m.delay_samples; tii[k].error = m.error; }
    }
    void onMessage(message_level_t, const std::string&, const std::string&) override {}
    void onInputFailure() override { failed = true; }
};

struct MemInput : InputInterface {
    const DSPCOMPLEX* data; int64_t n; int64_t pos = 0;
    MemInput(const float* iq, int64_t n_) : data((const DSPCOMPLEX*)iq), n(n_) {}
    void setFrequency(int) override {}
    int getFrequency() const override { return 0; }
    bool is_ok() override { return pos < n; }
    bool restart() override { pos = 0; return true; }
    void stop() override {}
    void reset() override {}
    int32_t getSamples(DSPCOMPLEX* b, int32_t size) override { int64_t k = size; if (k > n - pos) k = n - pos; memcpy(b, data + pos, k * sizeof(DSPCOMPLEX)); pos += k; return (int32_t)k; }
    std::vector<DSPCOMPLEX> getSpectrumSamples(int) override { return {}; }
    int32_t getSamplesToRead() override { int64_t k = n - pos; return (int32_t)(k > (1 << 20) ? (1 << 20) : k); }
    float setGain(int) override { return 0; }
    float getGain() const override { return 0; }
    int getGainCount() override { return 0; }
    void setAgc(bool) override {}
    std::string getDescription() override { return "mem"; }
};
}

extern "C" {
struct gpu_subch { int32_t subChId, startAddr, length, shortForm, uepTableIndex, uepLevel, eepProfileB, eepLevel, dabplus; char dump_path[256]; };
struct gpu_tii_event { int32_t frame, comb, pattern, delay_samples; float error; };
struct gpu_run_io {
    const float* iq; int64_t n_samples; int32_t disable_coarse, fft_placement;
    int32_t n_subch; const gpu_subch* subch;
    uint8_t* fib; int32_t fib_cap; float* cir; int32_t cir_cap; float* con; int32_t con_cap; float* snr; int32_t snr_cap; int32_t* corr; int32_t corr_cap;
    int32_t n_fib, n_cir, n_con, n_snr, n_corr, n_sync_true, n_sync_false, n_services;
    int32_t rs_calls[16], rs_uncorr[16], rs_corr[16];
    float* nul; int32_t nul_cap, n_nul;
    int32_t freqsync;                 // FreqsyncMethod (reference numbering)
    int32_t decode_tii; gpu_tii_event* tii; int32_t tii_cap, n_tii;   // RadioReceiverOptions::decodeTII and the onTIIMeasurement log
};

int gpu_receiver_run(gpu_run_io* io)
{
    Rec rec;
    rec.fib = io->fib; rec.fib_cap = io->fib_cap; rec.cir = io->cir; rec.cir_cap = io->cir_cap; rec.con = io->con; rec.con_cap = io->con_cap;
    rec.snr = io->snr; rec.snr_cap = io->snr_cap; rec.corr = io->corr; rec.corr_cap = io->corr_cap; rec.nul = io->nul; rec.nul_cap = io->nul_cap;
    MemInput in(io->iq, io->n_samples);
    RadioReceiverOptions rro;
    rro.decodeTII = io->decode_tii != 0; rro.disableCoarseCorrector = io->disable_coarse != 0;
    rec.tii = reinterpret_cast<Rec::TiiEv*>(io->tii); rec.tii_cap = io->tii_cap;
    rro.freqsyncMethod = io->freqsync == 0 ? FreqsyncMethod::GetMiddle : io->freqsync == 1 ? FreqsyncMethod::CorrelatePRS : FreqsyncMethod::PatternOfZeros;
    rro.fftPlacementMethod = io->fft_placement == 0 ? FFTPlacementMethod::StrongestPeak
                           : io->fft_placement == 1 ? FFTPlacementMethod::EarliestPeakWithBinning : FFTPlacementMethod::ThresholdBeforePeak;
    std::vector<NullProgrammeHandler> handlers(io->n_subch > 0 ? io->n_subch : 1);
    try {
        GpuRadioReceiver rx(rec, in, rro);
        rx.restart(false);
        for (int i = 0; i < io->n_subch; i++) {
            const gpu_subch& s = io->subch[i];
            Subchannel sub;
            sub.subChId = s.subChId; sub.startAddr = s.startAddr; sub.length = s.length;
            sub.protectionSettings.shortForm = s.shortForm != 0; sub.protectionSettings.uepTableIndex = s.uepTableIndex; sub.protectionSettings.uepLevel = s.uepLevel;
            sub.protectionSettings.eepProfile = s.eepProfileB ? EEPProtectionProfile::EEP_B : EEPProtectionProfile::EEP_A;
            sub.protectionSettings.eepLevel = (EEPProtectionLevel)s.eepLevel;
            rx.addSubchannel(handlers[i], s.dabplus ? AudioServiceComponentType::DABPlus : AudioServiceComponentType::DAB, std::string(s.dump_path), sub);
        }
        while (!rec.failed) std::this_thread::sleep_for(std::chrono::milliseconds(2));
        rx.stop();
    } catch (const std::exception& e) {
        fprintf(stderr, "gpu_receiver_run: %s\n", e.what());
        return -1;
    }
    io->n_fib = rec.n_fib; io->n_cir = rec.n_cir; io->n_con = rec.n_con; io->n_snr = rec.n_snr; io->n_corr = rec.n_corr; io->n_nul = rec.n_nul;
    io->n_sync_true = rec.n_sync_true; io->n_sync_false = rec.n_sync_false; io->n_services = rec.n_services; io->n_tii = rec.n_tii;
    for (int i = 0; i < io->n_subch && i < 16; i++) { io->rs_calls[i] = handlers[i].rs_calls; io->rs_uncorr[i] = handlers[i].rs_uncorr; io->rs_corr[i] = handlers[i].rs_corr; }
    return 0;
}

// ---- scan mode (RadioReceiver::restart(true)): the onSignalPresence calls the facade makes over a stream, in order (1 = true, 0 = false)
int gpu_scan_run(const float* iq, int64_t n_samples, int32_t* calls, int cap)
{
    struct ScanRec : Rec { int32_t* calls; int cap; std::atomic<int> n{0}; void onSignalPresence(bool v) override { int k = n++; if (k < cap) calls[k] = v ? 1 : 0; } };
    ScanRec rec; rec.calls = calls; rec.cap = cap;
    MemInput in(iq, n_samples);
    RadioReceiverOptions rro; rro.decodeTII = false;
    try {
        GpuRadioReceiver rx(rec, in, rro);
        rx.restart(true);
        while (!rec.failed) std::this_thread::sleep_for(std::chrono::milliseconds(2));
        rx.stop();
    } catch (const std::exception& e) { fprintf(stderr, "gpu_scan_run: %s\n", e.what()); return -1; }
    return rec.n;
}

// ---- worker-thread failure: an input whose getSamples throws after `fail_after` samples (a driver that dies under the receiver).  The
// reference reports input trouble through onInputFailure() and ends its processing thread (ofdm-processor.cpp:492-499); an exception
// that escaped the facade's std::thread would end the process instead.  Returns 1 when onInputFailure() arrived, the receiver could be
// stopped and a selected service could still be removed afterwards; 0 when the callback did not arrive in time.
int gpu_failing_input_run(const float* iq, int64_t n_samples, int64_t fail_after)
{
    struct Throwing : MemInput {
        int64_t limit;
        Throwing(const float* q, int64_t n_, int64_t lim) : MemInput(q, n_), limit(lim) {}
        int32_t getSamples(DSPCOMPLEX* b, int32_t size) override
        {
            if (pos >= limit) throw std::runtime_error("device gone");
            return MemInput::getSamples(b, size);
        }
    };
    Rec rec;
    Throwing in(iq, n_samples, fail_after);
    RadioReceiverOptions rro; rro.decodeTII = false;
    try {
        GpuRadioReceiver rx(rec, in, rro);
        rx.restart(false);
        for (int i = 0; i < 5000 && !rec.failed; i++) std::this_thread::sleep_for(std::chrono::milliseconds(2));
        const bool got = rec.failed;
        rx.stop();                                   // (must not hang or throw after the worker has ended itself)
        rx.restart_decoder();                        // clearSubchannels() with no worker running returns at once
        return got ? 1 : 0;
    } catch (const std::exception& e) { fprintf(stderr, "gpu_failing_input_run: %s\n", e.what()); return -1; }
}

// ---- batch mode: n_ens ensembles (streams of equal length, [n_ens][n_samples] cf32), each with its own FIBProcessor;
// out per ensemble: ensemble id, number of services listed, number of FIBs that passed the CRC, onServiceDetected calls
int gpu_batch_run2(const float* iq, int64_t n_samples, int n_ens, int frames_per_step, int n_steps, int signal_clock, int32_t* eid, int32_t* n_listed, int32_t* n_fib_ok, int32_t* n_detected, int32_t* n_tii);
// GpuNodeReceiver: n_ens ensembles (iq = [n_ens][n_samples]) sharded over `n_devices` shards on the devices listed (a device may repeat);
// outputs per GLOBAL ensemble as gpu_batch_run
int gpu_node_run(const float* iq, int64_t n_samples, int n_ens, const int32_t* devices, int n_devices, int frames_per_step, int n_steps,
                 int32_t* eid, int32_t* n_listed, int32_t* n_fib_ok, int32_t* n_detected, int32_t* n_shards)
{
    std::vector<std::unique_ptr<Rec>> recs;
    std::vector<RadioControllerInterface*> ctl;
    std::vector<int> fib_ok(n_ens, 0);
    struct CountRec : Rec { int* ok; void onFIBDecodeSuccess(bool good, const uint8_t* bits) override { if (good) (*ok)++; Rec::onFIBDecodeSuccess(good, bits); } };
    for (int e = 0; e < n_ens; e++) { auto r = std::unique_ptr<CountRec>(new CountRec); r->ok = &fib_ok[e]; ctl.push_back(r.get()); recs.push_back(std::move(r)); }
    try {
        RadioReceiverOptions rro;
        GpuNodeReceiver rx(ctl, (uint32_t)frames_per_step, rro, std::vector<int>(devices, devices + n_devices));
        *n_shards = (int32_t)rx.shards();
        for (size_t s = 0; s < rx.shards(); s++) {
            const size_t lo = rx.first_ensemble(s), n = rx.at(s).ensembles();
            if (dabphy_stream_upload(rx.at(s).phy(), iq + 2 * lo * (size_t)n_samples, (uint64_t)n_samples, 0) != DABPHY_OK) return -2;
            (void)n;
        }
        for (int k = 0; k < n_steps; k++) rx.process((uint32_t)frames_per_step);
        for (int e = 0; e < n_ens; e++) {
            eid[e] = rx.getEnsembleId(e); n_listed[e] = (int32_t)rx.getServiceList(e).size(); n_fib_ok[e] = fib_ok[e];
            n_detected[e] = recs[e]->n_services;
        }
    } catch (const std::exception& ex) {
        fprintf(stderr, "gpu_node_run: %s\n", ex.what());
        return -1;
    }
    return 0;
}
// ---- batch mode with services: every ensemble selects its own sub-channels (GpuBatchReceiver::addSubchannel / removeSubchannel =
// MscHandler's, per ensemble), possibly in mid-stream: sub i joins ensemble subs[i].ens before step add_step (0: from the start) and
// leaves before step remove_step (< 0: never).  Each has its own ProgrammeHandler (Reed-Solomon statistics of the reference's own
// SuperframeFilter behind DecoderAdapter) and dump file.
struct gpu_batch_sub { int32_t ens, add_step, remove_step; gpu_subch sub; };
int gpu_batch_msc_run(const float* iq, int64_t n_samples, int n_ens, int frames_per_step, int n_steps, const gpu_batch_sub* subs, int n_subs,
                      int32_t* rs_calls, int32_t* rs_uncorr, int32_t* rs_corr, int32_t* n_fib_ok)
{
    std::vector<std::unique_ptr<Rec>> recs;
    std::vector<RadioControllerInterface*> ctl;
    std::vector<int> fib_ok(n_ens, 0);
    struct CountRec : Rec { int* ok; void onFIBDecodeSuccess(bool good, const uint8_t* bits) override { if (good) (*ok)++; Rec::onFIBDecodeSuccess(good, bits); } };
    for (int e = 0; e < n_ens; e++) { auto r = std::unique_ptr<CountRec>(new CountRec); r->ok = &fib_ok[e]; ctl.push_back(r.get()); recs.push_back(std::move(r)); }
    std::vector<NullProgrammeHandler> handlers(n_subs > 0 ? n_subs : 1);
    try {
        RadioReceiverOptions rro; rro.decodeTII = false;
        GpuBatchReceiver rx(ctl, (uint32_t)frames_per_step, rro);
        if (dabphy_stream_upload(rx.phy(), iq, (uint64_t)n_samples, 0) != DABPHY_OK) return -2;
        for (int k = 0; k < n_steps; k++) {
            for (int i = 0; i < n_subs; i++) {
                const gpu_subch& s = subs[i].sub;
                if (subs[i].add_step == k) {
                    Subchannel sub;
                    sub.subChId = s.subChId; sub.startAddr = s.startAddr; sub.length = s.length;
                    sub.protectionSettings.shortForm = s.shortForm != 0; sub.protectionSettings.uepTableIndex = s.uepTableIndex; sub.protectionSettings.uepLevel = s.uepLevel;
                    sub.protectionSettings.eepProfile = s.eepProfileB ? EEPProtectionProfile::EEP_B : EEPProtectionProfile::EEP_A;
                    sub.protectionSettings.eepLevel = (EEPProtectionLevel)s.eepLevel;
                    if (!rx.addSubchannel((size_t)subs[i].ens, handlers[i], s.dabplus ? AudioServiceComponentType::DABPlus : AudioServiceComponentType::DAB, std::string(s.dump_path), sub)) return -3;
                }
                if (subs[i].remove_step == k && !rx.removeSubchannel((size_t)subs[i].ens, s.subChId)) return -4;
            }
            rx.process((uint32_t)frames_per_step);
        }
        for (int e = 0; e < n_ens; e++) n_fib_ok[e] = fib_ok[e];
    } catch (const std::exception& ex) {
        fprintf(stderr, "gpu_batch_msc_run: %s\n", ex.what());
        return -1;
    }
    // (the receiver is gone: every service's decoder thread has delivered its last frame and closed its dump)
    for (int i = 0; i < n_subs; i++) { rs_calls[i] = handlers[i].rs_calls; rs_uncorr[i] = handlers[i].rs_uncorr; rs_corr[i] = handlers[i].rs_corr; }
    return 0;
}

int gpu_batch_run(const float* iq, int64_t n_samples, int n_ens, int frames_per_step, int n_steps, int32_t* eid, int32_t* n_listed, int32_t* n_fib_ok, int32_t* n_detected, int32_t* n_tii)
{
    return gpu_batch_run2(iq, n_samples, n_ens, frames_per_step, n_steps, 1, eid, n_listed, n_fib_ok, n_detected, n_tii);
}
// signal_clock = 0: FIBProcessor ages its service counters by wall clock, as the reference does (for the contrast test)
int gpu_batch_run2(const float* iq, int64_t n_samples, int n_ens, int frames_per_step, int n_steps, int signal_clock, int32_t* eid, int32_t* n_listed, int32_t* n_fib_ok, int32_t* n_detected, int32_t* n_tii)
{
    std::vector<std::unique_ptr<Rec>> recs;
    std::vector<RadioControllerInterface*> ctl;
    std::vector<int> fib_ok(n_ens, 0);
    struct CountRec : Rec { int* ok; void onFIBDecodeSuccess(bool good, const uint8_t* bits) override { if (good) (*ok)++; Rec::onFIBDecodeSuccess(good, bits); } };
    for (int e = 0; e < n_ens; e++) { auto r = std::unique_ptr<CountRec>(new CountRec); r->ok = &fib_ok[e]; ctl.push_back(r.get()); recs.push_back(std::move(r)); }
    try {
        RadioReceiverOptions rro; rro.decodeTII = true;             // per-ensemble onTIIMeasurement calls are counted
        GpuBatchReceiver rx(ctl, (uint32_t)frames_per_step, rro);
        rx.setSignalClock(signal_clock != 0);
        if (dabphy_stream_upload(rx.phy(), iq, (uint64_t)n_samples, 0) != DABPHY_OK) return -2;
        for (int k = 0; k < n_steps; k++) rx.process((uint32_t)frames_per_step);
        for (int e = 0; e < n_ens; e++) {
            eid[e] = rx.getEnsembleId(e); n_listed[e] = (int32_t)rx.getServiceList(e).size(); n_fib_ok[e] = fib_ok[e];
            n_detected[e] = recs[e]->n_services; n_tii[e] = recs[e]->n_tii;
        }
    } catch (const std::exception& ex) {
        fprintf(stderr, "gpu_batch_run: %s\n", ex.what());
        return -1;
    }
    return 0;
}
}
