// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never shipped, never benchmarked as product).
//
// extern "C" taps around the UNMODIFIED welle.io reference sources (compiled where they lie under
// /root/reference by oracle/Makefile into oracle/_ref/libwelle_ref.so).  Nothing here restates an
// algorithm: every function instantiates the reference's own class and forwards the call.
//   Viterbi::deconvolve            src/backend/viterbi.cpp:227
//   FicHandler::processFicBlock    src/backend/fic-handler.cpp:111
//   EEPProtection/UEPProtection    src/backend/eep-protection.cpp:115, uep-protection.cpp:169
//   PhaseReference::findIndex      src/backend/phasereference.cpp:73
//   fft::Forward/Backward          src/various/fft.cpp:98-164 (KISS build)
//   FrequencyInterleaver::mapIn    src/backend/freq-interleaver.cpp:88
//   EnergyDispersal::dedisperse    src/backend/energy_dispersal.h:35
//   RSDecoder::DecodeSuperframe    src/backend/dabplus_decoder.cpp:326
//   RadioReceiver (end to end)     src/backend/radio-receiver.h:52, driven like src/tests/backend_tests.cpp:103-155
#include <cstring>
#include <cstdio>
#include <atomic>
#include <thread>
#include <sys/stat.h>
#include <chrono>
#include <vector>
#include <mutex>

// The end-to-end tap registers sub-channels directly on the MscHandler owned by RadioReceiver and
// reads the int16 correctors of OFDMProcessor; both are private members.  The reference sources are
// not edited -- this translation unit only relaxes access control for itself.
#include <complex>
#include <string>
#include <list>
#include <map>
#include <memory>
#include <sstream>
#include <iostream>
#include <fstream>
#include <condition_variable>
#include <functional>
#include <unordered_map>
#include <set>
#include <deque>
#include <array>
#include <algorithm>
#include <cmath>
#include <stdexcept>
#define private public
#define protected public
#include "radio-receiver.h"
#include "raw_file.h"
#include "dabplus_decoder.h"
#undef private
#undef protected
#include "energy_dispersal.h"
#include "eep-protection.h"
#include "uep-protection.h"
#include "protTables.h"
#include "dabplus_decoder.h"
#include "freq-interleaver.h"
#include "phasereference.h"

namespace {

struct NullProgrammeHandler : ProgrammeHandlerInterface {
    std::atomic<int> rs_calls{0}, rs_uncorr{0}, rs_corr{0}, frame_err{0};
    void onFrameErrors(int e) override { frame_err += e; }
    void onNewAudio(std::vector<int16_t>&&, int, const std::string&) override {}
    void onRsErrors(bool u, int n) override { rs_calls++; if (u) rs_uncorr++; rs_corr += n; }
    void onAacErrors(int) override {}
    void onNewDynamicLabel(const std::string&) override {}
    void onMOT(const mot_file_t&) override {}
    void onPADLengthError(size_t, size_t) override {}
};

struct Recorder : RadioControllerInterface {
    // capacity-bounded logs owned by the caller
    uint8_t* fib = nullptr; int fib_cap = 0; std::atomic<int> n_fib{0};        // 33 B each: ok, 32 packed bytes
    float* cir = nullptr; int cir_cap = 0; std::atomic<int> n_cir{0};          // 2048 f32 each
    float* con = nullptr; int con_cap = 0; std::atomic<int> n_con{0};          // 1200 cf32 each
    float* nul = nullptr; int nul_cap = 0; std::atomic<int> n_nul{0};          // 2656 cf32 each
    float* snr = nullptr; int snr_cap = 0; std::atomic<int> n_snr{0};
    int32_t* corr = nullptr; int corr_cap = 0;                                 // per frame: fine, coarse (read in onNewNullSymbol)
    std::atomic<int> n_sync_true{0}, n_sync_false{0};
    std::atomic<bool> failed{false};
    RadioReceiver* rx = nullptr;
    void onSNR(float s) override { int k = n_snr++; if (k < snr_cap) snr[k] = s; }
    void onFrequencyCorrectorChange(int, int) override {}
    void onSyncChange(char s) override { if (s) n_sync_true++; else n_sync_false++; }
    void onSignalPresence(bool) override {}
    void onServiceDetected(uint32_t) override {}
    void onNewEnsemble(uint16_t) override {}
    void onSetEnsembleLabel(DabLabel&) override {}
    void onDateTimeUpdate(const dab_date_time_t&) override {}
    void onFIBDecodeSuccess(bool ok, const uint8_t* bits) override {
        int k = n_fib++;
        if (k < fib_cap) {
            uint8_t* o = fib + 33 * (size_t)k;
            o[0] = ok;
            for (int i = 0; i < 32; i++) { uint8_t b = 0; for (int j = 0; j < 8; j++) b = (b << 1) | (bits[8 * i + j] & 1); o[1 + i] = b; }
        }
    }
    void onNewImpulseResponse(std::vector<float>&& d) override {
        int k = n_cir++; if (k < cir_cap && d.size() == 2048) memcpy(cir + 2048 * (size_t)k, d.data(), 2048 * 4);
    }
    void onConstellationPoints(std::vector<DSPCOMPLEX>&& d) override {
        int k = n_con.load(); if (k < con_cap && d.size() == 1200) memcpy(con + 2400 * (size_t)k, d.data(), 2400 * 4);
        n_con++;
    }
    void onNewNullSymbol(std::vector<DSPCOMPLEX>&& d) override {
        int k = n_nul.load();
        if (k < nul_cap && d.size() == 2656) memcpy(nul + 5312 * (size_t)k, d.data(), 5312 * 4);
        if (k < corr_cap && rx) { corr[2 * k] = rx->ofdmProcessor.fineCorrector; corr[2 * k + 1] = rx->ofdmProcessor.coarseCorrector; }
        n_nul++;
    }
    void onTIIMeasurement(tii_measurement_t&&) override {}
    void onMessage(message_level_t, const std::string&, const std::string&) override {}
    void onInputFailure() override { failed = true; }
};

// In-memory InputInterface with the lock-step gate of SURVEY.md section 8(c): never hand thread A a new
// frame while thread B still owes the previous one (the reference overwrites pending frames otherwise).
struct MemInput : InputInterface {
    const DSPCOMPLEX* data; int64_t n; int64_t pos = 0; Recorder* rec; std::atomic<bool> armed{false};
    MemInput(const float* iq, int64_t n_, Recorder* r) : data((const DSPCOMPLEX*)iq), n(n_), rec(r) {}
    void setFrequency(int) override {}
    int getFrequency() const override { return 0; }
    bool is_ok() override { return n - pos >= 2656; }   // a tail shorter than one request can never be served
    bool restart() override { return true; }
    void stop() override {}
    void reset() override {}
    int32_t getSamples(DSPCOMPLEX* b, int32_t size) override {
        int64_t k = size; if (k > n - pos) k = n - pos;
        memcpy(b, data + pos, k * sizeof(DSPCOMPLEX)); pos += k; return (int32_t)k;
    }
    std::vector<DSPCOMPLEX> getSpectrumSamples(int) override { return {}; }
    int32_t getSamplesToRead() override {
        if (!armed) return 0;
        if (rec->n_nul.load() > rec->n_con.load()) return 0;
        int64_t k = n - pos; if (k > 2656) k = 2656; return (int32_t)k;
    }
    float setGain(int) override { return 0; }
    float getGain() const override { return 0; }
    int getGainCount() override { return 0; }
    void setAgc(bool) override {}
    std::string getDescription() override { return "mem"; }
};

struct SilentLog { SilentLog() { } };
} // namespace

extern "C" {

struct ref_subch {       // mirrors Subchannel (dab-constants.h:164-198) fields the MSC path reads
    int32_t subChId, startAddr, length;
    int32_t shortForm, uepTableIndex, uepLevel;  // UEP
    int32_t eepProfileB, eepLevel;               // EEP (profile 0 = A, 1 = B; level 1..4)
    int32_t dabplus;                              // 1 = DAB+ (SuperframeFilter), 0 = MP2
    char dump_path[256];
};

struct ref_run_io {
    // in
    const float* iq; int64_t n_samples;
    int32_t disable_coarse, fft_placement /*0 strongest,1 earliest,2 threshold*/, freqsync /*0 GetMiddle,1 CorrelatePRS,2 PatternOfZeros*/;
    int32_t n_subch; const ref_subch* subch;
    // out (caller allocated)
    uint8_t* fib; int32_t fib_cap;
    float* cir; int32_t cir_cap;
    float* con; int32_t con_cap;
    float* nul; int32_t nul_cap;
    float* snr; int32_t snr_cap;
    int32_t* corr; int32_t corr_cap;
    // counts out
    int32_t n_fib, n_cir, n_con, n_nul, n_snr, n_sync_true, n_sync_false;
    int32_t rs_calls[16], rs_uncorr[16], rs_corr[16];
};

int ref_receiver_run(ref_run_io* io)
{
    Recorder rec;
    rec.fib = io->fib; rec.fib_cap = io->fib_cap; rec.cir = io->cir; rec.cir_cap = io->cir_cap;
    rec.con = io->con; rec.con_cap = io->con_cap; rec.nul = io->nul; rec.nul_cap = io->nul_cap;
    rec.snr = io->snr; rec.snr_cap = io->snr_cap; rec.corr = io->corr; rec.corr_cap = io->corr_cap;
    MemInput in(io->iq, io->n_samples, &rec);
    RadioReceiverOptions rro;
    rro.decodeTII = false;
    rro.disableCoarseCorrector = io->disable_coarse != 0;
    rro.fftPlacementMethod = io->fft_placement == 0 ? FFTPlacementMethod::StrongestPeak
                           : io->fft_placement == 1 ? FFTPlacementMethod::EarliestPeakWithBinning
                                                    : FFTPlacementMethod::ThresholdBeforePeak;
    rro.freqsyncMethod = io->freqsync == 0 ? FreqsyncMethod::GetMiddle
                       : io->freqsync == 1 ? FreqsyncMethod::CorrelatePRS : FreqsyncMethod::PatternOfZeros;
    std::vector<NullProgrammeHandler> handlers(io->n_subch > 0 ? io->n_subch : 1);
    {
        RadioReceiver rx(rec, in, rro);
        rec.rx = &rx;
        rx.restart(false);
        for (int i = 0; i < io->n_subch; i++) {
            const ref_subch& s = io->subch[i];
            Subchannel sub;
            sub.subChId = s.subChId; sub.startAddr = s.startAddr; sub.length = s.length;
            sub.protectionSettings.shortForm = s.shortForm != 0;
            sub.protectionSettings.uepTableIndex = s.uepTableIndex;
            sub.protectionSettings.uepLevel = s.uepLevel;
            sub.protectionSettings.eepProfile = s.eepProfileB ? EEPProtectionProfile::EEP_B : EEPProtectionProfile::EEP_A;
            sub.protectionSettings.eepLevel = (EEPProtectionLevel)s.eepLevel;
            rx.mscHandler.addSubchannel(handlers[i],
                    s.dabplus ? AudioServiceComponentType::DABPlus : AudioServiceComponentType::DAB,
                    std::string(s.dump_path), sub);
        }
        in.armed = true;
        while (!rec.failed) std::this_thread::sleep_for(std::chrono::milliseconds(2));
        // let thread B finish the frame in flight and thread C drain its ring buffers: until nothing has moved for 500 ms -- FIB count,
        // dump file sizes, superframes seen -- (a fixed 300 ms was too short for a slow decoder build on a loaded machine: RadioReceiver::stop drops
        // what the sub-channels' ring buffers still hold), at most 20 s
        {
            auto activity = [&]() {
                long long a = rec.n_fib;
                for (int i = 0; i < io->n_subch; i++) {
                    struct stat st;
                    if (io->subch[i].dump_path[0] && stat(io->subch[i].dump_path, &st) == 0) a += (long long)st.st_size;      // (buffered: moves every 4 KiB)
                    a += handlers[i].rs_calls;                                                                                // (every superframe of a DAB+ service)
                }
                return a;
            };
            long long last = activity(); int quiet_ms = 0;
            for (int waited = 0; quiet_ms < 500 && waited < 20000; waited += 20) {
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
                const long long now = activity();
                if (now != last) { last = now; quiet_ms = 0; } else quiet_ms += 20;
            }
        }
        rx.stop();
        rec.rx = nullptr;
    }   // ~RadioReceiver joins threads and closes the dump files
    io->n_fib = rec.n_fib; io->n_cir = rec.n_cir; io->n_con = rec.n_con; io->n_nul = rec.n_nul; io->n_snr = rec.n_snr;
    io->n_sync_true = rec.n_sync_true; io->n_sync_false = rec.n_sync_false;
    for (int i = 0; i < io->n_subch && i < 16; i++) {
        io->rs_calls[i] = handlers[i].rs_calls; io->rs_uncorr[i] = handlers[i].rs_uncorr; io->rs_corr[i] = handlers[i].rs_corr;
    }
    return 0;
}

// ---- unit-level taps -------------------------------------------------------------------------------

void ref_fft2048(float* io, int inverse)
{
    if (!inverse) { fft::Forward f(2048); memcpy(f.getVector(), io, 2048 * 8); f.do_FFT(); memcpy(io, f.getVector(), 2048 * 8); }
    else          { fft::Backward f(2048); memcpy(f.getVector(), io, 2048 * 8); f.do_IFFT(); memcpy(io, f.getVector(), 2048 * 8); }
}

void ref_prs_reftable(float* out /*2048 cf32*/)
{
    DABParams p(1); PhaseReference pr(p, FFTPlacementMethod::ThresholdBeforePeak);
    for (int i = 0; i < 2048; i++) { DSPCOMPLEX c = pr[i]; out[2 * i] = c.real(); out[2 * i + 1] = c.imag(); }
}

int ref_find_index(const float* v /*2048 cf32*/, int method, float* impulse /*2048*/)
{
    DABParams p(1);
    PhaseReference pr(p, method == 0 ? FFTPlacementMethod::StrongestPeak : method == 1 ? FFTPlacementMethod::EarliestPeakWithBinning
                                                                                       : FFTPlacementMethod::ThresholdBeforePeak);
    std::vector<DSPCOMPLEX> buf(2048); memcpy(buf.data(), v, 2048 * 8);
    std::vector<float> ir;
    int r = pr.findIndex(buf.data(), ir);
    if (impulse && ir.size() == 2048) memcpy(impulse, ir.data(), 2048 * 4);
    return r;
}

void ref_freq_perm(int16_t* out /*1536*/)
{
    DABParams p(1); FrequencyInterleaver fi(p);
    for (int i = 0; i < 1536; i++) out[i] = fi.mapIn(i);
}

void ref_pcodes(int idx /*0..23*/, int8_t* out /*32*/) { memcpy(out, getPCodes(idx), 32); }

void ref_viterbi(const int8_t* in /*4*(nbits+6)*/, int nbits, uint8_t* out /*nbits*/)
{
    Viterbi v(nbits);
    std::vector<softbit_t> tmp(in, in + 4 * (nbits + 6));
    v.deconvolve(tmp.data(), out);
}

// 3 FIC symbols (3 x 3072 soft bits) -> 12 FIBs: out_bits 12 x 256 bit-bytes, out_ok 12.  Returns the
// saturating success counter in percent after the block (fic-handler.cpp:236).
int ref_fic_decode(const int8_t* soft9216, uint8_t* out_bits, uint8_t* out_ok)
{
    struct R : Recorder {
        uint8_t* ob; uint8_t* ok; int k = 0;
        void onFIBDecodeSuccess(bool c, const uint8_t* bits) override { if (k < 12) { ok[k] = c; memcpy(ob + 256 * k, bits, 256); } k++; }
    } r;
    r.ob = out_bits; r.ok = out_ok;
    FicHandler f(r);
    for (int b = 1; b <= 3; b++) f.processFicBlock(soft9216 + 3072 * (b - 1), b);
    return f.getFicDecodeRatioPercent();
}

int ref_eep_deconvolve(int bitrate, int profile_b, int level, const int8_t* in, int in_len, uint8_t* out /*24*bitrate*/)
{
    EEPProtection p(bitrate, !profile_b, level);
    return p.deconvolve(in, in_len, out) ? 0 : -1;
}

int ref_uep_deconvolve(int bitrate, int level, const int8_t* in, int in_len, uint8_t* out)
{
    UEPProtection p(bitrate, level);
    return p.deconvolve(in, in_len, out) ? 0 : -1;
}

void ref_energy_dedisperse(uint8_t* bits, int n)
{
    EnergyDispersal e; std::vector<uint8_t> v(bits, bits + n); e.dedisperse(v); memcpy(bits, v.data(), n);
}

void ref_rs_superframe(uint8_t* sf, int len, int* corr, int* uncorr)
{
    RSDecoder d; int c = 0; bool u = false;
    d.DecodeSuperframe(sf, (size_t)len, c, u); *corr = c; *uncorr = u ? 1 : 0;
}

int ref_subch_bitrate(int shortForm, int uepTableIndex, int eepProfileB, int eepLevel, int length)
{
    Subchannel s; s.length = length; s.protectionSettings.shortForm = shortForm != 0;
    s.protectionSettings.uepTableIndex = uepTableIndex;
    s.protectionSettings.eepProfile = eepProfileB ? EEPProtectionProfile::EEP_B : EEPProtectionProfile::EEP_A;
    s.protectionSettings.eepLevel = (EEPProtectionLevel)eepLevel;
    return s.bitrate();
}

void ref_uep_protlevel(int idx, int* out3) { out3[0] = ProtLevel[idx][0]; out3[1] = ProtLevel[idx][1]; out3[2] = ProtLevel[idx][2]; }

// NCO table entry exactly as ofdm-processor.cpp:92-94 builds it (double cos/sin -> float)
void ref_nco_entry(int i, float* out2)
{
    DSPCOMPLEX c(cos(2.0 * M_PI * i / INPUT_RATE), sin(2.0 * M_PI * i / INPUT_RATE));
    out2[0] = c.real(); out2[1] = c.imag();
}

// ---- the real SuperframeFilter (dabplus_decoder.cpp) fed frame by frame; one record per decode attempt, filled from
// the observer callbacks (FECInfo, AudioError) and the filter's members after Feed() returns
struct ref_sf_event { int32_t cif, corrected, uncorrectable, sync, format, num_aus, au_start[7], au_crc_ok, sf_slot; };

namespace {
struct SfRecorder : SubchannelSinkObserver {
    bool attempted = false; int corr = 0; bool unc = false; unsigned bad_aus = 0;
    void FECInfo(int c, bool u) override { attempted = true; corr = c; unc = u; }
    void AudioError(const std::string& hint) override { if (hint.rfind("AU #", 0) == 0) bad_aus |= 1u << atoi(hint.c_str() + 4); }
};
}

int ref_superframe_run(const uint8_t* frames, int n_frames, int frame_len, ref_sf_event* events, int cap, uint8_t* sf_out)
{
    SfRecorder rec;
    SuperframeFilter f(&rec, false, false);
    int ne = 0;
    for (int i = 0; i < n_frames; i++) {
        rec.attempted = false; rec.bad_aus = 0;
        f.Feed(frames + (size_t)i * frame_len, (size_t)frame_len);
        if (!rec.attempted || ne >= cap) continue;
        ref_sf_event& e = events[ne];
        memset(&e, 0, sizeof e);
        e.cif = i; e.corrected = rec.corr; e.uncorrectable = rec.unc ? 1 : 0; e.sf_slot = -1;
        e.sync = f.frame_count == 0 ? 1 : 0;                     // Feed() ends with frame_count = 0 only after a successful CheckSync
        if (e.sync) {
            e.format = f.sf[2]; e.num_aus = f.num_aus;
            for (int k = 0; k <= f.num_aus; k++) e.au_start[k] = f.au_start[k];
            e.au_crc_ok = ((1u << f.num_aus) - 1) & ~rec.bad_aus;
            memcpy(sf_out + (size_t)ne * 5 * frame_len, f.sf, (size_t)5 * frame_len);
        }
        ne++;
    }
    return ne;
}

// ---- the real TIIDecoder (tii-decoder.cpp) fed one (NULL, PRS) pair at a time.  pushSymbols() drops a pair while the decoder
// thread is busy; the tap waits for State::Idle before the next pair so that every frame is analysed (what the restatement and
// the device path do).  Events: frame of the pair that triggered onTIIMeasurement, comb, pattern, delay_samples, error.
struct ref_tii_event { int32_t frame, comb, pattern, delay_samples; float error; };

namespace {
struct TiiRecorder : Recorder {
    std::mutex mu; ref_tii_event* ev = nullptr; int cap = 0, n = 0, frame = 0;
    void onTIIMeasurement(tii_measurement_t&& m) override {
        std::lock_guard<std::mutex> g(mu);
        if (n < cap) { ev[n].frame = frame; ev[n].comb = m.comb; ev[n].pattern = m.pattern; ev[n].delay_samples = m.delay_samples; ev[n].error = m.error; }
        n++;
    }
};
}

int ref_tii_run(const float* nulls, const float* prss, int n_pairs, ref_tii_event* events, int cap)
{
    TiiRecorder rec; rec.ev = events; rec.cap = cap;
    DABParams params(1);
    {
        TIIDecoder dec(params, rec);
        for (int i = 0; i < n_pairs; i++) {
            { std::lock_guard<std::mutex> g(rec.mu); rec.frame = i; }
            std::vector<complexf> nul((const complexf*)nulls + (size_t)i * 2656, (const complexf*)nulls + (size_t)(i + 1) * 2656);
            std::vector<complexf> prs((const complexf*)prss + (size_t)i * 2048, (const complexf*)prss + (size_t)(i + 1) * 2048);
            dec.pushSymbols(nul, prs);
            for (;;) {
                { std::unique_lock<std::mutex> lock(dec.m_state_mutex); if (dec.m_state == TIIDecoder::State::Idle) break; }
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
        }
    }
    return rec.n;
}

// ---- CRAWFile (input/raw_file.cpp): the reference's own file reader, reading `path` in `format` ("u8", "s8", "s16le", "s16be",
// "cf32", "auto") without throttling or rewind; out = the first `cap` samples CRAWFile::getSamples -> convertSamples
// (raw_file.cpp:203-214,324-366) hands over.  Pins the device's ingest conversion (k_ingest) to the real class.
int ref_rawfile_read(const char* path, const char* format, float* out /* cap x (re, im) */, int cap)
{
    Recorder rec;
    CRAWFile f(rec, false, false);
    f.setFileName(path, format);
    if (!f.restart()) return -1;
    int n = 0;
    for (int idle = 0; n < cap && idle < 2000;) {
        int32_t avail = f.getSamplesToRead();
        if (avail <= 0) { idle++; std::this_thread::sleep_for(std::chrono::milliseconds(1)); continue; }
        if (avail > cap - n) avail = cap - n;
        if (avail > 8192) avail = 8192;
        const int32_t got = f.getSamples(reinterpret_cast<DSPCOMPLEX*>(out) + n, avail);
        if (got <= 0) { idle++; continue; }
        n += got; idle = 0;
    }
    f.stop();
    return n;
}

// ---- the reference's RadioReceiver over a stream, optionally PACED IN REAL TIME (2.048 Msps of wall clock, like a live front end
// or welle-cli's throttled file input): number of services FIBProcessor lists at the end and of onServiceDetected calls.
// FIBProcessor ages its service-repeat counters by wall clock (fib-processor.cpp:290-309), so what it lists depends on the pace.
namespace {
struct ServiceRecorder : Recorder { std::atomic<int> n_detected{0}; void onServiceDetected(uint32_t) override { n_detected++; } };
struct PacedInput : MemInput {
    bool realtime; std::chrono::steady_clock::time_point start;
    PacedInput(const float* iq, int64_t n_, Recorder* r, bool rt) : MemInput(iq, n_, r), realtime(rt), start(std::chrono::steady_clock::now()) {}
    int32_t getSamplesToRead() override {
        const int32_t k = MemInput::getSamplesToRead();
        if (!realtime || k <= 0) return k;
        const double due = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count() * 2048000.0;
        return (double)(pos + k) <= due ? k : 0;
    }
};
}
int ref_service_list_run(const float* iq, int64_t n_samples, int realtime, int32_t* n_listed, int32_t* n_detected)
{
    ServiceRecorder rec;
    PacedInput in(iq, n_samples, &rec, realtime != 0);
    RadioReceiverOptions rro; rro.decodeTII = false;
    {
        RadioReceiver rx(rec, in, rro);
        rec.rx = &rx;
        in.start = std::chrono::steady_clock::now();
        rx.restart(false);
        in.armed = true;
        while (!rec.failed) std::this_thread::sleep_for(std::chrono::milliseconds(2));
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
        *n_listed = (int32_t)rx.getServiceList().size();
        rx.stop();
        rec.rx = nullptr;
    }
    *n_detected = rec.n_detected;
    return 0;
}

// ---- scan mode of the reference (RadioReceiver::restart(true), ofdm-processor.cpp:256-262,351-355): its onSignalPresence calls in order
int ref_scan_run(const float* iq, int64_t n_samples, int32_t* calls, int cap)
{
    struct ScanRec : Recorder { int32_t* calls; int cap; std::atomic<int> n{0}; void onSignalPresence(bool v) override { int k = n++; if (k < cap) calls[k] = v ? 1 : 0; } };
    ScanRec rec; rec.calls = calls; rec.cap = cap;
    MemInput in(iq, n_samples, &rec);
    RadioReceiverOptions rro; rro.decodeTII = false;
    {
        RadioReceiver rx(rec, in, rro);
        rec.rx = &rx;
        rx.restart(true);
        in.armed = true;
        while (!rec.failed) std::this_thread::sleep_for(std::chrono::milliseconds(2));
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
        rx.stop();
        rec.rx = nullptr;
    }
    return rec.n;
}

} // extern "C"
