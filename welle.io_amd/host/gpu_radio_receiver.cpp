// welle.io_amd/host/gpu_radio_receiver.cpp -- see gpu_radio_receiver.h.
//
// The worker pulls samples from InputInterface exactly like OFDMProcessor::getSamples (non-blocking count, then read, is_ok()
// while starving: ofdm-processor.cpp:186-207), appends them to the HBM ring and decodes frame by frame (dabphy_process(h, 1):
// with one frame per call the coarse-corrector feedback of ofdm-processor.cpp:397 is exact).  Callbacks are issued in the
// reference's order on this thread; callees must be thread-safe as before.
#include "gpu_radio_receiver.h"
#include "../../include/dabphy.h"

#include <chrono>
#include <cmath>
#include <cstring>
#include <iostream>
#include <stdexcept>

namespace {
// The sample ring in HBM holds 80 transmission frames (126 MB of a 288 GB device) although the worker only writes kAhead = 8 frames
// ahead of the synchroniser: the 72 frames behind it are the memory OFDMProcessor::sLevel needs.  The reference advances that level
// with every sample it pulls (ofdm-processor.cpp:174,216) -- a serial recurrence, 3 ms of one GPU lane per frame -- but reads it
// only after a loss of lock.  So nothing is spent on it while tracking; at a loss of lock k_acquire replays the samples pulled
// since the last acquisition (exactly; up to 64 frames back, beyond that two bracketing replays that meet: DESIGN.md section 7).
constexpr uint64_t kRing = 80ull * 196608;
constexpr uint64_t kAhead = 8ull * 196608;
constexpr int kPull = 65536;                         // samples per InputInterface::getSamples call

bool protection_of(const Subchannel& sub, dabphy_protection* p)
{
    dabphy_subchannel d;
    const bool ok = SubchannelStream::describe(sub, &d);
    *p = d.prot;
    return ok;
}

int placement_code(FFTPlacementMethod m)
{
    return m == FFTPlacementMethod::StrongestPeak ? 0 : m == FFTPlacementMethod::EarliestPeakWithBinning ? 1 : 2;
}
}

#ifndef DABPHY_NO_TOSTRING           // radio-receiver.cpp:36-62 (that translation unit is not part of a GPU build)
const char* fftPlacementMethodToString(FFTPlacementMethod fft_placement)
{
    switch (fft_placement) {
        case FFTPlacementMethod::StrongestPeak: return "StrongestPeak";
        case FFTPlacementMethod::EarliestPeakWithBinning: return "EarliestPeakWithBinning";
        case FFTPlacementMethod::ThresholdBeforePeak: return "ThresholdBeforePeak";
    }
    throw std::logic_error("Unhandled FFT placement");
}

const char* freqSyncMethodToString(FreqsyncMethod method)
{
    switch (method) {
        case FreqsyncMethod::GetMiddle: return "GetMiddle";
        case FreqsyncMethod::CorrelatePRS: return "CorrelatePRS";
        case FreqsyncMethod::PatternOfZeros: return "PatternOfZeros";
    }
    throw std::logic_error("Unhandled freqsyncMethod placement");
}
#endif

// ------------------------------------------------------------------------------------------------ facade
GpuRadioReceiver::GpuRadioReceiver(RadioControllerInterface& rci_, InputInterface& input_, RadioReceiverOptions rro, int transmission_mode) :
    fibProcessor(rci_), params(transmission_mode), rci(rci_), input(input_), options(rro)
{
    if (transmission_mode != 1) throw std::logic_error("GpuRadioReceiver: only transmission mode I is implemented");
    if (dabphy_abi_version() != DABPHY_ABI_VERSION) throw std::runtime_error("libdabphy_hip.so was built from another include/dabphy.h (ABI version)");
    dabphy_config cfg = DABPHY_CONFIG_INIT;
    cfg.n_ensembles = 1; cfg.max_frames = 1; cfg.device = 0;
    cfg.fft_placement = placement_code(rro.fftPlacementMethod);
    cfg.freqsync_method = (int32_t)rro.freqsyncMethod;                  // GetMiddle = 0, CorrelatePRS = 1, PatternOfZeros = 2
    cfg.disable_coarse = rro.disableCoarseCorrector;
    cfg.want_constellation = 1; cfg.want_impulse_response = 1;
    const int r = dabphy_create(&cfg, &phy);
    if (r != DABPHY_OK) throw std::runtime_error("GpuRadioReceiver: dabphy_create failed (no gfx950 device?)");
}

GpuRadioReceiver::~GpuRadioReceiver()
{
    stop();
    clearSubchannels();
    if (phy) dabphy_destroy(phy);
}

void GpuRadioReceiver::restart(bool doScan)
{
    stop();
    scan_mode = doScan;                                                 // radio-receiver.cpp:82-88
    clearSubchannels();
    fibProcessor.clearEnsemble();
    input.restart();
    if (dabphy_stream_open(phy, kRing) != DABPHY_OK) throw std::runtime_error(dabphy_last_error(phy));
    sample_count = 0;
    running = true;
    worker = std::thread(&GpuRadioReceiver::run, this);
}

void GpuRadioReceiver::restart_decoder()
{
    clearSubchannels();
    fibProcessor.clearEnsemble();
}

void GpuRadioReceiver::stop()
{
    {
        // under the mutex wait_until_released() evaluates its predicate under: a waiter that has just seen `running` true is inside
        // wait() by the time the flag flips, so the notification cannot fall between its check and its sleep
        std::lock_guard<std::mutex> lock(mutex);
        running = false;
    }
    sub_cv.notify_all();                                                // (nobody waits for a worker that is going away)
    // onInputFailure() runs on the worker and controllers answer it with stop() (ofdm-processor.cpp:497): never join ourselves
    if (worker.joinable() && worker.get_id() != std::this_thread::get_id()) worker.join();
}

void GpuRadioReceiver::setReceiverOptions(const RadioReceiverOptions rro)
{
    std::lock_guard<std::mutex> lock(mutex);
    options = rro;              // applied by the worker before its next frame (the handle is not thread-safe): OFDMProcessor::setReceiverOptions,
    options_dirty = true;       // ofdm-processor.cpp:518-529, incl. the restart when disableCoarseCorrector changes; decodeTII is read per frame
}

RadioReceiverStats GpuRadioReceiver::getReceiverStats() const
{
    RadioReceiverStats s;
    s.timeLastFCT0Frame = fibProcessor.getTimeLastFCT0Frame();
    return s;
}

bool GpuRadioReceiver::serviceHasAudioComponent(const Service& s) const
{
    for (const auto& sc : getComponents(s))
        if (sc.transportMode() == TransportMode::Audio &&
            (sc.audioType() == AudioServiceComponentType::DAB || sc.audioType() == AudioServiceComponentType::DABPlus)) return true;
    return false;
}

bool GpuRadioReceiver::playSingleProgramme(ProgrammeHandlerInterface& handler, const std::string& dumpFileName, const Service& s)
{
    return playProgramme(handler, s, dumpFileName, true);
}

bool GpuRadioReceiver::addServiceToDecode(ProgrammeHandlerInterface& handler, const std::string& dumpFileName, const Service& s)
{
    return playProgramme(handler, s, dumpFileName, false);
}

bool GpuRadioReceiver::removeServiceToDecode(const Service& s)
{
    // radio-receiver.cpp:130-144
    for (const auto& sc : fibProcessor.getComponents(s)) {
        if (sc.transportMode() != TransportMode::Audio) continue;
        const auto subch = fibProcessor.getSubchannel(sc);
        if (!subch.valid()) continue;
        std::shared_ptr<Stream> gone;
        {
            std::lock_guard<std::mutex> lock(mutex);
            for (auto it = streams.begin(); it != streams.end(); ++it)
                if ((*it)->sub.subChId == subch.subChId) { gone = *it; streams.erase(it); mark_subchannels_dirty(); break; }
        }
        if (!gone) return false;
        // the caller may destroy its ProgrammeHandler as soon as this returns (MscHandler::removeSubchannel joins the DabAudio
        // thread): wait until the worker has let go of the stream, then end its decoder thread here
        wait_until_released();
        gone.reset();
        return true;
    }
    return false;
}

bool GpuRadioReceiver::playProgramme(ProgrammeHandlerInterface& handler, const Service& s, const std::string& dumpFileName, bool unique)
{
    // radio-receiver.cpp:146-174
    for (const auto& sc : fibProcessor.getComponents(s)) {
        if (sc.transportMode() != TransportMode::Audio) continue;
        const auto subch = fibProcessor.getSubchannel(sc);
        if (!subch.valid()) continue;
        if (unique) clearSubchannels();
        if (sc.audioType() == AudioServiceComponentType::DAB || sc.audioType() == AudioServiceComponentType::DABPlus)
            return addSubchannel(handler, sc.audioType(), dumpFileName, subch);
    }
    return false;
}

bool GpuRadioReceiver::addSubchannel(ProgrammeHandlerInterface& handler, AudioServiceComponentType ascty,
                                     const std::string& dumpFileName, const Subchannel& sub)
{
    {
        std::lock_guard<std::mutex> lock(mutex);
        for (const auto& st : streams) if (st->sub.subChId == sub.subChId) return true;      // msc-handler.cpp:69-74
    }
    dabphy_protection p;
    if (!protection_of(sub, &p)) return false;          // no such protection profile: nothing the channel decoder could do with it
    auto st = std::make_shared<Stream>(handler, ascty, dumpFileName, sub);      // may throw like DecoderAdapter does (unknown component type)
    std::lock_guard<std::mutex> lock(mutex);
    streams.push_back(std::move(st));
    mark_subchannels_dirty();
    return true;
}

void GpuRadioReceiver::clearSubchannels()
{
    std::list<std::shared_ptr<Stream>> gone;
    {
        std::lock_guard<std::mutex> lock(mutex);
        gone.swap(streams);
        mark_subchannels_dirty();
    }
    // as in removeServiceToDecode: the handlers may go away once this returns
    wait_until_released();
}

// Blocks until the worker has applied every sub-channel change requested so far (it then holds no reference to a removed stream), or
// is not running: it applies them before every frame and while it starves, and signals sub_cv when it has.
void GpuRadioReceiver::wait_until_released()
{
    std::unique_lock<std::mutex> lock(mutex);
    const uint64_t want = sub_requested;
    sub_cv.wait(lock, [&] { return !running || sub_applied >= want; });
}

// Pending changes from other threads (options, sub-channel selection) are applied here, on the worker: the handle is not thread-safe,
// and its sub-channel list and `active` must change together.
void GpuRadioReceiver::apply_pending()
{
    std::lock_guard<std::mutex> lock(mutex);
    if (options_dirty) {
        int32_t restarted = 0;
        if (dabphy_set_options(phy, placement_code(options.fftPlacementMethod), (int32_t)options.freqsyncMethod,
                               options.disableCoarseCorrector, &restarted) != DABPHY_OK) throw std::runtime_error(dabphy_last_error(phy));
        if (restarted) input.restart();             // OFDMProcessor::setReceiverOptions -> restart() also restarts the input (ofdm-processor.cpp:115-132)
        options_dirty = false;
    }
    if (subchannels_dirty) {
        std::vector<dabphy_subchannel> list;
        std::vector<std::shared_ptr<Stream>> now;
        for (const auto& st : streams) {
            dabphy_subchannel d;
            d.subch_id = st->sub.subChId; d.start_cu = st->sub.startAddr; d.size_cu = st->sub.length;
            if (!protection_of(st->sub, &d.prot)) continue;
            list.push_back(d); now.push_back(st);
        }
        if (dabphy_set_subchannels(phy, list.data(), (uint32_t)list.size()) != DABPHY_OK) {
            // (a sub-channel the library refuses, e.g. one that runs past the CIF: decode none rather than the wrong ones)
            dabphy_set_subchannels(phy, nullptr, 0);
            now.clear();
        }
        active.swap(now);
        subchannels_dirty = false;
        sub_applied = sub_requested;
        sub_cv.notify_all();
    }
    tii_now = options.decodeTII;                                        // read once per frame, ofdm-processor.cpp:376-380
}

// One dabphy_process(1) + the reference's callbacks for that frame slot.  Returns false when nothing was decoded.
bool GpuRadioReceiver::decode_one_frame()
{
    apply_pending();
    const bool tii = tii_now;
    if (dabphy_set_tii(phy, tii) != DABPHY_OK) throw std::runtime_error(dabphy_last_error(phy));
    if (dabphy_process(phy, 1) != DABPHY_OK) throw std::runtime_error(dabphy_last_error(phy));
    dabphy_frame_info info;
    dabphy_get_frame_info(phy, &info);
    bool signal_found = false;
    if (scan_mode) {
        // ofdm-processor.cpp:256-262: the sixth entry into notSynced without a lock reports "no signal" and ends the scan; :351-355:
        // the first successful window search reports the signal (after that attempt's impulse response, below)
        int32_t attempts = 0, at_lock = -1;
        dabphy_get_scan_stats(phy, &attempts, &at_lock);
        if (at_lock >= 0 && at_lock <= 5) signal_found = true;
        else if (attempts > 5) { rci.onSignalPresence(false); scan_mode = false; }
    }
    if (info.valid == 3 || info.valid == 1) {
        std::vector<float> cir(2048);
        dabphy_get_impulse_response(phy, cir.data());
        rci.onNewImpulseResponse(std::move(cir));                       // ofdm-processor.cpp:344
    }
    if (info.valid == 3) { rci.onSyncChange(false); return true; }      // :347-350 -> notSynced (:282)
    if (info.valid != 1) return false;
    if (signal_found) { rci.onSignalPresence(true); scan_mode = false; }       // :351-355
    rci.onSyncChange(true);                                             // :369
    // thread B of the reference: FIBs in FIC order (fic-handler.cpp:215-229)
    uint8_t fib[12][32], ok[12];
    dabphy_get_fibs(phy, &fib[0][0], ok);
    for (int k = 0; k < 12; k++) {
        uint8_t bits[256];
        for (int i = 0; i < 256; i++) bits[i] = (fib[k][i >> 3] >> (7 - (i & 7))) & 1;
        rci.onFIBDecodeSuccess(ok[k] != 0, bits);
        if (ok[k]) fibProcessor.processFIB(bits, (uint16_t)(k / 3));
    }
    if (!std::isnan(info.snr)) rci.onSNR(info.snr);                     // ofdm-decoder.cpp:155-158
    {
        std::vector<DSPCOMPLEX> con(1200);
        dabphy_get_constellation(phy, reinterpret_cast<float*>(con.data()));
        rci.onConstellationPoints(std::move(con));                      // ofdm-decoder.cpp:119-121
    }
    {
        std::vector<DSPCOMPLEX> nul(2656);
        if (dabphy_get_null_symbols(phy, reinterpret_cast<float*>(nul.data())) == DABPHY_OK)
            rci.onNewNullSymbol(std::move(nul));                        // ofdm-processor.cpp:469
    }
    if (tii) {
        dabphy_tii_measurement m[9]; int32_t n = 0;
        if (dabphy_get_tii(phy, m, &n, 9) == DABPHY_OK)
            for (int i = 0; i < n && i < 9; i++) {
                tii_measurement_t t;
                t.comb = m[i].comb; t.pattern = m[i].pattern; t.delay_samples = m[i].delay_samples; t.error = m[i].error;
                rci.onTIIMeasurement(std::move(t));                     // tii-decoder.cpp:371-377
            }
    }
    // thread C: decoded logical frames, 4 per transmission frame, in CIF order (dab-audio.cpp:151-160), handed to the sub-channel's
    // own decoder thread.  `active` is exactly the list the handle decoded this frame with.
    for (size_t idx = 0; idx < active.size(); idx++) {
        Stream& st = *active[idx];
        std::vector<uint8_t> out(4 * (size_t)st.frame_bytes);
        int32_t first_valid = 0, n_rows = 0;
        if (dabphy_get_msc(phy, (uint32_t)idx, out.data(), out.size(), &first_valid, &n_rows) != DABPHY_OK) continue;
        for (int c = first_valid; c < n_rows; c++) st.push(out.data() + (size_t)c * st.frame_bytes, running);
    }
    // onFrequencyCorrectorChange every INPUT_RATE/5 samples (ofdm-processor.cpp:218-223)
    sample_count += 196608;
    if (sample_count > INPUT_RATE / 5) { rci.onFrequencyCorrectorChange(info.fine_corrector, info.coarse_corrector); sample_count = 0; }
    return true;
}

void GpuRadioReceiver::run()
{
    std::vector<DSPCOMPLEX> buf(kPull);
    uint64_t written = 0;
    const uint64_t frame_need = 2048 + 2047 + 75ull * 2552 + 2656;       // what one SyncOnPhase pass may touch
    bool input_done = false, failed = false;
    try {
    while (running) {
        apply_pending();                 // also while starving: a removed sub-channel's stream must be let go of
        // --- fill the ring like OFDMProcessor::getSamples pulls (count, then read; is_ok() while starving)
        uint64_t consumed = dabphy_stream_consumed(phy);
        bool pulled = false;
        while (running && !input_done && written - consumed + kPull <= kAhead) {
            int32_t avail = input.getSamplesToRead();
            if (avail <= 0) {
                if (!input.is_ok()) { input_done = true; break; }
                break;
            }
            if (avail > kPull) avail = kPull;
            const int32_t got = input.getSamples(buf.data(), avail);
            if (got <= 0) break;
            if (dabphy_stream_write(phy, reinterpret_cast<const float*>(buf.data()), (uint64_t)got) != DABPHY_OK)
                throw std::runtime_error(dabphy_last_error(phy));
            written += (uint64_t)got;
            pulled = true;
        }
        // --- decode while whole frames are buffered (or the input ended: flush what is decodable)
        consumed = dabphy_stream_consumed(phy);
        bool progressed = false;
        while (running && written - consumed >= frame_need) {
            const bool did = decode_one_frame();
            const uint64_t c2 = dabphy_stream_consumed(phy);
            progressed |= did || c2 != consumed;
            if (c2 == consumed && !did) break;
            consumed = c2;
        }
        if (input_done && !progressed) {
            running = false;
            rci.onInputFailure();                                        // ofdm-processor.cpp:495-499
            break;
        }
        if (!pulled && !progressed) std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    }
    catch (const std::exception& e) {
        // a failed library call (device lost, out of memory) or a throwing input: this is a std::thread, nothing above would catch it.
        // The reference ends its processing thread the same way for its InputFailure (ofdm-processor.cpp:492-499).
        std::clog << "GpuRadioReceiver: " << e.what() << ", closing down" << std::endl;
        failed = true;
    }
    {
        std::lock_guard<std::mutex> lock(mutex);
        active.clear();
        sub_applied = sub_requested;
    }
    if (failed) {
        running = false;                 // before onInputFailure: the controller may call stop() from it (ofdm-processor.cpp:497)
        sub_cv.notify_all();
        rci.onInputFailure();
        return;
    }
    sub_cv.notify_all();
}
