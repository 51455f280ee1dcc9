// welle.io_amd/host/gpu_radio_receiver.cpp -- see gpu_radio_receiver.h.
//
// Thread model: one worker thread replaces the reference's threads A (OFDMProcessor::run), B (OfdmDecoder) and C
// (DabAudio): it pulls samples from InputInterface exactly like OFDMProcessor::getSamples (non-blocking count, then
// read, is_ok() while starving: ofdm-processor.cpp:186-207), appends them to the HBM ring and decodes frame by frame
// (dabphy_process(h, 1): with one frame per call the coarse-corrector feedback of ofdm-processor.cpp:397 is exact).
// Callbacks are issued in the reference's order on this thread; callees must be thread-safe as before.
#include "gpu_radio_receiver.h"
#include "../../include/dabphy.h"

#include <chrono>
#include <cmath>
#include <cstring>
#include <stdexcept>

namespace {
constexpr uint64_t kRing = 8ull * 196608;           // 8 transmission frames of sample ring in HBM
constexpr int kPull = 65536;                         // samples per InputInterface::getSamples call

dabphy_protection protection_of(const Subchannel& sub)
{
    dabphy_protection p;
    const auto& ps = sub.protectionSettings;
    if (ps.shortForm) dabphy_protection_uep(&p, sub.bitrate(), ps.uepLevel);
    else dabphy_protection_eep(&p, sub.bitrate(), ps.eepProfile == EEPProtectionProfile::EEP_B, (int)ps.eepLevel);
    return p;
}
}

GpuRadioReceiver::GpuRadioReceiver(RadioControllerInterface& rci_, InputInterface& input_, RadioReceiverOptions rro, int transmission_mode) :
    fibProcessor(rci_), params(transmission_mode), rci(rci_), input(input_), options(rro)
{
    if (transmission_mode != 1) throw std::logic_error("GpuRadioReceiver: only transmission mode I is implemented");
    dabphy_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.n_ensembles = 1; cfg.max_frames = 1; cfg.device = 0;
    cfg.fft_placement = rro.fftPlacementMethod == FFTPlacementMethod::StrongestPeak ? 0
                      : rro.fftPlacementMethod == FFTPlacementMethod::EarliestPeakWithBinning ? 1 : 2;
    cfg.freqsync_method = (int32_t)rro.freqsyncMethod;                  // GetMiddle = 0, CorrelatePRS = 1, PatternOfZeros = 2
    cfg.disable_coarse = rro.disableCoarseCorrector;
    cfg.want_constellation = 1; cfg.want_impulse_response = 1;
    const int r = dabphy_create(&cfg, &phy);
    if (r != DABPHY_OK) throw std::runtime_error("GpuRadioReceiver: dabphy_create failed (no gfx950 device?)");
    dabphy_set_track_slevel(phy, 1);                                    // real-time receiver, small ring: keep sLevel exact frame by frame
}

GpuRadioReceiver::~GpuRadioReceiver()
{
    stop();
    if (phy) dabphy_destroy(phy);
}

void GpuRadioReceiver::restart(bool doScan)
{
    (void)doScan;       // scan mode only paces onSignalPresence in the reference (ofdm-processor.cpp:258-262,352-356)
    stop();
    clearSubchannels();
    fibProcessor.clearEnsemble();
    input.restart();
    if (dabphy_stream_open(phy, kRing) != DABPHY_OK) throw std::runtime_error(dabphy_last_error(phy));
    was_synced = false; sample_count = 0;
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
    running = false;
    if (worker.joinable()) worker.join();
}

void GpuRadioReceiver::setReceiverOptions(const RadioReceiverOptions rro)
{
    std::lock_guard<std::mutex> lock(mutex);
    options = rro;      // placement / coarse settings are fixed at construction on the GPU path; decodeTII is consulted per frame
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
    for (const auto& sc : fibProcessor.getComponents(s)) {
        if (sc.transportMode() != TransportMode::Audio) continue;
        const auto subch = fibProcessor.getSubchannel(sc);
        if (!subch.valid()) continue;
        std::lock_guard<std::mutex> lock(mutex);
        for (auto it = streams.begin(); it != streams.end(); ++it)
            if (it->sub.subChId == subch.subChId) { streams.erase(it); subchannels_dirty = true; return true; }
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
    std::lock_guard<std::mutex> lock(mutex);
    for (const auto& st : streams) if (st.sub.subChId == sub.subChId) return true;      // msc-handler.cpp:69-74
    Stream st;
    st.sub = sub;
    AudioServiceComponentType a = ascty;
    st.adapter = std::make_unique<DecoderAdapter>(handler, (int16_t)sub.bitrate(), a, dumpFileName);
    st.frame_bytes = 3 * sub.bitrate();
    streams.push_back(std::move(st));
    subchannels_dirty = true;
    return true;
}

void GpuRadioReceiver::clearSubchannels()
{
    std::lock_guard<std::mutex> lock(mutex);
    streams.clear();
    subchannels_dirty = true;
}

void GpuRadioReceiver::push_subchannels_locked()
{
    std::vector<dabphy_subchannel> list;
    for (const auto& st : streams) {
        dabphy_subchannel d;
        d.subch_id = st.sub.subChId; d.start_cu = st.sub.startAddr; d.size_cu = st.sub.length; d.prot = protection_of(st.sub);
        list.push_back(d);
    }
    dabphy_set_subchannels(phy, list.data(), (uint32_t)list.size());
    subchannels_dirty = false;
}

// One dabphy_process(1) + the reference's callbacks for that frame slot.  Returns false when nothing was decoded.
bool GpuRadioReceiver::decode_one_frame(uint64_t written)
{
    (void)written;
    {
        std::lock_guard<std::mutex> lock(mutex);
        if (subchannels_dirty) push_subchannels_locked();
    }
    bool tii;
    {
        std::lock_guard<std::mutex> lock(mutex);
        tii = options.decodeTII;                                        // read once per frame, ofdm-processor.cpp:376-380
    }
    if (dabphy_set_tii(phy, tii) != DABPHY_OK) throw std::runtime_error(dabphy_last_error(phy));
    if (dabphy_process(phy, 1) != DABPHY_OK) throw std::runtime_error(dabphy_last_error(phy));
    dabphy_frame_info info;
    dabphy_get_frame_info(phy, &info);
    if (info.valid == 3 || info.valid == 1) {
        std::vector<float> cir(2048);
        dabphy_get_impulse_response(phy, cir.data());
        rci.onNewImpulseResponse(std::move(cir));                       // ofdm-processor.cpp:344
    }
    if (info.valid == 3) { rci.onSyncChange(false); was_synced = false; return true; }      // :347-350 -> notSynced (:282)
    if (info.valid != 1) return false;
    if (!was_synced) was_synced = true;
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
    // thread C: decoded logical frames, 4 per transmission frame, in CIF order (dab-audio.cpp:151-160)
    {
        std::lock_guard<std::mutex> lock(mutex);
        uint32_t idx = 0;
        for (auto& st : streams) {
            std::vector<uint8_t> out(4 * (size_t)st.frame_bytes);
            int32_t first_valid = 0;
            if (dabphy_get_msc(phy, idx++, out.data(), &first_valid) != DABPHY_OK) continue;
            std::vector<uint8_t> bits(8 * (size_t)st.frame_bytes);
            for (int c = first_valid; c < 4; c++) {
                const uint8_t* p = out.data() + (size_t)c * st.frame_bytes;
                for (int i = 0; i < 8 * st.frame_bytes; i++) bits[i] = (p[i >> 3] >> (7 - (i & 7))) & 1;
                st.adapter->addtoFrame(bits.data());
            }
        }
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
    bool input_done = false;
    while (running) {
        // --- fill the ring like OFDMProcessor::getSamples pulls (count, then read; is_ok() while starving)
        uint64_t consumed = dabphy_stream_consumed(phy);
        bool pulled = false;
        while (running && !input_done && written - consumed + kPull <= kRing) {
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
            const bool did = decode_one_frame(written);
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
