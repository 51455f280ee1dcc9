// welle.io_amd/host/gpu_batch_receiver.cpp -- see gpu_batch_receiver.h
#include "gpu_batch_receiver.h"

#include <cmath>
#include <cstring>
#include <stdexcept>

#include "../../include/dabphy.h"
#include "signal_clock.h"

GpuBatchReceiver::GpuBatchReceiver(const std::vector<RadioControllerInterface*>& controllers, uint32_t max_frames_, RadioReceiverOptions rro, int device) :
    rci(controllers), synced(controllers.size(), 0), max_frames(max_frames_), streams(controllers.size()), active(controllers.size()), dirty(controllers.size(), 0)
{
    if (controllers.empty() || max_frames == 0) throw std::logic_error("GpuBatchReceiver: needs at least one ensemble and one frame");
    t0 = std::chrono::steady_clock::now();
    {
        dabphy_signal_clock::Scope at(t0);                 // FIBProcessor's constructor reads the clock (fib-processor.cpp:1271)
        for (auto* c : rci) fib.emplace_back(new FIBProcessor(*c));
    }
    if (dabphy_abi_version() != DABPHY_ABI_VERSION) throw std::runtime_error("libdabphy_hip.so was built from another include/dabphy.h (ABI version)");
    dabphy_config cfg = DABPHY_CONFIG_INIT;
    cfg.n_ensembles = (uint32_t)rci.size(); cfg.max_frames = max_frames; cfg.device = device;
    cfg.fft_placement = rro.fftPlacementMethod == FFTPlacementMethod::StrongestPeak ? 0
                      : rro.fftPlacementMethod == FFTPlacementMethod::EarliestPeakWithBinning ? 1 : 2;
    cfg.freqsync_method = (int32_t)rro.freqsyncMethod;
    cfg.disable_coarse = rro.disableCoarseCorrector;
    if (dabphy_create(&cfg, &handle) != DABPHY_OK) throw std::runtime_error("GpuBatchReceiver: dabphy_create failed (no gfx950 device?)");
    decode_tii = rro.decodeTII;
    if (decode_tii && dabphy_set_tii(handle, 1) != DABPHY_OK) throw std::runtime_error(dabphy_last_error(handle));
}

GpuBatchReceiver::~GpuBatchReceiver()
{
    alive = false;
    active.clear(); streams.clear();                    // the services' decoder threads end (after the frames already queued)
    dabphy_msc_drain_wait(handle);
    dabphy_destroy(handle);
    if (drain_buf) dabphy_host_free(drain_buf);
}

// ---- services: RadioReceiver's methods (radio-receiver.cpp:120-185), per ensemble
bool GpuBatchReceiver::playSingleProgramme(size_t e, ProgrammeHandlerInterface& handler, const std::string& dumpFileName, const Service& s)
{
    return playProgramme(e, handler, s, dumpFileName, true);
}

bool GpuBatchReceiver::addServiceToDecode(size_t e, ProgrammeHandlerInterface& handler, const std::string& dumpFileName, const Service& s)
{
    return playProgramme(e, handler, s, dumpFileName, false);
}

bool GpuBatchReceiver::playProgramme(size_t e, ProgrammeHandlerInterface& handler, const Service& s, const std::string& dumpFileName, bool unique)
{
    // radio-receiver.cpp:146-174
    for (const auto& sc : fib.at(e)->getComponents(s)) {
        if (sc.transportMode() != TransportMode::Audio) continue;
        const auto subch = fib[e]->getSubchannel(sc);
        if (!subch.valid()) continue;
        if (unique) clearSubchannels(e);
        if (sc.audioType() == AudioServiceComponentType::DAB || sc.audioType() == AudioServiceComponentType::DABPlus)
            return addSubchannel(e, handler, sc.audioType(), dumpFileName, subch);
    }
    return false;
}

bool GpuBatchReceiver::removeServiceToDecode(size_t e, const Service& s)
{
    // radio-receiver.cpp:130-144
    for (const auto& sc : fib.at(e)->getComponents(s)) {
        if (sc.transportMode() != TransportMode::Audio) continue;
        const auto subch = fib[e]->getSubchannel(sc);
        if (!subch.valid()) continue;
        return removeSubchannel(e, subch.subChId);
    }
    return false;
}

bool GpuBatchReceiver::addSubchannel(size_t e, ProgrammeHandlerInterface& handler, AudioServiceComponentType ascty, const std::string& dumpFileName, const Subchannel& sub)
{
    for (const auto& st : streams.at(e)) if (st->sub.subChId == sub.subChId) return true;       // msc-handler.cpp:69-74
    dabphy_subchannel d;
    if (!SubchannelStream::describe(sub, &d)) return false;           // no such protection profile: nothing the channel decoder could do with it
    streams[e].push_back(std::make_shared<SubchannelStream>(handler, ascty, dumpFileName, sub));   // may throw like DecoderAdapter does (unknown component type)
    dirty[e] = 1;
    return true;
}

bool GpuBatchReceiver::removeSubchannel(size_t e, int subChId)
{
    for (auto it = streams.at(e).begin(); it != streams[e].end(); ++it)
        if ((*it)->sub.subChId == subChId) {
            // the library keeps decoding the sub-channel until the next batch is set up, nobody reads it any more; the stream's own
            // thread ends here (MscHandler::removeSubchannel joins DabAudio, msc-handler.cpp:105-122)
            for (auto& a : active[e]) if (a == *it) a.reset();
            streams[e].erase(it);
            dirty[e] = 1;
            return true;
        }
    return false;
}

void GpuBatchReceiver::clearSubchannels(size_t e)
{
    if (streams.at(e).empty()) return;
    for (auto& a : active[e]) a.reset();
    streams[e].clear();
    dirty[e] = 1;
}

size_t GpuBatchReceiver::process(uint32_t n_frames)
{
    if (n_frames == 0 || n_frames > max_frames) throw std::out_of_range("GpuBatchReceiver::process: n_frames");
    const size_t B = rci.size();
    // the ensembles whose selection changed tell the library their new list: it decodes this batch with it, services that stay keep
    // their time de-interleaver and superframe state (include/dabphy.h: dabphy_set_subchannels_ensemble)
    for (size_t e = 0; e < B; e++) {
        if (!dirty[e]) continue;
        std::vector<dabphy_subchannel> list; std::vector<std::shared_ptr<SubchannelStream>> now;
        for (const auto& st : streams[e]) {
            dabphy_subchannel d;
            if (!SubchannelStream::describe(st->sub, &d)) continue;
            list.push_back(d); now.push_back(st);
        }
        if (dabphy_set_subchannels_ensemble(handle, (uint32_t)e, list.data(), (uint32_t)list.size()) != DABPHY_OK) {
            // (a sub-channel the library refuses, e.g. one that runs past the CIF: decode none of this ensemble rather than the wrong ones)
            dabphy_set_subchannels_ensemble(handle, (uint32_t)e, nullptr, 0);
            now.clear();
        }
        active[e].swap(now);
        dirty[e] = 0;
    }
    if (dabphy_process(handle, n_frames) != DABPHY_OK) throw std::runtime_error(dabphy_last_error(handle));
    // every selected service's logical frames of this batch start their way to the host now, in one pass (one copy per protection
    // class), and travel while the FIBs below are parsed
    uint32_t n_desc = 0; bool draining = false;
    {
        size_t need = 0; uint32_t nd = 0;
        if (dabphy_msc_batch_size(handle, &need, &nd) != DABPHY_OK) throw std::runtime_error(dabphy_last_error(handle));
        if (nd) {
            if (need > drain_cap) {
                if (drain_buf) dabphy_host_free(drain_buf);
                drain_buf = nullptr; drain_cap = 0;
                void* p = nullptr;
                if (dabphy_host_alloc(need + need / 4, &p) != DABPHY_OK) throw std::runtime_error("GpuBatchReceiver: dabphy_host_alloc failed");
                drain_buf = static_cast<uint8_t*>(p); drain_cap = need + need / 4;
            }
            if (drain_desc.size() < nd) drain_desc.resize(nd);
            if (dabphy_msc_drain_begin(handle, drain_desc.data(), (uint32_t)drain_desc.size(), &n_desc, drain_buf, drain_cap) != DABPHY_OK) throw std::runtime_error(dabphy_last_error(handle));
            draining = true;
        }
    }
    std::vector<dabphy_frame_info> info(B * n_frames);
    std::vector<uint8_t> fibs(B * n_frames * 12 * 32), ok(B * n_frames * 12);
    dabphy_get_frame_info(handle, info.data());
    dabphy_get_fibs(handle, fibs.data(), ok.data());
    // RadioReceiverOptions::decodeTII: the measurements of the batch, ordered by frame (tii-decoder.cpp:371-377 -> onTIIMeasurement)
    const uint32_t tii_cap = 9 * n_frames;
    std::vector<dabphy_tii_measurement> tii; std::vector<int32_t> n_tii(B, 0);
    if (decode_tii) { tii.resize(B * tii_cap); dabphy_get_tii(handle, tii.data(), n_tii.data(), tii_cap); }
    size_t decoded = 0;
    for (size_t e = 0; e < B; e++) {
        uint32_t next_tii = 0;
        for (uint32_t f = 0; f < n_frames; f++) {
            const dabphy_frame_info& fi = info[e * n_frames + f];
            if (fi.valid == 3) { if (synced[e]) { rci[e]->onSyncChange(false); synced[e] = 0; } continue; }   // ofdm-processor.cpp:347-350
            if (fi.valid != 1) continue;
            if (!synced[e]) { rci[e]->onSyncChange(true); synced[e] = 1; }                                      // :369
            decoded++;
            // signal time of this frame: its last sample's position in the ensemble's stream
            const auto t_sig = t0 + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>((double)(fi.sample_pos + 196608) / 2048000.0));
            std::unique_ptr<dabphy_signal_clock::Scope> at(use_signal_clock ? new dabphy_signal_clock::Scope(t_sig) : nullptr);
            for (int k = 0; k < 12; k++) {
                const uint8_t* p = &fibs[((e * n_frames + f) * 12 + k) * 32];
                uint8_t bits[256];
                for (int i = 0; i < 256; i++) bits[i] = (p[i >> 3] >> (7 - (i & 7))) & 1;
                const bool good = ok[(e * n_frames + f) * 12 + k] != 0;
                rci[e]->onFIBDecodeSuccess(good, bits);                                                          // fic-handler.cpp:215-218
                if (good) fib[e]->processFIB(bits, (uint16_t)(k / 3));                                           // :221-229
            }
            if (!std::isnan(fi.snr)) rci[e]->onSNR(fi.snr);
            for (; next_tii < (uint32_t)n_tii[e] && next_tii < tii_cap && tii[e * tii_cap + next_tii].frame == (int32_t)f; next_tii++) {
                const dabphy_tii_measurement& m = tii[e * tii_cap + next_tii];
                tii_measurement_t t; t.comb = m.comb; t.pattern = m.pattern; t.delay_samples = m.delay_samples; t.error = m.error;
                rci[e]->onTIIMeasurement(std::move(t));
            }
        }
    }
    // thread C of every selected service: its decoded logical frames of this batch, in CIF order (dab-audio.cpp:151-160).  Every
    // service gets its whole batch in one queue entry, none waits for another; then the back-pressure of DabAudio::process
    // (dab-audio.cpp:99-106), once per batch: the PHY goes on when every decoder is within kMaxQueued frames of it
    if (draining) {
        if (dabphy_msc_drain_wait(handle) != DABPHY_OK) throw std::runtime_error(dabphy_last_error(handle));
        for (uint32_t k = 0; k < n_desc; k++) {
            const dabphy_msc_desc& d = drain_desc[k];
            if (d.ensemble >= B || d.subch_index >= active[d.ensemble].size() || !active[d.ensemble][d.subch_index]) continue;   // removed since the batch was set up
            SubchannelStream& st = *active[d.ensemble][d.subch_index];
            if ((int)d.row_bytes != st.frame_bytes || d.n_rows <= d.first_valid) continue;
            st.push_rows(drain_buf + d.offset + (size_t)d.first_valid * d.row_bytes, d.n_rows - d.first_valid);
        }
        for (size_t e = 0; e < B; e++)
            for (auto& a : active[e]) if (a) a->wait_for_space(alive);
    }
    return decoded;
}
