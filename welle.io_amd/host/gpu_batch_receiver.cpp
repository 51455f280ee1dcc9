// welle.io_amd/host/gpu_batch_receiver.cpp -- see gpu_batch_receiver.h
#include "gpu_batch_receiver.h"

#include <cmath>
#include <cstring>
#include <stdexcept>

#include "../../include/dabphy.h"
#include "signal_clock.h"

GpuBatchReceiver::GpuBatchReceiver(const std::vector<RadioControllerInterface*>& controllers, uint32_t max_frames_, RadioReceiverOptions rro, int device) :
    rci(controllers), synced(controllers.size(), 0), max_frames(max_frames_)
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

GpuBatchReceiver::~GpuBatchReceiver() { dabphy_destroy(handle); }

size_t GpuBatchReceiver::process(uint32_t n_frames)
{
    if (n_frames == 0 || n_frames > max_frames) throw std::out_of_range("GpuBatchReceiver::process: n_frames");
    if (dabphy_process(handle, n_frames) != DABPHY_OK) throw std::runtime_error(dabphy_last_error(handle));
    const size_t B = rci.size();
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
    return decoded;
}
