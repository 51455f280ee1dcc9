// welle.io_amd/host/gpu_radio_receiver.h -- the reference's RadioReceiver facade on top of libdabphy_hip.so.
//
// Same constructor and public methods as RadioReceiver (src/backend/radio-receiver.h:52-116).  welle-cli compiles against it
// UNCHANGED: welle.io_amd/host/dropin/ holds a `radio-receiver.h` that shadows the reference's header and gives this class the
// name RadioReceiver (oracle/Makefile builds welle-cli both ways).  It is written against the reference's OWN
// headers (radio-controller.h, fib-processor.h, decoder_adapter.h, dab-constants.h); only the PHY hot path behind them is replaced:
//
//   reference object                      replaced by
//   OFDMProcessor (+PhaseReference)       dabphy_process: k_acquire, k_sync_find, k_sync_finish
//   OfdmDecoder                           k_demod, k_snr*
//   FicHandler (depuncture/Viterbi/CRC)   k_fic_gather, k_viterbi, k_fib_crc        -> FIBProcessor::processFIB stays
//   MscHandler + DabAudio + Protection    k_msc_gather, k_viterbi                   -> DecoderAdapter::addtoFrame stays
//
// What stays on the host, unchanged: FIBProcessor (FIG parsing, service database), DecoderAdapter and the audio /
// PAD decoders behind it, every front-end.
//
// Threads.  The reference runs thread A (OFDMProcessor::run), thread B (OfdmDecoder) and one thread C per selected
// sub-channel (DabAudio::run).  Here one worker replaces A + B (pull samples, dabphy_process(1), the per-frame callbacks in the
// reference's order) and every selected sub-channel keeps its own thread C: the worker queues the decoded logical frames, the
// sub-channel's thread feeds DecoderAdapter::addtoFrame -- audio decoding never holds up the PHY or another caller, and
// removeServiceToDecode returns only after that thread has ended (as MscHandler::removeSubchannel joins DabAudio).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <list>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "radio-controller.h"
#include "radio-receiver-options.h"
#include "dab-constants.h"
#include "fib-processor.h"
#include "decoder_adapter.h"
#include "subchannel_stream.h"


#ifndef DABPHY_HAVE_RECEIVER_STATS                  // radio-receiver.h:48-50 (the drop-in header replaces that file)
#define DABPHY_HAVE_RECEIVER_STATS
struct RadioReceiverStats {
    std::chrono::system_clock::time_point timeLastFCT0Frame;
};
const char* fftPlacementMethodToString(FFTPlacementMethod fft_placement);
const char* freqSyncMethodToString(FreqsyncMethod method);
#endif

class GpuRadioReceiver {
    public:
        GpuRadioReceiver(RadioControllerInterface& rci, InputInterface& input, RadioReceiverOptions rro,
                         int transmission_mode = 1);
        ~GpuRadioReceiver();
        GpuRadioReceiver(const GpuRadioReceiver&) = delete;
        GpuRadioReceiver& operator=(const GpuRadioReceiver&) = delete;

        void restart(bool doScan);
        void restart_decoder();
        void stop();
        void setReceiverOptions(const RadioReceiverOptions rro);

        bool playSingleProgramme(ProgrammeHandlerInterface& handler, const std::string& dumpFileName, const Service& s);
        bool addServiceToDecode(ProgrammeHandlerInterface& handler, const std::string& dumpFileName, const Service& s);
        bool removeServiceToDecode(const Service& s);

        uint16_t getEnsembleId() const { return fibProcessor.getEnsembleId(); }
        uint8_t getEnsembleEcc() const { return fibProcessor.getEnsembleEcc(); }
        DabLabel getEnsembleLabel() const { return fibProcessor.getEnsembleLabel(); }
        std::vector<Service> getServiceList() const { return fibProcessor.getServiceList(); }
        Service getService(uint32_t sId) const { return fibProcessor.getService(sId); }
        std::list<ServiceComponent> getComponents(const Service& s) const { return fibProcessor.getComponents(s); }
        bool serviceHasAudioComponent(const Service& s) const;
        Subchannel getSubchannel(const ServiceComponent& sc) const { return fibProcessor.getSubchannel(sc); }
        DABParams& getParams() { return params; }
        RadioReceiverStats getReceiverStats() const;                    // radio-receiver.cpp:225-229
        std::chrono::system_clock::time_point getTimeLastFCT0Frame() const { return fibProcessor.getTimeLastFCT0Frame(); }

        // MscHandler::addSubchannel / removeSubchannel (msc-handler.cpp:61-122), used by playProgramme
        bool addSubchannel(ProgrammeHandlerInterface& handler, AudioServiceComponentType ascty,
                           const std::string& dumpFileName, const Subchannel& sub);
        void clearSubchannels();

        FIBProcessor fibProcessor;      // public like FicHandler::fibProcessor (fic-handler.h:47)

    private:
        using Stream = SubchannelStream;     // one selected sub-channel: DabAudio's role after the channel decoder, on its own thread (subchannel_stream.h)
        bool playProgramme(ProgrammeHandlerInterface& handler, const Service& s, const std::string& dumpFileName, bool unique);
        void run();
        bool decode_one_frame();
        void apply_pending();

        DABParams params;
        RadioControllerInterface& rci;
        InputInterface& input;
        dabphy_handle* phy = nullptr;
        std::thread worker;
        std::atomic<bool> running{false};
        std::mutex mutex;                 // guards streams / subchannels_dirty / options / options_dirty
        RadioReceiverOptions options;
        bool options_dirty = false;
        std::list<std::shared_ptr<Stream>> streams;
        bool subchannels_dirty = false;
        // the worker has let go of every stream that was removed up to request number sub_requested (both under `mutex`, sub_cv)
        uint64_t sub_requested = 0, sub_applied = 0;
        std::condition_variable sub_cv;
        void mark_subchannels_dirty() { subchannels_dirty = true; ++sub_requested; }      // caller holds `mutex`
        void wait_until_released();
        std::vector<std::shared_ptr<Stream>> active;   // worker thread only: the streams whose sub-channels the handle currently decodes, in its order
        std::atomic<bool> scan_mode{false};            // OFDMProcessor::scanMode (ofdm-processor.h:110)
        bool tii_now = false;
        int sample_count = 0;             // OFDMProcessor::sampleCnt (onFrequencyCorrectorChange pacing)
};
