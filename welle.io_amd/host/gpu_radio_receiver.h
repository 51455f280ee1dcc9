// welle.io_amd/host/gpu_radio_receiver.h -- the reference's RadioReceiver facade on top of libdabphy_hip.so.
//
// Same constructor and public methods as RadioReceiver (src/backend/radio-receiver.h:52-116): welle-cli / the
// GUI compile against it unchanged once `using RadioReceiver = GpuRadioReceiver;` (or a rename) is in place --
// see INTEGRATION.md.  It is written against the reference's OWN headers (radio-controller.h, fib-processor.h,
// decoder_adapter.h, dab-constants.h); only the PHY hot path behind them is replaced:
//
//   reference object                      replaced by
//   OFDMProcessor (+PhaseReference)       dabphy_process: k_acquire, k_sync_find, k_cp_products, k_sync_finish
//   OfdmDecoder                           k_demod, k_snr*
//   FicHandler (depuncture/Viterbi/CRC)   k_fic_gather, k_viterbi, k_fib_crc        -> FIBProcessor::processFIB stays
//   MscHandler + DabAudio + Protection    k_msc_gather, k_viterbi                   -> DecoderAdapter::addtoFrame stays
//
// What stays on the host, unchanged: FIBProcessor (FIG parsing, service database), DecoderAdapter and the audio /
// PAD decoders behind it, every front-end.
#pragma once
#include <atomic>
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

struct dabphy_handle;

struct RadioReceiverStats;      // radio-receiver.h defines it when the facade replaces that header; see .cpp

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
        std::chrono::system_clock::time_point getTimeLastFCT0Frame() const { return fibProcessor.getTimeLastFCT0Frame(); }

        // MscHandler::addSubchannel / removeSubchannel (msc-handler.cpp:61-122), used by playProgramme
        bool addSubchannel(ProgrammeHandlerInterface& handler, AudioServiceComponentType ascty,
                           const std::string& dumpFileName, const Subchannel& sub);
        void clearSubchannels();

        FIBProcessor fibProcessor;      // public like FicHandler::fibProcessor (fic-handler.h:47)

    private:
        struct Stream {
            Subchannel sub;
            std::unique_ptr<DecoderAdapter> adapter;
            int frame_bytes = 0;
        };
        bool playProgramme(ProgrammeHandlerInterface& handler, const Service& s, const std::string& dumpFileName, bool unique);
        void run();
        void push_subchannels_locked();
        bool decode_one_frame(uint64_t written);

        DABParams params;
        RadioControllerInterface& rci;
        InputInterface& input;
        RadioReceiverOptions options;
        dabphy_handle* phy = nullptr;
        std::thread worker;
        std::atomic<bool> running{false};
        std::mutex mutex;                 // guards streams / subchannels_dirty / options
        std::list<Stream> streams;
        bool subchannels_dirty = false;
        bool was_synced = false;
        int sample_count = 0;             // OFDMProcessor::sampleCnt (onFrequencyCorrectorChange pacing)
};
