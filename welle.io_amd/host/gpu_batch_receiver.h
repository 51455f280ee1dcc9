// welle.io_amd/host/gpu_batch_receiver.h -- batch mode of the host side: N ensembles decoded in lock step on one GPU, each
// with its OWN instance of the reference's FIBProcessor (src/backend/fib-processor.h:39-57: FIG parsing and service database,
// unchanged) fed from the batched FIC output of libdabphy_hip.so.
//
// What the single-ensemble façade (gpu_radio_receiver.h) does per frame, this does per (ensemble, frame) of a batch:
//   dabphy_get_fibs  ->  RadioControllerInterface::onFIBDecodeSuccess(ok, 256 bit-bytes)   fic-handler.cpp:215-218
//                    ->  FIBProcessor::processFIB(bits, fib / 3)                             fic-handler.cpp:221-229
// in FIC order, so every ensemble's FIBProcessor sees exactly the call sequence FicHandler gives it in the reference.
// The per-ensemble RadioControllerInterface receives that ensemble's control-plane callbacks (onServiceDetected,
// onNewEnsemble, onSetEnsembleLabel, onDateTimeUpdate) and its FIB / SNR / sync callbacks.
//
// Service ageing: FIBProcessor decrements its service-repeat counters once per second of steady_clock
// (fib-processor.cpp:290-309).  A batch decoded at thousands of times real time passes hours of signal per wall-clock second, so
// every ensemble gets a SIGNAL clock instead: while its FIBs are processed, `steady_clock` inside fib-processor.cpp reads
// t0 + (samples of that ensemble consumed so far) / 2.048 MHz (signal_clock.h; fib-processor.cpp is compiled with that header
// force-included, unmodified).  The counters then age exactly as they do in a receiver running in real time, at any decode speed.
//
// Services (SURVEY 8a-11 ... 8a-16 for a batch): every ensemble selects ITS OWN, as every RadioReceiver of the reference does
// (radio-receiver.cpp:120-185 -> MscHandler::addSubchannel / removeSubchannel, msc-handler.cpp:61-127): addServiceToDecode(e, ...) /
// removeServiceToDecode(e, ...) mirror the facade's methods per ensemble, the selection reaches the library through
// dabphy_set_subchannels_ensemble before the next batch, and process() hands every selected sub-channel's logical frames to a
// DecoderAdapter of its own (subchannel_stream.h: one decoder thread per service, as DabAudio::run is).  All services of all ensembles
// leave the device in ONE bulk drain per batch (dabphy_msc_drain_begin: one copy per protection class into page-locked memory, in flight
// while the FIBs are parsed); every service is handed its whole batch without blocking, back-pressure is applied once per batch.
#pragma once
#include <atomic>
#include <chrono>
#include <list>
#include <memory>
#include <vector>

#include "radio-controller.h"
#include "radio-receiver-options.h"
#include "dab-constants.h"
#include "fib-processor.h"
#include "subchannel_stream.h"

class GpuBatchReceiver {
    public:
        // controllers.size() = number of ensembles; they must outlive the receiver
        GpuBatchReceiver(const std::vector<RadioControllerInterface*>& controllers, uint32_t max_frames, RadioReceiverOptions rro, int device = 0);
        ~GpuBatchReceiver();
        GpuBatchReceiver(const GpuBatchReceiver&) = delete;
        GpuBatchReceiver& operator=(const GpuBatchReceiver&) = delete;

        size_t ensembles() const { return fib.size(); }
        dabphy_handle* phy() { return handle; }             // sample input (dabphy_stream_*) and sub-channel selection go through the C ABI

        // decode the next n_frames of every ensemble and hand the FIBs to the control plane; returns the number of (ensemble,
        // frame) pairs that were demodulated
        size_t process(uint32_t n_frames);

        // the reference's getters, per ensemble (RadioReceiver::getEnsembleId() ... radio-receiver.h:88-104)
        uint16_t getEnsembleId(size_t e) const { return fib[e]->getEnsembleId(); }
        DabLabel getEnsembleLabel(size_t e) const { return fib[e]->getEnsembleLabel(); }
        std::vector<Service> getServiceList(size_t e) const { return fib[e]->getServiceList(); }
        std::list<ServiceComponent> getComponents(size_t e, const Service& s) const { return fib[e]->getComponents(s); }
        Subchannel getSubchannel(size_t e, const ServiceComponent& sc) const { return fib[e]->getSubchannel(sc); }

        // RadioReceiver::playSingleProgramme / addServiceToDecode / removeServiceToDecode (radio-receiver.cpp:120-185) of ensemble e.
        // Call them between process() calls, from the thread that calls process() (the handle is not thread-safe); the selection applies
        // from the next batch on.  A removed service's handler may be destroyed when removeServiceToDecode returns (its decoder thread
        // has ended, the frames already decoded have been delivered).
        bool playSingleProgramme(size_t e, ProgrammeHandlerInterface& handler, const std::string& dumpFileName, const Service& s);
        bool addServiceToDecode(size_t e, ProgrammeHandlerInterface& handler, const std::string& dumpFileName, const Service& s);
        bool removeServiceToDecode(size_t e, const Service& s);
        // MscHandler::addSubchannel / removeSubchannel (msc-handler.cpp:61-122) of ensemble e
        bool addSubchannel(size_t e, ProgrammeHandlerInterface& handler, AudioServiceComponentType ascty, const std::string& dumpFileName, const Subchannel& sub);
        bool removeSubchannel(size_t e, int subChId);
        void clearSubchannels(size_t e);

    private:
        std::vector<RadioControllerInterface*> rci;
        std::vector<std::unique_ptr<FIBProcessor>> fib;
        std::vector<char> synced;
        dabphy_handle* handle = nullptr;
        uint32_t max_frames;
        bool playProgramme(size_t e, ProgrammeHandlerInterface& handler, const Service& s, const std::string& dumpFileName, bool unique);
        std::vector<std::list<std::shared_ptr<SubchannelStream>>> streams;    // [ensemble]: the services selected, in the order they were added
        std::vector<std::vector<std::shared_ptr<SubchannelStream>>> active;   // [ensemble]: the list the library decodes the current batch with, in its order
        std::vector<char> dirty;                                              // [ensemble]: `streams` changed since the library was told
        std::atomic<bool> alive{true};
        // bulk drain of a batch's logical frames (dabphy_msc_drain_begin / _wait): page-locked landing buffer and index table, grown on demand
        uint8_t* drain_buf = nullptr; size_t drain_cap = 0;
        std::vector<dabphy_msc_desc> drain_desc;
        bool decode_tii = false;
        std::chrono::steady_clock::time_point t0;         // signal time 0 of every ensemble (construction time)
        bool use_signal_clock = true;
    public:
        void setSignalClock(bool on) { use_signal_clock = on; }   // off: FIBProcessor ages by wall clock as in the reference (tests)
};
