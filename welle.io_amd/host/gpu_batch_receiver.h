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
#pragma once
#include <chrono>
#include <memory>
#include <vector>

#include "radio-controller.h"
#include "radio-receiver-options.h"
#include "dab-constants.h"
#include "fib-processor.h"

struct dabphy_handle;

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

    private:
        std::vector<RadioControllerInterface*> rci;
        std::vector<std::unique_ptr<FIBProcessor>> fib;
        std::vector<char> synced;
        dabphy_handle* handle = nullptr;
        uint32_t max_frames;
        bool decode_tii = false;
        std::chrono::steady_clock::time_point t0;         // signal time 0 of every ensemble (construction time)
        bool use_signal_clock = true;
    public:
        void setSignalClock(bool on) { use_signal_clock = on; }   // off: FIBProcessor ages by wall clock as in the reference (tests)
};
