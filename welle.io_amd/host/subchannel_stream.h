// welle.io_amd/host/subchannel_stream.h -- one selected sub-channel above the channel decoder: DabAudio's role after
// Protection::deconvolve (src/backend/dab-audio.cpp:151-160) -- the decoded logical frames of libdabphy_hip.so go to the reference's
// unmodified DecoderAdapter (decoder_adapter.h) on the sub-channel's OWN thread, as DabAudio::run is one thread per sub-channel: audio
// decoding never holds up the PHY or another sub-channel, and destroying the stream joins that thread after the frames already queued
// have been delivered (MscHandler::removeSubchannel joins DabAudio, msc-handler.cpp:105-122).
// Used by GpuRadioReceiver (one ensemble, the facade) and GpuBatchReceiver (every ensemble of a batch selects its own services).
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "radio-controller.h"
#include "dab-constants.h"
#include "decoder_adapter.h"
#include "../../include/dabphy.h"

struct SubchannelStream {
    SubchannelStream(ProgrammeHandlerInterface& handler, AudioServiceComponentType ascty, const std::string& dumpFileName, const Subchannel& sub);
    ~SubchannelStream();
    SubchannelStream(const SubchannelStream&) = delete;
    SubchannelStream& operator=(const SubchannelStream&) = delete;
    // one logical frame (3 * bitrate bytes, MSB first).  A full queue holds the caller back as DabAudio::process does on a full mscBuffer
    // (dab-audio.cpp:99-106); the wait ends when keep_waiting turns false (a receiver that is being stopped: the frame is dropped with it)
    void push(const uint8_t* frame_bytes_msb_first, const std::atomic<bool>& keep_waiting);
    // a whole batch of logical frames (n_rows x frame_bytes, CIF order) in ONE queue entry, never blocking: the batch receiver feeds
    // every service of every ensemble first and applies the back-pressure afterwards, once per batch (wait_for_space), so that a slow
    // audio decoder holds up neither the other services' hand-over nor their decoding
    void push_rows(const uint8_t* rows, int n_rows);
    // blocks while more than kMaxQueued logical frames are waiting for this service's decoder (DabAudio::process waits the same way on
    // its ring buffer, dab-audio.cpp:99-106); ends when keep_waiting turns false
    void wait_for_space(const std::atomic<bool>& keep_waiting);
    Subchannel sub;
    int frame_bytes;
    static constexpr size_t kMaxQueued = 64;          // logical frames a sub-channel's decoder thread may lag behind the channel decoder
    // Subchannel -> the C ABI's record (protection profile via dabphy_protection_eep / _uep); false: no such profile
    static bool describe(const Subchannel& sub, dabphy_subchannel* out);
  private:
    void run();
    DecoderAdapter adapter;
    std::mutex m; std::condition_variable cv, cv_space;
    std::deque<std::vector<uint8_t>> q;                  // entries of one or more whole logical frames
    size_t queued_frames = 0;
    bool closing = false;
    std::thread thread;
};
