// welle.io_amd/host/subchannel_stream.cpp -- see subchannel_stream.h
#include "subchannel_stream.h"

#include <chrono>
#include <cstring>

bool SubchannelStream::describe(const Subchannel& sub, dabphy_subchannel* d)
{
    memset(d, 0, sizeof *d);
    d->subch_id = sub.subChId; d->start_cu = sub.startAddr; d->size_cu = sub.length;
    const auto& ps = sub.protectionSettings;
    const int r = ps.shortForm ? dabphy_protection_uep(&d->prot, sub.bitrate(), ps.uepLevel)
                               : dabphy_protection_eep(&d->prot, sub.bitrate(), ps.eepProfile == EEPProtectionProfile::EEP_B, (int)ps.eepLevel);
    return r == DABPHY_OK;
}

SubchannelStream::SubchannelStream(ProgrammeHandlerInterface& handler, AudioServiceComponentType ascty, const std::string& dumpFileName, const Subchannel& s) :
    sub(s), frame_bytes(3 * s.bitrate()), adapter(handler, (int16_t)s.bitrate(), ascty, dumpFileName)
{
    thread = std::thread(&SubchannelStream::run, this);
}

SubchannelStream::~SubchannelStream()
{
    {
        std::lock_guard<std::mutex> lock(m);
        closing = true;
    }
    cv.notify_all();
    if (thread.joinable()) thread.join();           // frames already queued are still delivered (a file ends with its last frames decoded)
}

void SubchannelStream::push(const uint8_t* p, const std::atomic<bool>& receiver_running)
{
    {
        // a full queue holds the channel decoder back (an unthrottled file input would otherwise run arbitrarily far ahead of the audio
        // decoder): DabAudio::process waits the same way on its ring buffer (dab-audio.cpp:99-106).  The wait is bounded: a receiver
        // that is being stopped must not sit behind a stalled audio decoder (the frame is then dropped with the receiver)
        std::unique_lock<std::mutex> lock(m);
        while (!cv_space.wait_for(lock, std::chrono::milliseconds(50), [&] { return closing || queued_frames < kMaxQueued; }))
            if (!receiver_running) return;
        if (closing) return;
        q.emplace_back(p, p + frame_bytes);
        queued_frames += 1;
    }
    cv.notify_one();
}

void SubchannelStream::push_rows(const uint8_t* rows, int n_rows)
{
    if (n_rows <= 0) return;
    {
        std::lock_guard<std::mutex> lock(m);
        if (closing) return;
        q.emplace_back(rows, rows + (size_t)n_rows * frame_bytes);
        queued_frames += (size_t)n_rows;
    }
    cv.notify_one();
}

void SubchannelStream::wait_for_space(const std::atomic<bool>& receiver_running)
{
    std::unique_lock<std::mutex> lock(m);
    while (!cv_space.wait_for(lock, std::chrono::milliseconds(50), [&] { return closing || queued_frames <= kMaxQueued; }))
        if (!receiver_running) return;
}

void SubchannelStream::run()
{
    std::vector<uint8_t> bits(8 * (size_t)frame_bytes);
    for (;;) {
        std::vector<uint8_t> chunk;
        {
            std::unique_lock<std::mutex> lock(m);
            cv.wait(lock, [&] { return closing || !q.empty(); });
            if (q.empty()) return;
            chunk = std::move(q.front()); q.pop_front();
        }
        for (size_t off = 0; off + (size_t)frame_bytes <= chunk.size(); off += (size_t)frame_bytes) {
            const uint8_t* f = chunk.data() + off;
            for (int i = 0; i < 8 * frame_bytes; i++) bits[i] = (f[i >> 3] >> (7 - (i & 7))) & 1;     // DabAudio hands over one bit per byte
            adapter.addtoFrame(bits.data());                                                          // dab-audio.cpp:157
            {
                std::lock_guard<std::mutex> lock(m);
                queued_frames -= 1;
            }
            cv_space.notify_all();
        }
    }
}
