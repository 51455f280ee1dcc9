// welle.io_amd/host/gpu_node_receiver.cpp -- see gpu_node_receiver.h
#include "gpu_node_receiver.h"

#include <exception>
#include <stdexcept>
#include <thread>

GpuNodeReceiver::GpuNodeReceiver(const std::vector<RadioControllerInterface*>& controllers, uint32_t max_frames, RadioReceiverOptions rro, const std::vector<int>& devices) :
    n_ens(controllers.size())
{
    if (controllers.empty() || devices.empty()) throw std::logic_error("GpuNodeReceiver: needs at least one ensemble and one device");
    const size_t want = devices.size() < n_ens ? devices.size() : n_ens;     // never an empty shard
    per = (n_ens + want - 1) / want;
    for (size_t s = 0; s * per < n_ens; s++) {
        const size_t lo = s * per, hi = lo + per < n_ens ? lo + per : n_ens;
        std::vector<RadioControllerInterface*> mine(controllers.begin() + (std::ptrdiff_t)lo, controllers.begin() + (std::ptrdiff_t)hi);
        first.push_back(lo);
        shard.emplace_back(new GpuBatchReceiver(mine, max_frames, rro, devices[s]));
    }
}

size_t GpuNodeReceiver::process(uint32_t n_frames)
{
    // one host thread per shard for the duration of the call: a shard's process() is dabphy_process (blocks until its device has decoded
    // the batch) followed by the host-side FIB hand-off, so the shards overlap in both; 50 us of thread start against milliseconds of work
    std::vector<size_t> decoded(shard.size(), 0);
    std::vector<std::exception_ptr> failed(shard.size());
    std::vector<std::thread> workers;
    for (size_t s = 1; s < shard.size(); s++)
        workers.emplace_back([&, s] { try { decoded[s] = shard[s]->process(n_frames); } catch (...) { failed[s] = std::current_exception(); } });
    try { decoded[0] = shard[0]->process(n_frames); } catch (...) { failed[0] = std::current_exception(); }
    for (auto& w : workers) w.join();
    for (auto& f : failed) if (f) std::rethrow_exception(f);
    size_t total = 0;
    for (size_t d : decoded) total += d;
    return total;
}
