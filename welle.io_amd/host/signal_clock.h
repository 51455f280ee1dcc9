// welle.io_amd/host/signal_clock.h -- a signal-time clock for the reference's FIBProcessor in batch mode.
//
// FIBProcessor ages its service-repeat counters by WALL clock: FIB_processor::FIG0Extension2 decrements them once per second of
// std::chrono::steady_clock (fib-processor.cpp:290-309) so that an SId seen once in a mis-decoded FIB is forgotten before it is seen
// again.  A batch decoded at tens of thousands of times real time passes hours of signal per wall-clock second: the counters never
// age and two stray sightings hours apart list a phantom service.  (The reference has the same problem whenever it runs faster than
// real time, e.g. welle-cli's unthrottled -t tests.)
//
// This header is FORCE-INCLUDED (-include) when fib-processor.cpp -- the reference's unmodified source -- is compiled for a GPU
// build: inside that translation unit the name `steady_clock` then denotes a clock that returns the time GpuBatchReceiver has set
// for the ensemble it is feeding (signal time = samples consumed / 2.048 MHz), and the real steady clock when no such time is set
// (the single-ensemble real-time facade).  The clock's time_point IS std::chrono::steady_clock::time_point, so the class layout of
// FIBProcessor (fib-processor.h:133) is the same in every translation unit.  No reference file is edited.
#pragma once
// everything of the standard library that mentions the real clock must be parsed before the name is redirected
#include <chrono>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <future>
#include <atomic>
#include <memory>
#include <string>
#include <vector>
#include <list>
#include <map>
#include <unordered_map>
#include <set>
#include <deque>
#include <algorithm>
#include <functional>
#include <sstream>
#include <iostream>
#include <fstream>

namespace dabphy_signal_clock {
// current thread's signal time; unset -> the real steady clock
void set(std::chrono::steady_clock::time_point t);
void clear();
std::chrono::steady_clock::time_point now() noexcept;
struct Scope {                                      // RAII: signal time for the calls made inside
    explicit Scope(std::chrono::steady_clock::time_point t) { set(t); }
    ~Scope() { clear(); }
    Scope(const Scope&) = delete; Scope& operator=(const Scope&) = delete;
};
}

#ifdef DABPHY_REDIRECT_STEADY_CLOCK                // defined only on the command line that compiles fib-processor.cpp
namespace std { namespace chrono {
struct dabphy_signal_steady_clock {
    typedef steady_clock::rep rep;
    typedef steady_clock::period period;
    typedef steady_clock::duration duration;
    typedef steady_clock::time_point time_point;
    static constexpr bool is_steady = true;
    static time_point now() noexcept { return dabphy_signal_clock::now(); }
};
} }
#define steady_clock dabphy_signal_steady_clock
#endif
