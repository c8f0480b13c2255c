// mm_trace.h -- MM_TRACE=1: phase timings of the host orchestration on stderr (diagnostics only).
#pragma once

#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace mm {

inline bool trace_on() { static const bool on = std::getenv("MM_TRACE") != nullptr; return on; }
struct TraceTimer {
    const char* what; std::chrono::steady_clock::time_point t0;
    bool done = false;
    explicit TraceTimer(const char* w) : what(w), t0(std::chrono::steady_clock::now()) {}
    void stop() {   // report now instead of at scope exit
        if (!done && trace_on())
            std::fprintf(stderr, "[mm trace] %-28s %9.3f ms\n", what,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        done = true;
    }
    ~TraceTimer() { stop(); }
};

}  // namespace mm
