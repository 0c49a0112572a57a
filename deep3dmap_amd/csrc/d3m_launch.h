// d3m_launch.h -- kernel launch macro with optional HIP-event timing (see d3m_timing_* in d3m_raster.h).
#pragma once
#include <hip/hip_runtime.h>
// ---------------------------------------------------------------------------------------------------
// optional per-kernel timing with HIP events on the launch stream (bench.py's live roofline figure).
// Off by default: LAUNCH is then a plain hipLaunchKernelGGL.
// ---------------------------------------------------------------------------------------------------
#include <map>
#include <mutex>
#include <string>
#include <vector>
struct TimedLaunch { const char* name; hipEvent_t start, stop; };
inline bool g_timing = false;
inline std::mutex g_timing_mu;
inline std::vector<TimedLaunch> g_timed;

struct LaunchTimer {
    const char* name; hipStream_t st; hipEvent_t a, b; bool on;
    LaunchTimer(const char* n, hipStream_t s) : name(n), st(s), on(g_timing) {
        if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, st); }
    }
    ~LaunchTimer() {
        if (on) {
            (void)hipEventRecord(b, st);
            std::lock_guard<std::mutex> lk(g_timing_mu);
            g_timed.push_back({name, a, b});
        }
    }
};
#define LAUNCH(name, kernel, grid, block, stream, ...)                          \
    do {                                                                        \
        LaunchTimer lt__(name, stream);                                         \
        hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);        \
    } while (0)

#define LAUNCH_SMEM(name, kernel, grid, block, smem, stream, ...)               \
    do {                                                                        \
        LaunchTimer lt__(name, stream);                                         \
        hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);     \
    } while (0)
