// d3m_launch.h -- kernel launch macro with optional HIP-event timing (see d3m_timing_* in d3m_raster.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
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
// Developer builds only (-DD3M_DEV_SKIP, tools_dev/): kernels whose name appears in $D3M_SKIP are not launched, to
// see what a kernel costs on the critical path of a captured step.  Results are then wrong by construction.
#ifdef D3M_DEV_SKIP
#include <cstdlib>
#include <cstring>
inline bool d3m_dev_skip(const char* name) {
    static const char* list = getenv("D3M_SKIP");
    return list && strstr(list, name);
}
#else
inline bool d3m_dev_skip(const char*) { return false; }
#endif
// D3M_TRACE_LAUNCHES=1 in the environment (debugging a hang): every launch is announced on stderr, the stream synchronised
// behind it (eager launches; inside a capture only announced) and the result reported -- the last line names the kernel that
// does not come back.  The switch is read ONCE, at the first launch of the process: set it before anything is launched.
#include <cstdio>
#include <cstdlib>
inline bool d3m_trace_launches() {
    static const bool on = [] { const char* e = getenv("D3M_TRACE_LAUNCHES"); return e && e[0] == '1'; }();
    return on;
}
#include <chrono>
inline std::chrono::steady_clock::time_point g_trace_t0;
inline void d3m_trace_begin(const char* name, dim3 g, dim3 b) {
    if (d3m_trace_launches()) {
        fprintf(stderr, "[d3m] launch %s grid (%u,%u,%u) block %u ... ", name, g.x, g.y, g.z, b.x);
        fflush(stderr);
        g_trace_t0 = std::chrono::steady_clock::now();
    }
}
inline void d3m_trace_end(hipStream_t st) {
    if (d3m_trace_launches()) {
        // (a stream that is being captured must not be synchronised: that invalidates the capture -- announce only)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
            fprintf(stderr, "recorded (stream is capturing)\n");
            fflush(stderr);
            return;
        }
        hipError_t e = hipStreamSynchronize(st);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_trace_t0).count();
        fprintf(stderr, "%s (%.3f ms incl. the host round trip)\n", e == hipSuccess ? "done" : hipGetErrorString(e), ms);
        fflush(stderr);
    }
}
// integer tuning switch from the environment (developer A/B runs; the defaults are the measured choices)
inline int d3m_env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && e[0]) ? atoi(e) : dflt;
}
#define LAUNCH(name, kernel, grid, block, stream, ...)                          \
    do {                                                                        \
        LaunchTimer lt__(name, stream);                                         \
        d3m_trace_begin(name, grid, block);                                     \
        if (!d3m_dev_skip(name)) hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);        \
        d3m_trace_end(stream);                                                  \
    } while (0)

#define LAUNCH_SMEM(name, kernel, grid, block, smem, stream, ...)               \
    do {                                                                        \
        LaunchTimer lt__(name, stream);                                         \
        d3m_trace_begin(name, grid, block);                                     \
        if (!d3m_dev_skip(name)) hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);     \
        d3m_trace_end(stream);                                                  \
    } while (0)

// Zero-fill as a KERNEL (not hipMemsetAsync): inside a captured HIP graph a memset becomes a memset node, and on
// this ROCm build its ordering against the next kernel node was not reliable when the graph is replayed from an
// idle non-default stream (tile counters were not cleared -> list overruns -> memory faults).  A fill kernel is an
// ordinary kernel node.  `bytes` must be a multiple of 4 and `p` 4-byte aligned (16-byte fast path when possible).
__global__ void __launch_bounds__(256) k_zero_fill(uint32_t* __restrict__ p, size_t n_words) {
    const size_t n4 = n_words >> 2;
    uint4* p4 = reinterpret_cast<uint4*>(p);
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) p4[i] = make_uint4(0, 0, 0, 0);
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += stride) p[i] = 0;
}

// Several clears in ONE launch (up to FILL_RANGES): with a step's kernels on one stream every launch is on its critical
// path, and a clear of a few KB costs what a clear of a few MB does (~4.5 us).  Same rules as zero_async per range
// (a multiple of 4 bytes), except that a range need only be 4-byte aligned (a view into a flat buffer): the words in front
// of the first 16-byte boundary are stored singly; empty ranges are skipped.
constexpr int FILL_RANGES = 6;
struct FillRanges {
    uint32_t* p[FILL_RANGES];
    unsigned long long words[FILL_RANGES];
};
__global__ void __launch_bounds__(256) k_zero_ranges(FillRanges r) {
    const size_t stride = (size_t)gridDim.x * 256, i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < FILL_RANGES; k++) {
        const size_t n_words = r.words[k];
        size_t head = ((16u - (unsigned)((uintptr_t)r.p[k] & 15u)) & 15u) >> 2;     // words up to the 16-byte boundary
        if (head > n_words) head = n_words;
        const size_t n4 = (n_words - head) >> 2;
        uint4* p4 = reinterpret_cast<uint4*>(r.p[k] + head);
        if (i0 < head) r.p[k][i0] = 0;
        for (size_t i = i0; i < n4; i += stride) p4[i] = make_uint4(0, 0, 0, 0);
        for (size_t i = head + (n4 << 2) + i0; i < n_words; i += stride) r.p[k][i] = 0;
    }
}
inline hipError_t zero_ranges_async(void* const* ptrs, const size_t* bytes, int n, hipStream_t st) {
    FillRanges r;
    size_t most = 0;
    int used = 0;
    for (int k = 0; k < n; k++) {
        if (!ptrs[k] || bytes[k] == 0) continue;
        if (used == FILL_RANGES || (bytes[k] & 3) || ((uintptr_t)ptrs[k] & 3)) return hipErrorInvalidValue;
        r.p[used] = (uint32_t*)ptrs[k];
        r.words[used] = bytes[k] >> 2;
        most = bytes[k] > most ? bytes[k] : most;
        used++;
    }
    if (used == 0) return hipSuccess;
    for (int k = used; k < FILL_RANGES; k++) { r.p[k] = nullptr; r.words[k] = 0; }
    size_t blocks = (most / 16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    LAUNCH("k_zero_fill", k_zero_ranges, dim3((unsigned)blocks), dim3(256), st, r);
    return hipGetLastError();
}

inline hipError_t zero_async(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    if ((bytes & 3) || ((uintptr_t)p & 15)) return hipMemsetAsync(p, 0, bytes, st);     // not used by this library
    const size_t words = bytes >> 2;
    size_t blocks = (words / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    LAUNCH("k_zero_fill", k_zero_fill, dim3((unsigned)blocks), dim3(256), st, (uint32_t*)p, words);
    return hipGetLastError();
}
