// d3m_raster.hip -- extern "C" entry points of libd3m_raster.so (see include/d3m_raster.h).
// gfx950 only.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/d3m_raster.h"
#include "d3m_launch.h"
#include "d3m_aux.h"
#include "d3m_textures.h"
#include "d3m_mesh.h"
#include "d3m_uv.h"
#include "d3m_backward.h"
#include "d3m_device.h"
#include "d3m_edge_grad.h"
#include "d3m_face_major.h"
#include "d3m_forward.h"
#include "d3m_lit.h"
#include "d3m_g2s.h"
#include "d3m_bid.h"
#include "d3m_front.h"
#include <climits>
#include <cstdlib>
#include <atomic>

using namespace d3m;

#define D3M_EXPORT extern "C" __attribute__((visibility("default")))

static thread_local int g_last_hip_error = 0;

static inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return D3M_ERR_LAUNCH;
    }
    return D3M_OK;
}
#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t e__ = (expr);                        \
        if (e__ != hipSuccess) {                        \
            g_last_hip_error = (int)e__;                \
            return D3M_ERR_LAUNCH;                      \
        }                                               \
    } while (0)

static inline unsigned blocks_for(long n, int threads) { return (unsigned)((n + threads - 1) / threads); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

D3M_EXPORT void d3m_timing_enable(int on) {
    std::lock_guard<std::mutex> lk(g_timing_mu);
    g_timing = on != 0;
}
// Synchronises the device, folds the recorded launches into per-kernel (count, total ms) and clears the
// record.  Writes up to max_entries rows; names are static strings.  Returns the number of rows.
D3M_EXPORT int d3m_timing_collect(const char** names, int* counts, float* total_ms, int max_entries) {
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lk(g_timing_mu);
    std::map<std::string, std::pair<int, double>> agg;
    std::map<std::string, const char*> keep;
    for (auto& t : g_timed) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
            auto& e = agg[t.name];
            e.first += 1; e.second += ms;
            keep[t.name] = t.name;
        }
        (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop);
    }
    g_timed.clear();
    int n = 0;
    for (auto& kv : agg) {
        if (n >= max_entries) break;
        names[n] = keep[kv.first]; counts[n] = kv.second.first; total_ms[n] = (float)kv.second.second;
        n++;
    }
    return n;
}

D3M_EXPORT const char* d3m_version(void) { return "d3m_raster 0.1 (gfx950)"; }
// up to six clears as one launch (a fill KERNEL, never a memset node: d3m_launch.h)
D3M_EXPORT int d3m_zero_ranges(void* const* ptrs, const size_t* bytes, int count, d3m_stream_t stream) {
    if (!ptrs || !bytes || count < 0 || count > FILL_RANGES) return D3M_ERR_INVALID;
    hipError_t e = zero_ranges_async(ptrs, bytes, count, (hipStream_t)stream);
    if (e == hipErrorInvalidValue) return D3M_ERR_INVALID;
    HIP_TRY(e);
    return check_launch();
}
D3M_EXPORT int d3m_last_hip_error(void) { return g_last_hip_error; }
D3M_EXPORT const char* d3m_error_string(int code) {
    switch (code) {
        case D3M_OK: return "ok";
        case D3M_ERR_INVALID: return "invalid argument";
        case D3M_ERR_WORKSPACE: return "workspace missing or too small";
        case D3M_ERR_LAUNCH: return "HIP call failed";
        default: return "unknown error";
    }
}

// ---------------------------------------------------------------------------------------------------
// workspace carving for the binned forward
// ---------------------------------------------------------------------------------------------------
static const int KCAP_DEFAULT = 16;   // a face covering more tiles than this is "large"
static const int KCAP_MAX = 64;
static const int RASTER_SMALL_GRID = 32768 * 64 / TILE_PX;  // up to this many tiles (2 M pixels) a tile gets 4 waves instead of 1 (d3m_forward.h)

// the streaming form of the tile pass (d3m_forward.h): waves per XCD for `per` tiles per XCD -- about two thirds as many, so
// that a wave takes one or two tiles, and such that its second tile lies two thirds of the screen away from its first, in
// both directions, whatever view it belongs to: the expensive tiles are the object's, in the middle of every view, and a wave
// that draws two of them is the pass's tail (the same number of waves with the second tile a third of the screen away: no gain)
static inline int raster_stream_waves(int per, int tiles_x, int tiles_y) {
    const int T = tiles_x * tiles_y, shift = (2 * tiles_y / 3) * tiles_x + 2 * tiles_x / 3;
    int w = (2 * per / 3) / T * T + shift;
    if (w < per / 2) w += T;
    return std::max(1, w);
}

struct FwdLayout {
    size_t zero_bytes;   // prefix that must be zeroed per call
    size_t off_count, off_cursor, off_big_count, off_alloc, off_offset, off_rect, off_big, off_pairs;
    size_t fixed_bytes;  // everything but pairs
};

static FwdLayout fwd_layout(int B, int F, int S) {
    const int tiles_x = (S + TILE_W - 1) / TILE_W, tiles_y = (S + TILE_H - 1) / TILE_H;
    const size_t nt = (size_t)B * tiles_x * tiles_y;
    FwdLayout L;
    size_t o = 0;
    L.off_count = o;     o += align_up(nt * 4, 256);
    L.off_cursor = o;    o += align_up(nt * 4, 256);
    L.off_big_count = o; o += align_up((size_t)B * 4, 256);
    L.off_alloc = o;     o += 256;
    L.zero_bytes = o;
    L.off_offset = o;    o += align_up(nt * 4, 256);
    L.off_rect = o;      o += align_up((size_t)B * F * 8, 256);
    L.off_big = o;       o += align_up((size_t)B * F * 4, 256);
    L.off_pairs = o;
    L.fixed_bytes = o;
    return L;
}

// what coverage by bidding keeps in the workspace instead (run_forward_mesh): z-buffer | big-face count | big-face list
static size_t bid_workspace_bytes(int B, int F, int S) {
    return align_up((size_t)B * S * S * 8, 256) + 256 + align_up((size_t)B * F * 4, 256);
}
// z-buffer | big-face count: what a bidding launch needs zeroed
static size_t bid_clear_bytes(int B, int F, int S) { (void)F; return align_up((size_t)B * S * S * 8, 256) + 256; }
D3M_EXPORT size_t d3m_forward_workspace_bytes(int B, int F, int S) {
    if (B <= 0 || F <= 0 || S <= 0) return 0;
    return std::max(fwd_layout(B, F, S).fixed_bytes + (size_t)KCAP_DEFAULT * B * F * 4, bid_workspace_bytes(B, F, S));
}
D3M_EXPORT size_t d3m_forward_workspace_min_bytes(int B, int F, int S) {
    if (B <= 0 || F <= 0 || S <= 0) return 0;
    return fwd_layout(B, F, S).fixed_bytes + (size_t)B * F * 4;
}

static int make_bins(BinBuffers& bb, int B, int F, int S, void* ws, size_t ws_bytes) {
    if (!ws) return D3M_ERR_WORKSPACE;
    const FwdLayout L = fwd_layout(B, F, S);
    if (ws_bytes < L.fixed_bytes + (size_t)B * F * 4) return D3M_ERR_WORKSPACE;
    long kcap = (long)((ws_bytes - L.fixed_bytes) / ((size_t)B * F * 4));
    if (kcap > KCAP_MAX) kcap = KCAP_MAX;
    // tile_offset / the cursors index `pairs` with an int
    while (kcap > 1 && kcap * (long)B * F >= (1l << 31)) kcap--;
    if ((long)B * F >= (1l << 31)) return D3M_ERR_INVALID;
    char* p = (char*)ws;
    bb.B = B; bb.F = F; bb.S = S;
    bb.tiles_x = (S + TILE_W - 1) / TILE_W;
    bb.tiles_y = (S + TILE_H - 1) / TILE_H;
    bb.T = bb.tiles_x * bb.tiles_y;
    bb.kcap = (int)kcap;
    bb.tile_count = (int*)(p + L.off_count);
    bb.tile_cursor = (int*)(p + L.off_cursor);
    bb.big_count = (int*)(p + L.off_big_count);
    bb.alloc_cursor = (int*)(p + L.off_alloc);
    bb.tile_offset = (int*)(p + L.off_offset);
    bb.rect = p + L.off_rect;            // (laid out for the 8-byte form; the 4-byte form uses the first half)
    bb.big_list = (int*)(p + L.off_big);
    bb.pairs = (int*)(p + L.off_pairs);
    return D3M_OK;
}

// binning workgroup: 1024 lanes (fewest global atomics on the tile counters) once that still fills the chip
static inline unsigned bin_threads(long lanes) { return lanes >= 1024l * 1024 ? BIN_THREADS : BIN_THREADS_SMALL; }

// the faces' tile rectangles in four bytes (d3m_forward.h Rect32) where the raster allows it
static inline bool rect32_ok(const BinBuffers& bb) { return bb.tiles_x <= 1024 && bb.tiles_y <= 1024 && bb.kcap <= 64; }
// the list-filling pass: more face (pairs) per thread once the launch stays large (see k_bin_fill)
template <bool PAIRED, class RECT>
static inline void launch_bin_fill_r(const BinBuffers& bb, long lanes, hipStream_t st) {
    const unsigned th = bin_threads(lanes);
    if (lanes >= 4l * 1024 * 1024)
        LAUNCH("k_bin_fill", (k_bin_fill<PAIRED, 4, RECT>), dim3(blocks_for(lanes, th * 4)), dim3(th), st, bb);
    else if (lanes >= 2l * 1024 * 1024)
        LAUNCH("k_bin_fill", (k_bin_fill<PAIRED, 2, RECT>), dim3(blocks_for(lanes, th * 2)), dim3(th), st, bb);
    else
        LAUNCH("k_bin_fill", (k_bin_fill<PAIRED, 1, RECT>), dim3(blocks_for(lanes, th)), dim3(th), st, bb);
}
template <bool PAIRED>
static inline void launch_bin_fill(const BinBuffers& bb, long lanes, hipStream_t st) {
    if (rect32_ok(bb)) launch_bin_fill_r<PAIRED, Rect32>(bb, lanes, st);
    else launch_bin_fill_r<PAIRED, Rect64>(bb, lanes, st);
}
template <class FS, bool PAIRED, class... Args>
static inline void launch_bin_count(const BinBuffers& bb, long lanes, hipStream_t st, FS fs, Args... args) {
    const unsigned th = bin_threads(lanes);
    if (rect32_ok(bb)) LAUNCH("k_bin_count", (k_bin_count<FS, PAIRED, Rect32>), dim3(blocks_for(lanes, th)), dim3(th), st, fs, bb, args...);
    else LAUNCH("k_bin_count", (k_bin_count<FS, PAIRED, Rect64>), dim3(blocks_for(lanes, th)), dim3(th), st, fs, bb, args...);
}

// grid of the per-pixel backward kernels (they stride): `sparse` = only the pixels of a few large faces have work, and
// normally there are none
static inline unsigned px_grid(long n, bool sparse) {
    const long b = (n + 255) / 256, cap = sparse ? 2048 : 16384;
    return (unsigned)(b < cap ? b : cap);
}

// COVERAGE BY BIDDING (d3m_bid.h) instead of binning -- same maps, bit for bit -- where it is the faster of the two
// (DESIGN.md 4.1, A/B on one box): meshes of (sub-)pixel triangles, fewer than 1.5 raster pixels each (the 1 M-triangle
// mesh at 1024^2: 0.95 ms per 8 views against 1.70), and batches of at most 65536 tiles, which the tile pass cannot fill
// the chip with as well (8 views of the 100 k mesh at 512^2: 0.145 ms against 0.21; 32 views: 0.47 against 0.45, so the big
// batches of ordinary meshes stay binned).  Its z-buffer and big-face list live in the forward
// workspace: a workspace too small for them means binning.  D3M_BID=1 / 0 forces / forbids it (measurements).
static const long BID_MAX_TILES = 65536;     // (8 x 8-pixel blocks) 16 views at 512^2: +1 % there, +1.5 % at 12, -2 % at 24 (A/B, round 3)
// which form of coverage the forward takes: -1 chosen per launch (above), 0 binned only, 1 bidding wherever its
// workspace fits.  Process-wide; D3M_BID=0 / 1 in the environment sets the initial value (measurements), and
// d3m_set_coverage_form() changes it between launches (the parity tests run every scene in both forms).
static std::atomic<int> g_coverage_form{-2};        // -2: not yet read from the environment
static int coverage_form() {
    int f = g_coverage_form.load(std::memory_order_relaxed);
    if (f == -2) {
        const char* e = getenv("D3M_BID");
        f = e ? (e[0] == '1' ? 1 : e[0] == '0' ? 0 : -1) : -1;
        int expected = -2;
        if (!g_coverage_form.compare_exchange_strong(expected, f)) f = expected;
    }
    return f;
}
D3M_EXPORT int d3m_set_coverage_form(int form) {
    if (form < -1 || form > 1) return D3M_ERR_INVALID;
    (void)coverage_form();
    g_coverage_form.store(form);
    return D3M_OK;
}
D3M_EXPORT int d3m_get_coverage_form(void) { return coverage_form(); }
D3M_EXPORT int d3m_set_deterministic(int on) {
    if (on != 0 && on != 1) return D3M_ERR_INVALID;
    (void)deterministic_mode();
    g_deterministic.store(on);
    return D3M_OK;
}
D3M_EXPORT int d3m_get_deterministic(void) { return deterministic_mode() ? 1 : 0; }
// Which form of coverage a launch takes (d3m_set_coverage_form: forced).  Sub-pixel triangles (more than two per three
// raster pixels: BASELINE config 5) bid whatever the batch; big batches of ordinary meshes (more than BID_MAX_TILES blocks
// of 8 x 8 pixels: the headline's 32 views) go through per-tile lists; small batches bid -- UNLESS the mesh is coarse (round
// 5): bidding hands a face's rows to the lanes of ONE wave, and a few thousand triangles of a hundred pixels each are a
// few dozen waves with everything to do (a 2 450-triangle mesh @512^2: k_bid_faces 237 us, the whole binned forward 95;
// 722 triangles: 341 against 175), where a tile's wave takes a tile-filling face in one step.  Measured crossover at ONE
// view: ~8 raster pixels per input triangle (19 602 triangles @512^2, 13 px each: lists 15 us ahead; 32 258, 8 px: equal;
// 53 138, 5 px: bidding 10 us ahead).
static bool big_batch(int B, long triangles, int S) {
    const int blocks_x = (S + 7) / 8;            // (the threshold is in blocks of 8 x 8 pixels, whatever the tile pass's tiles)
    return !((double)S * S < 1.5 * (double)triangles) && (long)B * blocks_x * blocks_x > BID_MAX_TILES;
}
static bool bidding_preferred(int B, long triangles, int S) {
    const int form = coverage_form();
    if (form >= 0) return form == 1;
    if ((double)S * S < 1.5 * (double)triangles) return true;
    // (the more views, the more waves bidding has to work with: the crossover moves from ~10 raster pixels per triangle at
    //  one view to ~25 at eight -- 19 602 triangles @512^2: lists 15 us ahead at one view, bidding 22 us ahead at eight)
    //  -- provided the batch has the faces for it: 8 views of 722 triangles @128^2 are 90 waves, 0.169 ms bidding, 0.12 binned)
    const double px_per_tri = (double)B * (double)triangles < 65536.0 ? 10.0 : std::min(10.0 + 2.0 * (B - 1), 32.0);
    return !big_batch(B, triangles, S) && (double)S * S <= px_per_tri * (double)triangles;
}
// Whether a launch is a "big batch" in that sense ("every kernel fills the chip by itself"), independent of the form of
// coverage a coarse mesh takes: what multiview.MultiViewFit keys its split exchange on (the lit render node took it for "run
// on one stream" in rounds 4-5; since round 6 its branches run beside each other at every size, rasterize._serial_branches).
D3M_EXPORT int d3m_forward_big_batch(int batch_size, int num_triangles, int image_size) {
    if (batch_size <= 0 || num_triangles <= 0 || image_size <= 0) return -1;
    return big_batch(batch_size, num_triangles, image_size) ? 1 : 0;
}
static bool bidding_wanted(int B, long triangles, int S, const void* ws, size_t ws_bytes, int F) {
    return bidding_preferred(B, triangles, S) && S <= 8192 && ws && ws_bytes >= bid_workspace_bytes(B, F, S);
}
D3M_EXPORT int d3m_forward_coverage_form(int batch_size, int num_triangles, int image_size) {
    if (batch_size <= 0 || num_triangles <= 0 || image_size <= 0) return -1;
    return bidding_preferred(batch_size, num_triangles, image_size) && image_size <= 8192 ? 1 : 0;
}
// FS: the faces as the caller has them (indexed mesh: faces_dense receives the dense copy; dense: faces_dense IS the input)
template <class FS, bool PAIRED>
static int run_bidding(FS fs, float* faces_dense, float* faces_dense_out, float* faces_inv, int B, int F, int S, float near,
                       float far, RasterOut out, void* ws, hipStream_t st, bool cleared = false, ModeOut mo = ModeOut{}) {
    const size_t zbytes = align_up((size_t)B * S * S * 8, 256);
    unsigned long long* zbuf = (unsigned long long*)ws;
    int* big_count = (int*)((char*)ws + zbytes);
    int* big_list = (int*)((char*)ws + zbytes + 256);
    if (!cleared) HIP_TRY(zero_async(zbuf, bid_clear_bytes(B, F, S), st));
    constexpr int PW = 64;          // a lane per face (pair) for the set-up: the boxes are a few rows each
    const long units = (long)B * (PAIRED ? F / 2 : F);
    LAUNCH("k_bid_faces", (k_bid_faces<FS, PW, PAIRED>), dim3(blocks_for(units, 4 * PW)), dim3(256), st, fs, zbuf, faces_dense_out,
           B, S, near, far, out.marks, out.marks_count, big_list, big_count, faces_inv);
    LAUNCH("k_bid_big", k_bid_big, dim3(128, (unsigned)((S + 255) / 256)), dim3(256), st, DenseFaces{faces_dense, F}, zbuf,
           (const int*)big_list, (const int*)big_count, S, near, far);
    LAUNCH("k_bid_resolve", k_bid_resolve, dim3(blocks_for((long)B * S * S, 256)), dim3(256), st, DenseFaces{faces_dense, F},
           (const unsigned long long*)zbuf, out, B, S, near, far, mo);
    return check_launch();
}

template <class FS>
static int run_forward(FS fs, int B, int F, int S, float near, float far, RasterOut out, float* faces_inv, void* ws,
                       size_t ws_bytes, hipStream_t st) {
    if (S > 8 * 65535) return D3M_ERR_INVALID;
    // (dense faces: a caller's fill_back copies are faces of their own here, two per triangle)
    if (bidding_wanted(B, F / 2, S, ws, ws_bytes, F))
        return run_bidding<FS, false>(fs, const_cast<float*>(fs.faces), nullptr, faces_inv, B, F, S, near, far, out, ws, st);
    BinBuffers bb;
    int rc = make_bins(bb, B, F, S, ws, ws_bytes);
    if (rc) return rc;
    HIP_TRY(zero_async(ws, fwd_layout(B, F, S).zero_bytes, st));
    const long nf = (long)B * F;
    launch_bin_count<FS, false>(bb, nf, st, fs, faces_inv, (float*)nullptr, (unsigned char*)nullptr, (int*)nullptr);
    LAUNCH("k_bin_alloc", k_bin_alloc, dim3(blocks_for((long)B * bb.T, BIN_ALLOC_THREADS)), dim3(BIN_ALLOC_THREADS), st, bb);
    launch_bin_fill<false>(bb, nf, st);
    const int n_tiles = B * bb.T;
    const int per = (n_tiles + 7) / 8;
    if (n_tiles <= RASTER_SMALL_GRID)
        LAUNCH("k_raster_tiles", (k_raster_tiles<FS, 4>), dim3(per * 8), dim3(256), st, fs, bb, out, near, far, ModeOut{});
    else
        LAUNCH("k_raster_tiles", (k_raster_tiles<FS, 1, true>), dim3(raster_stream_waves(per, bb.tiles_x, bb.tiles_y) * 8), dim3(64), st, fs, bb, out, near, far, ModeOut{});
    return check_launch();
}

// The same with the faces of an indexed mesh: the first pass reads them through the indices and leaves the dense
// copy (front-facing faces only) that the tile pass and every later operator use.
// cleared: the caller has zeroed the workspace's first d3m_forward_clear_bytes() bytes (of the form this launch takes)
static int run_forward_mesh(IndexedFaces ifs, float* faces_out, int B, int S, float near, float far, RasterOut out, void* ws,
                            size_t ws_bytes, hipStream_t st, bool cleared = false, ModeOut mo = ModeOut{}) {
    // out.marks (optional): zeroed by the first pass, set by the tile pass
    if (S > 8 * 65535) return D3M_ERR_INVALID;
    const int F = ifs.num_faces();
    if (bidding_wanted(B, ifs.Ft, S, ws, ws_bytes, F)) {
        if (ifs.fill_back) return run_bidding<IndexedFaces, true>(ifs, faces_out, faces_out, nullptr, B, F, S, near, far, out, ws, st, cleared, mo);
        return run_bidding<IndexedFaces, false>(ifs, faces_out, faces_out, nullptr, B, F, S, near, far, out, ws, st, cleared, mo);
    }
    BinBuffers bb;
    int rc = make_bins(bb, B, F, S, ws, ws_bytes);
    if (rc) return rc;
    if (!cleared) HIP_TRY(zero_async(ws, fwd_layout(B, F, S).zero_bytes, st));
    const long nf = (long)B * F;
#ifdef D3M_DEV_SKIP
    {   // developer builds: D3M_ABL_NO_DENSE -- from the third call on the binning pass stops writing the dense face copy
        // (the buffer keeps the earlier calls' contents: same mesh every step in bench.py), to time the pass without it
        static const char* abl = getenv("D3M_ABL_NO_DENSE");
        static int calls = 0;
        if (abl && ++calls > 2) {
            launch_bin_count<IndexedFaces, true>(bb, nf / 2, st, ifs, (float*)nullptr, (float*)nullptr, out.marks, out.marks_count);
            goto counted;
        }
    }
#endif
    if (ifs.fill_back)      // one lane per index triple, both copies
        launch_bin_count<IndexedFaces, true>(bb, nf / 2, st, ifs, (float*)nullptr, faces_out, out.marks, out.marks_count);
    else
        launch_bin_count<IndexedFaces, false>(bb, nf, st, ifs, (float*)nullptr, faces_out, out.marks, out.marks_count);
#ifdef D3M_DEV_SKIP
counted:
#endif
    LAUNCH("k_bin_alloc", k_bin_alloc, dim3(blocks_for((long)B * bb.T, BIN_ALLOC_THREADS)), dim3(BIN_ALLOC_THREADS), st, bb);
    if (ifs.fill_back) launch_bin_fill<true>(bb, nf / 2, st);
    else launch_bin_fill<false>(bb, nf, st);
    const int n_tiles = B * bb.T;
    const int per = (n_tiles + 7) / 8;
    DenseFaces fs{faces_out, F};
    const bool mode = mo.alpha_map || mo.alpha_out || mo.depth_out;
    if (n_tiles <= RASTER_SMALL_GRID) {
        if (mode) LAUNCH("k_raster_tiles", (k_raster_tiles<DenseFaces, 4, false, true>), dim3(per * 8), dim3(256), st, fs, bb, out, near, far, mo);
        else LAUNCH("k_raster_tiles", (k_raster_tiles<DenseFaces, 4>), dim3(per * 8), dim3(256), st, fs, bb, out, near, far, mo);
    } else {
        const dim3 grid(raster_stream_waves(per, bb.tiles_x, bb.tiles_y) * 8);
        if (mode) LAUNCH("k_raster_tiles", (k_raster_tiles<DenseFaces, 1, true, true>), grid, dim3(64), st, fs, bb, out, near, far, mo);
        else LAUNCH("k_raster_tiles", (k_raster_tiles<DenseFaces, 1, true>), grid, dim3(64), st, fs, bb, out, near, far, mo);
    }
    return check_launch();
}

D3M_EXPORT size_t d3m_visibility_bytes(int batch_size, int num_faces) {
    if (batch_size <= 0 || num_faces <= 0) return 0;
    return visibility_bytes((long)batch_size * num_faces);
}

D3M_EXPORT int d3m_visibility(const int32_t* face_index_map, void* visibility, size_t visibility_size, int batch_size,
                              int num_faces, int image_size, d3m_stream_t stream) {
    if (!visibility || batch_size <= 0 || num_faces <= 0 || image_size <= 0) return D3M_ERR_INVALID;
    if (visibility_size < d3m_visibility_bytes(batch_size, num_faces)) return D3M_ERR_WORKSPACE;
    const VisibilityView v = visibility_view(visibility, (long)batch_size * num_faces);
    HIP_TRY(run_visibility(face_index_map, v, batch_size, num_faces, image_size, (hipStream_t)stream));
    return check_launch();
}

// Index triples of an entry point: tri [tri_batch,Ft,3] with tri_batch = 1 or B -- or tri == NULL and tri_batch = -W:
// the implicit topology of a depth map's grid mesh with W vertices per row (d3m_device.h tri_ids), V = H*W vertices
// and Ft = 2 (H-1)(W-1) triangles.  Returns false when the combination is invalid; grid_w = 0 for explicit indices.
static bool tri_source_ok(const int32_t* tri, int tri_batch, int B, int num_vertices, int num_tri, int& grid_w) {
    grid_w = 0;
    if (tri) return tri_batch == 1 || tri_batch == B;
    if (tri_batch >= -1) return false;
    grid_w = -tri_batch;
    if (num_vertices % grid_w) return false;
    const int grid_h = num_vertices / grid_w;
    return grid_h >= 2 && num_tri == 2 * (grid_h - 1) * (grid_w - 1);
}

static int to_vertex_target(const d3m_vertex_target* h, int num_faces, VertexTarget& vt) {
    vt = VertexTarget{nullptr, nullptr, 0, 0, 1};
    if (!h) return D3M_OK;
    if (!h->grad_vertices || h->num_vertices <= 0 || h->num_tri <= 0) return D3M_ERR_INVALID;
    if ((h->fill_back ? 2 : 1) * h->num_tri != num_faces) return D3M_ERR_INVALID;
    int grid_w;
    if (!h->tri && !tri_source_ok(nullptr, h->tri_batch, 1, h->num_vertices, h->num_tri, grid_w)) return D3M_ERR_INVALID;
    vt = VertexTarget{h->grad_vertices, h->tri, h->num_vertices, h->num_tri, h->tri ? h->tri_batch : 1, h->tri ? 0 : -h->tri_batch};
    return D3M_OK;
}

// ---------------------------------------------------------------------------------------------------
// A. the five operators
// ---------------------------------------------------------------------------------------------------
D3M_EXPORT int d3m_forward_face_index_map(const float* faces, int32_t* face_index_map, float* weight_map,
                                          float* depth_map, float* face_inv_map, float* faces_inv, int batch_size,
                                          int num_faces, int image_size, float near, float far, int return_rgb,
                                          int return_alpha, int return_depth, void* workspace, size_t workspace_bytes,
                                          d3m_stream_t stream) {
    (void)return_rgb; (void)return_alpha;
    if (!faces || !face_index_map || !weight_map || !depth_map || batch_size <= 0 || num_faces <= 0 || image_size <= 0)
        return D3M_ERR_INVALID;
    DenseFaces fs{faces, num_faces};
    RasterOut out{face_index_map, weight_map, depth_map, return_depth ? face_inv_map : nullptr};
    return run_forward(fs, batch_size, num_faces, image_size, near, far, out, faces_inv, workspace, workspace_bytes,
                       (hipStream_t)stream);
}

D3M_EXPORT int d3m_forward_face_index_map_mesh(const float* vertices, const int32_t* tri, int tri_batch, int num_vertices,
                                               int num_tri, int fill_back, float* faces_out, int32_t* face_index_map,
                                               float* weight_map, float* depth_map, float* face_inv_map, int batch_size,
                                               int image_size, float near, float far, void* workspace,
                                               size_t workspace_bytes, void* visibility, size_t visibility_size,
                                               int flags, d3m_stream_t stream) {
    return d3m_forward_face_index_map_mesh_modes(vertices, tri, tri_batch, num_vertices, num_tri, fill_back, faces_out,
                                                 face_index_map, weight_map, depth_map, face_inv_map, batch_size, image_size,
                                                 near, far, workspace, workspace_bytes, visibility, visibility_size, nullptr,
                                                 nullptr, nullptr, flags, stream);
}
// ... which can also write, in its last pass, the output images of the renderer's silhouette / depth modes without
// anti-aliasing (d3m_output_epilogue's work for those modes: alpha_map [B,S,S] internal layout, alpha_out / depth_out
// [B,S,S] with the rows reversed; any may be NULL)
D3M_EXPORT int d3m_forward_face_index_map_mesh_modes(const float* vertices, const int32_t* tri, int tri_batch, int num_vertices,
                                                     int num_tri, int fill_back, float* faces_out, int32_t* face_index_map,
                                                     float* weight_map, float* depth_map, float* face_inv_map, int batch_size,
                                                     int image_size, float near, float far, void* workspace,
                                                     size_t workspace_bytes, void* visibility, size_t visibility_size,
                                                     float* alpha_map, float* alpha_out, float* depth_out, int flags,
                                                     d3m_stream_t stream) {
    if (!vertices || !faces_out || !face_index_map || !weight_map || !depth_map || batch_size <= 0 ||
        num_vertices <= 0 || num_tri <= 0 || image_size <= 0)
        return D3M_ERR_INVALID;
    int grid_w;
    if (!tri_source_ok(tri, tri_batch, batch_size, num_vertices, num_tri, grid_w)) return D3M_ERR_INVALID;
    IndexedFaces ifs{vertices, tri, num_vertices, num_tri, tri ? tri_batch : 1, fill_back ? 1 : 0, batch_size, grid_w};
    RasterOut out{face_index_map, weight_map, depth_map, face_inv_map, nullptr};
    if (visibility) {       // the first step of d3m_visibility rides along: finish it with d3m_visibility(NULL, ...)
        const long nf = (long)batch_size * ifs.num_faces();
        if (visibility_size < visibility_bytes(nf)) return D3M_ERR_WORKSPACE;
        out.marks = visibility_view(visibility, nf).marks;
        out.marks_count = visibility_view(visibility, nf).count;
    }
    if (face_inv_map && (alpha_map || alpha_out || depth_out)) return D3M_ERR_INVALID;      // one or the other
    ModeOut mo;
    mo.alpha_map = alpha_map; mo.alpha_out = alpha_out; mo.depth_out = depth_out;
    return run_forward_mesh(ifs, faces_out, batch_size, image_size, near, far, out, workspace, workspace_bytes,
                            (hipStream_t)stream, (flags & D3M_PRECLEARED) != 0, mo);
}
// The leading bytes of the forward workspace that d3m_forward_face_index_map_mesh zeroes in front of its kernels -- for the
// form of coverage THAT launch takes (it depends on the workspace's size too) --, or 0 for an invalid / too small workspace.
D3M_EXPORT size_t d3m_forward_clear_bytes(int batch_size, int num_tri, int fill_back, int image_size, size_t workspace_bytes) {
    if (batch_size <= 0 || num_tri <= 0 || image_size <= 0) return 0;
    const int F = (fill_back ? 2 : 1) * num_tri;
    if (bidding_wanted(batch_size, num_tri, image_size, (const void*)1, workspace_bytes, F)) return bid_clear_bytes(batch_size, F, image_size);
    const FwdLayout L = fwd_layout(batch_size, F, image_size);
    return workspace_bytes >= L.fixed_bytes + (size_t)batch_size * F * 4 ? L.zero_bytes : 0;
}

D3M_EXPORT int d3m_forward_texture_sampling(const float* faces, const float* textures, const int32_t* face_index_map,
                                            const float* weight_map, const float* depth_map, float* rgb_map,
                                            int32_t* sampling_index_map, float* sampling_weight_map, int batch_size,
                                            int num_faces, int image_size, int texture_size, float eps,
                                            d3m_stream_t stream) {
    if (!faces || !textures || !face_index_map || !weight_map || !depth_map || !rgb_map || batch_size <= 0 ||
        num_faces <= 0 || image_size <= 0 || texture_size <= 0)
        return D3M_ERR_INVALID;
    const long n = (long)batch_size * image_size * image_size;
    LAUNCH("k_texture_sampling", k_texture_sampling, dim3(blocks_for(n, 256)), dim3(256), (hipStream_t)stream, faces, textures,
                       face_index_map, weight_map, depth_map, rgb_map, sampling_index_map, sampling_weight_map,
                       batch_size, num_faces, image_size, texture_size, eps);
    return check_launch();
}

// the scalar factors the unscaled gradient maps of a fused objective still lack (NULL: the maps are final)
static GradScale to_grad_scale(const d3m_fit_targets* unscaled, int image_size) {
    if (!unscaled) return GradScale{nullptr, nullptr, 0.0f, 0, nullptr};
    return GradScale{unscaled->scratch, unscaled->grad_loss, (float)((long)image_size * image_size),
                     unscaled->edge_grad ? 1 : 0, unscaled->mask_sum};
}

D3M_EXPORT size_t d3m_backward_pixel_map_workspace_bytes(int batch_size, int num_faces, int image_size) {
    return edge_grad_workspace_bytes(batch_size, num_faces, image_size);
}

D3M_EXPORT int d3m_backward_pixel_map(const float* faces, const int32_t* face_index_map, const float* rgb_map,
                                      const float* alpha_map, const float* grad_rgb_map, const float* grad_alpha_map,
                                      float* grad_faces, int batch_size, int num_faces, int image_size, float eps,
                                      int return_rgb, int return_alpha, void* workspace, size_t workspace_bytes,
                                      const d3m_vertex_target* vertex_target, void* visibility, void* edge_plan,
                                      size_t edge_plan_size, const d3m_fit_targets* unscaled, d3m_stream_t stream) {
    if (edge_plan && !visibility) return D3M_ERR_INVALID;      // a plan indexes the list of its visibility blob
    // alpha only, the OUTPUT image's gradient as it is (flags & D3M_GRAD_OF_OUTPUT_IMAGE; | D3M_FIT_POOLED: of the 2x2-pooled image)
    const bool of_image = unscaled && !unscaled->scratch && !unscaled->edge_grad &&
                          (unscaled->flags & D3M_GRAD_OF_OUTPUT_IMAGE) && unscaled->grad_alpha_map;
    if (of_image && (return_rgb || !return_alpha || ((unscaled->flags & D3M_FIT_POOLED) && (image_size & 1))))
        return D3M_ERR_INVALID;
    if (unscaled && !unscaled->scratch && !unscaled->edge_grad && !of_image) return D3M_ERR_INVALID;    // (records without a scratch: final)
    if (!faces || !face_index_map || (!grad_faces && !vertex_target) || batch_size <= 0 || num_faces <= 0 ||
        image_size <= 0)
        return D3M_ERR_INVALID;
    VertexTarget vt;
    if (int rc = to_vertex_target(vertex_target, num_faces, vt)) return rc;
    const bool records = unscaled && unscaled->edge_grad;      // the gradient maps arrive as per-pixel records
    if (records && !(unscaled->edge_dot && unscaled->edge_nz_lo_inv && unscaled->edge_nz_hi1)) return D3M_ERR_INVALID;
    if (return_rgb && (!rgb_map || (!grad_rgb_map && !records))) return D3M_ERR_INVALID;
    if (return_alpha && (!alpha_map || (!grad_alpha_map && !records && !of_image))) return D3M_ERR_INVALID;
    if (!return_rgb && !return_alpha) return D3M_OK;    // rasterize.py:200-201
    DenseFaces fs{faces, num_faces};
    PixelMaps m{face_index_map, rgb_map, alpha_map, grad_rgb_map, grad_alpha_map, image_size, return_rgb != 0,
                return_alpha != 0};
    VisibilityView vis;
    if (visibility) vis = visibility_view(visibility, (long)batch_size * num_faces);
    EdgeRecords rec{nullptr, nullptr, nullptr, nullptr, nullptr};
    if (records)
        rec = EdgeRecords{(const float4*)unscaled->edge_grad, (const float2*)unscaled->edge_dot, unscaled->edge_nz_lo_inv,
                          unscaled->edge_nz_hi1, unscaled->grad_loss};
    if (of_image) {
        rec.ga_img = unscaled->grad_alpha_map;
        rec.img_aa = (unscaled->flags & D3M_FIT_POOLED) ? 1 : 0;
    }
    return run_edge_grad(fs, m, grad_faces, vt, visibility ? &vis : nullptr, edge_plan, edge_plan_size, rec,
                         to_grad_scale(of_image ? nullptr : unscaled, image_size), batch_size, eps, workspace, workspace_bytes,
                         (hipStream_t)stream, &g_last_hip_error);
}

D3M_EXPORT size_t d3m_backward_pixel_map_workspace_min_bytes(int batch_size, int num_faces, int image_size) {
    return edge_grad_workspace_min_bytes(batch_size, num_faces, image_size);
}

D3M_EXPORT size_t d3m_edge_plan_bytes(int batch_size, int num_faces, int image_size) {
    if (batch_size <= 0 || num_faces <= 0 || image_size <= 0) return 0;
    return edge_plan_bytes(batch_size, num_faces, image_size);
}
D3M_EXPORT size_t d3m_edge_plan_min_bytes(int batch_size, int num_faces, int image_size) {
    if (batch_size <= 0 || num_faces <= 0 || image_size <= 0) return 0;
    return edge_plan_min_bytes(batch_size, num_faces, image_size);
}
D3M_EXPORT size_t d3m_edge_plan_extents_offset(int batch_size, int num_faces, int image_size, size_t* bytes_each) {
    if (batch_size <= 0 || num_faces <= 0 || image_size <= 0) return 0;
    if (bytes_each) *bytes_each = eg_align((size_t)batch_size * 2 * image_size * 4);
    return edge_plan_layout(batch_size, num_faces, image_size).off_extents;
}
D3M_EXPORT size_t d3m_edge_plan_clear_bytes(int batch_size, int num_faces, int image_size) {
    if (batch_size <= 0 || num_faces <= 0 || image_size <= 0) return 0;
    return edge_plan_layout(batch_size, num_faces, image_size).zero_bytes;
}
D3M_EXPORT int d3m_edge_plan(const float* faces, const int32_t* face_index_map, void* visibility, void* edge_plan,
                             size_t edge_plan_size, int batch_size, int num_faces, int image_size, int flags,
                             d3m_stream_t stream) {
    if (!faces || !face_index_map || !visibility || !edge_plan || batch_size <= 0 || num_faces <= 0 || image_size <= 0) return D3M_ERR_INVALID;
    if (image_size > 65535 || num_faces > (1 << 25) || (long)batch_size * 2 * image_size >= (1l << 31)) return D3M_ERR_INVALID;
    const VisibilityView vis = visibility_view(visibility, (long)batch_size * num_faces);
    EdgePlan w;
    if (!edge_plan_view(edge_plan, edge_plan_size, vis, batch_size, num_faces, image_size, w)) return D3M_ERR_WORKSPACE;
    DenseFaces fs{faces, num_faces};
    HIP_TRY(run_edge_plan(fs, face_index_map, w, edge_plan, batch_size, image_size, (hipStream_t)stream,
                          (flags & D3M_PRECLEARED) != 0));
    return check_launch();
}

// scratch of the gathered (face-major) backward passes: one int per face
D3M_EXPORT size_t d3m_backward_faces_workspace_bytes(int batch_size, int num_faces) {
    if (batch_size <= 0 || num_faces <= 0) return 0;
    return (size_t)batch_size * num_faces * 4;
}

template <class FS>
static int run_backward_textures(FS fs, const float* faces_dummy, const int32_t* face_index_map, const float* sw,
                                 const int32_t* si, const float* grad_rgb_map, float* grad_textures, int B, int F, int S,
                                 int ts, int* flags, bool flags_ready, hipStream_t st) {
    (void)faces_dummy;
    const long n = (long)B * S * S, nf = (long)B * F;
    if (flags && ts == 2) {
        if (!flags_ready) {
            HIP_TRY(zero_async(flags, (size_t)nf * 4, st));
            LAUNCH("k_mark_visible", k_mark_visible, dim3(blocks_for(n, 256)), dim3(256), st, face_index_map, flags, B, F, S);
        }
        LAUNCH("k_backward_textures_faces", k_backward_textures_faces<FS>, dim3(blocks_for(nf, FM_FACES_PER_BLOCK)), dim3(256), st, fs,
               face_index_map, sw, si, grad_rgb_map, grad_textures, flags, B, S);
        LAUNCH("k_backward_textures", k_backward_textures, dim3(blocks_for(n, 256)), dim3(256), st, face_index_map, sw, si,
               grad_rgb_map, grad_textures, B, F, S, ts, (const int*)flags);
    } else {
        LAUNCH("k_backward_textures", k_backward_textures, dim3(blocks_for(n, 256)), dim3(256), st, face_index_map, sw, si,
               grad_rgb_map, grad_textures, B, F, S, ts, (const int*)nullptr);
    }
    return check_launch();
}

template <class FS>
static int run_backward_depth(FS fs, const float* depth_map, const int32_t* face_index_map, const float* face_inv_map,
                              const float* weight_map, const float* grad_depth_map, float* grad_faces, int B, int S,
                              int* flags, bool flags_ready, hipStream_t st) {
    const int F = fs.num_faces();
    const long n = (long)B * S * S, nf = (long)B * F;
    if (flags) {
        if (!flags_ready) {
            HIP_TRY(zero_async(flags, (size_t)nf * 4, st));
            LAUNCH("k_mark_visible", k_mark_visible, dim3(blocks_for(n, 256)), dim3(256), st, face_index_map, flags, B, F, S);
        }
        LAUNCH("k_backward_depth_faces", k_backward_depth_faces<FS>, dim3(blocks_for(nf, FM_FACES_PER_BLOCK)), dim3(256), st, fs, depth_map,
               face_index_map, weight_map, grad_depth_map, grad_faces, flags, B, S, (const int*)nullptr, (const int*)nullptr,
               VertexTarget{nullptr, nullptr, 0, 0, 1}, (int*)nullptr, 0,
               deterministic_mode() ? INT_MAX : FM_MAX_BBOX_AREA);      // (deterministic: no per-pixel atomic fallback)
    }
    LAUNCH("k_backward_depth_map", k_backward_depth_map<FS>, dim3(px_grid(n, false)), dim3(256), st, fs, depth_map,
           face_index_map, face_inv_map, weight_map, grad_depth_map, grad_faces, B, S, (const int*)flags,
           VertexTarget{nullptr, nullptr, 0, 0, 1});
    return check_launch();
}

D3M_EXPORT int d3m_backward_textures(const float* faces, const int32_t* face_index_map, const float* sampling_weight_map,
                                     const int32_t* sampling_index_map, const float* grad_rgb_map, float* grad_textures,
                                     int batch_size, int num_faces, int image_size, int texture_size, void* workspace,
                                     size_t workspace_bytes, d3m_stream_t stream) {
    if (!face_index_map || !sampling_weight_map || !sampling_index_map || !grad_rgb_map || !grad_textures ||
        batch_size <= 0 || num_faces <= 0 || image_size <= 0 || texture_size <= 0)
        return D3M_ERR_INVALID;
    int* flags = nullptr;
    if (faces && workspace && workspace_bytes >= (size_t)batch_size * num_faces * 4) flags = (int*)workspace;
    DenseFaces fs{faces, num_faces};
    return run_backward_textures(fs, faces, face_index_map, sampling_weight_map, sampling_index_map, grad_rgb_map,
                                 grad_textures, batch_size, num_faces, image_size, texture_size, flags, false,
                                 (hipStream_t)stream);
}

D3M_EXPORT int d3m_backward_depth_map(const float* faces, const float* depth_map, const int32_t* face_index_map,
                                      const float* face_inv_map, const float* weight_map, const float* grad_depth_map,
                                      float* grad_faces, int batch_size, int num_faces, int image_size, void* workspace,
                                      size_t workspace_bytes, d3m_stream_t stream) {
    if (!faces || !depth_map || !face_index_map || !weight_map || !grad_depth_map || !grad_faces || batch_size <= 0 ||
        num_faces <= 0 || image_size <= 0)
        return D3M_ERR_INVALID;
    int* flags = nullptr;
    if (workspace && workspace_bytes >= (size_t)batch_size * num_faces * 4) flags = (int*)workspace;
    DenseFaces fs{faces, num_faces};
    return run_backward_depth(fs, depth_map, face_index_map, face_inv_map, weight_map, grad_depth_map, grad_faces,
                              batch_size, image_size, flags, false, (hipStream_t)stream);
}

// The same operator for a mesh pipeline: over the compacted list of the faces that own a pixel (`visibility`, after
// d3m_visibility) instead of every face of the batch, and with its sums added straight into the gradient of the vertices the
// faces were gathered from (`vertex_target`) instead of a dense [B,F,3,3] array that a scatter-add pass then folds.
// large_counter: 256 zeroed bytes (the entry clears them unless flags & D3M_PRECLEARED).
D3M_EXPORT int d3m_backward_depth_map_mesh(const float* faces, const float* depth_map, const int32_t* face_index_map,
                                           const float* weight_map, const float* grad_depth_map, int batch_size, int num_faces,
                                           int image_size, const d3m_vertex_target* vertex_target, void* visibility,
                                           void* large_counter, int flags, d3m_stream_t stream) {
    if (!faces || !depth_map || !face_index_map || !weight_map || !grad_depth_map || !vertex_target || !visibility ||
        !large_counter || batch_size <= 0 || num_faces <= 0 || image_size <= 0)
        return D3M_ERR_INVALID;
    VertexTarget vt;
    if (int rc = to_vertex_target(vertex_target, num_faces, vt)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int B = batch_size, S = image_size;
    const long n = (long)B * S * S, nf = (long)B * num_faces;
    if (!(flags & D3M_PRECLEARED)) HIP_TRY(zero_async(large_counter, 256, st));
    const int flip = (flags & D3M_GRAD_OF_OUTPUT_IMAGE) ? 1 : 0;
    const VisibilityView v = visibility_view(visibility, nf);
    DenseFaces fs{faces, num_faces};
    // (a coarse mesh -- more than 48 raster pixels per triangle -- gives a face a whole wave instead of eight lanes)
    const bool coarse = (double)S * S > 48.0 * ((double)num_faces / (vt.Ft && num_faces == 2 * vt.Ft ? 2.0 : 1.0));
    const unsigned all_blocks = blocks_for(nf, coarse ? 4 : FM_FACES_PER_BLOCK);
    const dim3 g_faces(all_blocks > 4096 ? 4096 : (all_blocks + 7) / 8 * 8);
    if (coarse)
        LAUNCH("k_backward_depth_faces", (k_backward_depth_faces<DenseFaces, 64>), g_faces, dim3(256), st, fs, depth_map,
               face_index_map, weight_map, grad_depth_map, (float*)nullptr, v.flags, B, S, (const int*)v.list,
               (const int*)v.count, vt, (int*)large_counter, flip, FM_MAX_BBOX_AREA);
    else
        LAUNCH("k_backward_depth_faces", (k_backward_depth_faces<DenseFaces>), g_faces, dim3(256), st, fs, depth_map,
               face_index_map, weight_map, grad_depth_map, (float*)nullptr, v.flags, B, S, (const int*)v.list,
               (const int*)v.count, vt, (int*)large_counter, flip, FM_MAX_BBOX_AREA);
    LAUNCH("k_backward_depth_map", k_backward_depth_map<DenseFaces>, dim3(px_grid(n, true)), dim3(256), st, fs, depth_map,
           face_index_map, (const float*)nullptr, weight_map, grad_depth_map, (float*)nullptr, B, S, (const int*)v.flags, vt,
           GradScale{nullptr, nullptr, 0.0f, 0, nullptr}, (const int*)large_counter, flip);
    return check_launch();
}

// ---------------------------------------------------------------------------------------------------
// B. camera, gather / scatter, epilogue
// ---------------------------------------------------------------------------------------------------
static int to_cam(const d3m_camera* h, int B, Cam& c) {
    if (!h) return D3M_ERR_INVALID;
    c.mode = h->mode; c.perspective = h->perspective; c.width = h->tan_half_width; c.orig = h->orig_size;
    c.rot = h->rot; c.eye_or_t = h->eye_or_t; c.K = h->K; c.dist = h->dist;
    c.rot_b = h->rot_batch; c.eye_b = h->eye_batch; c.K_b = h->K_batch; c.dist_b = h->dist_batch;
    if (c.mode == D3M_CAMERA_NONE) return D3M_OK;
    if (c.mode != D3M_CAMERA_LOOK_AT && c.mode != D3M_CAMERA_LOOK && c.mode != D3M_CAMERA_PROJECTION) return D3M_ERR_INVALID;
    if (!c.rot || !c.eye_or_t) return D3M_ERR_INVALID;
    if ((c.rot_b != 1 && c.rot_b != B) || (c.eye_b != 1 && c.eye_b != B)) return D3M_ERR_INVALID;
    if (c.mode == D3M_CAMERA_PROJECTION) {
        if (!c.K || !c.dist) return D3M_ERR_INVALID;
        if ((c.K_b != 1 && c.K_b != B) || (c.dist_b != 1 && c.dist_b != B)) return D3M_ERR_INVALID;
    }
    return D3M_OK;
}

D3M_EXPORT int d3m_camera_basis(const float* eye, int eye_batch, const float* at_or_direction, int at_batch,
                                const float* up, int up_batch, int is_look_at, float* rot_out, int batch_size,
                                d3m_stream_t stream) {
    if (!eye || !at_or_direction || !up || !rot_out || batch_size <= 0) return D3M_ERR_INVALID;
    LAUNCH("k_camera_basis", k_camera_basis, dim3(blocks_for(batch_size, 64)), dim3(64), (hipStream_t)stream, eye, eye_batch,
                       at_or_direction, at_batch, up, up_batch, is_look_at, rot_out, batch_size);
    return check_launch();
}

D3M_EXPORT int d3m_camera_forward(const float* vertices, int vertices_batch, const d3m_camera* cam, float* out,
                                  int batch_size, int num_vertices, d3m_stream_t stream) {
    if (!vertices || !out || batch_size <= 0 || num_vertices <= 0) return D3M_ERR_INVALID;
    if (vertices_batch != 1 && vertices_batch != batch_size) return D3M_ERR_INVALID;
    Cam c;
    int rc = to_cam(cam, batch_size, c);
    if (rc) return rc;
    const long n = (long)batch_size * num_vertices;
    LAUNCH("k_camera_forward", k_camera_forward, dim3(blocks_for(n, 256)), dim3(256), (hipStream_t)stream, vertices,
                       vertices_batch, c, out, batch_size, num_vertices);
    return check_launch();
}

static int run_camera_backward(const float* vertices, int vertices_batch, const d3m_camera* cam, const float* grad_out,
                               float* grad_vertices, int batch_size, int num_vertices, bool accumulate, hipStream_t st) {
    if (!vertices || !grad_out || !grad_vertices || batch_size <= 0 || num_vertices <= 0) return D3M_ERR_INVALID;
    if (vertices_batch != 1 && vertices_batch != batch_size) return D3M_ERR_INVALID;
    Cam c;
    int rc = to_cam(cam, batch_size, c);
    if (rc) return rc;
    const long n = vertices_batch > 1 ? (long)batch_size * num_vertices : 8l * num_vertices;     // lanes: see the kernel
    LAUNCH("k_camera_backward", k_camera_backward, dim3(blocks_for(n, 256)), dim3(256), st, vertices, vertices_batch, c,
           grad_out, grad_vertices, batch_size, num_vertices, accumulate);
    return check_launch();
}
D3M_EXPORT int d3m_camera_backward(const float* vertices, int vertices_batch, const d3m_camera* cam,
                                   const float* grad_out, float* grad_vertices, int batch_size, int num_vertices,
                                   d3m_stream_t stream) {
    return run_camera_backward(vertices, vertices_batch, cam, grad_out, grad_vertices, batch_size, num_vertices, false,
                               (hipStream_t)stream);
}
D3M_EXPORT int d3m_camera_backward_add(const float* vertices, int vertices_batch, const d3m_camera* cam,
                                       const float* grad_out, float* grad_vertices, int batch_size, int num_vertices,
                                       d3m_stream_t stream) {
    return run_camera_backward(vertices, vertices_batch, cam, grad_out, grad_vertices, batch_size, num_vertices, true,
                               (hipStream_t)stream);
}

D3M_EXPORT int d3m_gather_faces(const float* vertices, const int32_t* tri, int tri_batch, float* faces_out,
                                int batch_size, int num_vertices, int num_tri, int fill_back, d3m_stream_t stream) {
    if (!vertices || !tri || !faces_out || batch_size <= 0 || num_vertices <= 0 || num_tri <= 0) return D3M_ERR_INVALID;
    if (tri_batch != 1 && tri_batch != batch_size) return D3M_ERR_INVALID;
    IndexedFaces fs{vertices, tri, num_vertices, num_tri, tri_batch, fill_back ? 1 : 0, batch_size};
    const long n = (long)batch_size * fs.num_faces() * 9;
    LAUNCH("k_gather_faces", k_gather_faces, dim3(blocks_for(n, 256)), dim3(256), (hipStream_t)stream, fs, faces_out,
                       batch_size);
    return check_launch();
}

D3M_EXPORT int d3m_scatter_face_grads(const float* grad_faces, const int32_t* tri, int tri_batch, float* grad_vertices,
                                      int batch_size, int num_vertices, int num_tri, int fill_back,
                                      d3m_stream_t stream) {
    if (!grad_faces || !tri || !grad_vertices || batch_size <= 0 || num_vertices <= 0 || num_tri <= 0)
        return D3M_ERR_INVALID;
    if (tri_batch != 1 && tri_batch != batch_size) return D3M_ERR_INVALID;
    IndexedFaces fs{nullptr, tri, num_vertices, num_tri, tri_batch, fill_back ? 1 : 0, batch_size};
    const long n = (long)batch_size * fs.num_faces() * 9;
    LAUNCH("k_scatter_face_grads", k_scatter_face_grads, dim3(blocks_for(n, 256)), dim3(256), (hipStream_t)stream, fs, grad_faces,
                       grad_vertices, batch_size);
    return check_launch();
}

static LightParams to_light(float ia, float id, const float* ca, const float* cd, const float* dir) {
    LightParams lp;
    lp.ia = ia; lp.id = id;
    for (int k = 0; k < 3; k++) { lp.ca[k] = ca[k]; lp.cd[k] = cd[k]; lp.dir[k] = dir[k]; }
    return lp;
}

D3M_EXPORT int d3m_lighting_forward(const float* faces, const float* textures_in, float* textures_out,
                                    float intensity_ambient, float intensity_directional, const float* color_ambient,
                                    const float* color_directional, const float* direction, long num_faces_total,
                                    int texture_size, d3m_stream_t stream) {
    if (!faces || !textures_in || !textures_out || !color_ambient || !color_directional || !direction ||
        num_faces_total <= 0 || texture_size <= 0)
        return D3M_ERR_INVALID;
    const LightParams lp = to_light(intensity_ambient, intensity_directional, color_ambient, color_directional, direction);
    LAUNCH("k_lighting_forward", k_lighting_forward, dim3(blocks_for(num_faces_total, 256)), dim3(256), (hipStream_t)stream, faces,
                       textures_in, textures_out, lp, num_faces_total, texture_size * texture_size * texture_size * 3);
    return check_launch();
}

D3M_EXPORT int d3m_lighting_backward(const float* faces, const float* textures_in, const float* grad_out,
                                     float* grad_textures, float* grad_faces, float intensity_ambient,
                                     float intensity_directional, const float* color_ambient,
                                     const float* color_directional, const float* direction, long num_faces_total,
                                     int texture_size, d3m_stream_t stream) {
    if (!faces || !textures_in || !grad_out || !color_ambient || !color_directional || !direction ||
        num_faces_total <= 0 || texture_size <= 0)
        return D3M_ERR_INVALID;
    const LightParams lp = to_light(intensity_ambient, intensity_directional, color_ambient, color_directional, direction);
    LAUNCH("k_lighting_backward", k_lighting_backward, dim3(blocks_for(num_faces_total, 256)), dim3(256), (hipStream_t)stream, faces,
                       textures_in, grad_out, grad_textures, grad_faces, lp, num_faces_total,
                       texture_size * texture_size * texture_size * 3);
    return check_launch();
}

D3M_EXPORT int d3m_view_transform(const float* view, int num_components, float* rot, float* trans, int batch_size,
                                  d3m_stream_t stream) {
    if (!view || !rot || !trans || batch_size <= 0) return D3M_ERR_INVALID;
    if (num_components != 3 && num_components != 5 && num_components != 6) return D3M_ERR_INVALID;
    LAUNCH("k_view_transform", k_view_transform, dim3(blocks_for(batch_size, 64)), dim3(64), (hipStream_t)stream, view,
           num_components, rot, trans, batch_size);
    return check_launch();
}

D3M_EXPORT int d3m_view_transform_backward(const float* view, int num_components, const float* grad_rot,
                                           const float* grad_trans, float* grad_view, int batch_size, d3m_stream_t stream) {
    if (!view || !grad_view || batch_size <= 0) return D3M_ERR_INVALID;
    if (num_components != 3 && num_components != 5 && num_components != 6) return D3M_ERR_INVALID;
    LAUNCH("k_view_transform_backward", k_view_transform_backward, dim3(blocks_for(batch_size, 64)), dim3(64),
           (hipStream_t)stream, view, num_components, grad_rot, grad_trans, grad_view, batch_size);
    return check_launch();
}

static int to_grid_warp(const float* depth, const float* inv_K, int inv_K_batch, const float* rot, const float* trans,
                        float rot_center_depth, const float* K, int K_batch, const int* crop, int B, int H, int W, GridWarp& g) {
    if (!depth || !inv_K || !rot || !trans || B <= 0 || H <= 0 || W <= 0) return D3M_ERR_INVALID;
    if ((inv_K_batch != 1 && inv_K_batch != B) || (K && K_batch != 1 && K_batch != B)) return D3M_ERR_INVALID;
    g = GridWarp{depth, inv_K, inv_K_batch, rot, trans, rot_center_depth, K, K_batch, 0, 0, 0, 0, B, H, W};
    if (crop) {
        if (crop[0] < 0 || crop[1] < 0 || crop[2] < 0 || crop[3] < 0 || crop[0] + crop[1] >= H || crop[2] + crop[3] >= W)
            return D3M_ERR_INVALID;
        g.crop_top = crop[0]; g.crop_bottom = crop[1]; g.crop_left = crop[2]; g.crop_right = crop[3];
    }
    return D3M_OK;
}

D3M_EXPORT int d3m_grid_warp(const float* depth, const float* inv_K, int inv_K_batch, const float* rot, const float* trans,
                             float rot_center_depth, const float* K, int K_batch, const int* crop, float* out,
                             int batch_size, int height, int width, d3m_stream_t stream) {
    GridWarp g;
    if (int rc = to_grid_warp(depth, inv_K, inv_K_batch, rot, trans, rot_center_depth, K, K_batch, crop, batch_size, height,
                              width, g))
        return rc;
    if (!out) return D3M_ERR_INVALID;
    LAUNCH("k_grid_warp", k_grid_warp, dim3(blocks_for((long)batch_size * height * width, 256)), dim3(256),
           (hipStream_t)stream, g, out);
    return check_launch();
}

D3M_EXPORT int d3m_grid_warp_backward(const float* depth, const float* inv_K, int inv_K_batch, const float* rot,
                                      const float* trans, float rot_center_depth, const float* K, int K_batch,
                                      const float* grad_out, float* grad_depth, float* grad_rot, float* grad_trans,
                                      int batch_size, int height, int width, d3m_stream_t stream) {
    GridWarp g;
    if (int rc = to_grid_warp(depth, inv_K, inv_K_batch, rot, trans, rot_center_depth, K, K_batch, nullptr, batch_size,
                              height, width, g))
        return rc;
    if (!grad_out) return D3M_ERR_INVALID;
    // enough workgroups to occupy the chip: up to 16 per batch entry, each with at least 512 pixels
    const long px = (long)height * width;
    int split = (int)std::min<long>(16, std::max<long>(1, px / 512));
    while (split > 1 && (long)batch_size * split > 2048) split /= 2;
    if (split > 1) {
        if (grad_rot) HIP_TRY(zero_async(grad_rot, (size_t)batch_size * 9 * 4, (hipStream_t)stream));
        if (grad_trans) HIP_TRY(zero_async(grad_trans, (size_t)batch_size * 3 * 4, (hipStream_t)stream));
    }
    LAUNCH("k_grid_warp_backward", k_grid_warp_backward, dim3(batch_size, split), dim3(256), (hipStream_t)stream, g, grad_out,
           grad_depth, grad_rot, grad_trans);
    return check_launch();
}

D3M_EXPORT int d3m_depth_normals(const float* depth, const float* inv_K, int inv_K_batch, float* normal, int batch_size,
                                 int height, int width, d3m_stream_t stream) {
    if (!depth || !inv_K || !normal || batch_size <= 0 || height <= 0 || width <= 0) return D3M_ERR_INVALID;
    if (inv_K_batch != 1 && inv_K_batch != batch_size) return D3M_ERR_INVALID;
    LAUNCH("k_depth_normals", k_depth_normals, dim3(blocks_for((long)batch_size * height * width, 256)), dim3(256),
           (hipStream_t)stream, depth, inv_K, inv_K_batch, normal, batch_size, height, width);
    return check_launch();
}

D3M_EXPORT int d3m_depth_normals_backward(const float* depth, const float* inv_K, int inv_K_batch, const float* grad_normal,
                                          float* grad_depth, int batch_size, int height, int width, d3m_stream_t stream) {
    if (!depth || !inv_K || !grad_normal || !grad_depth || batch_size <= 0 || height <= 0 || width <= 0) return D3M_ERR_INVALID;
    if (inv_K_batch != 1 && inv_K_batch != batch_size) return D3M_ERR_INVALID;
    LAUNCH("k_depth_normals_backward", k_depth_normals_backward, dim3(blocks_for((long)batch_size * height * width, 256)),
           dim3(256), (hipStream_t)stream, depth, inv_K, inv_K_batch, grad_normal, grad_depth, batch_size, height, width);
    return check_launch();
}

D3M_EXPORT int d3m_textures_from_im(const float* im, float* textures, int batch_size, int channels, int height, int width,
                                    int texture_size, d3m_stream_t stream) {
    if (!im || !textures || batch_size <= 0 || channels <= 0 || height < 2 || width < 2) return D3M_ERR_INVALID;
    if (texture_size != 1 && texture_size != 2) return D3M_ERR_INVALID;        // utils.py:106
    const long n = (long)batch_size * 2 * (height - 1) * (width - 1) * (texture_size == 2 ? 8 : 1);
    LAUNCH("k_textures_from_im", k_textures_from_im, dim3(blocks_for(n, 256)), dim3(256), (hipStream_t)stream, im, textures,
           batch_size, channels, height, width, texture_size);
    return check_launch();
}

D3M_EXPORT int d3m_textures_from_im_backward(const float* grad_textures, float* grad_im, int batch_size, int channels,
                                             int height, int width, int texture_size, d3m_stream_t stream) {
    if (!grad_textures || !grad_im || batch_size <= 0 || channels <= 0 || height < 2 || width < 2) return D3M_ERR_INVALID;
    if (texture_size != 1 && texture_size != 2) return D3M_ERR_INVALID;
    LAUNCH("k_textures_from_im_backward", k_textures_from_im_backward,
           dim3(blocks_for((long)batch_size * channels * height * width, 256)), dim3(256), (hipStream_t)stream, grad_textures,
           grad_im, batch_size, channels, height, width, texture_size);
    return check_launch();
}

// ---- Pt3dRenderer.sample's per-pixel pass (d3m_uv.h) ------------------------------------------------------------------
static int to_uv_unwrap(const int32_t* face_index_map, const float* weight_map, const int32_t* tri, const float* verts,
                        const float* vnormals, const float* uvs, const float* imgs, const int32_t* used, const float* light,
                        int B, int cov_B, int T, int F, int V, int C, int H, int W, UvUnwrap& a) {
    if (!face_index_map || !weight_map || !tri || !verts || !vnormals || !uvs || !imgs || !used || !light || B <= 0 ||
        T <= 0 || F <= 0 || V <= 0 || C <= 0 || C > 3 || H <= 0 || W <= 0 || (cov_B != 1 && cov_B != B))
        return D3M_ERR_INVALID;
    // light: location (3), camera centre (3), ambient, diffuse, specular, shininess
    a = UvUnwrap{face_index_map, weight_map, tri, verts, vnormals, uvs, imgs, used, {light[0], light[1], light[2]},
                 {light[3], light[4], light[5]}, light[6], light[7], light[8], light[9], B, T, F, V, C, H, W, cov_B};
    return D3M_OK;
}

D3M_EXPORT int d3m_uv_unwrap(const int32_t* face_index_map, const float* weight_map, const int32_t* tri, const float* verts,
                             const float* vnormals, const float* uvs, const float* imgs, const int32_t* used,
                             const float* light, float* out_img, float* out_mask, int batch_size, int coverage_batch,
                             int texture_size, int num_tri, int num_vertices, int channels, int height, int width,
                             d3m_stream_t stream) {
    UvUnwrap a;
    if (int rc = to_uv_unwrap(face_index_map, weight_map, tri, verts, vnormals, uvs, imgs, used, light, batch_size,
                              coverage_batch, texture_size, num_tri, num_vertices, channels, height, width, a))
        return rc;
    if (!out_img || !out_mask) return D3M_ERR_INVALID;
    LAUNCH("k_uv_unwrap", k_uv_unwrap, dim3(blocks_for((long)batch_size * texture_size * texture_size, 256)), dim3(256),
           (hipStream_t)stream, a, out_img, out_mask);
    return check_launch();
}

D3M_EXPORT int d3m_uv_unwrap_backward(const int32_t* face_index_map, const float* weight_map, const int32_t* tri,
                                      const float* verts, const float* vnormals, const float* uvs, const float* imgs,
                                      const int32_t* used, const float* light, const float* grad_img, float* grad_imgs,
                                      float* grad_uvs, int batch_size, int coverage_batch, int texture_size, int num_tri,
                                      int num_vertices, int channels, int height, int width, d3m_stream_t stream) {
    UvUnwrap a;
    if (int rc = to_uv_unwrap(face_index_map, weight_map, tri, verts, vnormals, uvs, imgs, used, light, batch_size,
                              coverage_batch, texture_size, num_tri, num_vertices, channels, height, width, a))
        return rc;
    if (!grad_img) return D3M_ERR_INVALID;
    LAUNCH("k_uv_unwrap_backward", k_uv_unwrap_backward, dim3(blocks_for((long)batch_size * texture_size * texture_size, 256)),
           dim3(256), (hipStream_t)stream, a, grad_img, grad_imgs, grad_uvs);
    return check_launch();
}

// ---- lit sampling: fill_back and lighting on the fly, shared textures (d3m_lit.h) ----------------------
D3M_EXPORT int d3m_face_light(const float* vertices, int vertices_batch, const int32_t* tri, int tri_batch, float* light,
                              float intensity_ambient, float intensity_directional, const float* color_ambient,
                              const float* color_directional, const float* direction, int light_batch, int num_vertices,
                              int num_tri, int fill_back, d3m_stream_t stream) {
    if (!vertices || !light || !color_ambient || !color_directional || !direction || light_batch <= 0 ||
        num_vertices <= 0 || num_tri <= 0)
        return D3M_ERR_INVALID;
    int grid_w;
    if (!tri_source_ok(tri, tri_batch, tri_batch, num_vertices, num_tri, grid_w)) return D3M_ERR_INVALID;
    IndexedFaces fs{vertices, tri, num_vertices, num_tri, tri ? tri_batch : 1, fill_back ? 1 : 0, vertices_batch, grid_w};
    const LightParams lp = to_light(intensity_ambient, intensity_directional, color_ambient, color_directional, direction);
    const long n = (long)light_batch * fs.num_faces();
    LAUNCH("k_face_light", k_face_light, dim3(blocks_for(n, 256)), dim3(256), (hipStream_t)stream, fs, lp, light, light_batch);
    return check_launch();
}

D3M_EXPORT int d3m_face_light_backward(const float* vertices, int vertices_batch, const int32_t* tri, int tri_batch,
                                       const float* grad_light, float* grad_vertices, float intensity_ambient,
                                       float intensity_directional, const float* color_ambient,
                                       const float* color_directional, const float* direction, int light_batch,
                                       int num_vertices, int num_tri, int fill_back, d3m_stream_t stream) {
    if (!vertices || !grad_light || !grad_vertices || !color_ambient || !color_directional || !direction ||
        light_batch <= 0 || num_vertices <= 0 || num_tri <= 0)
        return D3M_ERR_INVALID;
    int grid_w;
    if (!tri_source_ok(tri, tri_batch, tri_batch, num_vertices, num_tri, grid_w)) return D3M_ERR_INVALID;
    IndexedFaces fs{vertices, tri, num_vertices, num_tri, tri ? tri_batch : 1, fill_back ? 1 : 0, vertices_batch, grid_w};
    const LightParams lp = to_light(intensity_ambient, intensity_directional, color_ambient, color_directional, direction);
    const long n = (long)light_batch * fs.num_faces();
    LAUNCH("k_face_light_backward", k_face_light_backward, dim3(blocks_for(n, 256)), dim3(256), (hipStream_t)stream, fs, lp,
           grad_light, grad_vertices, vertices_batch, light_batch);
    return check_launch();
}

// ---- deterministic mode: the vertex sums gathered in a fixed order instead of scattered with float atomics ---------------
D3M_EXPORT int d3m_vertex_gather(const float* grad_faces_a, const float* grad_faces_b, const int32_t* adj_offsets,
                                 const int32_t* adj_items, float* grad_vertices, int batch_size, int num_vertices,
                                 int num_tri, int fill_back, const void* visibility, d3m_stream_t stream) {
    if ((!grad_faces_a && !grad_faces_b) || !adj_offsets || !adj_items || !grad_vertices || batch_size <= 0 ||
        num_vertices <= 0 || num_tri <= 0)
        return D3M_ERR_INVALID;
    LAUNCH("k_vertex_gather", k_vertex_gather, dim3(blocks_for((long)batch_size * num_vertices, 256)), dim3(256),
           (hipStream_t)stream, grad_faces_a, grad_faces_b, adj_offsets, adj_items, grad_vertices, batch_size, num_vertices,
           num_tri, fill_back ? 1 : 0,
           visibility ? (const int*)visibility_view(const_cast<void*>(visibility), (long)batch_size * (fill_back ? 2 : 1) * num_tri).flags
                      : (const int*)nullptr);
    return check_launch();
}
D3M_EXPORT int d3m_face_light_backward_gather(const float* vertices, const int32_t* tri, const int32_t* adj_offsets,
                                              const int32_t* adj_items, const float* grad_light, float* grad_vertices,
                                              float intensity_ambient, float intensity_directional,
                                              const float* color_ambient, const float* color_directional,
                                              const float* direction, int num_vertices, int num_tri, int fill_back,
                                              d3m_stream_t stream) {
    if (!vertices || !tri || !adj_offsets || !adj_items || !grad_light || !grad_vertices || !color_ambient ||
        !color_directional || !direction || num_vertices <= 0 || num_tri <= 0)
        return D3M_ERR_INVALID;
    IndexedFaces fs{vertices, tri, num_vertices, num_tri, 1, fill_back ? 1 : 0, 1, 0};
    const LightParams lp = to_light(intensity_ambient, intensity_directional, color_ambient, color_directional, direction);
    LAUNCH("k_face_light_backward_gather", k_face_light_backward_gather, dim3(blocks_for((long)num_vertices, 256)), dim3(256),
           (hipStream_t)stream, fs, lp, grad_light, adj_offsets, adj_items, grad_vertices);
    return check_launch();
}

// ---- the first launch of a lit render step: camera + per-face light + every clear (d3m_front.h) ------------------------
D3M_EXPORT int d3m_lit_front(const float* vertices, int vertices_batch, const d3m_camera* cam, const d3m_basis* basis,
                             float* screen_out, int batch_size, int num_vertices, const int32_t* tri, int tri_batch,
                             int num_tri, int fill_back, float* light, int light_batch, float intensity_ambient,
                             float intensity_directional, const float* color_ambient, const float* color_directional,
                             const float* direction, void* const* zero_ptrs, const size_t* zero_bytes, int zero_count,
                             d3m_stream_t stream) {
    if (!vertices || batch_size <= 0 || num_vertices <= 0 || zero_count < 0 || zero_count > FRONT_RANGES) return D3M_ERR_INVALID;
    if (zero_count && (!zero_ptrs || !zero_bytes)) return D3M_ERR_INVALID;
    FrontArgs a;
    memset(&a, 0, sizeof(a));
    a.vertices = vertices; a.vb = vertices_batch; a.B = batch_size; a.V = num_vertices;
    if (cam) {                                          // the camera part
        if (!screen_out || (vertices_batch != 1 && vertices_batch != batch_size)) return D3M_ERR_INVALID;
        d3m_camera c = *cam;
        if (basis) {                                    // the basis is computed here and left in cam->rot for the adjoint
            if (!basis->eye || !basis->at_or_direction || !basis->up || !cam->rot) return D3M_ERR_INVALID;
            if (cam->mode != D3M_CAMERA_LOOK_AT && cam->mode != D3M_CAMERA_LOOK) return D3M_ERR_INVALID;
            const int nb = cam->rot_batch;
            if ((nb != 1 && nb != batch_size) || (basis->eye_batch != 1 && basis->eye_batch != nb) ||
                (basis->at_batch != 1 && basis->at_batch != nb) || (basis->up_batch != 1 && basis->up_batch != nb))
                return D3M_ERR_INVALID;
            a.basis = FrontBasis{basis->eye, basis->at_or_direction, basis->up, basis->eye_batch, basis->at_batch,
                                 basis->up_batch, basis->is_look_at ? 1 : 0, const_cast<float*>(cam->rot)};
        }
        if (int rc = to_cam(&c, batch_size, a.cam)) return rc;
        a.screen = screen_out;
        a.nb_cam = blocks_for((long)batch_size * num_vertices, 256);
    }
    if (light) {                                        // the light part
        if (!color_ambient || !color_directional || !direction || light_batch <= 0 || num_tri <= 0) return D3M_ERR_INVALID;
        int grid_w;
        if (!tri_source_ok(tri, tri_batch, tri_batch, num_vertices, num_tri, grid_w)) return D3M_ERR_INVALID;
        a.faces = IndexedFaces{vertices, tri, num_vertices, num_tri, tri ? tri_batch : 1, fill_back ? 1 : 0, vertices_batch, grid_w};
        a.lp = to_light(intensity_ambient, intensity_directional, color_ambient, color_directional, direction);
        a.light = light; a.light_b = light_batch;
        a.nb_light = blocks_for((long)light_batch * a.faces.num_faces(), 256);
    }
    size_t most = 0;
    int used = 0;
    for (int k = 0; k < zero_count; k++) {
        if (!zero_ptrs[k] || zero_bytes[k] == 0) continue;
        if ((zero_bytes[k] & 3) || ((uintptr_t)zero_ptrs[k] & 3)) return D3M_ERR_INVALID;
        a.z_ptr[used] = (uint32_t*)zero_ptrs[k];
        a.z_words[used] = zero_bytes[k] >> 2;
        most = zero_bytes[k] > most ? zero_bytes[k] : most;
        used++;
    }
    size_t nb_zero = used ? (most / 16 + 255) / 256 : 0;
    if (nb_zero > 2048) nb_zero = 2048;
    if (used && nb_zero == 0) nb_zero = 1;
    const unsigned grid = a.nb_cam + a.nb_light + (unsigned)nb_zero;
    if (grid == 0) return D3M_OK;
    LAUNCH("k_lit_front", k_lit_front, dim3(grid), dim3(256), (hipStream_t)stream, a);
    return check_launch();
}

// ---- ... and its last launch: the camera's and the light's adjoints ADDED into grad_vertices (d3m_front.h k_lit_back) ------
D3M_EXPORT int d3m_lit_back(const float* vertices, int vertices_batch, const d3m_camera* cam, const float* grad_screen,
                            float* grad_vertices, int batch_size, int num_vertices, const int32_t* tri, int tri_batch,
                            int num_tri, int fill_back, const float* grad_light, int light_batch, float intensity_ambient,
                            float intensity_directional, const float* color_ambient, const float* color_directional,
                            const float* direction, d3m_stream_t stream) {
    if (!vertices || !cam || !grad_screen || !grad_vertices || !grad_light || !color_ambient || !color_directional ||
        !direction || batch_size <= 0 || num_vertices <= 0 || num_tri <= 0 || light_batch <= 0)
        return D3M_ERR_INVALID;
    if (vertices_batch != 1 && vertices_batch != batch_size) return D3M_ERR_INVALID;
    BackArgs a;
    memset(&a, 0, sizeof(a));
    if (int rc = to_cam(cam, batch_size, a.cam)) return rc;
    int grid_w;
    if (!tri_source_ok(tri, tri_batch, tri_batch, num_vertices, num_tri, grid_w)) return D3M_ERR_INVALID;
    a.vertices = vertices; a.vb = vertices_batch; a.grad_screen = grad_screen; a.grad_vertices = grad_vertices;
    a.B = batch_size; a.V = num_vertices;
    a.faces = IndexedFaces{vertices, tri, num_vertices, num_tri, tri ? tri_batch : 1, fill_back ? 1 : 0, vertices_batch, grid_w};
    a.lp = to_light(intensity_ambient, intensity_directional, color_ambient, color_directional, direction);
    a.grad_light = grad_light; a.light_b = light_batch;
    const long n_cam = vertices_batch > 1 ? (long)batch_size * num_vertices : 8l * num_vertices;     // lanes: see the kernel
    a.nb_cam = blocks_for(n_cam, 256);
    const unsigned nb_light = blocks_for((long)light_batch * a.faces.num_faces(), 256);
    LAUNCH("k_lit_back", k_lit_back, dim3(a.nb_cam + nb_light), dim3(256), (hipStream_t)stream, a);
    return check_launch();
}

static int make_lit(LitTextures& lt, const float* textures, int textures_batch, const float* light, int light_batch,
                    int num_tri, int texture_size, int fill_back, int B) {
    if (!textures || !light || num_tri <= 0 || texture_size <= 0) return D3M_ERR_INVALID;
    if (light_batch != 1 && light_batch != B) return D3M_ERR_INVALID;
    lt.textures = textures; lt.light = light; lt.F = num_tri; lt.Fp = fill_back ? 2 * num_tri : num_tri;
    lt.ts = texture_size; lt.tex_batch = textures_batch; lt.light_batch = light_batch; lt.fill_back = fill_back ? 1 : 0;
    lt.im = nullptr; lt.im_W = lt.im_H = 0;
    if (textures_batch < -1) {
        // the textures of a depth map's grid mesh taken from an image [B,3,H,W] on the fly (get_textures_from_im,
        // deep3dmap/core/renderer/utils.py:97-107, tx_size 2): `textures` is the image, W = -textures_batch
        const int W = -textures_batch;
        if (texture_size != 2 || W < 2 || num_tri % (2 * (W - 1))) return D3M_ERR_INVALID;
        lt.im = textures; lt.textures = nullptr; lt.im_W = W; lt.im_H = num_tri / (2 * (W - 1)) + 1; lt.tex_batch = B;
        return D3M_OK;
    }
    if (textures_batch != 1 && textures_batch != B) return D3M_ERR_INVALID;
    return D3M_OK;
}

D3M_EXPORT int d3m_forward_texture_sampling_lit(const float* faces, const float* textures, int textures_batch,
                                                const float* light, int light_batch, const int32_t* face_index_map,
                                                const float* weight_map, const float* depth_map, float* rgb_map,
                                                int batch_size, int num_tri, int fill_back, int image_size,
                                                int texture_size, float eps, d3m_stream_t stream) {
    if (!faces || !face_index_map || !weight_map || !depth_map || !rgb_map || batch_size <= 0 || image_size <= 0)
        return D3M_ERR_INVALID;
    LitTextures lt;
    int rc = make_lit(lt, textures, textures_batch, light, light_batch, num_tri, texture_size, fill_back, batch_size);
    if (rc) return rc;
    const long n = (long)batch_size * image_size * image_size;
    LAUNCH("k_texture_sampling_lit", k_texture_sampling_lit, dim3(blocks_for(n, 256)), dim3(256), (hipStream_t)stream, faces, lt,
           face_index_map, weight_map, depth_map, rgb_map, batch_size, image_size, eps);
    return check_launch();
}

// scratch of a fused objective: totals [8] | per-workgroup partial sums (float4) | group sums (float4) | ticket.
// One float4 of partials per 256-pixel workgroup of the plain epilogue, or per 32x32 tile and view of the records form
// (k_render_lit_fit_records) -- more of them than pixels / 256 when the image is smaller than a tile; a group (the unit of
// the objective's two-level finish, d3m_lit.h fit_finish_groups) is a view's tiles, or 256 consecutive workgroups.
struct FitScratch {
    size_t off_partials, off_group_sums, off_tickets, ticket_words, floats;
};
// Tile edge of the records form of the objective's pass (k_render_lit_fit_records<TILE>, k_fit_loss_records<TILE>): 16 -- one
// pixel per thread, four times the workgroups -- while the batch has at most FIT_TILE16_MAX_PIXELS pixels (a pass of a
// few hundred 32 x 32 tiles is one wave per SIMD with four dependent pixels each: 25-31 us for ANY small batch, round 5),
// 32 above.  A function of the launch's size alone: the objective's finish, which may run from another entry point, must
// find the same partial sums.
// Measured (round 6, same box, committed tree | this): one view of the 53 k mesh @256 0.108 -> 0.102 ms (the pass itself
// 24.4 -> 11 us), 8 views of the headline mesh 0.467 -> 0.458; 4 views 0.270 -> 0.273 (the pass is off that step's critical
// chain); 32 views keep the 32 x 32 tiles (one atomic per column and tile instead of four).
static const long FIT_TILE16_MAX_PIXELS = 4l << 20;
static int fit_tile(int B, int S) { return (long)B * S * S <= FIT_TILE16_MAX_PIXELS ? 16 : 32; }
static FitScratch fit_scratch_layout(int B, int S) {
    const size_t per_pixels = blocks_for((long)B * S * S, 256);
    const size_t tiles32 = (size_t)B * ((S + 31) / 32) * ((S + 31) / 32), tiles16 = (size_t)B * ((S + 15) / 16) * ((S + 15) / 16);
    const size_t tiles = tiles32 > tiles16 ? tiles32 : tiles16;       // (room for either tile size: see fit_tile)
    const size_t P = per_pixels > tiles ? per_pixels : tiles;
    const size_t G = std::max((size_t)B, (per_pixels + 255) / 256);
    FitScratch L;
    L.off_partials = 8;
    L.off_group_sums = 8 + 4 * P;
    L.off_tickets = L.off_group_sums + 4 * G;
    L.ticket_words = 4;
    L.floats = L.off_tickets + L.ticket_words;
    return L;
}
D3M_EXPORT size_t d3m_render_fit_scratch_floats(int batch_size, int image_size) {
    if (batch_size <= 0 || image_size <= 0) return 0;
    return fit_scratch_layout(batch_size, image_size).floats;
}
// the part of that scratch which must be ZERO when a pass with a fused objective starts (the finish's ticket): cleared by the
// pass's entry point itself unless the caller did (fit->flags & D3M_PRECLEARED)
D3M_EXPORT size_t d3m_render_fit_scratch_clear_range(int batch_size, int image_size, size_t* offset_floats) {
    if (batch_size <= 0 || image_size <= 0) return 0;
    const FitScratch L = fit_scratch_layout(batch_size, image_size);
    if (offset_floats) *offset_floats = L.off_tickets;
    return L.ticket_words;
}

static int to_fit_targets(const d3m_fit_targets* fit, FitTargets& ft) {
    ft = FitTargets{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (!fit) return D3M_OK;
    if (!fit->rgb_target || !fit->depth_target || !fit->alpha_target || !fit->mask || !fit->scratch || !fit->loss)
        return D3M_ERR_INVALID;
    const bool any = fit->grad_rgb_map || fit->grad_alpha_map || fit->grad_depth_map;
    if (any && !fit->edge_grad && !(fit->grad_rgb_map && fit->grad_alpha_map && fit->grad_depth_map)) return D3M_ERR_INVALID;
    ft = FitTargets{fit->rgb_target, fit->depth_target, fit->alpha_target, fit->mask, fit->scratch + 8,
                    fit->grad_rgb_map, fit->grad_alpha_map, fit->grad_depth_map};
    return D3M_OK;
}
// the finish of an objective whose pass left n_partials partial sums in groups of group_size (B views at internal size S,
// the objective on s x s output pixels per view)
static FitFin make_fit_fin(const d3m_fit_targets* fit, int B, int S, int s, int n_partials, int group_size) {
    const FitScratch L = fit_scratch_layout(B, S);
    return FitFin{(const float4*)(fit->scratch + L.off_partials), (float4*)(fit->scratch + L.off_group_sums),
                  (unsigned*)(fit->scratch + L.off_tickets), n_partials, group_size, (n_partials + group_size - 1) / group_size,
                  (float)((long)s * s), fit->mask_sum, fit->scratch, fit->loss};
}
static FitFin fit_fin_of_tiles(const d3m_fit_targets* fit, int B, int S) {     // the records form: a view's tiles
    // (the pooled form -- anti-aliasing -- keeps 32 x 32 internal pixels per workgroup)
    const int T = (fit->flags & D3M_FIT_POOLED) ? 32 : fit_tile(B, S);
    const int tiles = ((S + T - 1) / T) * ((S + T - 1) / T);
    return make_fit_fin(fit, B, S, (fit->flags & D3M_FIT_POOLED) ? S / 2 : S, B * tiles, tiles);
}
static int clear_fit_tickets(const d3m_fit_targets* fit, int B, int S, hipStream_t st) {
    if (fit->flags & D3M_PRECLEARED) return D3M_OK;
    const FitScratch L = fit_scratch_layout(B, S);
    HIP_TRY(zero_async(fit->scratch + L.off_tickets, L.ticket_words * 4, st));
    return D3M_OK;
}

D3M_EXPORT int d3m_render_lit_epilogue(const float* faces, const float* textures, int textures_batch, const float* light,
                                       int light_batch, const int32_t* face_index_map, const float* weight_map,
                                       const float* depth_map, const float* background, int background_batch,
                                       float* rgb_blended, float* alpha_map, float* rgb_out, float* alpha_out,
                                       float* depth_out, int batch_size, int num_tri, int fill_back, int image_size,
                                       int texture_size, float eps, int anti_aliasing, const d3m_fit_targets* fit,
                                       d3m_stream_t stream) {
    if (!faces || !face_index_map || !weight_map || !depth_map || !background || !rgb_blended || (!rgb_out && !fit) ||
        batch_size <= 0 || image_size <= 0)
        return D3M_ERR_INVALID;
    if (anti_aliasing && (image_size & 1)) return D3M_ERR_INVALID;
    if (fit && !alpha_map) return D3M_ERR_INVALID;
    // the records form of a pooled objective says so in the struct (its finish may run from another entry point)
    if (fit && fit->edge_grad && (anti_aliasing != 0) != ((fit->flags & D3M_FIT_POOLED) != 0)) return D3M_ERR_INVALID;
    if (background_batch != 1 && background_batch != batch_size) return D3M_ERR_INVALID;
    LitTextures lt;
    int rc = make_lit(lt, textures, textures_batch, light, light_batch, num_tri, texture_size, fill_back, batch_size);
    if (rc) return rc;
    FitTargets ft;
    const int s = anti_aliasing ? image_size / 2 : image_size;
    if ((rc = to_fit_targets(fit, ft))) return rc;
    const long n = (long)batch_size * s * s;
    hipStream_t st = (hipStream_t)stream;
    const int threads = 256;
    if (fit && (rc = clear_fit_tickets(fit, batch_size, image_size, st))) return rc;
    if (fit && fit->edge_grad) {
        // the objective's gradient leaves as the edge gradient's per-pixel records (and the depth gradient map)
        if (!fit->edge_dot || !fit->edge_nz_lo_inv || !fit->edge_nz_hi1 || !fit->mask_sum || !fit->grad_depth_map)
            return D3M_ERR_INVALID;
        const dim3 tiles((image_size + 31) / 32, (image_size + 31) / 32, batch_size);
        const dim3 tiles16((image_size + 15) / 16, (image_size + 15) / 16, batch_size);
        FitRecords rec{(float4*)fit->edge_grad, (float2*)fit->edge_dot, fit->edge_nz_lo_inv, fit->edge_nz_hi1, fit->mask_sum,
                       fit->grad_depth_map};
        if (anti_aliasing)
            LAUNCH("k_render_lit_fit_records", k_render_lit_fit_records_pooled, tiles, dim3(256), st, faces, lt,
                   face_index_map, weight_map, depth_map, background, background_batch, rgb_blended, alpha_map, rgb_out,
                   alpha_out, depth_out, batch_size, image_size, eps, ft, rec);
        else if (fit_tile(batch_size, image_size) == 16)
            LAUNCH("k_render_lit_fit_records", k_render_lit_fit_records<16>, tiles16, dim3(256), st, faces, lt, face_index_map,
                   weight_map, depth_map, background, background_batch, rgb_blended, alpha_map, rgb_out, alpha_out,
                   depth_out, batch_size, image_size, eps, ft, rec);
        else
            LAUNCH("k_render_lit_fit_records", k_render_lit_fit_records<32>, tiles, dim3(256), st, faces, lt, face_index_map,
                   weight_map, depth_map, background, background_batch, rgb_blended, alpha_map, rgb_out, alpha_out,
                   depth_out, batch_size, image_size, eps, ft, rec);
        // *fit->loss: a launch of its own (k_fit_finish, two levels), or -- D3M_FIT_FINISH_DEFERRED -- left to the caller's
        // d3m_backward_textures_lit, handed the same struct as `unscaled`
        if (!(fit->flags & D3M_FIT_FINISH_DEFERRED)) {
            const FitFin fin = fit_fin_of_tiles(fit, batch_size, image_size);
            LAUNCH("k_fit_finish", k_fit_finish, dim3(fin.n_groups), dim3(256), st, fin);
        }
        return check_launch();
    }
    if (fit && (fit->flags & D3M_FIT_FINISH_DEFERRED)) return D3M_ERR_INVALID;      // only the records form defers its finish
    LAUNCH("k_render_lit_epilogue", k_render_lit_epilogue, dim3(blocks_for(n, threads)), dim3(threads), st, faces, lt,
           face_index_map, weight_map, depth_map, background, background_batch, rgb_blended, alpha_map, rgb_out, alpha_out,
           depth_out, batch_size, image_size, anti_aliasing ? 1 : 0, eps, ft);
    if (fit) {
        const FitFin fin = make_fit_fin(fit, batch_size, image_size, s, (int)blocks_for(n, threads), 256);
        LAUNCH("k_fit_finish", k_fit_finish, dim3(fin.n_groups), dim3(256), st, fin);
    }
    return check_launch();
}

// The objective's finish as a call of its own, for a records-form pass made with D3M_FIT_FINISH_DEFERRED whose backward pass
// does not take the route that would have finished it (k_fit_finish: partial sums -> totals, *fit->loss).
D3M_EXPORT int d3m_fit_finish(const d3m_fit_targets* fit, int batch_size, int image_size, d3m_stream_t stream) {
    if (!fit || !fit->scratch || !fit->loss || !fit->edge_grad || batch_size <= 0 || image_size <= 0) return D3M_ERR_INVALID;
    const FitFin fin = fit_fin_of_tiles(fit, batch_size, image_size);
    LAUNCH("k_fit_finish", k_fit_finish, dim3(fin.n_groups), dim3(256), (hipStream_t)stream, fin);
    return check_launch();
}

// The fit objective of FINISHED images that came out of one lit render (no anti-aliasing), with its gradient left as the
// edge gradient's per-pixel records: see k_fit_loss_records.  fit: targets, mask, mask_sum, scratch
// (d3m_render_fit_scratch_floats), loss, edge_grad / edge_dot / edge_nz_* (zeroed) / grad_depth_map.
D3M_EXPORT int d3m_fit_loss_records(const float* rgb, const float* depth, const float* alpha, const int32_t* face_index_map,
                                    const d3m_fit_targets* fit, int batch_size, int image_size, d3m_stream_t stream) {
    if (!rgb || !depth || !alpha || !face_index_map || !fit || batch_size <= 0 || image_size <= 0) return D3M_ERR_INVALID;
    if (!fit->edge_grad || !fit->edge_dot || !fit->edge_nz_lo_inv || !fit->edge_nz_hi1 || !fit->mask_sum || !fit->grad_depth_map)
        return D3M_ERR_INVALID;
    FitTargets ft;
    if (int rc = to_fit_targets(fit, ft)) return rc;
    if (fit->flags & (D3M_FIT_FINISH_DEFERRED | D3M_FIT_POOLED)) return D3M_ERR_INVALID;      // (finished images: no pooling here)
    hipStream_t st = (hipStream_t)stream;
    if (int rc = clear_fit_tickets(fit, batch_size, image_size, st)) return rc;
    FitRecords rec{(float4*)fit->edge_grad, (float2*)fit->edge_dot, fit->edge_nz_lo_inv, fit->edge_nz_hi1, fit->mask_sum,
                   fit->grad_depth_map};
    if (fit_tile(batch_size, image_size) == 16)
        LAUNCH("k_fit_loss_records", k_fit_loss_records<16>, dim3((image_size + 15) / 16, (image_size + 15) / 16, batch_size),
               dim3(256), st, rgb, depth, alpha, face_index_map, batch_size, image_size, ft, rec);
    else
        LAUNCH("k_fit_loss_records", k_fit_loss_records<32>, dim3((image_size + 31) / 32, (image_size + 31) / 32, batch_size),
               dim3(256), st, rgb, depth, alpha, face_index_map, batch_size, image_size, ft, rec);
    const FitFin fin = fit_fin_of_tiles(fit, batch_size, image_size);
    LAUNCH("k_fit_finish", k_fit_finish, dim3(fin.n_groups), dim3(256), st, fin);
    return check_launch();
}

// workspace of the gathered lit backward: per-view gradients | flags | view masks | counter
struct LitWorkspace {
    float* gview;           // per-view texel gradients (shared textures)
    int* flags;             // when the caller brings no visibility blob
    unsigned* view_mask;
    size_t mask_bytes;
    int* n_large;           // 256 bytes
    size_t total;
};
static LitWorkspace lit_workspace(void* ws, int B, int num_tri, int fill_back, int texture_size) {
    const size_t ts3 = (size_t)texture_size * texture_size * texture_size;
    const size_t nf = (size_t)B * num_tri * (fill_back ? 2 : 1);
    char* p = (char*)ws;
    LitWorkspace L;
    L.gview = (float*)p;              p += align_up((size_t)B * num_tri * ts3 * 12, 256);
    L.flags = (int*)p;                p += align_up(nf * 4, 256);
    L.view_mask = (unsigned*)p;       L.mask_bytes = align_up((size_t)num_tri * ((B + 31) / 32) * 4, 256); p += L.mask_bytes;
    L.n_large = (int*)p;              p += 256;
    L.total = (size_t)(p - (char*)ws);
    return L;
}
D3M_EXPORT size_t d3m_backward_textures_lit_workspace_bytes(int batch_size, int num_tri, int fill_back, int texture_size) {
    if (batch_size <= 0 || num_tri <= 0 || texture_size <= 0) return 0;
    return lit_workspace(nullptr, batch_size, num_tri, fill_back, texture_size).total;
}
// what d3m_backward_textures_lit zeroes in front of its kernels (up to four ranges; returns how many)
static int lit_clear_ranges(const LitWorkspace& W, float* grad_textures, int textures_batch, float* grad_light, int light_batch,
                            int B, int num_tri, int fill_back, int texture_size, bool has_visibility, void** ptr, size_t* bytes) {
    const size_t ts3 = (size_t)texture_size * texture_size * texture_size, view_elems = (size_t)num_tri * ts3 * 3;
    const int Fp = (fill_back ? 2 : 1) * num_tri;
    // (gathered: texture sizes 2 -- and, over a visibility list, 3 and 4 -- store per-view sums instead of adding into zeros)
    const bool gathered = texture_size == 2 || (has_visibility && texture_size >= 3 && texture_size <= 4);
    const bool skip_zero = gathered && textures_batch == 1;
    int n = 0;
    if (!skip_zero) { ptr[n] = textures_batch > 1 ? grad_textures : W.gview; bytes[n++] = (size_t)B * view_elems * 4; }
    if (grad_light) { ptr[n] = grad_light; bytes[n++] = (size_t)light_batch * Fp * 12; }
    if (gathered) {
        // shared textures: the view masks (the sum over views reads only what was written) | counter, adjacent
        if (skip_zero) { ptr[n] = W.view_mask; bytes[n++] = W.mask_bytes + 256; }
        else { ptr[n] = W.n_large; bytes[n++] = 256; }
        if (!has_visibility) { ptr[n] = W.flags; bytes[n++] = (size_t)B * Fp * 4; }
    }
    return n;
}
D3M_EXPORT int d3m_backward_textures_lit_clear_ranges(float* grad_textures, int textures_batch, float* grad_light,
                                                      int light_batch, int batch_size, int num_tri, int fill_back,
                                                      int texture_size, void* workspace, int has_visibility, void** ptrs,
                                                      size_t* bytes) {
    if (batch_size <= 0 || num_tri <= 0 || texture_size <= 0 || !workspace || !ptrs || !bytes) return 0;
    const LitWorkspace W = lit_workspace(workspace, batch_size, num_tri, fill_back, texture_size);
    return lit_clear_ranges(W, grad_textures, textures_batch, grad_light, light_batch, batch_size, num_tri, fill_back,
                            texture_size, has_visibility != 0, ptrs, bytes);
}

D3M_EXPORT int d3m_backward_textures_lit(const float* faces, const float* textures, int textures_batch, const float* light,
                                         int light_batch, const int32_t* face_index_map, const float* weight_map,
                                         const float* depth_map, const float* grad_rgb_map, float* grad_textures,
                                         float* grad_light, const float* grad_depth_map, float* grad_faces, int batch_size,
                                         int num_tri, int fill_back, int image_size, int texture_size, float eps,
                                         void* workspace, size_t workspace_bytes, const d3m_vertex_target* vertex_target,
                                         void* visibility, const d3m_fit_targets* unscaled, int flags_in, d3m_stream_t stream) {
    if ((grad_depth_map != nullptr) != (grad_faces != nullptr || vertex_target != nullptr)) return D3M_ERR_INVALID;
    VertexTarget vt;
    if (int rcv = to_vertex_target(vertex_target, (fill_back ? 2 : 1) * num_tri, vt)) return rcv;
    const bool records = unscaled && unscaled->edge_grad;      // the rgb gradient is the yzw of the fit's per-pixel records
    if (!faces || !face_index_map || !weight_map || !depth_map || (!grad_rgb_map && !records) || !grad_textures ||
        batch_size <= 0 || image_size <= 0)
        return D3M_ERR_INVALID;
    const RgbGrad grad_rgb = records ? RgbGrad{(const float*)unscaled->edge_grad, 4, 1} : RgbGrad{grad_rgb_map, 3, 0};
    LitTextures lt;
    int rc = make_lit(lt, textures, textures_batch, light, light_batch, num_tri, texture_size, fill_back, batch_size);
    if (rc) return rc;
    if (lt.im) return D3M_ERR_INVALID;      // image-sourced textures are a forward-only form (the cube array is this pass's output)
    if (!workspace || workspace_bytes < d3m_backward_textures_lit_workspace_bytes(batch_size, num_tri, fill_back, texture_size))
        return D3M_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int B = batch_size, S = image_size;
    if (unscaled && !unscaled->scratch && !unscaled->edge_grad) return D3M_ERR_INVALID;    // (records without a scratch: final)
    const GradScale gs = to_grad_scale(unscaled, S);
    const size_t ts3 = (size_t)texture_size * texture_size * texture_size;
    const size_t view_elems = (size_t)num_tri * ts3 * 3;
    const LitWorkspace W = lit_workspace(workspace, B, num_tri, fill_back, texture_size);
    // per-view gradients: straight into grad_textures when every view has its own textures
    float* gview = textures_batch > 1 ? grad_textures : W.gview;
    int* flags = W.flags;
    const long n = (long)B * S * S, nf = (long)B * lt.Fp;
    unsigned* view_mask = W.view_mask;
    // the gathered pass stores (does not add) and the shared-texture sum skips unwritten entries by the flags
    const bool gathered_any = visibility && texture_size >= 3 && texture_size <= 4;      // (the LDS form: k_backward_textures_lit_faces_any)
    const bool skip_zero = (texture_size == 2 || gathered_any) && textures_batch == 1;
    // every clear of this entry point in one launch -- or none: a caller that zeroed the ranges of
    // d3m_backward_textures_lit_clear_ranges itself says so (D3M_PRECLEARED)
    if (!(flags_in & D3M_PRECLEARED)) {
        void* z_ptr[4];
        size_t z_bytes[4];
        const int nz = lit_clear_ranges(W, grad_textures, textures_batch, grad_light, light_batch, B, num_tri, fill_back,
                                        texture_size, visibility != nullptr, z_ptr, z_bytes);
        HIP_TRY(zero_ranges_async(z_ptr, z_bytes, nz, st));
    }
    const int* list = nullptr;
    const int* n_list = nullptr;
    if (visibility) {                                  // flags and the compacted list come from d3m_visibility
        const VisibilityView v = visibility_view(visibility, nf);
        flags = v.flags; list = v.list; n_list = v.count;
    }
    if (texture_size == 2) {
        const bool use_mask = skip_zero;               // shared textures: the sum over views reads only what was written
        if (!visibility)
            LAUNCH("k_mark_visible", k_mark_visible, dim3(blocks_for(n, 256)), dim3(256), st, face_index_map, flags, B, lt.Fp, S);
        LitFaceArgs fa{faces, lt, face_index_map, weight_map, depth_map, grad_rgb, gview, grad_light, grad_depth_map,
                       grad_faces, vt, flags, use_mask ? view_mask : nullptr, list, n_list, B, S, eps, gs, W.n_large,
                       deterministic_mode() ? INT_MAX : FM_MAX_BBOX_AREA};
        // (a coarse mesh -- more than 48 raster pixels per triangle -- gives a face a whole wave instead of eight lanes)
        const bool coarse = list && (double)S * S > 48.0 * (double)num_tri;
        const unsigned all_blocks = blocks_for(nf, coarse ? 4 : LIT_FACES_PER_BLOCK);
        const dim3 g_faces(list ? (all_blocks > 4096 ? 4096 : (all_blocks + 7) / 8 * 8) : all_blocks);
        if (coarse) LAUNCH("k_backward_textures_lit_faces", k_backward_textures_lit_faces<64>, g_faces, dim3(256), st, fa);
        else LAUNCH("k_backward_textures_lit_faces", k_backward_textures_lit_faces<>, g_faces, dim3(256), st, fa);
        // the texel and depth gradients of the faces the gathered pass marked LARGE (normally none: leaves at once) -- and, for
        // a fused objective whose forward pass deferred it (D3M_FIT_FINISH_DEFERRED), the objective's finish
        FitFin fin;
        memset(&fin, 0, sizeof(fin));
        if (records && unscaled->scratch && unscaled->loss && (unscaled->flags & D3M_FIT_FINISH_DEFERRED))
            fin = fit_fin_of_tiles(unscaled, B, S);
        LAUNCH("k_lit_large_faces", k_lit_large_faces, dim3(px_grid(n, true)), dim3(256), st, faces, lt, face_index_map,
               weight_map, depth_map, grad_rgb, gview, grad_light, (const int*)flags, B, S, eps, gs, (const int*)W.n_large,
               grad_depth_map, grad_faces, vt, fin);
    } else if (gathered_any) {
        // texture cubes of 3^3 / 4^3 texels over the visibility list: the same three launches as ts = 2, the face's sums in LDS
        LitFaceArgs fa{faces, lt, face_index_map, weight_map, depth_map, grad_rgb, gview, grad_light, grad_depth_map,
                       grad_faces, vt, flags, skip_zero ? view_mask : nullptr, list, n_list, B, S, eps, gs, W.n_large,
                       deterministic_mode() ? INT_MAX : FM_MAX_BBOX_AREA};
        const bool coarse = (double)S * S > 48.0 * (double)num_tri;          // (a wave per face: as at ts = 2)
        const unsigned all_blocks = blocks_for(nf, coarse ? 4 : LIT_FACES_PER_BLOCK);
        const dim3 g_faces(all_blocks > 4096 ? 4096 : (all_blocks + 7) / 8 * 8);
        if (coarse) LAUNCH("k_backward_textures_lit_faces", k_backward_textures_lit_faces_any<64>, g_faces, dim3(256), st, fa);
        else LAUNCH("k_backward_textures_lit_faces", k_backward_textures_lit_faces_any<>, g_faces, dim3(256), st, fa);
        FitFin fin;
        memset(&fin, 0, sizeof(fin));
        if (records && unscaled->scratch && unscaled->loss && (unscaled->flags & D3M_FIT_FINISH_DEFERRED))
            fin = fit_fin_of_tiles(unscaled, B, S);
        LAUNCH("k_lit_large_faces", k_lit_large_faces, dim3(px_grid(n, true)), dim3(256), st, faces, lt, face_index_map,
               weight_map, depth_map, grad_rgb, gview, grad_light, (const int*)flags, B, S, eps, gs, (const int*)W.n_large,
               grad_depth_map, grad_faces, vt, fin);
    } else {
        // (no kernel here for a deferred finish to ride in: a launch of its own)
        if (records && unscaled->scratch && unscaled->loss && (unscaled->flags & D3M_FIT_FINISH_DEFERRED)) {
            const FitFin fin = fit_fin_of_tiles(unscaled, B, S);
            LAUNCH("k_fit_finish", k_fit_finish, dim3(fin.n_groups), dim3(256), st, fin);
        }
        LAUNCH("k_backward_textures_lit_pixels", k_backward_textures_lit_pixels, dim3(px_grid(n, false)), dim3(256), st,
               faces, lt, face_index_map, weight_map, depth_map, grad_rgb, gview, grad_light, (const int*)nullptr, B, S, eps,
               gs, (const int*)nullptr);
        if (grad_depth_map) {
            DenseFaces fs{faces, lt.Fp};
            if (vt.gv) {
                LAUNCH("k_backward_depth_map", k_backward_depth_map<DenseFaces>, dim3(px_grid(n, false)), dim3(256), st, fs,
                       depth_map, face_index_map, (const float*)nullptr, weight_map, grad_depth_map, (float*)nullptr, B, S,
                       (const int*)nullptr, vt, gs);
            } else {
                if (unscaled) return D3M_ERR_INVALID;      // the dense-array depth path takes final gradient maps only
                const int rc2 = run_backward_depth(fs, depth_map, face_index_map, (const float*)nullptr, weight_map,
                                                   grad_depth_map, grad_faces, B, S, flags, false, st);
                if (rc2) return rc2;
            }
        }
    }
    if (textures_batch == 1) {
        if ((texture_size == 2 || texture_size == 4) && skip_zero && (((uintptr_t)gview | (uintptr_t)grad_textures) & 15) == 0)
            LAUNCH("k_sum_over_views", k_sum_over_views_ts2, dim3(blocks_for((long)view_elems / 4, 256)), dim3(256), st,
                   (const float4*)gview, (float4*)grad_textures, (long)view_elems / 4, B, (const unsigned*)view_mask,
                   (int)(ts3 * 3 / 4));
        else
            LAUNCH("k_sum_over_views", k_sum_over_views, dim3(blocks_for((long)view_elems, 256)), dim3(256), st,
                   (const float*)gview, grad_textures, (long)view_elems, B,
                   skip_zero ? (const int*)flags : (const int*)nullptr, num_tri, lt.Fp, (int)(ts3 * 3));
    }
    return check_launch();
}

D3M_EXPORT int d3m_output_epilogue(const int32_t* face_index_map, const float* rgb_map, const float* depth_map,
                                   const float* background, int background_batch, float* rgb_blended,
                                   float* alpha_map, float* rgb_out, float* alpha_out, float* depth_out, int batch_size,
                                   int image_size, int anti_aliasing, d3m_stream_t stream) {
    if (!face_index_map || batch_size <= 0 || image_size <= 0) return D3M_ERR_INVALID;
    if (anti_aliasing && (image_size & 1)) return D3M_ERR_INVALID;
    if (rgb_map && !background) return D3M_ERR_INVALID;
    if (background && background_batch != 1 && background_batch != batch_size) return D3M_ERR_INVALID;
    const int s = anti_aliasing ? image_size / 2 : image_size;
    const long n = (long)batch_size * s * s;
    LAUNCH("k_output_epilogue", k_output_epilogue, dim3(blocks_for(n, 256)), dim3(256), (hipStream_t)stream, face_index_map,
                       rgb_map, depth_map, background, background_batch, rgb_blended, alpha_map, rgb_out, alpha_out,
                       depth_out, batch_size, image_size, anti_aliasing ? 1 : 0);
    return check_launch();
}

D3M_EXPORT int d3m_output_epilogue_backward(const float* grad_rgb_out, const float* grad_alpha_out,
                                            const float* grad_depth_out, float* grad_rgb_map, float* grad_alpha_map,
                                            float* grad_depth_map, int batch_size, int image_size, int anti_aliasing,
                                            d3m_stream_t stream) {
    if (batch_size <= 0 || image_size <= 0) return D3M_ERR_INVALID;
    if ((grad_rgb_map && !grad_rgb_out) || (grad_alpha_map && !grad_alpha_out) || (grad_depth_map && !grad_depth_out))
        return D3M_ERR_INVALID;
    const long n = (long)batch_size * image_size * image_size;
    LAUNCH("k_output_epilogue_backward", k_output_epilogue_backward, dim3(blocks_for(n, 256)), dim3(256), (hipStream_t)stream,
                       grad_rgb_out, grad_alpha_out, grad_depth_out, grad_rgb_map, grad_alpha_map, grad_depth_map,
                       batch_size, image_size, anti_aliasing ? 1 : 0);
    return check_launch();
}

// The same adjoint with the rgb / alpha gradients leaving as the edge gradient's per-pixel records (no gradient maps:
// d3m_backward_pixel_map / d3m_backward_textures_lit take the records through `unscaled`, scratch NULL = final).
D3M_EXPORT int d3m_output_epilogue_backward_records(const float* grad_rgb_out, const float* grad_alpha_out,
                                                    const float* grad_depth_out, const int32_t* face_index_map,
                                                    const float* rgb_map, const float* alpha_map, void* edge_grad,
                                                    void* edge_dot, int* edge_nz_lo_inv, int* edge_nz_hi1,
                                                    float* grad_depth_map, int batch_size, int image_size,
                                                    int anti_aliasing, d3m_stream_t stream) {
    if (batch_size <= 0 || image_size <= 0 || !face_index_map || !edge_grad || !edge_dot || !edge_nz_lo_inv || !edge_nz_hi1)
        return D3M_ERR_INVALID;
    if ((grad_rgb_out && !rgb_map) || (grad_alpha_out && !alpha_map) || (grad_depth_map && !grad_depth_out)) return D3M_ERR_INVALID;
    if (anti_aliasing && (image_size & 1)) return D3M_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int S = image_size, B = batch_size;
    const size_t nz_bytes = (size_t)B * 2 * S * 4;
    void* z_ptr[2] = {edge_nz_lo_inv, edge_nz_hi1};
    size_t z_bytes[2] = {nz_bytes, nz_bytes};
    HIP_TRY(zero_ranges_async(z_ptr, z_bytes, 2, st));       // (one launch)
    const ImageGrads img{grad_rgb_out, grad_alpha_out, grad_depth_out, grad_depth_map, anti_aliasing ? S / 2 : S,
                         anti_aliasing ? 1 : 0, 1};
    LAUNCH("k_pack_maps", k_pack_maps, dim3((S + 31) / 32, (S + 31) / 32, B), dim3(256), st, face_index_map,
           grad_alpha_out ? alpha_map : (const float*)nullptr, (const float*)nullptr,
           grad_rgb_out ? rgb_map : (const float*)nullptr, (const float*)nullptr, (float4*)edge_grad, (float2*)edge_dot,
           edge_nz_lo_inv, edge_nz_hi1, S, GradScale{nullptr, nullptr, 0.0f, 0, nullptr}, img);
    return check_launch();
}

// ---------------------------------------------------------------------------------------------------
// C. losses
// ---------------------------------------------------------------------------------------------------
D3M_EXPORT int d3m_photometric_loss(const float* im1, const float* im2, const float* mask, const float* conf_sigma,
                                    float* loss, float* grad_im1, float* scratch, int batch_size, int channels,
                                    int height, int width, d3m_stream_t stream) {
    if (!im1 || !im2 || !loss || !scratch || batch_size <= 0 || channels <= 0 || height <= 0 || width <= 0)
        return D3M_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const long hw = (long)height * width;
    if (hw > 0x7FFFFFFF || (long)batch_size * channels > 65535) return D3M_ERR_INVALID;
    const unsigned planes = (unsigned)(batch_size * channels);
    unsigned gx = (unsigned)((hw + 2047) / 2048);
    const unsigned gx_cap = planes >= 1024 ? 1 : 1024 / planes;      // at most 1024 partial pairs (2048 floats)
    if (gx > gx_cap) gx = gx_cap;
    if (planes * gx > 1024) return D3M_ERR_INVALID;
    const dim3 grid(gx, planes);
    LAUNCH("k_photometric_reduce", k_photometric_reduce, grid, dim3(256), st, im1, im2, mask, conf_sigma, scratch, channels,
           (int)hw);
    LAUNCH("k_photometric_finish", k_photometric_finish, grad_im1 ? grid : dim3(1, 1), dim3(256), st, im1, im2, mask,
           conf_sigma, (const float*)scratch, (int)(planes * gx), loss, grad_im1, channels, (int)hw);
    return check_launch();
}

D3M_EXPORT int d3m_sum_squared_error(const float* a, const float* b, float* loss, float* grad_a, float* scratch, long n,
                                     d3m_stream_t stream) {
    if (!a || !b || !loss || !scratch || n <= 0) return D3M_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((n + 2047) / 2048 < 1024 ? (n + 2047) / 2048 : 1024);
    LAUNCH("k_sum_squared_error", k_sum_squared_error, dim3(grid), dim3(256), st, a, b, scratch, grad_a, n);
    LAUNCH("k_sum_partials", k_sum_partials, dim3(1), dim3(256), st, (const float*)scratch, (int)grid, loss);
    return check_launch();
}

static int smooth_dims(int B, int H, int W, float& n_xx, float& n_xy, float& n_yy) {
    if (B <= 0 || H < 3 || W < 3 || (long)B * H * W > 0x7FFFFFFFL) return D3M_ERR_INVALID;
    n_xx = (float)((long)B * H * (W - 2));
    n_xy = (float)((long)B * (H - 1) * (W - 1));
    n_yy = (float)((long)B * (H - 2) * W);
    return D3M_OK;
}

D3M_EXPORT int d3m_smooth_loss_forward(const float* pred, float* loss, float* scratch, int batch_size, int height, int width,
                                       d3m_stream_t stream) {
    if (!pred || !loss || !scratch) return D3M_ERR_INVALID;
    float n_xx, n_xy, n_yy;
    if (int rc = smooth_dims(batch_size, height, width, n_xx, n_xy, n_yy)) return rc;
    const long n = (long)batch_size * height * width;
    const unsigned grid = (unsigned)((n + 1023) / 1024 < 1024 ? (n + 1023) / 1024 : 1024);
    hipStream_t st = (hipStream_t)stream;
    LAUNCH("k_smooth_reduce", k_smooth_reduce, dim3(grid), dim3(256), st, pred, scratch, batch_size, height, width);
    LAUNCH("k_smooth_finish", k_smooth_finish, dim3(1), dim3(256), st, (const float*)scratch, (int)grid, n_xx, n_xy, n_yy, loss);
    return check_launch();
}

D3M_EXPORT int d3m_smooth_loss_backward(const float* pred, const float* grad_loss, float* grad_pred, int batch_size,
                                        int height, int width, d3m_stream_t stream) {
    if (!pred || !grad_loss || !grad_pred) return D3M_ERR_INVALID;
    float n_xx, n_xy, n_yy;
    if (int rc = smooth_dims(batch_size, height, width, n_xx, n_xy, n_yy)) return rc;
    const long n = (long)batch_size * height * width;
    LAUNCH("k_smooth_grad", k_smooth_grad, dim3(blocks_for(n, 256)), dim3(256), (hipStream_t)stream, pred, grad_loss, grad_pred,
           batch_size, height, width, n_xx, n_xy, n_yy);
    return check_launch();
}

// ---------------------------------------------------------------------------------------------------
// D. the gan2shape renderer block (d3m_g2s.h)
// ---------------------------------------------------------------------------------------------------
static inline int g2s_split(long pixels, int cap) {
    const long b = (pixels + 255) / 256;
    return (int)(b < 1 ? 1 : (b < cap ? b : cap));
}
struct G2SLayout { int split_s, split_f, off_sample, off_front, off_sback, total; };
static G2SLayout g2s_layout(int B, int H, int W, int s) {
    G2SLayout L;
    const long px = (long)s * s > (long)H * W ? (long)s * s : (long)H * W;     // the sample pass also sweeps the canonical maps
    L.split_s = g2s_split(px, 32);
    L.split_f = g2s_split((long)H * W, 32);
    int o = G2S_TOTALS;
    L.off_sample = o; o += G2S_SAMPLE_PART * B * L.split_s;
    L.off_front = o;  o += G2S_FRONT_SUMS * B * L.split_f;
    L.off_sback = o;  o += G2S_SBACK_SUMS * B * L.split_s;
    L.total = o;
    return L;
}
D3M_EXPORT size_t d3m_g2s_scratch_floats(int batch_size, int height, int width, int image_size) {
    if (batch_size <= 0 || height <= 0 || width <= 0 || image_size <= 0) return 0;
    return (size_t)g2s_layout(batch_size, height, width, image_size).total;
}

static int to_g2s(const d3m_g2s_block* h, G2S& g) {
    if (!h) return D3M_ERR_INVALID;
    const int B = h->batch_size, H = h->height, W = h->width, s = h->image_size;
    if (B <= 0 || H < 3 || W < 3 || s < 2 || (long)B * H * W > 0x3FFFFFFFL || (long)B * s * s > 0x0FFFFFFFL) return D3M_ERR_INVALID;
    if (h->flip && (B & 1)) return D3M_ERR_INVALID;
    if (s * (h->anti_aliasing ? 2 : 1) > 8192) return D3M_ERR_INVALID;      // the candidate walk packs pixel coordinates in 13 bits
    if (!h->inv_K || !h->K || !h->depth || !h->albedo || !h->light_a || !h->light_b || !h->light_d || !h->rot || !h->trans ||
        !h->diffuse_shading || !h->texture || !h->recon_im || !h->recon_im_mask || !h->losses || !h->screen_vertices ||
        !h->zbuffer || !h->scratch)
        return D3M_ERR_INVALID;
    if (h->view && h->view_components != 3 && h->view_components != 5 && h->view_components != 6) return D3M_ERR_INVALID;
    if ((h->inv_K_batch != 1 && h->inv_K_batch != B) || (h->K_batch != 1 && h->K_batch != B)) return D3M_ERR_INVALID;
    memset(&g, 0, sizeof(g));
    if (int rc = to_cam(h->camera, B, g.cam)) return rc;
    const G2SLayout L = g2s_layout(B, H, W, s);
    g.B = B; g.H = H; g.W = W; g.s = s; g.aa = h->anti_aliasing ? 1 : 0; g.S = g.aa ? 2 * s : s;
    g.flip = h->flip ? 1 : 0; g.Bh = g.flip ? B / 2 : B;
    g.inv_K = h->inv_K; g.invK_b = h->inv_K_batch; g.K = h->K; g.K_b = h->K_batch;
    g.center_z = h->rot_center_depth; g.depth_lo = h->depth_min; g.depth_hi = h->depth_max; g.near = h->near; g.far = h->far;
    g.view = h->view; g.view_n = h->view_components; g.rot = h->rot; g.trans = h->trans;
    g.depth = h->depth; g.albedo = h->albedo; g.light_a = h->light_a; g.light_b = h->light_b; g.light_d = h->light_d;
    g.target = h->target; g.extra_mask = h->extra_mask;
    g.normal = h->normal; g.diffuse = h->diffuse_shading; g.texture = h->texture; g.screen_vertices = h->screen_vertices;
    g.zbuf = (unsigned long long*)h->zbuffer; g.recon_depth = h->recon_depth; g.recon_im = h->recon_im;
    g.recon_mask = h->recon_im_mask; g.losses = h->losses; g.scratch = h->scratch;
    g.off_sample = L.off_sample; g.off_front = L.off_front; g.off_sback = L.off_sback;
    g.split_s = L.split_s; g.split_f = L.split_f;
    g.lam_smooth = h->lam_smooth; g.with_smooth = h->with_smooth ? 1 : 0;
    g.n_xx = g.n_xy = g.n_yy = 1.0f;
    if (g.with_smooth) if (int rc = smooth_dims(B, H, W, g.n_xx, g.n_xy, g.n_yy)) return rc;
    g.grad_recon_im = h->grad_recon_im; g.g_l1 = h->grad_l1; g.g_l1_flip = h->grad_l1_flip; g.g_smooth = h->grad_smooth;
    g.g_total = h->grad_total;
    g.grad_texture = h->grad_texture; g.grad_tri = h->grad_tri; g.grad_depth_map = h->grad_depth_map;
    g.grad_normal = h->grad_normal; g.grad_depth_mesh = h->grad_depth_mesh;
    g.grad_depth = h->grad_depth; g.grad_albedo = h->grad_albedo; g.grad_light_a = h->grad_light_a;
    g.grad_light_b = h->grad_light_b; g.grad_light_d = h->grad_light_d; g.grad_rot = h->grad_rot; g.grad_trans = h->grad_trans;
    g.grad_view = h->grad_view;
    return D3M_OK;
}

static int to_warp_resample(WarpResample& g, const float* depth, const float* inv_K, int inv_K_batch, const float* K, int K_batch,
                            const float* rot, const float* trans, float center_z, const float* src, int C, int B, int h, int w,
                            int H, int W) {
    if (!depth || !inv_K || !K || !rot || !trans || !src || B <= 0 || h < 2 || w < 2 || H <= 0 || W <= 0 || C <= 0 ||
        (long)B * h * w > 0x3FFFFFFFL)
        return D3M_ERR_INVALID;
    if ((inv_K_batch != 1 && inv_K_batch != B) || (K_batch != 1 && K_batch != B)) return D3M_ERR_INVALID;
    memset(&g, 0, sizeof(g));
    g.B = B; g.h = h; g.w = w; g.C = C; g.H = H; g.W = W; g.depth = depth; g.inv_K = inv_K; g.invK_b = inv_K_batch;
    g.K = K; g.K_b = K_batch; g.rot = rot; g.trans = trans; g.center_z = center_z; g.src = src;
    return D3M_OK;
}
D3M_EXPORT int d3m_warp_resample(const float* depth, const float* inv_K, int inv_K_batch, const float* K, int K_batch,
                                 const float* rot, const float* trans, float rot_center_depth, const float* src, int channels,
                                 const float* src_nearest, int channels_nearest, float* out, float* out_nearest,
                                 int batch_size, int height, int width, int src_height, int src_width, d3m_stream_t stream) {
    WarpResample g;
    if (int rc = to_warp_resample(g, depth, inv_K, inv_K_batch, K, K_batch, rot, trans, rot_center_depth, src, channels,
                                  batch_size, height, width, src_height, src_width))
        return rc;
    if (!out || (src_nearest && (!out_nearest || channels_nearest <= 0))) return D3M_ERR_INVALID;
    g.src_nearest = src_nearest; g.Cn = channels_nearest; g.out = out; g.out_nearest = out_nearest;
    LAUNCH("k_warp_resample", k_warp_resample, dim3(blocks_for((long)batch_size * height * width, 256)), dim3(256),
           (hipStream_t)stream, g);
    return check_launch();
}
D3M_EXPORT int d3m_warp_resample_partials(int height, int width) {
    return height > 0 && width > 0 ? g2s_split((long)height * width, 32) : 0;
}
D3M_EXPORT int d3m_warp_resample_backward(const float* depth, const float* inv_K, int inv_K_batch, const float* K, int K_batch,
                                          const float* rot, const float* trans, float rot_center_depth, const float* src,
                                          int channels, const float* grad_out, float* grad_src, float* grad_depth,
                                          float* partials, int batch_size, int height, int width, int src_height,
                                          int src_width, d3m_stream_t stream) {
    WarpResample g;
    if (int rc = to_warp_resample(g, depth, inv_K, inv_K_batch, K, K_batch, rot, trans, rot_center_depth, src, channels,
                                  batch_size, height, width, src_height, src_width))
        return rc;
    if (!grad_out || !partials) return D3M_ERR_INVALID;
    g.grad_out = grad_out; g.grad_src = grad_src; g.grad_depth = grad_depth; g.partials = partials;
    LAUNCH("k_warp_resample_backward", k_warp_resample_backward, dim3(d3m_warp_resample_partials(height, width), batch_size),
           dim3(256), (hipStream_t)stream, g);
    return check_launch();
}

D3M_EXPORT int d3m_g2s_forward(const d3m_g2s_block* block, d3m_stream_t stream) {
    G2S g;
    if (int rc = to_g2s(block, g)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int B = g.B, HW = g.H * g.W, Ft = 2 * (g.H - 1) * (g.W - 1);
    // the first pass also clears the z-buffer and (for the backward pass) the texture-gradient accumulator
    ZeroRanges z;
    z.p[0] = (uint32_t*)g.zbuf;         z.n[0] = (unsigned)((size_t)B * g.S * g.S * 2);
    z.p[1] = (uint32_t*)g.grad_texture; z.n[1] = g.grad_texture ? (unsigned)((size_t)B * 3 * HW) : 0u;
    LAUNCH("k_g2s_front", k_g2s_front, dim3(blocks_for((long)B * HW, 256)), dim3(256), st, g, z);
    LAUNCH("k_g2s_raster", k_g2s_raster, dim3(blocks_for((long)B * Ft, 4 * G2S_PW)), dim3(256), st, g);
    LAUNCH("k_g2s_sample", k_g2s_sample, dim3(g.split_s, g.Bh), dim3(256), st, g);
    LAUNCH("k_g2s_finish", k_g2s_finish, dim3(1), dim3(256), st, g);
    return check_launch();
}

D3M_EXPORT int d3m_g2s_backward(const d3m_g2s_block* block, d3m_stream_t stream) {
    G2S g;
    if (int rc = to_g2s(block, g)) return rc;
    if (!g.grad_texture || !g.grad_tri || !g.grad_depth_map || !g.grad_normal || !g.grad_depth_mesh || !g.grad_depth)
        return D3M_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int B = g.B, HW = g.H * g.W, Ft = 2 * (g.H - 1) * (g.W - 1);
    LAUNCH("k_g2s_sample_backward", k_g2s_sample_backward, dim3(g.split_s, B), dim3(256), st, g);
    LAUNCH("k_g2s_depth_faces", k_g2s_depth_faces, dim3(blocks_for((long)B * Ft, 4 * G2S_PW)), dim3(256), st, g);
    LAUNCH("k_g2s_front_backward", k_g2s_front_backward, dim3(g.split_f, B), dim3(256), st, g);
    // (+ B workgroups: the per-entry sums of the two passes above and the view's gradient -- g2s_finish_backward)
    LAUNCH("k_g2s_depth_backward", k_g2s_depth_backward, dim3(blocks_for((long)B * HW, 256) + B), dim3(256), st, g);
    return check_launch();
}

static int fit_loss_grid(int batch_size, long hw, dim3& grid) {
    if (batch_size <= 0 || hw <= 0 || hw > 0x7FFFFFFF || batch_size > 1024) return D3M_ERR_INVALID;
    unsigned gx = (unsigned)((hw + 1023) / 1024);
    const unsigned cap = 1024u / (unsigned)batch_size;            // at most 1024 partial records (4096 floats)
    if (gx > cap) gx = cap;
    if (gx == 0) gx = 1;
    grid = dim3(gx, (unsigned)batch_size);
    return D3M_OK;
}

D3M_EXPORT int d3m_fit_loss_forward(const float* rgb, const float* rgb_target, const float* depth, const float* depth_target,
                                    const float* alpha, const float* alpha_target, const float* mask, float* loss,
                                    float* scratch, const float* mask_sum, int batch_size, int height, int width,
                                    d3m_stream_t stream) {
    if (!rgb || !rgb_target || !depth || !depth_target || !alpha || !alpha_target || !mask || !loss || !scratch)
        return D3M_ERR_INVALID;
    dim3 grid;
    const long hw = (long)height * width;
    int rc = fit_loss_grid(batch_size, hw, grid);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    FitLossArgs a{rgb, rgb_target, depth, depth_target, alpha, alpha_target, mask, batch_size, (int)hw, (float)hw};
    LAUNCH("k_fit_loss_reduce", k_fit_loss_reduce, grid, dim3(256), st, a, scratch + 8);
    LAUNCH("k_fit_loss_finish", k_fit_loss_finish, dim3(1), dim3(256), st, (const float*)(scratch + 8),
           (int)(grid.x * grid.y), (float)hw, mask_sum, scratch, loss);
    return check_launch();
}

D3M_EXPORT int d3m_fit_loss_backward(const float* rgb, const float* rgb_target, const float* depth,
                                     const float* depth_target, const float* alpha, const float* alpha_target,
                                     const float* mask, const float* scratch, const float* grad_loss, float* grad_rgb,
                                     float* grad_depth, float* grad_alpha, int batch_size, int height, int width,
                                     d3m_stream_t stream) {
    if (!rgb || !rgb_target || !depth || !depth_target || !alpha || !alpha_target || !mask || !scratch)
        return D3M_ERR_INVALID;
    dim3 grid;
    const long hw = (long)height * width;
    int rc = fit_loss_grid(batch_size, hw, grid);
    if (rc) return rc;
    FitLossArgs a{rgb, rgb_target, depth, depth_target, alpha, alpha_target, mask, batch_size, (int)hw, (float)hw};
    LAUNCH("k_fit_loss_grad", k_fit_loss_grad, grid, dim3(256), (hipStream_t)stream, a, scratch, grad_loss, grad_rgb, grad_depth,
           grad_alpha);
    return check_launch();
}

// ---------------------------------------------------------------------------------------------------
// D. texture assets
// ---------------------------------------------------------------------------------------------------
D3M_EXPORT int d3m_load_textures(const float* image, const int32_t* is_update, const float* faces_uv, float* textures,
                                 int num_faces, int texture_size, int image_height, int image_width,
                                 int texture_wrapping, int use_bilinear, d3m_stream_t stream) {
    if (!image || !is_update || !faces_uv || !textures) return D3M_ERR_INVALID;
    if (num_faces <= 0 || texture_size < 2 || image_height <= 0 || image_width <= 0) return D3M_ERR_INVALID;
    if (texture_wrapping < 0 || texture_wrapping > 3) return D3M_ERR_INVALID;
    if ((long)texture_size * texture_size * texture_size > 0x7FFFFFFF) return D3M_ERR_INVALID;
    const long n = (long)num_faces * texture_size * texture_size * texture_size;
    LAUNCH("k_load_textures", k_load_textures, dim3(blocks_for(n, 256)), dim3(256), (hipStream_t)stream, image, is_update,
           faces_uv, textures, n, texture_size, image_height, image_width, texture_wrapping, use_bilinear ? 1 : 0);
    return check_launch();
}

D3M_EXPORT int d3m_create_texture_image(const float* vertices_all, const float* textures, float* image, int num_faces,
                                        int texture_size_in, int image_height, int image_width, int tile_width,
                                        float eps, d3m_stream_t stream) {
    if (!vertices_all || !textures || !image) return D3M_ERR_INVALID;
    if (num_faces <= 0 || texture_size_in <= 0 || image_height <= 0 || image_width <= 0 || tile_width <= 0)
        return D3M_ERR_INVALID;
    if (image_width % tile_width) return D3M_ERR_INVALID;
    const int tso = image_width / tile_width;
    if (image_height % tso) return D3M_ERR_INVALID;
    if ((long)(image_height / tso) * tile_width < num_faces) return D3M_ERR_INVALID;
    const long n = (long)image_height * image_width;
    LAUNCH("k_create_texture_image", k_create_texture_image, dim3(blocks_for(n, 256)), dim3(256), (hipStream_t)stream,
           vertices_all, textures, image, n, num_faces, texture_size_in, tso, tile_width, eps);
    return check_launch();
}

// ---------------------------------------------------------------------------------------------------
// E. the face3d utility rasterizer family (f64)
// ---------------------------------------------------------------------------------------------------
struct MeshWs {
    unsigned long long* zkey;     // [h*w]
    unsigned long long* tmp_key;  // [h*w]
    int *owner, *head;            // [h*w]
    int *big_list;                // [ntri]
    int *entries;                 // [3*ntri]
    int *next, *count, *cursor, *last_pixel;   // [nver(+1)]
    int *big_count;
    size_t bytes;
};

static MeshWs mesh_ws(void* base, int nver, int ntri, int h, int w) {
    const size_t px = (size_t)(h > 0 ? h : 0) * (w > 0 ? w : 0), nv = nver > 0 ? nver : 0, nt = ntri > 0 ? ntri : 0;
    char* p = (char*)base;
    size_t o = 0;
    auto take = [&](size_t n) { char* r = p + o; o += align_up(n, 256); return r; };
    MeshWs m;
    m.zkey = (unsigned long long*)take(px * 8);
    m.tmp_key = (unsigned long long*)take(px * 8);
    m.owner = (int*)take(px * 4);
    m.head = (int*)take(px * 4);
    m.big_list = (int*)take(nt * 4);
    m.entries = (int*)take(nt * 12);
    m.next = (int*)take(nv * 4);
    m.count = (int*)take((nv + 2) * 4);
    m.cursor = (int*)take(nv * 4);
    m.last_pixel = (int*)take(nv * 4);
    m.big_count = (int*)take(256);
    m.bytes = o;
    return m;
}

D3M_EXPORT size_t d3m_mesh_workspace_bytes(int nver, int ntri, int h, int w) { return mesh_ws(nullptr, nver, ntri, h, w).bytes; }

// the shared two-pass resolve: zkey / owner of every pixel
static int mesh_resolve(const MeshTris& m, const double* depth_buffer, const MeshWs& ws, hipStream_t st) {
    const long px = (long)m.h * m.w;
    LAUNCH("k_mesh_init", k_mesh_init, dim3(blocks_for(px, 256)), dim3(256), st, depth_buffer, ws.zkey, ws.owner, ws.big_count, px);
    const dim3 gt(blocks_for(m.ntri, 256)), gb(1024);
    LAUNCH("k_mesh_depth", k_mesh_tris<0>, gt, dim3(256), st, m, depth_buffer, ws.zkey, ws.owner, ws.big_list, ws.big_count);
    LAUNCH("k_mesh_depth_big", k_mesh_big_tris<0>, gb, dim3(256), st, m, depth_buffer, ws.zkey, ws.owner,
           (const int*)ws.big_list, (const int*)ws.big_count);
    LAUNCH("k_mesh_owner", k_mesh_tris<1>, gt, dim3(256), st, m, depth_buffer, ws.zkey, ws.owner, ws.big_list, ws.big_count);
    LAUNCH("k_mesh_owner_big", k_mesh_big_tris<1>, gb, dim3(256), st, m, depth_buffer, ws.zkey, ws.owner,
           (const int*)ws.big_list, (const int*)ws.big_count);
    return check_launch();
}

static int mesh_args_ok(const void* a, const void* b, const void* c, const void* d, int nver, int ntri, int h, int w,
                        const void* ws, size_t ws_bytes) {
    if (!a || !b || !c || !d || nver <= 0 || ntri <= 0 || h <= 0 || w <= 0) return D3M_ERR_INVALID;
    if ((long)h * w > 0x7FFFFFF0L) return D3M_ERR_INVALID;
    if (!ws || ws_bytes < d3m_mesh_workspace_bytes(nver, ntri, h, w)) return D3M_ERR_WORKSPACE;
    return D3M_OK;
}

D3M_EXPORT int d3m_mesh_render_colors(double* image, const double* vertices, const int32_t* triangles,
                                      const double* tri_depth, const double* tri_tex, double* depth_buffer, int nver,
                                      int ntri, int h, int w, int c, void* workspace, size_t workspace_bytes,
                                      d3m_stream_t stream) {
    if (int rc = mesh_args_ok(image, vertices, triangles, tri_depth, nver, ntri, h, w, workspace, workspace_bytes)) return rc;
    if (!tri_tex || !depth_buffer || c <= 0) return D3M_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const MeshWs ws = mesh_ws(workspace, nver, ntri, h, w);
    const MeshTris m{vertices, triangles, tri_depth, nver, ntri, h, w};
    if (int rc = mesh_resolve(m, depth_buffer, ws, st)) return rc;
    MeshShade s{};
    s.image = image; s.tri_tex = tri_tex; s.c = c;
    LAUNCH("k_mesh_shade", k_mesh_shade<MESH_COLORS>, dim3(blocks_for((long)h * w, 256)), dim3(256), st, m, s,
           (const int*)ws.owner, depth_buffer);
    return check_launch();
}

D3M_EXPORT int d3m_mesh_render_texture(double* image, const double* vertices, const int32_t* triangles,
                                       const double* texture, const double* tex_coords, const int32_t* tex_triangles,
                                       const double* tri_depth, double* depth_buffer, int nver, int tex_nver, int ntri, int h,
                                       int w, int c, int tex_h, int tex_w, int tex_c, int mapping_type, void* workspace,
                                       size_t workspace_bytes, d3m_stream_t stream) {
    if (int rc = mesh_args_ok(image, vertices, triangles, tri_depth, nver, ntri, h, w, workspace, workspace_bytes)) return rc;
    if (!texture || !tex_coords || !tex_triangles || !depth_buffer || c <= 0 || tex_h <= 0 || tex_w <= 0 || tex_c < c)
        return D3M_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const MeshWs ws = mesh_ws(workspace, nver, ntri, h, w);
    const MeshTris m{vertices, triangles, tri_depth, nver, ntri, h, w};
    if (int rc = mesh_resolve(m, depth_buffer, ws, st)) return rc;
    MeshShade s{};
    s.image = image; s.texture = texture; s.tex_coords = tex_coords; s.tex_triangles = tex_triangles; s.c = c;
    s.tex_nver = tex_nver; s.tex_h = tex_h; s.tex_w = tex_w; s.tex_c = tex_c; s.bilinear = mapping_type != 0;
    LAUNCH("k_mesh_shade", k_mesh_shade<MESH_TEXTURE>, dim3(blocks_for((long)h * w, 256)), dim3(256), st, m, s,
           (const int*)ws.owner, depth_buffer);
    return check_launch();
}

D3M_EXPORT int d3m_mesh_get_triangle_buffer(int32_t* triangle_buffer, const double* vertices, const int32_t* triangles,
                                            const double* tri_depth, double* depth_buffer, int nver, int ntri, int h, int w,
                                            void* workspace, size_t workspace_bytes, d3m_stream_t stream) {
    if (int rc = mesh_args_ok(triangle_buffer, vertices, triangles, tri_depth, nver, ntri, h, w, workspace, workspace_bytes))
        return rc;
    if (!depth_buffer) return D3M_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const MeshWs ws = mesh_ws(workspace, nver, ntri, h, w);
    const MeshTris m{vertices, triangles, tri_depth, nver, ntri, h, w};
    if (int rc = mesh_resolve(m, depth_buffer, ws, st)) return rc;
    MeshShade s{};
    s.triangle_buffer = triangle_buffer;
    LAUNCH("k_mesh_shade", k_mesh_shade<MESH_TRIANGLE_BUFFER>, dim3(blocks_for((long)h * w, 256)), dim3(256), st, m, s,
           (const int*)ws.owner, depth_buffer);
    return check_launch();
}

D3M_EXPORT int d3m_mesh_vis_of_vertices(double* vis, const double* vertices, const int32_t* triangles,
                                        const double* tri_depth, double* depth_buffer, double* depth_tmp, int nver, int ntri,
                                        int h, int w, void* workspace, size_t workspace_bytes, d3m_stream_t stream) {
    if (int rc = mesh_args_ok(vis, vertices, triangles, tri_depth, nver, ntri, h, w, workspace, workspace_bytes)) return rc;
    if (!depth_buffer || !depth_tmp) return D3M_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const MeshWs ws = mesh_ws(workspace, nver, ntri, h, w);
    const MeshTris m{vertices, triangles, tri_depth, nver, ntri, h, w};
    if (int rc = mesh_resolve(m, depth_buffer, ws, st)) return rc;
    const long px = (long)h * w;
    MeshShade s{};
    LAUNCH("k_mesh_shade", k_mesh_shade<MESH_DEPTH_ONLY>, dim3(blocks_for(px, 256)), dim3(256), st, m, s, (const int*)ws.owner,
           depth_buffer);
    LAUNCH("k_fill_i32", k_fill_i32, dim3(blocks_for(px, 256)), dim3(256), st, ws.head, -1, px);
    HIP_TRY(zero_async(ws.tmp_key, align_up((size_t)px * 8, 256), st));
    LAUNCH("k_mesh_vis_chain", k_mesh_vis_chain, dim3(blocks_for(nver, 256)), dim3(256), st, vertices, (const double*)depth_buffer,
           (const double*)depth_tmp, ws.head, ws.next, nver, h, w);
    LAUNCH("k_mesh_vis_resolve", k_mesh_vis_resolve, dim3(blocks_for(nver, 256)), dim3(256), st, vertices, (const int*)ws.head,
           (const int*)ws.next, vis, ws.tmp_key, nver, h, w);
    LAUNCH("k_mesh_vis_finish", k_mesh_vis_finish, dim3(blocks_for(px, 256)), dim3(256), st,
           (const unsigned long long*)ws.tmp_key, depth_tmp, px);
    return check_launch();
}

D3M_EXPORT int d3m_mesh_map_texture(double* dst_image, const double* src_image, const double* dst_vertices,
                                    const double* src_vertices, const int32_t* dst_triangle_buffer, const int32_t* triangles,
                                    int nver, int ntri, int sh, int sw, int sc, int h, int w, int c, d3m_stream_t stream) {
    if (!dst_image || !src_image || !dst_vertices || !src_vertices || !dst_triangle_buffer || !triangles || nver <= 0 ||
        ntri <= 0 || sh <= 0 || sw <= 0 || sc < c || h <= 0 || w <= 0 || c <= 0)
        return D3M_ERR_INVALID;
    LAUNCH("k_mesh_map_texture", k_mesh_map_texture, dim3(blocks_for((long)h * w, 256)), dim3(256), (hipStream_t)stream, dst_image,
           src_image, dst_vertices, src_vertices, dst_triangle_buffer, triangles, nver, ntri, sh, sw, sc, h, w, c);
    return check_launch();
}

D3M_EXPORT int d3m_mesh_get_norm_direction(double* norm, const double* tri_norm, const int32_t* triangles, int nver, int ntri,
                                           void* workspace, size_t workspace_bytes, d3m_stream_t stream) {
    if (!norm || !tri_norm || !triangles || nver <= 0 || ntri <= 0) return D3M_ERR_INVALID;
    if (!workspace || workspace_bytes < d3m_mesh_workspace_bytes(nver, ntri, 0, 0)) return D3M_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const MeshWs ws = mesh_ws(workspace, nver, ntri, 0, 0);
    HIP_TRY(zero_async(ws.count, align_up((size_t)(nver + 2) * 4, 256), st));
    HIP_TRY(zero_async(ws.cursor, align_up((size_t)nver * 4, 256), st));
    const dim3 ge(blocks_for(3L * ntri, 256));
    LAUNCH("k_mesh_incidence_count", k_mesh_incidence_count, ge, dim3(256), st, triangles, ws.count, ntri, nver);
    LAUNCH("k_scan_small", k_scan_small, dim3(1), dim3(1024), st, ws.count, nver, (const int*)nullptr, 1, ws.big_count);
    LAUNCH("k_mesh_incidence_fill", k_mesh_incidence_fill, ge, dim3(256), st, triangles, (const int*)ws.count, ws.cursor,
           ws.entries, ntri, nver);
    LAUNCH("k_mesh_normals", k_mesh_normals, dim3(blocks_for(nver, 256)), dim3(256), st, norm, tri_norm, (const int*)ws.count,
           ws.entries, nver, ntri);
    return check_launch();
}

D3M_EXPORT int d3m_mesh_get_correspondence(const double* image, const double* pncc_code, double* uv, int nver, int h, int w,
                                           int c, void* workspace, size_t workspace_bytes, d3m_stream_t stream) {
    if (!image || !pncc_code || !uv || nver <= 0 || h <= 0 || w <= 0 || c < 3) return D3M_ERR_INVALID;
    if (!workspace || workspace_bytes < d3m_mesh_workspace_bytes(nver, 0, 0, 0)) return D3M_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const MeshWs ws = mesh_ws(workspace, nver, 0, 0, 0);
    LAUNCH("k_fill_i32", k_fill_i32, dim3(blocks_for(nver, 256)), dim3(256), st, ws.last_pixel, -1, (long)nver);
    LAUNCH("k_mesh_nearest_code", k_mesh_nearest_code, dim3(blocks_for((long)h * w, 256)), dim3(256), st, image, pncc_code,
           ws.last_pixel, nver, h, w, c);
    LAUNCH("k_mesh_write_uv", k_mesh_write_uv, dim3(blocks_for(nver, 256)), dim3(256), st, (const int*)ws.last_pixel, uv, nver, w);
    return check_launch();
}

D3M_EXPORT int d3m_mesh_triangle_mean(const double* values, const int32_t* triangles, double* out, int channels, int nver,
                                      int ntri, d3m_stream_t stream) {
    if (!values || !triangles || !out || channels <= 0 || nver <= 0 || ntri <= 0) return D3M_ERR_INVALID;
    LAUNCH("k_mesh_triangle_mean", k_mesh_triangle_mean, dim3(blocks_for((long)channels * ntri, 256)), dim3(256),
           (hipStream_t)stream, values, triangles, out, channels, nver, ntri);
    return check_launch();
}

D3M_EXPORT int d3m_mesh_triangle_normals(const double* vertices, const int32_t* triangles, double* tri_norm, int nver, int ntri,
                                         d3m_stream_t stream) {
    if (!vertices || !triangles || !tri_norm || nver <= 0 || ntri <= 0) return D3M_ERR_INVALID;
    LAUNCH("k_mesh_triangle_normals", k_mesh_triangle_normals, dim3(blocks_for(ntri, 256)), dim3(256), (hipStream_t)stream,
           vertices, triangles, tri_norm, nver, ntri);
    return check_launch();
}

D3M_EXPORT int d3m_mesh_normalize(double* norm, int nver, d3m_stream_t stream) {
    if (!norm || nver <= 0) return D3M_ERR_INVALID;
    LAUNCH("k_mesh_normalize", k_mesh_normalize, dim3(blocks_for(nver, 256)), dim3(256), (hipStream_t)stream, norm, nver);
    return check_launch();
}
