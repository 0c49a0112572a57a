// d3m_forward.h -- forward coverage for gfx950: bin faces to 8x8 screen tiles, then one wave64 per tile.
//
// Replaces the reference's brute-force pair of kernels (KCU:24-169: every pixel streams every face)
// with:  k_bin_count -> k_bin_alloc -> k_bin_fill -> k_raster_tiles.
// The per-(pixel, face) arithmetic is d3m_device.h's restatement of KCU:110-139, so as long as the
// binning is conservative the selected face, weights and depth are bit-identical to brute force; the
// reference's "strict < in ascending face order" (KCU:142) is reproduced as a lexicographic minimum of
// (depth, face index) taken with 64-bit LDS atomics, which makes the result independent of list order.
#pragma once
#include "d3m_device.h"

namespace d3m {

struct BinBuffers {
    int B, F, S, tiles_x, tiles_y, T, kcap;     // T = tiles_x * tiles_y tiles of TILE_W x TILE_H pixels per view
    void* rect;         // [B*F]  tile rectangle of each face: 4 bytes (Rect32) on rasters of up to 1024 x 1024 tiles, else 8 (Rect64)
    int* tile_count;    // [B*T]  zeroed per call
    int* tile_cursor;   // [B*T]  zeroed per call
    int* big_count;     // [B]    zeroed per call
    int* alloc_cursor;  // [1]    zeroed per call
    int* tile_offset;   // [B*T]
    int* big_list;      // [B*F]  faces covering more than kcap tiles: scanned by every tile of the view
    int* pairs;         // [kcap*B*F] concatenated per-tile face lists
};

constexpr uint32_t RECT_NONE = 0xFFFFFFFFu;
// A face's tile rectangle as the counting pass leaves it for the list-filling pass: origin and extent in tiles, "none"
// (culled, or no pixel centre in its box) or "big" (more than kcap tiles: listed per view instead).  The 8-byte form
// (origin | far corner, 16 bits each) was 51 MB of stores and as many bytes of loads per headline step; on rasters of up to
// 1024 x 1024 tiles (S <= 8192) origin (10 + 10 bits) and extent - 1 (6 + 6 bits: a listed rectangle has at most kcap <= 64
// tiles) fit four bytes.
struct TileRect { int tx0, ty0, w, h; bool none, big; };
struct Rect64 {
    typedef uint2 T;
    static __device__ __forceinline__ T pack(int tx0, int ty0, int tx1, int ty1, bool big) {
        (void)big;
        return make_uint2((uint32_t)tx0 | ((uint32_t)ty0 << 16), (uint32_t)tx1 | ((uint32_t)ty1 << 16));
    }
    static __device__ __forceinline__ T none() { return make_uint2(RECT_NONE, 0); }
    static __device__ __forceinline__ TileRect unpack(T r, int kcap) {
        TileRect t;
        t.none = r.x == RECT_NONE;
        t.tx0 = r.x & 0xFFFF; t.ty0 = r.x >> 16;
        t.w = (int)(r.y & 0xFFFF) - t.tx0 + 1; t.h = (int)(r.y >> 16) - t.ty0 + 1;
        t.big = !t.none && t.w * t.h > kcap;
        return t;
    }
};
struct Rect32 {
    typedef uint32_t T;
    static constexpr uint32_t BIG = 0xFFFFFFFEu;          // (extent - 1 = 63 x 63: no listed rectangle)
    static __device__ __forceinline__ T pack(int tx0, int ty0, int tx1, int ty1, bool big) {
        if (big) return BIG;
        return (uint32_t)tx0 | ((uint32_t)ty0 << 10) | ((uint32_t)(tx1 - tx0) << 20) | ((uint32_t)(ty1 - ty0) << 26);
    }
    static __device__ __forceinline__ T none() { return RECT_NONE; }
    static __device__ __forceinline__ TileRect unpack(T r, int kcap) {
        (void)kcap;
        TileRect t;
        t.none = r == RECT_NONE; t.big = r == BIG;
        t.tx0 = r & 0x3FF; t.ty0 = (r >> 10) & 0x3FF; t.w = (int)((r >> 20) & 0x3F) + 1; t.h = (int)(r >> 26) + 1;
        return t;
    }
};

// Conservative pixel bounding box of a face, in the raster's pixel grid.  NDC -> pixel is
// p = (v*S + S - 1)/2 (KCU:47).  The box is dilated by a rounding margin so that no pixel that passes
// the f32 half-plane tests of KCU:115-117 can fall outside it (DESIGN.md "Binning is conservative").
// Returns false when the face cannot cover any pixel.
__device__ __forceinline__ bool pixel_bbox(const float* f, int S, int& x0, int& x1, int& y0, int& y1) {
    // three coincident xy: every entry of the face inverse is 0/0, weights are NaN, never selected
    if (f[0] == f[3] && f[3] == f[6] && f[1] == f[4] && f[4] == f[7]) return false;
    const bool finite = __builtin_isfinite(f[0]) && __builtin_isfinite(f[1]) && __builtin_isfinite(f[3]) &&
                        __builtin_isfinite(f[4]) && __builtin_isfinite(f[6]) && __builtin_isfinite(f[7]);
    if (!finite) {  // comparisons with NaN are false -> such a face "covers" everything; evaluate it everywhere
        x0 = 0; y0 = 0; x1 = S - 1; y1 = S - 1;
        return true;
    }
    const float xmn = fminf(f[0], fminf(f[3], f[6])), xmx = fmaxf(f[0], fmaxf(f[3], f[6]));
    const float ymn = fminf(f[1], fminf(f[4], f[7])), ymx = fmaxf(f[1], fmaxf(f[4], f[7]));
    const double m = 4e-6 * ((double)fmaxf(fmaxf(fabsf(xmn), fabsf(xmx)), fmaxf(fabsf(ymn), fabsf(ymx))) + 1.0);
    const double lx = ceil((((double)xmn - m) * S + S - 1) * 0.5), hx = floor((((double)xmx + m) * S + S - 1) * 0.5);
    const double ly = ceil((((double)ymn - m) * S + S - 1) * 0.5), hy = floor((((double)ymx + m) * S + S - 1) * 0.5);
    if (lx > S - 1 || hx < 0 || ly > S - 1 || hy < 0 || lx > hx || ly > hy) return false;
    x0 = d2i(fmax(lx, 0.0));
    y0 = d2i(fmax(ly, 0.0));
    x1 = d2i(fmin(hx, (double)(S - 1)));
    y1 = d2i(fmin(hy, (double)(S - 1)));
    return true;
}

// ---- per-workgroup aggregation of tile counters ---------------------------------------------------------------
// The 256 consecutive faces of a workgroup land in a few dozen tiles.  Instead of one (wave-merged) global atomic per
// (face, tile) pair, the pairs are first counted in a small LDS hash table keyed by the tile (LDS atomics), and each
// distinct tile then costs ONE global atomic per workgroup.  The wave-merged form spent ~110 us per binning pass on
// its serial group-discovery loops.
#ifndef D3M_BIN_THREADS
#define D3M_BIN_THREADS 1024
#endif
constexpr int BIN_THREADS = D3M_BIN_THREADS;      // faces (face pairs with fill_back) per binning workgroup, at most: its
                                                  // tile counters cost one global atomic each per distinct tile
constexpr int BIN_THREADS_SMALL = 256;            // ... when 1024 per workgroup would leave most of the chip idle
constexpr int TA_BITS = BIN_THREADS >= 1024 ? 11 : BIN_THREADS >= 512 ? 10 : 9;
constexpr int TA_SLOTS = 1 << TA_BITS;   // a workgroup touches far fewer distinct tiles
constexpr int TA_PROBES = 12;
constexpr int BIN_TABLE_TILES = 4;       // tiles of one face that are counted / ranked through the workgroup's table
// the part of the table a launch uses: two slots per lane (a 256-lane workgroup sweeps 512 slots, not 2048)
__device__ __forceinline__ int ta_bits() { return blockDim.x >= 1024 ? TA_BITS : blockDim.x >= 512 ? TA_BITS - 1 : TA_BITS - 2; }
struct TileAgg {
    int key[TA_SLOTS];                   // tile id + 1, 0 = empty
    int cnt[TA_SLOTS];
    int base[TA_SLOTS];
};

__device__ __forceinline__ void ta_clear(TileAgg& t) {
    for (int k = threadIdx.x; k < (1 << ta_bits()); k += blockDim.x) { t.key[k] = 0; t.cnt[k] = 0; }
    __syncthreads();
}

// slot of `tile` (inserted if new) and this pair's rank among the workgroup's pairs of that tile; -1: table too full
__device__ __forceinline__ int ta_add(TileAgg& t, int tile, int& rank) {
    const int bits = ta_bits();
    unsigned h = ((unsigned)tile * 2654435761u) >> (32 - bits);
    for (int p = 0; p < TA_PROBES; p++, h = (h + 1) & ((1u << bits) - 1)) {
        int k = t.key[h];
        if (k == 0) k = atomicCAS(&t.key[h], 0, tile + 1);
        if (k == 0 || k == tile + 1) {
            rank = atomicAdd(&t.cnt[h], 1);
            return (int)h;
        }
    }
    return -1;
}

// ---- pass 1: one lane per (view, face): cull, tile rectangle, per-tile counts ---------------------
// Also fills the reference's faces_inv scratch (KCU:24-67) when the caller passes it, and -- when the faces come
// from an indexed mesh -- the dense [B,F,3,3] copy of every front-facing face that the later passes read
// (vertices_to_faces + fill_back without a pass of its own; culled faces are never read again).
// PAIRED (an indexed mesh with fill_back): face Ft+f is face f with its vertices in reverse order, and at most one
// of the two is front-facing -- one lane loads the three vertices once and handles both copies, instead of a second
// lane repeating the index and vertex gathers only to find its copy culled.
template <class FS, bool PAIRED, class RECT>
__global__ void __launch_bounds__(BIN_THREADS) k_bin_count(FS fs, BinBuffers bb, float* __restrict__ faces_inv,
                                                  float* __restrict__ faces_dense_out,
                                                  unsigned char* __restrict__ marks = nullptr,
                                                  int* __restrict__ marks_count = nullptr) {
    __shared__ TileAgg agg;
    if (marks_count && blockIdx.x == 0 && threadIdx.x == 0) *marks_count = 0;
    ta_clear(agg);
    const int F = bb.F, Fl = PAIRED ? F / 2 : F;          // faces per view: all / handled by one lane each
    const long lane_i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = lane_i < (long)bb.B * Fl;
    const int b = in_range ? (int)(lane_i / Fl) : 0, f0 = in_range ? (int)(lane_i % Fl) : 0;
    float loaded[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (in_range) fs.load(b, f0, loaded);
#pragma unroll
    for (int side = 0; side < (PAIRED ? 2 : 1); side++) {
        const int f = f0 + side * Fl;
        const long i = (long)b * F + f;
        float face[9];
#pragma unroll
        for (int k = 0; k < 9; k++) face[k] = side == 0 ? loaded[k] : loaded[(2 - k / 3) * 3 + k % 3];
        typename RECT::T r = RECT::none();
        int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;     // empty
        bool small = false;
        if (in_range) {
            if (!backside(face)) {
                if (faces_inv) {
                    float fi[9];
                    face_inverse(face, bb.S, fi);
#pragma unroll
                    for (int k = 0; k < 9; k++) faces_inv[i * 9 + k] = fi[k];
                }
                int x0, x1, y0, y1;
                if (pixel_bbox(face, bb.S, x0, x1, y0, y1)) {
                    if (faces_dense_out) {            // (only faces whose box holds a pixel centre are ever read again)
#pragma unroll
                        for (int k = 0; k < 9; k++) faces_dense_out[i * 9 + k] = face[k];
                    }
                    tx0 = x0 / TILE_W; tx1 = x1 / TILE_W; ty0 = y0 / TILE_H; ty1 = y1 / TILE_H;
                    small = (tx1 - tx0 + 1) * (ty1 - ty0 + 1) <= bb.kcap;
                    r = RECT::pack(tx0, ty0, tx1, ty1, !small);
                    if (!small) {
                        const int pos = atomicAdd(&bb.big_count[b], 1);
                        bb.big_list[(size_t)b * F + pos] = f;
                    }
                }
            }
            reinterpret_cast<typename RECT::T*>(bb.rect)[i] = r;
            if (marks) marks[i] = 0;                  // "owns a pixel": set by the tile pass (RasterOut::marks)
        }
        // count the (at most kcap) tiles of every small face in the workgroup's table, then one atomic per distinct tile
        const int w = small ? tx1 - tx0 + 1 : 0, nt = small ? w * (ty1 - ty0 + 1) : 0;
        // (only a face's first BIN_TABLE_TILES tiles go through the table, as in k_bin_fill: the table is for the many small
        //  faces that share a tile; a coarse mesh's faces of 16-64 tiles each overflowed it and spent the pass probing --
        //  722 triangles @512^2: 109 us for three workgroups -- where plain atomics on their own tiles cost nothing)
        for (int s = 0; s < nt; s++) {
            const int tile = b * bb.T + (ty0 + s / w) * bb.tiles_x + tx0 + s % w;
            int rank;
            if (s >= BIN_TABLE_TILES || ta_add(agg, tile, rank) < 0) atomicAdd(&bb.tile_count[tile], 1);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < (1 << ta_bits()); k += blockDim.x)
        if (agg.key[k]) atomicAdd(&bb.tile_count[agg.key[k] - 1], agg.cnt[k]);
}

// ---- pass 2: give every tile a slice of `pairs` (one atomic per 256 tiles; order is irrelevant) ---
// 1024 tiles per workgroup: the workgroups' returning atomics all hit ONE address and serialise there (~15 ns each)
constexpr int BIN_ALLOC_THREADS = 1024;
__global__ void __launch_bounds__(BIN_ALLOC_THREADS) k_bin_alloc(BinBuffers bb) {
    __shared__ int s_wave[BIN_ALLOC_THREADS / 64];
    __shared__ int s_base;
    const int n = bb.B * bb.T;
    const int i = blockIdx.x * BIN_ALLOC_THREADS + threadIdx.x;
    const int c = i < n ? bb.tile_count[i] : 0;
    const int incl = wave_inclusive_scan(c);
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int k = 0; k < BIN_ALLOC_THREADS / 64; k++) { const int tk = s_wave[k]; s_wave[k] = run; run += tk; }
        s_base = atomicAdd(bb.alloc_cursor, run);
    }
    __syncthreads();
    if (i < n) bb.tile_offset[i] = s_base + s_wave[wv] + incl - c;
}

// ---- pass 3: scatter face ids into the per-tile lists ------------------------------------------
// PAIRED (fill_back): one lane per (face f, face F/2 + f) pair -- normally exactly one of the two has a rectangle, so
// every lane has work; should both have one (zero-area faces), the second goes straight to the global cursors.
// FPT face (pairs) per thread: the table then aggregates FPT x the faces, and the cursors' returning atomics -- one per
// distinct tile and workgroup, what bounds this pass on dense meshes (a workgroup's consecutive sub-pixel faces are a
// thin strip through hundreds of tiles) -- are shared by FPT x as many pairs.  (Reading a rectangle is cheap; the count
// pass, which loads and tests the faces, was slower with more than one face per thread.)
template <bool PAIRED, int FPT, class RECT>
__global__ void __launch_bounds__(BIN_THREADS) k_bin_fill(BinBuffers bb) {
    const typename RECT::T* __restrict__ rects = reinterpret_cast<const typename RECT::T*>(bb.rect);
    __shared__ TileAgg agg;
    ta_clear(agg);
    const int Fl = PAIRED ? bb.F / 2 : bb.F;
    constexpr int TA_LOCAL = BIN_TABLE_TILES;
    int packed[FPT][TA_LOCAL], face_of[FPT];
#pragma unroll
    for (int it = 0; it < FPT; it++) {
        const long lane_i = ((long)blockIdx.x * FPT + it) * blockDim.x + threadIdx.x;
        int tx0 = 0, ty0 = 0, w = 0, nt = 0, b = 0, f = 0;
        if (lane_i < (long)bb.B * Fl) {
            b = (int)(lane_i / Fl);
            const int f0 = (int)(lane_i % Fl);
            TileRect r = RECT::unpack(rects[(size_t)b * bb.F + f0], bb.kcap);
            f = f0;
            if (PAIRED) {
                const TileRect r1 = RECT::unpack(rects[(size_t)b * bb.F + f0 + Fl], bb.kcap);
                if (r.none) { r = r1; f = f0 + Fl; }
                else if (!r1.none && !r1.big) {           // both copies listed: the second one the plain way
                    const int an = r1.w * r1.h;
                    for (int s2 = 0; s2 < an; s2++) {
                        const int tile = b * bb.T + (r1.ty0 + s2 / r1.w) * bb.tiles_x + r1.tx0 + s2 % r1.w;
                        const int pos = atomicAdd(&bb.tile_cursor[tile], 1);
                        bb.pairs[(size_t)bb.tile_offset[tile] + pos] = f0 + Fl;
                    }
                }
            }
            if (!r.none && !r.big) {                      // (a big face lives in big_list)
                tx0 = r.tx0; ty0 = r.ty0;
                w = r.w;
                nt = w * r.h;
            }
        }
        // ranks within the workgroup from the LDS table (first TA_LOCAL tiles of a face; the rare further ones and a
        // full table go straight to the global cursor), one cursor atomic per distinct tile, then the scatter
        face_of[it] = f;
#pragma unroll
        for (int s = 0; s < TA_LOCAL; s++) packed[it][s] = -1;
        for (int s = 0; s < nt && s < TA_LOCAL; s++) {
            const int tile = b * bb.T + (ty0 + s / w) * bb.tiles_x + tx0 + s % w;
            int rank = 0;
            const int slot = ta_add(agg, tile, rank);
            if (slot >= 0) {
#pragma unroll
                for (int q = 0; q < TA_LOCAL; q++) if (q == s) packed[it][q] = (slot << 16) | rank;
            } else {
                const int pos = atomicAdd(&bb.tile_cursor[tile], 1);
                bb.pairs[(size_t)bb.tile_offset[tile] + pos] = f;
            }
        }
        // a coarse face's further tiles (up to kcap - TA_LOCAL): four cursor atomics in flight at a time -- one after the
        // other they are a chain of up to sixty round trips per lane (722 triangles @512^2: 41 us for three workgroups)
        for (int s0 = TA_LOCAL; s0 < nt; s0 += 4) {
            int tile[4], pos[4], off[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int s = min(s0 + j, nt - 1);
                tile[j] = b * bb.T + (ty0 + s / w) * bb.tiles_x + tx0 + s % w;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (s0 + j < nt) { pos[j] = atomicAdd(&bb.tile_cursor[tile[j]], 1); off[j] = bb.tile_offset[tile[j]]; }
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (s0 + j < nt) bb.pairs[(size_t)off[j] + pos[j]] = f;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < (1 << ta_bits()); k += blockDim.x)
        if (agg.key[k]) {
            const int tile = agg.key[k] - 1;
            agg.base[k] = bb.tile_offset[tile] + atomicAdd(&bb.tile_cursor[tile], agg.cnt[k]);
        }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < FPT; it++) {
#pragma unroll
        for (int s = 0; s < TA_LOCAL; s++)
            if (packed[it][s] >= 0) bb.pairs[(size_t)agg.base[packed[it][s] >> 16] + (packed[it][s] & 0xFFFF)] = face_of[it];
    }
}

// ---- pass 4: one wave64 per 8x8 tile ---------------------------------------------------------------
// Per chunk of 64 listed faces: lane j stages face j (NDC vertices, pixel-space inverse, its bounding
// box clipped to the tile) in LDS; the (face, pixel) candidates of the whole chunk are then flattened
// over the 64 lanes, so a chunk of sub-pixel triangles costs a few iterations and a tile-filling
// triangle costs one iteration per face.  Winners are kept in a 64-entry LDS z-buffer of
// (ordered depth bits << 32 | face index).
// order this wave's LDS writes before its later LDS reads (one wave: program order + a compiler/memory fence)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The output images of the silhouette / depth modes, no anti-aliasing (any may be NULL): alpha_map [B,S,S] = covered (internal
// layout, row 0 = bottom: what the edge gradient reads), alpha_out / depth_out [B,S,S] = the same and the depth map with
// the rows reversed (rasterize.py:311-317).
struct ModeOut {
    float* alpha_map = nullptr;
    float* alpha_out = nullptr;
    float* depth_out = nullptr;
    __device__ __forceinline__ void write(size_t i, size_t o, bool covered, float depth) const {
        const float a = covered ? 1.0f : 0.0f;
        if (alpha_map) alpha_map[i] = a;
        if (alpha_out) alpha_out[o] = a;
        if (depth_out) depth_out[o] = depth;
    }
};
struct RasterOut {
    int32_t* face_index_map;
    float* weight_map;
    float* depth_map;
    float* face_inv_map;   // NULL unless the caller wants the reference's [B,S,S,3,3] map
    unsigned char* marks;  // NULL, or [B*F] zeroed by k_bin_count: marks[b*F + f] = 1 for every face that owns a pixel
    int* marks_count = nullptr;   // with marks: the visibility list's length, cleared by k_bin_count for the compaction pass
};

// W waves per tile: W = 1 when there are enough tiles to fill the chip; W = 4 for small rasters (a few thousand
// tiles), where the hottest tiles' chunk chains would otherwise run on one wave each with nothing to overlap their
// latency: the chunks of a tile are dealt to its waves, which share the tile's LDS z-buffer and never wait for one
// another between chunks (each stages into its own arrays; only the z-buffer is common, through LDS atomics).
// STREAM (W == 1, big grids): a wave does not take one tile but a strided sequence of its XCD's tiles, and what a tile's
// first chunk needs before it can do anything -- its list's position and length, its lane's list entry, that face's nine
// floats: three DEPENDENT round trips to the L2, a third of the one-tile-per-wave pass's time with nothing of the tile's own
// to overlap them -- is requested while the tile BEFORE it is worked on: header two tiles ahead, list entry one tile ahead,
// the face behind the current tile's staging step.  (And 5632 workgroups are launched instead of 131072: 0.03 ms.)
#ifndef D3M_RT_STREAM_WAVES
#define D3M_RT_STREAM_WAVES 5      // waves per SIMD the streaming form is held to (its LDS leaves room for 5.5)
#endif
// MODE: the pass also writes the OUTPUT images of the renderer's silhouette / depth modes without anti-aliasing (ModeOut:
// what d3m_output_epilogue would make of its maps in a pass of its own) -- an instantiation of its own, so that the ordinary
// one (106 SGPRs in the streaming form) does not carry the pointers.
template <class FS, int W, bool STREAM = false, bool MODE = false>
__global__ void __launch_bounds__(64 * W, STREAM ? D3M_RT_STREAM_WAVES : 1) k_raster_tiles(FS fs, BinBuffers bb, RasterOut out, float near, float far,
                                                                                          ModeOut mo) {
    static_assert(!STREAM || W == 1, "the streaming form is the one-wave form");
    __shared__ float s_face_all[W][9][WAVE];
    __shared__ float s_finv_all[W][9][WAVE];
    __shared__ int s_fid_all[W][WAVE];
    __shared__ uint32_t s_box_all[W][WAVE];
    __shared__ uint32_t s_zkey_all[W][WAVE];
    __shared__ int s_pre_all[W][WAVE + 1];
    constexpr int RING = 2 * WAVE;             // survivors of the cheap tests waiting for the expensive ones
    __shared__ unsigned short s_ring_all[W][RING];
    constexpr int HEADS = 8 * WAVE;            // candidates per window of the owner marks (one uint2 per lane: with the
                                               // 128-entry z-buffer of a 16 x 8 tile, 20 workgroups still fit a CU's LDS)
    __shared__ __attribute__((aligned(16))) unsigned char s_head_all[W][HEADS];
    __shared__ unsigned long long s_z[TILE_PX];
    __shared__ float s_cx[TILE_W], s_cy[TILE_H];
    const int wv = W == 1 ? 0 : (int)(threadIdx.x >> 6);
    float (&s_face)[9][WAVE] = s_face_all[wv];
    float (&s_finv)[9][WAVE] = s_finv_all[wv];
    int (&s_fid)[WAVE] = s_fid_all[wv];
    uint32_t (&s_box)[WAVE] = s_box_all[wv];
    uint32_t (&s_zkey)[WAVE] = s_zkey_all[wv];
    int (&s_pre)[WAVE + 1] = s_pre_all[wv];
    unsigned short (&s_ring)[RING] = s_ring_all[wv];
    unsigned char (&s_head)[HEADS] = s_head_all[wv];

    // XCD-aware order: consecutive blocks are dealt round-robin to the 8 XCDs, so give XCD x the
    // contiguous tile range [x*per, (x+1)*per): neighbouring tiles (shared faces, shared vertex
    // lines) stay within one L2.
    const int n_tiles = bb.B * bb.T;
    const int per = (n_tiles + 7) >> 3;
    const int S = bb.S;
    const int lane = lane_id();
    // STREAM: the wave takes every (gridDim / 8)-th tile of its XCD's range -- one or two tiles (run_forward*: two thirds as
    // many waves per XCD as it has tiles, an odd number).  A wave's tiles are a SEQUENCE and tiles differ in cost by orders
    // of magnitude: as many waves as the chip holds with ~23 tiles each took 0.40-0.49 ms where one tile per workgroup takes
    // 0.31; tiles handed out from a per-XCD counter in batches of 1 to 8 0.44-0.84; two to four NEIGHBOURING tiles per wave
    // 0.38-0.48; a stride that pairs the same screen position of two views 0.37; 3 tiles per wave 0.29, 1.5 tiles 0.28.
    const int xcd = (int)(blockIdx.x & 7);
    const int tile_stride = STREAM ? (int)(gridDim.x >> 3) : 0;
    const int tile_end = STREAM ? min((xcd + 1) * per, n_tiles) : n_tiles;
    int tile = xcd * per + (int)(blockIdx.x >> 3);
    auto next_tile = [&](int from) { return from + tile_stride; };
    if (tile >= tile_end) return;
    // STREAM: (position, length) of the current tile's list, the lane's entry and its face; the same of the next tile
    int cnt_cur = 0, off_cur = 0, fid_cur = -1, tile_n = 0, cnt_n = 0, off_n = 0;
    float face_held[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};       // the current tile's until its first chunk is staged, then the next's
    if (STREAM) {
        cnt_cur = bb.tile_count[tile]; off_cur = bb.tile_offset[tile];
        if (lane < cnt_cur) { fid_cur = bb.pairs[off_cur + lane]; fs.load(tile / bb.T, fid_cur, face_held); }
        tile_n = next_tile(tile);
        if (tile_n < tile_end) { cnt_n = bb.tile_count[tile_n]; off_n = bb.tile_offset[tile_n]; }
    }
  for (;;) {
    int fid_n = -1, tile_nn = 0, cnt_nn = 0, off_nn = 0;
    bool next_requested = !STREAM;
    if (STREAM) {
        if (lane < cnt_n) fid_n = bb.pairs[off_n + lane];
        tile_nn = next_tile(tile_n);
        if (tile_nn < tile_end) { cnt_nn = bb.tile_count[tile_nn]; off_nn = bb.tile_offset[tile_nn]; }
    }
    const int b = tile / bb.T, t = tile % bb.T;
    const int px0 = (t % bb.tiles_x) * TILE_W, py0 = (t / bb.tiles_x) * TILE_H;

    if (wv == 0) {
        if (lane < TILE_W) s_cx[lane] = pixel_center(px0 + lane, S);
        else if (lane < TILE_W + TILE_H) s_cy[lane - TILE_W] = pixel_center(py0 + lane - TILE_W, S);
#pragma unroll
        for (int h = 0; h < TILE_PX / WAVE; h++) s_z[h * WAVE + lane] = ~0ull;
    }
    if (lane == 0) s_pre[0] = 0;
    if (STREAM) wave_lds_sync(); else __syncthreads();

    for (int which = 0; which < 2; which++) {
        const int* list = which == 0 ? bb.pairs + (STREAM ? off_cur : bb.tile_offset[tile]) : bb.big_list + (size_t)b * bb.F;
        const int n = which == 0 ? (STREAM ? cnt_cur : bb.tile_count[tile]) : bb.big_count[b];
        for (int base = wv * WAVE; base < n; base += W * WAVE) {
            int cnt = 0;
            if (base + lane < n) {
                const bool held = STREAM && which == 0 && base == 0;        // requested while the previous tile was worked on
                const int fid = held ? fid_cur : list[base + lane];
                float face[9];
                if (held) {
#pragma unroll
                    for (int k = 0; k < 9; k++) face[k] = face_held[k];
                } else {
                    fs.load(b, fid, face);
                }
                int x0, x1, y0, y1;
                if (pixel_bbox(face, S, x0, x1, y0, y1)) {
                    x0 = max(x0, px0); x1 = min(x1, px0 + TILE_W - 1);
                    y0 = max(y0, py0); y1 = min(y1, py0 + TILE_H - 1);
                    if (x0 <= x1 && y0 <= y1) {
                        const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
                        cnt = bw * bh;
                        float finv[9];
                        face_inverse(face, S, finv);
#pragma unroll
                        for (int k = 0; k < 9; k++) { s_face[k][lane] = face[k]; s_finv[k][lane] = finv[k]; }
                        s_fid[lane] = fid;
                        // EARLY Z.  The interpolated depth of KCU:136 is a weighted harmonic mean of the three vertex
                        // depths (weights clamped to [0,1] and renormalised), so for positive depths it cannot fall
                        // below their minimum by more than a few ulp.  A candidate whose (minimum - margin) already
                        // lies behind the pixel's current winner can neither win nor tie: skip its arithmetic.  With
                        // fill_back half of the listed faces are the far side of the mesh.
                        const float zmin = fminf(face[2], fminf(face[5], face[8]));
                        s_zkey[lane] = (zmin > 0.0f) ? ordered_bits(zmin * 0.99999f) : 0u;
                        // x | y<<4 | (width-1)<<8 | ceil(65536/width)<<12  (exact floor(c/width) for c < 4096 / width)
                        s_box[lane] = (uint32_t)(x0 - px0) | ((uint32_t)(y0 - py0) << 4) | ((uint32_t)(bw - 1) << 8) |
                                      ((uint32_t)((65536 + bw - 1) / bw) << 12);
                    }
                }
            }
            if (!next_requested) {          // the next tile's faces (its list entries were requested at the top of this tile)
                next_requested = true;
                if (fid_n >= 0) fs.load(tile_n / bb.T, fid_n, face_held);
            }
            const int incl = wave_inclusive_scan(cnt);
            s_pre[lane + 1] = incl;
            const int total = __shfl(incl, 63, 64);
            wave_lds_sync();                        // this wave's staging is complete (its own arrays: no block barrier)
            // Two phases.  A: every (face, pixel) candidate of the chunk takes the cheap tests (early z, the three
            // half-plane tests); the ~30 % that pass are appended to a small ring in LDS.  B: whenever the ring holds a
            // full wave of survivors (and at the end), all 64 lanes do the expensive part -- barycentrics and depth with
            // their seven correctly-rounded divisions -- instead of a third of them.
            int head = 0, count = 0;                   // wave-uniform
            auto resolve = [&](int n) {
                if (lane < n) {
                    const uint32_t ent = s_ring[(head + lane) & (RING - 1)];
                    const int lo = (int)(ent & 63), lx = (int)((ent >> 6) & 15), ly = (int)(ent >> 10);
                    float face[9], finv[9], w[3], zp;
#pragma unroll
                    for (int k = 0; k < 9; k++) { face[k] = s_face[k][lo]; finv[k] = s_finv[k][lo]; }
                    if (weights_depth(face, finv, px0 + lx, py0 + ly, near, far, w, zp)) {
                        const unsigned long long key = ((unsigned long long)ordered_bits(zp) << 32) | (uint32_t)s_fid[lo];
                        atomicMin(&s_z[ly * TILE_W + lx], key);
                    }
                }
            };
            // Which face a candidate belongs to: every face marks the first of its candidates with its lane number;
            // the owner of candidate c is then the running maximum of the marks up to c (six DPP steps and a carry)
            // instead of a binary search through six dependent LDS reads.  Windows of HEADS candidates.
            int carry = 0;                              // wave-uniform: mark of the last candidate so far
            for (int w0 = 0; w0 < total; w0 += HEADS) {
                reinterpret_cast<uint2*>(s_head)[lane] = make_uint2(0, 0);
                wave_lds_sync();
                const int start = incl - cnt;
                if (cnt > 0 && start >= w0 && start < w0 + HEADS) s_head[start - w0] = (unsigned char)(lane + 1);
                wave_lds_sync();
                const int wend = min(total, w0 + HEADS);
                for (int c0 = w0; c0 < wend; c0 += WAVE) {
                    const int c = c0 + lane;
                    uint32_t own = wave_max_scan(c < wend ? (uint32_t)s_head[c - w0] : 0u);
                    own = max(own, (uint32_t)carry);
                    carry = __builtin_amdgcn_readlane((int)own, 63);
                    bool pass = false;
                    uint32_t ent = 0;
                    if (c < wend) {
                        const int lo = (int)own - 1;
                        const int local = c - s_pre[lo];
                        const uint32_t box = s_box[lo];
                        const int bw = (int)((box >> 8) & 15) + 1;
                        const int row = (int)(((uint32_t)local * (box >> 12)) >> 16);
                        const int lx = (int)(box & 15) + (local - row * bw), ly = (int)((box >> 4) & 15) + row;
                        if (!(s_zkey[lo] > (uint32_t)(s_z[ly * TILE_W + lx] >> 32))) {    // early z (see above)
                            float face[9];
#pragma unroll
                            for (int k = 0; k < 9; k++) face[k] = (k % 3 == 2) ? 0.0f : s_face[k][lo];
                            pass = inside_face(face, s_cx[lx], s_cy[ly]);
                            ent = (uint32_t)lo | ((uint32_t)lx << 6) | ((uint32_t)ly << 10);
                        }
                    }
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
                    if (pass) s_ring[(head + count + mask_rank(m)) & (RING - 1)] = (unsigned short)ent;
                    count += __popcll(m);
                    wave_lds_sync();
                    if (count >= WAVE) {
                        resolve(WAVE);
                        head = (head + WAVE) & (RING - 1);
                        count -= WAVE;
                    }
                }
            }
            if (count > 0) resolve(count);
            wave_lds_sync();                        // before this wave restages
        }
    }
    if (!next_requested && fid_n >= 0) fs.load(tile_n / bb.T, fid_n, face_held);    // (a tile without a chunk)
    if (!STREAM) {
        __syncthreads();                            // every wave's bids are in
        if (wv != 0) return;
    }

    // resolve: lane = pixel; recompute the winner's weights (same arithmetic -> same bits) and store.  Pixels
    // nobody covers get the reference's initial values (rasterize.py:50-58: index -1, weights 0, depth far, inverse
    // 0), so the caller's pre-fill is not relied upon: every pixel of every map is written here.
#pragma unroll 1
    for (int part = 0; part < TILE_PX / WAVE; part++) {           // 64 pixels at a time: TILE_W wide, 64 / TILE_W rows
    const int in_tile = part * WAVE + lane;
    const unsigned long long key = s_z[in_tile];
    const int xi = px0 + in_tile % TILE_W, yi = py0 + in_tile / TILE_W;
    if (out.marks) {
        // which faces own a pixel (d3m_visibility's first step, for free here): a face's pixels form a small blob, and
        // only those of them that have neither the same face to their left nor above them within this part speak up
        const int fid = key != ~0ull ? (int)(uint32_t)(key & 0xFFFFFFFFull) : -1;
        const int left = __shfl(fid, lane - 1, 64), up = __shfl(fid, lane - TILE_W, 64);
        if (fid >= 0 && !((lane % TILE_W) && left == fid) && !(lane >= TILE_W && up == fid)) out.marks[(size_t)b * bb.F + fid] = 1;
    }
    if (xi < S && yi < S) {
        const size_t i = ((size_t)b * S + yi) * S + xi;
        if (key != ~0ull) {
            const int fid = (int)(uint32_t)(key & 0xFFFFFFFFull);
            float face[9], finv[9], w[3], zp;
            fs.load(b, fid, face);
            face_inverse(face, S, finv);
            weights_depth(face, finv, xi, yi, near, far, w, zp);
            out.depth_map[i] = zp;
            out.face_index_map[i] = fid;
            out.weight_map[3 * i + 0] = w[0];
            out.weight_map[3 * i + 1] = w[1];
            out.weight_map[3 * i + 2] = w[2];
            if constexpr (MODE) mo.write(i, ((size_t)b * S + (S - 1 - yi)) * S + xi, true, zp);
            if (!MODE && out.face_inv_map) {        // (MODE: the mode outputs instead -- the entry point refuses both)
#pragma unroll
                for (int k = 0; k < 9; k++) out.face_inv_map[9 * i + k] = finv[k];
            }
        } else {
            if constexpr (MODE) mo.write(i, ((size_t)b * S + (S - 1 - yi)) * S + xi, false, far);
            out.depth_map[i] = far;
            out.face_index_map[i] = -1;
            out.weight_map[3 * i + 0] = 0.0f;
            out.weight_map[3 * i + 1] = 0.0f;
            out.weight_map[3 * i + 2] = 0.0f;
            if (!MODE && out.face_inv_map) {
#pragma unroll
                for (int k = 0; k < 9; k++) out.face_inv_map[9 * i + k] = 0.0f;
            }
        }
    }    }
    if (!STREAM || tile_n >= tile_end) break;
    tile = tile_n; cnt_cur = cnt_n; off_cur = off_n; fid_cur = fid_n;
    tile_n = tile_nn; cnt_n = cnt_nn; off_n = off_nn;
    wave_lds_sync();                                // the z-buffer has been read: the next tile may clear it
  }
}

// ---- texture sampling (KCU:172-242), one lane per pixel ---------------------------------------------
// ts == 1 makes KCU:229-233 index texels 1..3 of a one-texel cube (the following faces' texels, with
// weights ~ -eps); that in-buffer bleed is reproduced, reads past the end of the buffer yield 0.
__global__ void __launch_bounds__(256) k_texture_sampling(const float* __restrict__ faces, const float* __restrict__ textures,
                                                         const int32_t* __restrict__ face_index_map,
                                                         const float* __restrict__ weight_map,
                                                         const float* __restrict__ depth_map, float* __restrict__ rgb_map,
                                                         int32_t* __restrict__ sampling_index_map,
                                                         float* __restrict__ sampling_weight_map, int B, int F, int S, int ts,
                                                         float eps) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * S * S) return;
    const int fi = face_index_map[i];
    if (fi < 0) return;
    const int bn = (int)(i / ((long)S * S));
    const float* face = faces + ((size_t)bn * F + fi) * 9;
    const size_t tex_base = ((size_t)bn * F + fi) * ts * ts * ts * 3;
    const size_t tex_total = (size_t)B * F * ts * ts * ts * 3;
    const float depth = depth_map[i];
    float tif[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float t = weight_map[3 * i + k] * (float)(ts - 1) * (depth / face[3 * k + 2]);
        t = fmaxf(t, 0.0f);
        t = fminf(t, (float)(ts - 1) - eps);
        tif[k] = t;
    }
    float px[3] = {0, 0, 0};
#pragma unroll
    for (int pn = 0; pn < 8; pn++) {
        float w = 1;
        int tii[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int fl = f2i(tif[k]);
            if (((pn >> k) & 1) == 0) { w *= 1 - (tif[k] - (float)fl); tii[k] = fl; }
            else                      { w *= tif[k] - (float)fl;       tii[k] = fl + 1; }
        }
        const int isc = tii[0] * ts * ts + tii[1] * ts + tii[2];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const size_t ti = tex_base + (size_t)((long)isc * 3 + k);
            px[k] += w * (ti < tex_total ? textures[ti] : 0.0f);
        }
        if (sampling_index_map) sampling_index_map[i * 8 + pn] = isc;
        if (sampling_weight_map) sampling_weight_map[i * 8 + pn] = w;
    }
    rgb_map[3 * i + 0] = px[0];
    rgb_map[3 * i + 1] = px[1];
    rgb_map[3 * i + 2] = px[2];
}

}  // namespace d3m
