// d3m_bid.h -- coverage by BIDDING: the rasterizer for meshes whose triangles are a pixel or a few.
//
// d3m_forward.h bins faces to 8x8 tiles and gives every tile a wave: three passes of set-up (count, allocate, fill)
// before the first pixel is tested, and every (face, tile) pair pays for the face's set-up again.  When the triangles
// are no bigger than a tile's pixels that is most of the work (a 1 M-triangle mesh at 1024^2: 0.62 ms of binning + 1.08 ms
// of tile pass per 8 views).  Here the faces come to the pixels instead: a wave stages PW triangles (pairs, with
// fill_back: at most one of a pair's two orientations faces the camera) in LDS -- vertices, pixel-space inverse,
// bounding box --, numbers the boxes' ROWS through, and takes the next 64 rows per step whichever face they belong to (a lane
// walks its row's span: bid_rows below); the pixels that pass the reference's tests (KCU:110-139 through d3m_device.h: same
// operations, same bits) BID ~((ordered depth bits << 32) | face) in a 64-bit z-buffer with atomicMax -- "nearest, lowest
// index among equals" (KCU:142), independent of order.  A per-pixel pass then turns the winners into the maps.
// Also where the tile pass cannot fill the chip (small batches); and the gan2shape block's two walks (d3m_g2s.h) use the
// same machinery on its implicit grid mesh.
#pragma once
#include "d3m_device.h"
#include "d3m_forward.h"

namespace d3m {

// z-buffer entries: the bid of (depth zp, face fid) is ~((ordered_bits(zp) << 32) | fid), so that atomicMax keeps the
// nearest face, the lowest index among equal depths (KCU:142), and 0 means "nothing here"
__device__ __forceinline__ unsigned long long bid_key(float zp, int fid) {
    return ~(((unsigned long long)ordered_bits(zp) << 32) | (uint32_t)fid);
}
__device__ __forceinline__ float bid_depth(unsigned long long e, float far) {
    if (e == 0ull) return far;                                  // uncovered pixels keep `far` (NR/rasterize.py:55)
    const uint32_t u = ~(uint32_t)(e >> 32);
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}
__device__ __forceinline__ int bid_face(unsigned long long e) { return e == 0ull ? -1 : (int)~(uint32_t)e; }

constexpr int BID_HEADS = 16 * WAVE;        // rows per window of owner marks (one uint4 per lane)
// what lane j < PW stages for its face; the walk's scratch.  FINV false: the pixel-space inverse is not staged (2.3 KB per
// wave at PW = 64) but set up again by the survivors' step -- LDS, not registers, bounds k_bid_faces' occupancy
template <int PW, bool FINV = true>
struct BidStage {
    float face[9][PW], finv[9][FINV ? PW : 1];
    int fid[PW], x0[PW], y0[PW], bw[PW];
    int pre[WAVE + 1];
    __attribute__((aligned(16))) unsigned char head[BID_HEADS];
    uint32_t ring[2 * WAVE];            // candidates that passed the cheap test, waiting for a full wave of them
};

// The walk.  Two phases: cheap(owner lane, x, y) -> bool on every candidate; the ones that pass wait in a ring until a full
// wave of them has gathered (and at the end), and costly(owner lane, x, y) then runs on 64 busy lanes instead of on the
// ~third of a step's candidates that survive (k_raster_tiles' scheme).  The units that are numbered through over the 64
// lanes are the ROWS of the staged boxes, not their pixels: a box of w x h pixels around a needle-shaped triangle holds few
// pixels of the triangle (the 1 M-triangle configuration: boxes of 5 x 7 = 38 candidates around 4 px^2 of area; the
// gan2shape meshes seen from 57 degrees: 14 x overdraw), and with the pixels numbered through every one of them paid the
// owner look-up (the first form of this walk: 1.46 against 0.93 ms on that configuration, 42 against 30 us in
// k_g2s_raster).  A lane takes one row of one face, intersects the row's line with the three half-planes of KCU:115-117
// -- approximately (v_rcp_f32), then widened by half a pixel on either side and clipped to the box: the exact test below
// still decides, the span only must not lose a pixel (NaN / horizontal edges drop out of the min / max and leave the box's
// bounds) -- and walks the span's pixels itself.  ~40 instructions per row once, then the cheap test alone per candidate,
// and about two candidates per row instead of the box's width.
template <int PW, bool FINV, class Cheap, class Costly>
__device__ __forceinline__ void bid_rows(BidStage<PW, FINV>& st, int rows, int S, Cheap&& cheap, Costly&& costly) {
    const int lane = lane_id();
    const int incl = wave_inclusive_scan(rows);
    if (lane == 0) st.pre[0] = 0;
    st.pre[lane + 1] = incl;
    const int total = __shfl(incl, 63, 64);
    int carry = 0, head = 0, waiting = 0;           // wave-uniform
    auto drain = [&](int n) {
        if (lane < n) {
            const uint32_t e = st.ring[(head + lane) & (2 * WAVE - 1)];
            costly((int)(e & 63u), (int)((e >> 6) & 0x1FFFu), (int)(e >> 19));
        }
    };
    const float half = 0.5f * (float)S;
    for (int w0 = 0; w0 < total; w0 += BID_HEADS) {
        reinterpret_cast<uint4*>(st.head)[lane] = make_uint4(0, 0, 0, 0);
        wave_lds_sync();
        const int start = incl - rows;
        if (rows > 0 && start >= w0 && start < w0 + BID_HEADS) st.head[start - w0] = (unsigned char)(lane + 1);
        wave_lds_sync();
        const int wend = min(total, w0 + BID_HEADS);
        for (int c0 = w0; c0 < wend; c0 += WAVE) {
            const int c = c0 + lane;
            uint32_t own = wave_max_scan(c < wend ? (uint32_t)st.head[c - w0] : 0u);
            own = max(own, (uint32_t)carry);
            carry = __builtin_amdgcn_readlane((int)own, 63);
            int lo = 0, xi = 0, xb = -1, yi = 0;
            if (c < wend) {
                lo = (int)own - 1;
                yi = st.y0[lo] + (c - st.pre[lo]);
                const float yp = pixel_center(yi, S);
                float x_lo = -3.0e38f, x_hi = 3.0e38f;
#pragma unroll
                for (int e = 0; e < 3; e++) {
                    const int n = (e + 1) % 3;
                    const float xa = st.face[3 * e][lo], ya = st.face[3 * e + 1][lo];
                    const float dy = st.face[3 * n + 1][lo] - ya;
                    const float x = xa + ((yp - ya) * (st.face[3 * n][lo] - xa)) * __builtin_amdgcn_rcpf(dy);
                    if (dy > 0.0f) x_hi = fminf(x_hi, x);            // (yp - ya)(xb - xa) >= (xp - xa) dy  <=>  xp <= x
                    else if (dy < 0.0f) x_lo = fmaxf(x_lo, x);       //                                     <=>  xp >= x
                }
                // NDC -> pixel index (KCU:47), half a pixel of slack, the box's bounds
                const float p_lo = fminf(fmaxf(x_lo * half + (half - 0.5f) - 0.5f, -1.0f), 65536.0f);
                const float p_hi = fminf(fmaxf(x_hi * half + (half - 0.5f) + 0.5f, -1.0f), 65536.0f);
                xi = max(st.x0[lo], (int)ceilf(p_lo));
                xb = min(st.x0[lo] + st.bw[lo] - 1, (int)floorf(p_hi));
            }
            const int steps = __builtin_amdgcn_readlane((int)wave_max_scan((uint32_t)max(xb - xi + 1, 0)), 63);
            for (int k = 0; k < steps; k++) {
                const bool pass = xi <= xb && cheap(lo, xi, yi);
                const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
                if (pass) st.ring[(head + waiting + mask_rank(m)) & (2 * WAVE - 1)] = (uint32_t)lo | ((uint32_t)xi << 6) | ((uint32_t)yi << 19);
                waiting += __popcll(m);
                wave_lds_sync();
                if (waiting >= WAVE) {
                    drain(WAVE);
                    head = (head + WAVE) & (2 * WAVE - 1);
                    waiting -= WAVE;
                }
                xi++;
            }
        }
        wave_lds_sync();                            // before the marks are cleared again
    }
    if (waiting > 0) drain(waiting);
}


// ---- the generic passes: any indexed mesh -----------------------------------------------------------------------------
// k_bid_faces: lane j < PW of a wave stages face (pair) j of the wave's PW -- read through the indices, oriented, its
// dense copy left in faces_dense_out for the later passes (as k_bin_count does) -- then the wave walks the candidates.
// PAIRED (fill_back): lane j handles index triple j in both orientations; at most one faces the camera.
// A face whose box holds more than BID_BIG_AREA pixels is not walked here -- PW such faces would keep one wave busy for
// milliseconds while the chip idles -- but listed (big_list: face index within the batch; big_count zeroed by the caller) for
// k_bid_big, which gives every 64 rows of such a box a wave of their own.
constexpr int BID_BIG_AREA = 1024;
template <class FS, int PW, bool PAIRED>
__global__ void __launch_bounds__(256) k_bid_faces(FS fs, unsigned long long* __restrict__ zbuf,
                                                   float* __restrict__ faces_dense_out, int B, int S, float near, float far,
                                                   unsigned char* __restrict__ marks, int* __restrict__ marks_count,
                                                   int* __restrict__ big_list, int* __restrict__ big_count,
                                                   float* __restrict__ faces_inv = nullptr) {
#ifdef D3M_BID_STAGE_FINV
    constexpr bool FINV = true;
#else
    constexpr bool FINV = false;
#endif
    __shared__ BidStage<PW, FINV> s_stage[4];
    if (marks_count && blockIdx.x == 0 && threadIdx.x == 0) *marks_count = 0;     // (as k_bin_count: see RasterOut)
    __shared__ int s_view[4][PW];
    BidStage<PW, FINV>& st = s_stage[threadIdx.x >> 6];
    int (&view)[PW] = s_view[threadIdx.x >> 6];
    const int lane = lane_id();
    const int F = fs.num_faces(), Fl = PAIRED ? F / 2 : F;
    const long unit = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * PW + lane;
    int cnt = 0;
    if (lane < PW && unit < (long)B * Fl) {
        const int b = (int)(unit / Fl), f0 = (int)(unit - (long)b * Fl);
        if (marks) {                                  // "owns a pixel": set by the resolve pass
            marks[(size_t)b * F + f0] = 0;
            if (PAIRED) marks[(size_t)b * F + f0 + Fl] = 0;
        }
        float v[9], face[9], finv[9];
        fs.load(b, f0, v);
        bool rev = false, front = !backside(v);
        if (!front && PAIRED) {
#pragma unroll
            for (int k = 0; k < 9; k++) face[k] = v[(2 - k / 3) * 3 + k % 3];
            rev = true;
            front = !backside(face);
        }
        if (!rev) {
#pragma unroll
            for (int k = 0; k < 9; k++) face[k] = v[k];
        }
        if (front) {
            const int fid = rev ? f0 + Fl : f0;
            if (faces_inv) {                      // the reference's K1 scratch (KCU:24-67), for the callers that pass it
                face_inverse(face, S, finv);
#pragma unroll
                for (int k = 0; k < 9; k++) faces_inv[((size_t)b * F + fid) * 9 + k] = finv[k];
            }
            int x0, x1, y0, y1;
            const bool boxed = pixel_bbox(face, S, x0, x1, y0, y1);
            // the dense copy the later passes read: only of faces whose box holds a pixel centre at all -- a face without
            // one owns no pixel and is never read again (of a 1 M-triangle mesh at 1024^2 that is most of them)
            if (boxed && faces_dense_out) {
                float* o = faces_dense_out + ((size_t)b * F + fid) * 9;
#pragma unroll
                for (int k = 0; k < 9; k++) o[k] = face[k];
            }
            if (boxed && (long)(x1 - x0 + 1) * (y1 - y0 + 1) > BID_BIG_AREA) {
                big_list[atomicAdd(big_count, 1)] = b * F + fid;
            } else if (boxed) {
                face_inverse(face, S, finv);
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    st.face[k][lane] = face[k];
                    if (FINV) st.finv[k][lane] = finv[k];
                }
                const int bw = x1 - x0 + 1;
                st.fid[lane] = fid; st.x0[lane] = x0; st.y0[lane] = y0; st.bw[lane] = bw;
                view[lane] = b;
                cnt = y1 - y0 + 1;                    // rows of the box (bid_rows)
            }
        }
    }
    bid_rows<PW>(st, cnt, S,
        [&](int lo, int xi, int yi) {              // cheap: the three half-plane tests
            // (Up to round 4 an early z followed: the pixel's current bid read from the z-buffer, candidates whose nearest
            //  vertex lies behind it dropped.  That read is a round trip to the L2 in the middle of every step of a wave
            //  that has nothing else to do: without it 0.0745 -> 0.0635 ms at 4 views of the 100 k mesh, 0.86 -> 0.77 on
            //  the 1 M-triangle mesh, the same maps -- the bids it saved are cheaper than the wait.)
            float face[9];
#pragma unroll
            for (int k = 0; k < 9; k++) face[k] = (k % 3 == 2) ? 0.0f : st.face[k][lo];
            return inside_face(face, pixel_center(xi, S), pixel_center(yi, S));
        },
        [&](int lo, int xi, int yi) {              // costly: barycentrics and depth (seven divisions), the bid
            float face[9], finv[9], w[3], zp;
#pragma unroll
            for (int k = 0; k < 9; k++) face[k] = st.face[k][lo];
            if (FINV) {
#pragma unroll
                for (int k = 0; k < 9; k++) finv[k] = st.finv[k][lo];
            } else {
                face_inverse(face, S, finv);          // (the same operations as at staging: the same bits)
            }
            if (!weights_depth(face, finv, xi, yi, near, far, w, zp)) return;
            const unsigned long long e = bid_key(zp, st.fid[lo]);
            unsigned long long* slot = zbuf + ((size_t)view[lo] * S + yi) * S + xi;
            if (e > *slot) atomicMax(slot, e);        // (the read stays: every bid as an atomic 0.77 -> 0.96 ms on the 1 M mesh)
        });
}

// k_bid_big: the listed big faces.  Workgroup (i, j) takes faces i, i + gridDim.x, ... and of each the rows
// y0 + 256 j .. y0 + 256 j + 255 of its box (and every 256 gridDim.y-th block of rows after them): a lane per row, the
// row's span as in bid_rows, every pixel of it tested and bid for by the lane itself (inside such a face most pass).
__global__ void __launch_bounds__(256) k_bid_big(DenseFaces fs, unsigned long long* __restrict__ zbuf, const int* __restrict__ big_list,
                                                 const int* __restrict__ big_count, int S, float near, float far) {
    const int n = *big_count, F = fs.num_faces();
    const float half = 0.5f * (float)S;
    for (int item = blockIdx.x; item < n; item += gridDim.x) {
        const int gi = big_list[item], b = gi / F, fid = gi - b * F;
        float face[9], finv[9];
        fs.load(b, fid, face);
        int x0, x1, y0, y1;
        if (!pixel_bbox(face, S, x0, x1, y0, y1)) continue;
        face_inverse(face, S, finv);
        const float zmin = fminf(face[2], fminf(face[5], face[8]));
        const uint32_t zkey = zmin > 0.0f ? (uint32_t)(~ordered_bits(zmin * 0.99999f)) : 0xFFFFFFFFu;
        for (int yi = y0 + (int)blockIdx.y * 256 + (int)threadIdx.x; yi <= y1; yi += 256 * (int)gridDim.y) {
            const float yp = pixel_center(yi, S);
            float x_lo = -3.0e38f, x_hi = 3.0e38f;
#pragma unroll
            for (int e = 0; e < 3; e++) {
                const int m = (e + 1) % 3;
                const float xa = face[3 * e], ya = face[3 * e + 1], dy = face[3 * m + 1] - ya;
                const float x = xa + ((yp - ya) * (face[3 * m] - xa)) * __builtin_amdgcn_rcpf(dy);
                if (dy > 0.0f) x_hi = fminf(x_hi, x);
                else if (dy < 0.0f) x_lo = fmaxf(x_lo, x);
            }
            const float p_lo = fminf(fmaxf(x_lo * half + (half - 0.5f) - 0.5f, -1.0f), 65536.0f);
            const float p_hi = fminf(fmaxf(x_hi * half + (half - 0.5f) + 0.5f, -1.0f), 65536.0f);
            const int xb = min(x1, (int)floorf(p_hi));
            unsigned long long* row = zbuf + ((size_t)b * S + yi) * S;
            float flat[9];
#pragma unroll
            for (int k = 0; k < 9; k++) flat[k] = (k % 3 == 2) ? 0.0f : face[k];
            for (int xi = max(x0, (int)ceilf(p_lo)); xi <= xb; xi++) {
                if (!inside_face(flat, pixel_center(xi, S), yp)) continue;
                if (zkey < (uint32_t)(row[xi] >> 32)) continue;                 // early z (k_raster_tiles: cannot win)
                float w[3], zp;
                if (!weights_depth(face, finv, xi, yi, near, far, w, zp)) continue;
                const unsigned long long e = bid_key(zp, fid);
                if (e > row[xi]) atomicMax(&row[xi], e);
            }
        }
    }
}

// k_bid_resolve: one lane per pixel: the winner's weights and depth recomputed (same arithmetic -> same bits), every
// pixel of every map written (uncovered: the reference's initial values), the faces that own a pixel marked.
__global__ void __launch_bounds__(256) k_bid_resolve(DenseFaces fs, const unsigned long long* __restrict__ zbuf, RasterOut out,
                                                    int B, int S, float near, float far, ModeOut mo) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const bool on = i < (long)B * S * S;
    const unsigned long long e = on ? zbuf[i] : 0ull;
    const int fid = bid_face(e);
    if (out.marks) {            // a run of pixels of one face along a row speaks up once
        const int left = __shfl_up(fid, 1, 64);
        if (fid >= 0 && !(lane_id() > 0 && left == fid)) out.marks[(size_t)(i / ((long)S * S)) * fs.num_faces() + fid] = 1;
    }
    if (!on) return;
    if (fid >= 0) {
        const int b = (int)(i / ((long)S * S)), pix = (int)(i - (long)b * S * S);
        float face[9], finv[9], w[3], zp;
        fs.load(b, fid, face);
        face_inverse(face, S, finv);
        weights_depth(face, finv, pix % S, pix / S, near, far, w, zp);
        out.depth_map[i] = zp;
        out.face_index_map[i] = fid;
        out.weight_map[3 * i + 0] = w[0]; out.weight_map[3 * i + 1] = w[1]; out.weight_map[3 * i + 2] = w[2];
        mo.write((size_t)i, ((size_t)b * S + (S - 1 - pix / S)) * S + pix % S, true, zp);
        if (out.face_inv_map) {
#pragma unroll
            for (int k = 0; k < 9; k++) out.face_inv_map[9 * i + k] = finv[k];
        }
    } else {
        const long pix = i % ((long)S * S);
        mo.write((size_t)i, (size_t)(i - pix) + (size_t)(S - 1 - pix / S) * S + pix % S, false, far);
        out.depth_map[i] = far;
        out.face_index_map[i] = -1;
        out.weight_map[3 * i + 0] = 0.0f; out.weight_map[3 * i + 1] = 0.0f; out.weight_map[3 * i + 2] = 0.0f;
        if (out.face_inv_map) {
#pragma unroll
            for (int k = 0; k < 9; k++) out.face_inv_map[9 * i + k] = 0.0f;
        }
    }
}

}  // namespace d3m
