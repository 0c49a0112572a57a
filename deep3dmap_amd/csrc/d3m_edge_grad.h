// d3m_edge_grad.h -- the edge / silhouette gradient (reference: backward_pixel_map_cuda_kernel,
// KCU:245-503) restructured for gfx950.
//
// The reference runs one thread per face; for every integer crossing d0 of every edge, on both axes,
// that thread walks the image from the edge out to the image BORDER (KCU:354-414) and inward to the
// opposite edge (KCU:417-495): 1.35 M walks of ~250 pixels for 8 views of the 100k-triangle mesh at 512^2.
//
// LINE-MAJOR formulation.  A walk only ever moves along ONE image line (a row for axis 1, a column for
// axis 0), and a line is shared by hundreds of walks.  So:
//   0. visibility     faces that own no pixel cannot contribute: flags + compacted list (shared with the gathered
//                     texture / depth pass through d3m_visibility);
//   1. k_pack_maps    what a walk reads per pixel -- (grad_alpha, grad_rgb), sum value*grad, owner -- packed in both
//                     orientations (rows, columns), plus each line's non-zero-gradient extent;
//   2. k_edge_count   a workgroup publishes the crossing ranges of its 42 faces x 6 (edge, axis) lanes in LDS and
//                     its threads take ONE crossing each per round; bounds the records each line will receive
//                     (two per crossing) without touching the maps;
//   3. k_scan_small / k_alloc_ranges   crossing base per workgroup, record slice per line (no same-address atomics);
//   4. k_edge_emit    same flattening: every crossing owns two result slots; segments are clipped to the line's
//                     extent, short ones walked in-thread, long ones written as 48-byte records in LINE order;
//   5. k_edge_lines   one workgroup per (view, axis, line): the line's records staged in LDS once, the line's
//                     segments ordered by length, sixteen segments per wave (four lanes each), factored distance;
//   6. k_edge_gather  six lanes per visible face add their crossings' slots in order; stored to grad_faces or
//                     accumulated into the vertex gradient (VertexTarget).
// Deterministic up to the final vertex atomics.  The reference OVERWRITES the 9 entries of every front-facing face
// (KCU:501-502) and leaves culled ones alone (KCU:270); with the caller's zero-initialised grad_faces
// (rasterize.py:111, a precondition of the C ABI) writing only the faces that own a pixel is the same thing: every
// other front-facing face would get zeros.  Per visited pixel the expressions are those of KCU:385-412 / :473-493
// regrouped (see "FACTORED DISTANCE" and k_pack_maps); the divisions inside the walk use v_rcp_f32 (1 ulp), far
// inside the 1e-3 gradient tolerance.
#pragma once
#include <type_traits>
#include "d3m_backward.h"
#include "d3m_face_major.h"
#include "d3m_launch.h"

namespace d3m {

constexpr int EG_INLINE_MAX = 6;   // segments of at most this many pixels are walked by the owning lane
#ifndef D3M_EG_LINE_PARTS
#define D3M_EG_LINE_PARTS 1
#endif
#ifndef D3M_EG_LINE_WAVES
#define D3M_EG_LINE_WAVES 8
#endif
constexpr int EG_LINE_PARTS = D3M_EG_LINE_PARTS;   // workgroups per line (items are dealt round-robin to the parts)
constexpr int EG_LINE_WAVES = D3M_EG_LINE_WAVES;   // waves per workgroup: parts x waves walk one line's items concurrently
constexpr int EG_ITEM_DW = 12;     // dwords per item

// What a walk reads per pixel, as one scan axis sees it: element (line d0, position d1) lives at b*S*S + d0*S + d1.
// k_pack_maps builds both orientations in one pass over the maps:
//   grad[i] = (grad_alpha, grad_r, grad_g, grad_b)              (0 for a disabled output)
//   dot[i]  = (sum value*grad of the pixel itself, owner face index bits)
// so that  diff_grad = dot.x - <reference values, grad>  (KCU:385-396 / :473-479 regrouped: 4 fma).
struct AxisMaps {
    const float4* grad;
    const float2* dot;
    __device__ __forceinline__ int owner(size_t i) const { return __float_as_int(dot[i].y); }
};

struct EdgeGradArgs {
    AxisMaps ax[2];   // [0]: axis 0 = column walks (transposed records); [1]: axis 1 = row walks
    const float* alpha_map;   // original [B,S,S] / [B,S,S,3] maps: reference values of a segment (one pixel each)
    const float* rgb_map;
    // per line (b*2 + axis)*S + d0: extent of the pixels whose gradients are not all zero, built by k_pack_maps with
    // atomicMax on zeroed arrays: nz_lo_inv = S - (first such d1), nz_hi1 = (last such d1) + 1; 0 = none.
    const int* nz_lo_inv;
    const int* nz_hi1;
    int S, use_rgb, use_alpha;
    float eps;
    unsigned n_lines;   // B*2*S
};

struct EdgeWork {
    int* visible;        // [B*F]   1 if the face owns a pixel (zeroed per call, set by k_mark_visible)
    int* visible_list;   // [B*F]   compacted indices of those faces
    int* n_visible;      // [1]     (zeroed per call)
    int2* lane_cross;    // [6*B*F] per (visible face, edge, axis) lane: first crossing within its workgroup, count
    float2* lane_partial;// [6*B*F] overflow sums of that lane (segments whose slot did not fit the workspace)
    int* line_count;     // [B*2*S] upper bound (2 per crossing) of the records each line receives = its slice (zeroed per call)
    int* line_cursor;    // [B*2*S] records written so far under each line (zeroed per call)
    int4* line_info;     // [B*2*S] what a crossing needs to know about its line, in one load: (S - first pixel of the
                         //         non-zero-gradient extent, last + 1, first record of the slice, slice length)
    int* alloc;          // [2] crossings (written by the block scan), line-slice cursor (zeroed per call)
    int* vis_block;      // [ceil(B*F/1024)+1] visible faces per 1024-face chunk, then (in place) their exclusive scan
    int* lane_block;     // [ceil(B*F/42)+1]   crossings per k_edge_count workgroup iteration, then their scan
    uint32_t* items;     // [cap * EG_ITEM_DW] records of the queued segments, grouped by line
    float2* results;     // [cap]
    int cap;
};

struct SegRef {
    float alpha, r, g, b;
};

// One walk segment.
struct Segment {
    int axis, d0, from, to, inward, f0, f1;
    float d1_cross, q0, q1;
    int ref_pos;     // d1 of the pixel whose value is the reference (in-pixel for outward, out-pixel for inward)
    int dir, d1_in;  // walk direction of KCU:297-308 and the in-pixel next to the crossing
    int oriented;    // every pixel lies on the expected side of the crossing (always true for outward walks)
};

// The d0 range of ONE (edge, axis) pair of a face (KCU:312-313): its crossings are d0_from .. d0_to.
__device__ __forceinline__ void crossing_range(float p00, float p10, int is, int& d0_from, int& d0_to) {
    d0_from = f2i(fmaxf(ceilf(fminf(p00, p10)), 0.0f));
    d0_to = f2i(fminf(fmaxf(p00, p10), (float)(is - 1)));
}

// The (at most two) walk segments of ONE crossing d0 of an (edge, axis) pair: KCU:314-362 (outward) and :417-431
// (inward).  p00..p21 = p[num][dim] of KCU:289-294 for that pair; owner(d0, d1) returns face_index_map at that
// line position.  has_out / has_in tell which of `out` / `in` were filled.
template <class Owner>
__device__ __forceinline__ void crossing_segments(float p00, float p01, float p10, float p11, float p20, float p21,
                                                  int axis, int fn, int is, int d0, int nz_lo, int nz_hi, Owner&& owner,
                                                  Segment& out, bool& has_out, Segment& in, bool& has_in) {
    // [nz_lo, nz_hi]: the line's pixels with a non-zero gradient.  Outside it diff_grad is exactly 0 and KCU:401/:481
    // skip the pixel, so every segment is clipped to it (with a masked loss the gradients vanish outside the object,
    // and the outward walks, which run to the image BORDER, lose most of their length or disappear).
    has_out = has_in = false;
    if (nz_hi < nz_lo) return;
    const int direction = (axis == 0) ? ((p00 < p10) ? -1 : 1) : ((p00 < p10) ? 1 : -1);   // KCU:297-308
    const float fd0 = (float)d0;
    const float d1_cross = (p11 - p01) / (p10 - p00) * (fd0 - p00) + p01;                 // KCU:317
    const int d1_in = (0 < direction) ? f2i(floorf(d1_cross)) : f2i(ceilf(d1_cross));
    const int d1_out = (int)((unsigned)d1_in + (unsigned)direction);
    if (d1_in < 0 || is <= d1_in || d1_out < 0 || is <= d1_out) return;                   // KCU:325-328
    Segment sg;
    sg.axis = axis;
    sg.d0 = d0;
    sg.dir = direction;
    sg.d1_in = d1_in;
    sg.d1_cross = d1_cross;
    sg.f0 = p10 != fd0;
    sg.f1 = p00 != fd0;
    sg.q0 = (p10 - p00) / (p10 - fd0);      // KCU:404 / :409: first factor of `dist`
    sg.q1 = (p10 - p00) / (fd0 - p00);
    // outward: out-pixel .. image border, only if the in-pixel belongs to this face (KCU:354-362)
    if (owner(d0, d1_in, direction) == fn) {
        const int d1_limit = (0 < direction) ? is - 1 : 0;
        out = sg;
        out.from = max(max(min(d1_out, d1_limit), 0), nz_lo);
        out.to = min(min(max(d1_out, d1_limit), is - 1), nz_hi);
        out.inward = 0;
        out.oriented = 1;
        out.ref_pos = d1_in;
        has_out = out.from <= out.to;
    }
    // inward: in-pixel .. opposite edge (KCU:417-431)
    float d0_cross2;
    if ((fd0 - p00) * (fd0 - p20) < 0) d0_cross2 = (p21 - p01) / (p20 - p00) * (fd0 - p00) + p01;
    else                               d0_cross2 = (p11 - p21) / (p10 - p20) * (fd0 - p20) + p21;
    const int d1_limit = (0 < direction) ? f2i(ceilf(d0_cross2)) : f2i(floorf(d0_cross2));
    in = sg;
    in.from = max(min(d1_in, d1_limit), 0);
    in.to = min(max(d1_in, d1_limit), is - 1);
    // all pixels on the expected side of the crossing (see "FACTORED DISTANCE"): decided before clipping
    in.oriented = (0 < direction) ? in.to == d1_in : in.from == d1_in;
    in.from = max(in.from, nz_lo);
    in.to = min(in.to, nz_hi);
    in.inward = 1;
    in.ref_pos = d1_out;
    has_in = in.from <= in.to;
}

// A segment is handed to the line kernel when it is long and its pixels lie on the expected side of the crossing
// (see "FACTORED DISTANCE" below); k_edge_count and k_edge_emit must agree on this.
__device__ __forceinline__ bool segment_queueable(const Segment& sg) {
    return sg.to - sg.from + 1 > EG_INLINE_MAX && sg.oriented;
}

// Accumulate one visited pixel: KCU:385-412 (outward) / :470-493 (inward).  Branch-free: a pixel whose
// diff_grad is <= 0 (KCU:401/:481) and a term the reference does not evaluate (f0 / f1 false: KCU:403/:408, the edge
// end lies on this very line) are dropped by SELECTS, never by a multiplication with 0: their quotient can be inf or
// NaN (an edge parallel to the walk on an integer coordinate makes d1_cross itself NaN).  NaN diff_grad still
// propagates, as in the reference.
__device__ __forceinline__ void visit_pixel(float diff, int d1, float d1_cross, float q0, float q1, bool f0, bool f1,
                                            float two_over_is, float eps, float& g0, float& g1) {
    const float t = (float)d1 - d1_cross;
    float dist0 = q0 * t * two_over_is;
    dist0 = (0 < dist0) ? dist0 + eps : dist0 - eps;
    float dist1 = q1 * t * two_over_is;
    dist1 = (0 < dist1) ? dist1 + eps : dist1 - eps;
    const float c0 = diff * __builtin_amdgcn_rcpf(dist0);
    const float c1 = diff * __builtin_amdgcn_rcpf(dist1);
    const bool skip = diff <= 0;
    g0 -= (skip || !f0) ? 0.0f : c0;
    g1 -= (skip || !f1) ? 0.0f : c1;
}

// reference values of a segment: pixel (line d0, position d1) of `axis` in the ORIGINAL maps
__device__ __forceinline__ SegRef load_ref(const EdgeGradArgs& a, int axis, size_t view_base, int d0, int d1) {
    const size_t idx = view_base + (axis ? (size_t)d0 * a.S + d1 : (size_t)d1 * a.S + d0);
    SegRef r = {0, 0, 0, 0};
    if (a.use_alpha) r.alpha = a.alpha_map[idx];
    if (a.use_rgb) { r.r = a.rgb_map[3 * idx + 0]; r.g = a.rgb_map[3 * idx + 1]; r.b = a.rgb_map[3 * idx + 2]; }
    return r;
}

// one pixel of an in-thread walk (record dt = (T, owner), g = gradients); `on` false = not part of the segment
__device__ __forceinline__ void inline_pixel(const Segment& sg, const SegRef& ref, int fn, const float2 dt, const float4 g,
                                             int d1, bool on, float two_over_is, float eps, float& g0, float& g1) {
    float diff = dt.x;
    diff = __builtin_fmaf(-ref.alpha, g.x, diff);
    diff = __builtin_fmaf(-ref.r, g.y, diff);
    diff = __builtin_fmaf(-ref.g, g.z, diff);
    diff = __builtin_fmaf(-ref.b, g.w, diff);
    // inward walks only count the face's own pixels (KCU:470); dropped by a select, like diff <= 0
    if (!on || (sg.inward && __float_as_int(dt.y) != fn)) diff = 0.0f;
    visit_pixel(diff, d1, sg.d1_cross, sg.f0 ? sg.q0 : 1.0f, sg.f1 ? sg.q1 : 1.0f, sg.f0 != 0, sg.f1 != 0, two_over_is, eps,
                g0, g1);
}

// short segment, walked straight from global memory by the owning thread: 8 + 16 bytes per pixel.  Two pixels per
// round, both records of both pixels requested before any is used, no branch inside: the walk is a chain of memory
// round trips and nothing else (the lazy, one-record-at-a-time form cost 0.21 ms of the 0.59 ms emit pass).
__device__ __forceinline__ void walk_inline(const EdgeGradArgs& a, const AxisMaps& m, size_t line_base, const Segment& sg,
                                            int from, int to, const SegRef& ref, int fn, float two_over_is, float& g0,
                                            float& g1) {
    for (int d1 = from; d1 <= to; d1 += 2) {
        const bool two = d1 + 1 <= to;
        const size_t ia = line_base + d1, ib = two ? ia + 1 : ia;
        const float2 dta = m.dot[ia], dtb = m.dot[ib];
        const float4 ga = m.grad[ia], gb = m.grad[ib];
        inline_pixel(sg, ref, fn, dta, ga, d1, true, two_over_is, a.eps, g0, g1);
        inline_pixel(sg, ref, fn, dtb, gb, d1 + 1, two, two_over_is, a.eps, g0, g1);
    }
}

// ---- 0. compact the faces that own at least one pixel -------------------------------------------------
// A face that owns no pixel cannot contribute: the outward walk needs its own in-pixel (KCU:354) and the
// inward walk only counts its own pixels (KCU:470).
//
// Slot allocation here and in k_edge_count goes through per-workgroup totals and ONE small scan kernel instead of
// an atomic cursor: thousands of returning atomics on one address serialise at the L2 atomic unit (~15 ns each:
// the 6272 of the old per-256-faces cursor cost 43 us, the 9200 per-wave ones of k_edge_count 130 us).
constexpr int EG_COMPACT_CHUNK = 1024;      // faces per workgroup (4 per lane)

__global__ void __launch_bounds__(256) k_count_visible(const int* __restrict__ visible, int* __restrict__ vis_block, long n) {
    __shared__ int s_wave[4];
    const long i0 = (long)blockIdx.x * EG_COMPACT_CHUNK + threadIdx.x * 4;
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) c += (i0 + k < n && visible[i0 + k] != 0) ? 1 : 0;
    const int incl = wave_inclusive_scan(c);
    if (lane_id() == 63) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    if (threadIdx.x == 0) vis_block[blockIdx.x] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}

// In-place exclusive scan of counts[0 .. n) by ONE workgroup; counts[n] and *total receive the sum.
// n comes from *n_ptr (divided by n_div, rounded up) when n_ptr is given: device-side sizes.
__global__ void __launch_bounds__(1024) k_scan_small(int* __restrict__ counts, int n_host, const int* __restrict__ n_ptr,
                                                    int n_div, int* __restrict__ total) {
    __shared__ int s_wave[16];
    __shared__ int s_run;
    const int n = n_ptr ? (*n_ptr + n_div - 1) / n_div : n_host;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    constexpr int PER = 8;                                  // consecutive entries per lane: 8192 per round
    for (int base = 0; base < n; base += 1024 * PER) {
        const int i0 = base + threadIdx.x * PER;
        int c[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) { c[k] = i0 + k < n ? counts[i0 + k] : 0; sum += c[k]; }
        const int incl = wave_inclusive_scan(sum);
        if (lane == 63) s_wave[wv] = incl;
        __syncthreads();
        int before = s_run;
        for (int k = 0; k < wv; k++) before += s_wave[k];
        int run = before + incl - sum;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            if (i0 + k < n) counts[i0 + k] = run;
            run += c[k];
        }
        __syncthreads();
        if (threadIdx.x == 1023) s_run = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) { counts[n] = s_run; *total = s_run; }
}

__global__ void __launch_bounds__(256) k_compact_visible(const int* __restrict__ visible, int* __restrict__ list,
                                                        const int* __restrict__ vis_block, long n) {
    __shared__ int s_wave[4];
    const long i0 = (long)blockIdx.x * EG_COMPACT_CHUNK + threadIdx.x * 4;
    bool v[4];
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = i0 + k < n && visible[i0 + k] != 0; c += v[k] ? 1 : 0; }
    const int incl = wave_inclusive_scan(c);
    const int wv = threadIdx.x >> 6;
    if (lane_id() == 63) s_wave[wv] = incl;
    __syncthreads();
    int pos = vis_block[blockIdx.x] + incl - c;
    for (int k = 0; k < wv; k++) pos += s_wave[k];
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (v[k]) list[pos++] = (int)(i0 + k);             // ascending face order: neighbours stay neighbours
}

// Six lanes per visible face, one per (edge, axis) pair; 42 faces per 256-thread workgroup.
constexpr int EG_FACES_PER_BLOCK = 42;

template <class FS>
__device__ __forceinline__ bool load_face_lane(const FS& fs, const EdgeWork& w, int blk, int is, int& pos, int& ea, long& gi,
                                               int& bn, int& fn, float& p00, float& p01, float& p10, float& p11,
                                               float& p20, float& p21) {
    const int t = threadIdx.x;
    if (t >= EG_FACES_PER_BLOCK * 6) return false;
    pos = blk * EG_FACES_PER_BLOCK + t / 6;
    ea = t % 6;
    if (pos >= *w.n_visible) return false;
    gi = w.visible_list[pos];
    const int F = fs.num_faces();
    bn = (int)(gi / F);
    fn = (int)(gi % F);
    float face[9];
    fs.load(bn, fn, face);
    if (backside(face)) return false;       // cannot own a pixel; kept for safety against inconsistent inputs
    float px[3], py[3];
#pragma unroll
    for (int n = 0; n < 3; n++) { px[n] = to_pixel(face[3 * n + 0], is); py[n] = to_pixel(face[3 * n + 1], is); }   // KCU:282
    const int edge = ea >> 1, axis = ea & 1;
    // vertex order (edge, edge+1, edge+2) mod 3 (KCU:278-279); dim 0 = x for axis 0, y for axis 1 (KCU:289-294)
    const float ax0 = edge == 0 ? px[0] : edge == 1 ? px[1] : px[2], ay0 = edge == 0 ? py[0] : edge == 1 ? py[1] : py[2];
    const float ax1 = edge == 0 ? px[1] : edge == 1 ? px[2] : px[0], ay1 = edge == 0 ? py[1] : edge == 1 ? py[2] : py[0];
    const float ax2 = edge == 0 ? px[2] : edge == 1 ? px[0] : px[1], ay2 = edge == 0 ? py[2] : edge == 1 ? py[0] : py[1];
    p00 = axis ? ay0 : ax0; p01 = axis ? ax0 : ay0;
    p10 = axis ? ay1 : ax1; p11 = axis ? ax1 : ay1;
    p20 = axis ? ay2 : ax2; p21 = axis ? ax2 : ay2;
    return true;
}

// ---- 1./3. crossings, flattened -----------------------------------------------------------------------
// A workgroup takes 42 visible faces = 252 (face, edge, axis) lanes.  The lanes publish their crossing ranges in
// LDS, a workgroup prefix turns them into one flat list of crossings (~900 per workgroup on the headline mesh), and
// the 256 threads then take ONE crossing each per round: no lane loops over its own d0 range, so a face with 30
// crossings no longer holds 63 lanes with 2 crossings hostage (per-wave max trip count was 3.3x the mean).
// Every crossing owns two result slots (2c: outward, 2c+1: inward) whether or not they end up queued; the slot is
// the item index.
struct LaneTable {
    float p[6][256];
    int fn[256], bn_axis[256], d0_from[256];
    int pre[257];           // exclusive prefix of the lanes' crossing counts; pre[256] = total
    int wave_tot[4];
};

template <class FS>
__device__ __forceinline__ int publish_lanes(const FS& fs, const EdgeWork& w, int blk, int is, LaneTable& t, bool& on, int& pos,
                                             int& ea, int& n_cross) {
    int bn = 0, fn = 0;
    long gi = 0;
    float p00 = 0, p01 = 0, p10 = 0, p11 = 0, p20 = 0, p21 = 0;
    on = load_face_lane(fs, w, blk, is, pos, ea, gi, bn, fn, p00, p01, p10, p11, p20, p21);
    int d0_from = 0, d0_to = -1;
    if (on) crossing_range(p00, p10, is, d0_from, d0_to);
    n_cross = on ? max(d0_to - d0_from + 1, 0) : 0;
    const int l = threadIdx.x;
    t.p[0][l] = p00; t.p[1][l] = p01; t.p[2][l] = p10; t.p[3][l] = p11; t.p[4][l] = p20; t.p[5][l] = p21;
    t.fn[l] = fn;
    t.bn_axis[l] = (bn << 1) | (ea & 1);
    t.d0_from[l] = d0_from;
    const int incl = wave_inclusive_scan(n_cross);
    const int wv = l >> 6;
    if (lane_id() == 63) t.wave_tot[wv] = incl;
    __syncthreads();
    int before = 0;
    for (int k = 0; k < wv; k++) before += t.wave_tot[k];
    t.pre[l] = before + incl - n_cross;
    if (l == 255) t.pre[256] = before + incl;
    __syncthreads();
    return t.pre[256];
}

// crossing c of the workgroup -> owning lane (last l with pre[l] <= c) ...
__device__ __forceinline__ int crossing_lane(const LaneTable& t, int c) {
    int lo = 0, hi = 256;
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int mid = (lo + hi) >> 1;
        if (t.pre[mid] <= c) lo = mid; else hi = mid;
    }
    return lo;
}

// ---- 1. per workgroup: how many crossings; per line: an upper bound of the segments it will receive ------------
// Every crossing yields at most two segments, so 2 x (crossings on the line) bounds the line's record slice without
// looking at the maps at all (no owner loads, no divisions: those happen once, in k_edge_emit).
template <class FS>
__global__ void __launch_bounds__(256) k_edge_count(FS fs, int is, EdgeWork w) {
    __shared__ LaneTable t;
    const int n_blocks = (*w.n_visible + EG_FACES_PER_BLOCK - 1) / EG_FACES_PER_BLOCK;
    const XcdOrder xo(n_blocks);
    for (int i = blockIdx.x; xo.more(i); i += gridDim.x) {              // fixed grid, uniform trip count per workgroup
        const int blk = xo.unit(i);
        if (blk >= n_blocks) continue;
        bool on;
        int pos = 0, ea = 0, n_cross = 0;
        const int total = publish_lanes(fs, w, blk, is, t, on, pos, ea, n_cross);
        // every lane of a listed face gets its record (n_cross is 0 for a lane that is not `on`)
        if (threadIdx.x < EG_FACES_PER_BLOCK * 6 && blk * EG_FACES_PER_BLOCK + (int)threadIdx.x / 6 < *w.n_visible) {
            const size_t lane6 = (size_t)blk * EG_FACES_PER_BLOCK * 6 + threadIdx.x;
            w.lane_cross[lane6] = make_int2(t.pre[threadIdx.x], n_cross);
        }
        if (threadIdx.x == 0) w.lane_block[blk] = total;
        for (int c0 = 0; c0 < total; c0 += 256) {
            const int c = c0 + threadIdx.x;
            size_t line = 0;
            if (c < total) {
                const int l = crossing_lane(t, c);
                line = ((size_t)(t.bn_axis[l] >> 1) * 2 + (t.bn_axis[l] & 1)) * is + t.d0_from[l] + (c - t.pre[l]);
            }
            // neighbouring crossings fall on the same lines: one atomic per distinct line of the wave (uniform call
            // site); every lane stands for the TWO segments its crossing can yield
            const unsigned long long same = wave_match_any((uint32_t)line, c < total);
            if (c < total && lane_id() == __builtin_ctzll(same)) atomicAdd(&w.line_count[line], 2 * __popcll(same));
        }
        __syncthreads();                                    // the table is rewritten by the next iteration
    }
}

// ---- 2. order-free range allocation: slice start for counts[i] (one atomic per 256), packed with the line's extent ----
__global__ void __launch_bounds__(256) k_alloc_ranges(const int* __restrict__ counts, const int* __restrict__ nz_lo_inv,
                                                     const int* __restrict__ nz_hi1, int4* __restrict__ line_info,
                                                     int* __restrict__ cursor, long n) {
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int c = i < n ? counts[i] : 0;
    const int incl = wave_inclusive_scan(c);
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t0 = s_wave[0], t1 = s_wave[1], t2 = s_wave[2], t3 = s_wave[3];
        s_base = atomicAdd(cursor, t0 + t1 + t2 + t3);
        s_wave[0] = 0; s_wave[1] = t0; s_wave[2] = t0 + t1; s_wave[3] = t0 + t1 + t2;
    }
    __syncthreads();
    if (i < n) line_info[i] = make_int4(nz_lo_inv[i], nz_hi1[i], s_base + s_wave[wv] + incl - c, c);
}

// ---- 3. walk short segments, emit long ones ------------------------------------------------------------
// Item (12 dwords): 0 bits = inward[0] f0[1] f1[2] fix_at_from[3] fix_at_to[4] | fn << 6; 1 inv0; 2 from | to<<16; 3 slot;
// 4 d1_cross; 5 u0; 6 u1; 7..10 reference alpha,r,g,b; 11 inv1.  (line and slots are implied by where the item is
// indexed.)
//
// FACTORED DISTANCE.  Along one queued segment t = d1 - d1_cross keeps its sign s_t (outward: the walk direction;
// inward: the opposite), so KCU:404-405's  dist = q*t*(2/is) +- eps  is  qc*(t + u)  with qc = q*2/is and the
// per-item constant u = s_t*eps/|qc|, and the walk's sum  -sum diff/dist  becomes  (-1/qc) * sum diff/(t + u):
// per pixel one packed add, two v_rcp and one packed fma for both vertices; inv = -1/qc is applied once per item.
// |t + u| >= |u| > 0, so no quotient is infinite.  The one pixel where t == 0 (an inward walk starting exactly on
// an integer crossing: the reference's `0 < dist` is false there, i.e. -eps whatever s_t says) is corrected after
// the loop (fix_at_*; only if that pixel survived the clip to the line's non-zero extent).  Inward segments whose
// limit lies on the unexpected side of the in-pixel (possible only within rounding of a vertex) are not queued.
template <class FS>
__global__ void __launch_bounds__(256) k_edge_emit(FS fs, EdgeGradArgs a, EdgeWork w) {
    __shared__ LaneTable t;
    const int is = a.S;
    const float two_over_is = 2.0f / (float)is;
    const int n_blocks = (*w.n_visible + EG_FACES_PER_BLOCK - 1) / EG_FACES_PER_BLOCK;
    const XcdOrder xo(n_blocks);
    for (int i = blockIdx.x; xo.more(i); i += gridDim.x) {
        const int blk = xo.unit(i);
        if (blk >= n_blocks) continue;
        bool on;
        int pos = 0, ea = 0, n_cross = 0;
        const int total = publish_lanes(fs, w, blk, is, t, on, pos, ea, n_cross);
        const long cbase = w.lane_block[blk];               // scanned: first crossing of this workgroup
        for (int c0 = 0; c0 < total; c0 += 256) {
            const int c = c0 + threadIdx.x;
            const bool active = c < total;
            Segment sg[2];
            bool has[2] = {false, false};
            int l = 0, fn = 0, axis = 0, d0 = 0;
            size_t base = 0, line = 0;
            int4 li = make_int4(0, 0, 0, 0);              // the line's extent and record slice
            float2 near_dot[2] = {make_float2(0, 0), make_float2(0, 0)};
            float4 near_grad[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
            if (active) {
                l = crossing_lane(t, c);
                d0 = t.d0_from[l] + (c - t.pre[l]);
                const int bn = t.bn_axis[l] >> 1;
                axis = t.bn_axis[l] & 1;
                fn = t.fn[l];
                base = (size_t)bn * is * is;
                line = ((size_t)bn * 2 + axis) * is + d0;
                li = w.line_info[line];
            }
            // the lanes of the wave that share this lane's line (found while the load above is in flight)
            const unsigned long long same = wave_match_any((uint32_t)line, active);     // uniform call site
            if (active) {
                const AxisMaps& mo = a.ax[axis];
                // The owner of the in-pixel decides about the outward walk.  Its record and that of its inward
                // neighbour -- the first two pixels of the inward walk, which on small faces is the whole walk -- are
                // requested in the same round trip.
                crossing_segments(t.p[0][l], t.p[1][l], t.p[2][l], t.p[3][l], t.p[4][l], t.p[5][l], axis, fn, is, d0,
                                  is - li.x, li.y - 1,
                                  [&](int e0, int e1, int dir) {
                                      const size_t i0 = base + (size_t)e0 * is + e1;
                                      const size_t i1 = base + (size_t)e0 * is + min(max(e1 - dir, 0), is - 1);
                                      near_dot[0] = mo.dot[i0]; near_grad[0] = mo.grad[i0];
                                      near_dot[1] = mo.dot[i1]; near_grad[1] = mo.grad[i1];
                                      return __float_as_int(near_dot[0].y);
                                  },
                                  sg[0], has[0], sg[1], has[1]);
            }
            const AxisMaps& m = a.ax[axis];
            const size_t line_base = base + (line % is) * is;
            // queued only if the slot and its line's whole slice fit the capacity the workspace gives; otherwise
            // this thread walks the segment itself (still correct, just serial)
            bool queued[2];
            SegRef refs[2];
#pragma unroll
            for (int which = 0; which < 2; which++) {       // 0: outward, 1: inward
                const long slot = 2 * (cbase + c) + which;
                // reference values of the segment (one pixel of the original maps): requested before anything waits
                refs[which] = SegRef{0, 0, 0, 0};
                if (active && has[which]) refs[which] = load_ref(a, axis, base, sg[which].d0, sg[which].ref_pos);
                queued[which] = active && has[which] && segment_queueable(sg[which]) && slot < (long)w.cap &&
                                (long)li.z + li.w <= (long)w.cap;
            }
            // Record positions: the lanes of the wave that share a line take consecutive places under that line's
            // cursor (outward segments first) with ONE atomic per distinct line, sent now and only waited for after
            // the work below.
            const unsigned long long q_out = same & __builtin_amdgcn_ballot_w64(queued[0]),
                                     q_in = same & __builtin_amdgcn_ballot_w64(queued[1]);
            const int n_out = __popcll(q_out), group_leader = active ? __builtin_ctzll(same) : 0;
            const int rank[2] = {mask_rank(q_out), n_out + mask_rank(q_in)};
            int cursor_base = 0;
            if (active && lane_id() == group_leader && n_out + __popcll(q_in) > 0)
                cursor_base = atomicAdd(&w.line_cursor[line], n_out + __popcll(q_in));
#pragma unroll
            for (int which = 0; which < 2; which++) {
                const long slot = 2 * (cbase + c) + which;
                uint4 rec0, rec1, rec2;
                if (queued[which]) {
                    const Segment& q = sg[which];
                    const SegRef& ref = refs[which];
                    const float qc0 = (q.f0 ? q.q0 : 1.0f) * two_over_is, qc1 = (q.f1 ? q.q1 : 1.0f) * two_over_is;
                    const float s_t = (float)(q.inward ? -q.dir : q.dir);
                    const float u0 = s_t * (a.eps / fabsf(qc0)), u1 = s_t * (a.eps / fabsf(qc1));
                    const bool fix = q.inward && (float)q.d1_in == q.d1_cross && q.from <= q.d1_in && q.d1_in <= q.to;
                    const uint32_t bits = (uint32_t)q.inward | ((uint32_t)q.f0 << 1) | ((uint32_t)q.f1 << 2) |
                                          ((fix && q.dir < 0) ? 8u : 0u) | ((fix && 0 < q.dir) ? 16u : 0u);
                    rec0 = make_uint4(bits | ((uint32_t)fn << 6), __float_as_uint(-1.0f / qc0),
                                      (uint32_t)q.from | ((uint32_t)q.to << 16), (uint32_t)slot);
                    rec1 = make_uint4(__float_as_uint(q.d1_cross), __float_as_uint(u0), __float_as_uint(u1),
                                      __float_as_uint(ref.alpha));
                    rec2 = make_uint4(__float_as_uint(ref.r), __float_as_uint(ref.g), __float_as_uint(ref.b),
                                      __float_as_uint(-1.0f / qc1));
                } else if (active) {
                    float g0 = 0, g1 = 0;
                    if (has[which]) {
                        const Segment& q = sg[which];
                        int from = q.from, to = q.to;
                        if (which == 1 && q.oriented) {     // the in-pixel and its inward neighbour are already here
                            const int n0 = q.d1_in, n1 = q.d1_in - q.dir;
                            inline_pixel(q, refs[1], fn, near_dot[0], near_grad[0], n0, from <= n0 && n0 <= to, two_over_is,
                                         a.eps, g0, g1);
                            inline_pixel(q, refs[1], fn, near_dot[1], near_grad[1], n1, from <= n1 && n1 <= to, two_over_is,
                                         a.eps, g0, g1);
                            if (0 < q.dir) to = min(to, n0 - 2); else from = max(from, n0 + 2);
                        }
                        walk_inline(a, m, line_base, q, from, to, refs[which], fn, two_over_is, g0, g1);
                    }
                    if (slot < (long)w.cap) {
                        w.results[slot] = make_float2(g0, g1);
                    } else if (g0 != 0 || g1 != 0) {        // no slot left: fold into the lane's overflow sum
                        const size_t lane_id6 = ((size_t)blk * EG_FACES_PER_BLOCK) * 6 + l;
                        atomicAdd(&w.lane_partial[lane_id6].x, g0);
                        atomicAdd(&w.lane_partial[lane_id6].y, g1);
                    }
                }
                // records are stored in LINE order (the line kernel streams its slice), slots in crossing order
                const int in_line = __shfl(cursor_base, group_leader, 64) + rank[which];
                if (queued[which]) {
                    uint4* rec = (uint4*)(w.items + ((size_t)li.z + in_line) * EG_ITEM_DW);
                    rec[0] = rec0; rec[1] = rec1; rec[2] = rec2;
                }
            }
        }
        __syncthreads();
    }
}

// Lanes that share one segment (16, 8 or 4: all inside a DPP row).  The per-segment work -- record decode, the
// trip count, the DPP sum, the result -- is per-LANE vector work shared by all the segments of a wave, so fewer lanes
// per segment means fewer instructions per segment: 16 -> 8 -> 4 lanes took the kernel from 0.51 to 0.45 to 0.44 ms on
// the headline step (263 M wave-instructions, 86 % VALU-issue-bound, at 16).
#ifndef D3M_EG_ROW
#define D3M_EG_ROW 4
#endif
constexpr int EG_ROW = D3M_EG_ROW;
constexpr int EG_SEG_PER_WAVE = 64 / EG_ROW;   // segments walked concurrently by one wave
constexpr int EG_SORT_CHUNK = 1024;  // segments ordered by length at a time (a multiple of the workgroup size)
__device__ __forceinline__ int from0_clamp(int from, int is) { return min(max(from, 0), is - 1); }

// ---- 4. one workgroup per (view, axis, line, part) ---------------------------------------------------------
// PAD: the LDS image of the line has 2*S + 16 entries (only the first S are filled), so the lanes of a row that has
// finished -- the segments of a wave advance in lock step with the longest -- keep reading inside the allocation and
// the per-iteration address clamp disappears; without PAD (large S) the index is clamped.
template <bool USE_RGB, bool USE_ALPHA, bool PAD>
__global__ void __launch_bounds__(EG_LINE_WAVES * 64) k_edge_lines(EdgeGradArgs a, EdgeWork w) {
    extern __shared__ __attribute__((aligned(16))) float s_line[];
    const int is = a.S;
    // (Keeping a line's parts on one XCD for L2 reuse was measured SLOWER: it piles a heavy line's work
    //  onto one XCD.  Parts of a line are consecutive workgroups, i.e. spread over the XCDs.)
    const int part = blockIdx.x % EG_LINE_PARTS;
    const size_t line = blockIdx.x / EG_LINE_PARTS;          // (b*2 + axis)*S + d0
    const int n_line = __builtin_amdgcn_readfirstlane(w.line_cursor[line]);      // items queued under this line
    // this workgroup's share of them: a contiguous range (EG_LINE_PARTS = 1: all)
    const int item_lo = (int)((long)n_line * part / EG_LINE_PARTS), item_hi = (int)((long)n_line * (part + 1) / EG_LINE_PARTS);
    const int n_items = item_hi - item_lo;
    if (n_items <= 0) return;                                 // nothing for this workgroup (uniform exit)
    const int wv = threadIdx.x >> 6, lane = lane_id();
    const int d0 = (int)(line % is);
    const int axis = (int)((line / is) & 1);
    const size_t bn = line / ((size_t)2 * is);
    const AxisMaps& m = a.ax[axis];
    const size_t line_base = bn * is * is + (size_t)d0 * is;
    const int4 li = w.line_info[line];
    const uint32_t* recs = w.items + ((size_t)li.z + item_lo) * EG_ITEM_DW;    // contiguous records
    // LDS image of the line: per pixel the float4 of gradients (alpha, r, g, b) and the pair (T/2, owner) with
    // T = sum value*grad of the pixel itself, so diff = T - <reference, gradients>: ds_read_b128 + ds_read_b64 per
    // visited pixel, contiguous within a segment's lane group (the groups of a wave start at unrelated offsets, so
    // bank conflicts between groups do occur; a single 32-byte record per pixel -- one address register instead of
    // two -- doubled them and was 8 % slower).  T is kept halved (exact) because the two packed fma below start BOTH
    // halves of the sum from it.
    const int n_lds = PAD ? 2 * is + 16 : is;
    float4* s_grd = (float4*)s_line;
    float2* s_df = (float2*)(s_grd + n_lds);
    // only the line's non-zero-gradient extent is staged: every segment was clipped to it (crossing_segments)
    const int p_lo = __builtin_amdgcn_readfirstlane(is - li.x);
    const int p_hi = __builtin_amdgcn_readfirstlane(li.y - 1);
    for (int p = p_lo + (int)threadIdx.x; p <= p_hi; p += EG_LINE_WAVES * 64) {
        s_grd[p] = m.grad[line_base + p];
        const float2 d = m.dot[line_base + p];
        s_df[p] = make_float2(0.5f * d.x, d.y);
    }
    __syncthreads();
    // EG_SEG_PER_WAVE segments per wave, EG_ROW lanes each: the segments are short once clipped (tens of pixels), so a
    // whole wave per segment spent most of its time on the per-segment prologue / reduction / epilogue.  Here those
    // are per-lane vector work shared by all the wave's segments, the reduction is a few DPP steps inside the lane
    // group, and a walk iteration covers EG_ROW pixels of each segment.
    typedef float v2f __attribute__((ext_vector_type(2)));
    const int row = lane / EG_ROW, rl = lane % EG_ROW;
    // The segments of a wave advance in lock step, so they should be about equally long: the workgroup first
    // orders its segments by length (counting sort on length / 16, longest first, in chunks of EG_SORT_CHUNK) and the
    // waves then take consecutive groups of that order.  Unsorted, a group of four ran at 64 % lane efficiency.
    __shared__ int s_hist[33];
    __shared__ unsigned short s_order[EG_SORT_CHUNK];
    for (int chunk0 = 0; chunk0 < n_items; chunk0 += EG_SORT_CHUNK) {
    const int nc = min(EG_SORT_CHUNK, n_items - chunk0);
    if (threadIdx.x < 33) s_hist[threadIdx.x] = 0;
    __syncthreads();
    int my_key[EG_SORT_CHUNK / (EG_LINE_WAVES * 64)], my_rank[EG_SORT_CHUNK / (EG_LINE_WAVES * 64)];
#pragma unroll
    for (int j = 0; j < EG_SORT_CHUNK / (EG_LINE_WAVES * 64); j++) {
        const int i = threadIdx.x + j * EG_LINE_WAVES * 64;
        my_key[j] = -1;
        if (i < nc) {
            const uint32_t ft = recs[(size_t)(chunk0 + i) * EG_ITEM_DW + 2];
            const int len = (int)(ft >> 16) - (int)(ft & 0xFFFF) + 1;
            my_key[j] = 31 - min(len >> 4, 31);
            my_rank[j] = atomicAdd(&s_hist[my_key[j]], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int k = 0; k < 32; k++) { const int c = s_hist[k]; s_hist[k] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EG_SORT_CHUNK / (EG_LINE_WAVES * 64); j++)
        if (my_key[j] >= 0) s_order[s_hist[my_key[j]] + my_rank[j]] = (unsigned short)(threadIdx.x + j * EG_LINE_WAVES * 64);
    __syncthreads();
    constexpr int STRIDE = EG_LINE_WAVES * EG_SEG_PER_WAVE;
    for (int base = wv * EG_SEG_PER_WAVE; base < nc; base += STRIDE) {
        const bool have = base + row < nc;
        const int it = chunk0 + (have ? s_order[base + row] : s_order[base]);
        const uint4* q = (const uint4*)(recs + (size_t)it * EG_ITEM_DW);
        const uint4 q0v = q[0], q1v = q[1], q2v = q[2];
        const uint32_t bits = q0v.x & 63u;
        const int from = (int)(q0v.z & 0xFFFF), to = have ? (int)(q0v.z >> 16) : -1, fn = (int)(q0v.x >> 6);
        const float d1_cross = __uint_as_float(q1v.x);
        const v2f u = {__uint_as_float(q1v.y), __uint_as_float(q1v.z)};
        const v2f nref_ar = {USE_ALPHA ? -__uint_as_float(q1v.w) : 0.0f, USE_RGB ? -__uint_as_float(q2v.x) : 0.0f};
        const v2f nref_gb = {USE_RGB ? -__uint_as_float(q2v.y) : 0.0f, USE_RGB ? -__uint_as_float(q2v.z) : 0.0f};
        // diff_grad of one pixel (KCU:385-396 / :473-479 regrouped): two packed fma + one add.
        auto diff_of = [&](const float4 g, const float2 d) {
            v2f p = {d.x, d.x};
            p = __builtin_elementwise_fma(v2f{g.x, g.y}, nref_ar, p);
            p = __builtin_elementwise_fma(v2f{g.z, g.w}, nref_gb, p);
            return p.x + p.y;
        };
        // A pixel counts if it lies in the segment, its diff_grad is not <= 0 (KCU:401/:481; NaN passes, as in the
        // reference) and -- inward walks only -- it belongs to the face (KCU:470).  The three conditions are combined
        // as wave masks on the scalar unit; the vector unit only issues the compares and ONE select.
        const unsigned long long m_outward = ~__builtin_amdgcn_ballot_w64((bits & 1u) != 0);
        // the rows advance together: as many 16-pixel steps as the longest of the four needs (a scalar trip count)
        const int len = to - from + 1;
        int max_len = __builtin_amdgcn_readlane(len, 0);
#pragma unroll
        for (int r = 1; r < EG_SEG_PER_WAVE; r++) max_len = max(max_len, __builtin_amdgcn_readlane(len, r * EG_ROW));
        const int n_iter = (max_len + EG_ROW - 1) / EG_ROW;
        float t = (float)(from + rl) - d1_cross;            // t = d1 - d1_cross advances by exact steps
        v2f acc = {0.0f, 0.0f};
        if (PAD) {
            const float4* pg = s_grd + from + rl;
            const float2* pd = s_df + from + rl;
            const float4* pg_to = s_grd + to;
            v2f den = u + t;                                // advances by exact steps of 16 as well
            if (m_outward == ~0ull) {                       // outward walks only (the common case): no owner test
                for (int k = 0; k < n_iter; k++) {
                    const float diff = diff_of(*pg, *pd);
                    const unsigned long long keep = __builtin_amdgcn_ballot_w64(pg <= pg_to) &
                                                    __builtin_amdgcn_ballot_w64(!(diff <= 0));
                    const v2f r = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
                    // under the lane mask, not by multiplying with 0: a lane outside its segment may sit exactly on
                    // den == 0 (1/0 = inf, 0 * inf = NaN)
                    if (__builtin_amdgcn_inverse_ballot_w64(keep)) acc = __builtin_elementwise_fma(v2f{diff, diff}, r, acc);
                    pg += EG_ROW;
                    pd += EG_ROW;
                    den += (float)EG_ROW;
                }
            } else {
                for (int k = 0; k < n_iter; k++) {
                    const float4 g = *pg;
                    const float2 d = *pd;
                    const float diff = diff_of(g, d);
                    const unsigned long long keep = __builtin_amdgcn_ballot_w64(pg <= pg_to) &
                                                    __builtin_amdgcn_ballot_w64(!(diff <= 0)) &
                                                    (__builtin_amdgcn_ballot_w64(__float_as_int(d.y) == fn) | m_outward);
                    const v2f r = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
                    // under the lane mask, not by multiplying with 0: a lane outside its segment may sit exactly on
                    // den == 0 (1/0 = inf, 0 * inf = NaN)
                    if (__builtin_amdgcn_inverse_ballot_w64(keep)) acc = __builtin_elementwise_fma(v2f{diff, diff}, r, acc);
                    pg += EG_ROW;
                    pd += EG_ROW;
                    den += (float)EG_ROW;
                }
            }
        } else {
            int d1 = from + rl;
            for (int k = 0; k < n_iter; k++) {
                const int dc = min(d1, is - 1);                       // keep the LDS address inside the line
                const float2 d = s_df[dc];
                const float diff = diff_of(s_grd[dc], d);
                const unsigned long long keep = __builtin_amdgcn_ballot_w64(d1 <= to) &
                                                __builtin_amdgcn_ballot_w64(!(diff <= 0)) &
                                                (__builtin_amdgcn_ballot_w64(__float_as_int(d.y) == fn) | m_outward);
                const v2f den = u + t;
                const v2f r = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
                if (__builtin_amdgcn_inverse_ballot_w64(keep)) acc = __builtin_elementwise_fma(v2f{diff, diff}, r, acc);
                d1 += EG_ROW;
                t += (float)EG_ROW;
            }
        }
        // row sums: quad swaps, then the two mirrors -> every lane of the row holds the segment's sum
        float s0 = acc.x, s1 = acc.y;
        s0 += dpp_f32<0xB1>(s0);  s1 += dpp_f32<0xB1>(s1);
        s0 += dpp_f32<0x4E>(s0);  s1 += dpp_f32<0x4E>(s1);
        if (EG_ROW >= 8) { s0 += dpp_f32<0x141>(s0); s1 += dpp_f32<0x141>(s1); }
        if (EG_ROW == 16) { s0 += dpp_f32<0x140>(s0); s1 += dpp_f32<0x140>(s1); }
        if (have && rl == 0) {
            const float inv0 = __uint_as_float(q0v.y), inv1 = __uint_as_float(q2v.w);
            if (bits & 24u) {                                  // the t == 0 pixel must use -eps (see item format)
                const int df = (bits & 8u) ? from : to;
                const float2 d = s_df[df];
                const float diff = diff_of(s_grd[df], d);
                const float dpos = (!(diff <= 0) && (!(bits & 1u) || __float_as_int(d.y) == fn)) ? diff : 0.0f;
                if (u.x * inv0 < 0) s0 -= 2.0f * (dpos * __builtin_amdgcn_rcpf(u.x));
                if (u.y * inv1 < 0) s1 -= 2.0f * (dpos * __builtin_amdgcn_rcpf(u.y));
            }
            w.results[(int)q0v.w] = make_float2((bits & 2u) ? inv0 * s0 : 0.0f, (bits & 4u) ? inv1 * s1 : 0.0f);
        }
    }
    __syncthreads();                                          // s_order / s_hist are rewritten by the next chunk
    }
}

// ---- 5. per visible face: the results of its six lanes' crossings, stored once ---------------------------
template <class FS>
__global__ void __launch_bounds__(256) k_edge_gather(FS fs, EdgeWork w, float* __restrict__ grad_faces, VertexTarget vt) {
    __shared__ float2 s_g[256];
    const int n_vis = *w.n_visible;
    const int n_blocks = (n_vis + EG_FACES_PER_BLOCK - 1) / EG_FACES_PER_BLOCK;
    const XcdOrder xo(n_blocks);
    for (int i = blockIdx.x; xo.more(i); i += gridDim.x) {
        const int blk = xo.unit(i);
        if (blk >= n_blocks) continue;
        const int t = threadIdx.x;
        const int pos = blk * EG_FACES_PER_BLOCK + t / 6, ea = t % 6;
        const bool on = t < EG_FACES_PER_BLOCK * 6 && pos < n_vis;
        float2 g = make_float2(0.0f, 0.0f);
        if (on) {
            const int2 lc = w.lane_cross[(size_t)pos * 6 + ea];
            g = w.lane_partial[(size_t)pos * 6 + ea];
            // slots (2c, 2c+1) of the lane's crossings c, contiguous and 16-byte aligned: one float4 per crossing, four
            // crossings requested per round, added in slot order; slots past cap were folded into lane_partial
            const long c_first = (long)w.lane_block[blk] + lc.x, c_last = min(c_first + (long)lc.y, (long)(w.cap >> 1));
            const float4* res4 = (const float4*)w.results;
            for (long c = c_first; c < c_last; c += 4) {
                float4 r[4];
#pragma unroll
                for (int j = 0; j < 4; j++) r[j] = c + j < c_last ? res4[c + j] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    g.x += r[j].x; g.y += r[j].y;
                    g.x += r[j].z; g.y += r[j].w;
                }
            }
        }
        s_g[t] = g;
        __syncthreads();
        if (on && ea == 0) {
            float acc[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 6; e++) {
                const int edge = e >> 1, axis = e & 1;
                const float2 r = s_g[t + e];
                acc[edge * 2 + (1 - axis)] += r.x;                    // vertex pi[0] = edge, component 1 - axis (KCU:406)
                acc[((edge + 1) % 3) * 2 + (1 - axis)] += r.y;        // vertex pi[1] = edge + 1            (KCU:411)
            }
            const long gi = w.visible_list[pos];
            if (vt.gv) {
                const int Fp = fs.num_faces();
                const int b = (int)(gi / Fp), f = (int)(gi % Fp);
#pragma unroll
                for (int v = 0; v < 3; v++) {
                    float* g = vt.vertex(b, f, v);
                    atomicAdd(&g[0], acc[2 * v + 0]);
                    atomicAdd(&g[1], acc[2 * v + 1]);
                }
            } else {
                float* gf = grad_faces + (size_t)gi * 9;
#pragma unroll
                for (int v = 0; v < 3; v++) {
                    gf[3 * v + 0] = acc[2 * v + 0];
                    gf[3 * v + 1] = acc[2 * v + 1];
                    gf[3 * v + 2] = 0.0f;
                }
            }
        }
        __syncthreads();
    }
}

// Per-pixel walk records in both orientations, one pass over the maps (replaces five transposes): a 32x32 tile per
// workgroup; the row-major records are written straight away, the column-major ones through an LDS tile.
__global__ void __launch_bounds__(256) k_pack_maps(const int32_t* __restrict__ fi, const float* __restrict__ alpha,
                                                  const float* __restrict__ galpha, const float* __restrict__ rgb,
                                                  const float* __restrict__ grgb, float4* __restrict__ grad_row,
                                                  float2* __restrict__ dot_row, float4* __restrict__ grad_col,
                                                  float2* __restrict__ dot_col, int* __restrict__ nz_lo_inv,
                                                  int* __restrict__ nz_hi1, int S, float2* __restrict__ lane_partial,
                                                  const int* __restrict__ n_visible, GradScale gs) {
    __shared__ float4 t_grad[32][33];
    __shared__ float2 t_dot[32][33];
    __shared__ int s_col_lo_inv[32], s_col_hi1[32];
    if (threadIdx.x < 32) { s_col_lo_inv[threadIdx.x] = 0; s_col_hi1[threadIdx.x] = 0; }
    __syncthreads();
    const int b = blockIdx.z;
    float s_rgb, s_alpha, s_depth;
    gs.get(s_rgb, s_alpha, s_depth);
    {   // the overflow sums of the (visible face, edge, axis) lanes start at zero (k_edge_emit adds, k_edge_gather reads)
        const long n_threads = (long)gridDim.x * gridDim.y * gridDim.z * 256;
        const long me = (((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        for (long i = me; i < (long)*n_visible * 6; i += n_threads) lane_partial[i] = make_float2(0.0f, 0.0f);
    }
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const size_t plane = (size_t)b * S * S;
    for (int r = ty; r < 32; r += 8) {
        const int y = y0 + r, x = x0 + tx;
        bool nz = false;
        if (y < S && x < S) {
            const size_t i = plane + (size_t)y * S + x;
            float4 g = make_float4(0, 0, 0, 0);
            float dot = 0;
            if (alpha) { g.x = galpha[i] * s_alpha; dot += alpha[i] * g.x; }
            if (rgb) {
                g.y = grgb[3 * i] * s_rgb; g.z = grgb[3 * i + 1] * s_rgb; g.w = grgb[3 * i + 2] * s_rgb;
                dot += rgb[3 * i] * g.y;
                dot += rgb[3 * i + 1] * g.z;
                dot += rgb[3 * i + 2] * g.w;
            }
            const float2 d = make_float2(dot, __int_as_float(fi[i]));
            grad_row[i] = g;
            dot_row[i] = d;
            t_grad[r][tx] = g;
            t_dot[r][tx] = d;
            nz = g.x != 0 || g.y != 0 || g.z != 0 || g.w != 0 || dot != 0;
        }
        // non-zero extents: this tile's share of row y (one half-wave = one tile row) and of its 32 columns
        const unsigned long long ball = __ballot(nz);
        const unsigned half = (threadIdx.x & 32) ? (unsigned)(ball >> 32) : (unsigned)ball;
        if (half != 0 && tx == 0) {
            const size_t line = ((size_t)b * 2 + 1) * S + (y0 + r);
            atomicMax(&nz_lo_inv[line], S - (x0 + (__ffs((int)half) - 1)));
            atomicMax(&nz_hi1[line], x0 + (32 - __clz((int)half)));
        }
        if (nz) {
            atomicMax(&s_col_lo_inv[tx], S - (y0 + r));
            atomicMax(&s_col_hi1[tx], y0 + r + 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < 32 && s_col_hi1[threadIdx.x] != 0 && x0 + (int)threadIdx.x < S) {
        const size_t line = ((size_t)b * 2 + 0) * S + (x0 + threadIdx.x);
        atomicMax(&nz_lo_inv[line], s_col_lo_inv[threadIdx.x]);
        atomicMax(&nz_hi1[line], s_col_hi1[threadIdx.x]);
    }
    for (int r = ty; r < 32; r += 8) {
        const int x = x0 + r, y = y0 + tx;
        if (x < S && y < S) {
            const size_t i = plane + (size_t)x * S + y;
            grad_col[i] = t_grad[tx][r];
            dot_col[i] = t_dot[tx][r];
        }
    }
}

inline size_t eg_align(size_t v) { return (v + 255) / 256 * 256; }

// ---- shared visibility ------------------------------------------------------------------------------------
// Which faces own a pixel depends only on face_index_map.  One blob (flags | compacted list | chunk scan | count)
// can be built once per forward result and handed to every backward operator (d3m_visibility in the C ABI).
struct VisibilityView {
    int* flags;        // [B*F]  FLAG_HIDDEN / FLAG_VISIBLE (FLAG_LARGE is written later by the gathered passes)
    int* list;         // [B*F]  ascending indices of the faces with flags != 0
    int* vis_block;    // [B*F/1024 + 2]
    int* count;        // [1]
};

inline size_t visibility_bytes(long nf) {
    return eg_align((size_t)nf * 4) * 2 + eg_align(((size_t)nf / EG_COMPACT_CHUNK + 2) * 4) + 256;
}

inline VisibilityView visibility_view(void* blob, long nf) {
    char* p = (char*)blob;
    VisibilityView v;
    v.flags = (int*)p;                        p += eg_align((size_t)nf * 4);
    v.list = (int*)p;                         p += eg_align((size_t)nf * 4);
    v.vis_block = (int*)p;                    p += eg_align(((size_t)nf / EG_COMPACT_CHUNK + 2) * 4);
    v.count = (int*)p;
    return v;
}

inline hipError_t run_visibility(const int32_t* face_index_map, const VisibilityView& v, int B, int F, int S, hipStream_t st) {
    const long nf = (long)B * F;
    hipError_t e = zero_async(v.flags, eg_align((size_t)nf * 4), st);
    if (e != hipSuccess) return e;
    LAUNCH("k_mark_visible", k_mark_visible, dim3((unsigned)(((long)B * S * S + 255) / 256)), dim3(256), st, face_index_map,
           v.flags, B, F, S);
    const int n_chunks = (int)((nf + EG_COMPACT_CHUNK - 1) / EG_COMPACT_CHUNK);
    LAUNCH("k_count_visible", k_count_visible, dim3(n_chunks), dim3(256), st, (const int*)v.flags, v.vis_block, nf);
    LAUNCH("k_scan_small", k_scan_small, dim3(1), dim3(1024), st, v.vis_block, n_chunks, (const int*)nullptr, 1, v.count);
    LAUNCH("k_compact_visible", k_compact_visible, dim3(n_chunks), dim3(256), st, (const int*)v.flags, v.list,
           (const int*)v.vis_block, nf);
    return hipGetLastError();
}

// count -> crossing base per workgroup -> record slice per line
template <class FS>
inline hipError_t launch_edge_count(FS fs, const EdgeWork& w, const int* nz_lo_inv, const int* nz_hi1, int B, int S,
                                    hipStream_t st) {
    const long nf = (long)B * fs.num_faces(), nl = (long)B * 2 * S;
    const long g6_full = (nf + EG_FACES_PER_BLOCK - 1) / EG_FACES_PER_BLOCK;
    const dim3 g6((unsigned)(g6_full < 8192 ? (g6_full + 7) / 8 * 8 : 8192));      // a multiple of 8: see XcdOrder
    LAUNCH("k_edge_count", k_edge_count<FS>, g6, dim3(256), st, fs, S, w);
    LAUNCH("k_scan_small", k_scan_small, dim3(1), dim3(1024), st, w.lane_block, 0, (const int*)w.n_visible,
           EG_FACES_PER_BLOCK, w.alloc);
    LAUNCH("k_alloc_ranges", k_alloc_ranges, dim3((unsigned)((nl + 255) / 256)), dim3(256), st, (const int*)w.line_count,
           nz_lo_inv, nz_hi1, w.line_info, w.alloc + 1, nl);
    return hipGetLastError();
}

// ---- host side ----------------------------------------------------------------------------------------
struct EdgeLayout {
    size_t off_grad_row, off_dot_row, off_grad_col, off_dot_col;
    size_t off_zero, zero_bytes;   // visible | line_count | line_cursor | alloc | n_visible
    size_t off_visible, off_line_count, off_line_cursor, off_nz_lo, off_nz_hi, off_alloc, off_visible_list, off_lane_cross,
        off_lane_partial, off_line_offset, off_vis_block, off_lane_block;
    size_t off_items;              // items | results follow, sized by capacity
    size_t fixed_bytes;
};


inline EdgeLayout edge_layout(int B, int F, int S) {
    const size_t px = (size_t)B * S * S, nf = (size_t)B * F, nl = (size_t)B * 2 * S;
    EdgeLayout L;
    size_t o = 0;
    L.off_grad_row = o; o += eg_align(px * 16);
    L.off_dot_row = o;  o += eg_align(px * 8);
    L.off_grad_col = o; o += eg_align(px * 16);
    L.off_dot_col = o;  o += eg_align(px * 8);
    L.off_zero = o;
    L.off_visible = o;      o += eg_align(nf * 4);
    L.off_line_count = o;   o += eg_align(nl * 4);
    L.off_line_cursor = o;  o += eg_align(nl * 4);
    L.off_nz_lo = o;        o += eg_align(nl * 4);
    L.off_nz_hi = o;        o += eg_align(nl * 4);
    L.off_alloc = o;        o += 256;                 // alloc[0], alloc[1], n_visible (alloc[2])
    L.zero_bytes = o - L.off_zero;
    L.off_visible_list = o; o += eg_align(nf * 4);
    // at most half of the faces can be front-facing AND own a pixel only if ... no such bound: size for all
    L.off_lane_cross = o;   o += eg_align(nf * 6 * 8);
    L.off_lane_partial = o; o += eg_align(nf * 6 * 8);
    L.off_line_offset = o;  o += eg_align(nl * 16);
    L.off_vis_block = o;    o += eg_align((nf / EG_COMPACT_CHUNK + 2) * 4);
    L.off_lane_block = o;   o += eg_align((nf / EG_FACES_PER_BLOCK + 2) * 4);
    L.off_items = o;
    L.fixed_bytes = o;
    return L;
}

constexpr size_t EG_BYTES_PER_ITEM = EG_ITEM_DW * 4 + 8;
constexpr int EG_ITEMS_PER_FACE_DEFAULT = 4;

inline size_t edge_grad_workspace_bytes(int B, int F, int S) {
    return edge_layout(B, F, S).fixed_bytes + eg_align((size_t)EG_ITEMS_PER_FACE_DEFAULT * B * F * EG_BYTES_PER_ITEM) + 1024;
}

template <class FS>
int run_edge_grad(FS fs, PixelMaps m, float* grad_faces, VertexTarget vt, const VisibilityView* shared_vis, GradScale gs,
                  int B, float eps, void* ws, size_t ws_bytes, hipStream_t st, int* last_err) {
    const int S = m.S, F = fs.num_faces();
    if (S > 65535 || F > (1 << 26) || (long)B * 2 * S >= (1l << 31)) return 1;   // item packing / line key limits (D3M_ERR_INVALID)
    const EdgeLayout L = edge_layout(B, F, S);
    if (!ws || ws_bytes < L.fixed_bytes + 1024) return 2;       // D3M_ERR_WORKSPACE
    char* p = (char*)ws;
    size_t cap = (ws_bytes - L.fixed_bytes - 768) / EG_BYTES_PER_ITEM;
    cap = cap > 256 ? cap - 128 : 0;                            // slack for the two 256-byte alignments below
    if (cap > 0x7FFFFF00) cap = 0x7FFFFF00;
    cap &= ~(size_t)1;                                          // a crossing's two slots stay together (k_edge_gather)
    EdgeWork w;
    w.visible = (int*)(p + L.off_visible);
    w.line_count = (int*)(p + L.off_line_count);
    w.line_cursor = (int*)(p + L.off_line_cursor);
    w.alloc = (int*)(p + L.off_alloc);
    w.n_visible = w.alloc + 2;
    w.visible_list = (int*)(p + L.off_visible_list);
    w.lane_cross = (int2*)(p + L.off_lane_cross);
    w.lane_partial = (float2*)(p + L.off_lane_partial);
    w.line_info = (int4*)(p + L.off_line_offset);
    w.vis_block = (int*)(p + L.off_vis_block);
    w.lane_block = (int*)(p + L.off_lane_block);
    w.items = (uint32_t*)(p + L.off_items);
    const size_t off_res = eg_align(L.off_items + cap * EG_ITEM_DW * 4);
    w.results = (float2*)(p + off_res);
    w.cap = (int)cap;

    float4* grad_row = (float4*)(p + L.off_grad_row);
    float2* dot_row = (float2*)(p + L.off_dot_row);
    float4* grad_col = (float4*)(p + L.off_grad_col);
    float2* dot_col = (float2*)(p + L.off_dot_col);
    // per-call counters; the visibility flags too unless the caller brought a d3m_visibility
    const size_t zero_from = shared_vis ? L.off_line_count : L.off_zero;
    hipError_t e = zero_async(p + zero_from, L.off_zero + L.zero_bytes - zero_from, st);
    if (e != hipSuccess) { *last_err = (int)e; return 3; }
    if (shared_vis) {
        w.visible = shared_vis->flags;
        w.visible_list = shared_vis->list;
        w.n_visible = shared_vis->count;
    }
    const long nf = (long)B * F, nl = (long)B * 2 * S;
    if (!shared_vis) {
        LAUNCH("k_mark_visible", k_mark_visible, dim3((unsigned)(((long)B * S * S + 255) / 256)), dim3(256), st,
               m.face_index_map, w.visible, B, F, S);
        const int n_chunks = (int)((nf + EG_COMPACT_CHUNK - 1) / EG_COMPACT_CHUNK);
        LAUNCH("k_count_visible", k_count_visible, dim3(n_chunks), dim3(256), st, (const int*)w.visible, w.vis_block, nf);
        LAUNCH("k_scan_small", k_scan_small, dim3(1), dim3(1024), st, w.vis_block, n_chunks, (const int*)nullptr, 1,
               w.n_visible);
        LAUNCH("k_compact_visible", k_compact_visible, dim3(n_chunks), dim3(256), st, (const int*)w.visible, w.visible_list,
               (const int*)w.vis_block, nf);
    }
    LAUNCH("k_pack_maps", k_pack_maps, dim3((S + 31) / 32, (S + 31) / 32, B), dim3(256), st, m.face_index_map,
           m.use_alpha ? m.alpha_map : nullptr, m.use_alpha ? m.grad_alpha_map : nullptr, m.use_rgb ? m.rgb_map : nullptr,
           m.use_rgb ? m.grad_rgb_map : nullptr, grad_row, dot_row, grad_col, dot_col, (int*)(p + L.off_nz_lo),
           (int*)(p + L.off_nz_hi), S, w.lane_partial, (const int*)w.n_visible, gs);
    EdgeGradArgs a;
    a.ax[0] = AxisMaps{grad_col, dot_col};
    a.ax[1] = AxisMaps{grad_row, dot_row};
    a.alpha_map = m.alpha_map; a.rgb_map = m.rgb_map;
    a.nz_lo_inv = (const int*)(p + L.off_nz_lo); a.nz_hi1 = (const int*)(p + L.off_nz_hi);
    a.S = S; a.use_rgb = m.use_rgb; a.use_alpha = m.use_alpha; a.eps = eps; a.n_lines = (unsigned)((long)B * 2 * S);
    // count / emit / gather walk the compacted list with a fixed grid (n_visible is only known on the device);
    // workgroups past n_visible exit on their first load
    const long g6_full = (nf + EG_FACES_PER_BLOCK - 1) / EG_FACES_PER_BLOCK;
    const dim3 g6((unsigned)(g6_full < 8192 ? (g6_full + 7) / 8 * 8 : 8192));      // a multiple of 8: see XcdOrder
    e = launch_edge_count(fs, w, a.nz_lo_inv, a.nz_hi1, B, S, st);
    if (e != hipSuccess) { *last_err = (int)e; return 3; }
    LAUNCH("k_edge_emit", k_edge_emit<FS>, g6, dim3(256), st, fs, a, w);
    const size_t smem_pad = (size_t)(2 * S + 16) * 24;
    const bool pad = smem_pad <= 36 * 1024;
    const size_t smem = pad ? smem_pad : (size_t)S * 24;
    const dim3 glines((unsigned)(nl * EG_LINE_PARTS));
#define D3M_LINES1(RGB, ALPHA, PADDED)                                                                               \
    do {                                                                                                             \
        if (smem > 64 * 1024) {                                                                                      \
            e = hipFuncSetAttribute((const void*)k_edge_lines<RGB, ALPHA, PADDED>,                                    \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                          \
            if (e != hipSuccess) { *last_err = (int)e; return 3; }                                                   \
        }                                                                                                            \
        LAUNCH_SMEM("k_edge_lines", (k_edge_lines<RGB, ALPHA, PADDED>), glines, dim3(EG_LINE_WAVES * 64), smem, st, a, w); \
    } while (0)
#define D3M_LINES(RGB, ALPHA)                                                                                        \
    do {                                                                                                             \
        if (pad) D3M_LINES1(RGB, ALPHA, true);                                                                       \
        else D3M_LINES1(RGB, ALPHA, false);                                                                          \
    } while (0)
    if (smem > 160 * 1024) return 1;                            // a line does not fit LDS (S > ~6800)
    if (m.use_rgb && m.use_alpha) D3M_LINES(true, true);
    else if (m.use_rgb) D3M_LINES(true, false);
    else D3M_LINES(false, true);
#undef D3M_LINES1
#undef D3M_LINES
    LAUNCH("k_edge_gather", k_edge_gather<FS>, g6, dim3(256), st, fs, w, grad_faces, vt);
    e = hipGetLastError();
    if (e != hipSuccess) { *last_err = (int)e; return 3; }
    return 0;
}

}  // namespace d3m
