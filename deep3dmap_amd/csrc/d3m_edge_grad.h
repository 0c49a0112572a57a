// d3m_edge_grad.h -- the edge / silhouette gradient (reference: backward_pixel_map_cuda_kernel,
// KCU:245-503) restructured for gfx950.
//
// The reference runs one thread per face; for every integer crossing d0 of every edge, on both axes,
// that thread walks the image from the edge out to the image BORDER (KCU:354-414) and inward to the
// opposite edge (KCU:417-495): 1.4 M crossings per 8 views of the 100k-triangle mesh at 512^2, walks of ~250 pixels.
//
// LINE-MAJOR formulation.  A walk only ever moves along ONE image line (a row for axis 1, a column for
// axis 0), and a line is shared by hundreds of walks.  So:
//   0. visibility     faces that own no pixel cannot contribute: one mark per face (left by the forward's tile pass, or
//                     by k_mark_visible_bytes from a face_index_map), then flags + compacted list, shared with the
//                     gathered texture / depth pass through d3m_visibility;
//  THE PLAN (geometry only: faces + face_index_map; d3m_edge_plan can build it right after the forward pass)
//   1. k_edge_count_window   a workgroup iteration takes 42 visible faces x 6 (edge, axis) lanes; a lane crosses the
//                     consecutive lines d0_from..d0_to, so it adds +1 / -1 at the ends of its range in an LDS array
//                     indexed by the line, a running sum turns the touched window into counts, and every counted
//                     line costs the workgroup one global atomic (k_edge_count: the by-key form, for S > 2048);
//   2. k_scan_small / k_alloc_ranges   crossing base per workgroup, record slice per line (no same-address atomics);
//   3. k_edge_scatter the lanes' crossings flattened over the threads (one crossing each per round): the geometry of a
//                     crossing's two walks, computed once, as one 16-byte record written in LINE order; a crossing
//                     whose walks cannot contribute gets none;
//  THE GRADIENT (needs the gradient maps)
//   4. k_pack_maps    what a walk reads per pixel -- (grad_alpha, grad_rgb), sum value*grad, owner -- plus each line's
//                     non-zero-gradient extent, from gradient maps or from the output images' gradients; the fused fit
//                     objective's epilogue writes the same records itself (d3m_lit.h) and this pass disappears;
//   5. k_edge_lines   one workgroup per (view, axis, line), lines in XCD-contiguous order: the line's per-pixel records
//                     and values staged in LDS once; a thread per (crossing, outward | inward) turns the crossing's
//                     record into a segment clipped to the extent, walks it if short, queues it in LDS if long; the
//                     queued segments are ordered by length and walked sixteen per wave (four lanes each) with the
//                     factored distance.  The walks never touch global memory;
//      k_edge_overflow  crossings that got no record because the workspace is smaller than the scene needs are walked
//                     from global memory by one thread each; leaves at once otherwise;
//   6. k_edge_gather  six lanes per visible face fetch their crossings' results (xpos) and add them in order; stored to
//                     grad_faces or accumulated into the vertex gradient (VertexTarget).
// Deterministic up to the final vertex atomics.  The reference OVERWRITES the 9 entries of every front-facing face
// (KCU:501-502) and leaves culled ones alone (KCU:270); with the caller's zero-initialised grad_faces
// (rasterize.py:111, a precondition of the C ABI) writing only the faces that own a pixel is the same thing: every
// other front-facing face would get zeros.  Per visited pixel the expressions are those of KCU:385-412 / :473-493
// regrouped (see "FACTORED DISTANCE" and k_pack_maps); the divisions inside the walk use v_rcp_f32 (1 ulp), far
// inside the 1e-3 gradient tolerance.
#pragma once
#include <algorithm>
#include <atomic>
#include <type_traits>
#include "d3m_backward.h"
#include "d3m_face_major.h"
#include "d3m_launch.h"

namespace d3m {

#ifndef D3M_EG_INLINE_MAX
#define D3M_EG_INLINE_MAX 6
#endif
constexpr int EG_INLINE_MAX = D3M_EG_INLINE_MAX;   // segments of at most this many pixels are walked by the owning lane
#ifndef D3M_EG_LINE_WAVES
#define D3M_EG_LINE_WAVES 8
#endif
constexpr int EG_LINE_WAVES = D3M_EG_LINE_WAVES;   // waves per workgroup: parts x waves walk one line's items concurrently
constexpr int EG_ITEM_DW = 8;      // dwords per item

// What a walk reads per pixel, row-major like the maps (pixel (y, x) of view b at b*S*S + y*S + x); k_pack_maps (or
// the fused fit epilogue, d3m_lit.h) builds it in one pass:
//   grad[i] = (grad_alpha, grad_r, grad_g, grad_b)              (0 for a disabled output)
//   dot[i]  = (sum value*grad of the pixel itself, owner face index bits)
// so that  diff_grad = dot.x - <reference values, grad>  (KCU:385-396 / :473-479 regrouped: 4 fma).
// A row line (axis 1) reads its pixels contiguously, a column line (axis 0) with stride S -- once, into LDS; transposed
// copies of the records (48 more bytes per pixel written and read) bought nothing measurable.
// `go` (NULL = 1): a scalar factor the records still lack -- the fused fit objective writes them before the gradient
// of the loss is known.  The walks are linear in it up to its SIGN (KCU:401/:481 keep a pixel iff diff_grad > 0): the
// sign is applied where records are read, the magnitude where the sums are gathered.
struct EdgeGradArgs {
    const float4* grad;
    const float2* dot;
    const float* go;
    __device__ __forceinline__ size_t pixel(int axis, size_t view_base, int d0, int d1) const {   // line d0, position d1
        return view_base + (axis ? (size_t)d0 * S + d1 : (size_t)d1 * S + d0);
    }
    __device__ __forceinline__ float go_sign() const { return (go && *go < 0) ? -1.0f : 1.0f; }
    __device__ __forceinline__ float go_abs() const { return go ? fabsf(*go) : 1.0f; }
    const float* alpha_map;   // original [B,S,S] / [B,S,S,3] maps: reference values of a segment (one pixel each)
    const float* rgb_map;
    // per line (b*2 + axis)*S + d0: extent of the pixels whose gradients are not all zero, built by k_pack_maps with
    // atomicMax on zeroed arrays: nz_lo_inv = S - (first such d1), nz_hi1 = (last such d1) + 1; 0 = none.
    const int* nz_lo_inv;
    const int* nz_hi1;
    int S, use_rgb, use_alpha;
    float eps;
    unsigned n_lines;   // B*2*S
    int sparse_max;     // alpha only: a line with at most this many pixels that can contribute to an OUTWARD walk takes the
                        // sparse form of those walks (k_edge_lines, "SPARSE OUTWARD WALKS"); 0 = always the dense walk
    // DIRECT (alpha only, final gradients: round 6).  No per-pixel records at all: the alpha gradient is read where it already
    // is -- an internal-layout map [B,S,S] (ga_map: the reference-shaped operator's grad_alpha_map) or the OUTPUT image's
    // gradient [B,s,s] through the row flip and the 2x2 pooling's adjoint (ga_img: the silhouette node) --, the owner from
    // face_index_map, T = alpha * gradient; the lines have no extents (nz_* NULL: the whole line).  k_pack_maps -- a pass
    // over every pixel writing 24 bytes each, 55 us of a 32-view silhouette step -- does not run.
    const float* ga_map;
    const float* ga_img;
    const int32_t* fi_direct;
    int img_s, img_aa;
    __device__ __forceinline__ bool direct() const { return ga_map != nullptr || ga_img != nullptr; }
    __device__ __forceinline__ float direct_ga(size_t pi) const {       // pi = (b*S + y)*S + x
        if (ga_map) return ga_map[pi];
        const size_t plane = (size_t)S * S;
        const int b = (int)(pi / plane), rem = (int)(pi - (size_t)b * plane), y = rem / S, x = rem - y * S;
        const int yo = img_aa ? (S - 1 - y) >> 1 : S - 1 - y, xo = img_aa ? x >> 1 : x;
        return ga_img[((size_t)b * img_s + yo) * img_s + xo] * (img_aa ? 0.25f : 1.0f);
    }
    // the same for position d1 of line (view, axis, d0) at pi = pixel(axis, view * S * S, d0, d1) -- the staging loops' form (no divisions)
    __device__ __forceinline__ float direct_ga_at(size_t view, int axis, int d0, int d1, size_t pi) const {
        if (ga_map) return ga_map[pi];
        const int y = axis ? d0 : d1, x = axis ? d1 : d0;
        const int yo = img_aa ? (S - 1 - y) >> 1 : S - 1 - y, xo = img_aa ? x >> 1 : x;
        return ga_img[(view * img_s + yo) * img_s + xo] * (img_aa ? 0.25f : 1.0f);
    }
    __device__ __forceinline__ float4 rec_grad(size_t pi) const {
        return direct() ? make_float4(direct_ga(pi), 0.0f, 0.0f, 0.0f) : grad[pi];
    }
    __device__ __forceinline__ float2 rec_dot(size_t pi) const {
        if (!direct()) return dot[pi];
        return make_float2(alpha_map[pi] * direct_ga(pi), __int_as_float(fi_direct[pi]));
    }
    __device__ __forceinline__ int extent_lo(size_t line) const { return nz_lo_inv ? S - nz_lo_inv[line] : 0; }
    __device__ __forceinline__ int extent_hi(size_t line) const { return nz_hi1 ? nz_hi1[line] - 1 : S - 1; }
    // alpha only: the lines k_edge_lines_alpha (in front of k_edge_lines) LEFT to it -- those of many contributing pixels --
    // as a list and its length (EdgePlan::alloc[2]: zero when the plan is built, handed back zeroed by k_edge_gather); NULL =
    // no such pass ran, k_edge_lines takes every line
    int* line_left;
    int* n_left;
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

// The geometry of ONE crossing d0 of an (edge, axis) pair: everything about its two walks (KCU:314-362 outward,
// :417-431 inward) that does not depend on the gradient maps.  p00..p21 = p[num][dim] of KCU:289-294 for that pair;
// owner(d0, d1) returns face_index_map at that line position.  Computed once per crossing (k_edge_scatter) and kept in
// the crossing's record.
enum : uint32_t { XG_ALIVE = 1, XG_DIRPOS = 2, XG_F0 = 4, XG_F1 = 8, XG_OWNER = 16, XG_ORIENTED = 32, XG_IDLE = 64 };
#ifndef D3M_XG_DEAD_SCAN
#define D3M_XG_DEAD_SCAN 6
#endif
constexpr int XG_DEAD_SCAN = D3M_XG_DEAD_SCAN;   // inward ranges of fewer pixels than this are checked for an owned pixel
struct XGeom {
    float d1_cross, q0, q1;     // crossing position along the line; first factors of `dist` (KCU:404 / :409)
    int d1_in;                  // the in-pixel next to the crossing
    int in_from, in_to;         // inward walk, in-pixel .. opposite edge, before clipping to the gradients' extent
    uint32_t bits;              // XG_ALIVE: both pixels next to the crossing lie in the image (KCU:325-328);
                                // XG_DIRPOS: walk direction +1 (KCU:297-308); XG_F0/F1: KCU:403/:408 evaluate the term;
                                // XG_OWNER: the in-pixel belongs to the face, i.e. the outward walk exists (KCU:354);
                                // XG_ORIENTED: every inward pixel lies on the expected side of the crossing
                                // XG_IDLE: alive, but neither walk can contribute (no record is written)
};

// The three quotients of KCU:317 / :427-429 that do not depend on the crossing: the slopes of the lane's edge and of
// the two other edges along the line axis.  One correctly rounded division each per (face, edge, axis) LANE instead of
// two per crossing; the crossing's expressions then multiply the same rounded quotients the reference forms first.
struct EdgeSlopes {
    float s01, s02, s12;
};
__device__ __forceinline__ EdgeSlopes edge_slopes(float p00, float p01, float p10, float p11, float p20, float p21) {
    return EdgeSlopes{(p11 - p01) / (p10 - p00), (p21 - p01) / (p20 - p00), (p11 - p21) / (p10 - p20)};
}

template <class Owner>
__device__ __forceinline__ XGeom crossing_geometry(float p00, float p01, float p10, float p11, float p20, float p21,
                                                   const EdgeSlopes sl, int axis, int fn, int is, int d0, Owner&& owner) {
    XGeom g;
    const int direction = (axis == 0) ? ((p00 < p10) ? -1 : 1) : ((p00 < p10) ? 1 : -1);   // KCU:297-308
    const float fd0 = (float)d0;
    g.d1_cross = sl.s01 * (fd0 - p00) + p01;                                              // KCU:317
    g.d1_in = (0 < direction) ? f2i(floorf(g.d1_cross)) : f2i(ceilf(g.d1_cross));
    const int d1_out = (int)((unsigned)g.d1_in + (unsigned)direction);
    g.bits = (0 < direction) ? XG_DIRPOS : 0u;
    g.q0 = g.q1 = 0.0f;
    g.in_from = g.in_to = 0;
    if (g.d1_in < 0 || is <= g.d1_in || d1_out < 0 || is <= d1_out) { g.d1_in = 0; return g; }   // KCU:325-328
    g.bits |= XG_ALIVE | (p10 != fd0 ? XG_F0 : 0u) | (p00 != fd0 ? XG_F1 : 0u);
    // KCU:404 / :409: first factors of `dist`, correctly rounded quotients (once per crossing; the record keeps ONE of them:
    // geometry_to_record)
    g.q0 = (p10 - p00) / (p10 - fd0);
    g.q1 = (p10 - p00) / (fd0 - p00);
    if (owner(d0, g.d1_in) == fn) g.bits |= XG_OWNER;                                       // KCU:354
    // inward: in-pixel .. opposite edge (KCU:417-431)
    float d0_cross2;
    if ((fd0 - p00) * (fd0 - p20) < 0) d0_cross2 = sl.s02 * (fd0 - p00) + p01;
    else                               d0_cross2 = sl.s12 * (fd0 - p20) + p21;
    const int d1_limit = (0 < direction) ? f2i(ceilf(d0_cross2)) : f2i(floorf(d0_cross2));
    g.in_from = max(min(g.d1_in, d1_limit), 0);
    g.in_to = min(max(g.d1_in, d1_limit), is - 1);
    // all pixels on the expected side of the crossing (see "FACTORED DISTANCE"): decided before clipping
    if ((0 < direction) ? g.in_to == g.d1_in : g.in_from == g.d1_in) g.bits |= XG_ORIENTED;
    // NEITHER WALK CAN CONTRIBUTE: the outward walk only exists when the in-pixel is the face's (KCU:354), and the inward
    // walk only counts pixels of the face (KCU:470).  A short inward range is looked through here (owners only, no
    // gradients needed); a crossing that fails both needs no record, no set-up and no result.
    if (!(g.bits & XG_OWNER) && g.in_to - g.in_from < XG_DEAD_SCAN) {
        bool own = false;
        for (int d1 = g.in_from; d1 <= g.in_to; d1++) own = own || owner(d0, d1) == fn;
        if (!own) g.bits |= XG_IDLE;
    }
    return g;
}

// record <-> geometry: SIXTEEN bytes per crossing (round 6; 32 up to round 5, and the records are the K4 chain's largest
// intermediate: written once by k_edge_scatter, read once or twice by k_edge_lines -- 8.3 M of them per headline step):
//   x  d1_cross
//   y  ONE of the two first factors: 1 / q0 + 1 / q1 = (p10 - d0 + d0 - p00) / (p10 - p00) = 1, so the other one is
//      q / (q - 1).  Kept: the LARGER (>= 2, so q - 1 is exact and the quotient well conditioned: an error of the kept
//      factor reaches the derived one divided by q - 1); when a term is not evaluated at all (KCU:403/:408: an edge end on
//      this very line -- the other factor is then exactly 1) the one that is.  XR_QSEL says which.
//   z  inward walk from | to << 16
//   w  bits (XG_ALIVE .. XG_ORIENTED, XR_QSEL) | face << 7        (the in-pixel follows from d1_cross and the direction,
//      the line is the reader's own, and nobody read the crossing's index)
enum : uint32_t { XR_QSEL = 64 };       // (XG_IDLE's bit: an idle crossing has no record)
__device__ __forceinline__ uint4 geometry_to_record(const XGeom& g, int fn) {
    const bool f0 = (g.bits & XG_F0) != 0, f1 = (g.bits & XG_F1) != 0;
    const bool keep1 = !f0 ? true : (!f1 ? false : g.q1 > g.q0);
    return make_uint4(__float_as_uint(g.d1_cross), __float_as_uint(keep1 ? g.q1 : g.q0),
                      (uint32_t)g.in_from | ((uint32_t)g.in_to << 16),
                      (g.bits & 0x3Fu) | (keep1 ? XR_QSEL : 0u) | ((uint32_t)fn << 7));
}
__device__ __forceinline__ XGeom record_to_geometry(const uint4 r) {
    XGeom g;
    g.d1_cross = __uint_as_float(r.x);
    g.bits = r.w & 0x3Fu;
    const float kept = __uint_as_float(r.y), d = kept - 1.0f;
    // kept / (kept - 1): v_rcp_f32 and one Newton step (the derived factor then carries about the rounding the kept one has)
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float other = kept * __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
    g.q0 = (r.w & XR_QSEL) ? other : kept;
    g.q1 = (r.w & XR_QSEL) ? kept : other;
    g.d1_in = (g.bits & XG_ALIVE) ? ((g.bits & XG_DIRPOS) ? f2i(floorf(g.d1_cross)) : f2i(ceilf(g.d1_cross))) : 0;
    g.in_from = (int)(r.z & 0xFFFFu); g.in_to = (int)(r.z >> 16);
    return g;
}
__device__ __forceinline__ int record_face(const uint4 r) { return (int)(r.w >> 7); }

// Walk `which` (0: outward, 1: inward) of a crossing, clipped to [nz_lo, nz_hi]: the line's pixels with a non-zero
// gradient.  Outside it diff_grad is exactly 0 and KCU:401/:481 skip the pixel (with a masked loss the gradients vanish
// outside the object, and the outward walks, which run to the image BORDER, lose most of their length or disappear).
__device__ __forceinline__ bool geometry_segment(const XGeom& g, int which, int axis, int d0, int is, int nz_lo, int nz_hi,
                                                 Segment& sg) {
    if (!(g.bits & XG_ALIVE) || nz_hi < nz_lo || (which == 0 && !(g.bits & XG_OWNER))) return false;
    const int direction = (g.bits & XG_DIRPOS) ? 1 : -1;
    const int d1_out = g.d1_in + direction;
    sg.axis = axis;
    sg.d0 = d0;
    sg.dir = direction;
    sg.d1_in = g.d1_in;
    sg.d1_cross = g.d1_cross;
    sg.f0 = (g.bits & XG_F0) != 0;
    sg.f1 = (g.bits & XG_F1) != 0;
    sg.q0 = g.q0;
    sg.q1 = g.q1;
    sg.inward = which;
    if (which == 0) {       // out-pixel .. image border (KCU:354-362)
        const int d1_limit = (0 < direction) ? is - 1 : 0;
        sg.from = max(max(min(d1_out, d1_limit), 0), nz_lo);
        sg.to = min(min(max(d1_out, d1_limit), is - 1), nz_hi);
        sg.oriented = 1;
        sg.ref_pos = g.d1_in;
    } else {
        sg.from = max(g.in_from, nz_lo);
        sg.to = min(g.in_to, nz_hi);
        sg.oriented = (g.bits & XG_ORIENTED) != 0;
        sg.ref_pos = d1_out;
    }
    return sg.from <= sg.to;
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

// the same with correctly rounded quotients (the sparse outward walks of the alpha-only mode: a handful of terms per walk,
// so the exact division costs nothing and the sum carries the reference's own rounding, KCU:404-412)
__device__ __forceinline__ void visit_pixel_div(float diff, int d1, float d1_cross, float q0, float q1, bool f0, bool f1,
                                                float two_over_is, float eps, float& g0, float& g1) {
    const float t = (float)d1 - d1_cross;
    float dist0 = q0 * t * two_over_is;
    dist0 = (0 < dist0) ? dist0 + eps : dist0 - eps;
    float dist1 = q1 * t * two_over_is;
    dist1 = (0 < dist1) ? dist1 + eps : dist1 - eps;
    const bool skip = diff <= 0;
    g0 -= (skip || !f0) ? 0.0f : diff / dist0;
    g1 -= (skip || !f1) ? 0.0f : diff / dist1;
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
__device__ __forceinline__ void walk_inline(const EdgeGradArgs& a, size_t view_base, const Segment& sg,
                                            int from, int to, const SegRef& ref, int fn, float two_over_is, float& g0,
                                            float& g1) {
    const float gs = a.go_sign();
    for (int d1 = from; d1 <= to; d1 += 2) {
        const bool two = d1 + 1 <= to;
        const size_t ia = a.pixel(sg.axis, view_base, sg.d0, d1), ib = two ? a.pixel(sg.axis, view_base, sg.d0, d1 + 1) : ia;
        float2 dta = a.rec_dot(ia), dtb = a.rec_dot(ib);
        float4 ga = a.rec_grad(ia), gb = a.rec_grad(ib);
        dta.x *= gs; dtb.x *= gs;
        ga.x *= gs; ga.y *= gs; ga.z *= gs; ga.w *= gs;
        gb.x *= gs; gb.y *= gs; gb.z *= gs; gb.w *= gs;
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

// Which faces own a pixel, as ONE BYTE per face: a face's pixels form a small blob, and only the pixels of it that have
// neither the same face to their left nor above them speak up (one or two plain byte stores per face, where a store per
// covered pixel into a dense int array cost 58 us plus 19 us for zeroing it; one BIT per face needs atomics: 78 us).
__global__ void __launch_bounds__(256) k_mark_visible_bytes(const int32_t* __restrict__ face_index_map,
                                                           unsigned char* __restrict__ marks, int B, int F, int S) {
    // a fixed grid striding over the pixels, four independent pixels in flight per lane: with one pixel per lane the
    // 131 k workgroups of the headline batch are bound by workgroup dispatch, not by memory
    const long n = (long)B * S * S, stride = (long)gridDim.x * 256;
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 4 * stride) {
        int fi[4], left[4], up[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const long i = i0 + k * stride;
            fi[k] = left[k] = up[k] = -1;
            if (i < n) {
                const int x = (int)(i % S), y = (int)((i / S) % S);
                fi[k] = face_index_map[i];
                if (x > 0) left[k] = face_index_map[i - 1];
                if (y > 0) up[k] = face_index_map[i - S];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (fi[k] < 0 || fi[k] == left[k] || fi[k] == up[k]) continue;
            marks[(size_t)((i0 + k * stride) / ((long)S * S)) * F + fi[k]] = 1;
        }
    }
}

// the four faces i0 .. i0+3 (i0 a multiple of 4) of the marks, as a 4-bit mask
__device__ __forceinline__ unsigned visible_nibble(const unsigned char* __restrict__ marks, long i0, long n) {
    if (i0 >= n) return 0u;
    const unsigned m = *(const unsigned*)(marks + i0);          // 0 or 1 per byte
    const unsigned valid = n - i0 >= 4 ? 15u : (1u << (int)(n - i0)) - 1u;      // (bytes past the last face are not the marks')
    return (m | (m >> 7) | (m >> 14) | (m >> 21)) & valid;
}

__global__ void __launch_bounds__(256) k_count_visible(const unsigned char* __restrict__ bits, int* __restrict__ vis_block, long n) {
    __shared__ int s_wave[4];
    const long i0 = (long)blockIdx.x * EG_COMPACT_CHUNK + threadIdx.x * 4;
    const int c = __popc(visible_nibble(bits, i0, n));
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

// The compacted list, and the dense flags (FLAG_HIDDEN / FLAG_VISIBLE for EVERY face: 16 contiguous bytes per lane).
// n is padded to a multiple of 4 by the caller's allocation (eg_align).
__global__ void __launch_bounds__(256) k_compact_visible(const unsigned char* __restrict__ bits, int* __restrict__ flags,
                                                        int* __restrict__ list, const int* __restrict__ vis_block, long n) {
    __shared__ int s_wave[4];
    const long i0 = (long)blockIdx.x * EG_COMPACT_CHUNK + threadIdx.x * 4;
    const unsigned nib = visible_nibble(bits, i0, n);
    bool v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = i0 + k < n && ((nib >> k) & 1u);
    const int c = __popc(nib);
    if (i0 < n) *(int4*)(flags + i0) = make_int4((int)(nib & 1u), (int)((nib >> 1) & 1u), (int)((nib >> 2) & 1u), (int)((nib >> 3) & 1u));
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

// The same list and flags in ONE launch (count -> scan -> compact were three, 57 us of launches and one-workgroup scans on
// the plan's critical path): a workgroup takes 8192 faces, counts its visible ones, and reserves its stretch of the list
// with ONE returning atomic on the list's length (the per-1024-faces cursor tried earlier needed 6272 of them).  The list is ascending inside a
// chunk and the chunks land in arrival order: neighbouring faces stay neighbours, which is all its readers rely on.
// (8192 faces per workgroup = 784 atomics: 42 us in the step, beside the sampling pass; 32768 per workgroup = 196
// atomics: 51 us -- the pass is bound by its 26 MB of flag stores and by what runs beside it, not by the cursor.)
// *count must be zero on entry (cleared with the marks).
#ifndef D3M_EG_COMPACT1_NIBBLES
#define D3M_EG_COMPACT1_NIBBLES 2
#endif
constexpr int EG_COMPACT1_NIB = D3M_EG_COMPACT1_NIBBLES;            // 4-face groups per thread
constexpr int EG_COMPACT1_FACES = 1024 * 4 * EG_COMPACT1_NIB;       // faces per workgroup (8192)
__global__ void __launch_bounds__(1024) k_compact_visible_atomic(const unsigned char* __restrict__ bits, int* __restrict__ flags,
                                                                int* __restrict__ list, int* __restrict__ count, long n) {
    __shared__ int s_wave[16];
    __shared__ int s_base;
    const long i0 = (long)blockIdx.x * EG_COMPACT1_FACES + threadIdx.x * (4 * EG_COMPACT1_NIB);
    unsigned nib[EG_COMPACT1_NIB];
    int c = 0;
#pragma unroll
    for (int h = 0; h < EG_COMPACT1_NIB; h++) {
        nib[h] = visible_nibble(bits, i0 + 4 * h, n);
        c += __popc(nib[h]);
    }
#pragma unroll
    for (int h = 0; h < EG_COMPACT1_NIB; h++)
        if (i0 + 4 * h < n)
            *(int4*)(flags + i0 + 4 * h) = make_int4((int)(nib[h] & 1u), (int)((nib[h] >> 1) & 1u), (int)((nib[h] >> 2) & 1u),
                                                     (int)((nib[h] >> 3) & 1u));
    const int incl = wave_inclusive_scan(c);
    const int wv = threadIdx.x >> 6;
    if (lane_id() == 63) s_wave[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int k = 0; k < 16; k++) { const int t = s_wave[k]; s_wave[k] = total; total += t; }
        s_base = total ? atomicAdd(count, total) : 0;
    }
    __syncthreads();
    int pos = s_base + s_wave[wv] + incl - c;
#pragma unroll
    for (int h = 0; h < EG_COMPACT1_NIB; h++)
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((nib[h] >> k) & 1u) list[pos++] = (int)(i0 + 4 * h + k);
}

// Six lanes per visible face, one per (edge, axis) pair; 42 faces per 256-thread workgroup.
constexpr int EG_FACES_PER_BLOCK = 42;

template <class FS>
__device__ __forceinline__ bool load_face_lane(const FS& fs, const int* __restrict__ visible_list,
                                               const int* __restrict__ n_visible, int blk, int is, int& pos, int& ea, long& gi,
                                               int& bn, int& fn, float& p00, float& p01, float& p10, float& p11,
                                               float& p20, float& p21) {
    const int t = threadIdx.x;
    if (t >= EG_FACES_PER_BLOCK * 6) return false;
    pos = blk * EG_FACES_PER_BLOCK + t / 6;
    ea = t % 6;
    if (pos >= *n_visible) return false;
    gi = visible_list[pos];
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
__device__ __forceinline__ int publish_lanes(const FS& fs, const int* __restrict__ visible_list,
                                             const int* __restrict__ n_visible, int blk, int is, LaneTable& t, bool& on,
                                             int& pos, int& ea, int& n_cross) {
    int bn = 0, fn = 0;
    long gi = 0;
    float p00 = 0, p01 = 0, p10 = 0, p11 = 0, p20 = 0, p21 = 0;
    on = load_face_lane(fs, visible_list, n_visible, blk, is, pos, ea, gi, bn, fn, p00, p01, p10, p11, p20, p21);
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

// ---- the plan: everything about the crossings that depends on GEOMETRY only --------------------------------
// faces + face_index_map decide which faces are visible, where their edges cross the pixel grid, and therefore
// which line every crossing belongs to.  None of that depends on the gradient maps, so the plan can be built as soon
// as the forward pass has produced face_index_map (d3m_edge_plan: on a side stream, beside the sampling pass) and
// the backward pass starts with the line kernel.  One record (16 bytes: geometry_to_record) per crossing, stored in LINE
// order:  xrec[i] = d1_cross, one first factor, inward walk from | to << 16, bits | face << 7
// The results of record i live at results[2*i + {0: outward, 1: inward}] -- LINE order too, so that the line kernel's
// stores are contiguous; xpos[crossing index] = i lets k_edge_gather find them (scattered READS of 16 bytes instead of
// scattered 8-byte writes, which cost the line kernel 0.15 ms of partial-line write traffic).
struct EdgePlan {
    int* visible;        // [B*F]   flags of the faces that own a pixel   \  the caller's d3m_visibility blob
    int* visible_list;   // [B*F]   their compacted indices                > (or one built in the workspace)
    int* n_visible;      // [1]                                            /
    int2* lane_cross;    // [6*B*F] per (visible face, edge, axis) lane: first crossing within its workgroup, count
    int* lane_block;     // [ceil(B*F/42)+1] crossings per k_edge_count workgroup iteration, then (in place) their scan
    int* line_count;     // [B*2*S] crossings on each line (zeroed per call)
    int* line_cursor;    // [B*2*S] records written so far under each line (zeroed per call)
    int2* line_slice;    // [B*2*S] (first record, number of records) of the line's slice of xrec
    int* alloc;          // [0] total crossings (written by the block scan), [1] slice cursor (zeroed per call)
    uint4* xrec;         // [cap]
    float2* results;     // [2 * cap] written by the line kernel (record order) or the overflow kernel (crossing order)
    int* xpos;           // [cap]     record position of every crossing (plan complete only)
    int cap;             // crossings the record / result arrays can hold; the rest is walked by k_edge_overflow
};

// The records are written iff ALL the batch's crossings fit (uniform over a launch): they are then dense and complete,
// xrec[0 .. alloc[0]), grouped by line.  Otherwise (a workspace smaller than the scene needs) k_edge_overflow walks
// every crossing the slow way.

__device__ __forceinline__ bool plan_complete(const EdgePlan& w) { return w.alloc[0] <= w.cap; }

// ---- the lines of one workgroup iteration, counted in LDS ---------------------------------------------------------
// A (face, edge, axis) lane crosses the CONSECUTIVE lines d0_from .. d0_to of its axis, and the 42 neighbouring faces of
// a workgroup iteration meet on a few hundred lines of one view.  So the lanes add +1 / -1 at the ends of their ranges
// in a per-axis LDS array indexed by the line (two LDS atomics per LANE), two waves turn the touched window into
// per-line counts with a running sum, and every line with a count costs the workgroup ONE global atomic -- instead of
// flattening the crossings and merging the counter updates of each wave by key (one global atomic per distinct line
// of every wave: 2.3 M per pass of the headline batch, which bounded both passes -- doubling them took k_edge_count
// from 0.110 to 0.157 ms -- after ~90 instructions of key matching per wave and round).
// Lanes of another view than the iteration's first face (a view boundary inside the 42 faces) update the global
// counters directly.  S <= EG_WINDOW_MAX_S (LDS: 16 bytes per line); beyond it the by-key kernels below are used.
constexpr int EG_WINDOW_MAX_S = 2048;
struct LineWindow {
    int* lds;           // [2][S + 1] per axis: range ends -> counts; all zero between iterations
    int stride;         // S + 1
    // (address arithmetic, not a two-pointer array: indexed by a run-time axis that array lived in scratch memory)
    __device__ __forceinline__ int* cnt(int axis) const { return lds + axis * stride; }
};
__device__ __forceinline__ void window_open(LineWindow& win, int* lds, int is) {
    win.lds = lds; win.stride = is + 1;
    for (int k = threadIdx.x; k < 2 * (is + 1); k += blockDim.x) lds[k] = 0;
}
// the lanes' range ends; returns after the barrier that makes them (and the window bounds) visible.  s_lo / s_hi:
// shared ints [2] each.
__device__ __forceinline__ void window_ranges(const LineWindow& win, const LaneTable& t, int is, bool on, int n_cross, int view,
                                              int* s_lo, int* s_hi) {
    if (threadIdx.x < 2) { s_lo[threadIdx.x] = is; s_hi[threadIdx.x] = -1; }
    __syncthreads();
    const int l = threadIdx.x;
    if (on && n_cross > 0 && (t.bn_axis[l] >> 1) == view) {
        const int axis = t.bn_axis[l] & 1, from = t.d0_from[l], to1 = from + n_cross;    // to + 1 <= is
        atomicAdd(&win.cnt(axis)[from], 1);
        atomicAdd(&win.cnt(axis)[to1], -1);
        atomicMin(&s_lo[axis], from);
        atomicMax(&s_hi[axis], to1);
    }
    __syncthreads();
}

// ---- 1. per workgroup: how many crossings; per line: how many records it will receive ------------------------
template <class FS>
__global__ void __launch_bounds__(256) k_edge_count_window(FS fs, int is, EdgePlan w) {
    extern __shared__ int s_window[];
    __shared__ LaneTable t;
    __shared__ int s_lo[2], s_hi[2];
    LineWindow win;
    window_open(win, s_window, is);
    const int n_blocks = (*w.n_visible + EG_FACES_PER_BLOCK - 1) / EG_FACES_PER_BLOCK;
    const XcdOrder xo(n_blocks);
    const int wv = threadIdx.x >> 6, lane = lane_id();
    for (int i = blockIdx.x; xo.more(i); i += gridDim.x) {              // fixed grid, uniform trip count per workgroup
        const int blk = xo.unit(i);
        if (blk >= n_blocks) continue;
        bool on;
        int pos = 0, ea = 0, n_cross = 0;
        const int total = publish_lanes(fs, w.visible_list, w.n_visible, blk, is, t, on, pos, ea, n_cross);
        // every lane of a listed face gets its record (n_cross is 0 for a lane that is not `on`)
        if (threadIdx.x < EG_FACES_PER_BLOCK * 6 && blk * EG_FACES_PER_BLOCK + (int)threadIdx.x / 6 < *w.n_visible) {
            const size_t lane6 = (size_t)blk * EG_FACES_PER_BLOCK * 6 + threadIdx.x;
            w.lane_cross[lane6] = make_int2(t.pre[threadIdx.x], n_cross);
        }
        if (threadIdx.x == 0) w.lane_block[blk] = total;
        const int view = t.bn_axis[0] >> 1;
        window_ranges(win, t, is, on, n_cross, view, s_lo, s_hi);
        if (wv < 2) {                                       // wave `axis`: running sum over the window, 64 lines per round
            const int axis = wv;
            int run = 0;
            for (int d0 = s_lo[axis] + lane; d0 - lane <= s_hi[axis]; d0 += 64) {
                const bool in = d0 <= s_hi[axis];
                const int v = in ? win.cnt(axis)[d0] : 0;
                const int incl = wave_inclusive_scan(v) + run;
                run = __builtin_amdgcn_readlane(incl, 63);
                if (in) {
                    win.cnt(axis)[d0] = 0;
                    if (incl > 0) atomicAdd(&w.line_count[((size_t)view * 2 + axis) * is + d0], incl);
                }
            }
        }
        // a lane of another view (a view boundary inside the iteration): straight to the global counters
        if (on && n_cross > 0 && (t.bn_axis[threadIdx.x] >> 1) != view) {
            const int l = threadIdx.x;
            const size_t line0 = ((size_t)(t.bn_axis[l] >> 1) * 2 + (t.bn_axis[l] & 1)) * is + t.d0_from[l];
            for (int k = 0; k < n_cross; k++) atomicAdd(&w.line_count[line0 + k], 1);
        }
        __syncthreads();                                    // the tables are rewritten by the next iteration
    }
}

// ---- 1. per workgroup: how many crossings; per line: how many records it will receive ------------------------
template <class FS>
__global__ void __launch_bounds__(256) k_edge_count(FS fs, int is, EdgePlan w) {
    __shared__ LaneTable t;
    const int n_blocks = (*w.n_visible + EG_FACES_PER_BLOCK - 1) / EG_FACES_PER_BLOCK;
    const XcdOrder xo(n_blocks);
    for (int i = blockIdx.x; xo.more(i); i += gridDim.x) {              // fixed grid, uniform trip count per workgroup
        const int blk = xo.unit(i);
        if (blk >= n_blocks) continue;
        bool on;
        int pos = 0, ea = 0, n_cross = 0;
        const int total = publish_lanes(fs, w.visible_list, w.n_visible, blk, is, t, on, pos, ea, n_cross);
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
            // neighbouring crossings fall on the same lines: one atomic per distinct line of the wave (uniform call site)
            const unsigned long long same = wave_match_any((uint32_t)line, c < total);
            if (c < total && lane_id() == __builtin_ctzll(same)) atomicAdd(&w.line_count[line], __popcll(same));
        }
        __syncthreads();                                    // the table is rewritten by the next iteration
    }
}

// ---- 2. order-free range allocation: slice start for counts[i] (one atomic per 256 lines) ---------------------
__global__ void __launch_bounds__(256) k_alloc_ranges(const int* __restrict__ counts, int2* __restrict__ line_slice,
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
    if (i < n) line_slice[i] = make_int2(s_base + s_wave[wv] + incl - c, c);
}

// ---- 2'. both allocations of the plan in ONE launch ---------------------------------------------------------------
// The count pass leaves two arrays of counts: crossings per workgroup iteration (lane_block: the scatter and gather
// passes need each iteration's first crossing) and crossings per line (line_count: the scatter pass needs each line's
// slice of the records).  Neither needs its slices in index order -- only disjoint --, so both are allocated order-free:
// 256 entries per workgroup, one returning atomic each (a few dozen per launch on one address, not thousands).  Round 3
// ran an ordered single-workgroup scan for the first (k_scan_small, 11 us of one workgroup on an otherwise idle chip) and
// k_alloc_ranges for the second; with the step's kernels on one stream that was 15 us of its critical path, now 5.
// Workgroups [0, blocks_a) take lane_block (in place: count -> first crossing; the grid is sized for EVERY face being
// visible, the workgroups past ceil(*n_visible / faces per iteration) leave at once), the rest line_count -> line_slice.
// cursor_a ends as the number of crossings (EdgePlan::alloc[0]: plan_complete()).
__global__ void __launch_bounds__(256) k_alloc_plan(int* __restrict__ lane_block, const int* __restrict__ n_visible, int per_block,
                                                   int* __restrict__ cursor_a, int blocks_a,
                                                   const int* __restrict__ line_count, int2* __restrict__ line_slice,
                                                   int* __restrict__ cursor_b, long n_lines) {
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const bool first = (int)blockIdx.x < blocks_a;
    const long n = first ? (long)((*n_visible + per_block - 1) / per_block) : n_lines;
    const long i = (long)(first ? blockIdx.x : blockIdx.x - blocks_a) * 256 + threadIdx.x;
    if (i - threadIdx.x >= n) return;                       // (uniform per workgroup)
    const int c = i < n ? (first ? lane_block[i] : line_count[i]) : 0;
    const int incl = wave_inclusive_scan(c);
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t0 = s_wave[0], t1 = s_wave[1], t2 = s_wave[2], t3 = s_wave[3];
        s_base = atomicAdd(first ? cursor_a : cursor_b, t0 + t1 + t2 + t3);
        s_wave[0] = 0; s_wave[1] = t0; s_wave[2] = t0 + t1; s_wave[3] = t0 + t1 + t2;
    }
    __syncthreads();
    if (i < n) {
        const int at = s_base + s_wave[wv] + incl - c;
        if (first) lane_block[i] = at;
        else line_slice[i] = make_int2(at, c);
    }
}

// ---- 3. the crossings' records, written in line order -------------------------------------------------------
// Same flattening as the count pass.  The lanes of a wave that share a line take consecutive places under that
// line's cursor with ONE atomic per distinct line.
// `parts` (>= 1): the rounds of one block of faces are dealt to that many workgroups -- a coarse mesh has few blocks (722
// triangles: 35) of faces with dozens of crossings per edge, i.e. a few workgroups with thirty rounds of dependent look-ups
// each while the rest of the chip idles (125 us); the host sets it from the batch's face count (1 for ordinary meshes).
template <class FS>
__global__ void __launch_bounds__(256) k_edge_scatter(FS fs, const int32_t* __restrict__ face_index_map, int is, EdgePlan w,
                                                     int parts) {
    __shared__ LaneTable t;
    __shared__ float s_slope[3][256];
    const int n_blocks = (*w.n_visible + EG_FACES_PER_BLOCK - 1) / EG_FACES_PER_BLOCK;
    const int n_units = n_blocks * parts;
    const XcdOrder xo(n_units);
    for (int i = blockIdx.x; xo.more(i); i += gridDim.x) {
        const int unit = xo.unit(i);
        if (unit >= n_units) continue;
        const int blk = unit / parts, part = unit - blk * parts;
        bool on;
        int pos = 0, ea = 0, n_cross = 0;
        const int total = publish_lanes(fs, w.visible_list, w.n_visible, blk, is, t, on, pos, ea, n_cross);
        const long cbase = w.lane_block[blk];               // scanned: first crossing of this workgroup
        {   // the lane's slopes, once (see EdgeSlopes)
            const int l = threadIdx.x;
            const EdgeSlopes sl = edge_slopes(t.p[0][l], t.p[1][l], t.p[2][l], t.p[3][l], t.p[4][l], t.p[5][l]);
            s_slope[0][l] = sl.s01; s_slope[1][l] = sl.s02; s_slope[2][l] = sl.s12;
        }
        __syncthreads();
        for (int c0 = part * 256; c0 < total; c0 += parts * 256) {
            const int c = c0 + threadIdx.x;
            const bool active = c < total;
            int l = 0;
            size_t line = 0;
            int2 slice = make_int2(0, 0);
            if (active) {
                l = crossing_lane(t, c);
                line = ((size_t)(t.bn_axis[l] >> 1) * 2 + (t.bn_axis[l] & 1)) * is + t.d0_from[l] + (c - t.pre[l]);
                slice = w.line_slice[line];
            }
            // the geometry first: a crossing that is outside the image or whose walks cannot contribute takes no place
            XGeom g;
            bool wants = false;
            if (active && plan_complete(w)) {
                const int bn = t.bn_axis[l] >> 1, axis = t.bn_axis[l] & 1, d0 = t.d0_from[l] + (c - t.pre[l]);
                const int32_t* view = face_index_map + (size_t)bn * is * is;
                g = crossing_geometry(t.p[0][l], t.p[1][l], t.p[2][l], t.p[3][l], t.p[4][l], t.p[5][l],
                                      EdgeSlopes{s_slope[0][l], s_slope[1][l], s_slope[2][l]}, axis, t.fn[l], is, d0,
                                      [&](int e0, int e1) { return view[axis ? (size_t)e0 * is + e1 : (size_t)e1 * is + e0]; });
                wants = (g.bits & XG_ALIVE) && !(g.bits & XG_IDLE);
                if (!wants) w.xpos[cbase + c] = -1;
            }
            const unsigned long long same = wave_match_any((uint32_t)line, wants);      // uniform call site
            const int leader = wants ? __builtin_ctzll(same) : 0, n = wants ? __popcll(same) : 0;
            int cursor_base = 0;
            if (wants && lane_id() == leader) cursor_base = atomicAdd(&w.line_cursor[line], n);
            const int in_line = __shfl(cursor_base, leader, 64) + mask_rank(same);
            if (wants) {
                w.xrec[(size_t)slice.x + in_line] = geometry_to_record(g, t.fn[l]);
                w.xpos[cbase + c] = slice.x + in_line;
            }
        }
        __syncthreads();
    }
}

// ---- 4. crossings that have no record: walked by the owning thread, straight from global memory ---------------
// Only when the workspace is too small for the scene (default capacity: eg_default_crossings, two crossings per face of the
// batch or three per raster pixel).
// Leaves at once otherwise.  Results: the crossing's own slots when it has them, else the lane's overflow sum.
template <class FS>
__global__ void __launch_bounds__(256) k_edge_overflow(FS fs, EdgeGradArgs a, EdgePlan w, float2* __restrict__ lane_partial) {
    if (plan_complete(w)) return;                           // every crossing has a record: k_edge_lines did it all (uniform exit)
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
        const int total = publish_lanes(fs, w.visible_list, w.n_visible, blk, is, t, on, pos, ea, n_cross);
        const long cbase = w.lane_block[blk];
        for (int c = threadIdx.x; c < total; c += 256) {
            const int l = crossing_lane(t, c);
            const int d0 = t.d0_from[l] + (c - t.pre[l]);
            const int bn = t.bn_axis[l] >> 1, axis = t.bn_axis[l] & 1, fn = t.fn[l];
            const size_t line = ((size_t)bn * 2 + axis) * is + d0;
            const size_t base = (size_t)bn * is * is;
            const XGeom xg = crossing_geometry(t.p[0][l], t.p[1][l], t.p[2][l], t.p[3][l], t.p[4][l], t.p[5][l],
                                               edge_slopes(t.p[0][l], t.p[1][l], t.p[2][l], t.p[3][l], t.p[4][l], t.p[5][l]),
                                               axis, fn, is, d0,
                                               [&](int e0, int e1) { return __float_as_int(a.rec_dot(a.pixel(axis, base, e0, e1)).y); });
#pragma unroll
            for (int which = 0; which < 2; which++) {
                float g0 = 0, g1 = 0;
                Segment sg;
                if (geometry_segment(xg, which, axis, d0, is, a.extent_lo(line), a.extent_hi(line), sg)) {
                    const SegRef ref = load_ref(a, axis, base, sg.d0, sg.ref_pos);
                    walk_inline(a, base, sg, sg.from, sg.to, ref, fn, two_over_is, g0, g1);
                }
                if (cbase + c < (long)w.cap) {
                    w.results[2 * (cbase + c) + which] = make_float2(g0, g1);
                } else if (g0 != 0 || g1 != 0) {
                    const size_t lane_id6 = ((size_t)blk * EG_FACES_PER_BLOCK) * 6 + l;
                    atomicAdd(&lane_partial[lane_id6].x, g0);
                    atomicAdd(&lane_partial[lane_id6].y, g1);
                }
            }
        }
        __syncthreads();
    }
}

// Lanes that share one long segment (16, 8 or 4: all inside a DPP row).  The per-segment work -- record decode, the
// trip count, the DPP sum, the result -- is per-LANE vector work shared by all the segments of a wave, so fewer lanes
// per segment means fewer instructions per segment: 16 -> 8 -> 4 lanes took the walk from 0.51 to 0.45 to 0.44 ms on
// the headline step (263 M wave-instructions, 86 % VALU-issue-bound, at 16).
#ifndef D3M_EG_ROW
#define D3M_EG_ROW 4
#endif
constexpr int EG_ROW = D3M_EG_ROW;
constexpr int EG_SEG_PER_WAVE = 64 / EG_ROW;   // segments walked concurrently by one wave
// per workgroup of `waves` waves: 64 * waves crossing records set up at a time (one thread per record), EG_QUEUE = 64 * waves
// long segments queued in LDS before they are walked (one sort key per thread)
constexpr size_t eg_line_static_lds(int waves) { return (size_t)waves * 64 * (EG_ITEM_DW * 4 + 2) + 1024; }   // queue + order + counters
constexpr size_t EG_LINE_STATIC_LDS = eg_line_static_lds(EG_LINE_WAVES);

// ---- 5. one workgroup per (view, axis, line): set up the line's crossings, walk their segments ------------------
// The line's per-pixel records are staged in LDS once (only its non-zero-gradient extent).  Then, a pass of
// EG_LINE_THREADS crossing records at a time:
//   SET-UP   one thread per record: the two segments of KCU:314-362 (outward) / :417-431 (inward) with the owner test
//            answered from LDS, clipped to the extent; an empty one stores its zero; the inward one (mean length 2
//            pixels) is walked by the thread itself from LDS unless it is long AND lies on one side of the crossing; the
//            outward one, and such a long inward one, are queued in LDS as 32-byte items;
//   WALK     the queued segments ordered by length (counting sort on length / 16, longest first), sixteen per wave,
//            four lanes each, FACTORED DISTANCE (below).
// Item (8 dwords): 0 bits = inward[0] f0[1] f1[2] fix_at_from[3] fix_at_to[4] s_t>0[5] | fn << 6; 1 inv0;
// 2 from | to<<16; 3 result slot; 4 d1_cross; 5 inv1; 6 position of the reference pixel; 7 unused.
//
// REFERENCE VALUES.  The pixel whose (alpha, rgb) a walk compares against (KCU:365-381 / :436-452) lies on the walk's
// own line, next to the crossing -- so the line's VALUES are staged in LDS as well (16 bytes per pixel, the whole
// line: that pixel need not lie inside the non-zero-gradient extent) and a segment's set-up is one memory round trip
// (its record), not two (record -> a 16-byte gather from the maps, 25 M of them per headline step).
//
// FACTORED DISTANCE.  Along one queued segment t = d1 - d1_cross keeps its sign s_t (outward: the walk direction;
// inward: the opposite), so KCU:404-405's  dist = q*t*(2/is) +- eps  is  qc*(t + u)  with qc = q*2/is and the
// per-item constant u = s_t*eps/|qc|, and the walk's sum  -sum diff/dist  becomes  (-1/qc) * sum diff/(t + u):
// per pixel one packed add, two v_rcp and one packed fma for both vertices; inv = -1/qc is applied once per item.
// |t + u| >= |u| > 0, so no quotient is infinite.  The one pixel where t == 0 (an inward walk starting exactly on
// an integer crossing: the reference's `0 < dist` is false there, i.e. -eps whatever s_t says) is corrected after
// the loop (fix_at_*; only if that pixel survived the clip to the line's non-zero extent).
//
// PAD: the lanes of a row that has finished -- the segments of a wave advance in lock step with the longest -- keep
// reading, up to entry 2*S + 16 of the image arrays; what they read is never used (their lanes are masked), it only
// has to lie inside the workgroup's LDS allocation.  The arrays are therefore laid out pairs | gradients | values |
// tail: an overrun of the 8-byte pairs ends in the gradients, one of the 16-byte gradients in the values and the
// tail, and the per-iteration address clamp disappears at the price of 16*16 bytes, not of a second image.  (Up to round 3
// rasters above 28 KB of line image walked a second, clamped-index form of the loop -- two reciprocals and a select per
// visit: the same LDS, and on the 1024^2 configuration 1.66 ms where this form takes 1.46.  Removed.)
// REGISTERS: at most 64 VGPRs (eight waves per SIMD), i.e. FOUR workgroups per CU where the line's LDS image leaves room
// for four (S <= ~600): the kernel's phases -- stage, set up, order, walk -- are separated by barriers and overlap only
// ACROSS workgroups.  Round 3 forced the bound onto a 70-register allocation, and the compiler met it with 24 bytes of
// scratch per lane (335 MB of spill stores per headline launch); since round 4 the walk re-reads what it needs after its
// loop from the queue item instead of carrying it, the allocation is 62-64 registers without the bound, and
// tests/test_host_logic.py fails on any scratch in the library.
template <bool USE_RGB, bool USE_ALPHA, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 8) k_edge_lines(EdgeGradArgs a, EdgePlan w, float2* __restrict__ lane_partial) {
    // WAVES per workgroup: 8 where four such workgroups fit a CU beside their lines' LDS images (S <= ~600), 16 above that
    // (two workgroups = the same eight waves per SIMD, where 8-wave workgroups would leave the CU at four or six)
    constexpr int EG_LINE_THREADS = WAVES * 64, EG_QUEUE = EG_LINE_THREADS, EG_LINE_WAVES = WAVES;
    extern __shared__ __attribute__((aligned(16))) float s_line[];
    if (!plan_complete(w)) {          // no records at all: k_edge_overflow walks every crossing; zero the sums it adds to
        const long n = (long)*w.n_visible * 6;
        for (long i = (long)blockIdx.x * EG_LINE_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * EG_LINE_THREADS)
            lane_partial[i] = make_float2(0.0f, 0.0f);
        return;
    }
    __shared__ __attribute__((aligned(16))) uint32_t s_items[EG_LINE_THREADS * EG_ITEM_DW];
    __shared__ int s_hist[33];
    __shared__ unsigned short s_order[EG_LINE_THREADS];
    __shared__ int s_nitems, s_pass[2];
    // SPARSE OUTWARD WALKS (alpha only: render_silhouettes and every return_rgb == 0 call).  An outward walk only exists
    // when its in-pixel is the face's own (KCU:354-355), i.e. covered: alpha_in = 1, and its terms are
    //     diff_grad(d) = (alpha(d) - 1) * grad_alpha(d)                                          (KCU:385-387)
    // -- independent of the crossing, and kept only where > 0 (KCU:401): at UNCOVERED pixels whose alpha gradient is
    // negative, the band where the silhouette should grow.  So the line's pixels that can contribute to ANY outward walk
    // are compacted once, in ascending order (s_nz: positions), and the thread that sets a crossing up adds its outward
    // walk's few terms on the spot -- binary search for the segment's first entry, then entries up to its end -- instead of
    // queueing a segment that visits every pixel between the crossing and the image border (a 32-view silhouette step walked
    // 1.3 G pixels per launch to find the ~2 % of them that count).  Exact: the skipped pixels are exactly those the
    // reference's `diff_grad <= 0` skips; the kept terms use its own expression with correctly rounded quotients
    // (visit_pixel_div).  A line with more such pixels than a.sparse_max (or than the list holds), and a walk whose
    // reference alpha is not exactly 1 (a caller's own alpha map), take the dense forms below.
    constexpr int NZ_CAP = USE_RGB ? 1 : WAVES * 64;
    __shared__ unsigned short s_nz[NZ_CAP];
    __shared__ int s_wcnt[USE_RGB ? 1 : WAVES];
    const int is = a.S;
    // XCD-aware order: consecutive workgroups are dealt round-robin to the 8 XCDs, and neighbouring COLUMN lines share
    // their pixels' cache lines (a 64-byte line holds the records of 4 neighbouring columns, the pairs of 8, the alphas
    // of 16) -- dealt to different XCDs every one of them is fetched into up to 8 L2s (1.74 GB of HBM-side traffic per
    // launch for 0.3 GB of maps).  XCD x takes the contiguous lines [x*per, (x+1)*per).
    auto do_line = [&](const size_t line) {                   // (b*2 + axis)*S + d0
    const int n_x = __builtin_amdgcn_readfirstlane(w.line_cursor[line]);      // crossing records under this line
    if (n_x <= 0) return;                                     // nothing to do (uniform exit)
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // the lane number where it is needed, from the thread index behind an opaque copy: a `lane` kept live through the set-up
    // passes is one of the registers those do not have
    auto lane_here = [] { int t = (int)threadIdx.x; asm volatile("" : "+v"(t)); return t & 63; };
    const int d0 = (int)(line % is);
    const int axis = (int)((line / is) & 1);
    const size_t bn = line / ((size_t)2 * is);
    const size_t view_base = bn * is * is;
    const int x_first = __builtin_amdgcn_readfirstlane(w.line_slice[line].x);
    const uint4* xrec = w.xrec + (size_t)x_first;
    const float two_over_is = 2.0f / (float)is;
    // only the line's non-zero-gradient extent is staged: every segment is clipped to it (geometry_segment)
    const int p_lo = __builtin_amdgcn_readfirstlane(a.extent_lo(line));
    const int p_hi = __builtin_amdgcn_readfirstlane(a.extent_hi(line));
    // ONE THREAD PER RECORD (round 4; before: one per (record, outward | inward), i.e. every record decoded by two
    // threads and EG_LINE_THREADS / 2 records per pass -- and the headline mesh puts a median of 260 records on a line,
    // four more than a pass of the 8-wave form took, so half the lines paid a second pass (three barriers, a round of
    // record loads) for a handful of crossings).  A thread decodes its record, takes the crossing's inward walk (mean
    // length 2 pixels: walked on the spot; the rare long one is queued) and classifies the outward one (96 % of the
    // records have one, nearly all of them long: queued).  Nothing but two wave masks stays live across the queue's
    // barriers: the threads that queue a segment read their record AGAIN behind them (16 bytes from the L2) and build the
    // item there -- with the geometry of both walks carried across, the 64-register budget spilled 100 bytes per lane
    // (and with the first pass's record requested before the line image is staged, as the two-thread form did, 20: the
    // record's round trip now follows the staging's; still the faster form, 0.50 -> 0.48 ms on the headline step).
    auto load_record = [&](int ci) {
        uint4 r = make_uint4(0, 0, 0, 0);   // bits 0: not alive -> neither walk exists
        if (ci < n_x) r = xrec[(uint32_t)ci];                     // (uniform base + 32-bit offset)
        return r;
    };
    // LDS image of the line: per pixel the float4 of gradients (alpha, r, g, b) and the pair (T/2, owner) with
    // T = sum value*grad of the pixel itself, so diff = T - <reference, gradients>: ds_read_b128 + ds_read_b64 per
    // visited pixel, contiguous within a segment's lane group.  T is kept halved (exact) because the two packed fma
    // below start BOTH halves of the sum from it.
    float2* s_df = (float2*)s_line;                            // [is] (T/2, owner)
    float4* s_grd = (float4*)(s_df + is + (is & 1));           // [is] gradients (alpha, r, g, b)
    float4* s_val = s_grd + is;                                // [is] values (alpha, r, g, b), then the PAD tail
    const float go_sign = a.go_sign();
    for (int p = (int)threadIdx.x; p < is; p += EG_LINE_THREADS) {
        const size_t pi = a.pixel(axis, view_base, d0, p);
        float4 v = make_float4(0, 0, 0, 0);
        if (USE_ALPHA) v.x = a.alpha_map[pi];
        if (USE_RGB) { v.y = a.rgb_map[3 * pi]; v.z = a.rgb_map[3 * pi + 1]; v.w = a.rgb_map[3 * pi + 2]; }
        s_val[p] = v;
    }
    for (int p = p_lo + (int)threadIdx.x; p <= p_hi; p += EG_LINE_THREADS) {
        const size_t pi = a.pixel(axis, view_base, d0, p);
        float4 g;
        float2 d;
        if (!USE_RGB && a.direct()) {
            const float ga = a.direct_ga_at(bn, axis, d0, p, pi);
            g = make_float4(ga, 0.0f, 0.0f, 0.0f);
            d = make_float2(a.alpha_map[pi] * ga, __int_as_float(a.fi_direct[pi]));
        } else {
            g = a.grad[pi];
            d = a.dot[pi];
        }
        g.x *= go_sign; g.y *= go_sign; g.z *= go_sign; g.w *= go_sign;
        s_grd[p] = g;
        s_df[p] = make_float2(0.5f * go_sign * d.x, d.y);
    }
    typedef float v2f __attribute__((ext_vector_type(2)));
    // diff_grad of one pixel (KCU:385-396 / :473-479 regrouped) from its LDS records: two packed fma + one add
    auto diff_of = [](const float4 g, const float2 d, const v2f nref_ar, const v2f nref_gb) {
        v2f p = {d.x, d.x};
        p = __builtin_elementwise_fma(v2f{g.x, g.y}, nref_ar, p);
        p = __builtin_elementwise_fma(v2f{g.z, g.w}, nref_gb, p);
        return p.x + p.y;
    };
    // the same sum in BOTH halves of a pair (one packed add of the pair with itself swapped)
    auto diff_of2 = [](const float4 g, const float2 d, const v2f nref_ar, const v2f nref_gb) {
        v2f p = {d.x, d.x};
        p = __builtin_elementwise_fma(v2f{g.x, g.y}, nref_ar, p);
        p = __builtin_elementwise_fma(v2f{g.z, g.w}, nref_gb, p);
        return p + __builtin_shufflevector(p, p, 1, 0);
    };
    // both quotients of a pixel from ONE v_rcp_f32 (a quarter-rate instruction): 1/a = b * 1/(a*b).  Two more roundings
    // (~1.5 ulp); |a*b| stays far inside the float range for distances of at most S pixels and |u| >= eps*S/2^k
#ifdef D3M_EG_TWO_RCP
    auto rcp2 = [](const v2f den) { return v2f{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)}; };
#else
    auto rcp2 = [](const v2f den) {
        const float r = __builtin_amdgcn_rcpf(den.x * den.y);
        return v2f{r, r} * v2f{den.y, den.x};
    };
#endif
    // acc += d2 * r in the lanes of `keep` only, as an EXEC-masked instruction: two scalar instructions around one
    // vector instruction, where the compiler's select form costs two more vector instructions per visit (-5 % kernel)
    // and its branch form a taken branch.  Never by multiplying with 0: a lane outside its segment may sit exactly on
    // den == 0 (1/0 = inf, 0 * inf = NaN).
    auto fma_where = [](v2f& acc, const v2f d2, const v2f r, const unsigned long long keep) {
#ifdef D3M_EG_SELECT_FMA
        if (__builtin_amdgcn_inverse_ballot_w64(keep)) acc = __builtin_elementwise_fma(d2, r, acc);
#else
        unsigned long long saved;
        asm volatile("s_and_saveexec_b64 %[sv], %[kp]\n\t"
                     "v_pk_fma_f32 %[a], %[d], %[rr], %[a]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [a] "+v"(acc), [sv] "=&s"(saved)
                     : [kp] "s"(keep), [d] "v"(d2), [rr] "v"(r)
                     : "scc");
#endif
    };
    // The queue of long segments (EG_QUEUE items in LDS) is filled by as many set-up passes as it takes -- a line of the
    // headline mesh has ~380 crossings, i.e. two passes, of which ~240 segments are long -- and walked when the next
    // pass might not fit, and at the end: the walk wants MANY segments at a time (sixteen per wave, ordered by length).
    auto walk_queue = [&]() {
        const int nc = s_nitems;                              // (the caller's barrier made it visible)
        if (nc > 0) {
            // ---- order the queued segments by length: the segments of a wave advance in lock step, so they should be
            // about equally long (unsorted, a group ran at 64 % lane efficiency) -----------------------------------------
            int key = -1, rank_in_key = 0;
            if ((int)threadIdx.x < nc) {
                const uint32_t ft = s_items[(size_t)threadIdx.x * EG_ITEM_DW + 2];
                key = 31 - min(((int)(ft >> 16) - (int)(ft & 0xFFFF) + 1) >> 4, 31);
                rank_in_key = atomicAdd(&s_hist[key], 1);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int run = 0;
                for (int k = 0; k < 32; k++) { const int c = s_hist[k]; s_hist[k] = run; run += c; }
            }
            __syncthreads();
            if (key >= 0) s_order[s_hist[key] + rank_in_key] = (unsigned short)threadIdx.x;
            __syncthreads();
            // ---- walk: EG_SEG_PER_WAVE segments per wave, EG_ROW lanes each ------------------------------------------
            constexpr int STRIDE = EG_LINE_WAVES * EG_SEG_PER_WAVE;
            const int row = lane_here() / EG_ROW, rl = lane_here() % EG_ROW;      // (see lane_here)
            for (int base = wv * EG_SEG_PER_WAVE; base < nc; base += STRIDE) {
                const bool have = base + row < nc;
                const int it = have ? s_order[base + row] : s_order[base];
                const uint4* q = (const uint4*)(s_items + (size_t)it * EG_ITEM_DW);
                const uint4 q0v = q[0], q1v = q[1];
                const float4 rv = s_val[q1v.z];
                const uint32_t bits = q0v.x & 63u;
                const int from = (int)(q0v.z & 0xFFFF), to = have ? (int)(q0v.z >> 16) : -1, fn = (int)(q0v.x >> 6);
                const float d1_cross = __uint_as_float(q1v.x);
                const float inv0 = __uint_as_float(q0v.y), inv1 = __uint_as_float(q1v.y);
                const float s_t = (bits & 32u) ? 1.0f : -1.0f;
                const v2f u = {s_t * (a.eps * fabsf(inv0)), s_t * (a.eps * fabsf(inv1))};      // see FACTORED DISTANCE
                const v2f nref_ar = {USE_ALPHA ? -rv.x : 0.0f, USE_RGB ? -rv.y : 0.0f};
                const v2f nref_gb = {USE_RGB ? -rv.z : 0.0f, USE_RGB ? -rv.w : 0.0f};
                // A pixel counts if it lies in the segment, its diff_grad is not <= 0 (KCU:401/:481; NaN passes, as in the
                // reference) and -- inward walks only -- it belongs to the face (KCU:470).  The three conditions are combined
                // as wave masks on the scalar unit; the vector unit only issues the compares and ONE select.
                const unsigned long long m_outward = ~__builtin_amdgcn_ballot_w64((bits & 1u) != 0);
                // the rows advance together: as many EG_ROW-pixel steps as the longest needs (a scalar trip count)
                const int len = to - from + 1;
                int max_len = __builtin_amdgcn_readlane(len, 0);
    #pragma unroll
                for (int r = 1; r < EG_SEG_PER_WAVE; r++) max_len = max(max_len, __builtin_amdgcn_readlane(len, r * EG_ROW));
                const int n_iter = (max_len + EG_ROW - 1) / EG_ROW;
                float t = (float)(from + rl) - d1_cross;            // t = d1 - d1_cross advances by exact steps
                v2f acc = {0.0f, 0.0f};
                {
                    const float4* pg = s_grd + from + rl;
                    const float2* pd = s_df + from + rl;
                    const float4* pg_to = s_grd + to;
                    v2f den = u + t;                                // advances by exact steps of EG_ROW as well
                    if (m_outward == ~0ull) {                       // outward walks only (the common case): no owner test
                        // two steps per round, each step's LDS reads issued a step ahead of its arithmetic (a wave's step is
                        // a round trip to LDS and then ~20 dependent vector instructions; with the reads in flight across the
                        // other step's arithmetic the wave waits for neither).  An odd trip count's extra step is masked.
                        float4 ga = pg[0];
                        float2 da = pd[0];
                        const float4* pg_to1 = pg_to - EG_ROW;
                        for (int k = 0; k < n_iter; k += 2) {
                            const float4 gb = pg[EG_ROW];
                            const float2 db = pd[EG_ROW];
                            {
                                const v2f d2 = diff_of2(ga, da, nref_ar, nref_gb);
                                const unsigned long long keep = __builtin_amdgcn_ballot_w64(pg <= pg_to) &
                                                                __builtin_amdgcn_ballot_w64(!(d2.x <= 0));
                                fma_where(acc, d2, rcp2(den), keep);
                                den += (float)EG_ROW;
                            }
                            ga = pg[2 * EG_ROW];
                            da = pd[2 * EG_ROW];
                            {
                                const v2f d2 = diff_of2(gb, db, nref_ar, nref_gb);
                                const unsigned long long keep = __builtin_amdgcn_ballot_w64(pg <= pg_to1) &
                                                                __builtin_amdgcn_ballot_w64(!(d2.x <= 0));
                                fma_where(acc, d2, rcp2(den), keep);
                                den += (float)EG_ROW;
                            }
                            pg += 2 * EG_ROW;
                            pd += 2 * EG_ROW;
                        }
                    } else {
                        for (int k = 0; k < n_iter; k++) {
                            const float4 g = *pg;
                            const float2 d = *pd;
                            const v2f d2 = diff_of2(g, d, nref_ar, nref_gb);
                            const unsigned long long keep = __builtin_amdgcn_ballot_w64(pg <= pg_to) &
                                                            __builtin_amdgcn_ballot_w64(!(d2.x <= 0)) &
                                                            (__builtin_amdgcn_ballot_w64(__float_as_int(d.y) == fn) | m_outward);
                            const v2f r = rcp2(den);
                            fma_where(acc, d2, r, keep);
                            pg += EG_ROW;
                            pd += EG_ROW;
                            den += (float)EG_ROW;
                        }
                    }
                }
                // row sums: quad swaps, then the two mirrors -> every lane of the row holds the segment's sum
                float s0 = acc.x, s1 = acc.y;
                s0 += dpp_f32<0xB1>(s0);  s1 += dpp_f32<0xB1>(s1);
                s0 += dpp_f32<0x4E>(s0);  s1 += dpp_f32<0x4E>(s1);
                if (EG_ROW >= 8) { s0 += dpp_f32<0x141>(s0); s1 += dpp_f32<0x141>(s1); }
                if (EG_ROW == 16) { s0 += dpp_f32<0x140>(s0); s1 += dpp_f32<0x140>(s1); }
                // What the loop did not need -- result slot, the item's flags, its reference values for the rare t == 0
                // correction -- is READ AGAIN from the item here, once per sixteen segments, instead of being carried through
                // the loop in registers: with them the kernel wants 70 VGPRs, i.e. 24 bytes of scratch per lane in the
                // 64-register form (four workgroups per CU) -- 335 MB of scratch stores per headline launch (round 3's
                // WRITE_SIZE 95 -> 353 MB).  The asm statement keeps the compiler from merging the two reads.
                asm volatile("" ::: "memory");
                if (base + row < nc && rl == 0) {
                    const uint4 z0 = q[0];
                    const uint32_t zbits = z0.x & 63u;
                    const float zinv0 = __uint_as_float(z0.y), zinv1 = __uint_as_float(q[1].y);
                    if (zbits & 24u) {                                 // the t == 0 pixel must use -eps (see item format)
                        const float4 zrv = s_val[q[1].z];
                        const v2f zref_ar = {USE_ALPHA ? -zrv.x : 0.0f, USE_RGB ? -zrv.y : 0.0f};
                        const v2f zref_gb = {USE_RGB ? -zrv.z : 0.0f, USE_RGB ? -zrv.w : 0.0f};
                        const int df = (zbits & 8u) ? (int)(z0.z & 0xFFFF) : (int)(z0.z >> 16);
                        const float2 d = s_df[df];
                        const float diff = diff_of(s_grd[df], d, zref_ar, zref_gb);
                        const float dpos = (!(diff <= 0) && (!(zbits & 1u) || __float_as_int(d.y) == (int)(z0.x >> 6))) ? diff : 0.0f;
                        const float zs = (zbits & 32u) ? 1.0f : -1.0f;
                        const float ux = zs * (a.eps * fabsf(zinv0)), uy = zs * (a.eps * fabsf(zinv1));
                        if (ux * zinv0 < 0) s0 -= 2.0f * (dpos * __builtin_amdgcn_rcpf(ux));
                        if (uy * zinv1 < 0) s1 -= 2.0f * (dpos * __builtin_amdgcn_rcpf(uy));
                    }
                    w.results[z0.w] = make_float2((zbits & 2u) ? zinv0 * s0 : 0.0f, (zbits & 4u) ? zinv1 * s1 : 0.0f);
                }
            }

        }
        __syncthreads();                                      // the queue has been read
        if (threadIdx.x == 0) {     // (one lane, constant addresses: a per-thread s_hist address would be one more register
            s_nitems = 0;           //  carried through the whole kernel -- and was the one that spilled)
#pragma unroll
            for (int k = 0; k < 33; k++) s_hist[k] = 0;
        }
        __syncthreads();
    };
    if (threadIdx.x == 0) {
        s_nitems = 0; s_pass[0] = 0; s_pass[1] = 0;
#pragma unroll
        for (int k = 0; k < 33; k++) s_hist[k] = 0;
    }
    __syncthreads();                                          // the image is staged
    int n_nz = -1;                                            // entries of s_nz; -1: this line walks densely
    if constexpr (!USE_RGB) {
        if (a.sparse_max > 0) {
            const v2f one_ar = {-1.0f, 0.0f}, none_gb = {0.0f, 0.0f};
            int run = 0;                                      // (uniform)
            for (int p0 = p_lo; p0 <= p_hi; p0 += EG_LINE_THREADS) {
                const int p = p0 + (int)threadIdx.x;
                bool on = false;
                if (p <= p_hi) on = !(diff_of(s_grd[p], s_df[p], one_ar, none_gb) <= 0);      // (NaN stays, as in the walk)
                const unsigned long long m = __builtin_amdgcn_ballot_w64(on);
                if (lane_here() == 0) s_wcnt[wv] = __popcll(m);
                __syncthreads();
                int before = run, total = 0;
#pragma unroll
                for (int k = 0; k < WAVES; k++) {
                    const int c = s_wcnt[k];
                    before += k < wv ? c : 0;
                    total += c;
                }
                const int at = before + mask_rank(m);
                if (on && at < NZ_CAP) s_nz[at] = (unsigned short)p;
                run = __builtin_amdgcn_readfirstlane(run + total);
                __syncthreads();
            }
            n_nz = run <= min(a.sparse_max, NZ_CAP) ? run : -1;
        }
    }
    // the queue item of a long segment (see "Item" above); 1 / qc by v_rcp_f32 (1 ulp), like every quotient of the walk
    auto make_item = [&](const Segment& q, int fn, uint32_t slot, uint4& rec0, uint4& rec1) {
        const float qc0 = (q.f0 ? q.q0 : 1.0f) * two_over_is, qc1 = (q.f1 ? q.q1 : 1.0f) * two_over_is;
        const float s_t = (float)(q.inward ? -q.dir : q.dir);
        const float rq0 = __builtin_amdgcn_rcpf(qc0), rq1 = __builtin_amdgcn_rcpf(qc1);
        const bool fix = q.inward && (float)q.d1_in == q.d1_cross && q.from <= q.d1_in && q.d1_in <= q.to;
        const uint32_t bits = (uint32_t)q.inward | ((uint32_t)q.f0 << 1) | ((uint32_t)q.f1 << 2) |
                              ((fix && q.dir < 0) ? 8u : 0u) | ((fix && 0 < q.dir) ? 16u : 0u) | (0 < s_t ? 32u : 0u);
        rec0 = make_uint4(bits | ((uint32_t)fn << 6), __float_as_uint(-rq0), (uint32_t)q.from | ((uint32_t)q.to << 16), slot);
        rec1 = make_uint4(__float_as_uint(q.d1_cross), __float_as_uint(-rq1), (uint32_t)q.ref_pos, 0u);
    };
    // a short (or not oriented) segment, walked by its thread from LDS
    auto walk_short = [&](const Segment& q, int fn, float& g0, float& g1) {
        const float4 rv = s_val[q.ref_pos];
        const v2f nref_ar = {USE_ALPHA ? -rv.x : 0.0f, USE_RGB ? -rv.y : 0.0f};
        const v2f nref_gb = {USE_RGB ? -rv.z : 0.0f, USE_RGB ? -rv.w : 0.0f};
        const float qq0 = q.f0 ? q.q0 : 1.0f, qq1 = q.f1 ? q.q1 : 1.0f;
        for (int d1 = q.from; d1 <= q.to; d1++) {
            const float2 d = s_df[d1];
            float diff = diff_of(s_grd[d1], d, nref_ar, nref_gb);
            // inward walks only count the face's own pixels (KCU:470); dropped by a select, like diff <= 0
            if (q.inward && __float_as_int(d.y) != fn) diff = 0.0f;
            visit_pixel(diff, d1, q.d1_cross, qq0, qq1, q.f0 != 0, q.f1 != 0, two_over_is, a.eps, g0, g1);
        }
    };
    for (int chunk0 = 0, pass = 0, step = EG_LINE_THREADS; chunk0 < n_x; chunk0 += step, pass ^= 1) {
        step = EG_LINE_THREADS;
        const int ci = chunk0 + (int)threadIdx.x;
        // ---- both walks classified: empty ones store their zero, short ones are walked here, long ones only flagged ----
        unsigned long long qm[2];
        {
            const uint4 rc = load_record(ci);
            const XGeom geo = record_to_geometry(rc);               // (bits 0 past the line's last record: no walk)
            const int fn = record_face(rc);
            // the outward walk: queued whenever it exists -- nearly all of them are long, and the queued walk takes a
            // segment of any length (outward walks are always oriented), so there is no in-thread form of it (but the
            // alpha-only mode's sparse one)
            auto outward = [&]() {
                bool queue_it = false;
                if (ci < n_x) {
                    Segment q;
                    queue_it = geometry_segment(geo, 0, axis, d0, is, p_lo, p_hi, q);
                    float g0 = 0, g1 = 0;
                    if constexpr (!USE_RGB) {
                        if (queue_it && n_nz >= 0 && s_val[q.ref_pos].x == 1.0f) {      // the sparse form (see s_nz)
                            queue_it = false;
                            int lo = 0, hi = n_nz;
                            while (lo < hi) {                         // first entry at or behind the segment's start
                                const int mid = (lo + hi) >> 1;
                                if ((int)s_nz[mid] < q.from) lo = mid + 1; else hi = mid;
                            }
                            const v2f one_ar = {-1.0f, 0.0f}, none_gb = {0.0f, 0.0f};
                            const float qq0 = q.f0 ? q.q0 : 1.0f, qq1 = q.f1 ? q.q1 : 1.0f;
                            for (int i = lo; i < n_nz; i++) {
                                const int d1 = (int)s_nz[i];
                                if (d1 > q.to) break;
                                visit_pixel_div(diff_of(s_grd[d1], s_df[d1], one_ar, none_gb), d1, q.d1_cross, qq0, qq1, q.f0 != 0,
                                                q.f1 != 0, 2.0f / (float)is, a.eps, g0, g1);
                            }
                        }
                    }
                    if (!queue_it) w.results[2u * (uint32_t)(x_first + ci)] = make_float2(g0, g1);
                }
                qm[0] = __builtin_amdgcn_ballot_w64(queue_it);
            };
            // the inward walk: short ones (mean length 2 pixels) on the spot, the rare long one flagged for the queue
            auto inward = [&]() {
                bool queue_it = false;
                if (ci < n_x) {
                    float g0 = 0, g1 = 0;
                    Segment q;
                    if (geometry_segment(geo, 1, axis, d0, is, p_lo, p_hi, q)) {
                        if (segment_queueable(q)) queue_it = true;
                        else walk_short(q, fn, g0, g1);
                    }
                    if (!queue_it) w.results[2u * (uint32_t)(x_first + ci) + 1u] = make_float2(g0, g1);
                }
                qm[1] = __builtin_amdgcn_ballot_w64(queue_it);
            };
            // (alpha only: the inward walk first -- what it needs of the record is dead before the sparse outward walk's loop
            //  starts; in the other order the 64-register form of this instantiation spilled two registers)
            if constexpr (USE_RGB) { outward(); inward(); } else { inward(); outward(); }
        }
        // the pass's long segments: if they do not fit behind what is queued already, that is walked first
        int n0 = __popcll(qm[0]), n1 = __popcll(qm[1]);
        if (lane_here() == 0 && (n0 | n1)) atomicAdd(&s_pass[pass], n0 + n1);
        __syncthreads();
        const int pass_total = s_pass[pass];
        const bool full = s_nitems + pass_total > EG_QUEUE;       // uniform: read by everyone before anyone appends
        __syncthreads();
        if (full) walk_queue();
        if (threadIdx.x == 0) s_pass[pass ^ 1] = 0;               // (next read: after the next pass's barrier)
        // A pass of EG_LINE_THREADS records can hold up to twice as many long segments as the (now empty) queue -- a line
        // crowded with faces wider than EG_INLINE_MAX pixels.  Then only the records of the first half of the threads
        // queue theirs now, and the other half is taken AGAIN by the next pass (its short walks are stored twice, the
        // same values).
        if (pass_total > EG_QUEUE) {                              // uniform
            step = EG_LINE_THREADS / 2;
            if ((int)threadIdx.x >= step) { qm[0] = 0; qm[1] = 0; n0 = 0; n1 = 0; }       // (whole waves)
        }
        if ((qm[0] | qm[1]) != 0) {                               // (wave-uniform) this wave queues: its records again
            const uint4 rc = load_record(ci);
            const XGeom geo = record_to_geometry(rc);
            const int fn = record_face(rc);
            // queue positions: one LDS atomic per wave, ranks from the ballots (outward items first, then inward)
            int base = 0;
            const int lane = lane_here();
            if (lane == 0 && (n0 | n1)) base = atomicAdd(&s_nitems, n0 + n1);
            base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
            for (int which = 0; which < 2; which++) {
                if ((qm[which] >> lane) & 1ull) {
                    Segment q;
                    geometry_segment(geo, which, axis, d0, is, p_lo, p_hi, q);
                    uint4 rec0, rec1;
                    make_item(q, fn, 2u * (uint32_t)(x_first + ci) + which, rec0, rec1);
                    uint4* it = (uint4*)(s_items + (size_t)(base + (which ? n0 : 0) + mask_rank(qm[which])) * EG_ITEM_DW);
                    it[0] = rec0; it[1] = rec1;
                }
            }
        }
        __syncthreads();
    }
    walk_queue();
    };      // do_line
    if constexpr (!USE_RGB) {
        // behind k_edge_lines_alpha: only the lines that pass left, from its list -- a small grid strides over it (a workgroup
        // per line of the image that leaves at once on nearly every one was 27 us of dispatch per 32-view launch)
        if (a.line_left) {
            const int n_left = __builtin_amdgcn_readfirstlane(*a.n_left);
            for (int item = (int)blockIdx.x; item < n_left; item += (int)gridDim.x) {
                do_line((size_t)a.line_left[item]);
                __syncthreads();                              // (the next line's image takes this one's place)
            }
            return;
        }
    }
#ifndef D3M_EG_LINES_PLAIN_ORDER
    const XcdOrder xo((int)a.n_lines);
    const size_t line = (size_t)xo.unit((int)blockIdx.x);
    if (line >= a.n_lines) return;
#else
    const size_t line = blockIdx.x;
#endif
    do_line(line);
}

// ---- 5a. the alpha-only mode (render_silhouettes, every return_rgb == 0 call) on lines of few contributing pixels ----------
// With alpha alone a walk's terms are  diff_grad(d) = (alpha(d) - alpha_ref) * grad_alpha(d)  (KCU:385-387 / :473-475), kept
// where > 0 (KCU:401/:481), and alpha is 0 or 1: against alpha_ref = 1 (every outward walk: its in-pixel is the face's own)
// only UNCOVERED pixels with a negative gradient count, against alpha_ref = 0 only COVERED pixels with a positive one --
// the two thin bands where a silhouette should grow or shrink.  k_edge_lines stages 40 bytes per pixel of the line, builds
// a queue, sorts it and walks four lanes per segment to find those few pixels (its sparse form of the outward walks took
// the 32-view silhouette step from 1.24 to 1.09 ms; what was left were the fixed phases of a 512-thread workgroup per
// line: 255 us).  Here a line is a workgroup of TWO waves holding 12 bytes per pixel (alpha, gradient, owner) and the two
// bands as sorted position lists; a thread per crossing record adds its two walks' handful of terms itself -- the reference's
// own expression, (alpha(d) - alpha_ref) first -- and stores the crossing's two result slots.
// No queue, no 16-wave barriers, sixteen lines in flight per CU instead of four.  A line whose bands hold more than
// EGA_LIST pixels each (a dense alpha gradient: every term counts) or whose alpha is not 0 / 1 somewhere (a caller's own
// map) is LEFT to k_edge_lines, which is launched behind this kernel and leaves at once on every line walked here
// (EdgeGradArgs::line_left).
constexpr int EGA_THREADS = 128;
constexpr int EGA_LIST = 128;
inline size_t edge_lines_alpha_lds(int S) { return (size_t)S * 12; }
__global__ void __launch_bounds__(EGA_THREADS) k_edge_lines_alpha(EdgeGradArgs a, EdgePlan w) {
    extern __shared__ __attribute__((aligned(16))) float s_aline[];
    __shared__ unsigned short s_raw[2][EGA_LIST], s_pos[2][EGA_LIST];     // [0]: counts against alpha_ref = 1, [1]: against 0
    __shared__ int s_n[2], s_odd;
    if (!plan_complete(w)) return;                            // (k_edge_lines / k_edge_overflow handle that case)
    const int is = a.S;
    const XcdOrder xo((int)a.n_lines);
    const size_t line = (size_t)xo.unit((int)blockIdx.x);     // (b*2 + axis)*S + d0
    if (line >= a.n_lines) return;
    const int n_x = __builtin_amdgcn_readfirstlane(w.line_cursor[line]);
    if (n_x <= 0) return;
    const int d0 = (int)(line % is), axis = (int)((line / is) & 1);
    const size_t bn = line / ((size_t)2 * is), view_base = bn * is * is;
    const int x_first = __builtin_amdgcn_readfirstlane(w.line_slice[line].x);
    const int p_lo = __builtin_amdgcn_readfirstlane(a.extent_lo(line));
    const int p_hi = __builtin_amdgcn_readfirstlane(a.extent_hi(line));
    float* s_al = s_aline;                                    // [S] alpha of the whole line (a walk's reference pixel may lie
    float* s_g = s_al + is;                                   //     outside the gradients' extent); [S] gradient; [S] owner
    int* s_own = (int*)(s_g + is);
    if (threadIdx.x < 2) s_n[threadIdx.x] = 0;
    if (threadIdx.x == 2) s_odd = 0;
    const float go_sign = a.go_sign();
    bool odd = false;
    // (four pixels per thread and round, every load of a round requested before any is used: a column line's pixels are a
    //  raster row apart, one round trip to the L2 each -- four in flight instead of four in a row)
    for (int p0 = (int)threadIdx.x; p0 < is; p0 += 4 * EGA_THREADS) {
        float al[4], gx[4], ow[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int p = p0 + k * EGA_THREADS;
            const size_t pi = a.pixel(axis, view_base, d0, min(p, is - 1));
            al[k] = a.alpha_map[pi];
            const bool in = p >= p_lo && p <= p_hi;
            gx[k] = in ? (a.direct() ? a.direct_ga_at(bn, axis, d0, min(p, is - 1), pi) : a.grad[pi].x) : 0.0f;
            ow[k] = in ? (a.direct() ? __int_as_float(a.fi_direct[pi]) : a.dot[pi].y) : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int p = p0 + k * EGA_THREADS;
            if (p < is) {
                s_al[p] = al[k];
                odd = odd || !(al[k] == 0.0f || al[k] == 1.0f);
                s_g[p] = go_sign * gx[k];
                s_own[p] = __float_as_int(ow[k]);
            }
        }
    }
    __syncthreads();
    if (odd) s_odd = 1;
    for (int p = p_lo + (int)threadIdx.x; p <= p_hi; p += EGA_THREADS) {
        const float al = s_al[p], g = s_g[p];
        if (!((al - 1.0f) * g <= 0)) {                        // (NaN stays, as in the walk)
            const int k = atomicAdd(&s_n[0], 1);
            if (k < EGA_LIST) s_raw[0][k] = (unsigned short)p;
        }
        if (!(al * g <= 0)) {
            const int k = atomicAdd(&s_n[1], 1);
            if (k < EGA_LIST) s_raw[1][k] = (unsigned short)p;
        }
    }
    __syncthreads();
    const int n_out = s_n[0], n_in = s_n[1];
    const bool left = n_out > EGA_LIST || n_in > EGA_LIST || s_odd != 0 || a.sparse_max <= 0;      // (uniform)
    if (left) {
        if (threadIdx.x == 0) a.line_left[atomicAdd(a.n_left, 1)] = (int)line;
        return;
    }
    // ascending order (the entries arrive in any order): rank = the number of smaller positions.  The walks then add their
    // terms in pixel order -- the same sums in every run
    for (int e = (int)threadIdx.x; e < n_out + n_in; e += EGA_THREADS) {
        const int which = e < n_out ? 0 : 1, i = which ? e - n_out : e, n = which ? n_in : n_out;
        const unsigned short mine = s_raw[which][i];
        int rank = 0;
        for (int j = 0; j < n; j++) rank += s_raw[which][j] < mine ? 1 : 0;
        s_pos[which][rank] = mine;
    }
    __syncthreads();
    const float two_over_is = 2.0f / (float)is;
    const uint4* xrec = w.xrec + (size_t)x_first;
    uint4 rc_next = make_uint4(0, 0, 0, 0);
    if ((int)threadIdx.x < n_x) rc_next = xrec[threadIdx.x];
    for (int ci = (int)threadIdx.x; ci < n_x; ci += EGA_THREADS) {
        const uint4 rc = rc_next;                             // (the next round's record is requested before this one's walks)
        if (ci + EGA_THREADS < n_x) rc_next = xrec[ci + EGA_THREADS];
        const XGeom geo = record_to_geometry(rc);
        const int fn = record_face(rc);
#pragma unroll
        for (int which = 0; which < 2; which++) {             // 0: outward, 1: inward
            float g0 = 0.0f, g1 = 0.0f;
            Segment q;
            if (geometry_segment(geo, which, axis, d0, is, p_lo, p_hi, q)) {
                const float ref = s_al[q.ref_pos];            // (0 or 1: the line holds nothing else)
                // (an inward walk against a COVERED out-pixel -- every interior edge -- has nothing to add: it counts the
                //  face's own, covered pixels only, whose terms (1 - 1) * g vanish)
                const int list = ref == 1.0f ? 0 : 1, n = (which == 1 && ref == 1.0f) ? 0 : (list ? n_in : n_out);
                const float qq0 = q.f0 ? q.q0 : 1.0f, qq1 = q.f1 ? q.q1 : 1.0f;
                int lo = 0, hi = n;
                while (lo < hi) {                             // first entry at or behind the segment's start
                    const int mid = (lo + hi) >> 1;
                    if ((int)s_pos[list][mid] < q.from) lo = mid + 1; else hi = mid;
                }
                // (v_rcp_f32 quotients, as in every other walk: with correctly rounded divisions this loop was the kernel --
                //  82 M wave VALU instructions per 32-view launch, 133 us of issue)
                for (int i = lo; i < n; i++) {
                    const int d1 = (int)s_pos[list][i];
                    if (d1 > q.to) break;
                    if (which == 1 && s_own[d1] != fn) continue;                  // KCU:470: the face's own pixels only
                    visit_pixel((s_al[d1] - ref) * s_g[d1], d1, q.d1_cross, qq0, qq1, q.f0 != 0, q.f1 != 0, two_over_is, a.eps,
                                g0, g1);
                }
            }
            w.results[2u * (uint32_t)(x_first + ci) + which] = make_float2(g0, g1);
        }
    }
}

// ---- 5. per visible face: the results of its six lanes' crossings, stored once ---------------------------
// `parts` (> 1 only with a vertex target, whose sums are ADDED): a block's lanes are dealt to that many workgroups, each
// taking every parts-th group of four crossings of every lane -- a coarse mesh's lane has hundreds of crossings to add up, one
// dependent round trip per four (8 triangles @1024^2: 165 us for one workgroup's six-times-eight lanes).
template <class FS>
__global__ void __launch_bounds__(256) k_edge_gather(FS fs, EdgePlan w, const float2* __restrict__ lane_partial,
                                                    const float* __restrict__ go, float* __restrict__ grad_faces,
                                                    VertexTarget vt, int parts) {
    __shared__ float2 s_g[256];
    if (blockIdx.x == 0 && threadIdx.x == 0) w.alloc[2] = 0;    // (k_edge_lines_alpha's list of left lines: EdgeGradArgs::n_left)
    const int n_vis = *w.n_visible;
    const int n_blocks = (n_vis + EG_FACES_PER_BLOCK - 1) / EG_FACES_PER_BLOCK;
    const int n_units = n_blocks * parts;
    const XcdOrder xo(n_units);
    const bool complete = plan_complete(w);      // results in record order (found through xpos), else in crossing order
    const float go_abs = go ? fabsf(*go) : 1.0f; // the magnitude of the factor the records lacked (EdgeGradArgs::go)
    for (int i = blockIdx.x; xo.more(i); i += gridDim.x) {
        const int unit = xo.unit(i);
        if (unit >= n_units) continue;
        const int blk = unit / parts, part = unit - blk * parts;
        const int t = threadIdx.x;
        const int pos = blk * EG_FACES_PER_BLOCK + t / 6, ea = t % 6;
        const bool on = t < EG_FACES_PER_BLOCK * 6 && pos < n_vis;
        float2 g = make_float2(0.0f, 0.0f);
        if (on) {
            const int2 lc = w.lane_cross[(size_t)pos * 6 + ea];
            if (!complete && part == 0) g = lane_partial[(size_t)pos * 6 + ea];
            // slots (2c, 2c+1) of the lane's crossings c, contiguous and 16-byte aligned: one float4 per crossing, four
            // crossings requested per round, added in slot order; crossings past cap were folded into lane_partial
            const long c_first = (long)w.lane_block[blk] + lc.x, c_last = min(c_first + (long)lc.y, (long)w.cap);
            const float4* res4 = (const float4*)w.results;
            for (long c = c_first + 4 * part; c < c_last; c += 4 * parts) {
                long at[4];
#pragma unroll
                for (int j = 0; j < 4; j++) at[j] = (complete && c + j < c_last) ? (long)w.xpos[c + j] : c + j;
                float4 r[4];
#pragma unroll
                for (int j = 0; j < 4; j++)       // (at < 0: the crossing got no record because its walks cannot contribute)
                    r[j] = (c + j < c_last && at[j] >= 0) ? res4[at[j]] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
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
                const float2 r = make_float2(s_g[t + e].x * go_abs, s_g[t + e].y * go_abs);
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

// Per-pixel walk records and the lines' non-zero extents, one pass over the five maps: a 32x32 tile per workgroup
// (the column extents are merged per tile in LDS: one atomic per column and tile).
// The gradients can also come straight from the OUTPUT images' gradients (ImageGrads: CHW, flipped, 2x2-pooled with
// anti-aliasing -- rasterize.py:305-326 undone on the fly, d3m_output_epilogue_backward_records): the adjoint of the
// output epilogue then has no pass and no [B,S,S,3] / [B,S,S] gradient maps of its own; the depth gradient, which the
// edge gradient does not read, is written out as the map its readers take.
struct ImageGrads {
    const float *g_rgb_out, *g_alpha_out, *g_depth_out;   // [B,3,s,s], [B,s,s], [B,s,s] or NULL
    float* g_depth_map;                                   // [B,S,S] out (with g_depth_out)
    int s, aa, on;
};
__global__ void __launch_bounds__(256) k_pack_maps(const int32_t* __restrict__ fi, const float* __restrict__ alpha,
                                                  const float* __restrict__ galpha, const float* __restrict__ rgb,
                                                  const float* __restrict__ grgb, float4* __restrict__ grad_row,
                                                  float2* __restrict__ dot_row, int* __restrict__ nz_lo_inv,
                                                  int* __restrict__ nz_hi1, int S, GradScale gs,
                                                  ImageGrads img = ImageGrads{nullptr, nullptr, nullptr, nullptr, 0, 0, 0}) {
    __shared__ int s_col_lo_inv[32], s_col_hi1[32];
    if (threadIdx.x < 32) { s_col_lo_inv[threadIdx.x] = 0; s_col_hi1[threadIdx.x] = 0; }
    __syncthreads();
    const int b = blockIdx.z;
    float s_rgb, s_alpha, s_depth;
    gs.get(s_rgb, s_alpha, s_depth);
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
            if (img.on) {       // output pixel of internal pixel (y, x): flipped, pooled 2x2 with anti-aliasing
                const int yo = img.aa ? (S - 1 - y) / 2 : S - 1 - y, xo = img.aa ? x / 2 : x;
                const float sc = img.aa ? 0.25f : 1.0f;
                const size_t o = ((size_t)b * img.s + yo) * img.s + xo;
                if (alpha && img.g_alpha_out) { g.x = img.g_alpha_out[o] * sc; dot += alpha[i] * g.x; }
                if (rgb && img.g_rgb_out) {
                    const size_t plane_o = (size_t)img.s * img.s, o0 = ((size_t)b * 3 * img.s + yo) * img.s + xo;
                    g.y = img.g_rgb_out[o0] * sc; g.z = img.g_rgb_out[o0 + plane_o] * sc; g.w = img.g_rgb_out[o0 + 2 * plane_o] * sc;
                    dot += rgb[3 * i] * g.y;
                    dot += rgb[3 * i + 1] * g.z;
                    dot += rgb[3 * i + 2] * g.w;
                }
                if (img.g_depth_map) img.g_depth_map[i] = img.g_depth_out[o] * sc;
            } else if (alpha) { g.x = galpha[i] * s_alpha; dot += alpha[i] * g.x; }
            if (!img.on && rgb) {
                g.y = grgb[3 * i] * s_rgb; g.z = grgb[3 * i + 1] * s_rgb; g.w = grgb[3 * i + 2] * s_rgb;
                dot += rgb[3 * i] * g.y;
                dot += rgb[3 * i + 1] * g.z;
                dot += rgb[3 * i + 2] * g.w;
            }
            const float2 d = make_float2(dot, __int_as_float(fi[i]));
            grad_row[i] = g;
            dot_row[i] = d;
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
    unsigned char* marks;   // [B*F] one byte per face, the intermediate the three arrays above are built from
};

inline size_t visibility_bytes(long nf) {
    return eg_align((size_t)nf * 4) * 2 + eg_align(((size_t)nf / EG_COMPACT_CHUNK + 2) * 4) + 256 +
           eg_align((size_t)nf + 4);
}

inline VisibilityView visibility_view(void* blob, long nf) {
    char* p = (char*)blob;
    VisibilityView v;
    v.flags = (int*)p;                        p += eg_align((size_t)nf * 4);
    v.list = (int*)p;                         p += eg_align((size_t)nf * 4);
    v.vis_block = (int*)p;                    p += eg_align(((size_t)nf / EG_COMPACT_CHUNK + 2) * 4);
    v.count = (int*)p;                        p += 256;
    v.marks = (unsigned char*)p;
    return v;
}

// D3M_DETERMINISTIC (d3m_set_deterministic): the list in ASCENDING face order -- count, scan, compact: three launches
// instead of one -- instead of chunks in arrival order.  Every pass over the list then adds its float atomics from the same
// workgroups in every run (see DESIGN.md section 6 for what that does and does not guarantee).
inline std::atomic<int> g_deterministic{-1};        // -1: not yet read from the environment
inline bool deterministic_mode() {
    int d = g_deterministic.load(std::memory_order_relaxed);
    if (d < 0) {
        const char* e = getenv("D3M_DETERMINISTIC");
        d = (e && e[0] == '1') ? 1 : 0;
        int expected = -1;
        if (!g_deterministic.compare_exchange_strong(expected, d)) d = expected;
    }
    return d == 1;
}
inline hipError_t run_visibility(const int32_t* face_index_map, const VisibilityView& v, int B, int F, int S, hipStream_t st) {
    const long nf = (long)B * F;
    if (face_index_map) {       // NULL: the marks were left (and the count cleared) by d3m_forward_face_index_map_mesh
        hipError_t e = zero_async(v.count, 256 + eg_align((size_t)nf + 4), st);      // count | marks are adjacent
        if (e != hipSuccess) return e;
        const long px_blocks = ((long)B * S * S + 255) / 256;
        LAUNCH("k_mark_visible", k_mark_visible_bytes, dim3((unsigned)std::min(px_blocks, 4096l)), dim3(256), st,
               face_index_map, v.marks, B, F, S);
    }
    if (deterministic_mode()) {
        const int n_blocks = (int)((nf + EG_COMPACT_CHUNK - 1) / EG_COMPACT_CHUNK);
        LAUNCH("k_count_visible", k_count_visible, dim3(n_blocks), dim3(256), st, (const unsigned char*)v.marks, v.vis_block, nf);
        LAUNCH("k_scan_small", k_scan_small, dim3(1), dim3(1024), st, v.vis_block, n_blocks, (const int*)nullptr, 1, v.count);
        LAUNCH("k_compact_visible", k_compact_visible, dim3(n_blocks), dim3(256), st, (const unsigned char*)v.marks, v.flags,
               v.list, (const int*)v.vis_block, nf);
        return hipGetLastError();
    }
    const int n_chunks = (int)((nf + EG_COMPACT1_FACES - 1) / EG_COMPACT1_FACES);
    LAUNCH("k_compact_visible", k_compact_visible_atomic, dim3(n_chunks), dim3(1024), st, (const unsigned char*)v.marks, v.flags,
           v.list, v.count, nf);
    return hipGetLastError();
}

// count -> crossing base per workgroup -> record slice per line -> records (geometry only: see struct EdgePlan)
struct EdgePlanLayout {
    size_t off_line_count, off_line_cursor, off_alloc, off_extents, zero_bytes;     // the zeroed prefix
    size_t off_lane_cross, off_lane_block, off_line_slice, off_xrec;   // xrec | results follow, sized by capacity
    size_t fixed_bytes;
};
constexpr size_t EG_BYTES_PER_CROSSING = 16 + 16 + 4;  // record + its two result slots + its position
// Default capacity of a plan, in crossings per view: two per face (visible or not) or three per raster pixel, whichever is
// more.  A mesh that covers P pixels with triangles of A pixels each has about 6 P / sqrt(A) crossings (every triangle's
// three edges, both axes, counted once per adjacent face): the 100 k-triangle mesh at 512^2 has 261 k per view of the 401 k
// its faces grant, but at 1024^2 -- the same faces four times as large -- about a million, and a plan that cannot hold the
// batch's crossings falls back to k_edge_overflow, which walks them from global memory (correct, and thirty times slower:
// 8.0 ms instead of 0.3 for eight such views, measured in round 4 when the capacity still went by faces alone).
inline size_t eg_default_crossings(int B, int F, int S) {
    return (size_t)B * std::max((size_t)2 * F, (size_t)3 * S * S);
}

inline EdgePlanLayout edge_plan_layout(int B, int F, int S) {
    const size_t nf = (size_t)B * F, nl = (size_t)B * 2 * S;
    EdgePlanLayout L;
    size_t o = 0;
    L.off_line_count = o;  o += eg_align(nl * 4);
    L.off_line_cursor = o; o += eg_align(nl * 4);
    L.off_alloc = o;       o += 256;
    // room for the lines' non-zero-gradient extents (nz_lo_inv | nz_hi1, [B,2,S] ints each, 256-byte aligned) of a fused fit
    // objective: they must be zero before the pass that fills them, and a caller that builds the plan BEFORE that pass on the
    // same stream (d3m_edge_plan_extents_offset) gets them cleared by the plan's own clear instead of by a launch of its own
    L.off_extents = o;     o += 2 * eg_align(nl * 4);
    L.zero_bytes = o;
    L.off_lane_cross = o;  o += eg_align(nf * 6 * 8);
    L.off_lane_block = o;  o += eg_align((nf / EG_FACES_PER_BLOCK + 2) * 4);
    L.off_line_slice = o;  o += eg_align(nl * 8);
    L.off_xrec = o;
    L.fixed_bytes = o;
    return L;
}

inline size_t edge_plan_min_bytes(int B, int F, int S) { return edge_plan_layout(B, F, S).fixed_bytes + 1024; }
inline size_t edge_plan_bytes(int B, int F, int S) {
    return edge_plan_min_bytes(B, F, S) + eg_align(eg_default_crossings(B, F, S) * EG_BYTES_PER_CROSSING);
}

// the plan as laid out in `blob` (flags / list / count come from a visibility blob)
inline bool edge_plan_view(void* blob, size_t bytes, const VisibilityView& v, int B, int F, int S, EdgePlan& w) {
    const EdgePlanLayout L = edge_plan_layout(B, F, S);
    if (!blob || bytes < L.fixed_bytes + 1024) return false;
    char* p = (char*)blob;
    size_t cap = (bytes - L.fixed_bytes - 512) / EG_BYTES_PER_CROSSING;
    cap = cap > 64 ? cap - 32 : 0;                             // slack for the 256-byte alignments below
    if (cap > 0x3FFFFF00) cap = 0x3FFFFF00;                    // 2 * cap result slots are indexed with an int
    w.visible = v.flags; w.visible_list = v.list; w.n_visible = v.count;
    w.line_count = (int*)(p + L.off_line_count);
    w.line_cursor = (int*)(p + L.off_line_cursor);
    w.alloc = (int*)(p + L.off_alloc);
    w.lane_cross = (int2*)(p + L.off_lane_cross);
    w.lane_block = (int*)(p + L.off_lane_block);
    w.line_slice = (int2*)(p + L.off_line_slice);
    w.xrec = (uint4*)(p + L.off_xrec);
    w.results = (float2*)(p + eg_align(L.off_xrec + cap * 16));
    w.xpos = (int*)(p + eg_align(eg_align(L.off_xrec + cap * 16) + cap * 16));
    w.cap = (int)cap;
    return true;
}

template <class FS>
inline hipError_t run_edge_plan(FS fs, const int32_t* face_index_map, const EdgePlan& w, void* blob, int B, int S,
                                hipStream_t st, bool cleared = false) {
    const EdgePlanLayout L = edge_plan_layout(B, fs.num_faces(), S);
    hipError_t e = cleared ? hipSuccess : zero_async(blob, L.zero_bytes, st);      // (cleared: by the caller, D3M_PRECLEARED)
    if (e != hipSuccess) return e;
    // the passes walk the compacted list with a fixed grid (n_visible is only known on the device); workgroups past
    // it leave on their first load
    const long nf = (long)B * fs.num_faces(), nl = (long)B * 2 * S;
    const long g6_full = (nf + EG_FACES_PER_BLOCK - 1) / EG_FACES_PER_BLOCK;
    const dim3 g6((unsigned)(g6_full < 8192 ? (g6_full + 7) / 8 * 8 : 8192));      // a multiple of 8: see XcdOrder
    const bool window = S <= EG_WINDOW_MAX_S;               // the workgroups' lines fit LDS
    if (window) LAUNCH_SMEM("k_edge_count", k_edge_count_window<FS>, g6, dim3(256), (size_t)2 * (S + 1) * 4, st, fs, S, w);
    else LAUNCH("k_edge_count", k_edge_count<FS>, g6, dim3(256), st, fs, S, w);
    const int blocks_a = (int)((g6_full + 255) / 256);
    LAUNCH("k_alloc_plan", k_alloc_plan, dim3((unsigned)(blocks_a + (nl + 255) / 256)), dim3(256), st, w.lane_block,
           (const int*)w.n_visible, EG_FACES_PER_BLOCK, w.alloc, blocks_a, (const int*)w.line_count, w.line_slice, w.alloc + 1, nl);
    // (the scatter pass keeps the by-key form: with the ranks taken from LDS cursors it was slower, 0.173 vs 0.150 ms)
    // (few blocks of faces: their rounds dealt to several workgroups each -- see the kernel)
    // (round 6: the same dealing decided on the DEVICE from the number of listed faces -- a small batch of a fine mesh leaves
    //  most of this grid idle too -- bought nothing: 4 views of the headline mesh 31 -> 33 us with two parts, 38 with four;
    //  the pass is a chain of dependent look-ups per round, and more workgroups only repeat its set-up.  docs/EXPERIMENTS.md E)
    const int parts = g6_full >= 2048 ? 1 : (int)std::min<long>(8, 2048 / std::max<long>(1, g6_full));
    const dim3 g_scatter((unsigned)std::min<long>(8192, (g6_full * parts + 7) / 8 * 8));
    LAUNCH("k_edge_scatter", k_edge_scatter<FS>, g_scatter, dim3(256), st, fs, face_index_map, S, w, parts);
    return hipGetLastError();
}

// ---- host side ----------------------------------------------------------------------------------------
// Workspace of d3m_backward_pixel_map: the per-pixel walk records, the lines' extents, the lanes' overflow sums, and
// room for a visibility blob and a plan of its own (used when the caller brings none).
struct EdgeLayout {
    size_t off_grad_row, off_dot_row;
    size_t off_nz_lo, off_nz_hi, nz_bytes;     // zeroed per call
    size_t off_lane_partial, off_line_left, off_visibility, off_plan;
    size_t fixed_bytes;                         // everything but the plan (sized by capacity)
};

inline EdgeLayout edge_layout(int B, int F, int S) {
    const size_t px = (size_t)B * S * S, nf = (size_t)B * F, nl = (size_t)B * 2 * S;
    EdgeLayout L;
    size_t o = 0;
    L.off_grad_row = o; o += eg_align(px * 16);
    L.off_dot_row = o;  o += eg_align(px * 8);
    L.off_nz_lo = o;    o += eg_align(nl * 4);
    L.off_nz_hi = o;    o += eg_align(nl * 4);
    L.nz_bytes = o - L.off_nz_lo;
    L.off_lane_partial = o; o += eg_align(nf * 6 * 8);
    L.off_line_left = o;    o += eg_align(nl * 4);
    L.off_visibility = o;   o += eg_align(visibility_bytes((long)nf));
    L.off_plan = o;
    L.fixed_bytes = o;
    return L;
}

inline size_t edge_grad_workspace_min_bytes(int B, int F, int S) {
    return edge_layout(B, F, S).fixed_bytes + edge_plan_min_bytes(B, F, S);
}
inline size_t edge_grad_workspace_bytes(int B, int F, int S) {
    return edge_layout(B, F, S).fixed_bytes + edge_plan_bytes(B, F, S);
}

// per-pixel records the caller already has (the fused fit epilogue wrote them): k_pack_maps is skipped
struct EdgeRecords {
    const float4* grad;
    const float2* dot;
    const int* nz_lo_inv;
    const int* nz_hi1;
    const float* go;       // the factor they still lack (device scalar; NULL = 1)
    // alpha only, no records at all (EdgeGradArgs "DIRECT"): the OUTPUT image's alpha gradient [B,s,s], s = S or S/2 (aa)
    const float* ga_img = nullptr;
    int img_aa = 0;
};

// the lanes' overflow sums start at zero -- only ever used when the plan is incomplete (leaves at once otherwise)
// dynamic LDS of k_edge_lines: pairs [S] (8 B, padded to 16), gradients [S] and values [S] (16 B); the gradients must be
// readable up to entry 2*S + 16 (PAD)
inline size_t edge_lines_lds(int S) {
    const size_t pairs = (size_t)(S + (S & 1)) * 8, body = pairs + (size_t)S * 32;
    return std::max(body, pairs + (size_t)(2 * S + 32) * 16);
}

template <class FS>
int run_edge_grad(FS fs, PixelMaps m, float* grad_faces, VertexTarget vt, const VisibilityView* shared_vis,
                  void* shared_plan, size_t shared_plan_bytes, EdgeRecords rec, GradScale gs, int B, float eps, void* ws,
                  size_t ws_bytes, hipStream_t st, int* last_err) {
    const int S = m.S, F = fs.num_faces();
    if (S > 65535 || F > (1 << 25) || (long)B * 2 * S >= (1l << 31)) return 1;   // item packing / line key limits (D3M_ERR_INVALID)
    const size_t smem = edge_lines_lds(S);
    if (smem + EG_LINE_STATIC_LDS > 160 * 1024) return 1;                // a line does not fit LDS beside the item queue (S > ~5600)
    const EdgeLayout L = edge_layout(B, F, S);
    if (!ws || ws_bytes < L.fixed_bytes + (shared_plan ? 0 : edge_plan_min_bytes(B, F, S))) return 2;   // D3M_ERR_WORKSPACE
    char* p = (char*)ws;
    const long nf = (long)B * F, nl = (long)B * 2 * S;
    hipError_t e;
    // which faces own a pixel: the caller's d3m_visibility, or one built here
    VisibilityView vis;
    if (shared_vis) {
        vis = *shared_vis;
    } else {
        vis = visibility_view(p + L.off_visibility, nf);
        e = run_visibility(m.face_index_map, vis, B, F, S, st);
        if (e != hipSuccess) { *last_err = (int)e; return 3; }
    }
    // the crossings' records: the caller's d3m_edge_plan, or one built here
    EdgePlan w;
    if (shared_plan) {
        if (!edge_plan_view(shared_plan, shared_plan_bytes, vis, B, F, S, w)) return 2;
    } else {
        if (!edge_plan_view(p + L.off_plan, ws_bytes - L.fixed_bytes, vis, B, F, S, w)) return 2;
        e = run_edge_plan(fs, m.face_index_map, w, p + L.off_plan, B, S, st);
        if (e != hipSuccess) { *last_err = (int)e; return 3; }
    }
    float4* grad_row = (float4*)(p + L.off_grad_row);
    float2* dot_row = (float2*)(p + L.off_dot_row);
    float2* lane_partial = (float2*)(p + L.off_lane_partial);
    EdgeGradArgs a;
    a.ga_map = a.ga_img = nullptr; a.fi_direct = m.face_index_map; a.img_s = S; a.img_aa = 0;
    // DIRECT: final alpha gradients need no records (D3M_EG_DIRECT=0: pack an internal-layout map all the same, for A/B
    // runs; the output image's gradient has no packed form)
    const bool direct = !rec.grad && !m.use_rgb && m.use_alpha && !gs.totals &&
                        (rec.ga_img || d3m_env_int("D3M_EG_DIRECT", 1) != 0);
    if (rec.grad) {
        a.grad = rec.grad; a.dot = rec.dot; a.go = rec.go;
        a.nz_lo_inv = rec.nz_lo_inv; a.nz_hi1 = rec.nz_hi1;
    } else if (direct) {
        a.grad = nullptr; a.dot = nullptr; a.go = nullptr; a.nz_lo_inv = a.nz_hi1 = nullptr;
        if (rec.ga_img) { a.ga_img = rec.ga_img; a.img_aa = rec.img_aa; a.img_s = rec.img_aa ? S / 2 : S; }
        else a.ga_map = m.grad_alpha_map;
    } else {
        e = zero_async(p + L.off_nz_lo, L.nz_bytes, st);
        if (e != hipSuccess) { *last_err = (int)e; return 3; }
        LAUNCH("k_pack_maps", k_pack_maps, dim3((S + 31) / 32, (S + 31) / 32, B), dim3(256), st, m.face_index_map,
               m.use_alpha ? m.alpha_map : nullptr, m.use_alpha ? m.grad_alpha_map : nullptr,
               m.use_rgb ? m.rgb_map : nullptr, m.use_rgb ? m.grad_rgb_map : nullptr, grad_row, dot_row,
               (int*)(p + L.off_nz_lo), (int*)(p + L.off_nz_hi), S, gs);
        a.grad = grad_row; a.dot = dot_row; a.go = nullptr;
        a.nz_lo_inv = (const int*)(p + L.off_nz_lo); a.nz_hi1 = (const int*)(p + L.off_nz_hi);
    }
    a.alpha_map = m.alpha_map; a.rgb_map = m.rgb_map;
    a.S = S; a.use_rgb = m.use_rgb; a.use_alpha = m.use_alpha; a.eps = eps; a.n_lines = (unsigned)nl;
    // (a quarter of the line: beyond that the list holds about as many entries as the dense walks would visit)
    a.sparse_max = m.use_rgb ? 0 : d3m_env_int("D3M_EG_SPARSE_MAX", S / 4);
    a.line_left = nullptr;
    a.n_left = nullptr;
    const dim3 glines((unsigned)((nl + 7) / 8 * 8));            // a multiple of 8: see XcdOrder
    // alpha only: the lines of few contributing pixels first, two waves per line (k_edge_lines_alpha); k_edge_lines behind it
    // takes what that pass leaves
    if (!m.use_rgb && m.use_alpha && a.sparse_max > 0 && edge_lines_alpha_lds(S) <= 64 * 1024) {
        a.line_left = (int*)(p + L.off_line_left);
        a.n_left = w.alloc + 2;
        LAUNCH_SMEM("k_edge_lines_alpha", k_edge_lines_alpha, glines, dim3(EGA_THREADS), edge_lines_alpha_lds(S), st, a, w);
    }
    // (behind that pass k_edge_lines strides over its list of left lines with a small grid)
    const dim3 glines_general(a.line_left ? std::min(glines.x, 2048u) : glines.x);
#define D3M_LINES(RGB, ALPHA, WV)                                                                                    \
    do {                                                                                                             \
        if (smem + eg_line_static_lds(WV) > 64 * 1024) {                                                             \
            e = hipFuncSetAttribute((const void*)k_edge_lines<RGB, ALPHA, WV>,                                       \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                          \
            if (e != hipSuccess) { *last_err = (int)e; return 3; }                                                   \
        }                                                                                                            \
        LAUNCH_SMEM("k_edge_lines", (k_edge_lines<RGB, ALPHA, WV>), glines_general, dim3(WV * 64), smem, st, a, w,   \
                    lane_partial);                                                                                   \
    } while (0)
    // Waves per workgroup: the narrow form while its workgroups keep a CU at 24 waves or more (their phases -- stage, set
    // up, order, walk -- overlap only ACROSS workgroups, and a barrier over 16 waves costs more than one over 8: 16 waves at
    // 512^2 took 0.60 ms against 0.50), the 16-wave form (two workgroups = 32 waves) where the lines' LDS images leave
    // fewer (S = 1024: two 8-wave workgroups = 16 waves; 1.08 -> 0.875 ms alone on BASELINE config 5)
#ifndef D3M_EG_NARROW_WAVES
#define D3M_EG_NARROW_WAVES 8
#endif
    const size_t lds_cu = 160 * 1024;
    const int wg_narrow = (int)std::min(lds_cu / (smem + eg_line_static_lds(D3M_EG_NARROW_WAVES)), (size_t)(32 / D3M_EG_NARROW_WAVES));
    const int wg_wide = (int)std::min(lds_cu / (smem + eg_line_static_lds(16)), (size_t)2);
    const bool wide = wg_narrow * D3M_EG_NARROW_WAVES < 24 && wg_wide * 16 > wg_narrow * D3M_EG_NARROW_WAVES;
#define D3M_LINES2(RGB, ALPHA) do { if (wide) D3M_LINES(RGB, ALPHA, 16); else D3M_LINES(RGB, ALPHA, D3M_EG_NARROW_WAVES); } while (0)
    if (m.use_rgb && m.use_alpha) D3M_LINES2(true, true);
    else if (m.use_rgb) D3M_LINES2(true, false);
    else D3M_LINES2(false, true);
#undef D3M_LINES2
#undef D3M_LINES
    const long g6_full = (nf + EG_FACES_PER_BLOCK - 1) / EG_FACES_PER_BLOCK;
    const dim3 g6((unsigned)(g6_full < 8192 ? (g6_full + 7) / 8 * 8 : 8192));      // a multiple of 8: see XcdOrder
    // crossings without a record (workspace smaller than the scene needs): leaves at once otherwise
    // (normally leaves at once: a small grid keeps that cheap; with an undersized workspace its workgroups stride)
    LAUNCH("k_edge_overflow", k_edge_overflow<FS>, dim3(std::min(g6.x, 512u)), dim3(256), st, fs, a, w, lane_partial);
    // (few blocks of faces, sums added into a vertex target: a block's lanes dealt to several workgroups -- see the kernel)
    const int gparts = (!vt.gv || g6_full >= 2048) ? 1 : (int)std::min<long>(8, 2048 / std::max<long>(1, g6_full));
    const dim3 g_gather((unsigned)std::min<long>(8192, (g6_full * gparts + 7) / 8 * 8));
    LAUNCH("k_edge_gather", k_edge_gather<FS>, g_gather, dim3(256), st, fs, w, (const float2*)lane_partial, a.go, grad_faces, vt,
           gparts);
    e = hipGetLastError();
    if (e != hipSuccess) { *last_err = (int)e; return 3; }
    return 0;
}

}  // namespace d3m
