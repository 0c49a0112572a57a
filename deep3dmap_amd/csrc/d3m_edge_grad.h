// d3m_edge_grad.h -- the edge / silhouette gradient (reference: backward_pixel_map_cuda_kernel,
// KCU:245-503) restructured for gfx950.
//
// The reference runs one thread per face; for every integer crossing d0 of every edge, on both axes,
// that thread walks the image from the edge out to the image BORDER (KCU:354-414) and inward to the
// opposite edge (KCU:417-495).  The outward walks are hundreds of pixels long whatever the triangle
// size: ~6e8 pixel visits x 32 B for 8 views of the 100k-triangle mesh at 512^2, i.e. tens of GB of
// cache traffic if every walk reads the maps itself.
//
// LINE-MAJOR formulation.  A walk only ever moves along ONE image line (a row for axis 1, a column for
// axis 0), and a line is shared by hundreds of walks.  So:
//   1. k_edge_count   one lane per face: enumerate crossings exactly as the reference does, count the
//                     LONG walk segments per face and per line;
//   2. k_alloc_ranges (x2) hand every face / every line a slice of the item arrays (order-free);
//   3. k_edge_emit    one lane per face: enumerate again; walk SHORT segments itself (a handful of
//                     pixels) and write each long segment as a 48-byte item, indexed under its line;
//   4. k_edge_lines   one workgroup per (view, axis, line, part): stage the line's maps in LDS ONCE
//                     (unit stride: column lines read transposed copies of the maps), then one wave
//                     per item walks 64 pixels per iteration out of LDS and wave-reduces;
//   5. k_edge_gather  one lane per face: add its items' results to the short-walk sums and store.
// No global atomics on the gradient, results are deterministic, and the reference's contract
// "overwrite the 9 entries of every front-facing face, leave culled faces alone" (KCU:270,501-502) is
// kept.  Per visited pixel the expressions are those of KCU:385-412 / :473-493; the two divisions
// inside the walk use v_rcp_f32 (1 ulp), far inside the 1e-3 gradient tolerance.
#pragma once
#include "d3m_backward.h"
#include "d3m_face_major.h"
#include "d3m_launch.h"

namespace d3m {

constexpr int EG_INLINE_MAX = 6;   // segments of at most this many pixels are walked by the owning lane
constexpr int EG_LINE_PARTS = 4;   // workgroups per line (items are dealt round-robin to the parts)
constexpr int EG_ITEM_DW = 12;     // dwords per item

// Maps as one scan axis sees them: element (line d0, position d1) lives at b*S*S + d0*S + d1.
struct AxisMaps {
    const int32_t* fi;
    const float* alpha;
    const float* galpha;
    const float* rgb;
    const float* grgb;
};

struct EdgeGradArgs {
    AxisMaps ax[2];   // [0]: axis 0 = column walks (transposed maps); [1]: axis 1 = row walks (original maps)
    int S, use_rgb, use_alpha;
    float eps;
};

struct EdgeWork {
    int* visible;        // [B*F]   1 if the face owns a pixel (zeroed per call, set by k_mark_visible)
    int* face_count;     // [B*F]   long segments per face (zeroed per call)
    int* face_offset;    // [B*F]
    int* line_count;     // [B*2*S] long segments per line (zeroed per call)
    int* line_cursor;    // [B*2*S] (zeroed per call)
    int* line_offset;    // [B*2*S]
    int* alloc;          // [2] cursors of the two range allocations (zeroed per call)
    uint32_t* items;     // [cap * EG_ITEM_DW]
    int* line_items;     // [cap] item indices grouped by line
    float2* results;     // [cap]
    int cap;
};

struct SegRef {
    float alpha, r, g, b;
};

// One walk segment.
struct Segment {
    int axis, slot0, slot1, d0, from, to, inward, f0, f1;
    float d1_cross, q0, q1;
    int ref_pos;     // d1 of the pixel whose value is the reference (in-pixel for outward, out-pixel for inward)
};

// Enumerates every walk segment of one face in the reference's order (edges, then axes, then d0).
// pp = pixel-space x0,y0,x1,y1,x2,y2 (KCU:282).  owner(axis, d0, d1) must return face_index_map at that
// line position; emit(const Segment&) is called for each non-empty segment.
template <class Owner, class Emit>
__device__ __forceinline__ void for_each_segment(const float* pp, int fn, int is, Owner&& owner, Emit&& emit) {
#pragma unroll
    for (int edge = 0; edge < 3; edge++) {
        const int i0 = edge, i1 = (edge + 1) % 3, i2 = (edge + 2) % 3;                    // pi[], KCU:278-279
#pragma unroll
        for (int axis = 0; axis < 2; axis++) {
            // p[num][dim] = pp[num][(dim + axis) % 2], KCU:289-294
            const float p00 = pp[2 * i0 + axis], p01 = pp[2 * i0 + 1 - axis];
            const float p10 = pp[2 * i1 + axis], p11 = pp[2 * i1 + 1 - axis];
            const float p20 = pp[2 * i2 + axis], p21 = pp[2 * i2 + 1 - axis];
            const int direction = (axis == 0) ? ((p00 < p10) ? -1 : 1) : ((p00 < p10) ? 1 : -1);   // KCU:297-308
            const int d0_from = f2i(fmaxf(ceilf(fminf(p00, p10)), 0.0f));                          // KCU:312
            const int d0_to = f2i(fminf(fmaxf(p00, p10), (float)(is - 1)));                        // KCU:313
            Segment sg;
            sg.axis = axis;
            sg.slot0 = i0 * 2 + (1 - axis);
            sg.slot1 = i1 * 2 + (1 - axis);
            for (int d0 = d0_from; d0 <= d0_to; d0++) {
                const float fd0 = (float)d0;
                const float d1_cross = (p11 - p01) / (p10 - p00) * (fd0 - p00) + p01;             // KCU:317
                const int d1_in = (0 < direction) ? f2i(floorf(d1_cross)) : f2i(ceilf(d1_cross));
                const int d1_out = (int)((unsigned)d1_in + (unsigned)direction);
                if (d1_in < 0 || is <= d1_in || d1_out < 0 || is <= d1_out) continue;             // KCU:325-328
                sg.d0 = d0;
                sg.d1_cross = d1_cross;
                sg.f0 = p10 != fd0;
                sg.f1 = p00 != fd0;
                sg.q0 = (p10 - p00) / (p10 - fd0);      // KCU:404 / :409: first factor of `dist`
                sg.q1 = (p10 - p00) / (fd0 - p00);
                // outward: out-pixel .. image border, only if the in-pixel belongs to this face (KCU:354-362)
                if (owner(axis, d0, d1_in) == fn) {
                    const int d1_limit = (0 < direction) ? is - 1 : 0;
                    sg.from = max(min(d1_out, d1_limit), 0);
                    sg.to = min(max(d1_out, d1_limit), is - 1);
                    sg.inward = 0;
                    sg.ref_pos = d1_in;
                    emit(sg);
                }
                // inward: in-pixel .. opposite edge (KCU:417-431)
                float d0_cross2;
                if ((fd0 - p00) * (fd0 - p20) < 0) d0_cross2 = (p21 - p01) / (p20 - p00) * (fd0 - p00) + p01;
                else                               d0_cross2 = (p11 - p21) / (p10 - p20) * (fd0 - p20) + p21;
                const int d1_limit = (0 < direction) ? f2i(ceilf(d0_cross2)) : f2i(floorf(d0_cross2));
                sg.from = max(min(d1_in, d1_limit), 0);
                sg.to = min(max(d1_in, d1_limit), is - 1);
                if (sg.from <= sg.to) {
                    sg.inward = 1;
                    sg.ref_pos = d1_out;
                    emit(sg);
                }
            }
        }
    }
}

// Accumulate one visited pixel: KCU:385-412 (outward) / :470-493 (inward).
__device__ __forceinline__ void visit_pixel(float diff, int d1, float d1_cross, float q0, float q1, bool f0, bool f1,
                                            float two_over_is, float eps, float& g0, float& g1) {
    if (diff <= 0) return;
    const float t = (float)d1 - d1_cross;
    if (f0) {
        float dist = q0 * t * two_over_is;
        dist = (0 < dist) ? dist + eps : dist - eps;
        g0 -= diff * __builtin_amdgcn_rcpf(dist);
    }
    if (f1) {
        float dist = q1 * t * two_over_is;
        dist = (0 < dist) ? dist + eps : dist - eps;
        g1 -= diff * __builtin_amdgcn_rcpf(dist);
    }
}

__device__ __forceinline__ SegRef load_ref(const AxisMaps& m, bool use_rgb, bool use_alpha, size_t idx) {
    SegRef r = {0, 0, 0, 0};
    if (use_alpha) r.alpha = m.alpha[idx];
    if (use_rgb) { r.r = m.rgb[3 * idx + 0]; r.g = m.rgb[3 * idx + 1]; r.b = m.rgb[3 * idx + 2]; }
    return r;
}

// short segment, walked straight from global memory by the owning lane
__device__ __forceinline__ void walk_inline(const AxisMaps& m, bool use_rgb, bool use_alpha, size_t line_base,
                                            const Segment& sg, int fn, const SegRef& ref, float two_over_is, float eps,
                                            float& g0, float& g1) {
    for (int d1 = sg.from; d1 <= sg.to; d1++) {
        const size_t idx = line_base + d1;
        if (sg.inward && m.fi[idx] != fn) continue;
        float diff = 0;
        if (use_alpha) diff += (m.alpha[idx] - ref.alpha) * m.galpha[idx];
        if (use_rgb) {
            diff += (m.rgb[3 * idx + 0] - ref.r) * m.grgb[3 * idx + 0];
            diff += (m.rgb[3 * idx + 1] - ref.g) * m.grgb[3 * idx + 1];
            diff += (m.rgb[3 * idx + 2] - ref.b) * m.grgb[3 * idx + 2];
        }
        visit_pixel(diff, d1, sg.d1_cross, sg.q0, sg.q1, sg.f0, sg.f1, two_over_is, eps, g0, g1);
    }
}

// A face that owns no pixel cannot contribute: the outward walk needs the in-pixel to be its own
// (KCU:354) and the inward walk only counts its own pixels (KCU:470).  `visible` makes that a 4-byte test.
template <class FS>
__device__ __forceinline__ bool load_face_pixels(const FS& fs, const int* visible, long gi, int B, int is, int& bn,
                                                 int& fn, float* pp, float* zero_out) {
    const int F = fs.num_faces();
    if (gi >= (long)B * F) return false;
    bn = (int)(gi / F);
    fn = (int)(gi % F);
    const bool vis = visible[gi] != 0;
    if (!vis && !zero_out) return false;
    float face[9];
    fs.load(bn, fn, face);
    if (backside(face)) return false;                      // KCU:270: culled faces are left untouched
    if (!vis) {                                            // front-facing but hidden: the reference stores zeros
#pragma unroll
        for (int k = 0; k < 9; k++) zero_out[(size_t)gi * 9 + k] = 0.0f;
        return false;
    }
#pragma unroll
    for (int n = 0; n < 3; n++) {
        pp[2 * n + 0] = to_pixel(face[3 * n + 0], is);     // KCU:282
        pp[2 * n + 1] = to_pixel(face[3 * n + 1], is);
    }
    return true;
}

// ---- 1. count long segments per face and per line --------------------------------------------------
// face_count[gi] = 1 + number of long segments for front-facing faces, 0 for culled ones (so that the
// gather pass can tell "front-facing without items" from "culled").
template <class FS>
__global__ void __launch_bounds__(256) k_edge_count(FS fs, EdgeGradArgs a, EdgeWork w, int B) {
    const long gi = (long)blockIdx.x * 256 + threadIdx.x;
    int bn, fn;
    float pp[6];
    if (!load_face_pixels(fs, w.visible, gi, B, a.S, bn, fn, pp, nullptr)) return;
    const int is = a.S;
    const size_t base = (size_t)bn * is * is;
    int n = 0;
    for_each_segment(
        pp, fn, is, [&](int axis, int d0, int d1) { return a.ax[axis].fi[base + (size_t)d0 * is + d1]; },
        [&](const Segment& sg) {
            if (sg.to - sg.from + 1 > EG_INLINE_MAX) {
                n++;
                atomicAdd(&w.line_count[((size_t)bn * 2 + sg.axis) * is + sg.d0], 1);
            }
        });
    w.face_count[gi] = n;
}

// ---- 2. order-free range allocation: offsets[i] = slice start for counts[i] (one atomic per 256) ----
__global__ void __launch_bounds__(256) k_alloc_ranges(const int* __restrict__ counts, int* __restrict__ offsets,
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
    if (i < n) offsets[i] = s_base + s_wave[wv] + incl - c;
}

// ---- 3. walk short segments, emit long ones ------------------------------------------------------------
// Item (12 dwords): 0 bits = slot0[0:3) slot1[3:6) inward[6] axis[7] f0[8] f1[9]; 1 d0 | b<<16;
// 2 from | to<<16; 3 fn; 4 d1_cross; 5 q0; 6 q1; 7..10 reference alpha,r,g,b; 11 unused.
template <class FS>
__global__ void __launch_bounds__(256) k_edge_emit(FS fs, EdgeGradArgs a, EdgeWork w, float* __restrict__ grad_faces, int B) {
    const long gi = (long)blockIdx.x * 256 + threadIdx.x;
    int bn, fn;
    float pp[6];
    if (!load_face_pixels(fs, w.visible, gi, B, a.S, bn, fn, pp, grad_faces)) return;
    const int is = a.S;
    const float two_over_is = 2.0f / (float)is;
    const bool use_rgb = a.use_rgb != 0, use_alpha = a.use_alpha != 0;
    const size_t base = (size_t)bn * is * is;
    float acc[6] = {0, 0, 0, 0, 0, 0};
    int k = 0;
    const int my_offset = w.face_offset[gi];
    for_each_segment(
        pp, fn, is, [&](int axis, int d0, int d1) { return a.ax[axis].fi[base + (size_t)d0 * is + d1]; },
        [&](const Segment& sg) {
            const AxisMaps& m = a.ax[sg.axis];
            const size_t line_base = base + (size_t)sg.d0 * is;
            const SegRef ref = load_ref(m, use_rgb, use_alpha, line_base + sg.ref_pos);
            const bool is_long = sg.to - sg.from + 1 > EG_INLINE_MAX;
            const int item = my_offset + k;
            if (is_long) k++;
            const size_t line = ((size_t)bn * 2 + sg.axis) * is + sg.d0;
            const uint32_t bits = (uint32_t)sg.slot0 | ((uint32_t)sg.slot1 << 3) | ((uint32_t)sg.inward << 6) |
                                  ((uint32_t)sg.axis << 7) | ((uint32_t)sg.f0 << 8) | ((uint32_t)sg.f1 << 9);
            // A long segment is queued only if both its item slot and its line's whole slice fit the
            // capacity the workspace gives; otherwise this lane walks it (still correct, just serial).
            const bool queued = is_long && item < w.cap &&
                                (long)w.line_offset[line] + w.line_count[line] <= (long)w.cap;
            if (!queued) {
                if (is_long && item < w.cap) {      // keep the gather pass well-defined for this slot
                    w.items[(size_t)item * EG_ITEM_DW] = bits;
                    w.results[item] = make_float2(0.0f, 0.0f);
                }
                float g0 = 0, g1 = 0;
                walk_inline(m, use_rgb, use_alpha, line_base, sg, fn, ref, two_over_is, a.eps, g0, g1);
                // slot indices are compile-time constants after for_each_segment is unrolled
#pragma unroll
                for (int s = 0; s < 6; s++) {
                    if (s == sg.slot0) acc[s] += g0;
                    if (s == sg.slot1) acc[s] += g1;
                }
                return;
            }
            uint4* q = (uint4*)(w.items + (size_t)item * EG_ITEM_DW);
            q[0] = make_uint4(bits, (uint32_t)sg.d0 | ((uint32_t)bn << 16), (uint32_t)sg.from | ((uint32_t)sg.to << 16),
                              (uint32_t)fn);
            q[1] = make_uint4(__float_as_uint(sg.d1_cross), __float_as_uint(sg.q0), __float_as_uint(sg.q1),
                              __float_as_uint(ref.alpha));
            q[2] = make_uint4(__float_as_uint(ref.r), __float_as_uint(ref.g), __float_as_uint(ref.b), 0u);
            const int pos = atomicAdd(&w.line_cursor[line], 1);
            w.line_items[(size_t)w.line_offset[line] + pos] = item;
        });
    float* gf = grad_faces + (size_t)gi * 9;
#pragma unroll
    for (int n = 0; n < 3; n++) {
        gf[3 * n + 0] = acc[2 * n + 0];
        gf[3 * n + 1] = acc[2 * n + 1];
        gf[3 * n + 2] = 0.0f;
    }
}

// ---- 4. one workgroup per (view, axis, line, part) ---------------------------------------------------------
// LDS: the line's maps, structure of arrays, 9 x S floats (fi, alpha, galpha, r, g, b, gr, gg, gb).
__global__ void __launch_bounds__(256) k_edge_lines(EdgeGradArgs a, EdgeWork w) {
    extern __shared__ __attribute__((aligned(16))) float s_line[];
    const int is = a.S;
    const int part = blockIdx.x % EG_LINE_PARTS;
    const size_t line = blockIdx.x / EG_LINE_PARTS;          // (b*2 + axis)*S + d0
    // items of this line that fit the capacity: a line's slice may straddle cap; entries past it were
    // never written (k_edge_emit walked those segments itself), and cursor counts only written ones
    const int n_items = w.line_cursor[line];
    if (part * 4 >= n_items) return;                          // nothing for this workgroup (uniform exit)
    const int wv = threadIdx.x >> 6, lane = lane_id();
    const int d0 = (int)(line % is);
    const int axis = (int)((line / is) & 1);
    const size_t bn = line / ((size_t)2 * is);
    const AxisMaps& m = a.ax[axis];
    const size_t line_base = bn * is * is + (size_t)d0 * is;
    const bool use_rgb = a.use_rgb != 0, use_alpha = a.use_alpha != 0;
    int* s_fi = (int*)s_line;
    float* s_alpha = s_line + is;
    float* s_galpha = s_line + 2 * is;
    float* s_rgb = s_line + 3 * is;      // r | g | b planes
    float* s_grgb = s_line + 6 * is;
    for (int p = threadIdx.x; p < is; p += 256) {
        s_fi[p] = m.fi[line_base + p];
        if (use_alpha) { s_alpha[p] = m.alpha[line_base + p]; s_galpha[p] = m.galpha[line_base + p]; }
    }
    if (use_rgb) {
        for (int e = threadIdx.x; e < 3 * is; e += 256) {      // coalesced over the interleaved rgb triplets
            const int p = e / 3, c = e - 3 * p;
            s_rgb[c * is + p] = m.rgb[3 * line_base + e];
            s_grgb[c * is + p] = m.grgb[3 * line_base + e];
        }
    }
    __syncthreads();
    const float two_over_is = 2.0f / (float)is;
    const int* list = w.line_items + w.line_offset[line];
    for (int it = part * 4 + wv; it < n_items; it += 4 * EG_LINE_PARTS) {
        const int item = list[it];
        const uint4* q = (const uint4*)(w.items + (size_t)item * EG_ITEM_DW);
        const uint4 q0v = q[0], q1v = q[1], q2v = q[2];
        const uint32_t bits = q0v.x;
        const int from = (int)(q0v.z & 0xFFFF), to = (int)(q0v.z >> 16), fn = (int)q0v.w;
        const bool inward = (bits >> 6) & 1, f0 = (bits >> 8) & 1, f1 = (bits >> 9) & 1;
        const float d1_cross = __uint_as_float(q1v.x), qq0 = __uint_as_float(q1v.y), qq1 = __uint_as_float(q1v.z);
        const float ra = __uint_as_float(q1v.w), rr = __uint_as_float(q2v.x), rg = __uint_as_float(q2v.y),
                    rb = __uint_as_float(q2v.z);
        float g0 = 0, g1 = 0;
        for (int d1 = from + lane; d1 <= to; d1 += 64) {
            if (inward && s_fi[d1] != fn) continue;
            float diff = 0;
            if (use_alpha) diff += (s_alpha[d1] - ra) * s_galpha[d1];
            if (use_rgb) {
                diff += (s_rgb[d1] - rr) * s_grgb[d1];
                diff += (s_rgb[is + d1] - rg) * s_grgb[is + d1];
                diff += (s_rgb[2 * is + d1] - rb) * s_grgb[2 * is + d1];
            }
            visit_pixel(diff, d1, d1_cross, qq0, qq1, f0, f1, two_over_is, a.eps, g0, g1);
        }
        g0 = wave_sum(g0);
        g1 = wave_sum(g1);
        if (lane == 0) w.results[item] = make_float2(g0, g1);
    }
}

// ---- 5. per face: short-walk sums (already in grad_faces) + its items' results ------------------------
template <class FS>
__global__ void __launch_bounds__(256) k_edge_gather(FS fs, EdgeWork w, float* __restrict__ grad_faces, int B) {
    const long gi = (long)blockIdx.x * 256 + threadIdx.x;
    if (gi >= (long)B * fs.num_faces()) return;
    const int n = w.face_count[gi];
    if (n == 0) return;                           // no long segments (or culled: count stays 0): nothing to add
    const int off = w.face_offset[gi];
    float acc[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < n; k++) {
        const int item = off + k;
        if (item >= w.cap) break;                 // those were walked inline by k_edge_emit
        const uint32_t bits = w.items[(size_t)item * EG_ITEM_DW];
        const float2 r = w.results[item];
        const int s0 = bits & 7, s1 = (bits >> 3) & 7;
#pragma unroll
        for (int s = 0; s < 6; s++) {
            if (s == s0) acc[s] += r.x;
            if (s == s1) acc[s] += r.y;
        }
    }
    float* gf = grad_faces + (size_t)gi * 9;
#pragma unroll
    for (int v = 0; v < 3; v++) {
        gf[3 * v + 0] += acc[2 * v + 0];
        gf[3 * v + 1] += acc[2 * v + 1];
    }
}

// [B,S,S,C] -> [B,S(x),S(y),C] through a 32x33 LDS tile; 4-byte elements (f32 or i32 bit patterns).
__global__ void __launch_bounds__(256) k_transpose_map(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int S,
                                                      int C) {
    __shared__ uint32_t tile[32][33];
    const int bc = blockIdx.z, b = bc / C, c = bc % C;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const size_t plane = (size_t)b * S * S;
    for (int r = ty; r < 32; r += 8) {
        const int y = y0 + r, x = x0 + tx;
        if (y < S && x < S) tile[r][tx] = src[(plane + (size_t)y * S + x) * C + c];
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int x = x0 + r, y = y0 + tx;
        if (x < S && y < S) dst[(plane + (size_t)x * S + y) * C + c] = tile[tx][r];
    }
}

// ---- host side ----------------------------------------------------------------------------------------
struct EdgeLayout {
    size_t off_fiT, off_alphaT, off_galphaT, off_rgbT, off_grgbT;
    size_t off_zero, zero_bytes;   // face_count | line_count | line_cursor | alloc
    size_t off_visible, off_face_count, off_line_count, off_line_cursor, off_alloc, off_face_offset, off_line_offset;
    size_t off_items;              // items | line_items | results follow, sized by capacity
    size_t fixed_bytes;
};

inline size_t eg_align(size_t v) { return (v + 255) / 256 * 256; }

inline EdgeLayout edge_layout(int B, int F, int S) {
    const size_t px = (size_t)B * S * S, nf = (size_t)B * F, nl = (size_t)B * 2 * S;
    EdgeLayout L;
    size_t o = 0;
    L.off_fiT = o;      o += eg_align(px * 4);
    L.off_alphaT = o;   o += eg_align(px * 4);
    L.off_galphaT = o;  o += eg_align(px * 4);
    L.off_rgbT = o;     o += eg_align(px * 12);
    L.off_grgbT = o;    o += eg_align(px * 12);
    L.off_zero = o;
    L.off_visible = o;      o += eg_align(nf * 4);
    L.off_face_count = o;   o += eg_align(nf * 4);
    L.off_line_count = o;   o += eg_align(nl * 4);
    L.off_line_cursor = o;  o += eg_align(nl * 4);
    L.off_alloc = o;        o += 256;
    L.zero_bytes = o - L.off_zero;
    L.off_face_offset = o;  o += eg_align(nf * 4);
    L.off_line_offset = o;  o += eg_align(nl * 4);
    L.off_items = o;
    L.fixed_bytes = o;
    return L;
}

constexpr size_t EG_BYTES_PER_ITEM = EG_ITEM_DW * 4 + 4 + 8;
constexpr int EG_ITEMS_PER_FACE_DEFAULT = 4;

inline size_t edge_grad_workspace_bytes(int B, int F, int S) {
    return edge_layout(B, F, S).fixed_bytes + eg_align((size_t)EG_ITEMS_PER_FACE_DEFAULT * B * F * EG_BYTES_PER_ITEM) + 1024;
}

template <class FS>
int run_edge_grad(FS fs, PixelMaps m, float* grad_faces, int B, float eps, void* ws, size_t ws_bytes, hipStream_t st,
                  int* last_err) {
    const int S = m.S, F = fs.num_faces();
    if (S > 65535 || B > 65535) return 1;                       // item packing limits (D3M_ERR_INVALID)
    const EdgeLayout L = edge_layout(B, F, S);
    if (!ws || ws_bytes < L.fixed_bytes + 1024) return 2;       // D3M_ERR_WORKSPACE
    char* p = (char*)ws;
    size_t cap = (ws_bytes - L.fixed_bytes - 768) / EG_BYTES_PER_ITEM;
    cap = cap > 256 ? cap - 128 : 0;                            // slack for the two 256-byte alignments below
    if (cap > 0x7FFFFF00) cap = 0x7FFFFF00;
    EdgeWork w;
    w.visible = (int*)(p + L.off_visible);
    w.face_count = (int*)(p + L.off_face_count);
    w.line_count = (int*)(p + L.off_line_count);
    w.line_cursor = (int*)(p + L.off_line_cursor);
    w.alloc = (int*)(p + L.off_alloc);
    w.face_offset = (int*)(p + L.off_face_offset);
    w.line_offset = (int*)(p + L.off_line_offset);
    w.items = (uint32_t*)(p + L.off_items);
    const size_t off_list = eg_align(L.off_items + cap * EG_ITEM_DW * 4);
    const size_t off_res = eg_align(off_list + cap * 4);
    w.line_items = (int*)(p + off_list);
    w.results = (float2*)(p + off_res);
    w.cap = (int)cap;

    int32_t* fiT = (int32_t*)(p + L.off_fiT);
    float* alphaT = (float*)(p + L.off_alphaT);
    float* galphaT = (float*)(p + L.off_galphaT);
    float* rgbT = (float*)(p + L.off_rgbT);
    float* grgbT = (float*)(p + L.off_grgbT);
    const dim3 grid1((S + 31) / 32, (S + 31) / 32, B), grid3((S + 31) / 32, (S + 31) / 32, B * 3);
    hipError_t e = hipMemsetAsync(p + L.off_zero, 0, L.zero_bytes, st);
    if (e != hipSuccess) { *last_err = (int)e; return 3; }
    LAUNCH("k_transpose_map", k_transpose_map, grid1, dim3(256), st, (const uint32_t*)m.face_index_map, (uint32_t*)fiT, S, 1);
    if (m.use_alpha) {
        LAUNCH("k_transpose_map", k_transpose_map, grid1, dim3(256), st, (const uint32_t*)m.alpha_map, (uint32_t*)alphaT, S, 1);
        LAUNCH("k_transpose_map", k_transpose_map, grid1, dim3(256), st, (const uint32_t*)m.grad_alpha_map, (uint32_t*)galphaT, S, 1);
    }
    if (m.use_rgb) {
        LAUNCH("k_transpose_map", k_transpose_map, grid3, dim3(256), st, (const uint32_t*)m.rgb_map, (uint32_t*)rgbT, S, 3);
        LAUNCH("k_transpose_map", k_transpose_map, grid3, dim3(256), st, (const uint32_t*)m.grad_rgb_map, (uint32_t*)grgbT, S, 3);
    }
    EdgeGradArgs a;
    a.ax[0] = AxisMaps{fiT, alphaT, galphaT, rgbT, grgbT};
    a.ax[1] = AxisMaps{m.face_index_map, m.alpha_map, m.grad_alpha_map, m.rgb_map, m.grad_rgb_map};
    a.S = S; a.use_rgb = m.use_rgb; a.use_alpha = m.use_alpha; a.eps = eps;
    const long nf = (long)B * F, nl = (long)B * 2 * S;
    const dim3 gf((unsigned)((nf + 255) / 256)), gl((unsigned)((nl + 255) / 256));
    LAUNCH("k_mark_visible", k_mark_visible, dim3((unsigned)(((long)B * S * S + 255) / 256)), dim3(256), st,
           m.face_index_map, w.visible, B, F, S);
    LAUNCH("k_edge_count", k_edge_count<FS>, gf, dim3(256), st, fs, a, w, B);
    LAUNCH("k_alloc_ranges", k_alloc_ranges, gf, dim3(256), st, (const int*)w.face_count, w.face_offset, w.alloc, nf);
    LAUNCH("k_alloc_ranges", k_alloc_ranges, gl, dim3(256), st, (const int*)w.line_count, w.line_offset, w.alloc + 1, nl);
    LAUNCH("k_edge_emit", k_edge_emit<FS>, gf, dim3(256), st, fs, a, w, grad_faces, B);
    const size_t smem = (size_t)9 * S * 4;
    if (smem > 64 * 1024) {
        if (smem > 160 * 1024) return 1;                        // line does not fit LDS (S > 4551)
        e = hipFuncSetAttribute((const void*)k_edge_lines, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) { *last_err = (int)e; return 3; }
    }
    LAUNCH_SMEM("k_edge_lines", k_edge_lines, dim3((unsigned)(nl * EG_LINE_PARTS)), dim3(256), smem, st, a, w);
    LAUNCH("k_edge_gather", k_edge_gather<FS>, gf, dim3(256), st, fs, w, grad_faces, B);
    e = hipGetLastError();
    if (e != hipSuccess) { *last_err = (int)e; return 3; }
    return 0;
}

}  // namespace d3m
