// d3m_edge_grad.h -- the edge / silhouette gradient (reference: backward_pixel_map_cuda_kernel,
// KCU:245-503) restructured for gfx950.
//
// The reference runs one thread per face; for every integer crossing d0 of every edge, on both axes,
// that thread walks the image from the edge out to the image BORDER (KCU:354-414) and inward to the
// opposite edge (KCU:417-495).  The outward walks are hundreds of pixels long whatever the triangle
// size, so one thread per face is both divergent and serial, and the column walks stride by a whole
// image row.
//
// Here a 256-thread workgroup owns 256 consecutive faces and alternates two phases:
//   A. each lane enumerates the crossings of its own face exactly as the reference does, walks the
//      SHORT segments itself, and posts every long segment as a 48-byte item in an LDS queue;
//   B. the four waves drain the queue: one wave per item, 64 consecutive pixels per iteration
//      (coalesced), wave-reduce, one LDS atomic per touched vertex component.
// Column walks read TRANSPOSED copies of the maps, so both axes are unit-stride.  Per-face sums are
// kept in LDS and written once: no global atomics, and the reference's "overwrite the 9 entries of
// every front-facing face" contract (KCU:501-502) is kept.
//
// Arithmetic: per visited pixel the same expressions as KCU:385-412 / :473-493; the divisions inside
// the walk use v_rcp_f32 (1 ulp), far inside the 1e-3 gradient tolerance; sums are re-associated.
#pragma once
#include "d3m_backward.h"
#include "d3m_launch.h"

namespace d3m {

constexpr int EG_BLOCK = 256;
constexpr int EG_QCAP = 384;      // queue items per round (12 dwords each: 18 KB of LDS)
constexpr int EG_MAX_ROUNDS = 1 << 14;
constexpr int EG_INLINE_MAX = 6;  // segments of at most this many pixels are walked by the owning lane

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

struct SegRef {
    float alpha, r, g, b;
};

// Walk positions start, start+stride, ... <= stop of one line; the body of KCU:384-413 (outward) or
// KCU:460-494 (inward, `fn >= 0`: only pixels owned by face fn contribute).
__device__ __forceinline__ void walk_segment(const AxisMaps& m, bool use_rgb, bool use_alpha, size_t line_base, int start,
                                             int stop, int stride, int fn_in, const SegRef& ref, float d1_cross, float q0,
                                             float q1, bool f0, bool f1, float two_over_is, float eps, float& g0,
                                             float& g1) {
    for (int d1 = start; d1 <= stop; d1 += stride) {
        const size_t idx = line_base + d1;
        if (fn_in >= 0 && m.fi[idx] != fn_in) continue;
        float diff = 0;
        if (use_alpha) diff += (m.alpha[idx] - ref.alpha) * m.galpha[idx];
        if (use_rgb) {
            diff += (m.rgb[3 * idx + 0] - ref.r) * m.grgb[3 * idx + 0];
            diff += (m.rgb[3 * idx + 1] - ref.g) * m.grgb[3 * idx + 1];
            diff += (m.rgb[3 * idx + 2] - ref.b) * m.grgb[3 * idx + 2];
        }
        if (diff <= 0) continue;
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
}

// LDS queue item = 13 dwords: bits (lf[0:8) slot0[8:11) slot1[11:14) inward[14] axis[15] f0[16] f1[17]),
// b, d0, from, to, fn, d1_cross, q0, q1, ref alpha, ref r, ref g, ref b.
template <class FS>
__global__ void __launch_bounds__(EG_BLOCK) k_edge_grad(FS fs, EdgeGradArgs a, float* __restrict__ grad_faces, int B) {
    __shared__ float s_pp[6][EG_BLOCK];          // pixel-space x0,y0,x1,y1,x2,y2 per face
    __shared__ float s_acc[6][EG_BLOCK];         // per face: (vertex, x|y) gradient sums
    __shared__ uint32_t s_q[EG_QCAP][13];        // 13 dwords: odd stride, conflict-free lane-per-item writes
    __shared__ int s_qcount, s_qhead, s_more;

    const int F = fs.num_faces();
    const long gi = (long)blockIdx.x * EG_BLOCK + threadIdx.x;
    const int lf = threadIdx.x;
    const int is = a.S;
    const float two_over_is = 2.0f / (float)is;
    const bool use_rgb = a.use_rgb != 0, use_alpha = a.use_alpha != 0;
    bool active = gi < (long)B * F;
    int bn = 0, fn = 0;
    if (active) {
        bn = (int)(gi / F);
        fn = (int)(gi % F);
        float face[9];
        fs.load(bn, fn, face);
        if (backside(face)) active = false;                          // KCU:270: culled faces are left untouched
        else {
#pragma unroll
            for (int n = 0; n < 3; n++) {
                s_pp[2 * n + 0][lf] = to_pixel(face[3 * n + 0], is);  // KCU:282
                s_pp[2 * n + 1][lf] = to_pixel(face[3 * n + 1], is);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 6; k++) s_acc[k][lf] = 0;
    const size_t base = (size_t)bn * is * is;

    // resumable enumeration state of this lane
    int ea = 0;            // edge*2 + axis, 0..5
    int d0 = 0, d0_to = -1;
    int stage = 0;         // 0: crossing not started, 1: outward part done, inward part pending
    bool have = false, done = !active;
    float p00 = 0, p01 = 0, p10 = 0, p11 = 0, p20 = 0, p21 = 0;
    int direction = 1, slot0 = 0, slot1 = 0, axis = 0;

    // Every loop below is bounded (the rounds by EG_MAX_ROUNDS, a lane's enumeration by the number of
    // crossings a face can have), so a logic error shows up as a NaN in the z slot instead of a hung GPU.
    bool failed = false;
    const long step_cap = 6L * is + 64;     // crossings a face can have, plus slack
    long steps = 0;
    for (int round = 0; round < EG_MAX_ROUNDS; round++) {
        if (threadIdx.x == 0) { s_qcount = 0; s_qhead = 0; s_more = 0; }
        __syncthreads();

        // ---------------- phase A: enumerate crossings ----------------
        bool full = false;
        while (!done && !full) {
            if (++steps > step_cap) { failed = true; done = true; break; }   // (a retry after a full queue is not counted)
            if (!have) {
                if (ea == 6) { done = true; break; }
                const int edge = ea >> 1;
                axis = ea & 1;
                const int i0 = edge, i1 = (edge + 1) % 3, i2 = (edge + 2) % 3;   // pi[], KCU:278-279
                p00 = s_pp[2 * i0 + axis][lf]; p01 = s_pp[2 * i0 + 1 - axis][lf]; // p[num][dim] = pp[num][(dim+axis)%2]
                p10 = s_pp[2 * i1 + axis][lf]; p11 = s_pp[2 * i1 + 1 - axis][lf];
                p20 = s_pp[2 * i2 + axis][lf]; p21 = s_pp[2 * i2 + 1 - axis][lf];
                direction = (axis == 0) ? ((p00 < p10) ? -1 : 1) : ((p00 < p10) ? 1 : -1);   // KCU:297-308
                d0 = f2i(fmaxf(ceilf(fminf(p00, p10)), 0.0f));                               // KCU:312
                d0_to = f2i(fminf(fmaxf(p00, p10), (float)(is - 1)));                         // KCU:313
                slot0 = i0 * 2 + (1 - axis);
                slot1 = i1 * 2 + (1 - axis);
                have = true;
                stage = 0;
            }
            if (d0 > d0_to) { have = false; ea++; continue; }
            const AxisMaps& m = a.ax[axis];
            const float fd0 = (float)d0;
            const float d1_cross = (p11 - p01) / (p10 - p00) * (fd0 - p00) + p01;             // KCU:317
            const int d1_in = (0 < direction) ? f2i(floorf(d1_cross)) : f2i(ceilf(d1_cross));
            const int d1_out = (int)((unsigned)d1_in + (unsigned)direction);
            if (d1_in < 0 || is <= d1_in || d1_out < 0 || is <= d1_out) { d0++; stage = 0; continue; }   // KCU:325-328
            const size_t line_base = base + (size_t)d0 * is;
            const bool f0 = p10 != fd0, f1 = p00 != fd0;
            const float q0 = (p10 - p00) / (p10 - fd0), q1 = (p10 - p00) / (fd0 - p00);     // KCU:404 / :409 prefixes

            if (stage == 0) {
                // outward: from the out-pixel to the border, only if the in-pixel belongs to this face (KCU:354)
                if (m.fi[line_base + d1_in] == fn) {
                    const int d1_limit = (0 < direction) ? is - 1 : 0;
                    const int from = max(min(d1_out, d1_limit), 0), to = min(max(d1_out, d1_limit), is - 1);
                    SegRef ref = {0, 0, 0, 0};
                    if (use_alpha) ref.alpha = m.alpha[line_base + d1_in];
                    if (use_rgb) {
                        ref.r = m.rgb[3 * (line_base + d1_in) + 0];
                        ref.g = m.rgb[3 * (line_base + d1_in) + 1];
                        ref.b = m.rgb[3 * (line_base + d1_in) + 2];
                    }
                    if (to - from + 1 <= EG_INLINE_MAX) {
                        float g0 = 0, g1 = 0;
                        walk_segment(m, use_rgb, use_alpha, line_base, from, to, 1, -1, ref, d1_cross, q0, q1, f0, f1,
                                     two_over_is, a.eps, g0, g1);
                        s_acc[slot0][lf] += g0;
                        s_acc[slot1][lf] += g1;
                    } else {
                        const int pos = atomicAdd(&s_qcount, 1);
                        if (pos >= EG_QCAP) { full = true; steps--; break; }
                        uint32_t* q = s_q[pos];
                        q[0] = (uint32_t)lf | ((uint32_t)slot0 << 8) | ((uint32_t)slot1 << 11) | (0u << 14) |
                               ((uint32_t)axis << 15) | ((uint32_t)f0 << 16) | ((uint32_t)f1 << 17);
                        q[1] = (uint32_t)bn; q[2] = (uint32_t)d0; q[3] = (uint32_t)from; q[4] = (uint32_t)to; q[5] = (uint32_t)fn;
                        q[6] = __float_as_uint(d1_cross); q[7] = __float_as_uint(q0); q[8] = __float_as_uint(q1);
                        q[9] = __float_as_uint(ref.alpha); q[10] = __float_as_uint(ref.r); q[11] = __float_as_uint(ref.g);
                        q[12] = __float_as_uint(ref.b);
                    }
                }
                stage = 1;
            }
            {
                // inward: from the in-pixel to the opposite edge (KCU:417-495); reference value = the out-pixel
                float d0_cross2;
                if ((fd0 - p00) * (fd0 - p20) < 0) d0_cross2 = (p21 - p01) / (p20 - p00) * (fd0 - p00) + p01;
                else                               d0_cross2 = (p11 - p21) / (p10 - p20) * (fd0 - p20) + p21;
                const int d1_limit = (0 < direction) ? f2i(ceilf(d0_cross2)) : f2i(floorf(d0_cross2));
                const int from = max(min(d1_in, d1_limit), 0), to = min(max(d1_in, d1_limit), is - 1);
                if (from <= to) {
                    SegRef ref = {0, 0, 0, 0};
                    if (use_alpha) ref.alpha = m.alpha[line_base + d1_out];
                    if (use_rgb) {
                        ref.r = m.rgb[3 * (line_base + d1_out) + 0];
                        ref.g = m.rgb[3 * (line_base + d1_out) + 1];
                        ref.b = m.rgb[3 * (line_base + d1_out) + 2];
                    }
                    if (to - from + 1 <= EG_INLINE_MAX) {
                        float g0 = 0, g1 = 0;
                        walk_segment(m, use_rgb, use_alpha, line_base, from, to, 1, fn, ref, d1_cross, q0, q1, f0, f1,
                                     two_over_is, a.eps, g0, g1);
                        s_acc[slot0][lf] += g0;
                        s_acc[slot1][lf] += g1;
                    } else {
                        const int pos = atomicAdd(&s_qcount, 1);
                        if (pos >= EG_QCAP) { full = true; steps--; break; }     // stage stays 1: resume at the inward part
                        uint32_t* q = s_q[pos];
                        q[0] = (uint32_t)lf | ((uint32_t)slot0 << 8) | ((uint32_t)slot1 << 11) | (1u << 14) |
                               ((uint32_t)axis << 15) | ((uint32_t)f0 << 16) | ((uint32_t)f1 << 17);
                        q[1] = (uint32_t)bn; q[2] = (uint32_t)d0; q[3] = (uint32_t)from; q[4] = (uint32_t)to; q[5] = (uint32_t)fn;
                        q[6] = __float_as_uint(d1_cross); q[7] = __float_as_uint(q0); q[8] = __float_as_uint(q1);
                        q[9] = __float_as_uint(ref.alpha); q[10] = __float_as_uint(ref.r); q[11] = __float_as_uint(ref.g);
                        q[12] = __float_as_uint(ref.b);
                    }
                }
            }
            d0++;
            stage = 0;
        }
        if (!done) s_more = 1;
        __syncthreads();

        // ---------------- phase B: one wave per queued segment ----------------
        const int n_items = min(s_qcount, EG_QCAP);
        const bool more = s_more != 0;
        const int lane = lane_id();
        for (int guard = 0; guard <= EG_QCAP; guard++) {
            int it = 0;
            if (lane == 0) it = atomicAdd(&s_qhead, 1);
            it = __shfl(it, 0, 64);
            if (it >= n_items) break;
            const uint32_t* q = s_q[it];
            const uint32_t bits = q[0];
            const int qaxis = (bits >> 15) & 1;
            const bool inward = (bits >> 14) & 1;
            const size_t line_base = (size_t)q[1] * is * is + (size_t)q[2] * is;
            SegRef ref = {__uint_as_float(q[9]), __uint_as_float(q[10]), __uint_as_float(q[11]), __uint_as_float(q[12])};
            float g0 = 0, g1 = 0;
            walk_segment(a.ax[qaxis], use_rgb, use_alpha, line_base, (int)q[3] + lane, (int)q[4], 64,
                         inward ? (int)q[5] : -1, ref, __uint_as_float(q[6]), __uint_as_float(q[7]), __uint_as_float(q[8]),
                         (bits >> 16) & 1, (bits >> 17) & 1, two_over_is, a.eps, g0, g1);
            g0 = wave_sum(g0);
            g1 = wave_sum(g1);
            if (lane == 0) {
                atomicAdd(&s_acc[(bits >> 8) & 7][bits & 255], g0);
                atomicAdd(&s_acc[(bits >> 11) & 7][bits & 255], g1);
            }
        }
        __syncthreads();
        if (!more) break;
        if (round == EG_MAX_ROUNDS - 1) failed = true;
    }

    if (active) {
        float* gf = grad_faces + (size_t)gi * 9;
#pragma unroll
        for (int n = 0; n < 3; n++) {
            gf[3 * n + 0] = s_acc[2 * n + 0][lf];
            gf[3 * n + 1] = s_acc[2 * n + 1][lf];
            gf[3 * n + 2] = failed ? __uint_as_float(0x7FC00000u) : 0.0f;
        }
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

inline size_t edge_grad_workspace_bytes(int B, int F, int S) {
    (void)F;
    // transposed face_index, alpha, grad_alpha (4 B each) and rgb, grad_rgb (12 B each)
    return (size_t)B * S * S * 36 + 1024;
}

template <class FS>
int run_edge_grad(FS fs, PixelMaps m, float* grad_faces, int B, float eps, void* ws, size_t ws_bytes, hipStream_t st,
                  int* last_err) {
    const int S = m.S;
    const size_t px = (size_t)B * S * S;
    if (!ws || ws_bytes < px * 36) return 2;   // D3M_ERR_WORKSPACE
    char* p = (char*)ws;
    int32_t* fiT = (int32_t*)p;            p += px * 4;
    float* alphaT = (float*)p;             p += px * 4;
    float* galphaT = (float*)p;            p += px * 4;
    float* rgbT = (float*)p;               p += px * 12;
    float* grgbT = (float*)p;
    const dim3 grid1((S + 31) / 32, (S + 31) / 32, B), grid3((S + 31) / 32, (S + 31) / 32, B * 3);
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
    const long n = (long)B * fs.num_faces();
    LAUNCH("k_edge_grad", k_edge_grad<FS>, dim3((unsigned)((n + EG_BLOCK - 1) / EG_BLOCK)), dim3(EG_BLOCK), st, fs, a,
           grad_faces, B);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { *last_err = (int)e; return 3; }
    return 0;
}

}  // namespace d3m
