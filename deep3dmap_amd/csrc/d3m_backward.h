// d3m_backward.h -- backward passes of the rasterizer for gfx950.
//
//   k_backward_pixel_map  : KCU:245-503 (edge / silhouette gradient wrt vertex x,y)
//   k_backward_textures   : KCU:506-540
//   k_backward_depth_map  : KCU:543-592
#pragma once
#include "d3m_device.h"

namespace d3m {

// Read-only view of the maps the edge gradient walks over.
struct PixelMaps {
    const int32_t* face_index_map;  // [B,S,S]
    const float* rgb_map;           // [B,S,S,3] or NULL
    const float* alpha_map;         // [B,S,S]   or NULL
    const float* grad_rgb_map;      // [B,S,S,3] or NULL
    const float* grad_alpha_map;    // [B,S,S]   or NULL
    int S;
    int use_rgb, use_alpha;
};

// diff_grad of KCU:385-392 / :473-480: (value(idx) - value(ref pixel)) . grad(idx)
__device__ __forceinline__ float edge_diff_grad(const PixelMaps& m, size_t idx, float alpha_ref, const float* rgb_ref) {
    float d = 0;
    if (m.use_alpha) d += (m.alpha_map[idx] - alpha_ref) * m.grad_alpha_map[idx];
    if (m.use_rgb) {
#pragma unroll
        for (int k = 0; k < 3; k++) d += (m.rgb_map[idx * 3 + k] - rgb_ref[k]) * m.grad_rgb_map[idx * 3 + k];
    }
    return d;
}

// -diff/dist terms of KCU:403-412 / :484-493 for one visited pixel d1 of a scan along d0
__device__ __forceinline__ void edge_accumulate(float diff_grad, float p0x, float p1x, int d0, int d1, float d1_cross,
                                                int is, float eps, float& g0, float& g1) {
    if (p1x != (float)d0) {
        float dist = (float)((double)((p1x - p0x) / (p1x - (float)d0) * ((float)d1 - d1_cross)) * 2. / is);
        dist = (0 < dist) ? dist + eps : dist - eps;
        g0 -= diff_grad / dist;
    }
    if (p0x != (float)d0) {
        float dist = (float)((double)((p1x - p0x) / ((float)d0 - p0x) * ((float)d1 - d1_cross)) * 2. / is);
        dist = (0 < dist) ? dist + eps : dist - eps;
        g1 -= diff_grad / dist;
    }
}

// One lane per face (first correct form; the strip-parallel form lives in d3m_edge_grad.h).
template <class FS>
__global__ void __launch_bounds__(256) k_backward_pixel_map(FS fs, PixelMaps m, float* __restrict__ grad_faces, int B,
                                                           float eps) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int F = fs.num_faces();
    if (i >= (long)B * F) return;
    const int bn = (int)(i / F), fn = (int)(i % F);
    const int is = m.S;
    float face[9];
    fs.load(bn, fn, face);
    if (backside(face)) return;
    float grad_face[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const size_t base = (size_t)bn * is * is;

    for (int edge_num = 0; edge_num < 3; edge_num++) {
        int pi[3];
        float pp[3][2];
        for (int num = 0; num < 3; num++) {
            pi[num] = (edge_num + num) % 3;
            pp[num][0] = to_pixel(face[3 * pi[num] + 0], is);
            pp[num][1] = to_pixel(face[3 * pi[num] + 1], is);
        }
        for (int axis = 0; axis < 2; axis++) {
            float p[3][2];
            for (int num = 0; num < 3; num++) {
                p[num][0] = pp[num][axis];
                p[num][1] = pp[num][1 - axis];
            }
            const int direction = (axis == 0) ? ((p[0][0] < p[1][0]) ? -1 : 1) : ((p[0][0] < p[1][0]) ? 1 : -1);
            const int d0_from = d2i(fmax((double)ceilf(fminf(p[0][0], p[1][0])), 0.));
            const int d0_to = d2i(fmin((double)fmaxf(p[0][0], p[1][0]), is - 1.));
            float g0 = 0, g1 = 0;   // running sums for vertices pi[0], pi[1], component (1 - axis)
            for (int d0 = d0_from; d0 <= d0_to; d0++) {
                const float d1_cross = (p[1][1] - p[0][1]) / (p[1][0] - p[0][0]) * ((float)d0 - p[0][0]) + p[0][1];
                const int d1_in = (0 < direction) ? f2i(floorf(d1_cross)) : f2i(ceilf(d1_cross));
                const int d1_out = (int)((unsigned)d1_in + (unsigned)direction);
                if (d1_in < 0 || is <= d1_in) continue;
                if (d1_out < 0 || is <= d1_out) continue;
                const size_t idx_in = (axis == 0) ? base + (size_t)d1_in * is + d0 : base + (size_t)d0 * is + d1_in;
                const size_t idx_out = (axis == 0) ? base + (size_t)d1_out * is + d0 : base + (size_t)d0 * is + d1_out;
                const size_t step = (axis == 0) ? (size_t)is : 1;
                float alpha_in = 0, alpha_out = 0, rgb_in[3] = {0, 0, 0}, rgb_out[3] = {0, 0, 0};
                if (m.use_alpha) { alpha_in = m.alpha_map[idx_in]; alpha_out = m.alpha_map[idx_out]; }
                if (m.use_rgb) {
                    for (int k = 0; k < 3; k++) { rgb_in[k] = m.rgb_map[idx_in * 3 + k]; rgb_out[k] = m.rgb_map[idx_out * 3 + k]; }
                }
                // out: from the out-pixel to the image border (KCU:354-414)
                if (m.face_index_map[idx_in] == fn) {
                    const int d1_limit = (0 < direction) ? is - 1 : 0;
                    const int d1_from = max(min(d1_out, d1_limit), 0);
                    const int d1_to = min(max(d1_out, d1_limit), is - 1);
                    size_t idx = (axis == 0) ? base + (size_t)d1_from * is + d0 : base + (size_t)d0 * is + d1_from;
                    for (int d1 = d1_from; d1 <= d1_to; d1++, idx += step) {
                        const float dg = edge_diff_grad(m, idx, alpha_in, rgb_in);
                        if (dg <= 0) continue;
                        edge_accumulate(dg, p[0][0], p[1][0], d0, d1, d1_cross, is, eps, g0, g1);
                    }
                }
                // in: from the in-pixel to the opposite edge (KCU:417-495)
                {
                    float d0_cross2;
                    if (((float)d0 - p[0][0]) * ((float)d0 - p[2][0]) < 0)
                        d0_cross2 = (p[2][1] - p[0][1]) / (p[2][0] - p[0][0]) * ((float)d0 - p[0][0]) + p[0][1];
                    else
                        d0_cross2 = (p[1][1] - p[2][1]) / (p[1][0] - p[2][0]) * ((float)d0 - p[2][0]) + p[2][1];
                    const int d1_limit = (0 < direction) ? f2i(ceilf(d0_cross2)) : f2i(floorf(d0_cross2));
                    const int d1_from = max(min(d1_in, d1_limit), 0);
                    const int d1_to = min(max(d1_in, d1_limit), is - 1);
                    size_t idx = (axis == 0) ? base + (size_t)d1_from * is + d0 : base + (size_t)d0 * is + d1_from;
                    for (int d1 = d1_from; d1 <= d1_to; d1++, idx += step) {
                        if (m.face_index_map[idx] != fn) continue;
                        const float dg = edge_diff_grad(m, idx, alpha_out, rgb_out);
                        if (dg <= 0) continue;
                        edge_accumulate(dg, p[0][0], p[1][0], d0, d1, d1_cross, is, eps, g0, g1);
                    }
                }
            }
            grad_face[pi[0] * 3 + (1 - axis)] += g0;
            grad_face[pi[1] * 3 + (1 - axis)] += g1;
        }
    }
#pragma unroll
    for (int k = 0; k < 9; k++) grad_faces[i * 9 + k] = grad_face[k];
}

// KCU:506-540: 8 texels x 3 channels of float atomics per covered pixel.
__global__ void __launch_bounds__(256) k_backward_textures(const int32_t* __restrict__ face_index_map,
                                                          const float* __restrict__ sampling_weight_map,
                                                          const int32_t* __restrict__ sampling_index_map,
                                                          const float* __restrict__ grad_rgb_map,
                                                          float* __restrict__ grad_textures, int B, int F, int S, int ts) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * S * S) return;
    const int fi = face_index_map[i];
    if (fi < 0) return;
    const int bn = (int)(i / ((long)S * S));
    const size_t tex_base = ((size_t)bn * F + fi) * ts * ts * ts * 3;
    const size_t tex_total = (size_t)B * F * ts * ts * ts * 3;   // ts == 1 bleed guard, see k_texture_sampling
    const float g[3] = {grad_rgb_map[3 * i + 0], grad_rgb_map[3 * i + 1], grad_rgb_map[3 * i + 2]};
#pragma unroll
    for (int pn = 0; pn < 8; pn++) {
        const float w = sampling_weight_map[i * 8 + pn];
        const int isc = sampling_index_map[i * 8 + pn];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const size_t ti = tex_base + (size_t)((long)isc * 3 + k);
            if (ti < tex_total) atomicAdd(&grad_textures[ti], w * g[k]);
        }
    }
}

// KCU:543-592: 9 float atomics per covered pixel into its face's gradient.
template <class FS>
__global__ void __launch_bounds__(256) k_backward_depth_map(FS fs, const float* __restrict__ depth_map,
                                                           const int32_t* __restrict__ face_index_map,
                                                           const float* __restrict__ face_inv_map,
                                                           const float* __restrict__ weight_map,
                                                           const float* __restrict__ grad_depth_map,
                                                           float* __restrict__ grad_faces, int B, int S) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * S * S) return;
    const int fn = face_index_map[i];
    if (fn < 0) return;
    const int bn = (int)(i / ((long)S * S));
    const int F = fs.num_faces();
    float face[9], finv[9];
    fs.load(bn, fn, face);
    if (face_inv_map) {
#pragma unroll
        for (int k = 0; k < 9; k++) finv[k] = face_inv_map[i * 9 + k];
    } else {
        face_inverse(face, S, finv);
    }
    const float depth = depth_map[i];
    const float depth2 = depth * depth;
    const float g = grad_depth_map[i];
    float* grad_face = grad_faces + ((size_t)bn * F + fn) * 9;
    float tmp[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int l = 0; l < 3; l++) tmp[k] += -finv[3 * l + k] / face[3 * l + 2];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float wk = weight_map[3 * i + k];
        const float z_k = face[3 * k + 2];
        atomicAdd(&grad_face[3 * k + 0], -g * tmp[0] * wk * depth2 * (float)S / 2.0f);
        atomicAdd(&grad_face[3 * k + 1], -g * tmp[1] * wk * depth2 * (float)S / 2.0f);
        atomicAdd(&grad_face[3 * k + 2], g * wk * depth2 / (z_k * z_k));
    }
}

}  // namespace d3m
