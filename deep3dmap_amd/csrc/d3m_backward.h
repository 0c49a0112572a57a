// d3m_backward.h -- backward passes of the rasterizer for gfx950.
//
//   (edge / silhouette gradient, KCU:245-503: d3m_edge_grad.h)
//   k_backward_textures   : KCU:506-540
//   k_backward_depth_map  : KCU:543-592
#pragma once
#include "d3m_device.h"

namespace d3m {

// Read-only view of the maps the edge gradient walks over (the kernels live in d3m_edge_grad.h).
struct PixelMaps {
    const int32_t* face_index_map;  // [B,S,S]
    const float* rgb_map;           // [B,S,S,3] or NULL
    const float* alpha_map;         // [B,S,S]   or NULL
    const float* grad_rgb_map;      // [B,S,S,3] or NULL
    const float* grad_alpha_map;    // [B,S,S]   or NULL
    int S;
    int use_rgb, use_alpha;
};

// KCU:506-540: 8 texels x 3 channels of float atomics per covered pixel.
__global__ void __launch_bounds__(256) k_backward_textures(const int32_t* __restrict__ face_index_map,
                                                          const float* __restrict__ sampling_weight_map,
                                                          const int32_t* __restrict__ sampling_index_map,
                                                          const float* __restrict__ grad_rgb_map,
                                                          float* __restrict__ grad_textures, int B, int F, int S, int ts,
                                                          const int* __restrict__ only_large) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * S * S) return;
    const int fi = face_index_map[i];
    if (fi < 0) return;
    const int bn = (int)(i / ((long)S * S));
    if (only_large && only_large[(size_t)bn * F + fi] != 2) return;   // the rest was gathered per face
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
__device__ __forceinline__ void backward_depth_map_pixels(FS fs, const float* __restrict__ depth_map,
                                                          const int32_t* __restrict__ face_index_map,
                                                          const float* __restrict__ face_inv_map,
                                                          const float* __restrict__ weight_map,
                                                          const float* __restrict__ grad_depth_map,
                                                          float* __restrict__ grad_faces, int B, int S,
                                                          const int* __restrict__ only_large, VertexTarget vt, GradScale gs,
                                                          int flip_rows = 0) {
    // (a fixed grid striding over the pixels, a wave at a time; a wave whose pixels all belong to ONE face -- the ordinary case
    //  where this pass runs at all: faces too large for the gathered pass -- sums its nine values over the wave and adds them
    //  once: see backward_textures_lit_pixels, d3m_lit.h)
    const long n = (long)B * S * S;
    const int F = fs.num_faces();
    const int lane = (int)(threadIdx.x & 63);
    __shared__ WgSums<9> wg;                // (per face: the workgroup's sums, flushed once -- see WgSums)
    wg.init();
    // (a contiguous run of pixels per workgroup -- a few image rows, i.e. few faces -- in steps of one workgroup)
    const long chunk = ((n + gridDim.x - 1) / gridDim.x + blockDim.x - 1) / blockDim.x * blockDim.x;
    const long run_end = min(n, (long)(blockIdx.x + 1) * chunk);
    for (long i0 = (long)blockIdx.x * chunk + (threadIdx.x & ~63u); i0 < run_end; i0 += blockDim.x) {
    const long i = i0 + lane;
    int fn = -1, bn = 0;
    bool active = i < run_end;
    if (active) { fn = face_index_map[i]; active = fn >= 0; }
    if (active) {
        bn = (int)(i / ((long)S * S));
        if (only_large && only_large[(size_t)bn * F + fn] != 2) active = false;   // the rest was gathered per face
    }
    const unsigned long long act = __builtin_amdgcn_ballot_w64(active);
    if (!act) continue;                                             // (wave-uniform)
    const int key = active ? bn * F + fn : -1;
    float v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (active) {
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
        float s_rgb, s_alpha, s_depth;
        gs.get(s_rgb, s_alpha, s_depth);
        // (flip_rows: the gradient of the OUTPUT image -- its row S-1-y is the map's row y)
        const long row = (i / S) % S;
        const float g = grad_depth_map[flip_rows ? i + ((long)S - 1 - 2 * row) * S : i] * s_depth;
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
            v[3 * k + 0] = -g * tmp[0] * wk * depth2 * (float)S / 2.0f;
            v[3 * k + 1] = -g * tmp[1] * wk * depth2 * (float)S / 2.0f;
            v[3 * k + 2] = g * wk * depth2 / (z_k * z_k);
        }
    }
    // one face of the wave at a time (a wave of 64 consecutive pixels of a row holds one to three), every lane in step
    // (up to four: a wave over micro-triangles holds dozens of faces, and those pixels add for themselves as they always did)
    unsigned long long todo = act;
    for (int round = 0; round < 4 && todo; round++) {               // (wave-uniform)
        const int lead = __builtin_ctzll(todo);
        const int key0 = __builtin_amdgcn_readlane(key, lead);
        const bool mine = active && key == key0;
        todo &= ~__builtin_amdgcn_ballot_w64(mine);
        const int bn_a = key0 / F, fn_a = key0 % F;
        float sv[9];
#pragma unroll
        for (int q = 0; q < 9; q++) sv[q] = wave_sum(mine ? v[q] : 0.0f);
        if (lane == lead && !wg.add(key0, sv)) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float* gk = vt.gv ? vt.vertex(bn_a, fn_a, k) : grad_faces + ((size_t)bn_a * F + fn_a) * 9 + 3 * k;
#pragma unroll
                for (int c = 0; c < 3; c++) atomicAdd(&gk[c], sv[3 * k + c]);
            }
        }
    }
    if ((todo >> lane) & 1ull) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float* gk = vt.gv ? vt.vertex(bn, fn, k) : grad_faces + ((size_t)bn * F + fn) * 9 + 3 * k;
#pragma unroll
            for (int c = 0; c < 3; c++) atomicAdd(&gk[c], v[3 * k + c]);
        }
    }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < wg.slots * 9; t += blockDim.x) {  // the workgroup's sums: one atomic per face and value
        const int slot = t / 9, q = t % 9, key = wg.key[slot];
        if (key < 0) continue;
        const int bn = key / F, fn = key % F;
        float* gk = vt.gv ? vt.vertex(bn, fn, q / 3) : grad_faces + ((size_t)bn * F + fn) * 9 + 3 * (q / 3);
        atomicAdd(&gk[q % 3], wg.v[slot][q]);
    }
    __syncthreads();                        // (the table may be initialised again by a caller's next pass)
}

template <class FS>
__global__ void __launch_bounds__(256) k_backward_depth_map(FS fs, const float* __restrict__ depth_map,
                                                           const int32_t* __restrict__ face_index_map,
                                                           const float* __restrict__ face_inv_map,
                                                           const float* __restrict__ weight_map,
                                                           const float* __restrict__ grad_depth_map,
                                                           float* __restrict__ grad_faces, int B, int S,
                                                           const int* __restrict__ only_large, VertexTarget vt,
                                                           GradScale gs = GradScale{nullptr, nullptr, 0.0f, 0, nullptr},
                                                           const int* __restrict__ n_large = nullptr, int flip_rows = 0) {
    if (n_large && *n_large == 0) return;          // no face was left to this kernel (uniform exit)
    backward_depth_map_pixels(fs, depth_map, face_index_map, face_inv_map, weight_map, grad_depth_map, grad_faces, B, S,
                              only_large, vt, gs, flip_rows);
}

}  // namespace d3m
