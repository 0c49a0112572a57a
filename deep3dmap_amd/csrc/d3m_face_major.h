// d3m_face_major.h -- atomic-free forms of the texture and depth backward passes.
//
// The reference scatters from pixels with float atomics: 24 per covered pixel for the texture cube
// (KCU:531-538), 9 for the depth gradient (KCU:573-590).  A face owns only the pixels inside its
// bounding box, so the sums can instead be GATHERED: FM_LANES lanes per visible face share the walk over its
// (small) bounding box, keep the sums in registers / LDS, combine them and store once.  Faces that own no pixel
// (culled, hidden or off screen: ~95% of the 2F' faces of a closed mesh) are skipped via a visibility
// flag; faces with a large bounding box fall back to the per-pixel atomic kernels.
#pragma once
#include "d3m_backward.h"
#include "d3m_forward.h"

namespace d3m {

#ifndef D3M_FM_MAX_BBOX_AREA
#define D3M_FM_MAX_BBOX_AREA 4096
#endif
constexpr int FM_MAX_BBOX_AREA = D3M_FM_MAX_BBOX_AREA;   // larger faces are left to the per-pixel atomic kernels
// Lanes per face.  The bounding-box scan is a chain of dependent loads (owner index, then the pixel's maps): one
// lane per face serialises ~10-25 of them.  FM_LANES adjacent lanes share a face, take every FM_LANES-th pixel of
// its box and combine their partial sums with quad shuffles / LDS; the visible faces of a mesh come in long index
// runs, so the waves stay dense.
constexpr int FM_LANES = 8;
constexpr int FM_FACES_PER_BLOCK = 256 / FM_LANES;

// pixel `i` (row-major) of the box, advanced by N (the lanes that share the face) per step
template <int N>
struct BoxCursorN {
    int x, y, x0, x1, bw;
    __device__ __forceinline__ BoxCursorN(int x0_, int x1_, int y0_, int start) : x0(x0_), x1(x1_), bw(x1_ - x0_ + 1) {
        y = y0_ + start / bw;
        x = x0_ + start % bw;
    }
    __device__ __forceinline__ void advance() {
        x += N;
        while (x > x1) { x -= bw; y++; }
    }
};
typedef BoxCursorN<FM_LANES> BoxCursor;

// sum over the FM_LANES (= 8) adjacent lanes of a face, in every one of them: two quad swaps and the half-row
// mirror, all DPP (no LDS crossbar)
__device__ __forceinline__ float quad_sum(float v) {
    static_assert(FM_LANES == 8, "quad_sum is written for 8 lanes per face");
    v += dpp_f32<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);     // row_half_mirror: the other quad of the 8
    return v;
}
constexpr int FLAG_HIDDEN = 0, FLAG_VISIBLE = 1, FLAG_LARGE = 2;

// flags[b*F + f] = 1 for every face that owns at least one pixel (plain stores of the same value)
__global__ void __launch_bounds__(256) k_mark_visible(const int32_t* __restrict__ face_index_map, int* __restrict__ flags,
                                                     int B, int F, int S) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * S * S) return;
    const int fi = face_index_map[i];
    if (fi >= 0) flags[(size_t)(i / ((long)S * S)) * F + fi] = FLAG_VISIBLE;
}

// Depth backward, gathered: KCU:543-592 summed over the pixels a face owns.  The sums go to grad_faces (+=) or, with a
// vertex target, straight into the gradient of the vertices the faces were gathered from (float atomics).
// LANES adjacent lanes share a face: FM_LANES (8) for ordinary meshes, a whole wave (64) for coarse ones, whose faces of
// hundreds of pixels were ninety steps of dependent loads for each of eight lanes (722 triangles @512^2: 234 us).
template <class FS, int LANES>
__device__ __forceinline__ void backward_depth_face(FS fs, const float* __restrict__ depth_map,
                                                    const int32_t* __restrict__ face_index_map,
                                                    const float* __restrict__ weight_map,
                                                    const float* __restrict__ grad_depth_map, float* __restrict__ grad_faces,
                                                    int* __restrict__ flags, int S, long gi, int sub, int F,
                                                    const VertexTarget& vt, int* __restrict__ n_large, int flip_rows,
                                                    int max_area) {
    const int bn = (int)(gi / F), fn = (int)(gi % F);
    float face[9], finv[9];
    fs.load(bn, fn, face);
    int x0, x1, y0, y1;
    if (!pixel_bbox(face, S, x0, x1, y0, y1)) return;
    const int area = (x1 - x0 + 1) * (y1 - y0 + 1);
    if (area > max_area) {      // (FM_MAX_BBOX_AREA; INT_MAX in the deterministic mode: a face's lanes own its sums whatever its size)                      // left to the per-pixel pass (counted, so that it can leave at once)
        flags[gi] = FLAG_LARGE;
        if (n_large && sub == 0) atomicAdd(n_large, 1);
        return;
    }
    face_inverse(face, S, finv);
    float tmp[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int l = 0; l < 3; l++) tmp[k] += -finv[3 * l + k] / face[3 * l + 2];     // KCU:582
    }
    float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const size_t base = (size_t)bn * S * S;
    BoxCursorN<LANES> c(x0, x1, y0, sub);
    for (int i = sub; i < area; i += LANES, c.advance()) {
        const size_t p = base + (size_t)c.y * S + c.x;
        // the pixel's maps are requested together with its owner: one round trip per step of the scan, not two
        const bool own = face_index_map[p] == fn;
        // (flip_rows: the gradient is that of the OUTPUT image, whose row S-1-y is the map's row y -- rasterize.py:311-317)
        const float ld = depth_map[p], lg = grad_depth_map[flip_rows ? base + (size_t)(S - 1 - c.y) * S + c.x : p];
        const float lw[3] = {weight_map[3 * p], weight_map[3 * p + 1], weight_map[3 * p + 2]};
        if (!__builtin_amdgcn_ballot_w64(own)) continue;
        const float depth = own ? ld : 1.0f, g = own ? lg : 0.0f;     // selected, not multiplied away
        const float depth2 = depth * depth;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float wk = own ? lw[k] : 0.0f, z_k = face[3 * k + 2];
            acc[3 * k + 0] += -g * tmp[0] * wk * depth2 * (float)S / 2.0f;          // KCU:588
            acc[3 * k + 1] += -g * tmp[1] * wk * depth2 * (float)S / 2.0f;
            acc[3 * k + 2] += g * wk * depth2 / (z_k * z_k);                        // KCU:575
        }
    }
#pragma unroll
    for (int k = 0; k < 9; k++) acc[k] = LANES == 64 ? wave_sum(acc[k]) : quad_sum(acc[k]);
    if (sub == 0) {
        if (vt.gv) {
#pragma unroll
            for (int v = 0; v < 3; v++) {
                float* g = vt.vertex(bn, fn, v);
#pragma unroll
                for (int k = 0; k < 3; k++) atomicAdd(&g[k], acc[3 * v + k]);
            }
        } else {
            float* gf = grad_faces + (size_t)gi * 9;
#pragma unroll
            for (int k = 0; k < 9; k++) gf[k] += acc[k];
        }
    }
}

// Over every face of the batch (flags decide), or -- `list`, the compacted list of d3m_visibility -- over the faces that own
// a pixel: a fixed grid striding either way (the FM_LANES lanes of a face stay together).
template <class FS, int LANES = FM_LANES>
__global__ void __launch_bounds__(256) k_backward_depth_faces(FS fs, const float* __restrict__ depth_map,
                                                             const int32_t* __restrict__ face_index_map,
                                                             const float* __restrict__ weight_map,
                                                             const float* __restrict__ grad_depth_map,
                                                             float* __restrict__ grad_faces, int* __restrict__ flags, int B,
                                                             int S, const int* __restrict__ list,
                                                             const int* __restrict__ n_list, VertexTarget vt,
                                                             int* __restrict__ n_large, int flip_rows, int max_area) {
    static_assert(LANES == FM_LANES || LANES == 64, "eight lanes per face, or a wave");
    const int sub = threadIdx.x % LANES;
    const int F = fs.num_faces();
    const long n_units = list ? (long)*n_list : (long)B * F;
    for (long u = (long)blockIdx.x * (256 / LANES) + threadIdx.x / LANES; u < n_units; u += (long)gridDim.x * (256 / LANES)) {
        const long gi = list ? (long)list[u] : u;
        if (!list && flags[gi] == FLAG_HIDDEN) continue;
        backward_depth_face<FS, LANES>(fs, depth_map, face_index_map, weight_map, grad_depth_map, grad_faces, flags, S, gi, sub, F, vt,
                            n_large, flip_rows, max_area);
    }
}

// Texture backward for ts == 2, gathered: 8 texels x 3 channels per face kept in LDS.  grad_textures += .
template <class FS>
__global__ void __launch_bounds__(256) k_backward_textures_faces(FS fs, const int32_t* __restrict__ face_index_map,
                                                                const float* __restrict__ sampling_weight_map,
                                                                const int32_t* __restrict__ sampling_index_map,
                                                                const float* __restrict__ grad_rgb_map,
                                                                float* __restrict__ grad_textures, int* __restrict__ flags,
                                                                int B, int S) {
    __shared__ float s_acc[24][256];
    const long gi = (long)blockIdx.x * FM_FACES_PER_BLOCK + threadIdx.x / FM_LANES;
    const int sub = threadIdx.x % FM_LANES;
    const int F = fs.num_faces();
    if (gi >= (long)B * F || flags[gi] == FLAG_HIDDEN) return;
    const int bn = (int)(gi / F), fn = (int)(gi % F);
    float face[9];
    fs.load(bn, fn, face);
    int x0, x1, y0, y1;
    if (!pixel_bbox(face, S, x0, x1, y0, y1)) return;
    const int area = (x1 - x0 + 1) * (y1 - y0 + 1);
    if (area > FM_MAX_BBOX_AREA) { flags[gi] = FLAG_LARGE; return; }
    const int l = threadIdx.x;
#pragma unroll
    for (int t = 0; t < 24; t++) s_acc[t][l] = 0;
    const size_t base = (size_t)bn * S * S;
    BoxCursor c(x0, x1, y0, sub);
    for (int i = sub; i < area; i += FM_LANES, c.advance()) {
        const size_t p = base + (size_t)c.y * S + c.x;
        if (face_index_map[p] != fn) continue;
        const float g0 = grad_rgb_map[3 * p + 0], g1 = grad_rgb_map[3 * p + 1], g2 = grad_rgb_map[3 * p + 2];
#pragma unroll
        for (int pn = 0; pn < 8; pn++) {
            const float w = sampling_weight_map[p * 8 + pn];
            const int isc = sampling_index_map[p * 8 + pn] & 7;                     // ts == 2: 0..7
            s_acc[isc * 3 + 0][l] += w * g0;                                        // KCU:537
            s_acc[isc * 3 + 1][l] += w * g1;
            s_acc[isc * 3 + 2][l] += w * g2;
        }
    }
    // the face's lanes sit in one wave: their LDS columns are complete once the loop has reconverged
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (sub == 0) {
        float* gt = grad_textures + (size_t)gi * 24;
#pragma unroll
        for (int t = 0; t < 24; t++) {
            float v = s_acc[t][l];
#pragma unroll
            for (int j = 1; j < FM_LANES; j++) v += s_acc[t][l + j];
            gt[t] += v;
        }
    }
}

}  // namespace d3m
