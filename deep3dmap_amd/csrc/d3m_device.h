// d3m_device.h -- device-side building blocks shared by the gfx950 rasterizer kernels.
//
// The per-(pixel, face) arithmetic below is the contract with the reference
// (pnpmodules/neural_renderer/neural_renderer/cuda/rasterize_cuda_kernel.cu, "KCU"): the same f32
// operations in the same order, with the same points of promotion to double, so that coverage
// decisions, barycentrics and depth come out bit-identical to a brute-force evaluation.  This file is
// compiled with -ffp-contract=off for that reason (no FMA contraction: every operation rounds once).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace d3m {

// Screen tile of the list form of coverage (d3m_forward.h): TILE_W x TILE_H pixels, one wave64 per tile (or four, on small
// rasters), which resolves it 64 pixels at a time.  8 x 8.  16 x 8 (-DD3M_TILE_W=16) is bit-identical and was measured in
// round 4 on the expectation that it would halve the staging's idle lanes (a tile's wave stages its listed faces 64 at a
// time, one lane each, and an 8 x 8 tile of the headline mesh lists ~35) and stage a 4 x 4-pixel face 1.6 times instead of
// 1.9: the 32-view step took 1.667 ms against 1.640 -- a 16 x 8 tile lists ~65 faces, i.e. one more than a chunk holds about
// as often as not, and the second chunk's staging costs what the first one's does.
#ifndef D3M_TILE_W
#define D3M_TILE_W 8
#endif
constexpr int TILE_W = D3M_TILE_W, TILE_H = 8, TILE_PX = TILE_W * TILE_H;
static_assert(TILE_W == 8 || TILE_W == 16, "the tile pass packs in-tile coordinates in 4 + 3 bits");
constexpr int WAVE = 64;

// ---- small helpers ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// v_cvt_i32_f32 / v_cvt_i32_f64 saturate and map NaN to 0, which is also what the reference's
// CUDA float->int conversions do (KCU:312-321, 427-429).
__device__ __forceinline__ int f2i(float v) { return (int)v; }
__device__ __forceinline__ int d2i(double v) { return (int)v; }

// order-preserving map f32 -> u32 (so that a u64 (key<<32 | face) min is a (depth, index) lexicographic min)
__device__ __forceinline__ uint32_t ordered_bits(float v) {
    uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// ---- the reference's per-face / per-pixel arithmetic ----------------------------------------------
// back-face predicate, KCU:40 / :111 / :270
__device__ __forceinline__ bool backside(const float* f) {
    return (f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0]);
}

// Where the reference promotes an f32 expression to double through a literal, the f32 form below is used only
// when it is PROVABLY bit-identical:
//   * x 0.5 is exact in either precision;
//   * a quotient of two values that are exactly representable in f32, computed in f64 and rounded to f32,
//     equals the correctly rounded f32 quotient (double rounding is innocuous for / when 53 >= 2*24 + 2);
//   * fmin/fmax against the constants 0 and 1 select the same value in either precision.
// The bit-exactness tests (tests/test_gpu_ops.py, array_equal against the brute-force oracle) guard this.

// NDC -> pixel coordinate, KCU:47 / :282: 0.5 * (v * is + is - 1)
__device__ __forceinline__ float to_pixel(float v, int is) { return 0.5f * (v * (float)is + (float)is - 1.0f); }

// pixel centre in NDC, KCU:96-97: (2.*i + 1 - is) / is -- an integer numerator below 2^24 over an integer
__device__ __forceinline__ float pixel_center(int i, int is) { return (float)(2 * i + 1 - is) / (float)is; }

// inverse of [[x0,x1,x2],[y0,y1,y2],[1,1,1]] in pixel space, KCU:44-62
__device__ __forceinline__ void face_inverse(const float* face, int is, float* out) {
    float p[3][2];
#pragma unroll
    for (int n = 0; n < 3; n++) {
        p[n][0] = to_pixel(face[3 * n + 0], is);
        p[n][1] = to_pixel(face[3 * n + 1], is);
    }
    const float den = (p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1]) + p[1][0] * (p[2][1] - p[0][1]));
    out[0] = (p[1][1] - p[2][1]) / den;
    out[1] = (p[2][0] - p[1][0]) / den;
    out[2] = (p[1][0] * p[2][1] - p[2][0] * p[1][1]) / den;
    out[3] = (p[2][1] - p[0][1]) / den;
    out[4] = (p[0][0] - p[2][0]) / den;
    out[5] = (p[2][0] * p[0][1] - p[0][0] * p[2][1]) / den;
    out[6] = (p[0][1] - p[1][1]) / den;
    out[7] = (p[1][0] - p[0][0]) / den;
    out[8] = (p[0][0] * p[1][1] - p[1][0] * p[0][1]) / den;
}

// three half-plane tests at the pixel centre, KCU:115-117 (a '<' rejects: edges are inclusive)
__device__ __forceinline__ bool inside_face(const float* face, float xp, float yp) {
    return !(((yp - face[1]) * (face[3] - face[0]) < (xp - face[0]) * (face[4] - face[1])) ||
             ((yp - face[4]) * (face[6] - face[3]) < (xp - face[3]) * (face[7] - face[4])) ||
             ((yp - face[7]) * (face[0] - face[6]) < (xp - face[6]) * (face[1] - face[7])));
}

// barycentrics (clamped, renormalised) and perspective-correct depth, KCU:120-139.
// Returns false when the depth falls outside (near, far) -- NaN depths (zero-area faces) also fail.
__device__ __forceinline__ bool weights_depth(const float* face, const float* finv, int xi, int yi, float near,
                                              float far, float* w, float& zp) {
    const float fx = (float)xi, fy = (float)yi;
    w[0] = finv[0] * fx + finv[1] * fy + finv[2];
    w[1] = finv[3] * fx + finv[4] * fy + finv[5];
    w[2] = finv[6] * fx + finv[7] * fy + finv[8];
    float w_sum = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        w[k] = fminf(fmaxf(w[k], 0.0f), 1.0f);            // KCU:129; fmax/fmin drop a NaN operand, as in CUDA
        w_sum += w[k];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) w[k] /= w_sum;
    zp = 1.0f / (w[0] / face[2] + w[1] / face[5] + w[2] / face[8]);                      // KCU:136
    return !(zp <= near || far <= zp) && (zp == zp);
}

// ---- face sources ----------------------------------------------------------------------------
// Dense: faces [B,F,3,3] as the reference extension receives them.
struct DenseFaces {
    const float* faces;
    int F;
    __host__ __device__ __forceinline__ int num_faces() const { return F; }
    __device__ __forceinline__ void load(int b, int f, float* out) const {
        const float* p = faces + ((size_t)b * F + f) * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) out[k] = p[k];
    }
};

// Index triple of input triangle fl of view b: from tri [tri_batch,Ft,3], or -- tri == NULL, grid_w > 0 -- from the
// implicit topology of a depth map's grid mesh (deep3dmap/core/renderer/utils.py:74-78: grid_w vertices per row, cell
// (y, x) carries (tl, bl, tr) in the first half of the list and (tr, bl, br) in the second), so that the adapter needs no
// index tensor at all.
__device__ __forceinline__ void tri_ids(const int32_t* tri, int tri_batch, int Ft, int grid_w, int b, int fl, int* ids) {
    if (tri) {
        const int32_t* t = tri + ((size_t)(tri_batch > 1 ? b : 0) * Ft + fl) * 3;
        ids[0] = t[0]; ids[1] = t[1]; ids[2] = t[2];
        return;
    }
    const int cells = Ft >> 1, second = fl >= cells, cell = second ? fl - cells : fl;
    const int y = cell / (grid_w - 1), x = cell - y * (grid_w - 1);
    const int tl = y * grid_w + x, bl = tl + grid_w;
    ids[0] = second ? tl + 1 : tl;
    ids[1] = bl;
    ids[2] = second ? bl + 1 : tl + 1;
}

// Indexed: projected vertices [B,V,3] + index triples [Bt,Ft,3]; the fill_back copy (face Ft+f =
// face f with vertex order reversed, renderer.py:86) is generated on the fly.
struct IndexedFaces {
    const float* verts;
    const int32_t* tri;
    int V, Ft, tri_batch, fill_back;
    int vert_batch;   // 1: one vertex array shared by every view (stride 0), otherwise one per view
    int grid_w = 0;   // tri == NULL: implicit grid topology with this many vertices per row (tri_ids)
    __host__ __device__ __forceinline__ int num_faces() const { return fill_back ? 2 * Ft : Ft; }
    __device__ __forceinline__ void vertex_ids(int b, int f, int* ids) const {
        const bool back = f >= Ft;
        int t[3];
        tri_ids(tri, tri_batch, Ft, grid_w, b, back ? f - Ft : f, t);
        ids[0] = back ? t[2] : t[0];
        ids[1] = t[1];
        ids[2] = back ? t[0] : t[2];
    }
    __device__ __forceinline__ void load(int b, int f, float* out) const {
        int ids[3];
        vertex_ids(b, f, ids);
#pragma unroll
        for (int n = 0; n < 3; n++) {
            const float* p = verts + ((size_t)(vert_batch > 1 ? b : 0) * V + ids[n]) * 3;
            out[3 * n + 0] = p[0];
            out[3 * n + 1] = p[1];
            out[3 * n + 2] = p[2];
        }
    }
};

// Where face gradients go.  gv == NULL: the dense [B,F',3,3] array of the reference operators.  Otherwise they are
// accumulated (float atomics) straight into the gradient of the vertices the faces were gathered from
// (vertices_to_faces + fill_back, renderer.py:86): no 36 B/face array to zero, fill and scatter afterwards.
struct VertexTarget {
    float* gv;             // [B, V, 3]
    const int32_t* tri;    // [tri_batch, Ft, 3]
    int V, Ft, tri_batch;
    int grid_w = 0;        // tri == NULL: implicit grid topology (tri_ids)
    __device__ __forceinline__ float* vertex(int b, int f, int n) const {      // vertex n of (virtual) face f
        const bool back = f >= Ft;
        int t[3];
        tri_ids(tri, tri_batch, Ft, grid_w, b, back ? f - Ft : f, t);
        const int m = back ? 2 - n : n;
        return gv + ((size_t)b * V + (m == 0 ? t[0] : (m == 1 ? t[1] : t[2]))) * 3;
    }
};

// ---- gradient maps that still lack their scalar factors ---------------------------------------------------
// The fused fit objective (k_render_lit_epilogue) leaves the objective's gradient in the maps WITHOUT the factors that
// are only known later -- 1/sum(mask) from the reduction, the incoming gradient of the loss -- so that no pass over the
// pixels is needed in backward: the operators that read the maps multiply by these three scalars.
struct GradScale {
    const float* totals;      // the objective's sums (totals[2] = sum of the mask); NULL: the maps are final
    const float* grad_out;    // gradient of the scalar objective; NULL = 1
    float pixels;             // pixels per view
    int records;              // the rgb / alpha gradients come from the edge gradient's per-pixel records
                              // (k_render_lit_fit_records), which already carry 1 / (3 sum(mask)) and 1 / pixels
    const float* den_given;   // the photometric normaliser when it was known before the render (fit->mask_sum): the
                              // backward pass then does not depend on the objective's reduction having finished
    __device__ __forceinline__ void get(float& s_rgb, float& s_alpha, float& s_depth) const {
        s_rgb = s_alpha = s_depth = 1.0f;
        if (totals) {
            const float go = grad_out ? *grad_out : 1.0f, den = den_given ? *den_given : totals[2];
            s_rgb = records ? go : go / (3.0f * den);
            s_depth = go / den;
            s_alpha = records ? go : go / pixels;
        }
    }
};
// the rgb gradient of a pixel: a [B,S,S,3] map, or the yzw of the [B,S,S] float4 records
struct RgbGrad {
    const float* p;
    int stride, off;
    __device__ __forceinline__ float get(size_t pixel, int k) const { return p[pixel * stride + off + k]; }
};

// ---- XCD-contiguous work order ---------------------------------------------------------------------
// Workgroups i and i + 8 share an XCD (observed dispatch order; a speed matter only).  A kernel that strides a fixed
// grid (a multiple of 8) over n work units hands XCD x the contiguous range [x*per, (x+1)*per): neighbouring units
// (neighbouring faces: shared map lines) then meet in ONE L2 instead of being fetched into all eight.
struct XcdOrder {
    int per;
    __device__ __forceinline__ explicit XcdOrder(int n) : per((n + 7) >> 3) {}
    __device__ __forceinline__ bool more(int i) const { return (i >> 3) < per; }
    __device__ __forceinline__ int unit(int i) const { return (i & 7) * per + (i >> 3); }
};

// ---- wave-level primitives -----------------------------------------------------------------------
__device__ __forceinline__ int wave_inclusive_scan(int v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}
// Sum over the 64 lanes, returned in every lane.  DPP within a 16-lane row (quad swaps, then the two mirrors: no
// LDS-crossbar round trips as __shfl_xor would take), v_readlane across the four rows.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);    // row_half_mirror
    v += dpp_f32<0x140>(v);    // row_mirror: every lane now holds its row's sum
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}

// Sums a workgroup collects per KEY before they go to global memory with one atomic each: the per-pixel fallbacks of the
// gathered backward passes add, for every (large) face, the same few values from every wave that holds pixels of it, and
// float atomics on ONE address from all over the chip take turns at the memory side (~150 ns each across the XCDs: eight
// views of an 8-triangle mesh @1024^2 spent 1.2 ms in 9 x 170 000 of them).  A table of SLOTS keys in LDS: the wave's
// leading lane adds its wave's sums under the face's key (LDS atomics), and the workgroup flushes once at its end.  The
// callers give a workgroup a CONTIGUOUS run of pixels (a few image rows), so that it meets few faces.
template <int NV, int SLOTS = 64>
struct WgSums {
    static constexpr int slots = SLOTS;
    int key[SLOTS];
    float v[SLOTS][NV];
    __device__ __forceinline__ void init() {                        // (all threads of the workgroup)
        for (int t = threadIdx.x; t < SLOTS; t += blockDim.x) key[t] = -1;
        for (int t = threadIdx.x; t < SLOTS * NV; t += blockDim.x) (&v[0][0])[t] = 0.0f;
        __syncthreads();
    }
    // one lane: false = the table is full (the caller adds to global memory itself)
    __device__ __forceinline__ bool add(int k, const float* vals) {
        static_assert((SLOTS & (SLOTS - 1)) == 0, "a power of two");
        unsigned s = ((unsigned)k * 2654435761u) >> 16;
        for (int p = 0; p < 8; p++, s++) {                          // (eight probes, then the caller's own atomics)
            s &= SLOTS - 1;
            int cur = key[s];
            if (cur == -1) cur = atomicCAS(&key[s], -1, k);
            if (cur == -1 || cur == k) {
#pragma unroll
                for (int j = 0; j < NV; j++) atomicAdd(&v[s][j], vals[j]);
                return true;
            }
        }
        return false;
    }
};

// Inclusive running maximum over the 64 lanes (unsigned; 0 is the identity): four row_shr steps inside each 16-lane
// row, then the two row broadcasts -- six DPP moves, no LDS crossbar.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32_or0(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_max_scan(uint32_t v) {
    v = max(v, dpp_u32_or0<0x111, 0xF>(v));     // row_shr:1
    v = max(v, dpp_u32_or0<0x112, 0xF>(v));     // row_shr:2
    v = max(v, dpp_u32_or0<0x114, 0xF>(v));     // row_shr:4
    v = max(v, dpp_u32_or0<0x118, 0xF>(v));     // row_shr:8
    v = max(v, dpp_u32_or0<0x142, 0xA>(v));     // row_bcast:15 -> rows 1 and 3
    v = max(v, dpp_u32_or0<0x143, 0xC>(v));     // row_bcast:31 -> rows 2 and 3
    return v;
}

// Match-any over a wave: the mask of the `has` lanes that hold the same 32-bit key as this lane (undefined in lanes
// without `has`).  One pass per DISTINCT key -- readlane, one compare, two selects -- and nothing else inside the
// loop: ranks, counts and leaders are bit counts on the returned mask.  Used to merge the per-line counter updates of
// a wave into one atomic per distinct line (neighbouring crossings fall on the same lines).
__device__ __forceinline__ unsigned long long wave_match_any(uint32_t key, bool has) {
    unsigned long long pending = __builtin_amdgcn_ballot_w64(has);
    unsigned long long mine = 0;
    while (pending) {
        const int leader = __builtin_ctzll(pending);                                  // wave-uniform
        const uint32_t lk = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
        const bool eq = key == lk;
        const unsigned long long same = __builtin_amdgcn_ballot_w64(has && eq);
        if (eq) mine = same;
        pending &= ~same;
    }
    return mine;
}
// number of set bits of `mask` below this lane
__device__ __forceinline__ int mask_rank(unsigned long long mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

}  // namespace d3m
