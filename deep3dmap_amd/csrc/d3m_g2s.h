// d3m_g2s.h -- the renderer block of the gan2shape training step (deep3dmap/models/frameworks/gan2shape.py:444,463-497,
// "G2S" below; NrRenderer = deep3dmap/core/renderer/renderer_nr.py, "CR") as eight launches (nine fused passes: the last two share a grid):
//
//   forward   k_g2s_front     per canonical pixel: the view's (R, t) (CR/utils.py:54-71), normal (CR:127-139) -> diffuse
//                             shading -> texture (G2S:463-466), and the pixel as a mesh vertex: back-projection, rigid
//                             motion (CR:90-100), the mesh renderer's camera (NR/projection.py); clears the z-buffer and
//                             the texture-gradient accumulator
//             k_g2s_raster    warp_canon_depth's rasterization (CR:116-125, KCU:24-169) for a mesh whose triangles are a
//                             few pixels each: a few lanes per triangle PAIR of the implicit grid topology (the fill_back
//                             copy is the other orientation of the same three vertices) walk its bounding box and bid
//                             (depth, face index) into a 64-bit z-buffer with atomicMax -- the per-(pixel, face)
//                             arithmetic and the "nearest, lowest index among equals" rule of d3m_forward.h, without its
//                             three binning passes
//             k_g2s_sample    per output pixel: pooled / flipped / clamped recon_depth (NR/rasterize.py:305-326, CR:122-124),
//                             border mask (G2S:478-482), inverse-warped sampling position (CR:102-114), bilinear lookup of
//                             the texture (F.grid_sample, G2S:483), clamp, masked-L1 sums (G2S:486,489); and the
//                             second-difference sums of smooth_loss(depth) + smooth_loss(diffuse_shading) (G2S:493-494)
//             k_g2s_finish    the loss values
//   backward  k_g2s_sample_backward   recon_im's gradient -> texture (atomics), recon_depth, the inverse view
//             k_g2s_depth_faces       K6 (KCU:543-592) gathered per triangle pair, stored per triangle (no atomics)
//             k_g2s_front_backward    per canonical pixel: its six incident triangles' gradients -> camera -> rigid
//                                     motion -> depth, view; texture gradient -> albedo, light, normal
//             k_g2s_depth_backward    normals' adjoint + smooth loss -> depth; its last B workgroups: per-entry sums, (R, t) -> view
//
// The reference runs this block as ~150 eager kernels forward and as many again in backward.  Every map of the block is
// a few hundred KB, so the passes are latency-bound: what counts is the number of launches, that nothing is
// materialised between them, and how many float atomics a pass issues (~37 G/s on this chip in these patterns).
#pragma once
#include "d3m_aux.h"
#include "d3m_bid.h"
#include "d3m_forward.h"

namespace d3m {

// Device view of d3m_g2s_block (include/d3m_raster.h).
struct G2S {
    int B, H, W;                  // canonical maps [B,H,W]
    int s, S, aa;                 // output images s x s; raster S x S (S = 2s with anti-aliasing)
    int Bh, flip;                 // flip: B = 2 Bh and entries (b, b + Bh) share the product of their border masks
    const float* inv_K; int invK_b;
    const float* K; int K_b;      // NrRenderer.K / inv_K (CR:35-46)
    float center_z, depth_lo, depth_hi, near, far;
    Cam cam;                      // the mesh renderer's camera (CR:47-54: projection)
    const float* view; int view_n;         // [B,view_n] or NULL: (rot, trans) are then inputs, else written by k_g2s_front
    float *rot, *trans;
    const float *depth, *albedo, *light_a, *light_b, *light_d, *target, *extra_mask;
    float *normal, *diffuse, *texture, *screen_vertices;
    unsigned long long* zbuf;     // [B,S,S] ~((ordered depth bits << 32) | face index), 0 = uncovered; row 0 = bottom
    float *recon_depth, *recon_im, *recon_mask, *losses;
    float* scratch;
    int off_sample, off_front, off_sback;                // partial sums inside scratch (floats)
    int split_s, split_f;                                // workgroups per batch entry (sample / front passes)
    float lam_smooth; int with_smooth;
    float n_xx, n_xy, n_yy;                              // element counts of the smooth loss's means
    // backward
    const float *grad_recon_im, *g_l1, *g_l1_flip, *g_smooth, *g_total;
    float *grad_texture, *grad_tri, *grad_depth_map, *grad_normal, *grad_depth_mesh;
    float *grad_depth, *grad_albedo, *grad_light_a, *grad_light_b, *grad_light_d, *grad_rot, *grad_trans, *grad_view;
};

constexpr int G2S_TOTALS = 16;          // scratch[0..16): num1, num2, den1, den2, then the 2 x 4 smooth sums
constexpr int G2S_SAMPLE_PART = 12;     // per workgroup of k_g2s_sample: the same twelve sums
constexpr int G2S_FRONT_SUMS = 17;      // grad_rot 9, grad_trans 3, light_a, light_b, light_d 3
constexpr int G2S_SBACK_SUMS = 12;      // gradient of the inverse view's (A', t')

struct ZeroRanges { uint32_t* p[2]; unsigned n[2]; };       // words

// N sums over a 256-thread workgroup with ONE barrier (DPP wave sums, then four partials per value through LDS); every
// thread gets the totals.  s_buf: 4 * N floats, not reused by the caller before its next barrier.
template <int N>
__device__ __forceinline__ void block_sums_256(float* v, float* s_buf) {
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N; k++) {
        v[k] = wave_sum(v[k]);
        if (lane_id() == 0) s_buf[wv * N + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = (s_buf[k] + s_buf[N + k]) + (s_buf[2 * N + k] + s_buf[3 * N + k]);
}

// ---- shared pieces -------------------------------------------------------------------------------------------------
// unit normal of the back-projected depth map at (x, y) (CR:127-139)
__device__ __forceinline__ void g2s_normal(const float* __restrict__ dview, const float* iK, int H, int W, int x, int y,
                                           float* n) {
    n[0] = 0.0f; n[1] = 0.0f; n[2] = 1.0f;
    if (x > 0 && x < W - 1 && y > 0 && y < H - 1) {
        float pr[3], pl[3], pd[3], pu[3], tu[3], tv[3];
        dn_point(dview, iK, W, x + 1, y, pr); dn_point(dview, iK, W, x - 1, y, pl);
        dn_point(dview, iK, W, x, y + 1, pd); dn_point(dview, iK, W, x, y - 1, pu);
#pragma unroll
        for (int k = 0; k < 3; k++) { tu[k] = pr[k] - pl[k]; tv[k] = pd[k] - pu[k]; }
        cross3f(tu, tv, n);
    }
    const float len = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) + DN_EPS;
    n[0] /= len; n[1] /= len; n[2] /= len;
}

// the inverse of the view's rigid motion, translate_pts(-t) then rotate_pts(R^T) (CR:102-107), as one (A', t'):
// A' = R^T, t' = -(t R) (row vector), the composition NrRenderer's Rigid.inverse() makes on [B,3,3]
struct G2SView { float A[9], t[3]; };
__device__ __forceinline__ void g2s_inverse_view(const float* R, const float* t, G2SView& v) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) v.A[3 * i + j] = R[3 * j + i];
        v.t[i] = -((t[0] * R[i] + t[1] * R[3 + i]) + t[2] * R[6 + i]);
    }
}

// recon_depth of output pixel (yo, xo): the raster's depth flipped and 2x2-pooled as rasterize_rgbad does
// (NR/rasterize.py:305-326, same summation order as k_output_epilogue), then clamped (CR:122-124).  `pooled` = before.
__device__ __forceinline__ float g2s_recon_depth(const G2S& g, int b, int yo, int xo, float& pooled) {
    const int S = g.S, n = g.aa ? 2 : 1;
    float acc = 0.0f;
    for (int dy = 0; dy < n; dy++)
        for (int dx = 0; dx < n; dx++)
            acc += bid_depth(g.zbuf[((size_t)b * S + (S - 1 - (yo * n + dy))) * S + xo * n + dx], g.far);
    pooled = acc * (g.aa ? 0.25f : 1.0f);
    return fminf(fmaxf(pooled, g.depth_lo), g.depth_hi);
}

// F.grid_sample's pixel coordinate (align_corners = False) and the four bilinear taps around it
struct Bilinear {
    int x0, y0;
    float fx, fy;                 // fractional position inside the cell
    __device__ __forceinline__ void at(float gx, float gy, int W, int H) {
        const float ix = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f, iy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
        const float flx = floorf(ix), fly = floorf(iy);
        // positions far outside the image (or NaN) contribute nothing: every tap is then out of bounds
        x0 = (flx >= -2.0f && flx <= (float)W) ? (int)flx : -2;
        y0 = (fly >= -2.0f && fly <= (float)H) ? (int)fly : -2;
        fx = ix - flx;
        fy = iy - fly;
    }
    __device__ __forceinline__ void weights(float* w) const {      // nw, ne, sw, se
        w[0] = (1.0f - fx) * (1.0f - fy); w[1] = fx * (1.0f - fy); w[2] = (1.0f - fx) * fy; w[3] = fx * fy;
    }
    __device__ __forceinline__ void offsets(int W, int H, int* o) const {   // plane offset of each tap, -1 = zero padding
        const bool xa = x0 >= 0 && x0 < W, xb = x0 + 1 >= 0 && x0 + 1 < W, ya = y0 >= 0 && y0 < H, yb = y0 + 1 >= 0 && y0 + 1 < H;
        o[0] = (xa && ya) ? y0 * W + x0 : -1;
        o[1] = (xb && ya) ? y0 * W + x0 + 1 : -1;
        o[2] = (xa && yb) ? (y0 + 1) * W + x0 : -1;
        o[3] = (xb && yb) ? (y0 + 1) * W + x0 + 1 : -1;
    }
};

// the sampling position of output pixel (yo, xo) with target-view depth d: back-project, move by `view`, project
// (get_inv_warped_2d_grid, CR:102-114); p = back-projected point shifted to the rotation centre, q = moved point
__device__ __forceinline__ void g2s_grid(const G2S& g, const float* iK, const float* K, const G2SView& view, int xo, int yo,
                                         float d, float* ray, float* p, float* q, float* uv) {
    gw_ray(iK, (float)xo, (float)yo, ray);
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = ray[k] * d;
    gw_rigid(p, view.A, view.t, g.center_z, q);
    gw_project(q, K, g.s, g.s, uv);
}

// view vector -> (R, t): the arithmetic of k_view_transform (get_transform_matrices, CR/utils.py:54-71)
__device__ __forceinline__ void g2s_view_rt(const float* v, int n_comp, float* R, float* t) {
    float mx[9], my[9], mz[9], yx[9];
    euler_factors(cosf(v[0]), sinf(v[0]), cosf(v[1]), sinf(v[1]), cosf(v[2]), sinf(v[2]), mx, my, mz);
    mat3_mul(my, mx, yx);
    mat3_mul(mz, yx, R);
#pragma unroll
    for (int k = 0; k < 3; k++) t[k] = 3 + k < n_comp ? v[3 + k] : 0.0f;
}

// The front-facing copy of triangle pair fl of view b, read through the implicit topology (tri_ids): face = its three
// screen-space vertices in the order of that copy, fid = its index in the fill_back'd face list (fl, or Ft + fl for the
// reversed copy, renderer.py:86).  False: both orientations are culled (KCU:40).
__device__ __forceinline__ bool g2s_front_face(const G2S& g, int b, int fl, int Ft, float* face, int& fid, bool& reversed) {
    int ids[3];
    tri_ids(nullptr, 1, Ft, g.W, b, fl, ids);
    const float* sv = g.screen_vertices + (size_t)b * g.H * g.W * 3;
    float v[9];
#pragma unroll
    for (int n = 0; n < 3; n++) {
        v[3 * n] = sv[3 * ids[n]]; v[3 * n + 1] = sv[3 * ids[n] + 1]; v[3 * n + 2] = sv[3 * ids[n] + 2];
    }
    reversed = backside(v);
#pragma unroll
    for (int k = 0; k < 9; k++) face[k] = reversed ? v[(2 - k / 3) * 3 + k % 3] : v[k];
    fid = reversed ? Ft + fl : fl;
    return !(reversed && backside(face));
}

// ---- forward ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_g2s_front(G2S g, ZeroRanges z) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
#pragma unroll
    for (int r = 0; r < 2; r++)
        for (long k = i; k < (long)z.n[r]; k += stride) z.p[r][k] = 0u;
    const int HW = g.H * g.W;
    if (i >= (long)g.B * HW) return;
    const int b = (int)(i / HW), pix = (int)(i - (long)b * HW);
    const int y = pix / g.W, x = pix - y * g.W;
    const float* iK = cam_ptr(g.inv_K, g.invK_b, b, 9);
    const float* dview = g.depth + (size_t)b * HW;
    // the view's rigid motion: given, or from the view vector here (every lane of the entry computes the same bits;
    // one of them leaves it for the later passes)
    float R[9], t[3];
    if (g.view) {
        g2s_view_rt(g.view + (size_t)b * g.view_n, g.view_n, R, t);
        if (pix == 0) {
#pragma unroll
            for (int k = 0; k < 9; k++) g.rot[(size_t)b * 9 + k] = R[k];
#pragma unroll
            for (int k = 0; k < 3; k++) g.trans[(size_t)b * 3 + k] = t[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 9; k++) R[k] = g.rot[(size_t)b * 9 + k];
#pragma unroll
        for (int k = 0; k < 3; k++) t[k] = g.trans[(size_t)b * 3 + k];
    }
    // shading (G2S:463-466)
    float n[3];
    g2s_normal(dview, iK, g.H, g.W, x, y, n);
    if (g.normal) { g.normal[3 * i] = n[0]; g.normal[3 * i + 1] = n[1]; g.normal[3 * i + 2] = n[2]; }
    const float* ld = g.light_d + 3 * b;
    const float diff = fmaxf((n[0] * ld[0] + n[1] * ld[1]) + n[2] * ld[2], 0.0f);
    g.diffuse[i] = diff;
    const float shading = g.light_a[b] + g.light_b[b] * diff;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const size_t o = ((size_t)b * 3 + c) * HW + pix;
        g.texture[o] = (g.albedo[o] / 2.0f + 0.5f) * shading * 2.0f - 1.0f;
    }
    // the pixel as a vertex of the grid mesh: get_warped_3d_grid (CR:90-100), then the mesh renderer's camera
    float ray[3], p[3], q[3], o[3];
    gw_ray(iK, (float)x, (float)y, ray);
    const float d = dview[pix];
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = ray[k] * d;
    gw_rigid(p, R, t, g.center_z, q);
    camera_point(g.cam, b, q, o, nullptr);
    g.screen_vertices[3 * i] = o[0]; g.screen_vertices[3 * i + 1] = o[1]; g.screen_vertices[3 * i + 2] = o[2];
}

// ---- the mesh's coverage by bidding (d3m_bid.h): G2S_PW triangle pairs of the implicit grid per wave ----------------------
// Pairs per wave.  A wave's walk is a chain of dependent steps (LDS marks -> owner -> z-buffer entry -> bid), ~0.5 us
// each, and the pass ends with its slowest wave: a depth map's mesh seen from 57 degrees has bounding boxes of 16 px in
// the median, 200 at the 99th percentile, > 1000 at the top.  64 pairs per wave left 2 waves per SIMD and tail waves of
// 100+ steps (0.09 ms for the pass); 16 pairs -- the other lanes only help with the walk -- give 8 waves per SIMD and
// short chains (0.05 ms; 4, 8 and 32 pairs: 0.055-0.059).
#ifndef D3M_G2S_PAIRS_PER_WAVE
#define D3M_G2S_PAIRS_PER_WAVE 16
#endif
constexpr int G2S_PW = D3M_G2S_PAIRS_PER_WAVE;
typedef BidStage<G2S_PW> G2SStage;

// lane j < G2S_PW <- pair (wave's first pair + j): returns the rows of its box (0: culled / off screen / out of range)
__device__ __forceinline__ int g2s_stage_pair(const G2S& g, G2SStage& st, long pair, int Ft, bool& reversed) {
    const int lane = lane_id(), S = g.S;
    int cnt = 0;
    reversed = false;
    if (lane < G2S_PW && pair < (long)g.B * Ft) {
        const int b = (int)(pair / Ft), fl = (int)(pair - (long)b * Ft);
        float face[9], finv[9];
        int fid, x0, x1, y0, y1;
        if (g2s_front_face(g, b, fl, Ft, face, fid, reversed) && pixel_bbox(face, S, x0, x1, y0, y1)) {
            face_inverse(face, S, finv);
#pragma unroll
            for (int k = 0; k < 9; k++) { st.face[k][lane] = face[k]; st.finv[k][lane] = finv[k]; }
            const int bw = x1 - x0 + 1;
            st.fid[lane] = fid; st.x0[lane] = x0; st.y0[lane] = y0; st.bw[lane] = bw;
            cnt = y1 - y0 + 1;                  // the box's rows (bid_rows)
        }
    }
    return cnt;
}

// warp_canon_depth's coverage: candidates that pass the reference's tests (KCU:110-139 through d3m_device.h: same
// operations, same bits) bid for their pixel.  One wave per G2S_PW triangle pairs, four independent waves per workgroup.
__global__ void __launch_bounds__(256) k_g2s_raster(G2S g) {
    __shared__ G2SStage s_stage[4];
    G2SStage& st = s_stage[threadIdx.x >> 6];
    const int Ft = 2 * (g.H - 1) * (g.W - 1), S = g.S;
    const long pair = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * G2S_PW + lane_id();
    bool reversed;
    const int cnt = g2s_stage_pair(g, st, pair, Ft, reversed);
    auto slot_of = [&](int lo, int xi, int yi) {
        const long owner_pair = pair - lane_id() + lo;                 // (the pairs of a wave may straddle two views)
        return g.zbuf + ((size_t)(owner_pair / Ft) * S + yi) * S + xi;
    };
    bid_rows(st, cnt, S,
        [&](int lo, int xi, int yi) {              // cheap: the three half-plane tests (no early z: see k_bid_faces)
            float face[9];
#pragma unroll
            for (int k = 0; k < 9; k++) face[k] = (k % 3 == 2) ? 0.0f : st.face[k][lo];
            return inside_face(face, pixel_center(xi, S), pixel_center(yi, S));
        },
        [&](int lo, int xi, int yi) {              // costly: barycentrics and depth (seven divisions), the bid
            float face[9], finv[9], w[3], zp;
#pragma unroll
            for (int k = 0; k < 9; k++) { face[k] = st.face[k][lo]; finv[k] = st.finv[k][lo]; }
            if (!weights_depth(face, finv, xi, yi, g.near, g.far, w, zp)) return;
            const unsigned long long e = bid_key(zp, st.fid[lo]);
            unsigned long long* slot = slot_of(lo, xi, yi);
            if (e > *slot) atomicMax(slot, e);
        });
}

// grid (split_s, Bh): with flip a lane handles its pixel in both halves of the batch, which share the mask product.
// The workgroups also take the smooth-loss sums of their entries' depth and shading maps (complete since k_g2s_front).
__global__ void __launch_bounds__(256) k_g2s_sample(G2S g) {
    __shared__ float s_buf[4 * G2S_SAMPLE_PART];
    const int s = g.s, npx = s * s, HW = g.H * g.W, halves = g.flip ? 2 : 1;
    float sums[G2S_SAMPLE_PART];       // num1, num2, den1, den2, 4 x smooth(depth), 4 x smooth(shading)
#pragma unroll
    for (int k = 0; k < G2S_SAMPLE_PART; k++) sums[k] = 0.0f;
    for (int pix = blockIdx.x * 256 + threadIdx.x; pix < npx; pix += gridDim.x * 256) {
        const int yo = pix / s, xo = pix - yo * s;
        float rd[2] = {0, 0}, m = 1.0f, pooled;
        for (int h = 0; h < halves; h++) {
            const int b = blockIdx.y + h * g.Bh;
            rd[h] = g2s_recon_depth(g, b, yo, xo, pooled);
            if (g.recon_depth) g.recon_depth[(size_t)b * npx + pix] = rd[h];
            m *= rd[h] < g.depth_hi ? 1.0f : 0.0f;                 // invalid border pixels were clamped to the limit (G2S:478)
        }
        for (int h = 0; h < halves; h++) {
            const int b = blockIdx.y + h * g.Bh;
            const float mh = g.extra_mask ? m * g.extra_mask[(size_t)b * npx + pix] : m;
            if (g.recon_mask) g.recon_mask[(size_t)b * npx + pix] = mh;
            G2SView view;
            g2s_inverse_view(g.rot + (size_t)b * 9, g.trans + (size_t)b * 3, view);
            float ray[3], p[3], q[3], uv[2], w[4];
            g2s_grid(g, cam_ptr(g.inv_K, g.invK_b, b, 9), cam_ptr(g.K, g.K_b, b, 9), view, xo, yo, rd[h], ray, p, q, uv);
            Bilinear bl;
            int o[4];
            bl.at(uv[0], uv[1], g.W, g.H);
            bl.weights(w);
            bl.offsets(g.W, g.H, o);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float* plane = g.texture + ((size_t)b * 3 + c) * HW;
                float raw = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; k++) if (o[k] >= 0) raw += plane[o[k]] * w[k];
                const float v = fminf(fmaxf(raw, -1.0f), 1.0f);
                g.recon_im[((size_t)b * 3 + c) * npx + pix] = v;
                if (g.target) sums[h] += fabsf(v - g.target[((size_t)blockIdx.y * 3 + c) * npx + pix]) * mh;
            }
            sums[2 + h] += mh;
        }
    }
    if (g.with_smooth) {
        for (int h = 0; h < halves; h++) {
            const size_t base = (size_t)(blockIdx.y + h * g.Bh) * HW;
            for (int pix = blockIdx.x * 256 + threadIdx.x; pix < HW; pix += gridDim.x * 256) {
                const int y = pix / g.W, x = pix - y * g.W;
#pragma unroll
                for (int mp = 0; mp < 2; mp++) {
                    const float* r0 = (mp ? g.diffuse : g.depth) + base + (size_t)y * g.W;
                    float* a = sums + 4 + 4 * mp;
                    if (x + 2 < g.W) a[0] += fabsf(sm_dxx(r0, x));
                    if (y + 1 < g.H && x + 1 < g.W) {
                        a[1] += fabsf(sm_dxy(r0, r0 + g.W, x));
                        a[2] += fabsf(sm_dyx(r0, r0 + g.W, x));
                    }
                    if (y + 2 < g.H) a[3] += fabsf(sm_dyy(r0, r0 + g.W, r0 + 2 * g.W, x));
                }
            }
        }
    }
    block_sums_256<G2S_SAMPLE_PART>(sums, s_buf);
    if (threadIdx.x < G2S_SAMPLE_PART)
        g.scratch[g.off_sample + G2S_SAMPLE_PART * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + threadIdx.x] = sums[threadIdx.x];
}

// one workgroup: losses = (l1, l1_flip, smooth, l1 + l1_flip + lam_smooth * smooth); totals kept for backward
__global__ void __launch_bounds__(256) k_g2s_finish(G2S g) {
    __shared__ float s_buf[4 * G2S_SAMPLE_PART];
    float acc[G2S_SAMPLE_PART];
#pragma unroll
    for (int k = 0; k < G2S_SAMPLE_PART; k++) acc[k] = 0.0f;
    const int n_s = g.Bh * g.split_s;
    for (int i = threadIdx.x; i < n_s; i += 256) {
#pragma unroll
        for (int k = 0; k < G2S_SAMPLE_PART; k++) acc[k] += g.scratch[g.off_sample + G2S_SAMPLE_PART * i + k];
    }
    block_sums_256<G2S_SAMPLE_PART>(acc, s_buf);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < G2S_SAMPLE_PART; k++) g.scratch[k] = acc[k];
        const float l1 = g.target ? acc[0] / (3.0f * acc[2]) : 0.0f;            // photometric_loss: the mask expands to 3 channels
        const float l1f = (g.target && g.flip) ? acc[1] / (3.0f * acc[3]) : 0.0f;
        float sm = 0.0f;
        if (g.with_smooth)
            sm = (((acc[4] / g.n_xx + acc[5] / g.n_xy) + acc[6] / g.n_xy) + acc[7] / g.n_yy) +
                 (((acc[8] / g.n_xx + acc[9] / g.n_xy) + acc[10] / g.n_xy) + acc[11] / g.n_yy);
        g.losses[0] = l1; g.losses[1] = l1f; g.losses[2] = sm;
        g.losses[3] = (l1 + l1f) + g.lam_smooth * sm;
    }
}

// ---- backward ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float g2s_scalar(const float* p) { return p ? *p : 0.0f; }

// grid (split_s, B).  Per output pixel: the gradient of recon_im (the caller's, e.g. from the perceptual loss, plus the
// masked-L1 terms) goes back through the clamp and the bilinear lookup to the texture (float atomics on the four taps:
// grad_texture arrives zeroed) and to the sampling position, from there through the inverse warp to recon_depth and
// the view, and through clamp / pooling / flip to the raster's depth map (every pixel of grad_depth_map is written).
__global__ void __launch_bounds__(256) k_g2s_sample_backward(G2S g) {
    __shared__ float s_buf[4 * G2S_SBACK_SUMS];
    const int b = blockIdx.y, s = g.s, npx = s * s, HW = g.H * g.W, S = g.S;
    const int h = (g.flip && b >= g.Bh) ? 1 : 0, bt = b - h * g.Bh;
    const float den = g.scratch[2 + h];
    const float gl1 = (h ? g2s_scalar(g.g_l1_flip) : g2s_scalar(g.g_l1)) + g2s_scalar(g.g_total);
    const float l1_scale = g.target ? gl1 / (3.0f * den) : 0.0f;
    const float* iK = cam_ptr(g.inv_K, g.invK_b, b, 9);
    const float* K = cam_ptr(g.K, g.K_b, b, 9);
    G2SView view;
    g2s_inverse_view(g.rot + (size_t)b * 9, g.trans + (size_t)b * 3, view);
    float acc[G2S_SBACK_SUMS];
#pragma unroll
    for (int k = 0; k < G2S_SBACK_SUMS; k++) acc[k] = 0.0f;
    for (int pix = blockIdx.x * 256 + threadIdx.x; pix < npx; pix += gridDim.x * 256) {
        const int yo = pix / s, xo = pix - yo * s;
        float pooled;
        const float rd = g2s_recon_depth(g, b, yo, xo, pooled);
        const float m = g.recon_mask[(size_t)b * npx + pix];
        float ray[3], p[3], q[3], uv[2], w[4];
        g2s_grid(g, iK, K, view, xo, yo, rd, ray, p, q, uv);
        Bilinear bl;
        int o[4];
        bl.at(uv[0], uv[1], g.W, g.H);
        bl.weights(w);
        bl.offsets(g.W, g.H, o);
        float gix = 0.0f, giy = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const size_t plane = ((size_t)b * 3 + c) * HW;
            float t[4], raw = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; k++) { t[k] = o[k] >= 0 ? g.texture[plane + o[k]] : 0.0f; raw += t[k] * w[k]; }
            const float v = fminf(fmaxf(raw, -1.0f), 1.0f);
            float gc = g.grad_recon_im ? g.grad_recon_im[((size_t)b * 3 + c) * npx + pix] : 0.0f;
            if (g.target) {
                const float dlt = v - g.target[((size_t)bt * 3 + c) * npx + pix];
                gc += (dlt > 0.0f ? 1.0f : (dlt < 0.0f ? -1.0f : 0.0f)) * m * l1_scale;
            }
            if (raw < -1.0f || raw > 1.0f) gc = 0.0f;                      // clamp(min=-1, max=1), G2S:483
            if (gc != 0.0f) {
#pragma unroll
                for (int k = 0; k < 4; k++) if (o[k] >= 0) atomicAdd(&g.grad_texture[plane + o[k]], w[k] * gc);
                gix += gc * ((t[1] - t[0]) * (1.0f - bl.fy) + (t[3] - t[2]) * bl.fy);
                giy += gc * ((t[2] - t[0]) * (1.0f - bl.fx) + (t[3] - t[1]) * bl.fx);
            }
        }
        // pixel coordinate -> normalised grid (align_corners = False) -> image plane -> moved point (CR:82-88)
        const float gu = gix * (float)g.W / 2.0f * 2.0f / (float)(s - 1), gv = giy * (float)g.H / 2.0f * 2.0f / (float)(s - 1);
        const float gxn = gu * K[0] + gv * K[3], gyn = gu * K[1] + gv * K[4];
        const float iz = 1.0f / q[2];
        const float gq[3] = {gxn * iz, gyn * iz, -(gxn * q[0] + gyn * q[1]) * iz * iz};
        float gp[3];
#pragma unroll
        for (int j = 0; j < 3; j++) gp[j] = gq[0] * view.A[j] + gq[1] * view.A[3 + j] + gq[2] * view.A[6 + j];
        float gd = gp[0] * ray[0] + gp[1] * ray[1] + gp[2] * ray[2];
        if (!(pooled >= g.depth_lo && pooled <= g.depth_hi)) gd = 0.0f;      // the clamp of CR:124
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
            for (int j = 0; j < 3; j++) acc[3 * k + j] += gq[k] * p[j];
            acc[9 + k] += gq[k];
        }
        const int n = g.aa ? 2 : 1;
        const float gpx = gd * (g.aa ? 0.25f : 1.0f);
        for (int dy = 0; dy < n; dy++)
            for (int dx = 0; dx < n; dx++)
                g.grad_depth_map[((size_t)b * S + (S - 1 - (yo * n + dy))) * S + xo * n + dx] = gpx;
    }
    block_sums_256<G2S_SBACK_SUMS>(acc, s_buf);
    if (threadIdx.x < G2S_SBACK_SUMS)
        g.scratch[g.off_sback + G2S_SBACK_SUMS * ((size_t)b * gridDim.x + blockIdx.x) + threadIdx.x] = acc[threadIdx.x];
}

// K6 (KCU:543-592), gathered: the candidates of 64 triangle pairs are walked as in k_g2s_raster; a pixel that the
// z-buffer says belongs to the candidate's triangle adds its share to the triangle's nine sums (LDS float atomics), and
// lane j stores pair j's sums per TRIANGLE, in the triangle's own vertex order (zeros when it owns nothing): grad_tri
// [B,Ft,3,3].  No global atomics; the vertices collect their six incident triangles in the next pass.  Depth and weights
// of a pixel are those of the forward pass (weights_depth recomputes them with the same operations).
__global__ void __launch_bounds__(256) k_g2s_depth_faces(G2S g) {
    __shared__ G2SStage s_stage[4];
    __shared__ float s_acc_all[4][9][G2S_PW];
    G2SStage& st = s_stage[threadIdx.x >> 6];
    float (&s_acc)[9][G2S_PW] = s_acc_all[threadIdx.x >> 6];
    const int Ft = 2 * (g.H - 1) * (g.W - 1), S = g.S, lane = lane_id();
    const long pair = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * G2S_PW + lane;
    bool reversed;
    const int cnt = g2s_stage_pair(g, st, pair, Ft, reversed);
    if (lane < G2S_PW) {
#pragma unroll
        for (int k = 0; k < 9; k++) s_acc[k][lane] = 0.0f;
    }
    auto pixel_of = [&](int lo, int xi, int yi) {
        const long owner_pair = pair - lane + lo;
        return ((size_t)(owner_pair / Ft) * S + yi) * S + xi;
    };
    bid_rows(st, cnt, S,
        [&](int lo, int xi, int yi) { return bid_face(g.zbuf[pixel_of(lo, xi, yi)]) == st.fid[lo]; },
        [&](int lo, int xi, int yi) {
            const size_t p = pixel_of(lo, xi, yi);
            float face[9], finv[9], w[3], zp;
#pragma unroll
            for (int k = 0; k < 9; k++) { face[k] = st.face[k][lo]; finv[k] = st.finv[k][lo]; }
            weights_depth(face, finv, xi, yi, g.near, g.far, w, zp);
            float tmp[2] = {0, 0};
#pragma unroll
            for (int k = 0; k < 2; k++) {
#pragma unroll
                for (int l = 0; l < 3; l++) tmp[k] += -finv[3 * l + k] / face[3 * l + 2];     // KCU:582
            }
            const float gd = g.grad_depth_map[p], depth2 = zp * zp;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float z_k = face[3 * k + 2];
                atomicAdd(&s_acc[3 * k + 0][lo], -gd * tmp[0] * w[k] * depth2 * (float)S / 2.0f);          // KCU:588
                atomicAdd(&s_acc[3 * k + 1][lo], -gd * tmp[1] * w[k] * depth2 * (float)S / 2.0f);
                atomicAdd(&s_acc[3 * k + 2][lo], gd * w[k] * depth2 / (z_k * z_k));                        // KCU:575
            }
        });
    wave_lds_sync();
    if (lane < G2S_PW && pair < (long)g.B * Ft) {
        float* o = g.grad_tri + (size_t)pair * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) o[k] = s_acc[reversed ? (2 - k / 3) * 3 + k % 3 : k][lane];
    }
}

// gradient of mesh vertex (y, x) of view b: the sum over its (up to six) incident triangles of the implicit topology
// (tri_ids: cell (cy, cx) carries (tl, bl, tr) and, Ft/2 later, (tr, bl, br))
__device__ __forceinline__ void g2s_vertex_gradient(const G2S& g, int b, int y, int x, float* gv) {
    const int Wc = g.W - 1, cells = (g.H - 1) * Wc;
    const float* gt = g.grad_tri + (size_t)b * 2 * cells * 9;
    gv[0] = gv[1] = gv[2] = 0.0f;
    auto add = [&](int second, int cy, int cx, int slot) {
        if (cy < 0 || cy >= g.H - 1 || cx < 0 || cx >= Wc) return;
        const float* t = gt + ((size_t)(second ? cells : 0) + (size_t)cy * Wc + cx) * 9 + 3 * slot;
        gv[0] += t[0]; gv[1] += t[1]; gv[2] += t[2];
    };
    add(0, y, x, 0);                            // tl of its own cell
    add(0, y, x - 1, 2); add(1, y, x - 1, 0);   // tr of the cell to the left
    add(0, y - 1, x, 1); add(1, y - 1, x, 1);   // bl of the cell above
    add(1, y - 1, x - 1, 2);                    // br of the cell above-left
}

// grid (split_f, B).  Per canonical pixel: what the mesh path left on the pixel's vertex goes back through the camera and
// the rigid motion to the depth (grad_depth_mesh) and the view; what the lookup left on the texture goes to the albedo,
// the light and -- together with the smooth loss of the shading -- to the normal (grad_normal, consumed by the next pass).
__global__ void __launch_bounds__(256) k_g2s_front_backward(G2S g) {
    __shared__ float s_buf[4 * G2S_FRONT_SUMS];
    const int b = blockIdx.y, HW = g.H * g.W;
    const float* iK = cam_ptr(g.inv_K, g.invK_b, b, 9);
    const float* dview = g.depth + (size_t)b * HW;
    const float* R = g.rot + (size_t)b * 9;
    const float* t = g.trans + (size_t)b * 3;
    const float* ld = g.light_d + 3 * b;
    const float la = g.light_a[b], lb = g.light_b[b];
    const float g_sm = g.with_smooth ? g2s_scalar(g.g_smooth) + g.lam_smooth * g2s_scalar(g.g_total) : 0.0f;
    float acc[G2S_FRONT_SUMS];
#pragma unroll
    for (int k = 0; k < G2S_FRONT_SUMS; k++) acc[k] = 0.0f;
    for (int pix = blockIdx.x * 256 + threadIdx.x; pix < HW; pix += gridDim.x * 256) {
        const int y = pix / g.W, x = pix - y * g.W;
        const size_t i = (size_t)b * HW + pix;
        // mesh path
        float ray[3], p[3], q[3], gv[3], gq[3], gp[3];
        gw_ray(iK, (float)x, (float)y, ray);
        const float d = dview[pix];
#pragma unroll
        for (int k = 0; k < 3; k++) p[k] = ray[k] * d;
        gw_rigid(p, R, t, g.center_z, q);
        g2s_vertex_gradient(g, b, y, x, gv);
        camera_point_adjoint(g.cam, b, q, gv, gq);
#pragma unroll
        for (int j = 0; j < 3; j++) gp[j] = gq[0] * R[j] + gq[1] * R[3 + j] + gq[2] * R[6 + j];
        g.grad_depth_mesh[i] = gp[0] * ray[0] + gp[1] * ray[1] + gp[2] * ray[2];
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
            for (int j = 0; j < 3; j++) acc[3 * k + j] += gq[k] * p[j];
            acc[9 + k] += gq[k];
        }
        // shading path
        float n[3];
        g2s_normal(dview, iK, g.H, g.W, x, y, n);
        const float dot = (n[0] * ld[0] + n[1] * ld[1]) + n[2] * ld[2];
        const float diff = fmaxf(dot, 0.0f), shading = la + lb * diff;
        float g_sh = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const size_t o = ((size_t)b * 3 + c) * HW + pix;
            const float gt = g.grad_texture[o];
            // the accumulator's one reader hands it back ZEROED: k_g2s_sample_backward ADDS into it, and a second backward
            // pass over the same forward (retain_graph, torch.autograd.grad per loss term) must not see the first one's sums
            g.grad_texture[o] = 0.0f;
            if (g.grad_albedo) g.grad_albedo[o] = gt * shading;           // d/d albedo of (albedo/2 + 0.5) * shading * 2 - 1
            g_sh += gt * ((g.albedo[o] / 2.0f + 0.5f) * 2.0f);
        }
        acc[12] += g_sh;
        acc[13] += g_sh * diff;
        float g_diff = g_sh * lb;
        if (g.with_smooth) g_diff += g_sm * sm_grad_at(g.diffuse + (i - x), x, y, g.H, g.W, g.n_xx, g.n_xy, g.n_yy);
        const float g_dot = dot >= 0.0f ? g_diff : 0.0f;                   // clamp(min=0)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            acc[14 + k] += g_dot * n[k];
            g.grad_normal[3 * i + k] = g_dot * ld[k];
        }
    }
    block_sums_256<G2S_FRONT_SUMS>(acc, s_buf);
    if (threadIdx.x < G2S_FRONT_SUMS)
        g.scratch[g.off_front + G2S_FRONT_SUMS * ((size_t)b * gridDim.x + blockIdx.x) + threadIdx.x] = acc[threadIdx.x];
}

// per canonical pixel: the normals' adjoint gathered from the four neighbours (as k_depth_normals_backward), the mesh
// path's share and the smooth loss of the depth map itself
// ... and, as the last B workgroups of the same grid (round 5: they read the partial sums of the two passes in FRONT of this
// one and nothing of its own -- a launch of ~4 us less on a nine-launch step), k_g2s_finish_backward's body, below.
__device__ __forceinline__ void g2s_finish_backward(const G2S& g, int b);
__global__ void __launch_bounds__(256) k_g2s_depth_backward(G2S g) {
    const int HW = g.H * g.W, H = g.H, W = g.W;
    const unsigned nb_depth = gridDim.x - (unsigned)g.B;
    if (blockIdx.x >= nb_depth) { g2s_finish_backward(g, (int)(blockIdx.x - nb_depth)); return; }
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)g.B * HW) return;
    const int b = (int)(i / HW), pix = (int)(i - (long)b * HW);
    const int y = pix / W, x = pix - y * W;
    const float* iK = cam_ptr(g.inv_K, g.invK_b, b, 9);
    const float* dview = g.depth + (size_t)b * HW;
    const float* gview = g.grad_normal + (size_t)b * HW * 3;
    float gp[3] = {0, 0, 0}, a[3], c[3];
    if (x - 1 >= 0) { dn_tangent_grads(dview, iK, gview, H, W, x - 1, y, a, c); for (int k = 0; k < 3; k++) gp[k] += a[k]; }
    if (x + 1 < W)  { dn_tangent_grads(dview, iK, gview, H, W, x + 1, y, a, c); for (int k = 0; k < 3; k++) gp[k] -= a[k]; }
    if (y - 1 >= 0) { dn_tangent_grads(dview, iK, gview, H, W, x, y - 1, a, c); for (int k = 0; k < 3; k++) gp[k] += c[k]; }
    if (y + 1 < H)  { dn_tangent_grads(dview, iK, gview, H, W, x, y + 1, a, c); for (int k = 0; k < 3; k++) gp[k] -= c[k]; }
    float ray[3];
    gw_ray(iK, (float)x, (float)y, ray);
    float gd = (gp[0] * ray[0] + gp[1] * ray[1] + gp[2] * ray[2]) + g.grad_depth_mesh[i];
    if (g.with_smooth) {
        const float g_sm = g2s_scalar(g.g_smooth) + g.lam_smooth * g2s_scalar(g.g_total);
        gd += g_sm * sm_grad_at(g.depth + (i - x), x, y, H, W, g.n_xx, g.n_xy, g.n_yy);
    }
    g.grad_depth[i] = gd;
}

// one workgroup per batch entry: add up the passes' partial sums (eight lanes per sum); the gradient of the inverse
// view's (A', t') = (R^T, -(t R)) joins that of (R, t); with a view vector, (R, t)'s gradient goes on to it
// (k_view_transform_backward's arithmetic)
__device__ __forceinline__ void g2s_finish_backward(const G2S& g, int b) {      // (all 256 threads of a workgroup)
    __shared__ float s_sum[32];
    const int j = threadIdx.x >> 3, sub = threadIdx.x & 7;      // sum j of 29, eight lanes each
    float v = 0.0f;
    if (j < G2S_FRONT_SUMS) {
        for (int k = sub; k < g.split_f; k += 8) v += g.scratch[g.off_front + G2S_FRONT_SUMS * ((size_t)b * g.split_f + k) + j];
    } else if (j < G2S_FRONT_SUMS + G2S_SBACK_SUMS) {
        for (int k = sub; k < g.split_s; k += 8)
            v += g.scratch[g.off_sback + G2S_SBACK_SUMS * ((size_t)b * g.split_s + k) + (j - G2S_FRONT_SUMS)];
    }
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    if (sub == 0) s_sum[j] = v;
    __syncthreads();
    const float* F = s_sum;
    const float* Sb = s_sum + G2S_FRONT_SUMS;
    const float* R = g.rot + (size_t)b * 9;
    const float* t = g.trans + (size_t)b * 3;
    const int lane = threadIdx.x;
    float g_rot[9], g_trans[3];
#pragma unroll
    for (int q = 0; q < 9; q++) {
        const int k = q / 3, i = q % 3;                   // R[k][i] enters A'[i][k] and t'[i] = -sum_k t[k] R[k][i]
        g_rot[q] = F[q] + Sb[3 * i + k] - t[k] * Sb[9 + i];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) g_trans[k] = F[9 + k] - ((Sb[9] * R[3 * k] + Sb[10] * R[3 * k + 1]) + Sb[11] * R[3 * k + 2]);
    if (lane < 9) {
        if (g.grad_rot) g.grad_rot[(size_t)b * 9 + lane] = g_rot[lane];
    } else if (lane < 12) {
        if (g.grad_trans) g.grad_trans[(size_t)b * 3 + (lane - 9)] = g_trans[lane - 9];
    } else if (lane == 12) {
        if (g.grad_light_a) g.grad_light_a[b] = F[12];
    } else if (lane == 13) {
        if (g.grad_light_b) g.grad_light_b[b] = F[13];
    } else if (lane < 17) {
        if (g.grad_light_d) g.grad_light_d[(size_t)b * 3 + (lane - 14)] = F[lane];
    } else if (lane == 17 && g.view && g.grad_view) {
        const float* vw = g.view + (size_t)b * g.view_n;
        const float cx = cosf(vw[0]), sx = sinf(vw[0]), cy = cosf(vw[1]), sy = sinf(vw[1]), cz = cosf(vw[2]), sz = sinf(vw[2]);
        float mx[9], my[9], mz[9], t0[9], t1[9];
        euler_factors(cx, sx, cy, sy, cz, sz, mx, my, mz);
        const float dx[9] = {0, 0, 0, 0, -sx, -cx, 0, cx, -sx};
        const float dy[9] = {-sy, 0, cy, 0, 0, 0, -cy, 0, -sy};
        const float dz[9] = {-sz, -cz, 0, cz, -sz, 0, 0, 0, 0};
        float out[3] = {0, 0, 0};
        mat3_mul(my, dx, t0); mat3_mul(mz, t0, t1);          // dR/drx = Rz Ry Rx'
        for (int k = 0; k < 9; k++) out[0] += g_rot[k] * t1[k];
        mat3_mul(dy, mx, t0); mat3_mul(mz, t0, t1);          // dR/dry = Rz Ry' Rx
        for (int k = 0; k < 9; k++) out[1] += g_rot[k] * t1[k];
        mat3_mul(my, mx, t0); mat3_mul(dz, t0, t1);          // dR/drz = Rz' Ry Rx
        for (int k = 0; k < 9; k++) out[2] += g_rot[k] * t1[k];
        float* o = g.grad_view + (size_t)b * g.view_n;
        o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
        for (int k = 3; k < g.view_n; k++) o[k] = g_trans[k - 3];
    }
}

// ---- NrRenderer's grid_sample frames (CR:180-184, 219-222, 263-267) as one pass each way ---------------------------------
//   grid = get_inv_warped_2d_grid(recon_depth)  (any composed rigid motion (A, t));  out = F.grid_sample(src, grid,
//   'bilinear');  out_nearest = F.grid_sample(src_nearest, grid, 'nearest')  -- zeros padding, align_corners = False.
struct WarpResample {
    int B, h, w;                  // recon_depth / outputs [B,h,w]
    int C, Cn, H, W;              // src [B,C,H,W], src_nearest [B,Cn,H,W]
    const float* depth;
    const float* inv_K; int invK_b;
    const float* K; int K_b;
    const float *rot, *trans;     // [B,3,3], [B,3]
    float center_z;
    const float *src, *src_nearest;
    float *out, *out_nearest;
    // backward
    const float* grad_out;
    float *grad_src, *grad_depth, *partials;     // partials [B][gridDim.x][12]
};

__device__ __forceinline__ void wr_grid(const WarpResample& g, int b, int xo, int yo, float* ray, float* p, float* q, float* uv) {
    gw_ray(cam_ptr(g.inv_K, g.invK_b, b, 9), (float)xo, (float)yo, ray);
    const float d = g.depth[((size_t)b * g.h + yo) * g.w + xo];
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = ray[k] * d;
    gw_rigid(p, g.rot + (size_t)b * 9, g.trans + (size_t)b * 3, g.center_z, q);
    gw_project(q, cam_ptr(g.K, g.K_b, b, 9), g.w, g.h, uv);
}

__global__ void __launch_bounds__(256) k_warp_resample(WarpResample g) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int npx = g.h * g.w, HW = g.H * g.W;
    if (i >= (long)g.B * npx) return;
    const int b = (int)(i / npx), pix = (int)(i - (long)b * npx);
    const int yo = pix / g.w, xo = pix - yo * g.w;
    float ray[3], p[3], q[3], uv[2], w[4];
    wr_grid(g, b, xo, yo, ray, p, q, uv);
    Bilinear bl;
    int o[4];
    bl.at(uv[0], uv[1], g.W, g.H);
    bl.weights(w);
    bl.offsets(g.W, g.H, o);
    for (int c = 0; c < g.C; c++) {
        const float* plane = g.src + ((size_t)b * g.C + c) * HW;
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; k++) if (o[k] >= 0) v += plane[o[k]] * w[k];
        g.out[((size_t)b * g.C + c) * npx + pix] = v;
    }
    if (g.src_nearest) {
        // mode='nearest': the pixel coordinate rounded half to even (nearbyint), zero outside the image
        const float ix = ((uv[0] + 1.0f) * (float)g.W - 1.0f) / 2.0f, iy = ((uv[1] + 1.0f) * (float)g.H - 1.0f) / 2.0f;
        const float rx = rintf(ix), ry = rintf(iy);
        const bool in = rx >= 0.0f && rx <= (float)(g.W - 1) && ry >= 0.0f && ry <= (float)(g.H - 1);
        for (int c = 0; c < g.Cn; c++)
            g.out_nearest[((size_t)b * g.Cn + c) * npx + pix] = in ? g.src_nearest[((size_t)b * g.Cn + c) * HW + (int)ry * g.W + (int)rx] : 0.0f;
    }
}

// grid (split, B): grad_src += (atomics, caller zeroes), grad_depth written, (A, t) partial sums per workgroup
__global__ void __launch_bounds__(256) k_warp_resample_backward(WarpResample g) {
    __shared__ float s_buf[4 * 12];
    const int b = blockIdx.y, npx = g.h * g.w, HW = g.H * g.W;
    const float* K = cam_ptr(g.K, g.K_b, b, 9);
    const float* A = g.rot + (size_t)b * 9;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; k++) acc[k] = 0.0f;
    for (int pix = blockIdx.x * 256 + threadIdx.x; pix < npx; pix += gridDim.x * 256) {
        const int yo = pix / g.w, xo = pix - yo * g.w;
        float ray[3], p[3], q[3], uv[2], w[4];
        wr_grid(g, b, xo, yo, ray, p, q, uv);
        Bilinear bl;
        int o[4];
        bl.at(uv[0], uv[1], g.W, g.H);
        bl.weights(w);
        bl.offsets(g.W, g.H, o);
        float gix = 0.0f, giy = 0.0f;
        for (int c = 0; c < g.C; c++) {
            const size_t plane = ((size_t)b * g.C + c) * HW;
            const float gc = g.grad_out[((size_t)b * g.C + c) * npx + pix];
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                t[k] = o[k] >= 0 ? g.src[plane + o[k]] : 0.0f;
                if (g.grad_src && o[k] >= 0 && gc != 0.0f) atomicAdd(&g.grad_src[plane + o[k]], w[k] * gc);
            }
            gix += gc * ((t[1] - t[0]) * (1.0f - bl.fy) + (t[3] - t[2]) * bl.fy);
            giy += gc * ((t[2] - t[0]) * (1.0f - bl.fx) + (t[3] - t[1]) * bl.fx);
        }
        const float gu = gix * (float)g.W / 2.0f * 2.0f / (float)(g.w - 1), gv = giy * (float)g.H / 2.0f * 2.0f / (float)(g.h - 1);
        const float gxn = gu * K[0] + gv * K[3], gyn = gu * K[1] + gv * K[4];
        const float iz = 1.0f / q[2];
        const float gq[3] = {gxn * iz, gyn * iz, -(gxn * q[0] + gyn * q[1]) * iz * iz};
        float gp[3];
#pragma unroll
        for (int j = 0; j < 3; j++) gp[j] = gq[0] * A[j] + gq[1] * A[3 + j] + gq[2] * A[6 + j];
        if (g.grad_depth) g.grad_depth[(size_t)b * npx + pix] = gp[0] * ray[0] + gp[1] * ray[1] + gp[2] * ray[2];
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
            for (int j = 0; j < 3; j++) acc[3 * k + j] += gq[k] * p[j];
            acc[9 + k] += gq[k];
        }
    }
    block_sums_256<12>(acc, s_buf);
    if (threadIdx.x < 12) g.partials[12 * ((size_t)b * gridDim.x + blockIdx.x) + threadIdx.x] = acc[threadIdx.x];
}

}  // namespace d3m
