// d3m_g2s.h -- the renderer block of the gan2shape training step (deep3dmap/models/frameworks/gan2shape.py:463-497,
// "G2S" below; NrRenderer = deep3dmap/core/renderer/renderer_nr.py, "CR") as a handful of fused passes:
//
//   forward   k_g2s_front     per canonical pixel: normal (CR:127-139) -> diffuse shading -> texture (G2S:463-466), and the
//                             pixel as a mesh vertex: back-projection, the view's rigid motion (CR:90-100), the mesh
//                             renderer's camera (NR/projection.py); clears the accumulators of the backward pass
//             k_g2s_smooth    the second-difference sums of smooth_loss(depth) + smooth_loss(diffuse_shading) (G2S:493-494)
//             (binning + tile pass of d3m_forward.h on the implicit grid topology: warp_canon_depth, CR:116-125)
//             k_g2s_sample    per output pixel: pooled / flipped / clamped recon_depth (NR/rasterize.py:305-326, CR:122-124),
//                             border mask (G2S:478-482), inverse-warped sampling position (CR:102-114), bilinear lookup of
//                             the texture (F.grid_sample, G2S:483), clamp, masked-L1 sums (G2S:486,489)
//             k_g2s_finish    the loss values
//   backward  k_g2s_sample_backward -> k_backward_depth_map (K6, KCU:543-592, onto the grid's vertices) ->
//             k_g2s_front_backward -> k_g2s_depth_backward -> k_g2s_finish_backward
//
// The reference runs this block as ~150 eager kernels forward and as many again in backward.  Every map of the block is
// a few hundred KB, so the passes are latency-bound: what counts is the number of launches and that nothing is
// materialised between them ([B,F,3,3] faces aside, which the tile pass reads).
#pragma once
#include "d3m_aux.h"
#include "d3m_backward.h"
#include "d3m_forward.h"

namespace d3m {

// Device view of d3m_g2s_block (include/d3m_raster.h).
struct G2S {
    int B, H, W;                  // canonical maps [B,H,W]
    int s, S, aa;                 // output images s x s; raster S x S (S = 2s with anti-aliasing)
    int Bh, flip;                 // flip: B = 2 Bh and entries (b, b + Bh) share the product of their border masks
    const float* inv_K; int invK_b;
    const float* K; int K_b;      // NrRenderer.K / inv_K (CR:35-46)
    float center_z, depth_lo, depth_hi;
    Cam cam;                      // the mesh renderer's camera (CR:47-54: projection)
    const float *depth, *albedo, *light_a, *light_b, *light_d, *rot, *trans, *target, *extra_mask;
    float *normal, *diffuse, *texture, *screen_vertices;
    const float* depth_map;       // [B,S,S] raster output, row 0 = bottom
    float *recon_depth, *recon_im, *recon_mask, *losses;
    float* scratch;
    int off_sample, off_smooth, off_front, off_sback;    // partial sums inside scratch (floats)
    int split_s, split_m, split_f;                       // workgroups per batch entry (sample / front) and per map (smooth)
    float lam_smooth; int with_smooth;
    // backward
    const float *grad_recon_im, *g_l1, *g_l1_flip, *g_smooth, *g_total;
    float *grad_texture, *grad_vertices, *grad_depth_map, *grad_normal, *grad_depth_mesh;
    float *grad_depth, *grad_albedo, *grad_light_a, *grad_light_b, *grad_light_d, *grad_rot, *grad_trans;
};

constexpr int G2S_TOTALS = 16;          // scratch[0..16): num1, num2, den1, den2, then the 2 x 4 smooth sums
constexpr int G2S_FRONT_SUMS = 17;      // grad_rot 9, grad_trans 3, light_a, light_b, light_d 3
constexpr int G2S_SAMPLE_SUMS = 12;     // gradient of the inverse view's (A', t')

struct ZeroRanges { uint32_t* p[3]; unsigned n[3]; };       // words

// ---- shared pieces -------------------------------------------------------------------------------------------------
// unit normal of the back-projected depth map at (x, y) (CR:127-139); n_raw = tu x tv before normalisation
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

// recon_depth of output pixel (yo, xo): the raster's depth map flipped and 2x2-pooled as rasterize_rgbad does
// (NR/rasterize.py:305-326, same summation order as k_output_epilogue), then clamped (CR:122-124).  `pooled` = before.
__device__ __forceinline__ float g2s_recon_depth(const G2S& g, int b, int yo, int xo, float& pooled) {
    const int S = g.S, n = g.aa ? 2 : 1;
    float acc = 0.0f;
    for (int dy = 0; dy < n; dy++)
        for (int dx = 0; dx < n; dx++)
            acc += g.depth_map[((size_t)b * S + (S - 1 - (yo * n + dy))) * S + xo * n + dx];
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

// ---- forward ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_g2s_front(G2S g, ZeroRanges z) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
#pragma unroll
    for (int r = 0; r < 3; r++)
        for (long k = i; k < (long)z.n[r]; k += stride) z.p[r][k] = 0u;
    const int HW = g.H * g.W;
    if (i >= (long)g.B * HW) return;
    const int b = (int)(i / HW), pix = (int)(i - (long)b * HW);
    const int y = pix / g.W, x = pix - y * g.W;
    const float* iK = cam_ptr(g.inv_K, g.invK_b, b, 9);
    const float* dview = g.depth + (size_t)b * HW;
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
    gw_rigid(p, g.rot + (size_t)b * 9, g.trans + (size_t)b * 3, g.center_z, q);
    camera_point(g.cam, b, q, o, nullptr);
    g.screen_vertices[3 * i] = o[0]; g.screen_vertices[3 * i + 1] = o[1]; g.screen_vertices[3 * i + 2] = o[2];
}

// smooth_loss sums of the depth map (blockIdx.y = 0) and of the diffuse shading (1): partials [2][split_m][4]
__global__ void __launch_bounds__(256) k_g2s_smooth(G2S g) {
    __shared__ float s_part[4];
    const float* pred = blockIdx.y == 0 ? g.depth : g.diffuse;
    const int H = g.H, W = g.W;
    float acc[4] = {0, 0, 0, 0};
    const long n = (long)g.B * H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const float* r0 = pred + (i - x);
        if (x + 2 < W) acc[0] += fabsf(sm_dxx(r0, x));
        if (y + 1 < H && x + 1 < W) {
            acc[1] += fabsf(sm_dxy(r0, r0 + W, x));
            acc[2] += fabsf(sm_dyx(r0, r0 + W, x));
        }
        if (y + 2 < H) acc[3] += fabsf(sm_dyy(r0, r0 + W, r0 + 2 * W, x));
    }
#pragma unroll
    for (int k = 0; k < 4; k++) acc[k] = block_sum_256(acc[k], s_part);
    if (threadIdx.x == 0) {
        float* o = g.scratch + g.off_smooth + 4 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = acc[k];
    }
}

// grid (split_s, Bh): with flip a lane handles its pixel in both halves of the batch, which share the mask product
__global__ void __launch_bounds__(256) k_g2s_sample(G2S g) {
    __shared__ float s_part[4];
    const int s = g.s, npx = s * s, HW = g.H * g.W, halves = g.flip ? 2 : 1;
    float num[2] = {0, 0}, den[2] = {0, 0};
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
                if (g.target) num[h] += fabsf(v - g.target[((size_t)blockIdx.y * 3 + c) * npx + pix]) * mh;
            }
            den[h] += mh;
        }
    }
    float sums[4] = {num[0], num[1], den[0], den[1]};
#pragma unroll
    for (int k = 0; k < 4; k++) sums[k] = block_sum_256(sums[k], s_part);
    if (threadIdx.x == 0) {
        float* o = g.scratch + g.off_sample + 4 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = sums[k];
    }
}

// one workgroup: losses = (l1, l1_flip, smooth, l1 + l1_flip + lam_smooth * smooth); totals kept for backward
__global__ void __launch_bounds__(256) k_g2s_finish(G2S g, float n_xx, float n_xy, float n_yy) {
    __shared__ float s_part[4];
    float acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int n_s = g.Bh * g.split_s;
    for (int i = threadIdx.x; i < n_s; i += 256) {
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] += g.scratch[g.off_sample + 4 * i + k];
    }
    if (g.with_smooth) {
        for (int i = threadIdx.x; i < g.split_m; i += 256) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                acc[4 + k] += g.scratch[g.off_smooth + 4 * i + k];
                acc[8 + k] += g.scratch[g.off_smooth + 4 * (g.split_m + i) + k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 12; k++) acc[k] = block_sum_256(acc[k], s_part);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 12; k++) g.scratch[k] = acc[k];
        const float l1 = g.target ? acc[0] / (3.0f * acc[2]) : 0.0f;            // photometric_loss: the mask expands to 3 channels
        const float l1f = (g.target && g.flip) ? acc[1] / (3.0f * acc[3]) : 0.0f;
        float sm = 0.0f;
        if (g.with_smooth)
            sm = (((acc[4] / n_xx + acc[5] / n_xy) + acc[6] / n_xy) + acc[7] / n_yy) +
                 (((acc[8] / n_xx + acc[9] / n_xy) + acc[10] / n_xy) + acc[11] / n_yy);
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
    __shared__ float s_part[4];
    const int b = blockIdx.y, s = g.s, npx = s * s, HW = g.H * g.W, S = g.S;
    const int h = (g.flip && b >= g.Bh) ? 1 : 0, bt = b - h * g.Bh;
    const float den = g.scratch[2 + h];
    const float gl1 = (h ? g2s_scalar(g.g_l1_flip) : g2s_scalar(g.g_l1)) + g2s_scalar(g.g_total);
    const float l1_scale = g.target ? gl1 / (3.0f * den) : 0.0f;
    const float* iK = cam_ptr(g.inv_K, g.invK_b, b, 9);
    const float* K = cam_ptr(g.K, g.K_b, b, 9);
    G2SView view;
    g2s_inverse_view(g.rot + (size_t)b * 9, g.trans + (size_t)b * 3, view);
    float acc[G2S_SAMPLE_SUMS];
#pragma unroll
    for (int k = 0; k < G2S_SAMPLE_SUMS; k++) acc[k] = 0.0f;
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
    float* out = g.scratch + g.off_sback + (size_t)G2S_SAMPLE_SUMS * ((size_t)b * gridDim.x + blockIdx.x);
#pragma unroll
    for (int k = 0; k < G2S_SAMPLE_SUMS; k++) {
        const float v = block_sum_256(acc[k], s_part);
        if (threadIdx.x == 0) out[k] = v;
    }
}

// grid (split_f, B).  Per canonical pixel: what the mesh path left on the pixel's vertex goes back through the camera and
// the rigid motion to the depth (grad_depth_mesh) and the view; what the lookup left on the texture goes to the albedo,
// the light and -- together with the smooth loss of the shading -- to the normal (grad_normal, consumed by the next pass).
__global__ void __launch_bounds__(256) k_g2s_front_backward(G2S g, float n_xx, float n_xy, float n_yy) {
    __shared__ float s_part[4];
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
        float ray[3], p[3], q[3], gq[3], gp[3];
        gw_ray(iK, (float)x, (float)y, ray);
        const float d = dview[pix];
#pragma unroll
        for (int k = 0; k < 3; k++) p[k] = ray[k] * d;
        gw_rigid(p, R, t, g.center_z, q);
        const float gv[3] = {g.grad_vertices[3 * i], g.grad_vertices[3 * i + 1], g.grad_vertices[3 * i + 2]};
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
            g.grad_albedo[o] = gt * shading;                              // d/d albedo of (albedo/2 + 0.5) * shading * 2 - 1
            g_sh += gt * ((g.albedo[o] / 2.0f + 0.5f) * 2.0f);
        }
        acc[12] += g_sh;
        acc[13] += g_sh * diff;
        float g_diff = g_sh * lb;
        if (g.with_smooth) g_diff += g_sm * sm_grad_at(g.diffuse + (i - x), x, y, g.H, g.W, n_xx, n_xy, n_yy);
        const float g_dot = dot >= 0.0f ? g_diff : 0.0f;                   // clamp(min=0)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            acc[14 + k] += g_dot * n[k];
            g.grad_normal[3 * i + k] = g_dot * ld[k];
        }
    }
    float* out = g.scratch + g.off_front + (size_t)G2S_FRONT_SUMS * ((size_t)b * gridDim.x + blockIdx.x);
#pragma unroll
    for (int k = 0; k < G2S_FRONT_SUMS; k++) {
        const float v = block_sum_256(acc[k], s_part);
        if (threadIdx.x == 0) out[k] = v;
    }
}

// per canonical pixel: the normals' adjoint gathered from the four neighbours (as k_depth_normals_backward), the mesh
// path's share and the smooth loss of the depth map itself
__global__ void __launch_bounds__(256) k_g2s_depth_backward(G2S g, float n_xx, float n_xy, float n_yy) {
    const int HW = g.H * g.W, H = g.H, W = g.W;
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
        gd += g_sm * sm_grad_at(g.depth + (i - x), x, y, H, W, n_xx, n_xy, n_yy);
    }
    g.grad_depth[i] = gd;
}

// one wave per batch entry: add up the workgroups' partial sums; the gradient of the inverse view's (A', t') =
// (R^T, -(t R)) joins that of (R, t)
__global__ void __launch_bounds__(64) k_g2s_finish_backward(G2S g) {
    __shared__ float s_sum[G2S_FRONT_SUMS + G2S_SAMPLE_SUMS];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (lane < G2S_FRONT_SUMS) {
        float v = 0.0f;
        for (int k = 0; k < g.split_f; k++) v += g.scratch[g.off_front + G2S_FRONT_SUMS * ((size_t)b * g.split_f + k) + lane];
        s_sum[lane] = v;
    } else if (lane < G2S_FRONT_SUMS + G2S_SAMPLE_SUMS) {
        const int j = lane - G2S_FRONT_SUMS;
        float v = 0.0f;
        for (int k = 0; k < g.split_s; k++) v += g.scratch[g.off_sback + G2S_SAMPLE_SUMS * ((size_t)b * g.split_s + k) + j];
        s_sum[lane] = v;
    }
    __syncthreads();
    const float* F = s_sum;
    const float* Sb = s_sum + G2S_FRONT_SUMS;
    const float* R = g.rot + (size_t)b * 9;
    const float* t = g.trans + (size_t)b * 3;
    if (lane < 9) {
        const int k = lane / 3, i = lane % 3;                 // R[k][i] enters A'[i][k] and t'[i] = -sum_k t[k] R[k][i]
        if (g.grad_rot) g.grad_rot[(size_t)b * 9 + lane] = F[lane] + Sb[3 * i + k] - t[k] * Sb[9 + i];
    } else if (lane < 12) {
        const int k = lane - 9;
        if (g.grad_trans)
            g.grad_trans[(size_t)b * 3 + k] = F[lane] - ((Sb[9] * R[3 * k] + Sb[10] * R[3 * k + 1]) + Sb[11] * R[3 * k + 2]);
    } else if (lane == 12) {
        if (g.grad_light_a) g.grad_light_a[b] = F[12];
    } else if (lane == 13) {
        if (g.grad_light_b) g.grad_light_b[b] = F[13];
    } else if (lane < 17) {
        if (g.grad_light_d) g.grad_light_d[(size_t)b * 3 + (lane - 14)] = F[lane];
    }
}

}  // namespace d3m
