// d3m_uv.h -- UV-space unwrapping for deep3dmap's Pt3dRenderer.sample (deep3dmap/core/renderer/renderer_pt3d.py:46-98,
// caller models/frameworks/imgs2mesh.py:100-124): the template mesh, laid out flat in UV space, is rasterized hard (one
// face per pixel) and every covered pixel looks the input image up at the barycentric interpolation of its face's
// per-vertex image coordinates, shaded by a diffuse point light.
//
// The reference does this with pytorch3d (pinned 0.6.1, not vendored): MeshRasterizer(blur_radius 0, faces_per_pixel
// 1) -> TexturesUV.sample_textures (F.grid_sample bilinear, align_corners=True, padding "border", the map flipped
// vertically) -> SoftPhongShader.  The shader is constructed WITHOUT the lights / materials the reference builds
// (renderer_pt3d.py:42-43 vs :87-91), so the library's defaults shade: PointLights at (0, 1, 0) with ambient 0.5,
// diffuse 0.3, specular 0.2, Materials all 1 with shininess 64, i.e. colour = (0.5 + 0.3 relu(n.l)) * texel +
// 0.2 relu(v.r)^64 [n.l > 0] with v towards the camera at (0, 0, 2.7) and r the reflected light direction.  The
// parameters are fields here (the host passes those defaults).  Its arithmetic is restated from the library's
// published behaviour; no reference vector exists (parity unpinned); the checker is an independent
// brute-force restatement (oracle/pt3d_oracle.py).  Coverage comes from the tile rasterizer of d3m_forward.h (same maps: row 0 = bottom); this
// pass is the per-pixel epilogue and its adjoint for the image and the per-vertex image coordinates.
#pragma once
#include "d3m_device.h"

namespace d3m {

struct UvUnwrap {
    const int32_t* face_index_map;   // [B,T,T]   internal orientation (row 0 = bottom), fill_back face ids
    const float* weight_map;         // [B,T,T,3]
    const int32_t* tri;              // [F,3]
    const float* verts;              // [V,3]     template_uvs3d
    const float* vnormals;           // [V,3]     vertex normals of that mesh (area-weighted face normals, normalised)
    const float* uvs;                // [B,V,2]   face_project: image coordinates in 0..1, v up
    const float* imgs;               // [B,C,H,W]
    const int32_t* used;             // [B]       0: the reference's triangle filter left this batch entry no faces
    float light[3];                  // PointLights.location
    float camera[3];                 // camera centre (specular term)
    float ambient, diffuse, specular, shininess;   // light colour x material colour per term (grey), Materials.shininess
    int B, T, F, V, C, H, W;
    int cov_B;                       // 1: one coverage map shared by the batch, else B
};

__device__ __forceinline__ void uv_vertex_ids(const UvUnwrap& a, int fi, int* ids) {
    const bool back = fi >= a.F;                        // fill_back copy: vertex order reversed (renderer.py:86)
    const int32_t* t = a.tri + (size_t)(back ? fi - a.F : fi) * 3;
    ids[0] = back ? t[2] : t[0]; ids[1] = t[1]; ids[2] = back ? t[0] : t[2];
}

// everything a pixel needs from its face: the texture coordinate, `tint` = ambient + diffuse * relu(n.l) (what
// multiplies the texel) and `spec` = specular * relu(v.r)^shininess [n.l > 0] (what is added): pytorch3d's
// phong_shading / _apply_lighting with PointLights.diffuse / .specular (vectors normalised with F.normalize's eps 1e-6)
__device__ __forceinline__ void uv_pixel_setup(const UvUnwrap& a, int b, const int* ids, const float* w, float* uv, float& tint,
                                               float& spec) {
    float n[3] = {0, 0, 0}, p[3] = {0, 0, 0};
    uv[0] = uv[1] = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float* vn = a.vnormals + (size_t)ids[k] * 3;
        const float* vp = a.verts + (size_t)ids[k] * 3;
        const float* vt = a.uvs + ((size_t)b * a.V + ids[k]) * 2;
#pragma unroll
        for (int c = 0; c < 3; c++) { n[c] += w[k] * vn[c]; p[c] += w[k] * vp[c]; }
        uv[0] += w[k] * vt[0];
        uv[1] += w[k] * vt[1];
    }
    float d[3] = {a.light[0] - p[0], a.light[1] - p[1], a.light[2] - p[2]};
    float v[3] = {a.camera[0] - p[0], a.camera[1] - p[1], a.camera[2] - p[2]};
    const float nl = fmaxf(sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), 1e-6f);      // F.normalize(eps=1e-6)
    const float dl = fmaxf(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), 1e-6f);
    const float vl = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-6f);
#pragma unroll
    for (int c = 0; c < 3; c++) { n[c] /= nl; d[c] /= dl; v[c] /= vl; }
    const float cs = n[0] * d[0] + n[1] * d[1] + n[2] * d[2];
    tint = a.ambient + a.diffuse * fmaxf(cs, 0.0f);
    float vr = 0.0f;                                            // view . reflect,  reflect = -d + 2 (n.d) n
#pragma unroll
    for (int c = 0; c < 3; c++) vr += v[c] * (-d[c] + 2.0f * (cs * n[c]));
    const float al = cs > 0.0f ? fmaxf(vr, 0.0f) : 0.0f;
    spec = a.specular * powf(al, a.shininess);
}

// bilinear taps of grid_sample(align_corners=True, padding_mode="border") on the vertically flipped map
struct UvTaps {
    int x0, x1, y0, y1;
    float fx, fy;
    bool in_x, in_y;        // the coordinate was not clamped (its derivative is not zero)
};
__device__ __forceinline__ UvTaps uv_taps(const UvUnwrap& a, const float* uv) {
    UvTaps t;
    float x = uv[0] * (float)(a.W - 1), y = (1.0f - uv[1]) * (float)(a.H - 1);
    t.in_x = x >= 0.0f && x <= (float)(a.W - 1);
    t.in_y = y >= 0.0f && y <= (float)(a.H - 1);
    x = fminf(fmaxf(x, 0.0f), (float)(a.W - 1));
    y = fminf(fmaxf(y, 0.0f), (float)(a.H - 1));
    t.x0 = min((int)floorf(x), a.W - 1); t.y0 = min((int)floorf(y), a.H - 1);
    t.x1 = min(t.x0 + 1, a.W - 1); t.y1 = min(t.y0 + 1, a.H - 1);
    t.fx = x - (float)t.x0; t.fy = y - (float)t.y0;
    return t;
}

// out_img / out_mask [B,T,T,4] (row 0 = top): rgb = texel * tint + spec / tint + spec, alpha = 1 where a face covers
__global__ void __launch_bounds__(256) k_uv_unwrap(UvUnwrap a, float* __restrict__ out_img, float* __restrict__ out_mask) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)a.B * a.T * a.T) return;
    const int b = (int)(i / ((long)a.T * a.T)), yo = (int)((i / a.T) % a.T), xo = (int)(i % a.T);
    const size_t p = ((size_t)(a.cov_B > 1 ? b : 0) * a.T + (a.T - 1 - yo)) * a.T + xo;      // internal maps are bottom-up
    float4 img = make_float4(0, 0, 0, 0), msk = make_float4(0, 0, 0, 0);        // BlendParams(background_color=(0,0,0))
    const int fi = a.face_index_map[p];
    if (fi >= 0 && a.used[b]) {
        int ids[3];
        uv_vertex_ids(a, fi, ids);
        const float w[3] = {a.weight_map[3 * p], a.weight_map[3 * p + 1], a.weight_map[3 * p + 2]};
        float uv[2], tint, spec;
        uv_pixel_setup(a, b, ids, w, uv, tint, spec);
        const UvTaps t = uv_taps(a, uv);
        float tex[3] = {0, 0, 0};
        for (int c = 0; c < a.C && c < 3; c++) {
            const float* ch = a.imgs + ((size_t)b * a.C + c) * a.H * a.W;
            const float top = ch[t.y0 * a.W + t.x0] * (1.0f - t.fx) + ch[t.y0 * a.W + t.x1] * t.fx;
            const float bot = ch[t.y1 * a.W + t.x0] * (1.0f - t.fx) + ch[t.y1 * a.W + t.x1] * t.fx;
            tex[c] = top * (1.0f - t.fy) + bot * t.fy;
        }
        img = make_float4(tex[0] * tint + spec, tex[1] * tint + spec, tex[2] * tint + spec, 1.0f);
        msk = make_float4(tint + spec, tint + spec, tint + spec, 1.0f);       // the same mesh textured with ones (:94-97)
    }
    reinterpret_cast<float4*>(out_img)[i] = img;
    reinterpret_cast<float4*>(out_mask)[i] = msk;
}

// adjoint wrt the image and the per-vertex image coordinates (the UV layout itself is a constant of the model)
__global__ void __launch_bounds__(256) k_uv_unwrap_backward(UvUnwrap a, const float* __restrict__ g_img,
                                                           float* __restrict__ g_imgs, float* __restrict__ g_uvs) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)a.B * a.T * a.T) return;
    const int b = (int)(i / ((long)a.T * a.T)), yo = (int)((i / a.T) % a.T), xo = (int)(i % a.T);
    const size_t p = ((size_t)(a.cov_B > 1 ? b : 0) * a.T + (a.T - 1 - yo)) * a.T + xo;
    const int fi = a.face_index_map[p];
    if (fi < 0 || !a.used[b]) return;
    int ids[3];
    uv_vertex_ids(a, fi, ids);
    const float w[3] = {a.weight_map[3 * p], a.weight_map[3 * p + 1], a.weight_map[3 * p + 2]};
    float uv[2], tint, spec;
    uv_pixel_setup(a, b, ids, w, uv, tint, spec);
    const UvTaps t = uv_taps(a, uv);
    float gx = 0, gy = 0;
    for (int c = 0; c < a.C && c < 3; c++) {
        const float gt = g_img[4 * i + c] * tint;
        if (gt == 0.0f) continue;
        const size_t base = ((size_t)b * a.C + c) * a.H * a.W;
        const float* ch = a.imgs + base;
        const float v00 = ch[t.y0 * a.W + t.x0], v01 = ch[t.y0 * a.W + t.x1], v10 = ch[t.y1 * a.W + t.x0], v11 = ch[t.y1 * a.W + t.x1];
        if (g_imgs) {
            atomicAdd(&g_imgs[base + t.y0 * a.W + t.x0], gt * (1.0f - t.fx) * (1.0f - t.fy));
            atomicAdd(&g_imgs[base + t.y0 * a.W + t.x1], gt * t.fx * (1.0f - t.fy));
            atomicAdd(&g_imgs[base + t.y1 * a.W + t.x0], gt * (1.0f - t.fx) * t.fy);
            atomicAdd(&g_imgs[base + t.y1 * a.W + t.x1], gt * t.fx * t.fy);
        }
        gx += gt * ((v01 - v00) * (1.0f - t.fy) + (v11 - v10) * t.fy);
        gy += gt * ((v10 - v00) * (1.0f - t.fx) + (v11 - v01) * t.fx);
    }
    if (g_uvs) {
        const float gu = t.in_x ? gx * (float)(a.W - 1) : 0.0f, gv = t.in_y ? -gy * (float)(a.H - 1) : 0.0f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float* g = g_uvs + ((size_t)b * a.V + ids[k]) * 2;
            atomicAdd(&g[0], w[k] * gu);
            atomicAdd(&g[1], w[k] * gv);
        }
    }
}

}  // namespace d3m
