// d3m_textures.h -- texture asset kernels: uv image -> per-face texture cubes, and texture cubes -> atlas image.
// Replaces NR/cuda/load_textures_cuda_kernel.cu:23-114 (LTK) and create_texture_image_cuda_kernel.cu:10-115 (CTK).
// Both are element-wise and HBM-bound (one texel / one atlas pixel per lane, outputs written coalesced).
#pragma once
#include "d3m_launch.h"

enum { D3M_WRAP_REPEAT = 0, D3M_WRAP_MIRRORED_REPEAT = 1, D3M_WRAP_CLAMP_TO_EDGE = 2, D3M_WRAP_CLAMP_TO_BORDER = 3 };

// LTK:6-14
__device__ __forceinline__ float tex_mod(float x, float y) { return x > 0 ? fmodf(x, y) : y + fmodf(x, y); }

// LTK:55-76, applied ONCE to a private copy of the uv coordinate.  The reference writes the wrapped value back
// into the shared `faces` array from every texel thread: a race that is harmless except at integer coordinates
// under REPEAT (0 -> 1 -> 0 ...), where its result depends on thread scheduling.  "Once" is what a thread that
// sees the caller's input computes; `faces` stays read-only here.
__device__ __forceinline__ float wrap_uv(float v, int wrapping) {
    if (wrapping == D3M_WRAP_REPEAT) return tex_mod(v, 1.0f);
    if (wrapping == D3M_WRAP_MIRRORED_REPEAT) return (tex_mod(v, 2.0f) < 1) ? tex_mod(v, 1.0f) : 1 - tex_mod(v, 1.0f);
    if (wrapping == D3M_WRAP_CLAMP_TO_EDGE) return fmaxf(fminf(v, 1.0f), 0.0f);
    return v;
}

// One lane per texel of textures [F, ts, ts, ts, 3]; faces with is_update == 0 are left untouched.
__global__ void __launch_bounds__(256) k_load_textures(const float* __restrict__ image,
                                                       const int32_t* __restrict__ is_update,
                                                       const float* __restrict__ faces, float* __restrict__ textures,
                                                       long n_texels, int ts, int image_height, int image_width,
                                                       int wrapping, int use_bilinear) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_texels) return;
    const int ts3 = ts * ts * ts;
    const int fn = (int)(i / ts3);
    if (is_update[fn] == 0) return;
    const int r = (int)(i - (long)fn * ts3);
    // barycentric direction of this texel (LTK:42-50); the reference divides in double and stores f32
    float dim0 = (float)((r / (ts * ts)) / (ts - 1.));
    float dim1 = (float)(((r / ts) % ts) / (ts - 1.));
    float dim2 = (float)((r % ts) / (ts - 1.));
    if (0 < dim0 + dim1 + dim2) {
        const float sum = dim0 + dim1 + dim2;
        dim0 /= sum; dim1 /= sum; dim2 /= sum;
    }
    float* tex = textures + i * 3;
    if (wrapping == D3M_WRAP_CLAMP_TO_BORDER) {                 // LTK:97,109: the reference writes zeros
        tex[0] = 0; tex[1] = 0; tex[2] = 0;
        return;
    }
    float uv[6];
#pragma unroll
    for (int k = 0; k < 6; k++) uv[k] = wrap_uv(faces[(long)fn * 6 + k], wrapping);
    const float pos_x = (uv[0] * dim0 + uv[2] * dim1 + uv[4] * dim2) * (float)(image_width - 1);
    const float pos_y = (uv[1] * dim0 + uv[3] * dim1 + uv[5] * dim2) * (float)(image_height - 1);
    if (use_bilinear) {
        const int xi = (int)pos_x, yi = (int)pos_y;
        const float wx1 = pos_x - (float)xi, wx0 = 1 - wx1, wy1 = pos_y - (float)yi, wy0 = 1 - wy1;
        const int y1 = min((int)(pos_y + 1), image_height - 1), x1 = min(xi + 1, image_width - 1);
        const float* p00 = image + ((long)yi * image_width + xi) * 3;
        const float* p10 = image + ((long)y1 * image_width + xi) * 3;
        const float* p01 = image + ((long)yi * image_width + x1) * 3;
        const float* p11 = image + ((long)y1 * image_width + x1) * 3;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float c = 0;
            c += p00[k] * (wx0 * wy0);
            c += p10[k] * (wx0 * wy1);
            c += p01[k] * (wx1 * wy0);
            c += p11[k] * (wx1 * wy1);
            tex[k] = c;
        }
    } else {
        const int xi = (int)roundf(pos_x), yi = (int)roundf(pos_y);
        const float* p = image + ((long)yi * image_width + xi) * 3;
        tex[0] = p[0]; tex[1] = p[1]; tex[2] = p[2];
    }
}

// One lane per atlas pixel.  The reference's second launch (CTK:97-115) copies the finished pixel (x-1, y) onto
// the pixels just right of each tile's diagonal; (x-1, y) is never such a pixel itself and lies in the same tile,
// so evaluating this lane at x-1 gives the identical value in one pass.  Padding tiles (fn >= num_faces, where the
// reference reads out of bounds) keep zeros.
__global__ void __launch_bounds__(256) k_create_texture_image(const float* __restrict__ vertices_all,
                                                              const float* __restrict__ textures,
                                                              float* __restrict__ image, long n_pixels, int num_faces,
                                                              int tsi, int tso, int tile_width, float eps) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pixels) return;
    const int width = tile_width * tso;
    int x = (int)(i % width);
    const int y = (int)(i / width);
    if ((y % tso + 1) == (x % tso)) x -= 1;
    const int fn = x / tso + (y / tso) * tile_width;
    float* out = image + i * 3;
    if (fn >= num_faces) { out[0] = 0; out[1] = 0; out[2] = 0; return; }
    const float* texture = textures + (long)fn * tsi * tsi * tsi * 3;
    const float* p0 = vertices_all + (long)fn * 6;
    const float* p1 = p0 + 2;
    const float* p2 = p0 + 4;
    float face_inv[9] = {
        p1[1] - p2[1], p2[0] - p1[0], p1[0] * p2[1] - p2[0] * p1[1],
        p2[1] - p0[1], p0[0] - p2[0], p2[0] * p0[1] - p0[0] * p2[1],
        p0[1] - p1[1], p1[0] - p0[0], p0[0] * p1[1] - p1[0] * p0[1]};
    const float den = p2[0] * (p0[1] - p1[1]) + p0[0] * (p1[1] - p2[1]) + p1[0] * (p2[1] - p0[1]);
#pragma unroll
    for (int k = 0; k < 9; k++) face_inv[k] /= den;
    float weight[3], weight_sum = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        weight[k] = face_inv[3 * k + 0] * (float)x + face_inv[3 * k + 1] * (float)y + face_inv[3 * k + 2];
        weight_sum += weight[k];
    }
    float tif[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        weight[k] /= (weight_sum + eps);
        float t = weight[k] * (float)(tsi - 1);
        t = fmaxf(t, 0.0f);
        t = fminf(t, (float)(tsi - 1) - eps);
        tif[k] = t;
    }
    float px[3] = {0, 0, 0};
#pragma unroll
    for (int pn = 0; pn < 8; pn++) {
        float w = 1;
        int tii[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int fl = (int)tif[k];
            if (((pn >> k) & 1) == 0) { w *= 1 - (tif[k] - (float)fl); tii[k] = fl; }
            else                      { w *= tif[k] - (float)fl;       tii[k] = fl + 1; }
        }
        // tsi == 1 makes the reference index one cube past this face (weight -eps); stay inside the array
        const int isc = tii[0] * tsi * tsi + tii[1] * tsi + tii[2];
        const bool in_range = (long)fn * tsi * tsi * tsi + isc < (long)num_faces * tsi * tsi * tsi;
#pragma unroll
        for (int k = 0; k < 3; k++) px[k] += w * (in_range ? texture[isc * 3 + k] : 0.0f);
    }
    out[0] = px[0]; out[1] = px[1]; out[2] = px[2];
}
