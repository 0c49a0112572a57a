// d3m_aux.h -- the steps either side of the rasterizer kernels that the reference runs as chains of
// eager torch ops: camera transforms (+ adjoint), vertices_to_faces gather / scatter-add, the output
// epilogue of rasterize_rgbad, and the losses.  Each is one pass over its data.
#pragma once
#include "d3m_device.h"
#include "../../include/d3m_raster.h"

namespace d3m {

// Camera block as the kernels see it (a by-value copy of the host struct).
struct Cam {
    int mode, perspective;
    float width, orig;
    const float *rot, *eye_or_t, *K, *dist;
    int rot_b, eye_b, K_b, dist_b;
};

__device__ __forceinline__ const float* cam_ptr(const float* p, int nb, int b, int stride) {
    return p + (size_t)(nb > 1 ? b : 0) * stride;
}

// look_at / look basis: rows (x, y, z) = normalise(cross(up, z)), normalise(cross(z, x)), z with
// z = normalise(at - eye) or normalise(direction); F.normalize semantics v / max(|v|, 1e-5)
// (neural_renderer/look_at.py:48-53, look.py:39-44).  One lane per view.
__device__ __forceinline__ void normalize3(float* v) {
    const float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float d = fmaxf(n, 1e-5f);
    v[0] /= d; v[1] /= d; v[2] /= d;
}
__device__ __forceinline__ void cross3(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
__global__ void k_camera_basis(const float* __restrict__ eye, int eye_b, const float* __restrict__ at_or_dir, int at_b,
                               const float* __restrict__ up, int up_b, int is_look_at, float* __restrict__ rot, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* e = cam_ptr(eye, eye_b, b, 3);
    const float* a = cam_ptr(at_or_dir, at_b, b, 3);
    const float* u = cam_ptr(up, up_b, b, 3);
    float z[3], x[3], y[3];
    for (int k = 0; k < 3; k++) z[k] = is_look_at ? a[k] - e[k] : a[k];
    normalize3(z);
    cross3(u, z, x);
    normalize3(x);
    cross3(z, x, y);
    normalize3(y);
    for (int k = 0; k < 3; k++) { rot[b * 9 + k] = x[k]; rot[b * 9 + 3 + k] = y[k]; rot[b * 9 + 6 + k] = z[k]; }
}

// Intermediate values of the projection chain that both directions need.
struct ProjTmp {
    float cx, cy, cz, zz, x_, y_, r2, radial, dr;
};

__device__ __forceinline__ void camera_point(const Cam& c, int b, const float* v, float* o, ProjTmp* tmp) {
    if (c.mode == D3M_CAMERA_NONE) {
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
        return;
    }
    const float* r = cam_ptr(c.rot, c.rot_b, b, 9);
    const float* e = cam_ptr(c.eye_or_t, c.eye_b, b, 3);
    if (c.mode == D3M_CAMERA_LOOK_AT || c.mode == D3M_CAMERA_LOOK) {
        const float d0 = v[0] - e[0], d1 = v[1] - e[1], d2 = v[2] - e[2];
        float x = d0 * r[0] + d1 * r[1] + d2 * r[2];
        float y = d0 * r[3] + d1 * r[4] + d2 * r[5];
        const float z = d0 * r[6] + d1 * r[7] + d2 * r[8];
        if (c.perspective) {            // perspective.py:18-19: x / z / width
            x = x / z / c.width;
            y = y / z / c.width;
        }
        o[0] = x; o[1] = y; o[2] = z;
        return;
    }
    // projection.py:19-42
    const float* K = cam_ptr(c.K, c.K_b, b, 9);
    const float* dc = cam_ptr(c.dist, c.dist_b, b, 5);
    const float cx = v[0] * r[0] + v[1] * r[1] + v[2] * r[2] + e[0];
    const float cy = v[0] * r[3] + v[1] * r[4] + v[2] * r[5] + e[1];
    const float cz = v[0] * r[6] + v[1] * r[7] + v[2] * r[8] + e[2];
    const float zz = cz + 1e-9f;
    const float x_ = cx / zz, y_ = cy / zz;
    const float k1 = dc[0], k2 = dc[1], p1 = dc[2], p2 = dc[3], k3 = dc[4];
    const float rr = sqrtf(x_ * x_ + y_ * y_);
    const float r2 = rr * rr, r4 = r2 * r2, r6 = r4 * r2;
    const float radial = 1 + k1 * r2 + k2 * r4 + k3 * r6;
    const float x__ = x_ * radial + 2 * p1 * x_ * y_ + p2 * (r2 + 2 * x_ * x_);
    const float y__ = y_ * radial + p1 * (r2 + 2 * y_ * y_) + 2 * p2 * x_ * y_;
    float u = x__ * K[0] + y__ * K[1] + K[2];
    float vv = x__ * K[3] + y__ * K[4] + K[5];
    vv = c.orig - vv;
    o[0] = 2 * (u - c.orig / 2.f) / c.orig;
    o[1] = 2 * (vv - c.orig / 2.f) / c.orig;
    o[2] = cz;
    if (tmp) {
        tmp->cx = cx; tmp->cy = cy; tmp->cz = cz; tmp->zz = zz; tmp->x_ = x_; tmp->y_ = y_;
        tmp->r2 = r2; tmp->radial = radial; tmp->dr = k1 + 2 * k2 * r2 + 3 * k3 * r4;
    }
}

// adjoint of camera_point at v: g (wrt output) -> gv (wrt input vertex)
__device__ __forceinline__ void camera_point_adjoint(const Cam& c, int b, const float* v, const float* g, float* gv) {
    if (c.mode == D3M_CAMERA_NONE) {
        gv[0] = g[0]; gv[1] = g[1]; gv[2] = g[2];
        return;
    }
    const float* r = cam_ptr(c.rot, c.rot_b, b, 9);
    const float* e = cam_ptr(c.eye_or_t, c.eye_b, b, 3);
    float gc[3];
    if (c.mode == D3M_CAMERA_LOOK_AT || c.mode == D3M_CAMERA_LOOK) {
        if (c.perspective) {
            const float d0 = v[0] - e[0], d1 = v[1] - e[1], d2 = v[2] - e[2];
            const float x = d0 * r[0] + d1 * r[1] + d2 * r[2];
            const float y = d0 * r[3] + d1 * r[4] + d2 * r[5];
            const float z = d0 * r[6] + d1 * r[7] + d2 * r[8];
            const float izw = 1.0f / (z * c.width);
            gc[0] = g[0] * izw;
            gc[1] = g[1] * izw;
            gc[2] = g[2] - (g[0] * x + g[1] * y) * izw / z;
        } else {
            gc[0] = g[0]; gc[1] = g[1]; gc[2] = g[2];
        }
    } else {
        ProjTmp t;
        float o[3];
        camera_point(c, b, v, o, &t);
        const float* K = cam_ptr(c.K, c.K_b, b, 9);
        const float* dc = cam_ptr(c.dist, c.dist_b, b, 5);
        const float p1 = dc[2], p2 = dc[3];
        const float gu = g[0] * 2.f / c.orig, gvp = -g[1] * 2.f / c.orig;
        const float gx2 = K[0] * gu + K[3] * gvp, gy2 = K[1] * gu + K[4] * gvp;
        const float x_ = t.x_, y_ = t.y_;
        const float dxx = t.radial + 2 * x_ * x_ * t.dr + 2 * p1 * y_ + 6 * p2 * x_;
        const float dxy = 2 * x_ * y_ * t.dr + 2 * p1 * x_ + 2 * p2 * y_;
        const float dyx = 2 * x_ * y_ * t.dr + 2 * p1 * x_ + 2 * p2 * y_;
        const float dyy = t.radial + 2 * y_ * y_ * t.dr + 6 * p1 * y_ + 2 * p2 * x_;
        const float gx1 = gx2 * dxx + gy2 * dyx, gy1 = gx2 * dxy + gy2 * dyy;
        gc[0] = gx1 / t.zz;
        gc[1] = gy1 / t.zz;
        gc[2] = g[2] - (gx1 * x_ + gy1 * y_) / t.zz;
    }
    gv[0] = r[0] * gc[0] + r[3] * gc[1] + r[6] * gc[2];
    gv[1] = r[1] * gc[0] + r[4] * gc[1] + r[7] * gc[2];
    gv[2] = r[2] * gc[0] + r[5] * gc[1] + r[8] * gc[2];
}

__global__ void __launch_bounds__(256) k_camera_forward(const float* __restrict__ vertices, int vb, Cam c,
                                                       float* __restrict__ out, int B, int V) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * V) return;
    const int b = (int)(i / V), v = (int)(i % V);
    const float* p = vertices + ((size_t)(vb > 1 ? b : 0) * V + v) * 3;
    const float in[3] = {p[0], p[1], p[2]};
    float o[3];
    camera_point(c, b, in, o, nullptr);
    out[i * 3 + 0] = o[0];
    out[i * 3 + 1] = o[1];
    out[i * 3 + 2] = o[2];
}

// Per-view meshes (vb > 1): one lane per (view, vertex).  A shared mesh (vb == 1): EIGHT lanes per vertex, lane s
// summing views s, s+8, ... and the eight partial sums combined in a fixed DPP tree -- the multi-view reduction needs
// no atomics, is deterministic, and runs on 8x as many lanes as one-lane-per-vertex (50 k vertices leave most of the
// chip idle, and each lane's 32 views were a chain of dependent loads).
__global__ void __launch_bounds__(256) k_camera_backward(const float* __restrict__ vertices, int vb, Cam c,
                                                        const float* __restrict__ grad_out,
                                                        float* __restrict__ grad_vertices, int B, int V,
                                                        bool accumulate = false) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool shared = vb <= 1;
    const long i = shared ? t >> 3 : t;
    const int sub = shared ? (int)(t & 7) : 0;
    const long n = (long)(shared ? 1 : B) * V;
    const bool on = i < n;
    const int v = on ? (int)(i % V) : 0;
    const int b_lo = shared ? sub : (int)(i / V), b_hi = shared ? B : b_lo + 1, b_step = shared ? 8 : 1;
    float acc[3] = {0, 0, 0};
    if (on) {
        const float* p = vertices + (size_t)i * 3;
        const float in[3] = {p[0], p[1], p[2]};
        for (int b = b_lo; b < b_hi; b += b_step) {
            const float* gp = grad_out + ((size_t)b * V + v) * 3;
            const float g[3] = {gp[0], gp[1], gp[2]};
            float gv[3];
            camera_point_adjoint(c, b, in, g, gv);
            acc[0] += gv[0]; acc[1] += gv[1]; acc[2] += gv[2];
        }
    }
    if (shared) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            acc[k] += dpp_f32<0xB1>(acc[k]);      // quad_perm [1,0,3,2]
            acc[k] += dpp_f32<0x4E>(acc[k]);      // quad_perm [2,3,0,1]
            acc[k] += dpp_f32<0x141>(acc[k]);     // row_half_mirror: the other quad of the 8
        }
    }
    if (on && sub == 0) {      // (one writer per entry: a plain read-modify-write when accumulating)
        grad_vertices[i * 3 + 0] = accumulate ? grad_vertices[i * 3 + 0] + acc[0] : acc[0];
        grad_vertices[i * 3 + 1] = accumulate ? grad_vertices[i * 3 + 1] + acc[1] : acc[1];
        grad_vertices[i * 3 + 2] = accumulate ? grad_vertices[i * 3 + 2] + acc[2] : acc[2];
    }
}

// vertices_to_faces.py:16-22 + fill_back (renderer.py:86): one lane per output float, coalesced stores
__global__ void __launch_bounds__(256) k_gather_faces(IndexedFaces fs, float* __restrict__ faces_out, int B) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Fp = fs.num_faces();
    if (i >= (long)B * Fp * 9) return;
    const int c = (int)(i % 3), n = (int)((i / 3) % 3);
    const long bf = i / 9;
    const int b = (int)(bf / Fp), f = (int)(bf % Fp);
    int ids[3];
    fs.vertex_ids(b, f, ids);
    faces_out[i] = fs.verts[((size_t)(fs.vert_batch > 1 ? b : 0) * fs.V + ids[n]) * 3 + c];
}

// Backward of the gather: grad_vertices[b, id, c] += grad_faces[b, f, n, c].  One lane per input float
// (coalesced loads); the three components of a vertex sit in adjacent lanes so one atomic
// wave-instruction touches ~21 vertices' 12-byte records.  Zero entries (culled faces) are skipped.
__global__ void __launch_bounds__(256) k_scatter_face_grads(IndexedFaces fs, const float* __restrict__ grad_faces,
                                                           float* __restrict__ grad_vertices, int B) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Fp = fs.num_faces();
    if (i >= (long)B * Fp * 9) return;
    const float g = grad_faces[i];
    if (g == 0.0f) return;
    const int c = (int)(i % 3), n = (int)((i / 3) % 3);
    const long bf = i / 9;
    const int b = (int)(bf / Fp), f = (int)(bf % Fp);
    int ids[3];
    fs.vertex_ids(b, f, ids);
    atomicAdd(&grad_vertices[((size_t)b * fs.V + ids[n]) * 3 + c], g);
}

// ---- the same sums GATHERED per vertex, in a fixed order (deterministic mode, d3m_set_deterministic) ----------------------
// The scatter above (and the vertex targets of the backward operators) add with float atomics: which workgroup's term lands
// first depends on timing, and a float sum depends on its order -- the vertex gradient differs by ~1e-7 of its largest entry
// from run to run.  Here every (view, vertex) is ONE lane that walks the vertex's incident (triangle, corner) pairs in
// ascending order -- a CSR adjacency of the index tensor, built once per topology on the host side -- and adds the entries
// of up to two per-face gradient arrays (K4's and K6's: the reference's own layout [B,F',3,3]) for the front copy, then for
// the fill_back copy (face F + f holds the triangle's vertices reversed: corner 2 - c).  Plain store: every run adds the
// same floats in the same order.
__global__ void __launch_bounds__(256) k_vertex_gather(const float* __restrict__ gf_a, const float* __restrict__ gf_b,
                                                      const int32_t* __restrict__ adj_offsets,
                                                      const int32_t* __restrict__ adj_items, float* __restrict__ grad_vertices,
                                                      int B, int V, int Ft, int fill_back, const int* __restrict__ flags) {
    // flags [B,F'] (optional: a d3m_visibility blob's): a face whose flag is 0 owns no pixel and its entries are zeros --
    // nine tenths of a fill_back mesh's (view, face) pairs, skipped on a 4-byte look-up instead of two 12-byte reads
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * V) return;
    const int b = (int)(i / V), v = (int)(i % V);
    const int Fp = fill_back ? 2 * Ft : Ft;
    const size_t view = (size_t)b * Fp * 9;
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (int e = adj_offsets[v]; e < adj_offsets[v + 1]; e++) {
        const int item = adj_items[e], f = item / 3, c = item - 3 * f;
#pragma unroll
        for (int copy = 0; copy < 2; copy++) {
            if (copy == 1 && !fill_back) break;
            if (flags && flags[(size_t)b * Fp + (copy ? Ft + f : f)] == 0) continue;
            const size_t at = view + (size_t)(copy ? Ft + f : f) * 9 + (size_t)(copy ? 2 - c : c) * 3;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (gf_a) acc[k] += gf_a[at + k];
                if (gf_b) acc[k] += gf_b[at + k];
            }
        }
    }
    float* out = grad_vertices + (size_t)i * 3;
    out[0] = acc[0]; out[1] = acc[1]; out[2] = acc[2];
}

// ---- lighting (neural_renderer/lighting.py:33-56) ---------------------------------------------------------
// light[b,f,:] = ia*ca + id*cd*relu(normal . direction), normal = normalise(cross(v0-v1, v2-v1), eps 1e-5);
// textures[b,f,...,:] *= light.  A block owns 256 consecutive faces: phase 1 one lane per face computes the
// light into LDS, phase 2 all lanes sweep the block's contiguous texel range (coalesced).
struct LightParams {
    float ia, id;
    float ca[3], cd[3], dir[3];
};

__device__ __forceinline__ void face_light(const float* f, const LightParams& lp, float* light, float* nrm, float* len,
                                           float* cosv) {
    const float a[3] = {f[0] - f[3], f[1] - f[4], f[2] - f[5]};     // v10 = v0 - v1
    const float b[3] = {f[6] - f[3], f[7] - f[4], f[8] - f[5]};     // v12 = v2 - v1
    float c[3];
    cross3(a, b, c);
    const float n = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    const float d = fmaxf(n, 1e-5f);
    const float nx = c[0] / d, ny = c[1] / d, nz = c[2] / d;
    const float cs = nx * lp.dir[0] + ny * lp.dir[1] + nz * lp.dir[2];
    const float r = fmaxf(cs, 0.0f);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float l = 0.0f;
        if (lp.ia != 0) l += lp.ia * lp.ca[k];
        if (lp.id != 0) l += lp.id * (lp.cd[k] * r);
        light[k] = l;
    }
    if (nrm) { nrm[0] = nx; nrm[1] = ny; nrm[2] = nz; *len = n; *cosv = cs; }
}

__global__ void __launch_bounds__(256) k_lighting_forward(const float* __restrict__ faces, const float* __restrict__ tex_in,
                                                         float* __restrict__ tex_out, LightParams lp, long n_faces,
                                                         int texels3 /* ts^3*3 */) {
    __shared__ float s_light[256][3];
    const long f0 = (long)blockIdx.x * 256;
    const long f = f0 + threadIdx.x;
    if (f < n_faces) {
        float fc[9], l[3];
#pragma unroll
        for (int k = 0; k < 9; k++) fc[k] = faces[f * 9 + k];
        face_light(fc, lp, l, nullptr, nullptr, nullptr);
        s_light[threadIdx.x][0] = l[0]; s_light[threadIdx.x][1] = l[1]; s_light[threadIdx.x][2] = l[2];
    }
    __syncthreads();
    const long nf_blk = min((long)256, n_faces - f0);
    const long n_el = nf_blk * texels3;
    const long base = f0 * texels3;
    for (long e = threadIdx.x; e < n_el; e += 256) {
        const int lf = (int)(e / texels3), ch = (int)(e % 3);
        tex_out[base + e] = tex_in[base + e] * s_light[lf][ch];
    }
}

// Adjoint: grad_tex_in = g * light;  grad_light[c] = sum_texels g*tex_in  -> chain rule through the normal.
__global__ void __launch_bounds__(256) k_lighting_backward(const float* __restrict__ faces, const float* __restrict__ tex_in,
                                                          const float* __restrict__ g_out, float* __restrict__ g_tex,
                                                          float* __restrict__ g_faces, LightParams lp, long n_faces,
                                                          int texels3) {
    __shared__ float s_light[256][3];
    __shared__ float s_gl[256][3];
    const long f0 = (long)blockIdx.x * 256;
    const long f = f0 + threadIdx.x;
    float fc[9], nrm[3], len = 0, cs = 0;
    if (f < n_faces) {
        float l[3];
#pragma unroll
        for (int k = 0; k < 9; k++) fc[k] = faces[f * 9 + k];
        face_light(fc, lp, l, nrm, &len, &cs);
        s_light[threadIdx.x][0] = l[0]; s_light[threadIdx.x][1] = l[1]; s_light[threadIdx.x][2] = l[2];
    }
    s_gl[threadIdx.x][0] = 0; s_gl[threadIdx.x][1] = 0; s_gl[threadIdx.x][2] = 0;
    __syncthreads();
    const long nf_blk = min((long)256, n_faces - f0);
    const long n_el = nf_blk * texels3;
    const long base = f0 * texels3;
    for (long e = threadIdx.x; e < n_el; e += 256) {
        const int lf = (int)(e / texels3), ch = (int)(e % 3);
        const float g = g_out[base + e];
        if (g_tex) g_tex[base + e] = g * s_light[lf][ch];
        if (g_faces) atomicAdd(&s_gl[lf][ch], g * tex_in[base + e]);
    }
    __syncthreads();
    if (!g_faces || f >= n_faces) return;
    float gf[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (lp.id != 0 && cs > 0) {
        const float g_cos = lp.id * (lp.cd[0] * s_gl[threadIdx.x][0] + lp.cd[1] * s_gl[threadIdx.x][1] +
                                     lp.cd[2] * s_gl[threadIdx.x][2]);
        const float gn[3] = {g_cos * lp.dir[0], g_cos * lp.dir[1], g_cos * lp.dir[2]};
        float gc[3];
        if (len > 1e-5f) {      // n = c/|c| : (I - n n^T)/|c| ; below the clamp n = c/eps
            const float dot = nrm[0] * gn[0] + nrm[1] * gn[1] + nrm[2] * gn[2];
#pragma unroll
            for (int k = 0; k < 3; k++) gc[k] = (gn[k] - nrm[k] * dot) / len;
        } else {
#pragma unroll
            for (int k = 0; k < 3; k++) gc[k] = gn[k] / 1e-5f;
        }
        const float a[3] = {fc[0] - fc[3], fc[1] - fc[4], fc[2] - fc[5]};
        const float b[3] = {fc[6] - fc[3], fc[7] - fc[4], fc[8] - fc[5]};
        float ga[3], gb[3];
        cross3(b, gc, ga);      // c = a x b : dc/da^T g = b x g
        cross3(gc, a, gb);      //             dc/db^T g = g x a
#pragma unroll
        for (int k = 0; k < 3; k++) { gf[k] = ga[k]; gf[6 + k] = gb[k]; gf[3 + k] = -(ga[k] + gb[k]); }
    }
#pragma unroll
    for (int k = 0; k < 9; k++) g_faces[f * 9 + k] = gf[k];
}

__device__ __forceinline__ float block_sum_256(float v, float* s_part) {
    v = wave_sum(v);
    const int wv = threadIdx.x >> 6;
    if (lane_id() == 0) s_part[wv] = v;
    __syncthreads();
    float r = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    __syncthreads();
    return r;
}

// ---- NrRenderer's depth -> warped pixel grid (renderer_nr.py:74-114, 141-158) ----------------------------------
// One pass for every variant of "depth_to_3d_grid, then rigid transforms, then maybe grid_3d_to_2d":
//   P(b, y, x) = depth * Kinv * (x, y, 1)                      (renderer_nr.py:74-80)
//   crop_mesh (renderer_nr.py:145-158, render_yaw): the top / bottom rows take the y and z of the first kept row, the
//                 left / right columns then take the x and z of the first kept column (of the row-adjusted grid), i.e.
//                 P.x from (y, cx), P.y from (ry, x), P.z from (ry, cx) with ry / cx the clamped row / column;
//   Q = A_b (P - c) + c + t_b,  c = (0, 0, center_z)          (rotate_pts + translate_pts; the callers compose inverse
//                                                              warps and chains of them into one (A, t) on [B,3,3])
//   out = Q                                         [B,H*W,3]   or, with K (grid_3d_to_2d, renderer_nr.py:82-88):
//   out = ((K (Q / Q.z)).xy / (W-1, H-1)) * 2 - 1   [B,H,W,2]   the sampling grid F.grid_sample takes.
struct GridWarp {
    const float* depth;
    const float* inv_K; int invK_b;
    const float* rot;                 // [B,3,3]
    const float* trans;               // [B,3]
    float center_z;
    const float* K; int K_b;          // NULL: 3-D output
    int crop_top, crop_bottom, crop_left, crop_right;
    int B, H, W;
};

__device__ __forceinline__ void gw_ray(const float* iK, float x, float y, float* ray) {
#pragma unroll
    for (int k = 0; k < 3; k++) ray[k] = x * iK[3 * k] + y * iK[3 * k + 1] + iK[3 * k + 2];        // renderer_nr.py:79
}

// Q = R (p - c) + c + t about c = (0, 0, center_z) (rotate_pts + translate_pts, renderer_nr.py:64-72); p is shifted in place
__device__ __forceinline__ void gw_rigid(float* p, const float* R, const float* t, float center_z, float* q) {
    p[2] -= center_z;                                                                         // renderer_nr.py:66
    q[0] = p[0] * R[0] + p[1] * R[1] + p[2] * R[2] + t[0];                                    // :67-68, :71
    q[1] = p[0] * R[3] + p[1] * R[4] + p[2] * R[5] + t[1];
    q[2] = p[0] * R[6] + p[1] * R[7] + p[2] * R[8] + center_z + t[2];
}
// grid_3d_to_2d (renderer_nr.py:82-88) of one point: normalised image coordinates of an s_w x s_h image
__device__ __forceinline__ void gw_project(const float* q, const float* K, int W, int H, float* uv) {
    const float xn = q[0] / q[2], yn = q[1] / q[2];                                           // :84
    const float u = xn * K[0] + yn * K[1] + K[2], v = xn * K[3] + yn * K[4] + K[5];           // :85
    uv[0] = u / (float)(W - 1) * 2.0f - 1.0f;                                                 // :86-87
    uv[1] = v / (float)(H - 1) * 2.0f - 1.0f;
}

__global__ void __launch_bounds__(256) k_grid_warp(GridWarp g, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)g.B * g.H * g.W) return;
    const int b = (int)(i / ((long)g.H * g.W)), pix = (int)(i % ((long)g.H * g.W));
    const int xi = pix % g.W, yi = pix / g.W;
    const float* iK = cam_ptr(g.inv_K, g.invK_b, b, 9);
    const float* dview = g.depth + (size_t)b * g.H * g.W;
    float p[3];
    const bool crop = (g.crop_top | g.crop_bottom | g.crop_left | g.crop_right) != 0;
    if (!crop) {
        float ray[3];
        gw_ray(iK, (float)xi, (float)yi, ray);
        const float d = dview[pix];
#pragma unroll
        for (int k = 0; k < 3; k++) p[k] = ray[k] * d;
    } else {
        const int ry = min(max(yi, g.crop_top), g.H - 1 - g.crop_bottom), cx = min(max(xi, g.crop_left), g.W - 1 - g.crop_right);
        float r0[3], r1[3], r2[3];
        gw_ray(iK, (float)cx, (float)yi, r0);      // x component: (y, cx)
        gw_ray(iK, (float)xi, (float)ry, r1);      // y component: (ry, x)
        gw_ray(iK, (float)cx, (float)ry, r2);      // z component: (ry, cx)
        p[0] = r0[0] * dview[yi * g.W + cx];
        p[1] = r1[1] * dview[ry * g.W + xi];
        p[2] = r2[2] * dview[ry * g.W + cx];
    }
    const float* R = g.rot + (size_t)b * 9;
    const float* t = g.trans + (size_t)b * 3;
    float q[3];
    gw_rigid(p, R, t, g.center_z, q);
    if (!g.K) {
        out[3 * i + 0] = q[0]; out[3 * i + 1] = q[1]; out[3 * i + 2] = q[2];
        return;
    }
    float uv[2];
    gw_project(q, cam_ptr(g.K, g.K_b, b, 9), g.W, g.H, uv);
    out[2 * i + 0] = uv[0];
    out[2 * i + 1] = uv[1];
}

// adjoint (no crop): gridDim.y workgroups per batch entry, each a strided share of the pixels; grad_depth per pixel,
// grad_rot [B,3,3] and grad_trans [B,3] by a workgroup reduction (added with atomics into zeroed arrays when a batch
// entry has more than one workgroup: one workgroup per entry left 240 of the 256 CUs idle for a batch of 16).
__global__ void __launch_bounds__(256) k_grid_warp_backward(GridWarp g, const float* __restrict__ g_out,
                                                           float* __restrict__ g_depth, float* __restrict__ g_rot,
                                                           float* __restrict__ g_trans) {
    __shared__ float s_part[4];
    const int b = blockIdx.x, H = g.H, W = g.W;
    const long base = (long)b * H * W;
    const float* iK = cam_ptr(g.inv_K, g.invK_b, b, 9);
    const float* R = g.rot + (size_t)b * 9;
    const float* t = g.trans + (size_t)b * 3;
    const float* K = g.K ? cam_ptr(g.K, g.K_b, b, 9) : nullptr;
    float acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int pix = blockIdx.y * 256 + threadIdx.x; pix < H * W; pix += 256 * gridDim.y) {
        float ray[3], p[3];
        gw_ray(iK, (float)(pix % W), (float)(pix / W), ray);
        const float d = g.depth[base + pix];
#pragma unroll
        for (int k = 0; k < 3; k++) p[k] = ray[k] * d;
        p[2] -= g.center_z;
        float gq[3];
        if (!K) {
            gq[0] = g_out[3 * (base + pix)]; gq[1] = g_out[3 * (base + pix) + 1]; gq[2] = g_out[3 * (base + pix) + 2];
        } else {
            float q[3];
            q[0] = p[0] * R[0] + p[1] * R[1] + p[2] * R[2] + t[0];
            q[1] = p[0] * R[3] + p[1] * R[4] + p[2] * R[5] + t[1];
            q[2] = p[0] * R[6] + p[1] * R[7] + p[2] * R[8] + g.center_z + t[2];
            const float gu = g_out[2 * (base + pix)] * 2.0f / (float)(W - 1), gv = g_out[2 * (base + pix) + 1] * 2.0f / (float)(H - 1);
            const float gxn = gu * K[0] + gv * K[3], gyn = gu * K[1] + gv * K[4];
            const float iz = 1.0f / q[2];
            gq[0] = gxn * iz;
            gq[1] = gyn * iz;
            gq[2] = -(gxn * q[0] + gyn * q[1]) * iz * iz;
        }
        float gp[3];
#pragma unroll
        for (int j = 0; j < 3; j++) gp[j] = gq[0] * R[j] + gq[1] * R[3 + j] + gq[2] * R[6 + j];
        if (g_depth) g_depth[base + pix] = gp[0] * ray[0] + gp[1] * ray[1] + gp[2] * ray[2];
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
            for (int j = 0; j < 3; j++) acc[3 * k + j] += gq[k] * p[j];
            acc[9 + k] += gq[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const float v = block_sum_256(acc[k], s_part);
        if (threadIdx.x == 0) {
            float* dst = k < 9 ? (g_rot ? g_rot + b * 9 + k : nullptr) : (g_trans ? g_trans + b * 3 + (k - 9) : nullptr);
            if (dst) { if (gridDim.y > 1) atomicAdd(dst, v); else *dst = v; }
        }
    }
}

// ---- NrRenderer.get_normal_from_depth (renderer_nr.py:127-139) -----------------------------------------------------
//   tu = P(y, x+1) - P(y, x-1),  tv = P(y+1, x) - P(y-1, x),  n = tu x tv,  normal = n / (|n| + 1e-7)   (interior)
//   normal = (0, 0, 1) / (1 + 1e-7) on the one-pixel border;  P = depth * Kinv (x, y, 1).
__device__ __forceinline__ void dn_point(const float* __restrict__ dview, const float* iK, int W, int x, int y, float* p) {
    float ray[3];
    gw_ray(iK, (float)x, (float)y, ray);
    const float d = dview[y * W + x];
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = ray[k] * d;
}
__device__ __forceinline__ void cross3f(const float* a, const float* b, float* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
constexpr float DN_EPS = 1e-7f;

__global__ void __launch_bounds__(256) k_depth_normals(const float* __restrict__ depth, const float* __restrict__ inv_K,
                                                      int invK_b, float* __restrict__ normal, int B, int H, int W) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * H * W) return;
    const int b = (int)(i / ((long)H * W)), pix = (int)(i % ((long)H * W));
    const int x = pix % W, y = pix / W;
    float n[3] = {0.0f, 0.0f, 1.0f};
    if (x > 0 && x < W - 1 && y > 0 && y < H - 1) {
        const float* iK = cam_ptr(inv_K, invK_b, b, 9);
        const float* dview = depth + (size_t)b * H * W;
        float pr[3], pl[3], pd[3], pu[3], tu[3], tv[3];
        dn_point(dview, iK, W, x + 1, y, pr); dn_point(dview, iK, W, x - 1, y, pl);
        dn_point(dview, iK, W, x, y + 1, pd); dn_point(dview, iK, W, x, y - 1, pu);
#pragma unroll
        for (int k = 0; k < 3; k++) { tu[k] = pr[k] - pl[k]; tv[k] = pd[k] - pu[k]; }
        cross3f(tu, tv, n);
    }
    const float len = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) + DN_EPS;
    normal[3 * i + 0] = n[0] / len; normal[3 * i + 1] = n[1] / len; normal[3 * i + 2] = n[2] / len;
}

// adjoint, gathered: the depth of pixel (y, x) enters the normals of its four neighbours (as their right / left /
// lower / upper point), so each lane recomputes those four normals' tangent gradients and adds up what reaches its own
// point -- no atomics, no scratch.
__device__ __forceinline__ void dn_tangent_grads(const float* __restrict__ dview, const float* iK,
                                                 const float* __restrict__ g_normal_view, int H, int W, int x, int y,
                                                 float* g_tu, float* g_tv) {
    g_tu[0] = g_tu[1] = g_tu[2] = g_tv[0] = g_tv[1] = g_tv[2] = 0.0f;
    if (!(x > 0 && x < W - 1 && y > 0 && y < H - 1)) return;      // border normals are constants
    float pr[3], pl[3], pd[3], pu[3], tu[3], tv[3], n[3];
    dn_point(dview, iK, W, x + 1, y, pr); dn_point(dview, iK, W, x - 1, y, pl);
    dn_point(dview, iK, W, x, y + 1, pd); dn_point(dview, iK, W, x, y - 1, pu);
#pragma unroll
    for (int k = 0; k < 3; k++) { tu[k] = pr[k] - pl[k]; tv[k] = pd[k] - pu[k]; }
    cross3f(tu, tv, n);
    const float norm = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), len = norm + DN_EPS;
    const float* go = g_normal_view + 3 * ((size_t)y * W + x);
    // out = n / len, len = |n| + eps:  g_n = go / len - n * (go . n) / (len^2 |n|)
    const float dot = go[0] * n[0] + go[1] * n[1] + go[2] * n[2];
    const float c = norm > 0.0f ? dot / (len * len * norm) : 0.0f;
    float gn[3];
#pragma unroll
    for (int k = 0; k < 3; k++) gn[k] = go[k] / len - n[k] * c;
    cross3f(tv, gn, g_tu);      // n = tu x tv:  g_tu = tv x g_n,  g_tv = g_n x tu
    cross3f(gn, tu, g_tv);
}

__global__ void __launch_bounds__(256) k_depth_normals_backward(const float* __restrict__ depth, const float* __restrict__ inv_K,
                                                               int invK_b, const float* __restrict__ g_normal,
                                                               float* __restrict__ g_depth, int B, int H, int W) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * H * W) return;
    const int b = (int)(i / ((long)H * W)), pix = (int)(i % ((long)H * W));
    const int x = pix % W, y = pix / W;
    const float* iK = cam_ptr(inv_K, invK_b, b, 9);
    const float* dview = depth + (size_t)b * H * W;
    const float* gview = g_normal + (size_t)b * H * W * 3;
    float gp[3] = {0, 0, 0}, a[3], c[3];
    if (x - 1 >= 0) { dn_tangent_grads(dview, iK, gview, H, W, x - 1, y, a, c); for (int k = 0; k < 3; k++) gp[k] += a[k]; }   // its right point
    if (x + 1 < W)  { dn_tangent_grads(dview, iK, gview, H, W, x + 1, y, a, c); for (int k = 0; k < 3; k++) gp[k] -= a[k]; }   // its left point
    if (y - 1 >= 0) { dn_tangent_grads(dview, iK, gview, H, W, x, y - 1, a, c); for (int k = 0; k < 3; k++) gp[k] += c[k]; }   // its lower point
    if (y + 1 < H)  { dn_tangent_grads(dview, iK, gview, H, W, x, y + 1, a, c); for (int k = 0; k < 3; k++) gp[k] -= c[k]; }   // its upper point
    float ray[3];
    gw_ray(iK, (float)x, (float)y, ray);
    g_depth[i] = gp[0] * ray[0] + gp[1] * ray[1] + gp[2] * ray[2];
}

// ---- get_textures_from_im (deep3dmap/core/renderer/utils.py:81-107) ----------------------------------------------
// Per-face textures of the implicit grid mesh from an image [B,C,H,W]: cell (y, x) holds two faces,
//   n = y (W-1) + x              vertex colours (im[y,x],   im[y,x+1], im[y+1,x])      (utils.py:102)
//   n + (H-1)(W-1)               vertex colours (im[y+1,x], im[y,x+1], im[y+1,x+1])    (utils.py:103)
// tx_size 2: texel idx of the 2x2x2 cube = sum_j CUBE[idx][j] * colour_j (utils.py:84-94) -> [B, 2 cells, 8, C];
// tx_size 1: the colour of im[y,x] / im[y+1,x+1] (utils.py:99-100) -> [B, 2 cells, 1, C].  One lane per (face, texel).
__constant__ float TFI_CUBE[8][3] = {{0.5f, 0.5f, 0.5f}, {0.f, 0.f, 1.f}, {0.f, 1.f, 0.f}, {-0.5f, 0.5f, 0.5f},
                                     {1.f, 0.f, 0.f}, {0.5f, -0.5f, 0.5f}, {0.5f, 0.5f, -0.5f}, {0.f, 0.f, 0.f}};
__device__ __forceinline__ void tfi_vertices(int second, int y, int x, int W, int* v) {     // pixel offsets in the image
    if (!second) { v[0] = y * W + x; v[1] = y * W + x + 1; v[2] = (y + 1) * W + x; }
    else         { v[0] = (y + 1) * W + x; v[1] = y * W + x + 1; v[2] = (y + 1) * W + x + 1; }
}

__global__ void __launch_bounds__(256) k_textures_from_im(const float* __restrict__ im, float* __restrict__ tex, int B, int C,
                                                         int H, int W, int ts) {
    const int cells = (H - 1) * (W - 1), per = ts == 2 ? 8 : 1;
    const long n = (long)B * 2 * cells * per;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int idx = (int)(i % per);
    const long f = i / per;
    const int b = (int)(f / (2 * cells)), fn = (int)(f % (2 * cells));
    const int second = fn >= cells, cell = second ? fn - cells : fn;
    const int y = cell / (W - 1), x = cell % (W - 1);
    const float* img = im + (size_t)b * C * H * W;
    float* o = tex + (size_t)i * C;
    if (ts == 2) {
        int v[3];
        tfi_vertices(second, y, x, W, v);
        for (int c = 0; c < C; c++) {
            const float* ch = img + (size_t)c * H * W;
            // the three products in the order of the reference's matmul row (coeffs . colours)
            o[c] = (TFI_CUBE[idx][0] * ch[v[0]] + TFI_CUBE[idx][1] * ch[v[1]]) + TFI_CUBE[idx][2] * ch[v[2]];
        }
    } else {
        const int px = second ? (y + 1) * W + x + 1 : y * W + x;
        for (int c = 0; c < C; c++) o[c] = img[(size_t)c * H * W + px];
    }
}

// adjoint, gathered per image pixel: it is a vertex of at most six faces (fixed pattern)
__global__ void __launch_bounds__(256) k_textures_from_im_backward(const float* __restrict__ g_tex, float* __restrict__ g_im,
                                                                  int B, int C, int H, int W, int ts) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * C * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H), c = (int)((i / ((long)W * H)) % C), b = (int)(i / ((long)W * H * C));
    const int cells = (H - 1) * (W - 1), per = ts == 2 ? 8 : 1;
    const float* gt = g_tex + (size_t)b * 2 * cells * per * C;
    float acc = 0.0f;
    auto add = [&](int second, int cy, int cx, int slot) {     // pixel (y, x) is vertex `slot` of that face of cell (cy, cx)
        if (cy < 0 || cy >= H - 1 || cx < 0 || cx >= W - 1) return;
        const size_t f = (size_t)(second ? cells : 0) + (size_t)cy * (W - 1) + cx;
        if (ts == 2) {
            for (int idx = 0; idx < 8; idx++) acc += TFI_CUBE[idx][slot] * gt[(f * 8 + idx) * C + c];
        } else if (slot == (second ? 2 : 0)) {
            acc += gt[f * C + c];
        }
    };
    add(0, y, x, 0); add(0, y, x - 1, 1); add(0, y - 1, x, 2);
    add(1, y - 1, x, 0); add(1, y, x - 1, 1); add(1, y - 1, x - 1, 2);
    g_im[i] = acc;
}

// ---- view vector -> (rotation, translation) (deep3dmap/core/renderer/utils.py:34-71) ----------------------------
// R = Rz(rz) Ry(ry) Rx(rx) with the reference's sign conventions, t = (tx, ty, tz) padded with zeros for 5- and
// 3-component views: one lane per batch entry instead of ~35 eager kernels (6 trig calls, 27 stacks, 2 batched matmuls)
// and as many again in backward.  The two 3x3 products are evaluated in the reference's association, Rz (Ry Rx).
__device__ __forceinline__ void mat3_mul(const float* a, const float* b, float* c) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) c[3 * i + j] = (a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j]) + a[3 * i + 2] * b[6 + j];
}
__device__ __forceinline__ void euler_factors(float cx, float sx, float cy, float sy, float cz, float sz, float* mx,
                                              float* my, float* mz) {
    const float x[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
    const float y[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
    const float z[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
#pragma unroll
    for (int k = 0; k < 9; k++) { mx[k] = x[k]; my[k] = y[k]; mz[k] = z[k]; }
}

__global__ void __launch_bounds__(64) k_view_transform(const float* __restrict__ view, int n_comp, float* __restrict__ rot,
                                                      float* __restrict__ trans, int B) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float* v = view + (size_t)b * n_comp;
    float mx[9], my[9], mz[9], yx[9], r[9];
    euler_factors(cosf(v[0]), sinf(v[0]), cosf(v[1]), sinf(v[1]), cosf(v[2]), sinf(v[2]), mx, my, mz);
    mat3_mul(my, mx, yx);
    mat3_mul(mz, yx, r);
#pragma unroll
    for (int k = 0; k < 9; k++) rot[(size_t)b * 9 + k] = r[k];
#pragma unroll
    for (int k = 0; k < 3; k++) trans[(size_t)b * 3 + k] = 3 + k < n_comp ? v[3 + k] : 0.0f;
}

// adjoint: g_view[b, 0..2] = <g_rot, dR/d(rx, ry, rz)>, g_view[b, 3..] = g_trans
__global__ void __launch_bounds__(64) k_view_transform_backward(const float* __restrict__ view, int n_comp,
                                                               const float* __restrict__ g_rot,
                                                               const float* __restrict__ g_trans, float* __restrict__ g_view,
                                                               int B) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float* v = view + (size_t)b * n_comp;
    const float cx = cosf(v[0]), sx = sinf(v[0]), cy = cosf(v[1]), sy = sinf(v[1]), cz = cosf(v[2]), sz = sinf(v[2]);
    float mx[9], my[9], mz[9];
    euler_factors(cx, sx, cy, sy, cz, sz, mx, my, mz);
    // derivatives of the factors: d/dtheta of (cos, sin) = (-sin, cos), constants -> 0
    const float dx[9] = {0, 0, 0, 0, -sx, -cx, 0, cx, -sx};
    const float dy[9] = {-sy, 0, cy, 0, 0, 0, -cy, 0, -sy};
    const float dz[9] = {-sz, -cz, 0, cz, -sz, 0, 0, 0, 0};
    float g[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (g_rot) {
#pragma unroll
        for (int k = 0; k < 9; k++) g[k] = g_rot[(size_t)b * 9 + k];
    }
    float t0[9], t1[9];
    float out[3];
    mat3_mul(my, dx, t0); mat3_mul(mz, t0, t1);          // dR/drx = Rz Ry Rx'
    out[0] = 0; for (int k = 0; k < 9; k++) out[0] += g[k] * t1[k];
    mat3_mul(dy, mx, t0); mat3_mul(mz, t0, t1);          // dR/dry = Rz Ry' Rx
    out[1] = 0; for (int k = 0; k < 9; k++) out[1] += g[k] * t1[k];
    mat3_mul(my, mx, t0); mat3_mul(dz, t0, t1);          // dR/drz = Rz' Ry Rx
    out[2] = 0; for (int k = 0; k < 9; k++) out[2] += g[k] * t1[k];
    float* o = g_view + (size_t)b * n_comp;
    o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
    for (int k = 3; k < n_comp; k++) o[k] = g_trans ? g_trans[(size_t)b * 3 + (k - 3)] : 0.0f;
}

// ---- output epilogue (rasterize.py:181-195, 305-326) -------------------------------------------------
// One lane per OUTPUT pixel.  Internal row r is output row S-1-r (vertical flip); with anti-aliasing
// the output pixel is the mean of its 2x2 internal pixels.
__global__ void __launch_bounds__(256) k_output_epilogue(const int32_t* __restrict__ face_index_map,
                                                        const float* __restrict__ rgb_map,
                                                        const float* __restrict__ depth_map,
                                                        const float* __restrict__ background, int bg_b,
                                                        float* __restrict__ rgb_blended, float* __restrict__ alpha_map,
                                                        float* __restrict__ rgb_out, float* __restrict__ alpha_out,
                                                        float* __restrict__ depth_out, int B, int S, int aa) {
    const int s = aa ? S / 2 : S;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * s * s) return;
    const int b = (int)(i / ((long)s * s));
    const int yo = (int)((i / s) % s), xo = (int)(i % s);
    const int n = aa ? 2 : 1;
    float acc_rgb[3] = {0, 0, 0}, acc_a = 0, acc_d = 0;
    const float* bg = background ? cam_ptr(background, bg_b, b, 3) : nullptr;
    for (int dy = 0; dy < n; dy++) {
        for (int dx = 0; dx < n; dx++) {
            const int yi = S - 1 - (yo * n + dy), xi = xo * n + dx;
            const size_t p = ((size_t)b * S + yi) * S + xi;
            const float mask = face_index_map[p] >= 0 ? 1.0f : 0.0f;
            if (rgb_map) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float v = rgb_map[3 * p + k] * mask + (1 - mask) * bg[k];     // rasterize.py:192
                    if (rgb_blended) rgb_blended[3 * p + k] = v;
                    acc_rgb[k] += v;
                }
            }
            if (alpha_map) alpha_map[p] = mask;
            acc_a += mask;
            if (depth_map) acc_d += depth_map[p];
        }
    }
    const float inv = aa ? 0.25f : 1.0f;
    if (rgb_out) {
#pragma unroll
        for (int k = 0; k < 3; k++) rgb_out[(((size_t)b * 3 + k) * s + yo) * s + xo] = acc_rgb[k] * inv;
    }
    if (alpha_out) alpha_out[i] = acc_a * inv;
    if (depth_out) depth_out[i] = acc_d * inv;
}

// Adjoint: one lane per INTERNAL pixel.  No coverage mask is applied: the reference blends the
// background inside RasterizeFunction.forward, so its backward kernels see the gradient wrt the
// blended image at every pixel (rasterize.py:83, 141-147).
__global__ void __launch_bounds__(256) k_output_epilogue_backward(const float* __restrict__ g_rgb_out,
                                                                 const float* __restrict__ g_alpha_out,
                                                                 const float* __restrict__ g_depth_out,
                                                                 float* __restrict__ g_rgb_map, float* __restrict__ g_alpha_map,
                                                                 float* __restrict__ g_depth_map, int B, int S, int aa) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long)B * S * S) return;
    const int s = aa ? S / 2 : S;
    const int b = (int)(p / ((long)S * S));
    const int yi = (int)((p / S) % S), xi = (int)(p % S);
    const int yo = aa ? (S - 1 - yi) / 2 : S - 1 - yi, xo = aa ? xi / 2 : xi;
    const float sc = aa ? 0.25f : 1.0f;
    const size_t o = ((size_t)b * s + yo) * s + xo;
    if (g_rgb_map) {
#pragma unroll
        for (int k = 0; k < 3; k++) g_rgb_map[3 * p + k] = g_rgb_out[(((size_t)b * 3 + k) * s + yo) * s + xo] * sc;
    }
    if (g_alpha_map) g_alpha_map[p] = g_alpha_out[o] * sc;
    if (g_depth_map) g_depth_map[p] = g_depth_out[o] * sc;
}

// ---- losses ------------------------------------------------------------------------------------------

// photometric_loss (deep3dmap/core/utils/utils.py:105-114): pass 1 accumulates
// per workgroup: scratch[2k] = sum(|a-b| [*sqrt2/(sigma+EPS) + log(sigma+EPS)] * mask), scratch[2k+1] = sum(mask).
// grid.y runs over the B*C image planes, so the per-element index math has no division.
__global__ void __launch_bounds__(256) k_photometric_reduce(const float* __restrict__ a, const float* __restrict__ b,
                                                           const float* __restrict__ mask,
                                                           const float* __restrict__ sigma, float* __restrict__ scratch,
                                                           int C, int hw) {
    __shared__ float s_part[4];
    const int plane = blockIdx.y, bn = plane / C;
    const float* pa = a + (size_t)plane * hw;
    const float* pb = b + (size_t)plane * hw;
    const float* pm = mask ? mask + (size_t)bn * hw : nullptr;
    const float* ps = sigma ? sigma + (size_t)bn * hw : nullptr;
    float num = 0, den = 0;
    auto term = [&](float va, float vb, float vm, float vs) {
        float l = fabsf(va - vb);
        if (ps) {
            const float sg = vs + 1e-7f;
            l = l * 1.41421356237309515f / sg + logf(sg);
        }
        num += l * vm;
        den += vm;
    };
    // planes whose size and base addresses allow it are streamed 16 bytes per lane
    const bool vec = (hw & 3) == 0 && (((uintptr_t)pa | (uintptr_t)pb | (uintptr_t)pm | (uintptr_t)ps) & 15) == 0;
    if (vec) {
        const float4 one = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < (hw >> 2); i += gridDim.x * 256) {
            const float4 va = ((const float4*)pa)[i], vb = ((const float4*)pb)[i];
            const float4 vm = pm ? ((const float4*)pm)[i] : one, vs = ps ? ((const float4*)ps)[i] : one;
            term(va.x, vb.x, vm.x, vs.x);
            term(va.y, vb.y, vm.y, vs.y);
            term(va.z, vb.z, vm.z, vs.z);
            term(va.w, vb.w, vm.w, vs.w);
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256)
            term(pa[i], pb[i], pm ? pm[i] : 1.0f, ps ? ps[i] : 1.0f);
    }
    num = block_sum_256(num, s_part);
    den = block_sum_256(den, s_part);
    if (threadIdx.x == 0) {      // one partial pair per workgroup: deterministic, no same-address atomics
        const int blk = blockIdx.y * gridDim.x + blockIdx.x;
        scratch[2 * blk + 0] = num;
        scratch[2 * blk + 1] = den;
    }
}
__global__ void __launch_bounds__(256) k_photometric_finish(const float* __restrict__ a, const float* __restrict__ b,
                                                           const float* __restrict__ mask,
                                                           const float* __restrict__ sigma,
                                                           const float* __restrict__ scratch, int n_partials,
                                                           float* __restrict__ loss, float* __restrict__ grad_a, int C,
                                                           int hw) {
    __shared__ float s_part[4];
    float num = 0, den = 0;      // every workgroup re-sums the (<= 1024) partial pairs in the same order
    for (int i = threadIdx.x; i < n_partials; i += 256) { num += scratch[2 * i]; den += scratch[2 * i + 1]; }
    num = block_sum_256(num, s_part);
    den = block_sum_256(den, s_part);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *loss = num / den;
    if (!grad_a) return;
    const int plane = blockIdx.y, bn = plane / C;
    const float* pa = a + (size_t)plane * hw;
    const float* pb = b + (size_t)plane * hw;
    const float* pm = mask ? mask + (size_t)bn * hw : nullptr;
    const float* ps = sigma ? sigma + (size_t)bn * hw : nullptr;
    float* pg = grad_a + (size_t)plane * hw;
    auto grad = [&](float va, float vb, float vm, float vs) {
        const float d = va - vb;
        float g = d > 0 ? 1.0f : (d < 0 ? -1.0f : 0.0f);
        if (ps) g = g * 1.41421356237309515f / (vs + 1e-7f);
        return g * vm / den;
    };
    const bool vec = (hw & 3) == 0 &&
                     (((uintptr_t)pa | (uintptr_t)pb | (uintptr_t)pm | (uintptr_t)ps | (uintptr_t)pg) & 15) == 0;
    if (vec) {
        const float4 one = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < (hw >> 2); i += gridDim.x * 256) {
            const float4 va = ((const float4*)pa)[i], vb = ((const float4*)pb)[i];
            const float4 vm = pm ? ((const float4*)pm)[i] : one, vs = ps ? ((const float4*)ps)[i] : one;
            ((float4*)pg)[i] = make_float4(grad(va.x, vb.x, vm.x, vs.x), grad(va.y, vb.y, vm.y, vs.y),
                                           grad(va.z, vb.z, vm.z, vs.z), grad(va.w, vb.w, vm.w, vs.w));
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256)
            pg[i] = grad(pa[i], pb[i], pm ? pm[i] : 1.0f, ps ? ps[i] : 1.0f);
    }
}

// ---- smooth_loss of one map (deep3dmap/core/utils/utils.py:82-102, one pyramid level) --------------------------
//   mean|dxx| + mean|dxy| + mean|dyx| + mean|dyy|   with the reference's nested first differences
// (dx = p[..,1:] - p[..,:-1], dy = p[:,1:] - p[:,:-1], then the same again), i.e. the same f32 subtractions in the
// same order.  The eager form is ~25 slice/abs/mean kernels forward and ~60 backward (every slice's backward is a
// zero fill plus a copy); here: one reduction, one finish, one gradient kernel.
__device__ __forceinline__ float sm_dxx(const float* p, int x) { return (p[x + 2] - p[x + 1]) - (p[x + 1] - p[x]); }
__device__ __forceinline__ float sm_dxy(const float* r0, const float* r1, int x) {      // gradient(dx)[1]
    return (r1[x + 1] - r1[x]) - (r0[x + 1] - r0[x]);
}
__device__ __forceinline__ float sm_dyx(const float* r0, const float* r1, int x) {      // gradient(dy)[0]
    return (r1[x + 1] - r0[x + 1]) - (r1[x] - r0[x]);
}
__device__ __forceinline__ float sm_dyy(const float* r0, const float* r1, const float* r2, int x) {
    return (r2[x] - r1[x]) - (r1[x] - r0[x]);
}
__device__ __forceinline__ float sm_sign(float v) { return v > 0 ? 1.0f : (v < 0 ? -1.0f : 0.0f); }

// partials: 4 floats per workgroup = sum|dxx|, sum|dxy|, sum|dyx|, sum|dyy|
__global__ void __launch_bounds__(256) k_smooth_reduce(const float* __restrict__ pred, float* __restrict__ partials, int B,
                                                      int H, int W) {
    __shared__ float s_part[4];
    float acc[4] = {0, 0, 0, 0};
    const long n = (long)B * H * W;
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
#pragma unroll
        for (int k = 0; k < 4; k++) partials[4 * blockIdx.x + k] = acc[k];
    }
}

__global__ void __launch_bounds__(256) k_smooth_finish(const float* __restrict__ partials, int n, float n_xx, float n_xy,
                                                      float n_yy, float* __restrict__ loss) {
    __shared__ float s_part[4];
    float acc[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < n; i += 256) {
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] += partials[4 * i + k];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) acc[k] = block_sum_256(acc[k], s_part);
    if (threadIdx.x == 0) *loss = ((acc[0] / n_xx + acc[1] / n_xy) + acc[2] / n_xy) + acc[3] / n_yy;
}

// d(sum of the four mean-abs second differences) / d pred[y, x]: the (up to 14) differences that contain the pixel,
// sign(difference) * coefficient / count; r0 = row y of the map
__device__ __forceinline__ float sm_grad_at(const float* __restrict__ r0, int x, int y, int H, int W, float n_xx, float n_xy,
                                            float n_yy) {
    float gxx = 0, gxy = 0, gyy = 0;
    // dxx terms starting at x, x-1, x-2 (coefficients +1, -2, +1)
    if (x + 2 < W) gxx += sm_sign(sm_dxx(r0, x));
    if (x >= 1 && x + 1 < W) gxx -= 2.0f * sm_sign(sm_dxx(r0, x - 1));
    if (x >= 2) gxx += sm_sign(sm_dxx(r0, x - 2));
    // dyy terms starting at y, y-1, y-2
    if (y + 2 < H) gyy += sm_sign(sm_dyy(r0, r0 + W, r0 + 2 * W, x));
    if (y >= 1 && y + 1 < H) gyy -= 2.0f * sm_sign(sm_dyy(r0 - W, r0, r0 + W, x));
    if (y >= 2) gyy += sm_sign(sm_dyy(r0 - 2 * W, r0 - W, r0, x));
    // the mixed terms with corner (y0, x0): +p11 - p10 - p01 + p00
    auto mixed = [&](const float* a0, const float* a1, int x0) { return sm_sign(sm_dxy(a0, a1, x0)) + sm_sign(sm_dyx(a0, a1, x0)); };
    if (y + 1 < H && x + 1 < W) gxy += mixed(r0, r0 + W, x);                 // pixel is p00
    if (y + 1 < H && x >= 1) gxy -= mixed(r0, r0 + W, x - 1);                  // p01
    if (y >= 1 && x + 1 < W) gxy -= mixed(r0 - W, r0, x);                      // p10
    if (y >= 1 && x >= 1) gxy += mixed(r0 - W, r0, x - 1);                     // p11
    return gxx / n_xx + gxy / n_xy + gyy / n_yy;
}

// grad_pred[b,y,x] = grad_loss * sum over the (up to 14) second differences that contain the pixel of
// sign(difference) * coefficient / count -- gathered, one lane per pixel
__global__ void __launch_bounds__(256) k_smooth_grad(const float* __restrict__ pred, const float* __restrict__ grad_loss,
                                                    float* __restrict__ grad_pred, int B, int H, int W, float n_xx,
                                                    float n_xy, float n_yy) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    grad_pred[i] = (*grad_loss) * sm_grad_at(pred + (i - x), x, y, H, W, n_xx, n_xy, n_yy);
}

// ---- the multi-view fit objective in three launches ------------------------------------------------------------
//   loss = photometric(rgb, rgb_t, mask) + sum((alpha - alpha_t)^2) / pixels + photometric(depth, depth_t, mask)
// (photometric_loss = utils.py:105-114 without sigma; the silhouette term = examples/example2.py:46), i.e. what
// core/losses.py composes from d3m_photometric_loss x2 + d3m_sum_squared_error + eager adds: 11 launches and two
// extra passes over the gradients.  Here: one reduction over all inputs, a one-workgroup finish, and (in the
// backward pass) one kernel that writes the three gradients already scaled by the incoming gradient.
// partials: 4 floats per workgroup = sum|rgb-rgb_t|*m, sum|d-d_t|*m, sum m, sum (a-a_t)^2.
struct FitLossArgs {
    const float *rgb, *rgb_t, *depth, *depth_t, *alpha, *alpha_t, *mask;
    int B, hw;
    float pixels;
};

__global__ void __launch_bounds__(256) k_fit_loss_reduce(FitLossArgs a, float* __restrict__ partials) {
    __shared__ float s_part[4];
    const int b = blockIdx.y, hw = a.hw;
    const float* pm = a.mask + (size_t)b * hw;
    float n_rgb = 0, n_d = 0, den = 0, sse = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
        const float m = pm[i];
        const size_t p = (size_t)b * hw + i;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const size_t q = ((size_t)b * 3 + c) * hw + i;
            n_rgb += fabsf(a.rgb[q] - a.rgb_t[q]) * m;
        }
        n_d += fabsf(a.depth[p] - a.depth_t[p]) * m;
        den += m;
        const float d = a.alpha[p] - a.alpha_t[p];
        sse += d * d;
    }
    n_rgb = block_sum_256(n_rgb, s_part);
    n_d = block_sum_256(n_d, s_part);
    den = block_sum_256(den, s_part);
    sse = block_sum_256(sse, s_part);
    if (threadIdx.x == 0) {
        float* o = partials + 4 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        o[0] = n_rgb; o[1] = n_d; o[2] = den; o[3] = sse;
    }
}

// totals[0..3] = the four sums; totals[4] = loss
__global__ void __launch_bounds__(256) k_fit_loss_finish(const float* __restrict__ partials, int n, float pixels,
                                                        const float* __restrict__ mask_sum,
                                                        float* __restrict__ totals, float* __restrict__ loss) {
    __shared__ float s_part[4];
    float acc[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < n; i += 256) {
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] += partials[4 * i + k];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) acc[k] = block_sum_256(acc[k], s_part);
    if (threadIdx.x == 0) {
        if (mask_sum) acc[2] = *mask_sum;    // this batch is a shard: normalise by the mask of the whole objective
#pragma unroll
        for (int k = 0; k < 4; k++) totals[k] = acc[k];
        const float l = (acc[0] / (3.0f * acc[2]) + acc[3] / pixels) + acc[1] / acc[2];
        totals[4] = l;
        *loss = l;
    }
}

__global__ void __launch_bounds__(256) k_fit_loss_grad(FitLossArgs a, const float* __restrict__ totals,
                                                      const float* __restrict__ grad_out, float* __restrict__ g_rgb,
                                                      float* __restrict__ g_depth, float* __restrict__ g_alpha) {
    const int b = blockIdx.y, hw = a.hw;
    const float go = grad_out ? *grad_out : 1.0f;
    const float den = totals[2];
    const float* pm = a.mask + (size_t)b * hw;
    auto sgn = [](float d) { return d > 0 ? 1.0f : (d < 0 ? -1.0f : 0.0f); };
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
        const float m = pm[i];
        const size_t p = (size_t)b * hw + i;
        if (g_rgb) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const size_t q = ((size_t)b * 3 + c) * hw + i;
                g_rgb[q] = sgn(a.rgb[q] - a.rgb_t[q]) * m / (3.0f * den) * go;
            }
        }
        if (g_depth) g_depth[p] = sgn(a.depth[p] - a.depth_t[p]) * m / den * go;
        if (g_alpha) g_alpha[p] = 2.0f * (a.alpha[p] - a.alpha_t[p]) / a.pixels * go;
    }
}

// sum((a-b)^2): per-workgroup partials, then one small workgroup adds them (no same-address atomics)
__global__ void __launch_bounds__(256) k_sum_squared_error(const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ partials, float* __restrict__ grad_a, long n) {
    __shared__ float s_part[4];
    float acc = 0;
    const bool vec = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)grad_a) & 15) == 0;
    const long n4 = vec ? n >> 2 : 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 va = ((const float4*)a)[i], vb = ((const float4*)b)[i];
        const float4 d = make_float4(va.x - vb.x, va.y - vb.y, va.z - vb.z, va.w - vb.w);
        acc += d.x * d.x;
        acc += d.y * d.y;
        acc += d.z * d.z;
        acc += d.w * d.w;
        if (grad_a) ((float4*)grad_a)[i] = make_float4(2.0f * d.x, 2.0f * d.y, 2.0f * d.z, 2.0f * d.w);
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float d = a[i] - b[i];
        acc += d * d;
        if (grad_a) grad_a[i] = 2.0f * d;
    }
    acc = block_sum_256(acc, s_part);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_sum_partials(const float* __restrict__ partials, int n, float* __restrict__ out) {
    __shared__ float s_part[4];
    float acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partials[i];
    acc = block_sum_256(acc, s_part);
    if (threadIdx.x == 0) *out = acc;
}

}  // namespace d3m
