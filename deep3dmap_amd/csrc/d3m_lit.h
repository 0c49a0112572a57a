// d3m_lit.h -- texture sampling and its backward with fill_back and lighting applied ON THE FLY.
//
// The reference materialises, per view, cat(textures, textures.permute(0,1,4,3,2,5)) (renderer.py:156,204)
// and multiplies it in place by the per-face light (lighting.py:55-56) before sampling: 2 x B copies of the
// texture array per step.  Here the sampler reads the ORIGINAL textures [Bx,F,ts,ts,ts,3] (Bx = 1: one mesh
// shared by all views), maps a back face F+f to face f with the first and third cube axes swapped, and
// multiplies each fetched texel by light[face] -- the same f32 product the reference stores, so the sampled
// colours are bit-identical.  The backward gathers per visible face (no atomics for ts == 2).
#pragma once
#include "d3m_aux.h"
#include "d3m_tail.h"
#include "d3m_face_major.h"

namespace d3m {

struct LitTextures {
    const float* textures;   // [Bx, F, ts^3, 3]
    const float* light;      // [Bm, F', 3]
    int F, Fp, ts, tex_batch, light_batch, fill_back;
    // textures == NULL: the 2x2x2 cubes of a grid mesh's faces are evaluated from an image [B,3,im_H,im_W] where they are
    // sampled (get_textures_from_im, CR/utils.py:81-107: the products and sums of k_textures_from_im, so the same bits)
    const float* im;
    int im_H, im_W;
};

// texel `isc` of virtual face f' of view b -> offset into `textures`, or -1 when outside the virtual array
// (ts == 1 only: KCU:229-233 then runs into the following faces, see k_texture_sampling).
__device__ __forceinline__ long lit_texel(const LitTextures& t, int B, int b, int fp, int isc, int* light_row) {
    const int ts3 = t.ts * t.ts * t.ts;
    long vf = (long)b * t.Fp + fp + isc / ts3;       // face reached in the virtual [B,F'] array
    isc %= ts3;
    if (vf >= (long)B * t.Fp) return -1;
    const int bb = (int)(vf / t.Fp), ff = (int)(vf % t.Fp);
    *light_row = (t.light_batch > 1 ? bb : 0) * t.Fp + ff;
    int fo = ff, idx = isc;
    if (ff >= t.F) {                                  // back copy: texel (a,b,c) of it = texel (c,b,a) of face ff - F
        fo = ff - t.F;
        const int a = isc / (t.ts * t.ts), bq = (isc / t.ts) % t.ts, c = isc % t.ts;
        idx = c * t.ts * t.ts + bq * t.ts + a;
    }
    return (((long)(t.tex_batch > 1 ? bb : 0) * t.F + fo) * ts3 + idx) * 3;
}

// per-face light of the (virtual, fill_back) face array on WORLD-space vertices (renderer.py:159-167)
__global__ void __launch_bounds__(256) k_face_light(IndexedFaces fs, LightParams lp, float* __restrict__ light, int Bm) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int Fp = fs.num_faces();
    if (i >= (long)Bm * Fp) return;
    float fc[9], l[3];
    fs.load((int)(i / Fp), (int)(i % Fp), fc);
    face_light(fc, lp, l, nullptr, nullptr, nullptr);
    light[3 * i + 0] = l[0]; light[3 * i + 1] = l[1]; light[3 * i + 2] = l[2];
}

// grad_light [Bm,F',3] -> world-space vertex gradients through the face normal (atomics: a vertex is shared
// by ~6 faces; only faces that received a gradient do anything)
__global__ void __launch_bounds__(256) k_face_light_backward(IndexedFaces fs, LightParams lp, const float* __restrict__ g_light,
                                                            float* __restrict__ grad_vertices, int vertices_batch, int Bm) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int Fp = fs.num_faces();
    if (i >= (long)Bm * Fp || lp.id == 0) return;
    const float gl[3] = {g_light[3 * i], g_light[3 * i + 1], g_light[3 * i + 2]};
    if (gl[0] == 0 && gl[1] == 0 && gl[2] == 0) return;
    const int b = (int)(i / Fp), f = (int)(i % Fp);
    float fc[9], l[3], nrm[3], len, cs;
    fs.load(b, f, fc);
    face_light(fc, lp, l, nrm, &len, &cs);
    if (!(cs > 0)) return;
    const float g_cos = lp.id * (lp.cd[0] * gl[0] + lp.cd[1] * gl[1] + lp.cd[2] * gl[2]);
    const float gn[3] = {g_cos * lp.dir[0], g_cos * lp.dir[1], g_cos * lp.dir[2]};
    float gc[3];
    if (len > 1e-5f) {
        const float dot = nrm[0] * gn[0] + nrm[1] * gn[1] + nrm[2] * gn[2];
        for (int k = 0; k < 3; k++) gc[k] = (gn[k] - nrm[k] * dot) / len;
    } else {
        for (int k = 0; k < 3; k++) gc[k] = gn[k] / 1e-5f;
    }
    const float a[3] = {fc[0] - fc[3], fc[1] - fc[4], fc[2] - fc[5]};
    const float bb[3] = {fc[6] - fc[3], fc[7] - fc[4], fc[8] - fc[5]};
    float ga[3], gb[3];
    cross3(bb, gc, ga);
    cross3(gc, a, gb);
    int ids[3];
    fs.vertex_ids(b, f, ids);
    float* base = grad_vertices + (size_t)(vertices_batch > 1 ? b : 0) * fs.V * 3;
    for (int k = 0; k < 3; k++) {
        atomicAdd(&base[(size_t)ids[0] * 3 + k], ga[k]);
        atomicAdd(&base[(size_t)ids[2] * 3 + k], gb[k]);
        atomicAdd(&base[(size_t)ids[1] * 3 + k], -(ga[k] + gb[k]));
    }
}

// The same adjoint GATHERED per vertex in a fixed order (deterministic mode; see k_vertex_gather): one lane per vertex of
// ONE shared mesh walks the vertex's incident (triangle, corner) pairs and adds what the front copy and the fill_back copy of
// each triangle send to that corner -- the per-face expressions of k_face_light_backward, recomputed per incident face
// (six per vertex on a grid mesh) instead of scattered with float atomics.  grad_vertices [V,3] is WRITTEN.
__device__ __forceinline__ void face_light_corner_grad(const IndexedFaces& fs, const LightParams& lp,
                                                       const float* __restrict__ g_light, int f, int corner, float* acc) {
    const float gl[3] = {g_light[3 * (size_t)f], g_light[3 * (size_t)f + 1], g_light[3 * (size_t)f + 2]};
    if (gl[0] == 0 && gl[1] == 0 && gl[2] == 0) return;
    float fc[9], l[3], nrm[3], len, cs;
    fs.load(0, f, fc);
    face_light(fc, lp, l, nrm, &len, &cs);
    if (!(cs > 0)) return;
    const float g_cos = lp.id * (lp.cd[0] * gl[0] + lp.cd[1] * gl[1] + lp.cd[2] * gl[2]);
    const float gn[3] = {g_cos * lp.dir[0], g_cos * lp.dir[1], g_cos * lp.dir[2]};
    float gc[3];
    if (len > 1e-5f) {
        const float dot = nrm[0] * gn[0] + nrm[1] * gn[1] + nrm[2] * gn[2];
        for (int k = 0; k < 3; k++) gc[k] = (gn[k] - nrm[k] * dot) / len;
    } else {
        for (int k = 0; k < 3; k++) gc[k] = gn[k] / 1e-5f;
    }
    const float a[3] = {fc[0] - fc[3], fc[1] - fc[4], fc[2] - fc[5]};
    const float bb[3] = {fc[6] - fc[3], fc[7] - fc[4], fc[8] - fc[5]};
    float ga[3], gb[3];
    cross3(bb, gc, ga);
    cross3(gc, a, gb);
    for (int k = 0; k < 3; k++) acc[k] += corner == 0 ? ga[k] : (corner == 2 ? gb[k] : -(ga[k] + gb[k]));
}
__global__ void __launch_bounds__(256) k_face_light_backward_gather(IndexedFaces fs, LightParams lp,
                                                                   const float* __restrict__ g_light,
                                                                   const int32_t* __restrict__ adj_offsets,
                                                                   const int32_t* __restrict__ adj_items,
                                                                   float* __restrict__ grad_vertices) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= fs.V) return;
    float acc[3] = {0.0f, 0.0f, 0.0f};
    if (lp.id != 0) {
        for (int e = adj_offsets[v]; e < adj_offsets[v + 1]; e++) {
            const int item = adj_items[e], f = item / 3, c = item - 3 * f;
            face_light_corner_grad(fs, lp, g_light, f, c, acc);
            if (fs.fill_back) face_light_corner_grad(fs, lp, g_light, fs.Ft + f, 2 - c, acc);
        }
    }
    grad_vertices[3 * (size_t)v + 0] = acc[0]; grad_vertices[3 * (size_t)v + 1] = acc[1]; grad_vertices[3 * (size_t)v + 2] = acc[2];
}

// trilinear sample positions of one covered pixel: KCU:209-231 (shared by the sampler and its backward)
__device__ __forceinline__ void sample_setup(const float* face, const float* weight, float depth, int ts, float eps,
                                             int* fl, float* fr) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float t = weight[k] * (float)(ts - 1) * (depth / face[3 * k + 2]);
        t = fmaxf(t, 0.0f);                              // max(NaN, 0.) = 0 as in CUDA
        t = fminf(t, (float)(ts - 1) - eps);
        fl[k] = f2i(t);
        fr[k] = t - (float)fl[k];
    }
}
__device__ __forceinline__ void sample_corner(int pn, int ts, const int* fl, const float* fr, float& w, int& isc) {
    w = 1;
    int tii[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (((pn >> k) & 1) == 0) { w *= 1 - fr[k]; tii[k] = fl[k]; }
        else                      { w *= fr[k];     tii[k] = fl[k] + 1; }
    }
    isc = tii[0] * ts * ts + tii[1] * ts + tii[2];
}

// colour of one covered pixel: sum_corners w * (texel * light)   (KCU:217-240 on the virtual lit array)
__device__ __forceinline__ void sample_pixel_lit(const float* __restrict__ faces, const LitTextures& lt, int B, int bn, int fi,
                                                 const float* weight, float depth, float eps, float* px) {
    const float* face = faces + ((size_t)bn * lt.Fp + fi) * 9;
    int fl[3];
    float fr[3];
    sample_setup(face, weight, depth, lt.ts, eps, fl, fr);
    px[0] = px[1] = px[2] = 0;
    if (lt.ts == 2) {
        // ts = 2: the sample position is clamped below 1, its integer part is 0, so corner pn is always texel
        // (pn&1, pn>>1&1, pn>>2&1) of THIS face: the whole 2x2x2 cube is 24 contiguous floats (six 16-byte loads) and
        // the face's light is loaded once; same products and the same corner order as the general loop below
        const bool back = fi >= lt.F;
        const int fo = back ? fi - lt.F : fi;
        float tex[24];
        if (lt.im) {
            const int cells = lt.F >> 1, second = fo >= cells, cell = second ? fo - cells : fo;
            const int cy = cell / (lt.im_W - 1), cx = cell - cy * (lt.im_W - 1);
            int v[3];
            tfi_vertices(second, cy, cx, lt.im_W, v);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float* ch = lt.im + ((size_t)bn * 3 + c) * lt.im_H * lt.im_W;
                const float c0 = ch[v[0]], c1 = ch[v[1]], c2 = ch[v[2]];
#pragma unroll
                for (int idx = 0; idx < 8; idx++)
                    tex[idx * 3 + c] = (TFI_CUBE[idx][0] * c0 + TFI_CUBE[idx][1] * c1) + TFI_CUBE[idx][2] * c2;
            }
        } else {
            const float4* cube = (const float4*)(lt.textures + ((size_t)(lt.tex_batch > 1 ? bn : 0) * lt.F + fo) * 24);
#pragma unroll
            for (int q = 0; q < 6; q++) {
                const float4 v = cube[q];
                tex[4 * q] = v.x; tex[4 * q + 1] = v.y; tex[4 * q + 2] = v.z; tex[4 * q + 3] = v.w;
            }
        }
        const float* li = lt.light + 3 * ((size_t)(lt.light_batch > 1 ? bn : 0) * lt.Fp + fi);
        const float l0 = li[0], l1 = li[1], l2 = li[2];
#pragma unroll
        for (int pn = 0; pn < 8; pn++) {
            float w;
            int isc;
            sample_corner(pn, 2, fl, fr, w, isc);
            const int rev = ((pn & 1) << 2) | (pn & 2) | ((pn >> 2) & 1);      // = isc
            const int idx = back ? pn : rev;                                    // back copy: texel (c,b,a)
            px[0] += w * (tex[idx * 3 + 0] * l0);
            px[1] += w * (tex[idx * 3 + 1] * l1);
            px[2] += w * (tex[idx * 3 + 2] * l2);
        }
        return;
    }
#pragma unroll
    for (int pn = 0; pn < 8; pn++) {
        float w;
        int isc, lrow = 0;
        sample_corner(pn, lt.ts, fl, fr, w, isc);
        const long off = lit_texel(lt, B, bn, fi, isc, &lrow);
        if (off >= 0) {
#pragma unroll
            for (int k = 0; k < 3; k++) px[k] += w * (lt.textures[off + k] * lt.light[3 * (size_t)lrow + k]);
        }
    }
}

// forward: rgb_map[b,y,x,:] of every covered pixel
__global__ void __launch_bounds__(256) k_texture_sampling_lit(const float* __restrict__ faces, LitTextures lt,
                                                             const int32_t* __restrict__ face_index_map,
                                                             const float* __restrict__ weight_map,
                                                             const float* __restrict__ depth_map, float* __restrict__ rgb_map,
                                                             int B, int S, float eps) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * S * S) return;
    const int fi = face_index_map[i];
    if (fi < 0) return;
    const int bn = (int)(i / ((long)S * S));
    const float weight[3] = {weight_map[3 * i], weight_map[3 * i + 1], weight_map[3 * i + 2]};
    float px[3];
    sample_pixel_lit(faces, lt, B, bn, fi, weight, depth_map[i], eps, px);
    rgb_map[3 * i + 0] = px[0];
    rgb_map[3 * i + 1] = px[1];
    rgb_map[3 * i + 2] = px[2];
}

// forward, sampling and the output epilogue of rasterize_rgbad in ONE pass (no rgb_sampled round trip): one lane
// per OUTPUT pixel; per internal pixel  rgb = covered ? sampled : background  (what rasterize.py:187-195 computes
// from a zero-initialised rgb_map), alpha = covered; then vertical flip, HWC->CHW and the 2x2 mean when
// anti-aliasing (rasterize.py:305-326).  rgb_blended / alpha_map are the internal-resolution maps the backward
// pass reads.
// The multi-view fit objective (deep3dmap_amd/multiview.py; the composition of photometric_loss on rgb and depth and
// silhouette_loss that k_fit_loss_* evaluate on finished images) evaluated where the images are produced.
struct FitTargets {
    const float *rgb_t, *depth_t, *alpha_t, *mask;   // OUTPUT image layout [B,3,s,s] / [B,s,s]; all NULL = no objective
    float* partials;                                  // [4 * gridDim.x]: sum |rgb - t| m, sum |depth - t| m, sum m, sum (alpha - t)^2
    // the objective's gradient wrt the internal maps, WITHOUT its scalar factors (see GradScale); NULL = not wanted
    float *g_rgb, *g_alpha, *g_depth;
};

// The objective's last step: per-workgroup partial sums -> totals and *loss, in two levels and in a FIXED order
// (deterministic): workgroup g adds up the partials of group g (a view's tiles; 256 consecutive workgroups of the 1-D pass)
// and publishes the sum (sc1) behind a ticket; the workgroup whose ticket is the last adds up the group sums -- nobody
// waits (d3m_tail.h).  Round 4's one-workgroup kernel took 69 us on average (337 at worst) behind the 32768 partials of
// BASELINE config 5.  Run either by k_fit_finish behind the pass that leaves the partials, or -- a caller that runs the
// backward pass right behind the forward pass -- by the first workgroups of a kernel the backward pass launches anyway and
// that normally has nothing to do (k_lit_large_faces): then it costs the step no launch at all.
struct FitFin {
    const float4* partials;   // NULL: nothing to finish
    float4* group_sums;       // [n_groups]
    unsigned* tickets;        // [1] ZEROED: groups done
    int n_partials, group_size, n_groups;
    float pixels;             // output pixels per view (the silhouette term's divisor)
    const float* mask_sum;    // NULL: the batch's own sum of the mask
    float* totals;            // [8]: the four sums, the objective
    float* loss;
};
__device__ __forceinline__ float4 block_sum4_256(float4 v, float4* s_part /*[4]*/) {      // -> every thread
    const float4 w = make_float4(wave_sum(v.x), wave_sum(v.y), wave_sum(v.z), wave_sum(v.w));
    if (lane_id() == 0) s_part[threadIdx.x >> 6] = w;
    __syncthreads();
    float4 t = s_part[0];
    for (int k = 1; k < 4; k++) { t.x += s_part[k].x; t.y += s_part[k].y; t.z += s_part[k].z; t.w += s_part[k].w; }
    __syncthreads();
    return t;
}
// called by all 256 threads of workgroup `block` of a launch of `blocks` workgroups (uniformly)
__device__ __forceinline__ void fit_finish_groups(const FitFin& f, unsigned block, unsigned blocks) {
    __shared__ float4 s_part[4];
    __shared__ int s_last;
    for (unsigned g = block; g < (unsigned)f.n_groups; g += blocks) {
        const int first = (int)g * f.group_size, size = min(f.group_size, f.n_partials - first);
        float4 acc = make_float4(0, 0, 0, 0);
        for (int i = threadIdx.x; i < size; i += 256) {
            const float4 v = f.partials[first + i];           // (written by an earlier launch)
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        acc = block_sum4_256(acc, s_part);
        if (threadIdx.x == 0) {
            tail_store4(&f.group_sums[g], acc);               // sc1: read by another workgroup of this launch
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_last = __hip_atomic_fetch_add(f.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)f.n_groups - 1;
        }
        __syncthreads();
        if (!s_last) continue;                                // (uniform)
        acc = make_float4(0, 0, 0, 0);
        for (int i = threadIdx.x; i < f.n_groups; i += 256) {
            const float4 v = tail_load4(&f.group_sums[i]);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        acc = block_sum4_256(acc, s_part);
        if (threadIdx.x == 0) {
            if (f.mask_sum) acc.z = *f.mask_sum;     // this batch is a shard: normalise by the mask of the whole objective
            f.totals[0] = acc.x; f.totals[1] = acc.y; f.totals[2] = acc.z; f.totals[3] = acc.w;
            const float l = (acc.x / (3.0f * acc.z) + acc.w / f.pixels) + acc.y / acc.z;
            f.totals[4] = l;
            *f.loss = l;
        }
    }
}
__global__ void __launch_bounds__(256) k_fit_finish(FitFin f) { fit_finish_groups(f, blockIdx.x, gridDim.x); }

#ifndef D3M_EPI_MINWAVES
#define D3M_EPI_MINWAVES 4
#endif
__global__ void __launch_bounds__(256, D3M_EPI_MINWAVES) k_render_lit_epilogue(const float* __restrict__ faces, LitTextures lt,
                                                            const int32_t* __restrict__ face_index_map,
                                                            const float* __restrict__ weight_map,
                                                            const float* __restrict__ depth_map,
                                                            const float* __restrict__ background, int bg_b,
                                                            float* __restrict__ rgb_blended, float* __restrict__ alpha_map,
                                                            float* __restrict__ rgb_out, float* __restrict__ alpha_out,
                                                            float* __restrict__ depth_out, int B, int S, int aa, float eps,
                                                            FitTargets fit) {
    __shared__ float4 s_part[4];
    const int s = aa ? S / 2 : S;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    float t_rgb = 0, t_d = 0, t_m = 0, t_sse = 0;
    if (i < (long)B * s * s) {
        const int b = (int)(i / ((long)s * s));
        const int yo = (int)((i / s) % s), xo = (int)(i % s);
        const int n = aa ? 2 : 1;
        const float* bg = background + (size_t)(bg_b > 1 ? b : 0) * 3;
        float acc_rgb[3] = {0, 0, 0}, acc_a = 0, acc_d = 0;
        // the objective's targets are requested first: they then arrive under the dependent loads of the sampling
        float tg[6] = {0, 0, 0, 0, 0, 0};                 // rgb_t x3, depth_t, alpha_t, mask
        if (fit.partials) {
#pragma unroll
            for (int k = 0; k < 3; k++) tg[k] = fit.rgb_t[(((size_t)b * 3 + k) * s + yo) * s + xo];
            tg[3] = fit.depth_t[i]; tg[4] = fit.alpha_t[i]; tg[5] = fit.mask[i];
        }
        for (int dy = 0; dy < n; dy++) {
            for (int dx = 0; dx < n; dx++) {
                const int yi = S - 1 - (yo * n + dy), xi = xo * n + dx;
                const size_t p = ((size_t)b * S + yi) * S + xi;
                const int fi = face_index_map[p];
                const float depth = depth_map[p];
                float v[3] = {bg[0], bg[1], bg[2]};
                if (fi >= 0) {
                    const float weight[3] = {weight_map[3 * p], weight_map[3 * p + 1], weight_map[3 * p + 2]};
                    sample_pixel_lit(faces, lt, B, b, fi, weight, depth, eps, v);
                }
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    rgb_blended[3 * p + k] = v[k];
                    acc_rgb[k] += v[k];
                }
                const float mask = fi >= 0 ? 1.0f : 0.0f;
                if (alpha_map) alpha_map[p] = mask;
                acc_a += mask;
                acc_d += depth;
            }
        }
        const float inv = aa ? 0.25f : 1.0f;
        if (rgb_out) {
#pragma unroll
            for (int k = 0; k < 3; k++) rgb_out[(((size_t)b * 3 + k) * s + yo) * s + xo] = acc_rgb[k] * inv;
        }
        if (alpha_out) alpha_out[i] = acc_a * inv;
        if (depth_out) depth_out[i] = acc_d * inv;
        if (fit.partials) {                               // same terms as k_fit_loss_reduce
            const float m = tg[5];
#pragma unroll
            for (int k = 0; k < 3; k++) t_rgb += fabsf(acc_rgb[k] * inv - tg[k]) * m;
            t_d = fabsf(acc_d * inv - tg[3]) * m;
            t_m = m;
            const float d = acc_a * inv - tg[4];
            t_sse = d * d;
            if (fit.g_rgb) {
                // the unscaled gradient wrt the INTERNAL maps: every internal pixel of the output pixel gets `inv` of the
                // output pixel's gradient (the adjoint of the 2x2 mean).  The readers' scalars (GradScale) are those of the
                // output images except 1 / pixels of the silhouette term, which they take from the INTERNAL size S^2 =
                // s^2 / inv: the alpha gradient therefore carries inv * (1 / inv) = 1.
                auto sgn = [](float x) { return x > 0 ? 1.0f : (x < 0 ? -1.0f : 0.0f); };
                float gr[3];
#pragma unroll
                for (int k = 0; k < 3; k++) gr[k] = sgn(acc_rgb[k] * inv - tg[k]) * m * inv;
                const float gd = sgn(acc_d * inv - tg[3]) * m * inv, ga = 2.0f * d;
                for (int dy = 0; dy < n; dy++) {
                    for (int dx = 0; dx < n; dx++) {
                        const size_t p = ((size_t)b * S + (S - 1 - (yo * n + dy))) * S + xo * n + dx;
#pragma unroll
                        for (int k = 0; k < 3; k++) fit.g_rgb[3 * p + k] = gr[k];
                        fit.g_depth[p] = gd;
                        fit.g_alpha[p] = ga;
                    }
                }
            }
        }
    }
    if (fit.partials) {                                   // four wave sums (DPP), one exchange through LDS
        const float4 t = block_sum4_256(make_float4(t_rgb, t_d, t_m, t_sse), s_part);
        if (threadIdx.x == 0) reinterpret_cast<float4*>(fit.partials)[blockIdx.x] = t;
    }
}

// The same pass for the fused fit objective WITH the edge gradient's per-pixel records (d3m_edge_grad.h, struct
// EdgeGradArgs) as its gradient output: no anti-aliasing, so an output pixel is an internal pixel, and with the
// normaliser of the photometric terms known up front (fit.mask_sum: the mask is a constant of the targets) everything
// a walk reads per pixel is known here except the gradient of the loss itself, which the readers apply (`go`):
//   grad = (2 (alpha - alpha_t) / pixels,  sign(rgb - rgb_t) mask / (3 mask_sum))
//   dot  = (<(alpha, rgb), grad>, owner)
// plus each line's non-zero extent.  k_pack_maps and the unscaled g_rgb / g_alpha maps (16 B per pixel written here,
// 36 B read there) disappear from the step; the gathered texture pass reads grad.yzw (one 16-byte load instead of three
// dwords).  A 32x32 tile per workgroup, so that the column extents are merged in LDS (one atomic per column and tile).
// the lanes of this thread's tile row (TILE consecutive lanes of the wave) out of a wave ballot
template <int TILE>
__device__ __forceinline__ unsigned tile_row_mask(unsigned long long ball) {
    const int sub = (int)(threadIdx.x & 63) / TILE;
    return (unsigned)(ball >> (sub * TILE)) & (TILE == 32 ? 0xFFFFFFFFu : (1u << (TILE & 31)) - 1u);
}
struct FitRecords {
    float4* grad;          // [B,S,S]
    float2* dot;           // [B,S,S]
    int* nz_lo_inv;        // [B,2,S] zeroed: S - (first pixel with a non-zero record), per line (b*2 + axis)*S + d0
    int* nz_hi1;           // [B,2,S] zeroed: last such pixel + 1
    const float* mask_sum; // [1]
    float* g_depth;        // [B,S,S] sign(depth - target) * mask (the depth gradient stays a map: k_backward_textures_lit)
};

// (4 waves per SIMD: the compiler wants 134 registers, six more than four waves allow; held to 128 the pass -- bound by its
//  memory round trips, i.e. by how many of them are in flight -- takes 0.198 instead of 0.233 ms alone; 5 waves spill: 0.35)
#ifndef D3M_FIT_MINWAVES
#define D3M_FIT_MINWAVES 4
#endif
// TILE: 32 (four pixels per thread, eight rows apart) for big batches; 16 (one pixel per thread) for small ones -- one view
// at 512^2 is 256 tiles of 32 x 32, ONE wave per SIMD with four dependent pixels each: 25-31 us whatever the mesh (round 5's
// floor of every small configuration); as 1024 tiles of 16 x 16 the same pixels are four waves per SIMD with one pixel each.
template <int TILE>
__global__ void __launch_bounds__(256, D3M_FIT_MINWAVES) k_render_lit_fit_records(const float* __restrict__ faces, LitTextures lt,
                                                               const int32_t* __restrict__ face_index_map,
                                                               const float* __restrict__ weight_map,
                                                               const float* __restrict__ depth_map,
                                                               const float* __restrict__ background, int bg_b,
                                                               float* __restrict__ rgb_blended, float* __restrict__ alpha_map,
                                                               float* __restrict__ rgb_out, float* __restrict__ alpha_out,
                                                               float* __restrict__ depth_out,
                                                               int B, int S, float eps, FitTargets fit, FitRecords rec) {
    static_assert(TILE == 32 || TILE == 16, "tile of 32 x 32 or 16 x 16 pixels");
    constexpr int ROWS = 256 / TILE;                          // rows of the tile per step of the workgroup
    __shared__ float4 s_part[4];
    __shared__ int s_col_lo_inv[TILE], s_col_hi1[TILE];
    if (threadIdx.x < TILE) { s_col_lo_inv[threadIdx.x] = 0; s_col_hi1[threadIdx.x] = 0; }
    __syncthreads();
    const int b = blockIdx.z, x0 = blockIdx.x * TILE, y0 = blockIdx.y * TILE;
    const int tx = threadIdx.x % TILE, ty = threadIdx.x / TILE;
    const float* bg = background + (size_t)(bg_b > 1 ? b : 0) * 3;
    const float inv_pixels = 1.0f / (float)((long)S * S), inv_3den = 1.0f / (3.0f * *rec.mask_sum);
    float t_rgb = 0, t_d = 0, t_m = 0, t_sse = 0;
    auto sgn = [](float x) { return x > 0 ? 1.0f : (x < 0 ? -1.0f : 0.0f); };
    for (int r = ty; r < TILE; r += ROWS) {
        const int yi = y0 + r, xi = x0 + tx;                  // internal pixel; row 0 = bottom (rasterize.py:311-317)
        bool nz = false;
        if (yi < S && xi < S) {
            const int yo = S - 1 - yi;                        // output row
            const size_t p = ((size_t)b * S + yi) * S + xi, o = ((size_t)b * S + yo) * S + xi;
            // the objective's targets are requested first: they then arrive under the dependent loads of the sampling
            float tg[6];
#pragma unroll
            for (int k = 0; k < 3; k++) tg[k] = fit.rgb_t[(((size_t)b * 3 + k) * S + yo) * S + xi];
            tg[3] = fit.depth_t[o]; tg[4] = fit.alpha_t[o]; tg[5] = fit.mask[o];
            const int fi = face_index_map[p];
            const float depth = depth_map[p];
            float v[3] = {bg[0], bg[1], bg[2]};
            if (fi >= 0) {
                const float weight[3] = {weight_map[3 * p], weight_map[3 * p + 1], weight_map[3 * p + 2]};
                sample_pixel_lit(faces, lt, B, b, fi, weight, depth, eps, v);
            }
            const float alpha = fi >= 0 ? 1.0f : 0.0f;
#pragma unroll
            for (int k = 0; k < 3; k++) rgb_blended[3 * p + k] = v[k];
            alpha_map[p] = alpha;
            if (rgb_out) {                                    // the images as well (CHW, flipped: rasterize.py:305-317)
#pragma unroll
                for (int k = 0; k < 3; k++) rgb_out[(((size_t)b * 3 + k) * S + yo) * S + xi] = v[k];
            }
            if (alpha_out) alpha_out[o] = alpha;
            if (depth_out) depth_out[o] = depth;
            const float m = tg[5], d = alpha - tg[4];
#pragma unroll
            for (int k = 0; k < 3; k++) t_rgb += fabsf(v[k] - tg[k]) * m;      // same terms as k_fit_loss_reduce
            t_d += fabsf(depth - tg[3]) * m;
            t_m += m;
            t_sse += d * d;
            if (rec.grad) {
                float4 g;
                g.x = (2.0f * d) * inv_pixels;
                g.y = (sgn(v[0] - tg[0]) * m) * inv_3den;
                g.z = (sgn(v[1] - tg[1]) * m) * inv_3den;
                g.w = (sgn(v[2] - tg[2]) * m) * inv_3den;
                float dot = alpha * g.x;
                dot += v[0] * g.y;
                dot += v[1] * g.z;
                dot += v[2] * g.w;
                rec.grad[p] = g;
                rec.dot[p] = make_float2(dot, __int_as_float(fi));
                rec.g_depth[p] = sgn(depth - tg[3]) * m;
                nz = g.x != 0 || g.y != 0 || g.z != 0 || g.w != 0 || dot != 0;
            }
        }
        if (rec.grad) {     // non-zero extents: this tile's share of row yi (TILE lanes of a wave = one tile row) and of its columns
            const unsigned half = tile_row_mask<TILE>(__ballot(nz));
            if (half != 0 && tx == 0) {
                const size_t line = ((size_t)b * 2 + 1) * S + (y0 + r);
                atomicMax(&rec.nz_lo_inv[line], S - (x0 + (__ffs((int)half) - 1)));
                atomicMax(&rec.nz_hi1[line], x0 + (32 - __clz((int)half)));
            }
            if (nz) {
                atomicMax(&s_col_lo_inv[tx], S - (y0 + r));
                atomicMax(&s_col_hi1[tx], y0 + r + 1);
            }
        }
    }
    __syncthreads();
    if (rec.grad && threadIdx.x < TILE && s_col_hi1[threadIdx.x] != 0 && x0 + (int)threadIdx.x < S) {
        const size_t line = ((size_t)b * 2 + 0) * S + (x0 + threadIdx.x);
        atomicMax(&rec.nz_lo_inv[line], s_col_lo_inv[threadIdx.x]);
        atomicMax(&rec.nz_hi1[line], s_col_hi1[threadIdx.x]);
    }
    // four wave sums (DPP), one exchange through LDS (a view's tiles are consecutive: one group of the finish)
    const float4 t = block_sum4_256(make_float4(t_rgb, t_d, t_m, t_sse), s_part);
    if (threadIdx.x == 0)
        reinterpret_cast<float4*>(fit.partials)[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t;
}

// The records form WITH anti-aliasing: the objective is taken on the 2x2-pooled images (s = S / 2), the records stay per
// INTERNAL pixel -- the adjoint of the 2x2 mean hands each of an output pixel's four internal pixels a quarter of its
// gradient, and 1 / pixels of the silhouette term times that quarter is 1 / S^2:
//   grad = (2 (alpha_o - alpha_t) / S^2,  sign(rgb_o - rgb_t) mask / 4 / (3 mask_sum))      (the same for all four)
//   dot  = (<(alpha_i, rgb_i), grad>, owner_i)                                              (per internal pixel)
// One thread per output pixel of a 32x32 internal tile (16x16 output pixels); the four internal pixels are sampled in the
// order of k_render_lit_epilogue (upper row first), so the pooled images are bit-identical to its own.  Round 4 wrote
// unscaled gradient MAPS here and packed them in backward (a clear + k_pack_maps: 52 B per internal pixel moved twice).
__global__ void __launch_bounds__(256, 3) k_render_lit_fit_records_pooled(
    const float* __restrict__ faces, LitTextures lt, const int32_t* __restrict__ face_index_map,
    const float* __restrict__ weight_map, const float* __restrict__ depth_map, const float* __restrict__ background, int bg_b,
    float* __restrict__ rgb_blended, float* __restrict__ alpha_map, float* __restrict__ rgb_out, float* __restrict__ alpha_out,
    float* __restrict__ depth_out, int B, int S, float eps, FitTargets fit, FitRecords rec) {
    __shared__ float4 s_part[4];
    __shared__ int s_lo_inv[2][32], s_hi1[2][32];              // [axis][line of the tile]: axis 0 = columns, 1 = rows
    if (threadIdx.x < 64) { (&s_lo_inv[0][0])[threadIdx.x] = 0; (&s_hi1[0][0])[threadIdx.x] = 0; }
    __syncthreads();
    const int b = blockIdx.z, x0 = blockIdx.x * 32, y0 = blockIdx.y * 32, s = S / 2;
    const int ox = threadIdx.x & 15, oy = threadIdx.x >> 4;    // 16 x 16 output pixels
    const int xi0 = x0 + 2 * ox, yi0 = y0 + 2 * oy;            // the lower left of the four internal pixels
    const float* bg = background + (size_t)(bg_b > 1 ? b : 0) * 3;
    const float inv_pixels = 1.0f / (float)((long)S * S), inv_3den = 1.0f / (3.0f * *rec.mask_sum);
    float t_rgb = 0, t_d = 0, t_m = 0, t_sse = 0;
    auto sgn = [](float x) { return x > 0 ? 1.0f : (x < 0 ? -1.0f : 0.0f); };
    if (xi0 < S && yi0 < S) {
        const int xo = xi0 >> 1, yo = s - 1 - (yi0 >> 1);      // output pixel; internal row 0 = bottom (rasterize.py:311-317)
        const size_t o = ((size_t)b * s + yo) * s + xo;
        float tg[6];
#pragma unroll
        for (int k = 0; k < 3; k++) tg[k] = fit.rgb_t[(((size_t)b * 3 + k) * s + yo) * s + xo];
        tg[3] = fit.depth_t[o]; tg[4] = fit.alpha_t[o]; tg[5] = fit.mask[o];
        float vv[4][3], aa[4], acc_rgb[3] = {0, 0, 0}, acc_a = 0, acc_d = 0;
        int own[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {                          // q = dy * 2 + dx of k_render_lit_epilogue: yi = S-1-(2 yo + dy)
            const int yi = yi0 + 1 - (q >> 1), xi = xi0 + (q & 1);
            const size_t p = ((size_t)b * S + yi) * S + xi;
            const int fi = face_index_map[p];
            const float depth = depth_map[p];
            float v[3] = {bg[0], bg[1], bg[2]};
            if (fi >= 0) {
                const float weight[3] = {weight_map[3 * p], weight_map[3 * p + 1], weight_map[3 * p + 2]};
                sample_pixel_lit(faces, lt, B, b, fi, weight, depth, eps, v);
            }
            const float alpha = fi >= 0 ? 1.0f : 0.0f;
#pragma unroll
            for (int k = 0; k < 3; k++) { rgb_blended[3 * p + k] = v[k]; acc_rgb[k] += v[k]; vv[q][k] = v[k]; }
            alpha_map[p] = alpha;
            acc_a += alpha; acc_d += depth;
            aa[q] = alpha; own[q] = fi;
        }
        const float inv = 0.25f;
        if (rgb_out) {
#pragma unroll
            for (int k = 0; k < 3; k++) rgb_out[(((size_t)b * 3 + k) * s + yo) * s + xo] = acc_rgb[k] * inv;
        }
        if (alpha_out) alpha_out[o] = acc_a * inv;
        if (depth_out) depth_out[o] = acc_d * inv;
        const float m = tg[5], d = acc_a * inv - tg[4];
#pragma unroll
        for (int k = 0; k < 3; k++) t_rgb += fabsf(acc_rgb[k] * inv - tg[k]) * m;      // same terms as k_render_lit_epilogue
        t_d = fabsf(acc_d * inv - tg[3]) * m;
        t_m = m;
        t_sse = d * d;
        float4 g;
        g.x = (2.0f * d) * inv_pixels;
        g.y = (sgn(acc_rgb[0] * inv - tg[0]) * m * inv) * inv_3den;
        g.z = (sgn(acc_rgb[1] * inv - tg[1]) * m * inv) * inv_3den;
        g.w = (sgn(acc_rgb[2] * inv - tg[2]) * m * inv) * inv_3den;
        const float gd = sgn(acc_d * inv - tg[3]) * m * inv;
        const bool g_nz = g.x != 0 || g.y != 0 || g.z != 0 || g.w != 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int yi = yi0 + 1 - (q >> 1), xi = xi0 + (q & 1);
            const size_t p = ((size_t)b * S + yi) * S + xi;
            float dot = aa[q] * g.x;
            dot += vv[q][0] * g.y;
            dot += vv[q][1] * g.z;
            dot += vv[q][2] * g.w;
            rec.grad[p] = g;
            rec.dot[p] = make_float2(dot, __int_as_float(own[q]));
            rec.g_depth[p] = gd;
            if (g_nz || dot != 0) {                            // this tile's share of the pixel's row and column extents
                const int r = yi - y0, c = xi - x0;
                atomicMax(&s_lo_inv[1][r], S - xi); atomicMax(&s_hi1[1][r], xi + 1);
                atomicMax(&s_lo_inv[0][c], S - yi); atomicMax(&s_hi1[0][c], yi + 1);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int axis = threadIdx.x >> 5, l = threadIdx.x & 31, d0 = (axis ? y0 : x0) + l;
        if (s_hi1[axis][l] != 0 && d0 < S) {
            const size_t line = ((size_t)b * 2 + axis) * S + d0;
            atomicMax(&rec.nz_lo_inv[line], s_lo_inv[axis][l]);
            atomicMax(&rec.nz_hi1[line], s_hi1[axis][l]);
        }
    }
    const float4 t = block_sum4_256(make_float4(t_rgb, t_d, t_m, t_sse), s_part);
    if (threadIdx.x == 0)
        reinterpret_cast<float4*>(fit.partials)[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t;
}

// The same objective and the same records from FINISHED images: what multiview_fit_loss(*Renderer.render(...)) runs when
// its three images come straight from one lit render node (core/losses.py).  The images hold the very floats the fused pass
// above has in registers (rgb_out = the blended colour, alpha_out = covered, depth_out = the depth map; no anti-aliasing:
// an output pixel is an internal pixel), so partial sums, records, extents and the depth gradient are bit-identical to
// its own -- and backward takes the records route: no gradient images, no k_fit_loss_grad, no k_pack_maps.  Reads
// 20 B (images) + 24 B (targets) + 4 B (owner) per pixel, writes 28 B.
template <int TILE>
__global__ void __launch_bounds__(256) k_fit_loss_records(const float* __restrict__ rgb_im, const float* __restrict__ depth_im,
                                                         const float* __restrict__ alpha_im,
                                                         const int32_t* __restrict__ face_index_map, int B, int S,
                                                         FitTargets fit, FitRecords rec) {
    constexpr int ROWS = 256 / TILE;
    __shared__ float4 s_part[4];
    __shared__ int s_col_lo_inv[TILE], s_col_hi1[TILE];
    if (threadIdx.x < TILE) { s_col_lo_inv[threadIdx.x] = 0; s_col_hi1[threadIdx.x] = 0; }
    __syncthreads();
    const int b = blockIdx.z, x0 = blockIdx.x * TILE, y0 = blockIdx.y * TILE;
    const int tx = threadIdx.x % TILE, ty = threadIdx.x / TILE;
    const float inv_pixels = 1.0f / (float)((long)S * S), inv_3den = 1.0f / (3.0f * *rec.mask_sum);
    float t_rgb = 0, t_d = 0, t_m = 0, t_sse = 0;
    auto sgn = [](float x) { return x > 0 ? 1.0f : (x < 0 ? -1.0f : 0.0f); };
    for (int r = ty; r < TILE; r += ROWS) {
        const int yi = y0 + r, xi = x0 + tx;                  // internal pixel; row 0 = bottom (rasterize.py:311-317)
        bool nz = false;
        if (yi < S && xi < S) {
            const int yo = S - 1 - yi;                        // output row
            const size_t p = ((size_t)b * S + yi) * S + xi, o = ((size_t)b * S + yo) * S + xi;
            float tg[6], v[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                tg[k] = fit.rgb_t[(((size_t)b * 3 + k) * S + yo) * S + xi];
                v[k] = rgb_im[(((size_t)b * 3 + k) * S + yo) * S + xi];
            }
            tg[3] = fit.depth_t[o]; tg[4] = fit.alpha_t[o]; tg[5] = fit.mask[o];
            const int fi = face_index_map[p];
            const float depth = depth_im[o], alpha = alpha_im[o];
            const float m = tg[5], d = alpha - tg[4];
#pragma unroll
            for (int k = 0; k < 3; k++) t_rgb += fabsf(v[k] - tg[k]) * m;      // same terms as k_fit_loss_reduce
            t_d += fabsf(depth - tg[3]) * m;
            t_m += m;
            t_sse += d * d;
            float4 g;
            g.x = (2.0f * d) * inv_pixels;
            g.y = (sgn(v[0] - tg[0]) * m) * inv_3den;
            g.z = (sgn(v[1] - tg[1]) * m) * inv_3den;
            g.w = (sgn(v[2] - tg[2]) * m) * inv_3den;
            float dot = alpha * g.x;
            dot += v[0] * g.y;
            dot += v[1] * g.z;
            dot += v[2] * g.w;
            rec.grad[p] = g;
            rec.dot[p] = make_float2(dot, __int_as_float(fi));
            rec.g_depth[p] = sgn(depth - tg[3]) * m;
            nz = g.x != 0 || g.y != 0 || g.z != 0 || g.w != 0 || dot != 0;
        }
        const unsigned half = tile_row_mask<TILE>(__ballot(nz));
        if (half != 0 && tx == 0) {
            const size_t line = ((size_t)b * 2 + 1) * S + (y0 + r);
            atomicMax(&rec.nz_lo_inv[line], S - (x0 + (__ffs((int)half) - 1)));
            atomicMax(&rec.nz_hi1[line], x0 + (32 - __clz((int)half)));
        }
        if (nz) {
            atomicMax(&s_col_lo_inv[tx], S - (y0 + r));
            atomicMax(&s_col_hi1[tx], y0 + r + 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < TILE && s_col_hi1[threadIdx.x] != 0 && x0 + (int)threadIdx.x < S) {
        const size_t line = ((size_t)b * 2 + 0) * S + (x0 + threadIdx.x);
        atomicMax(&rec.nz_lo_inv[line], s_col_lo_inv[threadIdx.x]);
        atomicMax(&rec.nz_hi1[line], s_col_hi1[threadIdx.x]);
    }
    // four wave sums (DPP), one exchange through LDS (a view's tiles are consecutive: one group of the finish)
    const float4 t = block_sum4_256(make_float4(t_rgb, t_d, t_m, t_sse), s_part);
    if (threadIdx.x == 0)
        reinterpret_cast<float4*>(fit.partials)[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t;
}

// backward, gathered per visible face (ts == 2): sampling weights are recomputed, the 24 sums of
// w * grad_rgb live in LDS, then   grad_textures[view, f, texel] = sum * light   (plain store: within a view
// at most one of the two copies of a face is front-facing) and  grad_light[face] += sum * texel.
// Entries of faces that own no pixel are NOT written: with shared textures the per-view buffer is scratch and
// k_sum_over_views skips them by the visibility flags (no 77 MB zero fill per step on the headline workload);
// a caller-visible per-view gradient is zero-filled by the host wrapper first.
// Lanes per face in the gathered lit pass.  What is done once per face -- bounding box, inverse, the 24 + 9 DPP sums,
// the stores -- is per-lane vector work shared by all the faces of a wave, so fewer lanes per face means fewer
// instructions per face, at the price of more scan steps per lane (each a memory round trip).  Measured on the
// headline step: 8 lanes 2.135 ms, 4 lanes 2.155 ms.
#ifndef D3M_LIT_LANES
#define D3M_LIT_LANES 8
#endif
constexpr int LIT_LANES = D3M_LIT_LANES;
constexpr int LIT_FACES_PER_BLOCK = 256 / LIT_LANES;
__device__ __forceinline__ float lit_sum(float v) {       // over the LIT_LANES adjacent lanes of a face, in every one
    v += dpp_f32<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);      // quad_perm [2,3,0,1]
    if (LIT_LANES == 8) v += dpp_f32<0x141>(v);     // row_half_mirror: the other quad of the 8
    return v;
}

struct LitFaceArgs {
    const float* faces;
    LitTextures lt;
    const int32_t* face_index_map;
    const float *weight_map, *depth_map;
    RgbGrad grad_rgb;
    float* gtex_view;              // [B,F,24]
    float* grad_light;             // [Bm,F',3] zeroed, or NULL
    const float* grad_depth_map;   // or NULL
    float* grad_faces;             // [B,F',9], += (depth gradient), unless vt.gv
    VertexTarget vt;
    int* flags;
    unsigned* view_mask;           // [F, ceil(B/32)] zeroed, or NULL: bit b of word b/32 = "view b wrote gtex_view[b, f]"
    const int* list;               // compacted visible faces + their count, or NULL: every face is tried
    const int* n_list;
    int B, S;
    float eps;
    GradScale gs;
    int* n_large;                  // [1] zeroed: faces handed to the per-pixel kernels (which leave at once when it stays 0)
    int max_area = FM_MAX_BBOX_AREA;   // faces with a larger box go to the per-pixel kernels; INT_MAX in the deterministic mode
                                       // (those add with float atomics in arrival order; here a face's lanes own its sums)
};

// LANES adjacent lanes share a face: LIT_LANES (8) for ordinary meshes, a whole wave (64) for coarse ones, whose faces of
// hundreds of pixels were ninety steps of dependent loads for each of eight lanes (722 triangles @512^2: 210 us of a 337 us
// step).  The arithmetic per pixel and the order of a lane's sums are the same; the sum over the lanes is a tree either way.
template <int LANES>
__device__ __forceinline__ float lit_lanes_sum(float v) { return LANES == 64 ? wave_sum(v) : lit_sum(v); }
template <int LANES>
__device__ __forceinline__ void lit_face_backward(const LitFaceArgs& a, long gi, int sub) {
    const float* __restrict__ faces = a.faces;
    const LitTextures& lt = a.lt;
    const int32_t* __restrict__ face_index_map = a.face_index_map;
    const float* __restrict__ weight_map = a.weight_map;
    const float* __restrict__ depth_map = a.depth_map;
    const RgbGrad grad_rgb = a.grad_rgb;
    float* __restrict__ gtex_view = a.gtex_view;
    float* __restrict__ grad_light = a.grad_light;
    const float* __restrict__ grad_depth_map = a.grad_depth_map;
    float* __restrict__ grad_faces = a.grad_faces;
    const VertexTarget& vt = a.vt;
    int* __restrict__ flags = a.flags;
    const int S = a.S;
    const float eps = a.eps;
    const int Fp = lt.Fp;
    float s_rgb, s_alpha, s_depth;
    a.gs.get(s_rgb, s_alpha, s_depth);
    if (!a.list && flags[gi] == FLAG_HIDDEN) return;              // a listed face is visible: no flag round trip
    const int bn = (int)(gi / Fp), fn = (int)(gi % Fp);
    const float* face = faces + (size_t)gi * 9;
    float fc[9];
#pragma unroll
    for (int k = 0; k < 9; k++) fc[k] = face[k];
    int x0, x1, y0, y1;
    if (!pixel_bbox(fc, S, x0, x1, y0, y1)) return;
    const int area = (x1 - x0 + 1) * (y1 - y0 + 1);
    const int fo = fn >= lt.F ? fn - lt.F : fn;
    float* gt = gtex_view + ((size_t)bn * lt.F + fo) * 24;
    if (sub == 0 && a.view_mask) atomicOr(&a.view_mask[(size_t)fo * ((a.B + 31) >> 5) + (bn >> 5)], 1u << (bn & 31));
    if (area > a.max_area) {              // left to the per-pixel atomic kernel, which adds: give it zeros
        flags[gi] = FLAG_LARGE;
        if (sub == 0) atomicAdd(a.n_large, 1);
        for (int t = sub; t < 24; t += LANES) gt[t] = 0.0f;
        return;
    }
    // ts == 2: the sample position is clamped below 1 (KCU:222-223), so its integer part is 0 and corner pn of the
    // trilinear stencil is ALWAYS texel (pn&1, pn>>1&1, pn>>2&1): the 24 sums have static indices and live in registers
    float acc[24];
#pragma unroll
    for (int t = 0; t < 24; t++) acc[t] = 0.0f;
    const size_t base = (size_t)bn * S * S;
    // the depth gradient (KCU:543-592) rides along when asked for: same pixels, same weights, same depth
    float dacc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dtmp[3] = {0, 0, 0};
    // 1 / z of the three vertices, once per face: the quotients below (KCU:222, :575, :582) become products -- one more
    // rounding each, in a gradient held to 1e-3; the forward pass keeps the reference's divisions.  v_rcp_f32 (1 ulp, one
    // instruction) instead of the correctly rounded quotient (~10): all eight lanes of a face repeat this set-up.
    const float rz[3] = {__builtin_amdgcn_rcpf(fc[2]), __builtin_amdgcn_rcpf(fc[5]), __builtin_amdgcn_rcpf(fc[8])};
    if (grad_depth_map) {
        // the x and y columns of the face inverse (face_inverse(): KCU:24-67) over their common denominator
        float px[3], py[3];
#pragma unroll
        for (int n = 0; n < 3; n++) { px[n] = to_pixel(fc[3 * n], S); py[n] = to_pixel(fc[3 * n + 1], S); }
        const float rden = __builtin_amdgcn_rcpf(px[2] * (py[0] - py[1]) + px[0] * (py[1] - py[2]) + px[1] * (py[2] - py[0]));
        const float ix[3] = {(py[1] - py[2]) * rden, (py[2] - py[0]) * rden, (py[0] - py[1]) * rden};
        const float iy[3] = {(px[2] - px[1]) * rden, (px[0] - px[2]) * rden, (px[1] - px[0]) * rden};
#pragma unroll
        for (int l2 = 0; l2 < 3; l2++) {                                                     // KCU:582
            dtmp[0] += -ix[l2] * rz[l2];
            dtmp[1] += -iy[l2] * rz[l2];
        }
    }
    // what the epilogue needs, requested now: this lane's texel (sub) of the face's cube and the face's light
    const int lrow = (lt.light_batch > 1 ? bn : 0) * Fp + fn;
    const float li[3] = {lt.light[3 * (size_t)lrow], lt.light[3 * (size_t)lrow + 1], lt.light[3 * (size_t)lrow + 2]};
    constexpr int TPL = LANES >= 8 ? 1 : 8 / LANES;       // texels per lane in the epilogue: sub, sub + LANES (lanes >= 8: none)
    const float* tex_face = lt.textures + ((size_t)(lt.tex_batch > 1 ? bn : 0) * lt.F + fo) * 24;
    int to[TPL];
    float tx[TPL][3];
#pragma unroll
    for (int j = 0; j < TPL; j++) {
        const int t = (sub + j * LANES) & 7;               // (lanes 8 .. 63 of a wave-wide face: unused copies)
        to[j] = fn >= lt.F ? ((t & 1) << 2) | (t & 2) | ((t >> 2) & 1) : t;   // (a,b,c) -> (c,b,a) for ts = 2
#pragma unroll
        for (int c3 = 0; c3 < 3; c3++) tx[j][c3] = tex_face[to[j] * 3 + c3];
    }
    BoxCursorN<LANES> c(x0, x1, y0, sub);
    for (int i = sub; i < area; i += LANES, c.advance()) {
        const size_t p = base + (size_t)c.y * S + c.x;
        // Everything the pixel could contribute is requested together with its owner (ONE round trip per step of
        // the scan instead of two); a pixel of another face then computes on stand-in values with zero gradients
        // (selected, never multiplied away: its own weights / depth belong to a different triangle).
        const bool own = face_index_map[p] == fn;
        const float lw0 = weight_map[3 * p], lw1 = weight_map[3 * p + 1], lw2 = weight_map[3 * p + 2];
        const float lg0 = grad_rgb.get(p, 0) * s_rgb, lg1 = grad_rgb.get(p, 1) * s_rgb, lg2 = grad_rgb.get(p, 2) * s_rgb;
        const float ld = depth_map[p], lgd = grad_depth_map ? grad_depth_map[p] * s_depth : 0.0f;
        if (!__builtin_amdgcn_ballot_w64(own)) continue;                  // nobody in the wave owns its pixel
        const float third = 1.0f / 3.0f;
        const float weight[3] = {own ? lw0 : third, own ? lw1 : third, own ? lw2 : third};
        const float g0 = own ? lg0 : 0.0f, g1 = own ? lg1 : 0.0f, g2 = own ? lg2 : 0.0f;
        const float depth = own ? ld : 1.0f;
        if (grad_depth_map) {
            const float g = own ? lgd : 0.0f, depth2 = depth * depth;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float z_k = fc[3 * k + 2];
                dacc[3 * k + 0] += -g * dtmp[0] * weight[k] * depth2 * (float)S / 2.0f;      // KCU:588
                dacc[3 * k + 1] += -g * dtmp[1] * weight[k] * depth2 * (float)S / 2.0f;
                dacc[3 * k + 2] += g * weight[k] * depth2 * (rz[k] * rz[k]);                // KCU:575
            }
        }
        int fl[3] = {0, 0, 0};                    // ts = 2: the position is clamped below 1 (KCU:222-223)
        float fr[3];
#pragma unroll
        for (int k = 0; k < 3; k++) fr[k] = fminf(fmaxf(weight[k] * (depth * rz[k]), 0.0f), 1.0f - eps);
#pragma unroll
        for (int pn = 0; pn < 8; pn++) {
            float w;
            int isc_dyn;
            sample_corner(pn, 2, fl, fr, w, isc_dyn);
            const int isc = ((pn & 1) << 2) | (pn & 2) | ((pn >> 2) & 1);        // = isc_dyn, since fl == 0
            acc[isc * 3 + 0] += w * g0;
            acc[isc * 3 + 1] += w * g1;
            acc[isc * 3 + 2] += w * g2;
        }
    }
#pragma unroll
    for (int t = 0; t < 24; t++) acc[t] = lit_lanes_sum<LANES>(acc[t]);
    if (grad_depth_map) {
#pragma unroll
        for (int k = 0; k < 9; k++) dacc[k] = lit_lanes_sum<LANES>(dacc[k]);
        if (sub == 0) {
            if (vt.gv) {
#pragma unroll
                for (int n = 0; n < 3; n++) {
                    float* g = vt.vertex(bn, fn, n);
#pragma unroll
                    for (int k = 0; k < 3; k++) atomicAdd(&g[k], dacc[3 * n + k]);
                }
            } else {
                float* gf = grad_faces + (size_t)gi * 9;
#pragma unroll
                for (int k = 0; k < 9; k++) gf[k] += dacc[k];
            }
        }
    }
    // epilogue, 8 / LIT_LANES texels per lane (every lane of the face holds all 24 sums after lit_sum): light and
    // texels were requested before the scan
    float gl[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < TPL; j++) {
        float mine[3] = {0, 0, 0};
#pragma unroll
        for (int t = 0; t < 8; t++) {
            if (sub + j * LANES == t) { mine[0] = acc[3 * t]; mine[1] = acc[3 * t + 1]; mine[2] = acc[3 * t + 2]; }
        }
        if (sub + j * LANES < 8) {
#pragma unroll
            for (int c3 = 0; c3 < 3; c3++) {
                gt[to[j] * 3 + c3] = mine[c3] * li[c3];      // plain store: see the kernel comment
                gl[c3] += mine[c3] * tx[j][c3];
            }
        }
    }
#pragma unroll
    for (int c3 = 0; c3 < 3; c3++) gl[c3] = lit_lanes_sum<LANES>(gl[c3]);
    if (grad_light && sub == 0) {
        atomicAdd(&grad_light[3 * (size_t)lrow + 0], gl[0]);
        atomicAdd(&grad_light[3 * (size_t)lrow + 1], gl[1]);
        atomicAdd(&grad_light[3 * (size_t)lrow + 2], gl[2]);
    }
}

// FM_LANES lanes per face.  Without a list every face of [B,F'] gets its lanes (hidden ones leave at once: ~95 % of
// a fill_back mesh, i.e. mostly idle waves); with the compacted list of a d3m_visibility only faces that own a pixel
// do, on a fixed grid that strides over the list.
// (111 registers, 4 waves per SIMD; held to 5 or 6 waves it spills and loses: 0.26 / 0.34 ms against 0.22)
template <int LANES = LIT_LANES>
__global__ void __launch_bounds__(256) k_backward_textures_lit_faces(LitFaceArgs a) {
    static_assert(LANES == LIT_LANES || LANES == 64, "LIT_LANES lanes per face, or a wave");
    constexpr int PER_BLOCK = 256 / LANES;
    const int sub = threadIdx.x % LANES, slot = threadIdx.x / LANES;
    if (a.list) {
        const int n = *a.n_list;
        const XcdOrder xo((n + PER_BLOCK - 1) / PER_BLOCK);     // neighbouring faces share map lines
        for (int i = blockIdx.x; xo.more(i); i += gridDim.x) {
            const long base = (long)xo.unit(i) * PER_BLOCK;
            if (base + slot < n) lit_face_backward<LANES>(a, a.list[base + slot], sub);
        }
    } else {
        const long gi = (long)blockIdx.x * PER_BLOCK + slot;
        if (gi < (long)a.B * a.lt.Fp) lit_face_backward<LANES>(a, gi, sub);
    }
}

// ---- the gathered pass for texture cubes of 3^3 and 4^3 texels (round 5) ----------------------------------------------
// neural_renderer's own default is texture_size 4 (NR/load_obj.py), and only ts = 2 had a gathered backward: every other
// size took the per-pixel atomic passes -- 24 float atomics per covered pixel into a zero-filled [B,F,ts^3,3] buffer, nine
// more for the depth gradient, a sum over ALL views behind them: 8 views of the headline mesh 2.8 ms against 0.49 at ts = 2,
// a 2 450-triangle mesh 4.1 against 0.32.  Same structure as lit_face_backward -- eight lanes per listed face share the scan
// of its box, the depth gradient rides along, one plain store per texel at the end -- with the face's ts^3 x 3 texel sums in
// LDS (192 floats at ts = 4, thirty-two faces per workgroup) instead of registers: a corner's texel depends on the pixel.
// (ts = 1 keeps the per-pixel pass: its corners reach into the FOLLOWING faces' texels, KCU:229-233.)
constexpr int LIT_ANY_MAX_TEXELS = 64;
// (LANES: LIT_LANES, or a whole wave per face on coarse meshes -- as lit_face_backward)
template <int LANES>
__device__ __forceinline__ void lit_face_backward_any(const LitFaceArgs& a, long gi, int sub, float* __restrict__ acc_lds) {
    const LitTextures& lt = a.lt;
    const int S = a.S, Fp = lt.Fp, ts = lt.ts, ts3 = ts * ts * ts;
    float s_rgb, s_alpha, s_depth;
    a.gs.get(s_rgb, s_alpha, s_depth);
    const int bn = (int)(gi / Fp), fn = (int)(gi % Fp);
    float fc[9];
#pragma unroll
    for (int k = 0; k < 9; k++) fc[k] = a.faces[(size_t)gi * 9 + k];
    int x0, x1, y0, y1;
    if (!pixel_bbox(fc, S, x0, x1, y0, y1)) return;
    const int area = (x1 - x0 + 1) * (y1 - y0 + 1);
    const bool back = fn >= lt.F;
    const int fo = back ? fn - lt.F : fn;
    float* gt = a.gtex_view + ((size_t)bn * lt.F + fo) * ts3 * 3;
    if (sub == 0 && a.view_mask) atomicOr(&a.view_mask[(size_t)fo * ((a.B + 31) >> 5) + (bn >> 5)], 1u << (bn & 31));
    if (area > a.max_area) {              // left to the per-pixel atomic kernel, which adds: give it zeros
        a.flags[gi] = FLAG_LARGE;
        if (sub == 0) atomicAdd(a.n_large, 1);
        for (int t = sub; t < ts3 * 3; t += LANES) gt[t] = 0.0f;
        return;
    }
    for (int t = sub; t < ts3 * 3; t += LANES) acc_lds[t] = 0.0f;
    float dacc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dtmp[3] = {0, 0, 0};
    if (a.grad_depth_map) {
        float finv[9];
        face_inverse(fc, S, finv);
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
            for (int l = 0; l < 3; l++) dtmp[k] += -finv[3 * l + k] / fc[3 * l + 2];     // KCU:582
        }
    }
    const int lrow = (lt.light_batch > 1 ? bn : 0) * Fp + fn;
    const float li[3] = {lt.light[3 * (size_t)lrow], lt.light[3 * (size_t)lrow + 1], lt.light[3 * (size_t)lrow + 2]};
    const float* tex_face = lt.textures + ((size_t)(lt.tex_batch > 1 ? bn : 0) * lt.F + fo) * ts3 * 3;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // (the zeroes before the first sums: one wave, in order)
    __builtin_amdgcn_wave_barrier();
    const size_t base = (size_t)bn * S * S;
    BoxCursorN<LANES> c(x0, x1, y0, sub);
    for (int i = sub; i < area; i += LANES, c.advance()) {
        const size_t p = base + (size_t)c.y * S + c.x;
        // (the pixel's maps are requested together with its owner: one round trip per step of the scan)
        const bool own = a.face_index_map[p] == fn;
        const float lw[3] = {a.weight_map[3 * p], a.weight_map[3 * p + 1], a.weight_map[3 * p + 2]};
        const float lg[3] = {a.grad_rgb.get(p, 0) * s_rgb, a.grad_rgb.get(p, 1) * s_rgb, a.grad_rgb.get(p, 2) * s_rgb};
        const float ld = a.depth_map[p], lgd = a.grad_depth_map ? a.grad_depth_map[p] * s_depth : 0.0f;
        if (!own) continue;
        if (a.grad_depth_map) {
            const float depth2 = ld * ld;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float z_k = fc[3 * k + 2];
                dacc[3 * k + 0] += -lgd * dtmp[0] * lw[k] * depth2 * (float)S / 2.0f;      // KCU:588
                dacc[3 * k + 1] += -lgd * dtmp[1] * lw[k] * depth2 * (float)S / 2.0f;
                dacc[3 * k + 2] += lgd * lw[k] * depth2 / (z_k * z_k);                     // KCU:575
            }
        }
        int fl[3];
        float fr[3];
        sample_setup(fc, lw, ld, ts, a.eps, fl, fr);
#pragma unroll
        for (int pn = 0; pn < 8; pn++) {
            float w;
            int isc;
            sample_corner(pn, ts, fl, fr, w, isc);                  // (ts >= 2: the position is clamped below ts - 1: isc < ts^3)
#pragma unroll
            for (int k = 0; k < 3; k++) atomicAdd(&acc_lds[isc * 3 + k], w * lg[k]);
        }
    }
    if (a.grad_depth_map) {
#pragma unroll
        for (int k = 0; k < 9; k++) dacc[k] = lit_lanes_sum<LANES>(dacc[k]);
        if (sub == 0) {
            if (a.vt.gv) {
#pragma unroll
                for (int n = 0; n < 3; n++) {
                    float* g = a.vt.vertex(bn, fn, n);
#pragma unroll
                    for (int k = 0; k < 3; k++) atomicAdd(&g[k], dacc[3 * n + k]);
                }
            } else {
                float* gf = a.grad_faces + (size_t)gi * 9;
#pragma unroll
                for (int k = 0; k < 9; k++) gf[k] += dacc[k];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // (every lane's sums before any lane reads them)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float gl[3] = {0, 0, 0};
    for (int t = sub; t < ts3; t += LANES) {
        // texel t = (a,b,c) of the virtual face is texel (c,b,a) of the original one for the back copy (NR/renderer.py:156)
        const int to = back ? (t % ts) * ts * ts + ((t / ts) % ts) * ts + t / (ts * ts) : t;
#pragma unroll
        for (int c3 = 0; c3 < 3; c3++) {
            const float m = acc_lds[t * 3 + c3];
            gt[to * 3 + c3] = m * li[c3];                           // plain store (see k_backward_textures_lit_faces)
            gl[c3] += m * tex_face[to * 3 + c3];
        }
    }
#pragma unroll
    for (int c3 = 0; c3 < 3; c3++) gl[c3] = lit_lanes_sum<LANES>(gl[c3]);
    if (a.grad_light && sub == 0) {
#pragma unroll
        for (int c3 = 0; c3 < 3; c3++) atomicAdd(&a.grad_light[3 * (size_t)lrow + c3], gl[c3]);
    }
}
// over the compacted visibility list only (a fixed grid striding, as k_backward_textures_lit_faces with a list)
template <int LANES = LIT_LANES>
__global__ void __launch_bounds__(256) k_backward_textures_lit_faces_any(LitFaceArgs a) {
    constexpr int PER_BLOCK = 256 / LANES;
    __shared__ float s_acc[PER_BLOCK][LIT_ANY_MAX_TEXELS * 3];
    const int sub = threadIdx.x % LANES, slot = threadIdx.x / LANES;
    const int n = *a.n_list;
    const XcdOrder xo((n + PER_BLOCK - 1) / PER_BLOCK);
    for (int i = blockIdx.x; xo.more(i); i += gridDim.x) {
        const long base = (long)xo.unit(i) * PER_BLOCK;
        if (base + slot < n) lit_face_backward_any<LANES>(a, a.list[base + slot], sub, s_acc[slot]);
    }
}

// ts = 2: the 24 + 3 sums of one face (corner pn's channels at 3 * pn, the light's at 24) into the per-view texel
// gradients and the light's gradient
__device__ __forceinline__ void lit_large_flush(const LitTextures& lt, int B, float* __restrict__ gtex_view,
                                                float* __restrict__ grad_light, int bn, int fi, int lrow, const float* sums) {
#pragma unroll
    for (int pn = 0; pn < 8; pn++) {
        const int isc = ((pn & 1) << 2) | (pn & 2) | ((pn >> 2) & 1);
        int lr = 0;
        const long off = lit_texel(lt, B, bn, fi, isc, &lr);
        float* gt = gtex_view + (size_t)bn * lt.F * 24 + off % ((long)lt.F * 24);
#pragma unroll
        for (int k = 0; k < 3; k++) atomicAdd(&gt[k], sums[3 * pn + k]);
    }
    if (grad_light) {
#pragma unroll
        for (int k = 0; k < 3; k++) atomicAdd(&grad_light[3 * (size_t)lrow + k], sums[24 + k]);
    }
}

// backward, per pixel with float atomics: any ts, and the faces the gathered form marked LARGE
__device__ __forceinline__ void backward_textures_lit_pixels(const float* __restrict__ faces, const LitTextures& lt,
                                                             const int32_t* __restrict__ face_index_map,
                                                             const float* __restrict__ weight_map,
                                                             const float* __restrict__ depth_map, RgbGrad grad_rgb,
                                                             float* __restrict__ gtex_view /*[B,F,ts^3,3]*/,
                                                             float* __restrict__ grad_light,
                                                             const int* __restrict__ only_large, int B, int S, float eps,
                                                             GradScale gs) {
    // a fixed grid striding over the pixels: in only_large mode the launch normally has nothing to do, and 32 k
    // workgroups that leave at once still cost 12 us of dispatch.
    // WAVE AGGREGATION (round 5): the pixels of a wave normally belong to ONE face here -- this is the pass of faces whose
    // box exceeds the gathered pass's limit -- so for ts = 2 (where a corner's texel is the same for every pixel of the
    // face) the 24 + 3 sums are taken over the wave first and its first lane adds them: 27 atomics per wave instead of
    // 27 x 64 on the same few addresses (an 8-triangle mesh filling a 512^2 image: 14.75 ms -> see EXPERIMENTS).  A wave with
    // pixels of several faces, and any other ts, adds per pixel as before.
    const long n = (long)B * S * S;
    __shared__ WgSums<27> wg;               // (per face: the workgroup's 24 texel + 3 light sums, flushed once -- see WgSums)
    wg.init();
    // (a contiguous run of pixels per workgroup -- a few image rows, i.e. few faces -- in steps of one workgroup)
    const long chunk = ((n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const long run_end = min(n, (long)(blockIdx.x + 1) * chunk);
    for (long i0 = (long)blockIdx.x * chunk + (threadIdx.x & ~63); i0 < run_end; i0 += 256) {
    const long i = i0 + (threadIdx.x & 63);
    int fi = -1, bn = 0;
    bool active = i < run_end;
    if (active) { fi = face_index_map[i]; active = fi >= 0; }
    if (active) {
        bn = (int)(i / ((long)S * S));
        if (only_large && only_large[(size_t)bn * lt.Fp + fi] != FLAG_LARGE) active = false;
    }
    const unsigned long long act = __builtin_amdgcn_ballot_w64(active);
    if (!act) continue;                                             // (wave-uniform)
    const int key = active ? bn * lt.Fp + fi : -1;
    const int lane = (int)(threadIdx.x & 63);
    float g[3] = {0, 0, 0};
    int fl[3] = {0, 0, 0};
    float fr[3] = {0, 0, 0};
    if (active) {
        const float* face = faces + ((size_t)bn * lt.Fp + fi) * 9;
        const float weight[3] = {weight_map[3 * i], weight_map[3 * i + 1], weight_map[3 * i + 2]};
        float s_rgb, s_alpha, s_depth;
        gs.get(s_rgb, s_alpha, s_depth);
        g[0] = grad_rgb.get(i, 0) * s_rgb; g[1] = grad_rgb.get(i, 1) * s_rgb; g[2] = grad_rgb.get(i, 2) * s_rgb;
        sample_setup(face, weight, depth_map[i], lt.ts, eps, fl, fr);
    }
    const int ts3 = lt.ts * lt.ts * lt.ts;
    if (lt.ts != 2) {                       // a corner's texel depends on the pixel: per-pixel atomics
        if (!active) continue;
#pragma unroll
        for (int pn = 0; pn < 8; pn++) {
            float w;
            int isc, lrow = 0;
            sample_corner(pn, lt.ts, fl, fr, w, isc);
            const long off = lit_texel(lt, B, bn, fi, isc, &lrow);
            if (off < 0) continue;
            // off addresses `textures`; the per-view gradient buffer has the same [.., F, ts^3, 3] tail but batch B
            const long in_batch = off % ((long)lt.F * ts3 * 3);
            long vb = ((long)bn * lt.Fp + fi + isc / ts3) / lt.Fp;       // view of the face actually reached
            float* gt = gtex_view + (size_t)vb * lt.F * ts3 * 3 + in_batch;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                atomicAdd(&gt[k], w * g[k] * lt.light[3 * (size_t)lrow + k]);
                if (grad_light) atomicAdd(&grad_light[3 * (size_t)lrow + k], w * g[k] * lt.textures[off + k]);
            }
        }
        continue;
    }
    // ts = 2: one face of the wave at a time (a wave of 64 consecutive pixels of a row holds one to three), every lane in step
    // (up to four; a wave with more -- small faces beside a large one -- lets the remaining pixels add for themselves)
    unsigned long long todo = act;
    for (int round = 0; round < 4 && todo; round++) {               // (wave-uniform)
        const int lead = __builtin_ctzll(todo);
        const int key0 = __builtin_amdgcn_readlane(key, lead);
        const bool mine = active && key == key0;
        todo &= ~__builtin_amdgcn_ballot_w64(mine);
        // (the face's texels and light addressed once, without lit_texel's 64-bit quotients: ts = 2 stays inside the face)
        const int bn_a = key0 / lt.Fp, fi_a = key0 - bn_a * lt.Fp;
        const bool back = fi_a >= lt.F;
        const float* tex_face = lt.textures + ((size_t)(lt.tex_batch > 1 ? bn_a : 0) * lt.F + (back ? fi_a - lt.F : fi_a)) * 24;
        const int lrow_a = (lt.light_batch > 1 ? bn_a : 0) * lt.Fp + fi_a;
        const float li[3] = {lt.light[3 * (size_t)lrow_a], lt.light[3 * (size_t)lrow_a + 1], lt.light[3 * (size_t)lrow_a + 2]};
        float gl_sum[3] = {0, 0, 0}, sums[27];      // sums: corner pn's three channels at 3 * pn, the light's at 24
#pragma unroll
        for (int pn = 0; pn < 8; pn++) {
            float w;
            int isc;
            sample_corner(pn, 2, fl, fr, w, isc);                   // (fl = 0: isc does not depend on the pixel)
            isc = ((pn & 1) << 2) | (pn & 2) | ((pn >> 2) & 1);
            if (!mine) w = 0.0f;
            const int idx = back ? ((isc & 1) << 2) | (isc & 2) | ((isc >> 2) & 1) : isc;   // back copy: texel (c,b,a)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                sums[3 * pn + k] = wave_sum(w * g[k]) * li[k];
                gl_sum[k] += w * g[k] * tex_face[idx * 3 + k];
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) sums[24 + k] = grad_light ? wave_sum(gl_sum[k]) : 0.0f;
        if (lane == lead && !wg.add(key0, sums))
            lit_large_flush(lt, B, gtex_view, grad_light, bn_a, fi_a, lrow_a, sums);
    }
    if ((todo >> lane) & 1ull) {
#pragma unroll
        for (int pn = 0; pn < 8; pn++) {
            float w;
            int isc, lrow = 0;
            sample_corner(pn, 2, fl, fr, w, isc);
            const long off = lit_texel(lt, B, bn, fi, isc, &lrow);
            float* gt = gtex_view + (size_t)bn * lt.F * ts3 * 3 + off % ((long)lt.F * ts3 * 3);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                atomicAdd(&gt[k], w * g[k] * lt.light[3 * (size_t)lrow + k]);
                if (grad_light) atomicAdd(&grad_light[3 * (size_t)lrow + k], w * g[k] * lt.textures[off + k]);
            }
        }
    }
    }
    __syncthreads();
    if (lt.ts == 2) {
        for (int slot = threadIdx.x; slot < wg.slots; slot += blockDim.x) {    // the workgroup's sums: 27 atomics per face
            const int key = wg.key[slot];
            if (key < 0) continue;
            int lrow = 0;
            lit_texel(lt, B, key / lt.Fp, key % lt.Fp, 0, &lrow);
            lit_large_flush(lt, B, gtex_view, grad_light, key / lt.Fp, key % lt.Fp, lrow, wg.v[slot]);
        }
    }
    __syncthreads();                        // (the table may be initialised again by a caller's next pass)
}

__global__ void __launch_bounds__(256) k_backward_textures_lit_pixels(const float* __restrict__ faces, LitTextures lt,
                                                                     const int32_t* __restrict__ face_index_map,
                                                                     const float* __restrict__ weight_map,
                                                                     const float* __restrict__ depth_map, RgbGrad grad_rgb,
                                                                     float* __restrict__ gtex_view /*[B,F,ts^3,3]*/,
                                                                     float* __restrict__ grad_light,
                                                                     const int* __restrict__ only_large, int B, int S,
                                                                     float eps, GradScale gs,
                                                                     const int* __restrict__ n_large) {
    if (n_large && *n_large == 0) return;          // only_large mode and no such face: nothing to do (uniform exit)
    backward_textures_lit_pixels(faces, lt, face_index_map, weight_map, depth_map, grad_rgb, gtex_view, grad_light,
                                 only_large, B, S, eps, gs);
}

// The gathered pass's two fallbacks for the faces it marked LARGE -- texel gradients and the depth gradient, per pixel with
// float atomics -- as ONE launch: normally there is no such face and the launch leaves at once, and with the step's kernels
// on one stream every launch that leaves at once is still ~4.5 us of the step.
__global__ void __launch_bounds__(256) k_lit_large_faces(const float* __restrict__ faces, LitTextures lt,
                                                        const int32_t* __restrict__ face_index_map,
                                                        const float* __restrict__ weight_map,
                                                        const float* __restrict__ depth_map, RgbGrad grad_rgb,
                                                        float* __restrict__ gtex_view, float* __restrict__ grad_light,
                                                        const int* __restrict__ flags, int B, int S, float eps, GradScale gs,
                                                        const int* __restrict__ n_large,
                                                        const float* __restrict__ grad_depth_map, float* __restrict__ grad_faces,
                                                        VertexTarget vt, FitFin fin) {
    // (a fused objective whose finish the forward pass left to the backward pass: this launch's first workgroups do it)
    if (fin.partials) fit_finish_groups(fin, blockIdx.x, gridDim.x);
    if (*n_large == 0) return;                     // (uniform exit)
    backward_textures_lit_pixels(faces, lt, face_index_map, weight_map, depth_map, grad_rgb, gtex_view, grad_light, flags, B, S,
                                 eps, gs);
    if (grad_depth_map)
        backward_depth_map_pixels(DenseFaces{faces, lt.Fp}, depth_map, face_index_map, (const float*)nullptr, weight_map,
                                  grad_depth_map, grad_faces, B, S, flags, vt, gs);
}

// out[j] = sum_b in[b, j]  (shared textures: per-view gradients -> one gradient).  With `flags` ([B, F'] visibility
// of the front and, at +F, back copies) only the views in which the face owns a pixel are read: the others were
// never written.
__global__ void __launch_bounds__(256) k_sum_over_views(const float* __restrict__ in, float* __restrict__ out, long n, int B,
                                                       const int* __restrict__ flags, int F, int Fp, int per_face) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int f = (int)(j / per_face);
    float acc = 0;
    for (int b = 0; b < B; b++) {
        if (flags) {
            const int* fl = flags + (size_t)b * Fp;
            if (fl[f] == FLAG_HIDDEN && (Fp == F || fl[F + f] == FLAG_HIDDEN)) continue;
        }
        acc += in[(size_t)b * n + j];
    }
    out[j] = acc;
}

// The ts == 2 form of the above: one lane per 4 consecutive floats of a face's 24 (six lanes per face), 16-byte loads.
// view_mask (written by k_backward_textures_lit_faces) names the views that wrote this face's entry, in view order:
// typically 4 of 32, found with ffs instead of 2 x B flag loads.
// (per_face4 float4s per face: 6 at ts = 2, 48 at ts = 4 -- 3^3 x 3 floats are no multiple of four: the form above)
__global__ void __launch_bounds__(256) k_sum_over_views_ts2(const float4* __restrict__ in, float4* __restrict__ out, long n4,
                                                           int B, const unsigned* __restrict__ view_mask, int per_face4) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n4) return;
    const int f = (int)(j / per_face4), words = (B + 31) >> 5;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int wd = 0; wd < words; wd++) {
        unsigned m = view_mask[(size_t)f * words + wd];
        while (m) {                                       // up to four views' entries requested per round, added in view order
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v[k] = make_float4(0, 0, 0, 0);
                if (m) {
                    const int b = (wd << 5) + __ffs((int)m) - 1;
                    m &= m - 1;
                    v[k] = in[(size_t)b * n4 + j];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
        }
    }
    out[j] = acc;
}

}  // namespace d3m
