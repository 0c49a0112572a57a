/*
 * nr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, f32) of the differentiable mesh rasterizer that
 * achao2013/deep3dmap vendors as pnpmodules/neural_renderer.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (deep3dmap_amd/) never does.
 *
 * Reference file (abbreviated KCU below):
 *   pnpmodules/neural_renderer/neural_renderer/cuda/rasterize_cuda_kernel.cu
 *
 * Parity pins (see DESIGN.md "Oracle"):
 *   - the reference's hand-written known-answer gradients
 *     (pnpmodules/neural_renderer/tests/test_rasterize_silhouettes.py:37-99,
 *      tests/test_rasterize.py:84-156),
 *   - golden vectors under tests/golden/: kern_golden.npz / tex_golden.npz
 *     made by the reference's OWN kernels run on the device
 *     (oracle/_ref/libnr_ref_hip.so = their text compiled by hipcc where it
 *     lies, oracle/Makefile `ref_hip`; tests/golden/make_golden_kern.py), and
 *     nr_golden.npz made by importing the reference's pure-torch modules;
 *   - on the GPU box, fresh random scenes against that device build
 *     (tests/test_gpu_reference.py).
 *
 * Arithmetic notes.  The reference is templated on scalar_t; this restates
 * the scalar_t = float instantiation, keeping every place where a double
 * literal promotes the expression (KCU:47,96-97,129,136,312-313,404) and the
 * CUDA conversion rules the kernels rely on: fmax/fmin drop a NaN operand,
 * and float/double -> int conversion saturates with NaN -> 0.
 * Build with -ffp-contract=off so every f32 operation rounds once.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* CUDA float->int conversion: round toward zero, saturate, NaN -> 0. */
static inline int cuda_d2i(double v) {
    if (v != v) return 0;
    if (v >= 2147483647.0) return INT_MAX;
    if (v <= -2147483648.0) return INT_MIN;
    return (int)v;
}
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORC_API void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* back-face predicate shared by K1/K2/K4 (KCU:40, 111, 270). */
static inline int is_backside(const float *f) {
    return (f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0]);
}

/* ---------------------------------------------------------------------------
 * K1: per-face inverse of [[x0,x1,x2],[y0,y1,y2],[1,1,1]] in pixel space.
 * Follows forward_face_index_map_cuda_kernel_1, KCU:24-67.  Culled faces are
 * left untouched (the caller pre-zeroes faces_inv, rasterize.py:161).
 * ------------------------------------------------------------------------- */
static inline void face_inverse(const float *face, int is, float *out) {
    float p[3][2];
    for (int n = 0; n < 3; n++)
        for (int d = 0; d < 2; d++)
            p[n][d] = (float)(0.5 * (double)(face[3 * n + d] * (float)is + (float)is - 1.0f));   /* KCU:47 */
    float fi[9] = {
        p[1][1] - p[2][1], p[2][0] - p[1][0], p[1][0] * p[2][1] - p[2][0] * p[1][1],
        p[2][1] - p[0][1], p[0][0] - p[2][0], p[2][0] * p[0][1] - p[0][0] * p[2][1],
        p[0][1] - p[1][1], p[1][0] - p[0][0], p[0][0] * p[1][1] - p[1][0] * p[0][1]};      /* KCU:52-55 */
    float den = (p[2][0] * (p[0][1] - p[1][1]) +
                 p[0][0] * (p[1][1] - p[2][1]) +
                 p[1][0] * (p[2][1] - p[0][1]));                                            /* KCU:56-59 */
    for (int k = 0; k < 9; k++) out[k] = fi[k] / den;                                       /* KCU:60-62 */
}

ORC_API void orc_face_inverse(const float *faces, float *faces_inv, int batch_size, int num_faces,
                              int image_size) {
    const long n = (long)batch_size * num_faces;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; i++) {
        const float *face = faces + i * 9;
        if (is_backside(face)) continue;                                                    /* KCU:40 */
        face_inverse(face, image_size, faces_inv + i * 9);
    }
}

/* One (pixel, face) evaluation of K2's loop body, KCU:110-139.
 * Returns 1 and fills w[3], *zp_out when the face covers the pixel inside
 * (near, far); 0 otherwise.  xi, yi are integer pixel coordinates, xp, yp the
 * NDC pixel centre (KCU:96-97). */
static inline int pixel_face_eval(const float *face, const float *face_inv, int xi, int yi,
                                  float xp, float yp, float near, float far, float *w, float *zp_out) {
    if (is_backside(face)) return 0;                                                        /* KCU:111 */
    if (((yp - face[1]) * (face[3] - face[0]) < (xp - face[0]) * (face[4] - face[1])) ||
        ((yp - face[4]) * (face[6] - face[3]) < (xp - face[3]) * (face[7] - face[4])) ||
        ((yp - face[7]) * (face[0] - face[6]) < (xp - face[6]) * (face[1] - face[7])))
        return 0;                                                                           /* KCU:115-118 */
    const float fx = (float)xi, fy = (float)yi;
    w[0] = face_inv[0] * fx + face_inv[1] * fy + face_inv[2];
    w[1] = face_inv[3] * fx + face_inv[4] * fy + face_inv[5];
    w[2] = face_inv[6] * fx + face_inv[7] * fy + face_inv[8];                               /* KCU:122-124 */
    float w_sum = 0;
    for (int k = 0; k < 3; k++) {
        w[k] = (float)fmin(fmax((double)w[k], 0.), 1.);                                     /* KCU:129 */
        w_sum += w[k];
    }
    for (int k = 0; k < 3; k++) w[k] /= w_sum;                                              /* KCU:132-134 */
    const float zp = (float)(1. / (double)(w[0] / face[2] + w[1] / face[5] + w[2] / face[8]));  /* KCU:136 */
    if (zp <= near || far <= zp) return 0;                                                  /* KCU:137 */
    *zp_out = zp;
    return 1;
}

static inline float pixel_center(int i, int is) {
    return (float)((2. * i + 1 - is) / is);                                                 /* KCU:96-97 */
}

/* ---------------------------------------------------------------------------
 * K2, brute force: every pixel loops over every face in index order.
 * Follows forward_face_index_map_cuda_kernel_2, KCU:70-169.  Outputs must be
 * pre-filled by the caller (face_index -1, weight 0, depth far, face_inv 0:
 * rasterize.py:50-69); only covered pixels are written.
 * face_inv_map may be NULL (return_depth == 0).
 * ------------------------------------------------------------------------- */
static void face_index_map_range(const float *faces, const float *faces_inv, int32_t *face_index_map,
                                 float *weight_map, float *depth_map, float *face_inv_map,
                                 int num_faces, int image_size, float near, float far,
                                 int return_depth, long pixel_begin, long pixel_end) {
    const int is = image_size, nf = num_faces;
#pragma omp parallel for schedule(dynamic, 64)
    for (long i = pixel_begin; i < pixel_end; i++) {
        const int bn = (int)(i / ((long)is * is));
        const int pn = (int)(i % ((long)is * is));
        const int yi = pn / is, xi = pn % is;
        const float yp = pixel_center(yi, is), xp = pixel_center(xi, is);
        float depth_min = far;                                                              /* KCU:101 */
        int face_index_min = -1;
        float weight_min[3] = {0, 0, 0};
        for (int fn = 0; fn < nf; fn++) {
            const float *face = faces + ((long)bn * nf + fn) * 9;
            const float *finv = faces_inv + ((long)bn * nf + fn) * 9;
            float w[3], zp;
            if (!pixel_face_eval(face, finv, xi, yi, xp, yp, near, far, w, &zp)) continue;
            if (zp < depth_min) {                                                           /* KCU:142 */
                depth_min = zp;
                face_index_min = fn;
                weight_min[0] = w[0]; weight_min[1] = w[1]; weight_min[2] = w[2];
            }
        }
        if (0 <= face_index_min) {                                                          /* KCU:157-168 */
            depth_map[i] = depth_min;
            face_index_map[i] = face_index_min;
            for (int k = 0; k < 3; k++) weight_map[3 * i + k] = weight_min[k];
            if (return_depth && face_inv_map) {
                const float *finv = faces_inv + ((long)bn * nf + face_index_min) * 9;
                for (int k = 0; k < 9; k++) face_inv_map[9 * i + k] = finv[k];
            }
        }
    }
}

ORC_API void orc_face_index_map(const float *faces, const float *faces_inv, int32_t *face_index_map,
                                float *weight_map, float *depth_map, float *face_inv_map,
                                int batch_size, int num_faces, int image_size, float near, float far,
                                int return_depth) {
    face_index_map_range(faces, faces_inv, face_index_map, weight_map, depth_map, face_inv_map, num_faces,
                         image_size, near, far, return_depth, 0, (long)batch_size * image_size * image_size);
}

/* The same brute force over the flat pixel range [pixel_begin, pixel_end) only: a bounded sample of a full-size
 * workload for bench.py's cpu_baseline (the full loop is O(pixels * faces): ~50 s per 512x512 view and thread). */
ORC_API void orc_face_index_map_range(const float *faces, const float *faces_inv, int32_t *face_index_map,
                                      float *weight_map, float *depth_map, float *face_inv_map,
                                      int batch_size, int num_faces, int image_size, float near, float far,
                                      int return_depth, long pixel_begin, long pixel_end) {
    const long npix = (long)batch_size * image_size * image_size;
    if (pixel_begin < 0) pixel_begin = 0;
    if (pixel_end > npix) pixel_end = npix;
    face_index_map_range(faces, faces_inv, face_index_map, weight_map, depth_map, face_inv_map, num_faces,
                         image_size, near, far, return_depth, pixel_begin, pixel_end);
}

/* ---------------------------------------------------------------------------
 * K2, per-face bounding-box variant: same per-(pixel, face) arithmetic, but
 * each face only visits the pixels of its (conservatively dilated) bounding
 * box, the loop structure of the reference's native CPU rasterizer
 * (deep3dmap/core/renderer/renderer_demo/mesh_cython/render.cpp:333-366).
 * The winner per pixel is the lexicographic minimum of (zp, face index),
 * which is exactly what KCU:142's strict '<' in ascending face order selects.
 * Used as the "cpu_tiled" baseline; checked against the brute-force form in
 * tests/test_oracle.py.  Parallel over image row bands (no write conflicts).
 * ------------------------------------------------------------------------- */
ORC_API void orc_face_index_map_bbox(const float *faces, const float *faces_inv, int32_t *face_index_map,
                                     float *weight_map, float *depth_map, float *face_inv_map,
                                     int batch_size, int num_faces, int image_size, float near, float far,
                                     int return_depth) {
    const int is = image_size, nf = num_faces;
    const int band = 16;
    const int nbands = (is + band - 1) / band;
    float *centers = (float *)malloc(sizeof(float) * (size_t)is);
    for (int i = 0; i < is; i++) centers[i] = pixel_center(i, is);
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int bn = 0; bn < batch_size; bn++) {
        for (int bd = 0; bd < nbands; bd++) {
            const int ylo = bd * band, yhi = imin(is, ylo + band) - 1;
            for (int fn = 0; fn < nf; fn++) {
                const float *face = faces + ((long)bn * nf + fn) * 9;
                if (is_backside(face)) continue;
                const float *finv = faces_inv + ((long)bn * nf + fn) * 9;
                /* conservative pixel bbox: NDC -> pixel is p = (v*is + is - 1)/2; dilate by a
                 * rounding margin so that no pixel passing KCU:115-117 is skipped. */
                float xmn = fminf(face[0], fminf(face[3], face[6])), xmx = fmaxf(face[0], fmaxf(face[3], face[6]));
                float ymn = fminf(face[1], fminf(face[4], face[7])), ymx = fmaxf(face[1], fmaxf(face[4], face[7]));
                int x0 = 0, x1 = is - 1, y0 = ylo, y1 = yhi;
                if (xmn == xmn && xmx == xmx && ymn == ymn && ymx == ymx && isfinite(xmn) && isfinite(xmx) &&
                    isfinite(ymn) && isfinite(ymx) && !(face[0] != face[0]) && !(face[3] != face[3]) &&
                    !(face[6] != face[6]) && !(face[1] != face[1]) && !(face[4] != face[4]) && !(face[7] != face[7])) {
                    double m = 4e-6 * (fmax(fmax(fabs(xmn), fabs(xmx)), fmax(fabs(ymn), fabs(ymx))) + 1.0);
                    x0 = imax(x0, cuda_d2i(ceil(((xmn - m) * is + is - 1) * 0.5)));
                    x1 = imin(x1, cuda_d2i(floor(((xmx + m) * is + is - 1) * 0.5)));
                    y0 = imax(y0, cuda_d2i(ceil(((ymn - m) * is + is - 1) * 0.5)));
                    y1 = imin(y1, cuda_d2i(floor(((ymx + m) * is + is - 1) * 0.5)));
                }
                for (int yi = y0; yi <= y1; yi++) {
                    for (int xi = x0; xi <= x1; xi++) {
                        float w[3], zp;
                        if (!pixel_face_eval(face, finv, xi, yi, centers[xi], centers[yi], near, far, w, &zp))
                            continue;
                        const long i = ((long)bn * is + yi) * is + xi;
                        const int cur = face_index_map[i];
                        /* depth_map is pre-filled with far, so 'zp < depth' also covers the empty pixel */
                        if (zp < depth_map[i] || (cur >= 0 && zp == depth_map[i] && fn < cur)) {
                            depth_map[i] = zp;
                            face_index_map[i] = fn;
                            for (int k = 0; k < 3; k++) weight_map[3 * i + k] = w[k];
                            if (return_depth && face_inv_map)
                                for (int k = 0; k < 9; k++) face_inv_map[9 * i + k] = finv[k];
                        }
                    }
                }
            }
        }
    }
    free(centers);
}

/* ---------------------------------------------------------------------------
 * K3: trilinear sampling of the winning face's ts^3 texture cube.
 * Follows forward_texture_sampling_cuda_kernel, KCU:172-242.
 * ------------------------------------------------------------------------- */
ORC_API void orc_texture_sampling(const float *faces, const float *textures, const int32_t *face_index_map,
                                  const float *weight_map, const float *depth_map, float *rgb_map,
                                  int32_t *sampling_index_map, float *sampling_weight_map, int batch_size,
                                  int num_faces, int image_size, int texture_size, float eps) {
    const int is = image_size, nf = num_faces, ts = texture_size;
    const long npix = (long)batch_size * is * is;
    /* ts == 1 makes KCU:229-233 index texels 1..3 of a one-texel cube, i.e. the next faces' texels;
     * that in-buffer bleed is reproduced, reads past the end of the whole buffer yield 0. */
    const long tex_total = (long)batch_size * nf * ts * ts * ts * 3;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < npix; i++) {
        const int face_index = face_index_map[i];
        if (face_index < 0) continue;                                                       /* KCU:192 */
        const int bn = (int)(i / ((long)is * is));
        const float *face = faces + ((long)bn * nf + face_index) * 9;
        const float *texture = textures + ((long)bn * nf + face_index) * ts * ts * ts * 3;
        const float *weight = weight_map + i * 3;
        const float depth = depth_map[i];
        float tif[3];
        for (int k = 0; k < 3; k++) {
            float t = weight[k] * (float)(ts - 1) * (depth / face[3 * k + 2]);              /* KCU:211 */
            t = (float)fmax((double)t, 0.);                                                 /* KCU:212 */
            t = fminf(t, (float)(ts - 1) - eps);                                            /* KCU:213 */
            tif[k] = t;
        }
        float new_pixel[3] = {0, 0, 0};
        for (int pn = 0; pn < 8; pn++) {
            float w = 1;
            int tii[3];
            for (int k = 0; k < 3; k++) {
                const int fl = cuda_d2i((double)tif[k]);
                if (((pn >> k) % 2) == 0) {
                    w *= 1 - (tif[k] - (float)fl);
                    tii[k] = fl;
                } else {
                    w *= tif[k] - (float)fl;
                    tii[k] = fl + 1;
                }
            }                                                                               /* KCU:222-231 */
            const int isc = tii[0] * ts * ts + tii[1] * ts + tii[2];
            for (int k = 0; k < 3; k++) {
                const long ti = (texture - textures) + (long)isc * 3 + k;
                new_pixel[k] += w * (ti < tex_total ? textures[ti] : 0.0f);                 /* KCU:235 */
            }
            sampling_index_map[i * 8 + pn] = isc;
            sampling_weight_map[i * 8 + pn] = w;
        }
        for (int k = 0; k < 3; k++) rgb_map[i * 3 + k] = new_pixel[k];
    }
}

/* ---------------------------------------------------------------------------
 * K4: Kato's approximate gradient of rgb/alpha wrt the x,y of each face's
 * vertices.  Follows backward_pixel_map_cuda_kernel, KCU:245-503, one face at
 * a time.  grad_faces entries of front-facing faces are OVERWRITTEN
 * (KCU:501-502); culled faces are left untouched.
 * ------------------------------------------------------------------------- */
ORC_API void orc_backward_pixel_map(const float *faces, const int32_t *face_index_map, const float *rgb_map,
                                    const float *alpha_map, const float *grad_rgb_map,
                                    const float *grad_alpha_map, float *grad_faces, int batch_size,
                                    int num_faces, int image_size, float eps, int return_rgb,
                                    int return_alpha) {
    const int is = image_size;
    const long n = (long)batch_size * num_faces;
#pragma omp parallel for schedule(dynamic, 16)
    for (long i = 0; i < n; i++) {
        const int bn = (int)(i / num_faces);
        const int fn = (int)(i % num_faces);
        const float *face = faces + i * 9;
        float grad_face[9] = {0};
        if (is_backside(face)) continue;                                                    /* KCU:270 */
        const long base = (long)bn * is * is;

        for (int edge_num = 0; edge_num < 3; edge_num++) {
            int pi[3];
            float pp[3][2];
            for (int num = 0; num < 3; num++) pi[num] = (edge_num + num) % 3;
            for (int num = 0; num < 3; num++)
                for (int dim = 0; dim < 2; dim++)
                    pp[num][dim] = (float)(0.5 * (double)(face[3 * pi[num] + dim] * (float)is + (float)is - 1.0f)); /* KCU:282 */

            for (int axis = 0; axis < 2; axis++) {
                float p[3][2];
                for (int num = 0; num < 3; num++)
                    for (int dim = 0; dim < 2; dim++) p[num][dim] = pp[num][(dim + axis) % 2];

                int direction;
                if (axis == 0) direction = (p[0][0] < p[1][0]) ? -1 : 1;
                else           direction = (p[0][0] < p[1][0]) ? 1 : -1;                     /* KCU:297-308 */

                const int d0_from = cuda_d2i(fmax((double)ceilf(fminf(p[0][0], p[1][0])), 0.));       /* KCU:312 */
                const int d0_to = cuda_d2i(fmin((double)fmaxf(p[0][0], p[1][0]), is - 1.));            /* KCU:313 */
                for (int d0 = d0_from; d0 <= d0_to; d0++) {
                    const float d1_cross =
                        (p[1][1] - p[0][1]) / (p[1][0] - p[0][0]) * ((float)d0 - p[0][0]) + p[0][1];   /* KCU:317 */
                    int d1_in, d1_out;
                    if (0 < direction) d1_in = cuda_d2i((double)floorf(d1_cross));
                    else               d1_in = cuda_d2i((double)ceilf(d1_cross));
                    d1_out = (int)((unsigned)d1_in + (unsigned)direction);                  /* wraps like CUDA */
                    if (d1_in < 0 || is <= d1_in) continue;
                    if (d1_out < 0 || is <= d1_out) continue;                               /* KCU:325-328 */

                    long map_index_in, map_index_out;
                    if (axis == 0) {
                        map_index_in = base + (long)d1_in * is + d0;
                        map_index_out = base + (long)d1_out * is + d0;
                    } else {
                        map_index_in = base + (long)d0 * is + d1_in;
                        map_index_out = base + (long)d0 * is + d1_out;
                    }
                    float alpha_in = 0, alpha_out = 0;
                    const float *rgb_in = NULL, *rgb_out = NULL;
                    if (return_alpha) { alpha_in = alpha_map[map_index_in]; alpha_out = alpha_map[map_index_out]; }
                    if (return_rgb) { rgb_in = rgb_map + map_index_in * 3; rgb_out = rgb_map + map_index_out * 3; }
                    const long map_offset = (axis == 0) ? is : 1;

                    /* out: from the out-pixel to the image border (KCU:354-414) */
                    if (face_index_map[map_index_in] == fn) {
                        const int d1_limit = (0 < direction) ? is - 1 : 0;
                        const int d1_from = imax(imin(d1_out, d1_limit), 0);
                        const int d1_to = imin(imax(d1_out, d1_limit), is - 1);
                        long idx = (axis == 0) ? base + (long)d1_from * is + d0 : base + (long)d0 * is + d1_from;
                        for (int d1 = d1_from; d1 <= d1_to; d1++, idx += map_offset) {
                            float diff_grad = 0;
                            if (return_alpha) diff_grad += (alpha_map[idx] - alpha_in) * grad_alpha_map[idx];
                            if (return_rgb)
                                for (int k = 0; k < 3; k++)
                                    diff_grad += (rgb_map[idx * 3 + k] - rgb_in[k]) * grad_rgb_map[idx * 3 + k];
                            if (diff_grad <= 0) continue;                                   /* KCU:401 */
                            if (p[1][0] != (float)d0) {
                                float dist = (float)((double)((p[1][0] - p[0][0]) / (p[1][0] - (float)d0) *
                                                              ((float)d1 - d1_cross)) * 2. / is);      /* KCU:404 */
                                dist = (0 < dist) ? dist + eps : dist - eps;
                                grad_face[pi[0] * 3 + (1 - axis)] -= diff_grad / dist;
                            }
                            if (p[0][0] != (float)d0) {
                                float dist = (float)((double)((p[1][0] - p[0][0]) / ((float)d0 - p[0][0]) *
                                                              ((float)d1 - d1_cross)) * 2. / is);      /* KCU:409 */
                                dist = (0 < dist) ? dist + eps : dist - eps;
                                grad_face[pi[1] * 3 + (1 - axis)] -= diff_grad / dist;
                            }
                        }
                    }

                    /* in: from the in-pixel to the opposite edge (KCU:417-495) */
                    {
                        float d0_cross2;
                        if (((float)d0 - p[0][0]) * ((float)d0 - p[2][0]) < 0)
                            d0_cross2 = (p[2][1] - p[0][1]) / (p[2][0] - p[0][0]) * ((float)d0 - p[0][0]) + p[0][1];
                        else
                            d0_cross2 = (p[1][1] - p[2][1]) / (p[1][0] - p[2][0]) * ((float)d0 - p[2][0]) + p[2][1];
                        int d1_limit;
                        if (0 < direction) d1_limit = cuda_d2i((double)ceilf(d0_cross2));
                        else               d1_limit = cuda_d2i((double)floorf(d0_cross2));
                        const int d1_from = imax(imin(d1_in, d1_limit), 0);
                        const int d1_to = imin(imax(d1_in, d1_limit), is - 1);
                        long idx = (axis == 0) ? base + (long)d1_from * is + d0 : base + (long)d0 * is + d1_from;
                        for (int d1 = d1_from; d1 <= d1_to; d1++, idx += map_offset) {
                            if (face_index_map[idx] != fn) continue;                        /* KCU:470 */
                            float diff_grad = 0;
                            if (return_alpha) diff_grad += (alpha_map[idx] - alpha_out) * grad_alpha_map[idx];
                            if (return_rgb)
                                for (int k = 0; k < 3; k++)
                                    diff_grad += (rgb_map[idx * 3 + k] - rgb_out[k]) * grad_rgb_map[idx * 3 + k];
                            if (diff_grad <= 0) continue;                                   /* KCU:481 */
                            if (p[1][0] != (float)d0) {
                                float dist = (float)((double)((p[1][0] - p[0][0]) / (p[1][0] - (float)d0) *
                                                              ((float)d1 - d1_cross)) * 2. / is);      /* KCU:485 */
                                dist = (0 < dist) ? dist + eps : dist - eps;
                                grad_face[pi[0] * 3 + (1 - axis)] -= diff_grad / dist;
                            }
                            if (p[0][0] != (float)d0) {
                                float dist = (float)((double)((p[1][0] - p[0][0]) / ((float)d0 - p[0][0]) *
                                                              ((float)d1 - d1_cross)) * 2. / is);      /* KCU:490 */
                                dist = (0 < dist) ? dist + eps : dist - eps;
                                grad_face[pi[1] * 3 + (1 - axis)] -= diff_grad / dist;
                            }
                        }
                    }
                }
            }
        }
        for (int k = 0; k < 9; k++) grad_faces[i * 9 + k] = grad_face[k];                   /* KCU:501-502 */
    }
}

/* ---------------------------------------------------------------------------
 * K5: scatter-add of w * grad_rgb into the 8 sampled texels.
 * Follows backward_textures_cuda_kernel, KCU:506-540.  Serial accumulation in
 * pixel order (the reference's atomics have no defined order).
 * ------------------------------------------------------------------------- */
ORC_API void orc_backward_textures(const int32_t *face_index_map, const float *sampling_weight_map,
                                   const int32_t *sampling_index_map, const float *grad_rgb_map,
                                   float *grad_textures, int batch_size, int num_faces, int image_size,
                                   int texture_size) {
    const int is = image_size, nf = num_faces, ts = texture_size;
    const long tex_total = (long)batch_size * nf * ts * ts * ts * 3;   /* see orc_texture_sampling on ts == 1 */
    /* serial over the batch as well: with ts == 1 a face's bleed can cross into the next batch entry */
    for (int bn = 0; bn < batch_size; bn++) {
        for (long pn = 0; pn < (long)is * is; pn++) {
            const long i = (long)bn * is * is + pn;
            const int face_index = face_index_map[i];
            if (face_index < 0) continue;
            float *grad_texture = grad_textures + ((long)bn * nf + face_index) * ts * ts * ts * 3;
            for (int s = 0; s < 8; s++) {
                const float w = sampling_weight_map[i * 8 + s];
                const int isc = sampling_index_map[i * 8 + s];
                for (int k = 0; k < 3; k++) {
                    const long ti = (grad_texture - grad_textures) + (long)isc * 3 + k;
                    if (ti < tex_total) grad_textures[ti] += w * grad_rgb_map[i * 3 + k];               /* KCU:537 */
                }
            }
        }
    }
}

/* ---------------------------------------------------------------------------
 * K6: gradient of the perspective-correct depth wrt the covering face.
 * Follows backward_depth_map_cuda_kernel, KCU:543-592.  ADDS into grad_faces
 * (after K4 has overwritten, rasterize.py:141-151).
 * ------------------------------------------------------------------------- */
ORC_API void orc_backward_depth_map(const float *faces, const float *depth_map, const int32_t *face_index_map,
                                    const float *face_inv_map, const float *weight_map,
                                    const float *grad_depth_map, float *grad_faces, int batch_size,
                                    int num_faces, int image_size) {
    const int is = image_size, nf = num_faces;
#pragma omp parallel for schedule(static)
    for (int bn = 0; bn < batch_size; bn++) {
        for (long pn = 0; pn < (long)is * is; pn++) {
            const long i = (long)bn * is * is + pn;
            const int fn = face_index_map[i];
            if (fn < 0) continue;
            const float *face = faces + ((long)bn * nf + fn) * 9;
            const float depth = depth_map[i];
            const float depth2 = depth * depth;
            const float *face_inv = face_inv_map + i * 9;
            const float *weight = weight_map + i * 3;
            const float grad_depth = grad_depth_map[i];
            float *grad_face = grad_faces + ((long)bn * nf + fn) * 9;
            for (int k = 0; k < 3; k++) {
                const float z_k = face[3 * k + 2];
                grad_face[3 * k + 2] += grad_depth * weight[k] * depth2 / (z_k * z_k);       /* KCU:575 */
            }
            float tmp[3] = {0, 0, 0};
            for (int k = 0; k < 3; k++)
                for (int l = 0; l < 3; l++) tmp[k] += -face_inv[3 * l + k] / face[3 * l + 2];           /* KCU:582 */
            for (int k = 0; k < 3; k++)
                for (int l = 0; l < 2; l++)
                    grad_face[3 * k + l] += -grad_depth * tmp[l] * weight[k] * depth2 * (float)is / 2.0f; /* KCU:588 */
        }
    }
}

/* ===========================================================================
 * Texture asset kernels.  LTK = NR/cuda/load_textures_cuda_kernel.cu,
 * CTK = NR/cuda/create_texture_image_cuda_kernel.cu.
 * ========================================================================= */

/* LTK:6-14 */
static inline float tex_mod(float x, float y) { return x > 0 ? fmodf(x, y) : y + fmodf(x, y); }

/* load_textures_cuda_kernel, LTK:23-114: fills the ts^3 texture cube of every face whose is_update flag is set by
 * sampling `image` [H,W,3] at barycentric combinations of the face's uv coordinates `faces` [F,3,2].
 * The reference writes the wrapped uv back into the shared `faces` array from every texel thread: a race that is
 * harmless except at integer coordinates under REPEAT (0 -> 1 -> 0 ...), where its result depends on scheduling.
 * The port wraps a private copy ONCE, i.e. what a thread that sees the caller's input computes. */
ORC_API void orc_load_textures(const float *image, const int32_t *is_update, const float *faces, float *textures,
                               int num_faces, int texture_size, int image_height, int image_width,
                               int texture_wrapping, int use_bilinear) {
    const int ts = texture_size;
    const long n = (long)num_faces * ts * ts * ts;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; i++) {
        const int fn = (int)(i / ((long)ts * ts * ts));
        float dim0 = (float)(((i / (ts * ts)) % ts) / (ts - 1.));
        float dim1 = (float)(((i / ts) % ts) / (ts - 1.));
        float dim2 = (float)((i % ts) / (ts - 1.));
        if (0 < dim0 + dim1 + dim2) {
            const float sum = dim0 + dim1 + dim2;
            dim0 /= sum; dim1 /= sum; dim2 /= sum;
        }
        if (is_update[fn] == 0) continue;
        float face[6];
        for (int k = 0; k < 6; k++) face[k] = faces[fn * 6 + k];
        if (texture_wrapping == 0) {                                           /* REPEAT */
            for (int k = 0; k < 6; k++) face[k] = tex_mod(face[k], 1.0f);
        } else if (texture_wrapping == 1) {                                    /* MIRRORED_REPEAT */
            for (int k = 0; k < 6; k++)
                face[k] = (tex_mod(face[k], 2.0f) < 1) ? tex_mod(face[k], 1.0f) : 1 - tex_mod(face[k], 1.0f);
        } else if (texture_wrapping == 2) {                                    /* CLAMP_TO_EDGE */
            for (int k = 0; k < 6; k++) face[k] = fmaxf(fminf(face[k], 1.0f), 0.0f);
        }
        const float pos_x = (face[0] * dim0 + face[2] * dim1 + face[4] * dim2) * (float)(image_width - 1);
        const float pos_y = (face[1] * dim0 + face[3] * dim1 + face[5] * dim2) * (float)(image_height - 1);
        float *tex = textures + i * 3;
        if (texture_wrapping == 3) {                                           /* CLAMP_TO_BORDER: zeros (LTK:97,109) */
            tex[0] = tex[1] = tex[2] = 0;
            continue;
        }
        if (use_bilinear) {
            const int xi = cuda_d2i((double)pos_x), yi = cuda_d2i((double)pos_y);
            const float wx1 = pos_x - (float)xi, wx0 = 1 - wx1, wy1 = pos_y - (float)yi, wy0 = 1 - wy1;
            const int y1 = imin(cuda_d2i((double)(pos_y + 1)), image_height - 1), x1 = imin(xi + 1, image_width - 1);
            for (int k = 0; k < 3; k++) {
                float c = 0;
                c += image[((long)yi * image_width + xi) * 3 + k] * (wx0 * wy0);
                c += image[((long)y1 * image_width + xi) * 3 + k] * (wx0 * wy1);
                c += image[((long)yi * image_width + x1) * 3 + k] * (wx1 * wy0);
                c += image[((long)y1 * image_width + x1) * 3 + k] * (wx1 * wy1);
                tex[k] = c;
            }
        } else {
            const int xi = cuda_d2i((double)roundf(pos_x)), yi = cuda_d2i((double)roundf(pos_y));
            for (int k = 0; k < 3; k++) tex[k] = image[((long)yi * image_width + xi) * 3 + k];
        }
    }
}

/* create_texture_image_cuda_kernel + _boundary_, CTK:10-115: renders every face's texture cube into its tile of a
 * texture atlas `image` [tile_h*tso, tile_w*tso, 3]; vertices_all [F,3,2] are the tile-space triangle corners. */
ORC_API void orc_create_texture_image(const float *vertices_all, const float *textures, float *image, int num_faces,
                                      int texture_size_in, int image_height, int image_width, int tile_width,
                                      float eps) {
    const int tsi = texture_size_in, tso = image_width / tile_width;
    const long npx = (long)image_height * image_width;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < npx; i++) {
        const int x = (int)(i % ((long)tile_width * tso)), y = (int)(i / ((long)tile_width * tso));
        const int fn = x / tso + (y / tso) * tile_width;
        /* The atlas has tile_width*tile_height >= num_faces tiles; for the padding tiles the reference reads
         * past the end of `textures` and `vertices_all` (undefined values).  Defined here: padding tiles keep the
         * zeros the image was created with (NR/save_obj.py:16). */
        if (fn >= num_faces) { for (int k = 0; k < 3; k++) image[i * 3 + k] = 0.f; continue; }
        const float *texture = textures + (long)fn * tsi * tsi * tsi * 3;
        const float *p0 = vertices_all + (long)fn * 6, *p1 = p0 + 2, *p2 = p0 + 4;
        float face_inv[9] = {
            p1[1] - p2[1], p2[0] - p1[0], p1[0] * p2[1] - p2[0] * p1[1],
            p2[1] - p0[1], p0[0] - p2[0], p2[0] * p0[1] - p0[0] * p2[1],
            p0[1] - p1[1], p1[0] - p0[0], p0[0] * p1[1] - p1[0] * p0[1]};
        const float den = p2[0] * (p0[1] - p1[1]) + p0[0] * (p1[1] - p2[1]) + p1[0] * (p2[1] - p0[1]);
        for (int k = 0; k < 9; k++) face_inv[k] /= den;
        float weight[3], weight_sum = 0;
        for (int k = 0; k < 3; k++) {
            weight[k] = face_inv[3 * k + 0] * (float)x + face_inv[3 * k + 1] * (float)y + face_inv[3 * k + 2];
            weight_sum += weight[k];
        }
        for (int k = 0; k < 3; k++) weight[k] /= (weight_sum + eps);
        float tif[3];
        for (int k = 0; k < 3; k++) {
            float t = weight[k] * (float)(tsi - 1);
            t = (float)fmax((double)t, 0.);
            t = fminf(t, (float)(tsi - 1) - eps);
            tif[k] = t;
        }
        float px[3] = {0, 0, 0};
        for (int pn = 0; pn < 8; pn++) {
            float w = 1;
            int tii[3];
            for (int k = 0; k < 3; k++) {
                const int fl = cuda_d2i((double)tif[k]);
                if (((pn >> k) % 2) == 0) { w *= 1 - (tif[k] - (float)fl); tii[k] = fl; }
                else                      { w *= tif[k] - (float)fl;       tii[k] = fl + 1; }
            }
            const int isc = tii[0] * tsi * tsi + tii[1] * tsi + tii[2];
            /* texture_size_in == 1 indexes one cube past this face in the reference; stay inside the array */
            const int in_range = (long)fn * tsi * tsi * tsi + isc < (long)num_faces * tsi * tsi * tsi;
            for (int k = 0; k < 3; k++) px[k] += w * (in_range ? texture[isc * 3 + k] : 0.0f);
        }
        for (int k = 0; k < 3; k++) image[i * 3 + k] = px[k];
    }
    /* boundary fix-up, CTK:97-115 (a second launch in the reference: reads the finished image) */
    for (long i = 0; i < npx; i++) {
        const int x = (int)(i % ((long)tile_width * tso)), y = (int)(i / ((long)tile_width * tso));
        if ((y % tso + 1) == (x % tso))
            for (int k = 0; k < 3; k++) image[i * 3 + k] = image[((long)y * tile_width * tso + (x - 1)) * 3 + k];
    }
}

/* ===========================================================================
 * Camera transforms and NrRenderer's depth -> mesh vertices, in plain f32 with
 * every operation rounded once (-ffp-contract=off), written out operation by
 * operation.  The reference evaluates these formulas through torch (matmul,
 * broadcasting), i.e. in whatever association the tensor library picks; any
 * fixed f32 association is an equally valid restatement.  This one uses the
 * same association as the HIP kernels (csrc/d3m_aux.h camera_point /
 * k_camera_basis / k_grid_warp), so that the oracle and the product hand
 * BIT-IDENTICAL screen-space vertices to their rasterizers: coverage then
 * matches pixel for pixel and gradients can be compared at 1e-3 instead of
 * the few per cent that flipped edge pixels cost.  Checked against the
 * reference modules' own outputs in tests/test_oracle.py (cam/..., d3m/...).
 *   look_at / look     NR/look_at.py:47-60, NR/look.py:39-51
 *   perspective        NR/perspective.py:15-20
 *   projection         NR/projection.py:19-42
 *   depth -> vertices  deep3dmap/core/renderer/renderer_nr.py:64-80,95-100
 * ========================================================================= */
static void orc_normalize3(float *v) {                    /* F.normalize(eps=1e-5): v / max(|v|, 1e-5) */
    const float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float d = fmaxf(n, 1e-5f);
    v[0] /= d; v[1] /= d; v[2] /= d;
}
static void orc_cross3(const float *a, const float *b, float *o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static const float *orc_bptr(const float *p, int nb, int b, int stride) { return p + (size_t)(nb > 1 ? b : 0) * stride; }

/* rows (x, y, z) of the camera frame: z = normalize(at - eye) or normalize(direction) */
ORC_API void orc_camera_basis(const float *eye, int eye_b, const float *at_or_dir, int at_b, const float *up, int up_b,
                              int is_look_at, float *rot, int B) {
    for (int b = 0; b < B; b++) {
        const float *e = orc_bptr(eye, eye_b, b, 3), *a = orc_bptr(at_or_dir, at_b, b, 3), *u = orc_bptr(up, up_b, b, 3);
        float z[3], x[3], y[3];
        for (int k = 0; k < 3; k++) z[k] = is_look_at ? a[k] - e[k] : a[k];
        orc_normalize3(z);
        orc_cross3(u, z, x);
        orc_normalize3(x);
        orc_cross3(z, x, y);
        orc_normalize3(y);
        for (int k = 0; k < 3; k++) { rot[b * 9 + k] = x[k]; rot[b * 9 + 3 + k] = y[k]; rot[b * 9 + 6 + k] = z[k]; }
    }
}

/* mode 1: look_at / look (rot = frame, eye_or_t = eye, optional perspective division by z and `width`);
 * mode 3: projection (rot = R, eye_or_t = t, K, dist, orig).  vertices [vb,V,3] -> out [B,V,3]. */
ORC_API void orc_camera_points(const float *vertices, int vb, int mode, int perspective, float width, float orig,
                               const float *rot, int rot_b, const float *eye_or_t, int eye_b, const float *K, int K_b,
                               const float *dist, int dist_b, float *out, int B, int V) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)B * V; i++) {
        const int b = (int)(i / V), vi = (int)(i % V);
        const float *v = vertices + ((size_t)(vb > 1 ? b : 0) * V + vi) * 3;
        const float *r = orc_bptr(rot, rot_b, b, 9), *e = orc_bptr(eye_or_t, eye_b, b, 3);
        float *o = out + 3 * i;
        if (mode == 1) {
            const float d0 = v[0] - e[0], d1 = v[1] - e[1], d2 = v[2] - e[2];
            float x = d0 * r[0] + d1 * r[1] + d2 * r[2];
            float y = d0 * r[3] + d1 * r[4] + d2 * r[5];
            const float z = d0 * r[6] + d1 * r[7] + d2 * r[8];
            if (perspective) { x = x / z / width; y = y / z / width; }
            o[0] = x; o[1] = y; o[2] = z;
        } else {
            const float *Kb = orc_bptr(K, K_b, b, 9), *dc = orc_bptr(dist, dist_b, b, 5);
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
            const float u = x__ * Kb[0] + y__ * Kb[1] + Kb[2];
            float vv = x__ * Kb[3] + y__ * Kb[4] + Kb[5];
            vv = orig - vv;
            o[0] = 2 * (u - orig / 2.f) / orig;
            o[1] = 2 * (vv - orig / 2.f) / orig;
            o[2] = cz;
        }
    }
}

/* vertices[b, y*W+x] = rot_b (depth * inv_K (x, y, 1) - c) + c + trans_b, c = (0, 0, center_z) */
ORC_API void orc_depth_to_vertices(const float *depth, const float *inv_K, int invK_b, const float *rot, const float *trans,
                                   float center_z, float *out, int B, int H, int W) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)B * H * W; i++) {
        const int b = (int)(i / ((long)H * W)), pix = (int)(i % ((long)H * W));
        const float x = (float)(pix % W), y = (float)(pix / W);
        const float *iK = orc_bptr(inv_K, invK_b, b, 9);
        const float d = depth[i];
        float p[3];
        for (int k = 0; k < 3; k++) p[k] = (x * iK[3 * k] + y * iK[3 * k + 1] + iK[3 * k + 2]) * d;
        p[2] -= center_z;
        const float *R = rot + (size_t)b * 9, *t = trans + (size_t)b * 3;
        out[3 * i + 0] = p[0] * R[0] + p[1] * R[1] + p[2] * R[2] + t[0];
        out[3 * i + 1] = p[0] * R[3] + p[1] * R[4] + p[2] * R[5] + t[1];
        out[3 * i + 2] = p[0] * R[6] + p[1] * R[7] + p[2] * R[8] + center_z + t[2];
    }
}
