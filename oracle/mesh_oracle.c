/* oracle/mesh_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C (f64) restatement of the reference's face3d CPU rasterizer family,
 * MC = deep3dmap/core/renderer/renderer_demo/mesh_cython/render.cpp.  Each function cites the MC lines it follows.
 * Array layouts are the reference's: vertices [3, nver] and triangles [3, ntri] coordinate-major, images [h, w, c].
 * Pinned against the reference itself: `make -C oracle ref` compiles MC from its own sources into
 * oracle/_ref/libmesh_ref.so; tests/test_mesh_oracle.py compares the two bit for bit and against the committed
 * vectors (tests/golden/mesh_golden.npz).  Where the reference indexes a texture without a bounds check, this port
 * (and the HIP kernels) clamp the index: identical on the in-range inputs the callers produce. */
#include <math.h>
#include <stdint.h>
#define ORC_API __attribute__((visibility("default")))

typedef struct { double x, y; } pt;
static inline pt pt_sub(pt a, pt b) { pt r = {a.x - b.x, a.y - b.y}; return r; }
static inline double pt_dot(pt a, pt b) { return a.x * b.x + a.y * b.y; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* MC:372-403: true within 2 pixels of the image border (a quirk of the reference), barycentric test otherwise */
static int point_in_tri(pt p, pt p0, pt p1, pt p2, int h, int w) {
    if (p.x < 2 || p.x > w - 3 || p.y < 2 || p.y > h - 3) return 1;
    const pt v0 = pt_sub(p2, p0), v1 = pt_sub(p1, p0), v2 = pt_sub(p, p0);
    const double dot00 = pt_dot(v0, v0), dot01 = pt_dot(v0, v1), dot02 = pt_dot(v0, v2), dot11 = pt_dot(v1, v1),
                 dot12 = pt_dot(v1, v2);
    double inv;
    if (dot00 * dot11 - dot01 * dot01 == 0) inv = 0;
    else inv = 1 / (dot00 * dot11 - dot01 * dot01);
    const double u = (dot11 * dot02 - dot01 * dot12) * inv, v = (dot00 * dot12 - dot01 * dot02) * inv;
    return (u >= 0) && (v >= 0) && (u + v < 1);
}

/* MC:406-434 */
static void point_weight(double* weight, pt p, pt p0, pt p1, pt p2) {
    const pt v0 = pt_sub(p2, p0), v1 = pt_sub(p1, p0), v2 = pt_sub(p, p0);
    const double dot00 = pt_dot(v0, v0), dot01 = pt_dot(v0, v1), dot02 = pt_dot(v0, v2), dot11 = pt_dot(v1, v1),
                 dot12 = pt_dot(v1, v2);
    double inv;
    if (dot00 * dot11 - dot01 * dot01 == 0) inv = 0;
    else inv = 1 / (dot00 * dot11 - dot01 * dot01);
    const double u = (dot11 * dot02 - dot01 * dot12) * inv, v = (dot00 * dot12 - dot01 * dot02) * inv;
    weight[0] = 1 - u - v;
    weight[1] = v;
    weight[2] = u;
}

/* triangle i: corner points and its pixel box (MC:58-72 and the same lines in every core) */
static int tri_box(const double* vertices, const int* triangles, int nver, int ntri, int i, int h, int w, pt* p0, pt* p1,
                   pt* p2, int* x_min, int* x_max, int* y_min, int* y_max) {
    const int a = triangles[i], b = triangles[ntri + i], c = triangles[2 * ntri + i];
    p0->x = vertices[a]; p0->y = vertices[nver + a];
    p1->x = vertices[b]; p1->y = vertices[nver + b];
    p2->x = vertices[c]; p2->y = vertices[nver + c];
    *x_min = imax((int)ceil(fmin(p0->x, fmin(p1->x, p2->x))), 0);
    *x_max = imin((int)floor(fmax(p0->x, fmax(p1->x, p2->x))), w - 1);
    *y_min = imax((int)ceil(fmin(p0->y, fmin(p1->y, p2->y))), 0);
    *y_max = imin((int)floor(fmax(p0->y, fmax(p1->y, p2->y))), h - 1);
    return !(*x_max < *x_min || *y_max < *y_min);
}

/* MC:4-24 */
ORC_API void orc_mesh_get_norm_direction(double* norm, const double* tri_norm, const int* triangles, int nver, int ntri) {
    for (int i = 0; i < ntri; i++) {
        const int a = triangles[i], b = triangles[ntri + i], c = triangles[2 * ntri + i];
        for (int j = 0; j < 3; j++) {
            norm[j * nver + a] = norm[j * nver + a] + tri_norm[j * ntri + i];
            norm[j * nver + b] = norm[j * nver + b] + tri_norm[j * ntri + i];
            norm[j * nver + c] = norm[j * nver + c] + tri_norm[j * ntri + i];
        }
    }
}

/* MC:27-91: painter with a z-buffer of per-triangle depths; LARGER depth is closer, strict > keeps the first of equals */
ORC_API void orc_mesh_render_colors(double* image, const double* vertices, const int* triangles, const double* tri_depth,
                                    const double* tri_tex, double* depth_buffer, int nver, int ntri, int h, int w, int c) {
    for (int i = 0; i < ntri; i++) {
        pt p0, p1, p2, p;
        int x_min, x_max, y_min, y_max;
        if (!tri_box(vertices, triangles, nver, ntri, i, h, w, &p0, &p1, &p2, &x_min, &x_max, &y_min, &y_max)) continue;
        for (int y = y_min; y <= y_max; y++)
            for (int x = x_min; x <= x_max; x++) {
                p.x = x; p.y = y;
                if (tri_depth[i] > depth_buffer[y * w + x] && point_in_tri(p, p0, p1, p2, h, w)) {
                    depth_buffer[y * w + x] = tri_depth[i];
                    for (int k = 0; k < c; k++) image[(y * w + x) * c + k] = tri_tex[k * ntri + i];
                }
            }
    }
}

/* texel fetch shared by MC:160-175 and :238-248 (nearest / bilinear), indices clamped (see the header) */
static double fetch_texel(const double* tex, int th, int tw, int tc, double ty, double tx, int k, int bilinear) {
    if (!bilinear) {
        const int yi = clampi((int)round(ty), 0, th - 1), xi = clampi((int)round(tx), 0, tw - 1);
        return tex[(yi * tw + xi) * tc + k];
    }
    const double yd = ty - floor(ty), xd = tx - floor(tx);
    const int y0 = clampi((int)floor(ty), 0, th - 1), y1 = clampi((int)ceil(ty), 0, th - 1);
    const int x0 = clampi((int)floor(tx), 0, tw - 1), x1 = clampi((int)ceil(tx), 0, tw - 1);
    const double ul = tex[(y0 * tw + x0) * tc + k], ur = tex[(y0 * tw + x1) * tc + k];
    const double dl = tex[(y1 * tw + x0) * tc + k], dr = tex[(y1 * tw + x1) * tc + k];
    return ul * (1 - xd) * (1 - yd) + ur * xd * (1 - yd) + dl * (1 - xd) * yd + dr * xd * yd;
}

/* MC:94-185.  The y of a texture corner is read with the MESH vertex index (MC:131-133), as in the reference. */
ORC_API void orc_mesh_render_texture(double* image, const double* vertices, const int* triangles, const double* texture,
                                     const double* tex_coords, const int* tex_triangles, const double* tri_depth,
                                     double* depth_buffer, int nver, int tex_nver, int ntri, int h, int w, int c, int tex_h,
                                     int tex_w, int tex_c, int mapping_type) {
    for (int i = 0; i < ntri; i++) {
        pt p0, p1, p2, p;
        int x_min, x_max, y_min, y_max;
        const int a = triangles[i], b = triangles[ntri + i], cc = triangles[2 * ntri + i];
        const int ta = tex_triangles[i], tb = tex_triangles[ntri + i], tcn = tex_triangles[2 * ntri + i];
        const pt t0 = {tex_coords[ta], tex_coords[tex_nver + a]}, t1 = {tex_coords[tb], tex_coords[tex_nver + b]},
                 t2 = {tex_coords[tcn], tex_coords[tex_nver + cc]};
        if (!tri_box(vertices, triangles, nver, ntri, i, h, w, &p0, &p1, &p2, &x_min, &x_max, &y_min, &y_max)) continue;
        for (int y = y_min; y <= y_max; y++)
            for (int x = x_min; x <= x_max; x++) {
                p.x = x; p.y = y;
                if (tri_depth[i] > depth_buffer[y * w + x] && point_in_tri(p, p0, p1, p2, h, w)) {
                    double weight[3];
                    point_weight(weight, p, p0, p1, p2);
                    /* point*double then sums, MC:154 */
                    const double tx = (weight[0] * t0.x + weight[1] * t1.x) + weight[2] * t2.x;
                    const double ty = (weight[0] * t0.y + weight[1] * t1.y) + weight[2] * t2.y;
                    for (int k = 0; k < c; k++)
                        image[(y * w + x) * c + k] = fetch_texel(texture, tex_h, tex_w, tex_c, ty, tx, k, mapping_type != 0);
                    depth_buffer[y * w + x] = tri_depth[i];
                }
            }
    }
}

/* MC:188-250 */
ORC_API void orc_mesh_map_texture(double* dst_image, const double* src_image, const double* dst_vertices,
                                  const double* src_vertices, const int* dst_triangle_buffer, const int* triangles,
                                  int nver, int ntri, int sh, int sw, int sc, int h, int w, int c) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int t = dst_triangle_buffer[y * w + x];
            if (t < 0) continue;
            const int a = triangles[t], b = triangles[ntri + t], cc = triangles[2 * ntri + t];
            pt p = {x, y};
            pt p0 = {dst_vertices[a], dst_vertices[nver + a]}, p1 = {dst_vertices[b], dst_vertices[nver + b]},
               p2 = {dst_vertices[cc], dst_vertices[nver + cc]};
            double weight[3];
            point_weight(weight, p, p0, p1, p2);
            const pt s0 = {src_vertices[a], src_vertices[nver + a]}, s1 = {src_vertices[b], src_vertices[nver + b]},
                     s2 = {src_vertices[cc], src_vertices[nver + cc]};
            const double tx = (weight[0] * s0.x + weight[1] * s1.x) + weight[2] * s2.x;
            const double ty = (weight[0] * s0.y + weight[1] * s1.y) + weight[2] * s2.y;
            if (tx < 0 || tx > sw - 1 || ty < 0 || ty > sh - 1) continue;
            for (int k = 0; k < c; k++) dst_image[(y * w + x) * c + k] = fetch_texel(src_image, sh, sw, sc, ty, tx, k, 1);
        }
}

/* MC:253-318.  x, y, z of the vertex loop are INT variables in the reference (MC:260): coordinates are truncated. */
ORC_API void orc_mesh_vis_of_vertices(double* vis, const double* vertices, const int* triangles, const double* tri_depth,
                                      double* depth_buffer, double* depth_tmp, int nver, int ntri, int h, int w, int c) {
    (void)c;
    for (int i = 0; i < ntri; i++) {
        pt p0, p1, p2, p;
        int x_min, x_max, y_min, y_max;
        if (!tri_box(vertices, triangles, nver, ntri, i, h, w, &p0, &p1, &p2, &x_min, &x_max, &y_min, &y_max)) continue;
        for (int y = y_min; y <= y_max; y++)
            for (int x = x_min; x <= x_max; x++) {
                p.x = x; p.y = y;
                if (tri_depth[i] > depth_buffer[y * w + x] && point_in_tri(p, p0, p1, p2, h, w))
                    depth_buffer[y * w + x] = tri_depth[i];
            }
    }
    for (int i = 0; i < nver; i++) {
        const int x = (int)vertices[i], y = (int)vertices[nver + i];
        if (x < 0 || x > w - 1 || y < 0 || y > h - 1) continue;
        const int z = (int)vertices[nver * 2 + i];
        if (z < depth_tmp[y * w + x]) continue;
        if (fabs(z - depth_buffer[y * w + x]) < 1.5) {
            vis[i] = 1;
            depth_tmp[y * w + x] = z;
        }
    }
}

/* MC:321-365 */
ORC_API void orc_mesh_get_triangle_buffer(int* triangle_buffer, const double* vertices, const int* triangles,
                                          const double* tri_depth, double* depth_buffer, int nver, int ntri, int h, int w,
                                          int c) {
    (void)c;
    for (int i = 0; i < ntri; i++) {
        pt p0, p1, p2, p;
        int x_min, x_max, y_min, y_max;
        if (!tri_box(vertices, triangles, nver, ntri, i, h, w, &p0, &p1, &p2, &x_min, &x_max, &y_min, &y_max)) continue;
        for (int y = y_min; y <= y_max; y++)
            for (int x = x_min; x <= x_max; x++) {
                p.x = x; p.y = y;
                if (tri_depth[i] > depth_buffer[y * w + x] && point_in_tri(p, p0, p1, p2, h, w)) {
                    depth_buffer[y * w + x] = tri_depth[i];
                    triangle_buffer[y * w + x] = i;
                }
            }
    }
}

/* MC:441-488 */
ORC_API void orc_mesh_get_correspondence(const double* image, const double* pncc_code, double* uv, int nver, int h, int w,
                                         int c) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const double r = image[(y * w + x) * c], g = image[(y * w + x) * c + 1], b = image[(y * w + x) * c + 2];
            const double sum = r + g + b;
            if (sum < 0.07) continue;
            double min_dis = h + w;
            int min_ind = 0;
            for (int i = 0; i < nver; i++) {
                const double dr = r - pncc_code[i], dg = g - pncc_code[nver + i], db = b - pncc_code[2 * nver + i];
                const double dis = dr * dr + dg * dg + db * db;
                if (dis < min_dis) { min_dis = dis; min_ind = i; }
            }
            if (min_dis > 0.08) continue;
            uv[min_ind] = x;
            uv[nver + min_ind] = y;
        }
}
