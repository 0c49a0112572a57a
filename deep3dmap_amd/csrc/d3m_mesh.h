// d3m_mesh.h -- the face3d utility rasterizer family on the GPU (f64, like the reference).
// Replaces MC = deep3dmap/core/renderer/renderer_demo/mesh_cython/render.cpp (cores called from render_cython.pyx:55-161
// by render.py:124-299; caller tools/data_gen/prnet.py:110).  Layouts are the reference's: vertices [3, nver] and
// triangles [3, ntri] coordinate-major, images [h, w, c].
//
// The reference paints triangles one after the other into a z-buffer of PER-TRIANGLE depths ("larger is closer",
// strict >), so the pixel's owner is the covering triangle with the largest depth and, among equals, the lowest index,
// provided that depth beats the buffer's initial value.  That is order-free:
//   1. k_mesh_depth   one lane per triangle walks its pixel box: atomicMax of the order-preserving u64 image of the
//                     depth (boxes over 256 pixels are walked by a whole wave);
//   2. k_mesh_owner   same walk: where the triangle's depth equals the pixel's maximum, atomicMin of its index;
//   3. a per-pixel epilogue that does what the reference does when it paints the owner (colour, texture lookup,
//      triangle index ...) -- identical f64 expressions, -ffp-contract=off, so images are bit-identical.
#pragma once
#include "d3m_launch.h"

namespace d3m {

struct pt2 {
    double x, y;
};
__device__ __forceinline__ pt2 pt_sub(pt2 a, pt2 b) { return pt2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ double pt_dot(pt2 a, pt2 b) { return a.x * b.x + a.y * b.y; }

// MC:372-403: true within 2 pixels of the image border (a quirk of the reference), barycentric test otherwise
__device__ __forceinline__ bool mesh_point_in_tri(pt2 p, pt2 p0, pt2 p1, pt2 p2, int h, int w) {
    if (p.x < 2 || p.x > w - 3 || p.y < 2 || p.y > h - 3) return true;
    const pt2 v0 = pt_sub(p2, p0), v1 = pt_sub(p1, p0), v2 = pt_sub(p, p0);
    const double dot00 = pt_dot(v0, v0), dot01 = pt_dot(v0, v1), dot02 = pt_dot(v0, v2), dot11 = pt_dot(v1, v1),
                 dot12 = pt_dot(v1, v2);
    double inv;
    if (dot00 * dot11 - dot01 * dot01 == 0) inv = 0;
    else inv = 1 / (dot00 * dot11 - dot01 * dot01);
    const double u = (dot11 * dot02 - dot01 * dot12) * inv, v = (dot00 * dot12 - dot01 * dot02) * inv;
    return (u >= 0) && (v >= 0) && (u + v < 1);
}

// MC:406-434
__device__ __forceinline__ void mesh_point_weight(double* weight, pt2 p, pt2 p0, pt2 p1, pt2 p2) {
    const pt2 v0 = pt_sub(p2, p0), v1 = pt_sub(p1, p0), v2 = pt_sub(p, p0);
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

// order-preserving u64 image of a double (no NaN)
__device__ __forceinline__ unsigned long long ordered_u64(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

struct MeshTris {
    const double* vertices;    // [3, nver]
    const int32_t* triangles;  // [3, ntri]
    const double* tri_depth;   // [ntri]
    int nver, ntri, h, w;
    __device__ __forceinline__ void corners(int i, pt2& p0, pt2& p1, pt2& p2) const {
        const int a = triangles[i], b = triangles[ntri + i], c = triangles[2 * ntri + i];
        p0 = pt2{vertices[a], vertices[nver + a]};
        p1 = pt2{vertices[b], vertices[nver + b]};
        p2 = pt2{vertices[c], vertices[nver + c]};
    }
    // MC:64-72: the pixel box of a triangle; false when it is empty
    __device__ __forceinline__ bool box(pt2 p0, pt2 p1, pt2 p2, int& x0, int& x1, int& y0, int& y1) const {
        x0 = max((int)ceil(fmin(p0.x, fmin(p1.x, p2.x))), 0);
        x1 = min((int)floor(fmax(p0.x, fmax(p1.x, p2.x))), w - 1);
        y0 = max((int)ceil(fmin(p0.y, fmin(p1.y, p2.y))), 0);
        y1 = min((int)floor(fmax(p0.y, fmax(p1.y, p2.y))), h - 1);
        return !(x1 < x0 || y1 < y0);
    }
};

constexpr int MESH_BIG_BOX = 256;     // boxes with more pixels are walked by a whole wave,
constexpr int MESH_HUGE_BOX = 16384;  // and beyond this by the whole grid

// zkey[p] = ordered(depth_buffer[p]) (the buffer's initial contents), owner[p] = none
__global__ void __launch_bounds__(256) k_mesh_init(const double* __restrict__ depth_buffer, unsigned long long* __restrict__ zkey,
                                                  int* __restrict__ owner, int* __restrict__ big_count, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i == 0) { big_count[0] = 0; big_count[1] = 0; }
    if (i >= n) return;
    const double d = depth_buffer[i];
    zkey[i] = d == d ? ordered_u64(d) : ~0ull;          // a NaN in the buffer never lets `>` pass
    owner[i] = 0x7FFFFFFF;
}

// One walk over the box of triangle i with lanes [lane, lane + stride, ...] of its row-major pixel list.
// PASS 0: depth maximum.  PASS 1: owner = lowest index among the triangles that reach the maximum.
template <int PASS>
__device__ __forceinline__ void mesh_walk(const MeshTris& m, const double* __restrict__ depth_buffer,
                                          unsigned long long* __restrict__ zkey, int* __restrict__ owner, int i, pt2 p0, pt2 p1,
                                          pt2 p2, int x0, int x1, int y0, int y1, int lane, int stride) {
    const double d = m.tri_depth[i];
    if (!(d == d)) return;                               // NaN depth: `tri_depth > depth_buffer` is never true
    const unsigned long long key = ordered_u64(d);
    const int bw = x1 - x0 + 1, n = bw * (y1 - y0 + 1);
    for (int k = lane; k < n; k += stride) {
        const int x = x0 + k % bw, y = y0 + k / bw;
        const long p = (long)y * m.w + x;
        if (PASS == 0) {
            if (key <= zkey[p]) continue;                // cannot raise the maximum (it only grows)
        } else {
            if (key != zkey[p] || !(d > depth_buffer[p])) continue;
        }
        if (!mesh_point_in_tri(pt2{(double)x, (double)y}, p0, p1, p2, m.h, m.w)) continue;
        if (PASS == 0) atomicMax(&zkey[p], key);
        else atomicMin(&owner[p], i);
    }
}

template <int PASS>
__global__ void __launch_bounds__(256) k_mesh_tris(MeshTris m, const double* __restrict__ depth_buffer,
                                                  unsigned long long* __restrict__ zkey, int* __restrict__ owner,
                                                  int* __restrict__ big_list, int* __restrict__ big_count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m.ntri) return;
    pt2 p0, p1, p2;
    int x0, x1, y0, y1;
    m.corners(i, p0, p1, p2);
    if (!m.box(p0, p1, p2, x0, x1, y0, y1)) return;
    const long area = (long)(x1 - x0 + 1) * (y1 - y0 + 1);
    if (area > MESH_BIG_BOX) {      // both passes walk these with a wave each / the whole grid; the lists are built once
        if (PASS == 0) {
            if (area > MESH_HUGE_BOX) big_list[m.ntri - 1 - atomicAdd(big_count + 1, 1)] = i;    // huge: from the end
            else big_list[atomicAdd(big_count, 1)] = i;
        }
        return;
    }
    mesh_walk<PASS>(m, depth_buffer, zkey, owner, i, p0, p1, p2, x0, x1, y0, y1, 0, 1);
}

template <int PASS>
__global__ void __launch_bounds__(256) k_mesh_big_tris(MeshTris m, const double* __restrict__ depth_buffer,
                                                      unsigned long long* __restrict__ zkey, int* __restrict__ owner,
                                                      const int* __restrict__ big_list, const int* __restrict__ big_count) {
    const int n = big_count[0], n_huge = big_count[1];
    const int lane = threadIdx.x & 63;
    for (int j = blockIdx.x * 4 + (threadIdx.x >> 6); j < n; j += gridDim.x * 4) {
        const int i = big_list[j];
        pt2 p0, p1, p2;
        int x0, x1, y0, y1;
        m.corners(i, p0, p1, p2);
        m.box(p0, p1, p2, x0, x1, y0, y1);
        mesh_walk<PASS>(m, depth_buffer, zkey, owner, i, p0, p1, p2, x0, x1, y0, y1, lane, 64);
    }
    // image-sized triangles: every lane of the grid takes a share of each
    for (int j = 0; j < n_huge; j++) {
        const int i = big_list[m.ntri - 1 - j];
        pt2 p0, p1, p2;
        int x0, x1, y0, y1;
        m.corners(i, p0, p1, p2);
        m.box(p0, p1, p2, x0, x1, y0, y1);
        mesh_walk<PASS>(m, depth_buffer, zkey, owner, i, p0, p1, p2, x0, x1, y0, y1, blockIdx.x * 256 + threadIdx.x,
                        gridDim.x * 256);
    }
}

// texel fetch of MC:160-175 / :238-248 (indices clamped: the reference does not check them)
__device__ __forceinline__ double mesh_fetch_texel(const double* __restrict__ tex, int th, int tw, int tc, double ty,
                                                   double tx, int k, bool bilinear) {
    if (!bilinear) {
        const int yi = min(max((int)round(ty), 0), th - 1), xi = min(max((int)round(tx), 0), tw - 1);
        return tex[((long)yi * tw + xi) * tc + k];
    }
    const double yd = ty - floor(ty), xd = tx - floor(tx);
    const int y0 = min(max((int)floor(ty), 0), th - 1), y1 = min(max((int)ceil(ty), 0), th - 1);
    const int x0 = min(max((int)floor(tx), 0), tw - 1), x1 = min(max((int)ceil(tx), 0), tw - 1);
    const double ul = tex[((long)y0 * tw + x0) * tc + k], ur = tex[((long)y0 * tw + x1) * tc + k];
    const double dl = tex[((long)y1 * tw + x0) * tc + k], dr = tex[((long)y1 * tw + x1) * tc + k];
    return ul * (1 - xd) * (1 - yd) + ur * xd * (1 - yd) + dl * (1 - xd) * yd + dr * xd * yd;
}

// ---- per-pixel epilogues --------------------------------------------------------------------------------------
enum { MESH_COLORS = 0, MESH_TEXTURE = 1, MESH_TRIANGLE_BUFFER = 2, MESH_DEPTH_ONLY = 3 };

struct MeshShade {
    double* image;                 // [h, w, c]
    const double* tri_tex;         // [c, ntri]                          (colours)
    const double* texture;         // [tex_h, tex_w, tex_c]              (texture)
    const double* tex_coords;      // [2, tex_nver]
    const int32_t* tex_triangles;  // [3, ntri]
    int32_t* triangle_buffer;      // [h, w]                             (triangle buffer)
    int c, tex_nver, tex_h, tex_w, tex_c, bilinear;
};

template <int MODE>
__global__ void __launch_bounds__(256) k_mesh_shade(MeshTris m, MeshShade s, const int* __restrict__ owner,
                                                   double* __restrict__ depth_buffer) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)m.h * m.w) return;
    const int i = owner[p];
    if (i == 0x7FFFFFFF) return;                          // nothing beat the buffer here: pixel untouched
    depth_buffer[p] = m.tri_depth[i];
    if (MODE == MESH_COLORS) {
        for (int k = 0; k < s.c; k++) s.image[p * s.c + k] = s.tri_tex[(long)k * m.ntri + i];      // MC:82-85
    } else if (MODE == MESH_TRIANGLE_BUFFER) {
        s.triangle_buffer[p] = i;                                                               // MC:360
    } else if (MODE == MESH_TEXTURE) {
        pt2 p0, p1, p2;
        m.corners(i, p0, p1, p2);
        const int a = m.triangles[i], b = m.triangles[m.ntri + i], cc = m.triangles[2 * m.ntri + i];
        const int ta = s.tex_triangles[i], tb = s.tex_triangles[m.ntri + i], tcn = s.tex_triangles[2 * m.ntri + i];
        // MC:131-133: the y of a texture corner is read with the MESH vertex index, as in the reference
        const pt2 t0 = pt2{s.tex_coords[ta], s.tex_coords[s.tex_nver + a]}, t1 = pt2{s.tex_coords[tb], s.tex_coords[s.tex_nver + b]},
                  t2 = pt2{s.tex_coords[tcn], s.tex_coords[s.tex_nver + cc]};
        const int x = (int)(p % m.w), y = (int)(p / m.w);
        double weight[3];
        mesh_point_weight(weight, pt2{(double)x, (double)y}, p0, p1, p2);
        const double tx = (weight[0] * t0.x + weight[1] * t1.x) + weight[2] * t2.x;             // MC:154
        const double ty = (weight[0] * t0.y + weight[1] * t1.y) + weight[2] * t2.y;
        for (int k = 0; k < s.c; k++)
            s.image[p * s.c + k] = mesh_fetch_texel(s.texture, s.tex_h, s.tex_w, s.tex_c, ty, tx, k, s.bilinear != 0);
    }
}

// MC:188-250: one lane per destination pixel
__global__ void __launch_bounds__(256) k_mesh_map_texture(double* __restrict__ dst_image, const double* __restrict__ src_image,
                                                         const double* __restrict__ dst_vertices,
                                                         const double* __restrict__ src_vertices,
                                                         const int32_t* __restrict__ dst_triangle_buffer,
                                                         const int32_t* __restrict__ triangles, int nver, int ntri, int sh,
                                                         int sw, int sc, int h, int w, int c) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)h * w) return;
    const int t = dst_triangle_buffer[p];
    if (t < 0 || t >= ntri) return;
    const int a = triangles[t], b = triangles[ntri + t], cc = triangles[2 * ntri + t];
    const int x = (int)(p % w), y = (int)(p / w);
    double weight[3];
    mesh_point_weight(weight, pt2{(double)x, (double)y}, pt2{dst_vertices[a], dst_vertices[nver + a]},
                      pt2{dst_vertices[b], dst_vertices[nver + b]}, pt2{dst_vertices[cc], dst_vertices[nver + cc]});
    const double tx = (weight[0] * src_vertices[a] + weight[1] * src_vertices[b]) + weight[2] * src_vertices[cc];
    const double ty = (weight[0] * src_vertices[nver + a] + weight[1] * src_vertices[nver + b]) + weight[2] * src_vertices[nver + cc];
    if (tx < 0 || tx > sw - 1 || ty < 0 || ty > sh - 1) return;
    for (int k = 0; k < c; k++) dst_image[p * c + k] = mesh_fetch_texel(src_image, sh, sw, sc, ty, tx, k, true);
}

// ---- vertex visibility (MC:296-317) ------------------------------------------------------------------------------
// The reference visits the vertices in index order with a per-pixel running maximum depth_tmp: a vertex is visible
// iff it is a CANDIDATE -- inside the image, |z - depth_buffer| < 1.5, z >= the initial depth_tmp of its pixel -- and
// no earlier candidate of the same pixel has a larger z (x, y, z are ints in the reference, MC:260: truncation).
// Candidates are chained per pixel (atomicExch), then every candidate scans its pixel's short chain.
__device__ __forceinline__ bool mesh_vertex_pixel(const double* __restrict__ vertices, int nver, int i, int h, int w, int& pix,
                                                  int& z) {
    const int x = (int)vertices[i], y = (int)vertices[nver + i];
    if (x < 0 || x > w - 1 || y < 0 || y > h - 1) return false;
    pix = y * w + x;
    z = (int)vertices[2 * nver + i];
    return true;
}

__global__ void __launch_bounds__(256) k_mesh_vis_chain(const double* __restrict__ vertices, const double* __restrict__ depth_buffer,
                                                       const double* __restrict__ depth_tmp, int* __restrict__ head,
                                                       int* __restrict__ next, int nver, int h, int w) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nver) return;
    int pix, z;
    next[i] = -2;                                         // not a candidate
    if (!mesh_vertex_pixel(vertices, nver, i, h, w, pix, z)) return;
    if ((double)z < depth_tmp[pix] || !(fabs((double)z - depth_buffer[pix]) < 1.5)) return;
    next[i] = atomicExch(&head[pix], i);
}

__global__ void __launch_bounds__(256) k_mesh_vis_resolve(const double* __restrict__ vertices, const int* __restrict__ head,
                                                         const int* __restrict__ next, double* __restrict__ vis,
                                                         unsigned long long* __restrict__ tmp_key, int nver, int h, int w) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nver || next[i] == -2) return;
    int pix, z;
    mesh_vertex_pixel(vertices, nver, i, h, w, pix, z);
    bool beaten = false;
    for (int j = head[pix]; j >= 0; j = next[j]) {
        if (j < i && (int)vertices[2 * nver + j] > z) { beaten = true; break; }
    }
    if (beaten) return;
    vis[i] = 1;
    atomicMax(&tmp_key[pix], ordered_u64((double)z));     // depth_tmp ends as the largest accepted z
}

__global__ void __launch_bounds__(256) k_mesh_vis_finish(const unsigned long long* __restrict__ tmp_key,
                                                        double* __restrict__ depth_tmp, long n) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n || tmp_key[p] == 0ull) return;
    const unsigned long long k = tmp_key[p];
    const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    depth_tmp[p] = __longlong_as_double((long long)b);
}

// ---- vertex normals (MC:4-24) ---------------------------------------------------------------------------------------
// norm[:, v] += tri_norm[:, i] for the triangles i around v IN INDEX ORDER (f64 sums are order-sensitive): count the
// incidences, scan, fill (triangle*3 + corner), then one lane per vertex sorts its handful of entries and adds them.
__global__ void __launch_bounds__(256) k_mesh_incidence_count(const int32_t* __restrict__ triangles, int* __restrict__ count,
                                                             int ntri, int nver) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 3 * ntri) return;
    const int v = triangles[e];
    if (v >= 0 && v < nver) atomicAdd(&count[v], 1);
}

__global__ void __launch_bounds__(256) k_mesh_incidence_fill(const int32_t* __restrict__ triangles, const int* __restrict__ offset,
                                                            int* __restrict__ cursor, int* __restrict__ entries, int ntri,
                                                            int nver) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 3 * ntri) return;
    const int v = triangles[e];
    if (v < 0 || v >= nver) return;
    const int corner = e / ntri, tri = e % ntri;           // triangles is [3, ntri]
    entries[offset[v] + atomicAdd(&cursor[v], 1)] = tri * 3 + corner;
}

__global__ void __launch_bounds__(256) k_mesh_normals(double* __restrict__ norm, const double* __restrict__ tri_norm,
                                                     const int* __restrict__ offset, int* __restrict__ entries, int nver,
                                                     int ntri) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nver) return;
    const int lo = offset[v], hi = offset[v + 1];
    for (int a = lo + 1; a < hi; a++) {                      // insertion sort: valence is ~6
        const int key = entries[a];
        int b = a - 1;
        while (b >= lo && entries[b] > key) { entries[b + 1] = entries[b]; b--; }
        entries[b + 1] = key;
    }
    double n0 = norm[v], n1 = norm[nver + v], n2 = norm[2 * nver + v];
    for (int a = lo; a < hi; a++) {
        const int t = entries[a] / 3;
        n0 = n0 + tri_norm[t];
        n1 = n1 + tri_norm[ntri + t];
        n2 = n2 + tri_norm[2 * ntri + t];
    }
    norm[v] = n0; norm[nver + v] = n1; norm[2 * nver + v] = n2;
}

// ---- PNCC correspondence (MC:441-488) --------------------------------------------------------------------------------
// per pixel the first nearest code; the reference's raster-order loop leaves, per vertex, the LAST pixel that chose it
__global__ void __launch_bounds__(256) k_mesh_nearest_code(const double* __restrict__ image, const double* __restrict__ pncc,
                                                          int* __restrict__ last_pixel, int nver, int h, int w, int c) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= h * w) return;
    const double r = image[(long)p * c], g = image[(long)p * c + 1], b = image[(long)p * c + 2];
    const double sum = r + g + b;
    if (sum < 0.07) return;
    double min_dis = h + w;
    int min_ind = 0;
    for (int i = 0; i < nver; i++) {
        const double dr = r - pncc[i], dg = g - pncc[nver + i], db = b - pncc[2 * nver + i];
        const double dis = dr * dr + dg * dg + db * db;
        if (dis < min_dis) { min_dis = dis; min_ind = i; }
    }
    if (min_dis > 0.08) return;
    atomicMax(&last_pixel[min_ind], p);
}

__global__ void __launch_bounds__(256) k_mesh_write_uv(const int* __restrict__ last_pixel, double* __restrict__ uv, int nver,
                                                      int w) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nver || last_pixel[v] < 0) return;
    uv[v] = (double)(last_pixel[v] % w);
    uv[nver + v] = (double)(last_pixel[v] / w);
}

// ---- the numpy glue of render.py around the cores, with the same roundings ---------------------------------------
// out[k, i] = (values[k, a] + values[k, b] + values[k, c]) / 3.   (MP:141-142: tri_depth, tri_tex)
__global__ void __launch_bounds__(256) k_mesh_triangle_mean(const double* __restrict__ values, const int32_t* __restrict__ triangles,
                                                           double* __restrict__ out, int channels, int nver, int ntri) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)channels * ntri) return;
    const int k = (int)(e / ntri), i = (int)(e % ntri);
    const double* v = values + (long)k * nver;
    out[e] = (v[triangles[i]] + v[triangles[ntri + i]] + v[triangles[2 * ntri + i]]) / 3.;
}

// tri_norm[:, i] = cross(pt0 - pt1, pt0 - pt2)   (MP:6-9; products and differences rounded separately, like numpy)
__global__ void __launch_bounds__(256) k_mesh_triangle_normals(const double* __restrict__ vertices,
                                                              const int32_t* __restrict__ triangles,
                                                              double* __restrict__ tri_norm, int nver, int ntri) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ntri) return;
    const int a = triangles[i], b = triangles[ntri + i], c = triangles[2 * ntri + i];
    double u[3], v[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double p0 = vertices[(long)k * nver + a];
        u[k] = p0 - vertices[(long)k * nver + b];
        v[k] = p0 - vertices[(long)k * nver + c];
    }
    tri_norm[i] = u[1] * v[2] - u[2] * v[1];
    tri_norm[ntri + i] = u[2] * v[0] - u[0] * v[2];
    tri_norm[2 * ntri + i] = u[0] * v[1] - u[1] * v[0];
}

// MP:19-26: unit length; an all-zero normal becomes (1, 0, 0)
__global__ void __launch_bounds__(256) k_mesh_normalize(double* __restrict__ norm, int nver) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nver) return;
    double x = norm[v];
    const double y = norm[nver + v], z = norm[2 * nver + v];
    double mag = x * x + y * y + z * z;
    if (mag == 0) { mag = 1; x = 1; }
    const double r = sqrt(mag);
    norm[v] = x / r;
    norm[nver + v] = y / r;
    norm[2 * nver + v] = z / r;
}

__global__ void __launch_bounds__(256) k_fill_i32(int* __restrict__ p, int value, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = value;
}

}  // namespace d3m
