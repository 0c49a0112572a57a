// d3m_front.h -- the first launch of a lit render step: camera (basis + transform), per-face light and EVERY clear of
// the step, as one kernel.
//
// Round 4's step opened with four launches that each finish in a few microseconds -- k_camera_basis (one lane per view),
// k_camera_forward, k_face_light, a fill -- and carried four more fills further in (the plan's counters, the backward's
// accumulators, the gathered pass's masks): with the step's kernels on one stream each is ~5 us of its critical path, more
// than the work it does.  Here they are block ranges of ONE grid: nothing in them depends on another, the operators behind
// are told that their scratch is already zero (D3M_PRECLEARED), and the step's first dependent kernel starts ~35 us earlier.
//
//   the first blocks (as many as the   zero fill of up to FRONT_RANGES ranges, strided
//    largest range needs, <= 2 048)
//   the next nb_cam                      camera_point per (view, vertex); the look_at / look basis is recomputed per block
//                                        from eye / at / up (~60 flops, the block's first lane) instead of being a launch of
//                                        its own, and the block that holds a view's vertex 0 stores it for the backward pass
//   the last nb_light                    face_light per (light batch entry, face)
#pragma once
#include "d3m_aux.h"
#include "d3m_lit.h"

namespace d3m {

constexpr int FRONT_RANGES = 10;
struct FrontBasis {             // look_at / look: the basis from the caller's vectors (eye == NULL: cam.rot holds it already)
    const float *eye, *at_or_dir, *up;
    int eye_b, at_b, up_b, is_look_at;
    float* rot_out;             // [B,3,3]: stored for the backward pass (cam.rot points here too)
};
struct FrontArgs {
    // camera
    const float* vertices; int vb; Cam cam; FrontBasis basis; float* screen; int B, V;
    // light (light == NULL: none)
    IndexedFaces faces; LightParams lp; float* light; int light_b;
    // clears
    uint32_t* z_ptr[FRONT_RANGES];
    unsigned long long z_words[FRONT_RANGES];
    unsigned nb_cam, nb_light;
};

__device__ __forceinline__ void front_basis(const FrontBasis& fb, int b, float* rot /*[9]*/) {
    const float* e = cam_ptr(fb.eye, fb.eye_b, b, 3);
    const float* a = cam_ptr(fb.at_or_dir, fb.at_b, b, 3);
    const float* u = cam_ptr(fb.up, fb.up_b, b, 3);
    float z[3], x[3], y[3];
    for (int k = 0; k < 3; k++) z[k] = fb.is_look_at ? a[k] - e[k] : a[k];     // the operations of k_camera_basis, in its order
    normalize3(z);
    cross3(u, z, x);
    normalize3(x);
    cross3(z, x, y);
    normalize3(y);
    for (int k = 0; k < 3; k++) { rot[k] = x[k]; rot[3 + k] = y[k]; rot[6 + k] = z[k]; }
}

// Block ranges: the clears FIRST -- their stores leave the CUs at once and drain while the camera blocks, which wait for
// their view's basis (a chain of dependent loads, three normalisations), occupy them: 26.3 -> 23.9 us at 32 views.
// (Camera blocks of 1 024 entries, four per thread with their loads issued together, to get through the chip in one round
// of blocks instead of three: 35.7 us.  Dropped.)
__global__ void __launch_bounds__(256) k_lit_front(FrontArgs a) {
    const unsigned nb_zero = gridDim.x - a.nb_cam - a.nb_light;
    if (blockIdx.x < nb_zero) {
        // as k_zero_ranges (d3m_launch.h), over this part's blocks
        const size_t stride = (size_t)nb_zero * 256, i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
#pragma unroll
        for (int k = 0; k < FRONT_RANGES; k++) {
            const size_t n_words = a.z_words[k];
            if (n_words == 0) continue;
            size_t head = ((16u - (unsigned)((uintptr_t)a.z_ptr[k] & 15u)) & 15u) >> 2;     // words up to the 16-byte boundary
            if (head > n_words) head = n_words;
            const size_t n4 = (n_words - head) >> 2;
            uint4* p4 = reinterpret_cast<uint4*>(a.z_ptr[k] + head);
            if (i0 < head) a.z_ptr[k][i0] = 0;
            for (size_t i = i0; i < n4; i += stride) p4[i] = make_uint4(0, 0, 0, 0);
            for (size_t i = head + (n4 << 2) + i0; i < n_words; i += stride) a.z_ptr[k][i] = 0;
        }
        return;
    }
    const unsigned bx = blockIdx.x - nb_zero;
    if (bx < a.nb_cam) {
        __shared__ float s_rot[2][9];
        const long i0 = (long)bx * 256, n = (long)a.B * a.V;
        const int b0 = (int)(i0 / a.V);
        Cam c = a.cam;
        if (a.basis.eye) {
            // a block's 256 consecutive (view, vertex) entries belong to one view or (V < 256: to several; then every lane
            // computes its own) two: lanes 0 and 1 compute the bases of views b0 and b0 + 1
            const bool two_at_most = a.V >= 256;
            if (two_at_most) {
                if (threadIdx.x < 2 && b0 + (int)threadIdx.x < a.B) front_basis(a.basis, b0 + threadIdx.x, s_rot[threadIdx.x]);
                __syncthreads();
            }
            const long i = i0 + threadIdx.x;
            if (i >= n) return;
            const int b = (int)(i / a.V), v = (int)(i - (long)b * a.V);
            float rot[9];
            if (two_at_most) {
#pragma unroll
                for (int k = 0; k < 9; k++) rot[k] = s_rot[b - b0][k];
            } else {
                front_basis(a.basis, b, rot);
            }
            if (v == 0) {
#pragma unroll
                for (int k = 0; k < 9; k++) a.basis.rot_out[(size_t)(a.cam.rot_b > 1 ? b : 0) * 9 + k] = rot[k];   // (one camera for all views: the same nine floats from every view)
            }
            c.rot = rot; c.rot_b = 1;               // camera_point reads the basis through the pointer: this lane's copy
            const float* p = a.vertices + ((size_t)(a.vb > 1 ? b : 0) * a.V + v) * 3;
            const float in[3] = {p[0], p[1], p[2]};
            float o[3];
            camera_point(c, b, in, o, nullptr);
            a.screen[i * 3 + 0] = o[0]; a.screen[i * 3 + 1] = o[1]; a.screen[i * 3 + 2] = o[2];
            return;
        }
        const long i = i0 + threadIdx.x;
        if (i >= n) return;
        const int b = (int)(i / a.V), v = (int)(i - (long)b * a.V);
        const float* p = a.vertices + ((size_t)(a.vb > 1 ? b : 0) * a.V + v) * 3;
        const float in[3] = {p[0], p[1], p[2]};
        float o[3];
        camera_point(c, b, in, o, nullptr);
        a.screen[i * 3 + 0] = o[0]; a.screen[i * 3 + 1] = o[1]; a.screen[i * 3 + 2] = o[2];
        return;
    }
    {
        const long i = (long)(bx - a.nb_cam) * 256 + threadIdx.x;
        const int Fp = a.faces.num_faces();
        if (i >= (long)a.light_b * Fp) return;
        float fc[9], l[3];
        a.faces.load((int)(i / Fp), (int)(i % Fp), fc);
        face_light(fc, a.lp, l, nullptr, nullptr, nullptr);
        a.light[3 * i + 0] = l[0]; a.light[3 * i + 1] = l[1]; a.light[3 * i + 2] = l[2];
        return;
    }
}

// ---- ... and its LAST launch: the camera's adjoint and the light's, both into the mesh's gradient ------------------------
// Round 4 ended a step with k_face_light_backward (float atomics into grad_vertices) and, behind it, k_camera_backward_add
// (one writer per entry, a plain read-modify-write): two launches of 11 and 13 us.  As block ranges of one grid both ADD with
// float atomics into a gradient that is zero when the launch starts (the step's first launch cleared it) -- a vertex gets the
// camera's sum over views from one lane and the light's contributions from its ~6 faces, in any order.
//   blocks [0, nb_cam)       k_camera_backward's body (eight lanes per vertex of a shared mesh, DPP sum over the views)
//   the rest                 k_face_light_backward's body
struct BackArgs {
    const float* vertices; int vb; Cam cam; const float* grad_screen; float* grad_vertices; int B, V;
    IndexedFaces faces; LightParams lp; const float* grad_light; int light_b;
    unsigned nb_cam;
};
__global__ void __launch_bounds__(256) k_lit_back(BackArgs a) {
    if (blockIdx.x < a.nb_cam) {
        const long t = (long)blockIdx.x * 256 + threadIdx.x;
        const bool shared = a.vb <= 1;
        const long i = shared ? t >> 3 : t;
        const int sub = shared ? (int)(t & 7) : 0;
        const long n = (long)(shared ? 1 : a.B) * a.V;
        const bool on = i < n;
        const int v = on ? (int)(i % a.V) : 0;
        const int b_lo = shared ? sub : (int)(i / a.V), b_hi = shared ? a.B : b_lo + 1, b_step = shared ? 8 : 1;
        float acc[3] = {0, 0, 0};
        if (on) {
            const float* p = a.vertices + (size_t)i * 3;
            const float in[3] = {p[0], p[1], p[2]};
            for (int b = b_lo; b < b_hi; b += b_step) {
                const float* gp = a.grad_screen + ((size_t)b * a.V + v) * 3;
                const float g[3] = {gp[0], gp[1], gp[2]};
                float gv[3];
                camera_point_adjoint(a.cam, b, in, g, gv);
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
        if (on && sub == 0) {
            atomicAdd(&a.grad_vertices[i * 3 + 0], acc[0]);
            atomicAdd(&a.grad_vertices[i * 3 + 1], acc[1]);
            atomicAdd(&a.grad_vertices[i * 3 + 2], acc[2]);
        }
        return;
    }
    // the light's adjoint: k_face_light_backward's body
    const long i = (long)(blockIdx.x - a.nb_cam) * 256 + threadIdx.x;
    const IndexedFaces& fs = a.faces;
    const LightParams& lp = a.lp;
    const int Fp = fs.num_faces();
    if (i >= (long)a.light_b * Fp || lp.id == 0) return;
    const float gl[3] = {a.grad_light[3 * i], a.grad_light[3 * i + 1], a.grad_light[3 * i + 2]};
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
    const float ea[3] = {fc[0] - fc[3], fc[1] - fc[4], fc[2] - fc[5]};
    const float eb[3] = {fc[6] - fc[3], fc[7] - fc[4], fc[8] - fc[5]};
    float ga[3], gb[3];
    cross3(eb, gc, ga);
    cross3(gc, ea, gb);
    int ids[3];
    fs.vertex_ids(b, f, ids);
    float* base = a.grad_vertices + (size_t)(a.vb > 1 ? b : 0) * fs.V * 3;
    for (int k = 0; k < 3; k++) {
        atomicAdd(&base[(size_t)ids[0] * 3 + k], ga[k]);
        atomicAdd(&base[(size_t)ids[2] * 3 + k], gb[k]);
        atomicAdd(&base[(size_t)ids[1] * 3 + k], -(ga[k] + gb[k]));
    }
}

}  // namespace d3m
