/*
 * d3m_raster.h -- C ABI of libd3m_raster.so, the MI355X (gfx950) differentiable mesh rasterizer.
 *
 * This is the drop-in boundary for the rasterization hot path of achao2013/deep3dmap.  Plain
 * pointers and sizes only: every pointer is a DEVICE pointer unless it says "host"; `stream` is a
 * hipStream_t passed as void* (NULL = the legacy default stream, which is what the reference's
 * `<<<blocks, threads>>>` launches use, rasterize_cuda_kernel.cu:615).  No ATen / torch types.
 *
 * Conventions kept from the reference extension (pnpmodules/neural_renderer/neural_renderer/cuda/
 * rasterize_cuda.cpp, "KCPP" below; kernels in rasterize_cuda_kernel.cu, "KCU"):
 *   - the caller allocates AND pre-fills every output (face_index -1, weight 0, depth = far, rgb 0,
 *     sampling maps 0, face_inv 0, grad buffers 0: neural_renderer/rasterize.py:50-69,111-115);
 *     entry points mutate in place;
 *   - tensors are contiguous f32 / i32 in the layouts named per argument;
 *   - disabled outputs may be NULL (the reference passes 1-element dummies, rasterize.py:46,59-69);
 *   - nothing here synchronises the device; errors detected on the host are returned at once,
 *     launch failures surface as D3M_ERR_LAUNCH with hipGetLastError() kept in d3m_last_hip_error().
 *
 * Return value of every int function: 0 on success, one of the D3M_ERR_* codes otherwise.
 */
#ifndef D3M_RASTER_H
#define D3M_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* d3m_stream_t; /* hipStream_t */

enum {
    D3M_OK = 0,
    D3M_ERR_INVALID = 1,   /* bad size / NULL where data is required / unsupported value */
    D3M_ERR_WORKSPACE = 2, /* workspace missing or smaller than d3m_*_workspace_bytes() */
    D3M_ERR_LAUNCH = 3     /* a HIP call failed; see d3m_last_hip_error() */
};

/* camera_mode values of d3m_camera_* (neural_renderer/renderer.py:88-112) */
enum { D3M_CAMERA_NONE = 0, D3M_CAMERA_LOOK_AT = 1, D3M_CAMERA_LOOK = 2, D3M_CAMERA_PROJECTION = 3 };

const char* d3m_version(void);
int d3m_last_hip_error(void);         /* hipError_t of the most recent failed HIP call, 0 if none */
/* Zero-fill up to six device ranges with ONE kernel launch (host arrays of `count` device pointers / byte counts; every
 * range 4-byte aligned and a multiple of 4 bytes, NULL / 0 entries skipped).  The library never uses memset nodes (a
 * captured hipMemsetAsync was not ordered reliably against the next kernel node on ROCm 7.2); callers that clear several
 * small scratch buffers per step (the reference does the same with torch.zeros, NR/rasterize.py:111-115) save a launch
 * per buffer. */
int d3m_zero_ranges(void* const* ptrs, const size_t* bytes, int count, d3m_stream_t stream);
/* CLEARS.  An operator that needs zeroed scratch (counters, arrival tickets, accumulators) clears it itself with one fill
 * launch in front of its kernels.  A caller that runs a whole step may do all of a step's clears in ONE launch instead --
 * d3m_lit_front, the step's first kernel, takes a list of ranges beside the camera and the light -- : it asks every operator
 * for what that operator would clear (d3m_forward_clear_bytes, d3m_edge_plan_clear_bytes, d3m_render_fit_scratch_clear_range,
 * d3m_backward_textures_lit_clear_ranges), zeroes those ranges, and passes D3M_PRECLEARED in the operator's `flags`. */
#define D3M_PRECLEARED 1
const char* d3m_error_string(int code);

/* Per-kernel timing with HIP events recorded on each launch's own stream (used by bench.py for the
 * live roofline figure; off by default).  d3m_timing_collect synchronises the device, returns one row
 * per kernel name (static strings) with its launch count and summed duration, and clears the record. */
void d3m_timing_enable(int on);
int d3m_timing_collect(const char** names, int* counts, float* total_ms, int max_entries);

/* ------------------------------------------------------------------------------------------------
 * A. The five operators of `neural_renderer.cuda.rasterize` (KCPP:193-199), same argument order.
 * ---------------------------------------------------------------------------------------------- */

/* Bytes of scratch d3m_forward_face_index_map(_mesh) needs for (batch_size, num_faces, image_size):
 * per-tile face lists built on the device -- or, for the meshes and batch sizes where d3m_forward_face_index_map_mesh
 * covers by bidding instead of binning, a 64-bit z-buffer and a list of big faces: the larger of the two.
 * A smaller buffer is accepted down to d3m_forward_workspace_min_bytes(); it lowers the size above which a face is handled
 * as "large" (scanned by every tile of its view) and rules the bidding form out.  Contents need no initialisation. */
size_t d3m_forward_workspace_bytes(int batch_size, int num_faces, int image_size);
size_t d3m_forward_workspace_min_bytes(int batch_size, int num_faces, int image_size);

/* Which of the two forms of coverage d3m_forward_face_index_map(_mesh) runs (both replace KCU:70-169 and
 * produce the same maps bit for bit): -1 = chosen per launch by mesh density and batch size (the default), 0 = per-tile
 * face lists only (k_bin_* -> k_raster_tiles), 1 = bidding into a 64-bit z-buffer wherever its workspace fits
 * (k_bid_faces -> k_bid_big -> k_bid_resolve).  Process-wide, read at every launch; the environment variable D3M_BID=0/1
 * sets the initial value.  Returns D3M_ERR_INVALID for any other value. */
int d3m_set_coverage_form(int form);
int d3m_get_coverage_form(void);

/* Run-to-run reproducibility (no reference counterpart: the reference's float atomics, KCU:533-538,586-589, are unordered
 * too).  on = 1: d3m_visibility (and the list a forward launch leaves) is built in ASCENDING face order by three launches
 * -- count, scan, compact -- instead of one whose chunks land in arrival order, so that every pass over the list issues its
 * float atomics from the same workgroups, in the same list order, in every run; the gathered passes of
 * d3m_backward_textures_lit and d3m_backward_depth_map take faces of ANY size (no per-pixel atomic fallback for boxes beyond
 * 4096 pixels), so that a face's sums are its own lanes'.  Process-wide, read at every launch; the
 * environment variable D3M_DETERMINISTIC=1 sets the initial value.  Returns D3M_ERR_INVALID for values other than 0 / 1. */
int d3m_set_deterministic(int on);
int d3m_get_deterministic(void);
/* The per-vertex sums of the deterministic mode (deep3dmap_amd/neural_renderer/rasterize.py, "DETERMINISTIC"): the adjoint
 * of vertices_to_faces + fill_back (NR/vertices_to_faces.py:16-22, NR/renderer.py:86) GATHERED per vertex in a fixed order
 * instead of scattered with float atomics.  adj_offsets [V+1], adj_items [3 F]: CSR adjacency of ONE index tensor tri [F,3]
 * -- item = 3 f + c for "corner c of triangle f", the items of a vertex in ascending order.
 * d3m_vertex_gather: grad_vertices [B,V,3] (WRITTEN) = the sum over a vertex's items of grad_faces_a[b,f,c,:] +
 * grad_faces_b[b,f,c,:] (either array may be NULL; both [B,F',3,3]) and, with fill_back, of the copy's [b,F+f,2-c,:];
 * `visibility` (optional): the d3m_visibility blob of the forward result -- faces that own no pixel are skipped unread.
 * d3m_face_light_backward_gather: d3m_face_light_backward for ONE shared mesh (vertices [V,3], grad_light [F',3]), the same
 * per-face terms gathered per vertex: grad_vertices [V,3] is WRITTEN. */
int d3m_vertex_gather(const float* grad_faces_a, const float* grad_faces_b, const int32_t* adj_offsets,
                      const int32_t* adj_items, float* grad_vertices, int batch_size, int num_vertices, int num_tri,
                      int fill_back, const void* visibility, d3m_stream_t stream);
int d3m_face_light_backward_gather(const float* vertices, const int32_t* tri, const int32_t* adj_offsets,
                                   const int32_t* adj_items, const float* grad_light, float* grad_vertices,
                                   float intensity_ambient, float intensity_directional, const float* color_ambient,
                                   const float* color_directional, const float* direction, int num_vertices, int num_tri,
                                   int fill_back, d3m_stream_t stream);
/* The form (0 | 1) a launch of d3m_forward_face_index_map_mesh on `batch_size` views of a mesh of `num_triangles` triangles
 * (before fill_back) at `image_size` takes with a workspace of d3m_forward_workspace_bytes(); -1 for invalid sizes.  The
 * automatic choice: bidding for sub-pixel triangles (more than two per three raster pixels) whatever the batch; per-tile
 * lists for big batches (d3m_forward_big_batch) and for coarse meshes (more than 10 raster pixels per triangle; with at
 * least 65 536 (view, triangle) pairs in the batch more than 10 + 2 (views - 1), at most 32); bidding otherwise. */
int d3m_forward_coverage_form(int batch_size, int num_triangles, int image_size);
/* 1 when the launch is a BIG BATCH of an ordinary mesh (more than 65 536 blocks of 8 x 8 pixels, triangles not sub-pixel):
 * every kernel of a step fills the chip by itself.  (Rounds 4-5 ran such a step's branches on one stream; since round 6
 * they run beside each other at every size.  The camera-sharded fit still keys its split exchange on it.) */
int d3m_forward_big_batch(int batch_size, int num_triangles, int image_size);

/* Replaces forward_face_index_map (KCPP:70-95 -> KCU:24-169: kernels 1 and 2).
 *   faces          [B,F,3,3] f32 in   NDC x,y in [-1,1] (+y up), z = depth
 *   face_index_map [B,S,S]   i32 out  index of the nearest covering face, -1 where uncovered
 *   weight_map     [B,S,S,3] f32 out  clamped+renormalised barycentrics of that face, 0 where uncovered
 *   depth_map      [B,S,S]   f32 out  perspective-correct depth, `far` where uncovered
 *   face_inv_map   [B,S,S,3,3] f32 out, written only if return_depth != 0 (may be NULL otherwise), 0 where uncovered
 *   (the uncovered values are the reference's pre-fill, rasterize.py:50-58: every pixel is written, so a caller
 *   that pre-fills as the reference does and one that passes uninitialised memory get the same maps)
 *   faces_inv      [B,F,3,3] f32 i/o  scratch the reference fills for front-facing faces (may be NULL)
 * Row 0 of the maps is the BOTTOM of the image (the flip happens in Python, rasterize.py:305-317). */
int d3m_forward_face_index_map(const float* faces, int32_t* face_index_map, float* weight_map,
                               float* depth_map, float* face_inv_map, float* faces_inv, int batch_size,
                               int num_faces, int image_size, float near, float far, int return_rgb,
                               int return_alpha, int return_depth, void* workspace, size_t workspace_bytes,
                               d3m_stream_t stream);

/* d3m_forward_face_index_map on an INDEXED mesh (an addition): vertices [B,V,3] (screen space), tri [Bt,Ft,3]
 * (Bt = 1: shared), fill_back appends the reversed-winding copies (renderer.py:86), i.e. num_faces =
 * (fill_back ? 2 : 1) * num_tri.  The faces are read through the indices and faces_out [B,num_faces,3,3] receives
 * the dense copy of the faces that CAN OWN A PIXEL -- front-facing, and with a pixel centre inside their (dilated) box:
 * what vertices_to_faces would have produced for the faces any later operator can touch -- so the gather needs no
 * pass of its own.  Every other entry of faces_out is left UNDEFINED (whatever the buffer held: it may be NaN); a
 * caller that sweeps the whole array must fill it first, or use d3m_gather_faces.  Workspace as
 * d3m_forward_workspace_bytes(B, num_faces, S). */
int d3m_forward_face_index_map_mesh(const float* vertices, const int32_t* tri, int tri_batch, int num_vertices,
                                    int num_tri, int fill_back, float* faces_out, int32_t* face_index_map,
                                    float* weight_map, float* depth_map, float* face_inv_map, int batch_size,
                                    int image_size, float near, float far, void* workspace, size_t workspace_bytes,
                                    void* visibility, size_t visibility_size, int flags, d3m_stream_t stream);
/* The same, also writing -- in its last pass -- the output images of the renderer's silhouette / depth modes WITHOUT
 * anti-aliasing (NR/renderer.py:114-183; what d3m_output_epilogue makes of the maps in a pass of its own): alpha_map [B,S,S]
 * (1 where covered; internal layout, row 0 = bottom: what the edge gradient reads), alpha_out and depth_out [B,S,S] (the
 * same and depth_map with the rows reversed, rasterize.py:311-317).  Any of the three may be NULL; face_inv_map must be
 * NULL when one is given (D3M_ERR_INVALID otherwise). */
int d3m_forward_face_index_map_mesh_modes(const float* vertices, const int32_t* tri, int tri_batch, int num_vertices,
                                          int num_tri, int fill_back, float* faces_out, int32_t* face_index_map,
                                          float* weight_map, float* depth_map, float* face_inv_map, int batch_size,
                                          int image_size, float near, float far, void* workspace, size_t workspace_bytes,
                                          void* visibility, size_t visibility_size, float* alpha_map, float* alpha_out,
                                          float* depth_out, int flags, d3m_stream_t stream);
/* flags: D3M_PRECLEARED -- the caller has zeroed the first d3m_forward_clear_bytes(...) bytes of `workspace` (the tile
 * counters and arrival tickets of the per-tile lists, or the z-buffer of the bidding form: whichever form a launch with
 * THIS workspace size takes); 0: the operator clears them itself. */
size_t d3m_forward_clear_bytes(int batch_size, int num_tri, int fill_back, int image_size, size_t workspace_bytes);
/* tri == NULL with tri_batch = -W (here, in d3m_face_light(_backward) and in d3m_vertex_target): the IMPLICIT topology of
 * a depth map's grid mesh with W vertices per row (deep3dmap/core/renderer/utils.py:74-78: num_vertices = H*W, num_tri =
 * 2 (H-1)(W-1), cell (y, x) carries (tl, bl, tr) in the first half of the list and (tr, bl, br) in the second) -- NrRenderer's
 * meshes need no index tensor.
 * `visibility` (NULL, or a blob of d3m_visibility_bytes(B, num_faces)): the tile pass then also leaves the first step
 * of d3m_visibility -- which faces own a pixel -- in the blob, and d3m_visibility(NULL, blob, ...) finishes it without a
 * pass over face_index_map. */

/* Replaces forward_texture_sampling (KCPP:97-124 -> KCU:172-242).
 *   textures [B,F,ts,ts,ts,3] f32 in; rgb_map [B,S,S,3] f32 i/o; sampling_index_map [B,S,S,8] i32 i/o;
 *   sampling_weight_map [B,S,S,8] f32 i/o (the last two may be NULL: they are recomputable). */
int d3m_forward_texture_sampling(const float* faces, const float* textures, const int32_t* face_index_map,
                                 const float* weight_map, const float* depth_map, float* rgb_map,
                                 int32_t* sampling_index_map, float* sampling_weight_map, int batch_size,
                                 int num_faces, int image_size, int texture_size, float eps,
                                 d3m_stream_t stream);

/* Replaces backward_pixel_map (KCPP:126-150 -> KCU:245-503).  Writes (not accumulates) the 9 entries of
 * every face that owns at least one pixel in grad_faces [B,F,3,3] (x,y carry the gradient, z is set to
 * 0).  grad_faces must arrive zero-initialised, as RasterizeFunction.backward provides it
 * (rasterize.py:111): the reference also stores zeros for front-facing faces without pixels and leaves
 * culled faces untouched, which is then identical.  rgb_map/grad_rgb_map may be NULL when return_rgb == 0, alpha
 * likewise.  workspace: d3m_backward_pixel_map_workspace_bytes() bytes of scratch (no init needed); with less -- down to
 * d3m_backward_pixel_map_workspace_min_bytes() -- the crossings that find no room are walked one thread each (slow,
 * same result); below that D3M_ERR_WORKSPACE. */
size_t d3m_backward_pixel_map_workspace_bytes(int batch_size, int num_faces, int image_size);
size_t d3m_backward_pixel_map_workspace_min_bytes(int batch_size, int num_faces, int image_size);
/* Optional destination of face gradients (an ADDITION to the reference's interface; NULL = the dense grad_faces):
 * when `faces` was gathered from vertices (vertices_to_faces + fill_back, renderer.py:86 / vertices_to_faces.py),
 * the gradient of each face corner is accumulated straight into grad_vertices [B,num_vertices,3] (+=, float
 * atomics) - the composition with the adjoint of that gather - and grad_faces may be NULL. */
typedef struct {
    float* grad_vertices;
    const int32_t* tri;     /* [tri_batch, num_tri, 3]; num_faces = (fill_back ? 2 : 1) * num_tri */
    int num_vertices, num_tri, tri_batch, fill_back;
} d3m_vertex_target;
/* `unscaled` (NULL = the gradient maps are final): the maps are the unscaled ones a fused fit objective left
 * (d3m_render_lit_epilogue with fit->grad_*_map, see struct d3m_fit_targets below); their scalar factors are applied
 * as the maps are read.
 * Alpha only (return_alpha, not return_rgb) with FINAL gradients takes no pass over the pixels to pack them: the walks read
 * grad_alpha_map and face_index_map as they are.  The same for the gradient of the OUTPUT image instead of the internal map:
 * `unscaled` with only grad_alpha_map = that image's gradient [B,s,s] and flags = D3M_GRAD_OF_OUTPUT_IMAGE (s = image_size:
 * the image is the alpha map with its rows reversed) or D3M_GRAD_OF_OUTPUT_IMAGE | D3M_FIT_POOLED (s = image_size / 2: its 2x2
 * average); the argument grad_alpha_map is then not read (NULL).  Any other `unscaled` without scratch or records: D3M_ERR_INVALID. */
typedef struct d3m_fit_targets d3m_fit_targets;
int d3m_backward_pixel_map(const float* faces, const int32_t* face_index_map, const float* rgb_map,
                           const float* alpha_map, const float* grad_rgb_map, const float* grad_alpha_map,
                           float* grad_faces, int batch_size, int num_faces, int image_size, float eps,
                           int return_rgb, int return_alpha, void* workspace, size_t workspace_bytes,
                           const d3m_vertex_target* vertex_target, void* visibility, void* edge_plan,
                           size_t edge_plan_size, const d3m_fit_targets* unscaled, d3m_stream_t stream);

/* Which faces own a pixel depends on face_index_map only.  d3m_visibility builds, once per forward result, the
 * flags and the compacted list of those faces in a caller-owned blob of d3m_visibility_bytes(); backward operators
 * that are handed the blob (`visibility`, NULL = each builds its own) skip that work and run over the list.
 * face_index_map NULL: the blob went through d3m_forward_face_index_map_mesh, which left the first step in it (the
 * marks) and cleared the list's counter -- ONE finishing call per forward: the list is appended under that counter, so a
 * second d3m_visibility(NULL, ...) on the same blob without a new forward would append the list a second time (call with
 * the face_index_map instead, which clears first).  The list is ascending within chunks of 8192 faces; the chunks land in
 * arrival order, so the ORDER of the list -- and with it the order of the float atomics of the passes that run over it
 * -- may differ from run to run (the sums agree to rounding; the flags and the set of listed faces do not vary). */
size_t d3m_visibility_bytes(int batch_size, int num_faces);
int d3m_visibility(const int32_t* face_index_map, void* visibility, size_t visibility_size, int batch_size,
                   int num_faces, int image_size, d3m_stream_t stream);
/* Where the edges of the visible faces cross the pixel grid -- every walk of KCU:312-362 / :417-431 starts at such a
 * crossing -- depends on `faces` and face_index_map only, not on the gradient maps.  d3m_edge_plan builds, once per
 * forward result, the crossings' records (position, the in-pixel and whether the face owns it, the inward walk's end,
 * the first factors of KCU:404/:409's `dist`) grouped by image line in a caller-owned blob of d3m_edge_plan_bytes() (less is
 * accepted down to d3m_edge_plan_min_bytes(): crossings without room are then walked the slow way);
 * d3m_backward_pixel_map handed the blob (`edge_plan`, NULL = it builds its own in its workspace) starts with the
 * line walk.  `visibility`: the d3m_visibility blob of the same forward result (required). */
size_t d3m_edge_plan_bytes(int batch_size, int num_faces, int image_size);
/* Two arrays of [B,2,S] ints inside the plan blob's zeroed prefix, at the returned byte offset and *bytes_each behind it,
 * which d3m_edge_plan clears together with its own counters: a caller that builds the plan BEFORE it renders with a fused
 * fit objective on the same stream may use them as d3m_fit_targets.edge_nz_lo_inv / edge_nz_hi1 ("zeroed by the caller")
 * and saves the launch that clears them.  The plan itself does not use them. */
size_t d3m_edge_plan_extents_offset(int batch_size, int num_faces, int image_size, size_t* bytes_each);
size_t d3m_edge_plan_min_bytes(int batch_size, int num_faces, int image_size);
/* flags: D3M_PRECLEARED -- the caller has zeroed the blob's first d3m_edge_plan_clear_bytes() bytes (line counters,
 * cursors, arrival tickets and the extents above). */
size_t d3m_edge_plan_clear_bytes(int batch_size, int num_faces, int image_size);
int d3m_edge_plan(const float* faces, const int32_t* face_index_map, void* visibility, void* edge_plan,
                  size_t edge_plan_size, int batch_size, int num_faces, int image_size, int flags, d3m_stream_t stream);

/* Scratch for the two entry points below: one int per face.  With it the sums are GATHERED per visible
 * face (no atomics; faces with a very large bounding box still use the per-pixel atomic kernel);
 * without it (NULL) the reference's per-pixel float atomics are used throughout. */
size_t d3m_backward_faces_workspace_bytes(int batch_size, int num_faces);

/* Replaces backward_textures (KCPP:152-168 -> KCU:506-540): grad_textures [B,F,ts,ts,ts,3] += .
 * `faces` [B,F,3,3] is an ADDITION to the reference's argument list (needed to gather per face; pass
 * NULL to force the atomic form). */
int d3m_backward_textures(const float* faces, const int32_t* face_index_map, const float* sampling_weight_map,
                          const int32_t* sampling_index_map, const float* grad_rgb_map, float* grad_textures,
                          int batch_size, int num_faces, int image_size, int texture_size, void* workspace,
                          size_t workspace_bytes, d3m_stream_t stream);

/* Replaces backward_depth_map (KCPP:170-191 -> KCU:543-592): grad_faces += (after backward_pixel_map).
 * face_inv_map may be NULL: the face inverse is then recomputed from `faces` (bit-identical values). */
int d3m_backward_depth_map(const float* faces, const float* depth_map, const int32_t* face_index_map,
                           const float* face_inv_map, const float* weight_map, const float* grad_depth_map,
                           float* grad_faces, int batch_size, int num_faces, int image_size, void* workspace,
                           size_t workspace_bytes, d3m_stream_t stream);

/* d3m_backward_depth_map for a mesh pipeline (faces = the dense copy d3m_forward_face_index_map_mesh left, visibility = the
 * blob it marked and d3m_visibility finished): runs over the listed faces only and ADDS its sums straight into
 * vertex_target->grad_vertices (float atomics) -- no dense grad_faces, no scatter-add pass behind it.  large_counter: 256
 * bytes, zero when the kernels start (cleared here unless flags & D3M_PRECLEARED).  flags & D3M_GRAD_OF_OUTPUT_IMAGE:
 * grad_depth_map is the gradient of the OUTPUT depth image [B,S,S] of a render without anti-aliasing (rows top to bottom,
 * rasterize.py:311-317) -- read with the flip undone, so the adjoint of the output epilogue needs no pass of its own. */
#define D3M_GRAD_OF_OUTPUT_IMAGE 32
int d3m_backward_depth_map_mesh(const float* faces, const float* depth_map, const int32_t* face_index_map,
                                const float* weight_map, const float* grad_depth_map, int batch_size, int num_faces,
                                int image_size, const d3m_vertex_target* vertex_target, void* visibility,
                                void* large_counter, int flags, d3m_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * B. The eager-torch steps either side of those operators, as single HIP passes.
 * ---------------------------------------------------------------------------------------------- */

/* Camera parameter block, HOST memory, copied by value into the launch.
 *   LOOK_AT / LOOK : rot = rows (x_axis, y_axis, z_axis) per view, eye per view
 *                    (neural_renderer/look_at.py:48-60, look.py:39-51), then, if perspective != 0,
 *                    x,y /= z * tan(viewing_angle) (perspective.py:13-20; `inv_width`= 1/tan is NOT
 *                    used: the divisions are kept as the reference writes them).
 *   PROJECTION     : v*R^T + t, x/(z+1e-9), distortion (k1,k2,p1,p2,k3), K, v = orig-v, map to
 *                    [-1,1] (projection.py:19-42).
 * Per-view arrays live on the DEVICE: rot [Bc,3,3], eye_or_t [Bc,3], K [Bc,3,3], dist [Bc,5]; Bc is
 * either 1 (broadcast) or batch_size. */
typedef struct d3m_camera {
    int mode;              /* D3M_CAMERA_* */
    int perspective;       /* look / look_at only */
    float tan_half_width;  /* tan(viewing_angle), look / look_at only */
    float orig_size;       /* projection only */
    const float* rot;      /* device: look/look_at basis rows, or projection R */
    const float* eye_or_t; /* device: look/look_at eye, or projection t */
    const float* K;        /* device: projection only */
    const float* dist;     /* device: projection only */
    int rot_batch, eye_batch, K_batch, dist_batch; /* 1 or batch_size each */
} d3m_camera;

/* look_at / look basis on the device (look_at.py:48-53, look.py:39-44): rot_out [B,3,3] rows
 * (x_axis, y_axis, z_axis); eye/at_or_direction/up are [n,3] device arrays with n = 1 or B.
 * is_look_at != 0: z = normalise(at - eye); else z = normalise(direction). */
int d3m_camera_basis(const float* eye, int eye_batch, const float* at_or_direction, int at_batch, const float* up,
                     int up_batch, int is_look_at, float* rot_out, int batch_size, d3m_stream_t stream);

/* vertices [Bv,V,3] (Bv = 1 broadcasts one mesh to every view) -> out [B,V,3] in NDC + depth. */
/* d3m_lit_front -- the FIRST launch of a lit render step, three jobs in one grid (each optional):
 *   camera   screen_out [B,V,3] = camera(vertices) as d3m_camera_forward (cam NULL: none).  `basis` (look_at / look only,
 *            NULL: cam->rot holds the basis): the basis is computed here from eye / at_or_direction / up exactly as
 *            d3m_camera_basis does and ALSO stored to cam->rot ([rot_batch,3,3], writable) for d3m_camera_backward;
 *   light    light [light_batch, F', 3] as d3m_face_light (light NULL: none);
 *   clears   up to 10 (pointer, bytes) ranges zeroed (4-byte multiples, 4-byte aligned) -- see "Clears" above.
 * (NR/look_at.py:48-60 + NR/lighting.py:33-56 + the torch.zeros of NR/rasterize.py:50-58,111-115, as one kernel.) */
typedef struct d3m_basis {
    const float* eye;             /* [eye_batch,3] */
    const float* at_or_direction; /* [at_batch,3]: `at` (look_at) or the viewing direction (look) */
    const float* up;              /* [up_batch,3] */
    int eye_batch, at_batch, up_batch, is_look_at;
} d3m_basis;
int d3m_lit_front(const float* vertices, int vertices_batch, const d3m_camera* cam, const d3m_basis* basis, float* screen_out,
                  int batch_size, int num_vertices, const int32_t* tri, int tri_batch, int num_tri, int fill_back,
                  float* light, int light_batch, float intensity_ambient, float intensity_directional,
                  const float* color_ambient, const float* color_directional, const float* direction,
                  void* const* zero_ptrs, const size_t* zero_bytes, int zero_count, d3m_stream_t stream);
/* d3m_lit_back -- the LAST launch of such a step: grad_vertices [vertices_batch,V,3] += the camera's adjoint of grad_screen
 * [B,V,3] (d3m_camera_backward) + the light's adjoint of grad_light [light_batch,F',3] (d3m_face_light_backward), both with
 * float atomics in one grid: grad_vertices must hold zeros (or other contributions) when the launch starts. */
int d3m_lit_back(const float* vertices, int vertices_batch, const d3m_camera* cam, const float* grad_screen,
                 float* grad_vertices, int batch_size, int num_vertices, const int32_t* tri, int tri_batch, int num_tri,
                 int fill_back, const float* grad_light, int light_batch, float intensity_ambient,
                 float intensity_directional, const float* color_ambient, const float* color_directional,
                 const float* direction, d3m_stream_t stream);
int d3m_camera_forward(const float* vertices, int vertices_batch, const d3m_camera* cam, float* out,
                       int batch_size, int num_vertices, d3m_stream_t stream);
/* grad_vertices [Bv,V,3] = d(out)/d(vertices)^T grad_out; with Bv == 1 the B views are summed. */
int d3m_camera_backward(const float* vertices, int vertices_batch, const d3m_camera* cam,
                        const float* grad_out, float* grad_vertices, int batch_size, int num_vertices,
                        d3m_stream_t stream);
/* the same, ADDED to grad_vertices (which then already holds another path's share, e.g. d3m_face_light_backward's: the two
 * gradients of a mesh that is both projected and lit need no separate sum) */
int d3m_camera_backward_add(const float* vertices, int vertices_batch, const d3m_camera* cam,
                            const float* grad_out, float* grad_vertices, int batch_size, int num_vertices,
                            d3m_stream_t stream);

/* vertices_to_faces (neural_renderer/vertices_to_faces.py:16-22) with the fill_back copy made on the
 * fly (renderer.py:86): faces_out [B,F',3,3], F' = 2F if fill_back else F; face F+f is face f with
 * vertex order reversed.  tri [Bt,F,3] i32, Bt = 1 or B. */
int d3m_gather_faces(const float* vertices, const int32_t* tri, int tri_batch, float* faces_out,
                     int batch_size, int num_vertices, int num_tri, int fill_back, d3m_stream_t stream);
/* Its backward: the per-vertex scatter-add.  grad_vertices [B,V,3] += ; wave-level pre-reduction of
 * lanes that hit the same vertex, then float atomics. */
int d3m_scatter_face_grads(const float* grad_faces, const int32_t* tri, int tri_batch, float* grad_vertices,
                           int batch_size, int num_vertices, int num_tri, int fill_back,
                           d3m_stream_t stream);

/* lighting (neural_renderer/lighting.py:33-56): light = ia*ca + id*cd*relu(normal . direction) per face,
 * normal = normalise(cross(v0-v1, v2-v1)) with F.normalize's eps 1e-5, on WORLD-space faces
 * (renderer.py:159); textures_out = textures_in * light (may alias for the reference's in-place form).
 *   faces [N,3,3], textures [N,ts,ts,ts,3] with N = num_faces_total (= B*F'); the three colour /
 *   direction vectors are HOST arrays of 3 floats. */
int d3m_lighting_forward(const float* faces, const float* textures_in, float* textures_out,
                         float intensity_ambient, float intensity_directional, const float* color_ambient,
                         const float* color_directional, const float* direction, long num_faces_total,
                         int texture_size, d3m_stream_t stream);
/* grad_textures [N,ts,ts,ts,3] = grad_out*light (NULL to skip); grad_faces [N,3,3] = gradient through
 * the face normal (NULL to skip; written, not accumulated). */
int d3m_lighting_backward(const float* faces, const float* textures_in, const float* grad_out,
                          float* grad_textures, float* grad_faces, float intensity_ambient,
                          float intensity_directional, const float* color_ambient, const float* color_directional,
                          const float* direction, long num_faces_total, int texture_size, d3m_stream_t stream);

/* get_transform_matrices (deep3dmap/core/renderer/utils.py:34-71): view [B,n] = (rx, ry, rz[, tx, ty[, tz]]), n = 3, 5
 * or 6 -> rot [B,3,3] = Rz Ry Rx and trans [B,3] (missing components 0), and its adjoint (grad_rot / grad_trans may be
 * NULL = zero).  One launch each way instead of ~35 eager kernels. */
int d3m_view_transform(const float* view, int num_components, float* rot, float* trans, int batch_size,
                       d3m_stream_t stream);
int d3m_view_transform_backward(const float* view, int num_components, const float* grad_rot, const float* grad_trans,
                                float* grad_view, int batch_size, d3m_stream_t stream);
/* NrRenderer's depth map -> warped pixel grid (deep3dmap/core/renderer/renderer_nr.py:74-114,141-158): every
 * "depth_to_3d_grid, then rigid transforms, then maybe grid_3d_to_2d" of the class in one pass.
 *   P = depth * inv_K (x, y, 1)^T;   Q = rot_b (P - c) + c + trans_b,  c = (0, 0, rot_center_depth)
 *   K == NULL: out [B,H*W,3] = Q;    K [1|B,3,3]: out [B,H,W,2] = ((K (Q / Q.z)).xy / (W-1, H-1)) * 2 - 1, the sampling
 *   grid of F.grid_sample (renderer_nr.py:82-88).
 * crop (NULL or {top, bottom, left, right}: render_yaw's crop_mesh, renderer_nr.py:145-158): border rows / columns take
 * the y,z / x,z of the first kept row / column.  Inverse warps and chains of transforms are composed into (rot, trans)
 * by the caller.  The adjoint (no crop) WRITES grad_depth [B,H,W], grad_rot [B,3,3], grad_trans [B,3] (each may be NULL). */
int d3m_grid_warp(const float* depth, const float* inv_K, int inv_K_batch, const float* rot, const float* trans,
                  float rot_center_depth, const float* K, int K_batch, const int* crop, float* out, int batch_size,
                  int height, int width, d3m_stream_t stream);
int d3m_grid_warp_backward(const float* depth, const float* inv_K, int inv_K_batch, const float* rot, const float* trans,
                           float rot_center_depth, const float* K, int K_batch, const float* grad_out, float* grad_depth,
                           float* grad_rot, float* grad_trans, int batch_size, int height, int width, d3m_stream_t stream);
/* NrRenderer.get_normal_from_depth (renderer_nr.py:127-139): normal [B,H,W,3] of the back-projected depth map, central
 * differences, (0,0,1) on the one-pixel border, normalised with EPS = 1e-7; and its adjoint (grad_depth WRITTEN). */
int d3m_depth_normals(const float* depth, const float* inv_K, int inv_K_batch, float* normal, int batch_size, int height,
                      int width, d3m_stream_t stream);
int d3m_depth_normals_backward(const float* depth, const float* inv_K, int inv_K_batch, const float* grad_normal,
                               float* grad_depth, int batch_size, int height, int width, d3m_stream_t stream);
/* get_textures_from_im (deep3dmap/core/renderer/utils.py:81-107): textures [B, 2(H-1)(W-1), ts^3, C] of the implicit
 * grid mesh from im [B,C,H,W], ts = 1 or 2 (anything else: D3M_ERR_INVALID, as utils.py:106 raises); and its adjoint
 * (grad_im WRITTEN). */
int d3m_textures_from_im(const float* im, float* textures, int batch_size, int channels, int height, int width,
                         int texture_size, d3m_stream_t stream);
int d3m_textures_from_im_backward(const float* grad_textures, float* grad_im, int batch_size, int channels, int height,
                                  int width, int texture_size, d3m_stream_t stream);

/* Pt3dRenderer.sample's per-pixel pass (deep3dmap/core/renderer/renderer_pt3d.py:75-97; the reference runs it through
 * pytorch3d 0.6.1's TexturesUV + SoftPhongShader): for the template mesh rasterized in UV space -- face_index_map /
 * weight_map [coverage_batch = 1|B, T,T(,3)] of d3m_forward_face_index_map on its fill_back faces -- every covered pixel samples
 * imgs [B,C<=3,H,W] bilinearly (align_corners, border padding, v up) at the barycentric mix of its face's per-vertex
 * image coordinates uvs [B,V,2] and shades it with pytorch3d's Phong model against the interpolated vertex normals
 * vnormals [V,3] at the interpolated position verts [V,3]:
 *     colour = (ambient + diffuse * relu(n.l)) * texel + specular * relu(v.r)^shininess [n.l > 0]
 * `light` = HOST array of 10 floats: light location (3), camera centre (3), ambient, diffuse, specular, shininess (the
 * products light colour x material colour; the reference's shader is built without lights / materials, so pytorch3d's
 * defaults apply: (0,1,0), (0,0,2.7), 0.5, 0.3, 0.2, 64).  out_img / out_mask [B,T,T,4] (row 0 = top): rgb = the colour
 * above / the same with texel = 1, alpha = coverage; zeros where nothing covers or used[b] == 0.
 * The adjoint ADDS into grad_imgs [B,C,H,W] and grad_uvs [B,V,2] (either may be NULL; caller zeroes). */
int d3m_uv_unwrap(const int32_t* face_index_map, const float* weight_map, const int32_t* tri, const float* verts,
                  const float* vnormals, const float* uvs, const float* imgs, const int32_t* used, const float* light,
                  float* out_img, float* out_mask, int batch_size, int coverage_batch, int texture_size, int num_tri,
                  int num_vertices, int channels, int height, int width, d3m_stream_t stream);
int d3m_uv_unwrap_backward(const int32_t* face_index_map, const float* weight_map, const int32_t* tri, const float* verts,
                           const float* vnormals, const float* uvs, const float* imgs, const int32_t* used,
                           const float* light, const float* grad_img, float* grad_imgs, float* grad_uvs, int batch_size,
                           int coverage_batch, int texture_size, int num_tri, int num_vertices, int channels, int height,
                           int width, d3m_stream_t stream);

/* --- lighting and fill_back applied on the fly (instead of renderer.py:155-167,203-215 materialising
 * cat(textures, textures.permute(0,1,4,3,2,5)) * light per view) ---------------------------------------
 * light [Bl,F',3]: per-face light of the fill_back'd face array on WORLD vertices [Bv,V,3] / tri [Bt,F,3];
 * Bl = 1 when one mesh is shared by every view, else the batch size. */
int d3m_face_light(const float* vertices, int vertices_batch, const int32_t* tri, int tri_batch, float* light,
                   float intensity_ambient, float intensity_directional, const float* color_ambient,
                   const float* color_directional, const float* direction, int light_batch, int num_vertices,
                   int num_tri, int fill_back, d3m_stream_t stream);
/* grad_light [Bl,F',3] -> grad_vertices [Bv,V,3] += (through the face normals; float atomics). */
int d3m_face_light_backward(const float* vertices, int vertices_batch, const int32_t* tri, int tri_batch,
                            const float* grad_light, float* grad_vertices, float intensity_ambient,
                            float intensity_directional, const float* color_ambient, const float* color_directional,
                            const float* direction, int light_batch, int num_vertices, int num_tri, int fill_back,
                            d3m_stream_t stream);
/* forward_texture_sampling on the VIRTUAL lit array: faces [B,F',3,3] (F' = 2*num_tri if fill_back),
 * textures [Bx,num_tri,ts,ts,ts,3] (Bx = 1: shared), light [Bl,F',3]; writes rgb_map [B,S,S,3] for covered
 * pixels.  Colours are bit-identical to sampling the materialised array (same f32 products). */
int d3m_forward_texture_sampling_lit(const float* faces, const float* textures, int textures_batch, const float* light,
                                     int light_batch, const int32_t* face_index_map, const float* weight_map,
                                     const float* depth_map, float* rgb_map, int batch_size, int num_tri,
                                     int fill_back, int image_size, int texture_size, float eps, d3m_stream_t stream);
/* (forward sampling entry points only) textures_batch = -W, W >= 2: `textures` is an IMAGE [B,3,H,W] and the 2x2x2 cubes of
 * the grid mesh's faces (num_tri = 2 (H-1)(W-1), texture_size 2) are evaluated from it where they are sampled --
 * get_textures_from_im (deep3dmap/core/renderer/utils.py:81-107) without its [B,F,2,2,2,3] array; same bits.
 * d3m_forward_texture_sampling_lit followed by d3m_output_epilogue in one pass, without the intermediate
 * rgb_map: writes the blended internal-resolution rgb_blended [B,S,S,3] (covered ? sampled : background) and
 * alpha_map [B,S,S] (NULL to skip) that the backward pass reads, and the output images rgb_out [B,3,s,s],
 * alpha_out / depth_out [B,s,s] (NULL to skip; s = S/2 when anti_aliasing), flipped as rasterize.py:305-326.
 *
 * `fit` (NULL = none; an ADDITION to the reference's structure, not its interface): the multi-view fit objective
 *     photometric_loss(rgb, rgb_target, mask) + sum((alpha - alpha_target)^2) / (s*s) + photometric_loss(depth, ...)
 * (deep3dmap/core/utils/utils.py:105-114 composed as d3m_fit_loss_forward does) evaluated in the same pass, where
 * the images are produced: *fit->loss receives the value, the images themselves need not be written (rgb_out NULL)
 * and are not read again.  alpha_map is required.  With anti_aliasing the targets are at the output size s = S/2 and the
 * objective is that of the pooled images (the records form below then carries D3M_FIT_POOLED).
 * With fit->grad_*_map set the pass also leaves the objective's gradient wrt the internal-resolution maps
 * (rgb_blended, alpha_map, depth_map) there, WITHOUT the scalar factors that are only known later - sign(rgb - target)
 * * mask, 2 (alpha - target), sign(depth - target) * mask - so that backward needs no pass over the pixels of its
 * own: d3m_backward_pixel_map / d3m_backward_textures_lit handed these maps AND the same struct as `unscaled` (with
 * grad_loss filled in; NULL = 1) multiply by grad_loss / (3 sum(mask)), grad_loss / (s*s), grad_loss / sum(mask)
 * as they read them.  (d3m_fit_loss_backward followed by d3m_output_epilogue_backward, without either.) */
struct d3m_fit_targets {
    const float* rgb_target;     /* [B,3,S,S]  output image layout (flipped, channel-major) */
    const float* depth_target;   /* [B,S,S] */
    const float* alpha_target;   /* [B,S,S] */
    const float* mask;           /* [B,S,S] */
    float* scratch;              /* d3m_render_fit_scratch_floats() floats, kept from forward to backward */
    float* loss;                 /* [1] */
    float* grad_rgb_map;         /* [B,S,S,3] \                                                        */
    float* grad_alpha_map;       /* [B,S,S]    > unscaled gradient maps, written by the epilogue; all or none */
    float* grad_depth_map;       /* [B,S,S]   /                                                         */
    const float* grad_loss;      /* [1] device scalar, read by the backward operators; NULL = 1 */
    const float* mask_sum;       /* [1] device scalar or NULL.  NULL: the photometric terms are normalised by the sum of
                                  * `mask` over THIS batch.  When the batch is one rank's shard of a larger objective
                                  * (camera-sharded fit), the sum of the mask over ALL shards: the shard's value is then
                                  * its additive part of the global objective and gradients add up across ranks. */
    /* The gradient wrt rgb_blended / alpha_map in the form the edge gradient reads it (all four or none; needs mask_sum
     * and grad_depth_map; grad_rgb_map / grad_alpha_map are then not written and may be NULL): per pixel
     *   edge_grad = (2 (alpha - target) / (S*S), sign(rgb - target) mask / (3 mask_sum))      float4 [B,S,S]
     *   edge_dot  = (<(alpha, rgb), edge_grad>, face index bits)                               float2 [B,S,S]
     * and per image line (b*2 + axis)*S + d0 the extent of its non-zero records (caller-zeroed int [B,2,S] each:
     * S - first, last + 1).  Everything but grad_loss is in there; d3m_backward_pixel_map / d3m_backward_textures_lit
     * handed the struct as `unscaled` read the records instead of packing the maps (no pass over the pixels).
     * With anti_aliasing (flags & D3M_FIT_POOLED, required then and only then) the records stay per INTERNAL pixel: each of
     * an output pixel's four gets a quarter of its gradient -- (2 (alpha_o - target) / (S*S), sign(rgb_o - target) mask / 4 /
     * (3 mask_sum)) -- and grad_depth_map sign(depth_o - target) mask / 4. */
    void* edge_grad;
    void* edge_dot;
    int* edge_nz_lo_inv;
    int* edge_nz_hi1;
    int flags;                   /* D3M_PRECLEARED: the caller has zeroed the scratch's ticket word
                                  * (d3m_render_fit_scratch_clear_range) -- see "Clears" above; else the pass does it.
                                  * D3M_FIT_FINISH_DEFERRED (records form of d3m_render_lit_epilogue only): the pass leaves
                                  * its partial sums in `scratch` and does NOT complete *loss; d3m_backward_textures_lit,
                                  * handed this struct (same flag) as `unscaled`, completes it in a kernel it launches anyway --
                                  * for callers that run the backward pass right behind the forward pass: no finishing launch
                                  * D3M_FIT_POOLED: the records belong to an objective on the 2x2-pooled images (see above) */
};
#define D3M_FIT_FINISH_DEFERRED 2
#define D3M_FIT_POOLED 4
/* ... or, should the backward pass not come that way after all, by this call (the ticket must still be zero: one finish). */
int d3m_fit_finish(const d3m_fit_targets* fit, int batch_size, int image_size, d3m_stream_t stream);
/* scratch: totals | partial sums | group sums | ticket. */
size_t d3m_render_fit_scratch_floats(int batch_size, int image_size);
size_t d3m_render_fit_scratch_clear_range(int batch_size, int image_size, size_t* offset_floats);   /* returns a count of floats */
/* The same objective evaluated on FINISHED images -- rgb [B,3,S,S], depth / alpha [B,S,S], row 0 = top: what
 * d3m_render_lit_epilogue wrote as rgb_out / depth_out / alpha_out without anti-aliasing -- with its gradient left as the
 * edge gradient's per-pixel records, exactly as the fused pass leaves them (fit->edge_grad, edge_dot, edge_nz_* zeroed by
 * the caller, grad_depth_map, mask_sum; *fit->loss is complete on return of the stream).  face_index_map [B,S,S] (row 0 =
 * bottom) supplies the records' owner.  This is how the reference-shaped composition  loss(*renderer.render(...))
 * (NR/renderer.py:200-246 + the caller's loss) reaches the records route of d3m_backward_pixel_map /
 * d3m_backward_textures_lit without gradient images. */
int d3m_fit_loss_records(const float* rgb, const float* depth, const float* alpha, const int32_t* face_index_map,
                         const d3m_fit_targets* fit, int batch_size, int image_size, d3m_stream_t stream);
int d3m_render_lit_epilogue(const float* faces, const float* textures, int textures_batch, const float* light,
                            int light_batch, const int32_t* face_index_map, const float* weight_map,
                            const float* depth_map, const float* background, int background_batch,
                            float* rgb_blended, float* alpha_map, float* rgb_out, float* alpha_out, float* depth_out,
                            int batch_size, int num_tri, int fill_back, int image_size, int texture_size, float eps,
                            int anti_aliasing, const d3m_fit_targets* fit, d3m_stream_t stream);
/* Its backward (replaces backward_textures + the adjoint of lighting and of the fill_back cat):
 * grad_textures [Bx,num_tri,ts^3,3] is WRITTEN (summed over views when Bx = 1); grad_light [Bl,F',3] is
 * written when not NULL.  Sampling weights are recomputed from weight_map / depth_map (no 64 B/pixel
 * sampling maps).  Gathered per visible face for ts = 2, per-pixel float atomics otherwise.
 * grad_depth_map [B,S,S] / grad_faces [B,F',3,3] (both or neither): when given, the depth gradient of
 * backward_depth_map (KCU:543-592) is ADDED to grad_faces in the same pass over the faces' pixels, i.e. this one
 * call then stands for backward_textures + backward_depth_map of rasterize.py:146-151.  With vertex_target the
 * depth gradient goes to grad_vertices instead and grad_faces may be NULL. */
size_t d3m_backward_textures_lit_workspace_bytes(int batch_size, int num_tri, int fill_back, int texture_size);
int d3m_backward_textures_lit(const float* faces, const float* textures, int textures_batch, const float* light,
                              int light_batch, const int32_t* face_index_map, const float* weight_map,
                              const float* depth_map, const float* grad_rgb_map, float* grad_textures, float* grad_light,
                              const float* grad_depth_map, float* grad_faces, int batch_size, int num_tri, int fill_back,
                              int image_size, int texture_size, float eps, void* workspace, size_t workspace_bytes,
                              const d3m_vertex_target* vertex_target, void* visibility,
                              const d3m_fit_targets* unscaled, int flags, d3m_stream_t stream);
/* flags: D3M_PRECLEARED -- the caller has zeroed the (up to four) ranges d3m_backward_textures_lit_clear_ranges reports for
 * the same arguments: grad_light, the workspace's view masks / counter / arrival tickets (and, where they apply, the
 * per-view gradients and the flags).  Returns the number of ranges written to ptrs / bytes (room for four). */
int d3m_backward_textures_lit_clear_ranges(float* grad_textures, int textures_batch, float* grad_light, int light_batch,
                                           int batch_size, int num_tri, int fill_back, int texture_size, void* workspace,
                                           int has_visibility, void** ptrs, size_t* bytes);

/* Output epilogue of rasterize_rgbad (rasterize.py:305-326) in one pass: background blend + alpha
 * (rasterize.py:181-195), HWC->CHW, vertical flip, optional 2x2 average pool.
 *   in : face_index_map [B,S,S], rgb_map [B,S,S,3] (sampled, NOT yet blended; NULL if !rgb),
 *        depth_map [B,S,S] (NULL if !depth); background [Bb,3] device, Bb = 1 or B.
 *   out: rgb_out [B,3,s,s], alpha_out [B,s,s], depth_out [B,s,s]  (s = S/2 if anti_aliasing else S);
 *        rgb_blended [B,S,S,3] and alpha_map [B,S,S] are the internal-resolution maps the backward
 *        needs (either may be NULL when the mode does not use it). */
int d3m_output_epilogue(const int32_t* face_index_map, const float* rgb_map, const float* depth_map,
                        const float* background, int background_batch, float* rgb_blended, float* alpha_map,
                        float* rgb_out, float* alpha_out, float* depth_out, int batch_size, int image_size,
                        int anti_aliasing, d3m_stream_t stream);
/* Its adjoint: output-resolution grads -> internal maps grad_rgb_map [B,S,S,3], grad_alpha_map [B,S,S],
 * grad_depth_map [B,S,S] (un-pool, un-flip, CHW->HWC).  No coverage mask is applied: the reference
 * blends the background inside RasterizeFunction.forward, so its backward kernels receive the
 * gradient wrt the blended image at every pixel (rasterize.py:83,141-147).  NULL pairs are skipped. */
int d3m_output_epilogue_backward(const float* grad_rgb_out, const float* grad_alpha_out,
                                 const float* grad_depth_out, float* grad_rgb_map, float* grad_alpha_map,
                                 float* grad_depth_map, int batch_size, int image_size, int anti_aliasing,
                                 d3m_stream_t stream);
/* The same adjoint, with the rgb / alpha gradients leaving as the per-pixel records the edge gradient walks over
 * (edge_grad [B,S,S] float4, edge_dot [B,S,S] float2, the lines' non-zero extents edge_nz_* [B,2,S] i32: see
 * d3m_fit_targets) instead of gradient maps; the depth gradient still leaves as a map.  d3m_backward_pixel_map and
 * d3m_backward_textures_lit take the records through `unscaled` with scratch = grad_loss = mask_sum = NULL (final). */
int d3m_output_epilogue_backward_records(const float* grad_rgb_out, const float* grad_alpha_out,
                                         const float* grad_depth_out, const int32_t* face_index_map,
                                         const float* rgb_map, const float* alpha_map, void* edge_grad, void* edge_dot,
                                         int* edge_nz_lo_inv, int* edge_nz_hi1, float* grad_depth_map, int batch_size,
                                         int image_size, int anti_aliasing, d3m_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * C. Losses on the path (deep3dmap/core/utils/utils.py:82-114; examples/example2.py:43-47).
 *    Each writes ONE f32 to `loss` (device) and, if grad_* != NULL, the gradient for d(loss) = 1.
 * ---------------------------------------------------------------------------------------------- */
/* photometric_loss(im1, im2, mask, conf_sigma): im [B,C,H,W]; mask/conf_sigma [B,1,H,W] or NULL.
 * scratch: 2048 floats of device memory (per-workgroup partial sums; no initialisation needed);
 * batch_size * channels must not exceed 1024. */
int d3m_photometric_loss(const float* im1, const float* im2, const float* mask, const float* conf_sigma,
                         float* loss, float* grad_im1, float* scratch, int batch_size, int channels,
                         int height, int width, d3m_stream_t stream);
/* sum((a-b)^2) over n elements (silhouette loss of the examples).  scratch: 1024 floats. */
int d3m_sum_squared_error(const float* a, const float* b, float* loss, float* grad_a, float* scratch, long n,
                          d3m_stream_t stream);

/* smooth_loss of ONE map pred [B,H,W] (deep3dmap/core/utils/utils.py:82-102, one pyramid level, weight 1):
 * mean|dxx| + mean|dxy| + mean|dyx| + mean|dyy| of the nested first differences; H, W >= 3.  scratch: 4096 floats.
 * backward: grad_pred = *grad_loss * d loss / d pred (grad_loss is a device scalar). */
int d3m_smooth_loss_forward(const float* pred, float* loss, float* scratch, int batch_size, int height, int width,
                            d3m_stream_t stream);
int d3m_smooth_loss_backward(const float* pred, const float* grad_loss, float* grad_pred, int batch_size, int height,
                             int width, d3m_stream_t stream);

/* The multi-view fit objective in one reduction + one finish launch, and its gradients in one
 * more:  loss = photometric_loss(rgb, rgb_target, mask) + sum((alpha - alpha_target)^2) / (H*W)
 *             + photometric_loss(depth, depth_target, mask)
 * with rgb [B,3,H,W], depth / alpha / mask [B,H,W].  Same value and gradients as composing the three operators
 * above.  scratch: 4104 floats, written by forward and read by backward (it keeps the reduction totals).
 * grad_loss: device scalar (NULL = 1).  Any of the grad_* outputs may be NULL.
 * mask_sum: device scalar replacing sum(mask) as the photometric terms' normaliser (NULL = sum over this batch; see
 * struct d3m_fit_targets). */
int d3m_fit_loss_forward(const float* rgb, const float* rgb_target, const float* depth, const float* depth_target,
                         const float* alpha, const float* alpha_target, const float* mask, float* loss, float* scratch,
                         const float* mask_sum, int batch_size, int height, int width, d3m_stream_t stream);
int d3m_fit_loss_backward(const float* rgb, const float* rgb_target, const float* depth, const float* depth_target,
                          const float* alpha, const float* alpha_target, const float* mask, const float* scratch,
                          const float* grad_loss, float* grad_rgb, float* grad_depth, float* grad_alpha, int batch_size,
                          int height, int width, d3m_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * D. The renderer block of the gan2shape training step as fused passes (deep3dmap/models/frameworks/gan2shape.py:444,
 *    463-497 through NrRenderer, deep3dmap/core/renderer/renderer_nr.py:61-139):
 *
 *      rot, trans      = get_transform_matrices(view)   (when `view` is given; else rot / trans are inputs)   utils.py:54-71
 *      normal          = get_normal_from_depth(depth)                                           renderer_nr.py:127-139
 *      diffuse_shading = clamp((normal . light_d), min=0);  shading = light_a + light_b * diffuse_shading
 *      texture         = (albedo / 2 + 0.5) * shading * 2 - 1                                   gan2shape.py:463-466
 *      recon_depth     = warp_canon_depth(depth)    the depth map's grid mesh (implicit topology, fill_back) moved by
 *                        (rot, trans), projected by `camera`, rasterized in depth mode, clamped  renderer_nr.py:116-125
 *      recon_im_mask   = (recon_depth < depth_max) [* the other half's with `flip`] [* extra_mask]   gan2shape.py:476-482
 *      recon_im        = clamp(grid_sample(texture, get_inv_warped_2d_grid(recon_depth)), -1, 1)  gan2shape.py:475,483
 *      losses[0]       = photometric_loss(recon_im[:b], target, mask[:b])    (b = batch_size, or half of it with flip)
 *      losses[1]       = photometric_loss(recon_im[b:], target, mask[b:])    (flip only, else 0)    gan2shape.py:486,489
 *      losses[2]       = smooth_loss(depth) + smooth_loss(diffuse_shading)   (with_smooth)          gan2shape.py:493-494
 *      losses[3]       = losses[0] + losses[1] + lam_smooth * losses[2]
 *
 *    d3m_g2s_forward: 4 launches; d3m_g2s_backward: 5 (gradients for depth, albedo, light_a, light_b, light_d and the
 *    view -- as rot / trans and, with `view`, the view vector -- from the gradients of recon_im and of the four loss
 *    values).  The mesh is rasterized by bidding (depth, face) into a 64-bit z-buffer (its triangles are a few pixels
 *    each): same coverage, depth and tie rule as d3m_forward_face_index_map.  Device pointers unless marked. */
typedef struct d3m_g2s_block {
    int batch_size, height, width;       /* canonical maps: depth [B,H,W], albedo [B,3,H,W] */
    int image_size, anti_aliasing;       /* output images [B,.,s,s]; the mesh is rasterized at 2s with anti-aliasing */
    int flip;                            /* entries (b, b + B/2) are an image and its mirror: shared mask product */
    const float* inv_K; int inv_K_batch; /* NrRenderer.inv_K / K, [1|B,3,3] */
    const float* K; int K_batch;
    float rot_center_depth;
    float depth_min, depth_max;          /* recon_depth is clamped to [depth_min, depth_max] (= min_depth - margin,
                                          * max_depth + margin of renderer_nr.py:122-124) */
    float near, far;                     /* of the depth rasterization (the rasterizer's defaults, NR/renderer.py:149) */
    const d3m_camera* camera;            /* HOST: the mesh renderer's camera (renderer_nr.py:47-54) */
    /* inputs */
    const float* view; int view_components;   /* [B, 3|5|6] view vectors, or NULL */
    float *rot, *trans;                  /* [B,3,3], [B,3]: inputs when view == NULL, else written (set_transform_matrices) */
    const float *depth, *albedo;
    const float *light_a, *light_b;      /* [B] */
    const float* light_d;                /* [B,3] */
    const float* target;                 /* [b,3,s,s] input_im, or NULL (no photometric terms) */
    const float* extra_mask;             /* [B,s,s] multiplied into recon_im_mask (gan2shape.py:672), or NULL */
    /* forward outputs (normal, recon_depth may be NULL) */
    float* normal;                       /* [B,H,W,3] */
    float* diffuse_shading;              /* [B,H,W] */
    float* texture;                      /* [B,3,H,W] */
    float* recon_depth;                  /* [B,s,s] */
    float* recon_im;                     /* [B,3,s,s] */
    float* recon_im_mask;                /* [B,s,s] */
    float* losses;                       /* [4] */
    float lam_smooth; int with_smooth;
    /* kept from forward to backward (caller-allocated, no initialisation) */
    float* screen_vertices;              /* [B,H*W,3] */
    uint64_t* zbuffer;                   /* [B,S,S], S = image_size * (anti_aliasing ? 2 : 1) */
    float* scratch;                      /* d3m_g2s_scratch_floats() */
    /* backward inputs: gradient of recon_im [B,3,s,s] and of the four loss values (device scalars); NULL = zero */
    const float *grad_recon_im, *grad_l1, *grad_l1_flip, *grad_smooth, *grad_total;
    /* backward scratch; grad_texture must already be handed to d3m_g2s_forward, which clears it; d3m_g2s_backward
     * hands it back zeroed, so any number of backward passes may follow one forward */
    float* grad_texture;                 /* [B,3,H,W] */
    float* grad_tri;                     /* [B, 2 (H-1)(W-1), 3, 3] */
    float* grad_depth_map;               /* [B,S,S] */
    float* grad_normal;                  /* [B,H,W,3] */
    float* grad_depth_mesh;              /* [B,H,W] */
    /* backward outputs (WRITTEN; any of light / rot / trans / view may be NULL) */
    float *grad_depth, *grad_albedo, *grad_light_a, *grad_light_b, *grad_light_d, *grad_rot, *grad_trans, *grad_view;
} d3m_g2s_block;
size_t d3m_g2s_scratch_floats(int batch_size, int height, int width, int image_size);
int d3m_g2s_forward(const d3m_g2s_block* block, d3m_stream_t stream);
int d3m_g2s_backward(const d3m_g2s_block* block, d3m_stream_t stream);

/* NrRenderer's grid_sample frames (renderer_nr.py:180-184, 219-222, 263-267) in one pass:
 *   grid = ((K (Q / Q.z)).xy / (w-1, h-1)) * 2 - 1,  Q = rot_b (depth * inv_K (x, y, 1) - c) + c + trans_b        (as d3m_grid_warp)
 *   out [B,C,h,w] = F.grid_sample(src [B,C,H,W], grid, mode='bilinear');  out_nearest [B,Cn,h,w] = F.grid_sample(src_nearest,
 *   grid, mode='nearest') (src_nearest may be NULL) -- zeros padding, align_corners = False.  depth [B,h,w] is the target
 *   view's depth (recon_depth), (rot, trans) [B,3,3] / [B,3] the composed rigid motion (the inverse view for the frames).
 * The adjoint ADDS into grad_src [B,C,H,W] (float atomics; caller zeroes; may be NULL) and WRITES grad_depth [B,h,w] (may be
 * NULL) and, per workgroup, the partial sums of the gradient of (rot, trans): partials [B, d3m_warp_resample_partials(h, w), 12]
 * (9 + 3; the caller adds them up). */
int d3m_warp_resample(const float* depth, const float* inv_K, int inv_K_batch, const float* K, int K_batch, const float* rot,
                      const float* trans, float rot_center_depth, const float* src, int channels, const float* src_nearest,
                      int channels_nearest, float* out, float* out_nearest, int batch_size, int height, int width,
                      int src_height, int src_width, d3m_stream_t stream);
int d3m_warp_resample_partials(int height, int width);
int d3m_warp_resample_backward(const float* depth, const float* inv_K, int inv_K_batch, const float* K, int K_batch,
                               const float* rot, const float* trans, float rot_center_depth, const float* src, int channels,
                               const float* grad_out, float* grad_src, float* grad_depth, float* partials, int batch_size,
                               int height, int width, int src_height, int src_width, d3m_stream_t stream);

/* ---- texture assets ----------------------------------------------------------------------------------------- */
/* Replaces load_textures_cuda (NR/cuda/load_textures_cuda.cpp:6-37, kernel load_textures_cuda_kernel.cu:23-114):
 * fills textures [F, ts, ts, ts, 3] of every face with is_update[f] != 0 by sampling image [H, W, 3] at
 * barycentric combinations of the face's uv corners faces_uv [F, 3, 2].  texture_wrapping: 0 REPEAT,
 * 1 MIRRORED_REPEAT, 2 CLAMP_TO_EDGE, 3 CLAMP_TO_BORDER (writes zeros, like the reference).  faces_uv is
 * read-only here (the reference wraps it in place; see DESIGN.md for the one case where that matters). */
int d3m_load_textures(const float* image, const int32_t* is_update, const float* faces_uv, float* textures,
                      int num_faces, int texture_size, int image_height, int image_width, int texture_wrapping,
                      int use_bilinear, d3m_stream_t stream);
/* Replaces create_texture_image_cuda (NR/cuda/create_texture_image_cuda.cpp:6-30, kernels
 * create_texture_image_cuda_kernel.cu:10-115, both launches in one pass): renders textures
 * [F, tsi, tsi, tsi, 3] into the atlas image [image_height, image_width, 3] of tile_width tiles per row
 * (tile edge = image_width / tile_width); vertices_all [F, 3, 2] are the tile-space corners.  Tiles beyond
 * num_faces are zero-filled. */
int d3m_create_texture_image(const float* vertices_all, const float* textures, float* image, int num_faces,
                             int texture_size_in, int image_height, int image_width, int tile_width, float eps,
                             d3m_stream_t stream);

/* ---- the face3d utility rasterizer family (f64) ----------------------------------------------------------------
 * Replace the C++ cores of deep3dmap/core/renderer/renderer_demo/mesh_cython/render.cpp (MC) that render_cython.pyx
 * exposes (:55-161) and render.py drives (:124-299); same argument order and array layouts (vertices [3,nver],
 * triangles [3,ntri] coordinate-major, images [h,w,c], everything double / int32), device pointers, plus a workspace
 * of d3m_mesh_workspace_bytes(nver, ntri, h, w) bytes (no initialisation needed; pass 0 for the dimensions a
 * function does not have) and a stream.  In/out buffers (image, depth_buffer, depth_tmp, vis, triangle_buffer, norm,
 * uv) are pre-filled by the caller as render.py does and updated exactly as the reference's sequential loops would. */
size_t d3m_mesh_workspace_bytes(int nver, int ntri, int h, int w);
/* _render_colors_core, MC:27-91 */
int d3m_mesh_render_colors(double* image, const double* vertices, const int32_t* triangles, const double* tri_depth,
                           const double* tri_tex, double* depth_buffer, int nver, int ntri, int h, int w, int c,
                           void* workspace, size_t workspace_bytes, d3m_stream_t stream);
/* _render_texture_core, MC:94-185 (mapping_type 0 nearest, 1 bilinear; texel indices are clamped) */
int d3m_mesh_render_texture(double* image, const double* vertices, const int32_t* triangles, const double* texture,
                            const double* tex_coords, const int32_t* tex_triangles, const double* tri_depth,
                            double* depth_buffer, int nver, int tex_nver, int ntri, int h, int w, int c, int tex_h,
                            int tex_w, int tex_c, int mapping_type, void* workspace, size_t workspace_bytes,
                            d3m_stream_t stream);
/* _map_texture_core, MC:188-250 */
int d3m_mesh_map_texture(double* dst_image, const double* src_image, const double* dst_vertices,
                         const double* src_vertices, const int32_t* dst_triangle_buffer, const int32_t* triangles,
                         int nver, int ntri, int sh, int sw, int sc, int h, int w, int c, d3m_stream_t stream);
/* _vis_of_vertices_core, MC:253-318 */
int d3m_mesh_vis_of_vertices(double* vis, const double* vertices, const int32_t* triangles, const double* tri_depth,
                             double* depth_buffer, double* depth_tmp, int nver, int ntri, int h, int w, void* workspace,
                             size_t workspace_bytes, d3m_stream_t stream);
/* _get_triangle_buffer_core, MC:321-365 */
int d3m_mesh_get_triangle_buffer(int32_t* triangle_buffer, const double* vertices, const int32_t* triangles,
                                 const double* tri_depth, double* depth_buffer, int nver, int ntri, int h, int w,
                                 void* workspace, size_t workspace_bytes, d3m_stream_t stream);
/* _get_norm_direction_core, MC:4-24 (sums in triangle-index order, as the reference's loop) */
int d3m_mesh_get_norm_direction(double* norm, const double* tri_norm, const int32_t* triangles, int nver, int ntri,
                                void* workspace, size_t workspace_bytes, d3m_stream_t stream);
/* The numpy statements render.py wraps around the cores, with numpy's roundings (no fused multiply-add, true
 * division): per-triangle mean of per-vertex values [channels, nver] -> [channels, ntri] (render.py:141-142),
 * cross(pt0 - pt1, pt0 - pt2) per triangle (:6-9), and the unit-length step with the zero-normal rule (:19-26). */
int d3m_mesh_triangle_mean(const double* values, const int32_t* triangles, double* out, int channels, int nver, int ntri,
                           d3m_stream_t stream);
int d3m_mesh_triangle_normals(const double* vertices, const int32_t* triangles, double* tri_norm, int nver, int ntri,
                              d3m_stream_t stream);
int d3m_mesh_normalize(double* norm, int nver, d3m_stream_t stream);
/* _get_correspondence_core, MC:441-488 */
int d3m_mesh_get_correspondence(const double* image, const double* pncc_code, double* uv, int nver, int h, int w, int c,
                                void* workspace, size_t workspace_bytes, d3m_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* D3M_RASTER_H */
