// extern "C" doors onto the reference's own face3d rasterizer functions
// (deep3dmap/core/renderer/renderer_demo/mesh_cython/render.cpp, declared in its render.h which is found through -I
// at build time).  No reference code lives here: `make -C oracle ref` compiles render.cpp where it lies and links it
// with this file into oracle/_ref/libmesh_ref.so.  TEST INFRASTRUCTURE.
#include "render.h"
#define DOOR extern "C" __attribute__((visibility("default")))

DOOR void ref_mesh_get_norm_direction(double* norm, double* tri_norm, int* triangles, int nver, int ntri) {
    _get_norm_direction_core(norm, tri_norm, triangles, nver, ntri);
}
DOOR void ref_mesh_render_colors(double* image, double* vertices, int* triangles, double* tri_depth, double* tri_tex,
                                 double* depth_buffer, int nver, int ntri, int h, int w, int c) {
    _render_colors_core(image, vertices, triangles, tri_depth, tri_tex, depth_buffer, nver, ntri, h, w, c);
}
DOOR void ref_mesh_render_texture(double* image, double* vertices, int* triangles, double* texture, double* tex_coords,
                                  int* tex_triangles, double* tri_depth, double* depth_buffer, int nver, int tex_nver,
                                  int ntri, int h, int w, int c, int tex_h, int tex_w, int tex_c, int mapping_type) {
    _render_texture_core(image, vertices, triangles, texture, tex_coords, tex_triangles, tri_depth, depth_buffer, nver,
                         tex_nver, ntri, h, w, c, tex_h, tex_w, tex_c, mapping_type);
}
DOOR void ref_mesh_map_texture(double* dst_image, double* src_image, double* dst_vertices, double* src_vertices,
                               int* dst_triangle_buffer, int* triangles, int nver, int ntri, int sh, int sw, int sc, int h,
                               int w, int c) {
    _map_texture_core(dst_image, src_image, dst_vertices, src_vertices, dst_triangle_buffer, triangles, nver, ntri, sh, sw,
                      sc, h, w, c);
}
DOOR void ref_mesh_vis_of_vertices(double* vis, double* vertices, int* triangles, double* tri_depth, double* depth_buffer,
                                   double* depth_tmp, int nver, int ntri, int h, int w, int c) {
    _vis_of_vertices_core(vis, vertices, triangles, tri_depth, depth_buffer, depth_tmp, nver, ntri, h, w, c);
}
DOOR void ref_mesh_get_triangle_buffer(int* triangle_buffer, double* vertices, int* triangles, double* tri_depth,
                                       double* depth_buffer, int nver, int ntri, int h, int w, int c) {
    _get_triangle_buffer_core(triangle_buffer, vertices, triangles, tri_depth, depth_buffer, nver, ntri, h, w, c);
}
DOOR void ref_mesh_get_correspondence(double* image, double* pncc_code, double* uv, int nver, int h, int w, int c) {
    _get_correspondence_core(image, pncc_code, uv, nver, h, w, c);
}
