"""The numpy front end of the reference's face3d rasterizer family
(deep3dmap/core/renderer/renderer_demo/mesh_cython/render.py, MP), with the C++ cores it calls through
render_cython replaced by the HIP entry points `d3m_mesh_*` of libd3m_raster.so.

Same function names, arguments, array layouts (vertices [3, nver], triangles [3, ntri], images [h, w, c], float64 /
int32) and return values as MP.  Inputs may be numpy arrays (as in the reference; results come back as numpy) or CUDA
tensors (results stay on the device).  There is no CPU path: without a GPU and the HIP library these functions raise.
"""
import numpy as np
import torch

from .. import _lib

_workspace = {}


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("deep3dmap_amd.mesh_cython runs on the GPU (libd3m_raster.so): no GPU found")
    return torch.device("cuda")


def _in(a, dtype):
    """-> (contiguous device tensor of `dtype`, came_from_numpy)"""
    if torch.is_tensor(a):
        if not a.is_cuda:
            raise RuntimeError("tensor arguments must be CUDA tensors (numpy arrays are uploaded)")
        return a.to(dtype).contiguous(), False
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64 if dtype == torch.float64 else np.int32)).to(_dev()), True


def _out(t, as_numpy):
    return t.cpu().numpy() if as_numpy else t


def _ws(nver, ntri, h, w):
    n = _lib.lib().d3m_mesh_workspace_bytes(int(nver), int(ntri), int(h), int(w))
    buf = _workspace.get("ws")
    if buf is None or buf.numel() < n:
        buf = torch.empty(int(n), dtype=torch.uint8, device=_dev())
        _workspace["ws"] = buf
    return buf


def _tri_mean(values, triangles):
    """(values[:, t0] + values[:, t1] + values[:, t2]) / 3. with numpy's rounding (MP:141-142)."""
    ch, nver = values.shape
    out = torch.empty((ch, triangles.shape[1]), dtype=torch.float64, device=values.device)
    _lib.check(_lib.lib().d3m_mesh_triangle_mean(_lib.ptr(values), _lib.ptr(triangles), _lib.ptr(out), ch, nver,
                                                 triangles.shape[1], _lib.stream_ptr()), "d3m_mesh_triangle_mean")
    return out


def _tri_depth(vertices, triangles):
    return _tri_mean(vertices[2:3].contiguous(), triangles).reshape(-1)                             # MP:141


def get_norm_direction(vertices, triangles):
    """MP:5-27: unit per-vertex normals = normalised sum of the incident triangles' normals (summed on the device in
    triangle order, like the reference's loop)."""
    v, np_in = _in(vertices, torch.float64)
    t, _ = _in(triangles, torch.int32)
    L = _lib.lib()
    tri_norm = torch.empty((3, t.shape[1]), dtype=torch.float64, device=v.device)
    _lib.check(L.d3m_mesh_triangle_normals(_lib.ptr(v), _lib.ptr(t), _lib.ptr(tri_norm), v.shape[1], t.shape[1],
                                           _lib.stream_ptr()), "d3m_mesh_triangle_normals")
    norm = torch.zeros_like(v)
    ws = _ws(v.shape[1], t.shape[1], 0, 0)
    _lib.check(L.d3m_mesh_get_norm_direction(_lib.ptr(norm), _lib.ptr(tri_norm), _lib.ptr(t), v.shape[1], t.shape[1],
                                             _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "d3m_mesh_get_norm_direction")
    _lib.check(L.d3m_mesh_normalize(_lib.ptr(norm), v.shape[1], _lib.stream_ptr()), "d3m_mesh_normalize")
    return _out(norm, np_in)


def render_colors(vertices, triangles, colors, h, w, c=3, isBG=False):
    ''' render mesh with colors by z buffer (MP:124-150)
    Args:
        vertices: 3 x nver
        colors: 3 x nver
        triangles: 3 x ntri
        h: height
        w: width
    '''
    v, np_in = _in(vertices, torch.float64)
    t, _ = _in(triangles, torch.int32)
    col, _ = _in(colors, torch.float64)
    image = torch.ones((h, w, c), dtype=torch.float64, device=v.device) if isBG else \
        torch.zeros((h, w, c), dtype=torch.float64, device=v.device)
    depth_buffer = torch.full((h, w), -999999., dtype=torch.float64, device=v.device)
    tri_tex = _tri_mean(col, t)
    tri_depth = _tri_depth(v, t)
    ws = _ws(v.shape[1], t.shape[1], h, w)
    _lib.check(_lib.lib().d3m_mesh_render_colors(_lib.ptr(image), _lib.ptr(v), _lib.ptr(t), _lib.ptr(tri_depth),
                                                 _lib.ptr(tri_tex), _lib.ptr(depth_buffer), v.shape[1], t.shape[1], h, w, c,
                                                 _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "d3m_mesh_render_colors")
    return _out(image, np_in)


def render_texture(vertices, triangles, texture, tex_coords, tex_triangles, h, w, c=3, mapping_type='nearest', isBG=False):
    ''' render mesh with texture map by z buffer (MP:153-193)
    Args:
        vertices: 3 x nver
        triangles: 3 x ntri
        texture: tex_h x tex_w x 3
        tex_coords: 2 x ntexver
        tex_triangles: 3 x ntri
    '''
    v, np_in = _in(vertices, torch.float64)
    t, _ = _in(triangles, torch.int32)
    tex, _ = _in(texture, torch.float64)
    tc, _ = _in(tex_coords, torch.float64)
    tt, _ = _in(tex_triangles, torch.int32)
    image = torch.ones((h, w, c), dtype=torch.float64, device=v.device) if isBG else \
        torch.zeros((h, w, c), dtype=torch.float64, device=v.device)
    depth_buffer = torch.full((h, w), -999999., dtype=torch.float64, device=v.device)
    tri_depth = _tri_depth(v, t)
    tex_h, tex_w, tex_c = tex.shape
    mt = 1 if mapping_type == 'bilinear' else 0
    # MP:187 passes tex_coords.shape[0] as `tex_nver`; the reads stay inside tex_coords only if it is this small
    tex_nver = tc.shape[0]
    if tc.numel() < tex_nver + v.shape[1]:
        raise ValueError("tex_coords is too small for the reference's indexing (tex_coords[shape[0] + vertex index])")
    ws = _ws(v.shape[1], t.shape[1], h, w)
    _lib.check(_lib.lib().d3m_mesh_render_texture(
        _lib.ptr(image), _lib.ptr(v), _lib.ptr(t), _lib.ptr(tex), _lib.ptr(tc), _lib.ptr(tt), _lib.ptr(tri_depth),
        _lib.ptr(depth_buffer), v.shape[1], tex_nver, t.shape[1], h, w, c, tex_h, tex_w, tex_c, mt, _lib.ptr(ws), ws.numel(),
        _lib.stream_ptr()), "d3m_mesh_render_texture")
    return _out(image, np_in)


def map_texture(src_image, src_vertices, dst_vertices, dst_triangle_buffer, triangles, h, w, c=3, mapping_type='bilinear'):
    '''
    Args:
        triangles: 3 x ntri
        src_image: height x width x nchannels
        src_vertices: 3 x nver
        dst_vertices: 3 x nver
        dst_triangle_buffer: height x width. the triangle index of each pixel in dst image
    Returns:
        dst_image: height x width x nchannels   (MP:196-220)
    '''
    src, np_in = _in(src_image, torch.float64)
    sv, _ = _in(src_vertices, torch.float64)
    dv, _ = _in(dst_vertices, torch.float64)
    tb, _ = _in(dst_triangle_buffer, torch.int32)
    t, _ = _in(triangles, torch.int32)
    dst = torch.zeros((h, w, c), dtype=torch.float64, device=src.device)
    _lib.check(_lib.lib().d3m_mesh_map_texture(_lib.ptr(dst), _lib.ptr(src), _lib.ptr(dv), _lib.ptr(sv), _lib.ptr(tb), _lib.ptr(t),
                                               sv.shape[1], t.shape[1], src.shape[0], src.shape[1], src.shape[2], h, w, c,
                                               _lib.stream_ptr()), "d3m_mesh_map_texture")
    return _out(dst, np_in)


def vis_of_vertices(vertices, triangles, h, w):
    '''
    Returns:
        vertices_vis: nver. the visibility of each vertex   (MP:224-245)
    '''
    v, np_in = _in(vertices, torch.float64)
    t, _ = _in(triangles, torch.int32)
    vis = torch.zeros(v.shape[1], dtype=torch.float64, device=v.device)
    depth_buffer = torch.full((h, w), -999999., dtype=torch.float64, device=v.device)
    depth_tmp = torch.full((h, w), -9999., dtype=torch.float64, device=v.device)
    tri_depth = _tri_depth(v, t)
    ws = _ws(v.shape[1], t.shape[1], h, w)
    _lib.check(_lib.lib().d3m_mesh_vis_of_vertices(_lib.ptr(vis), _lib.ptr(v), _lib.ptr(t), _lib.ptr(tri_depth),
                                                   _lib.ptr(depth_buffer), _lib.ptr(depth_tmp), v.shape[1], t.shape[1], h, w,
                                                   _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "d3m_mesh_vis_of_vertices")
    return _out(vis, np_in)


def get_triangle_buffer(vertices, triangles, h, w):
    '''
    Returns:
        triangle_buffer: height x width, the index of the triangle drawn at each pixel, -1 for none (MP:249-275).
        Here, the bigger the z, the fronter the point.
    '''
    v, np_in = _in(vertices, torch.float64)
    t, _ = _in(triangles, torch.int32)
    depth_buffer = torch.full((h, w), -999999., dtype=torch.float64, device=v.device)
    triangle_buffer = torch.full((h, w), -1, dtype=torch.int32, device=v.device)
    tri_depth = _tri_depth(v, t)
    ws = _ws(v.shape[1], t.shape[1], h, w)
    _lib.check(_lib.lib().d3m_mesh_get_triangle_buffer(_lib.ptr(triangle_buffer), _lib.ptr(v), _lib.ptr(t), _lib.ptr(tri_depth),
                                                       _lib.ptr(depth_buffer), v.shape[1], t.shape[1], h, w, _lib.ptr(ws),
                                                       ws.numel(), _lib.stream_ptr()), "d3m_mesh_get_triangle_buffer")
    return _out(triangle_buffer, np_in)


def get_correspondence(image, pncc_code):
    """MP:288-299: (pixel coordinates [2, n] with the row flipped, vertex indices [n]) of the vertices whose PNCC code
    is the nearest one of some pixel."""
    img, np_in = _in(image, torch.float64)
    code, _ = _in(pncc_code, torch.float64)
    nver = code.shape[1]
    h, w, c = img.shape
    uv = torch.zeros((2, nver), dtype=torch.float64, device=img.device)
    ws = _ws(nver, 0, 0, 0)
    _lib.check(_lib.lib().d3m_mesh_get_correspondence(_lib.ptr(img), _lib.ptr(code), _lib.ptr(uv), nver, h, w, c, _lib.ptr(ws),
                                                      ws.numel(), _lib.stream_ptr()), "d3m_mesh_get_correspondence")
    X_ind = torch.nonzero(uv)[:, 1]
    # np.nonzero(uv)[1] lists a column once per non-zero ROW entry, in row-major order (MP:296)
    x = uv[:, X_ind]
    x[1, :] = h - 1 - x[1, :]
    return _out(x, np_in), _out(X_ind, np_in)
