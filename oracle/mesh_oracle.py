"""
oracle/mesh_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The face3d CPU rasterizer family of the reference (MC = deep3dmap/core/renderer/renderer_demo/mesh_cython/render.cpp,
its numpy front end MP = .../mesh_cython/render.py) behind one interface with two back ends:

  * "ref" : oracle/_ref/libmesh_ref.so -- MC itself, compiled from its own sources by `make -C oracle ref` in the
            build container (plain C++, no stand-ins); travels to the GPU box as a built file;
  * "port": oracle/_build/libmesh_oracle.so -- oracle/mesh_oracle.c, this repo's restatement (builds anywhere).

The functions below restate MP:124-273 / :5-27 (what is computed in numpy before and after the C cores).
Arrays follow the reference: vertices [3, nver] f64, triangles [3, ntri] i32, images [h, w, c] f64.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PORT_SO = os.path.join(_HERE, "_build", "libmesh_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libmesh_ref.so")
_D = ctypes.POINTER(ctypes.c_double)
_Ip = ctypes.POINTER(ctypes.c_int32)
_I = ctypes.c_int
_SIGS = {
    "get_norm_direction": [_D, _D, _Ip, _I, _I],
    "render_colors": [_D, _D, _Ip, _D, _D, _D, _I, _I, _I, _I, _I],
    "render_texture": [_D, _D, _Ip, _D, _D, _Ip, _D, _D] + [_I] * 10,
    "map_texture": [_D, _D, _D, _D, _Ip, _Ip] + [_I] * 8,
    "vis_of_vertices": [_D, _D, _Ip, _D, _D, _D, _I, _I, _I, _I, _I],
    "get_triangle_buffer": [_Ip, _D, _Ip, _D, _D, _I, _I, _I, _I, _I],
    "get_correspondence": [_D, _D, _D, _I, _I, _I, _I],
}
_LIBS = {}


def have_ref():
    return os.path.exists(_REF_SO)


def default_backend():
    """The reference itself where its build is present, the port otherwise."""
    return "ref" if have_ref() else "port"


def _lib(backend):
    if backend not in _LIBS:
        if backend == "port":
            src = os.path.join(_HERE, "mesh_oracle.c")
            if not os.path.exists(_PORT_SO) or os.path.getmtime(_PORT_SO) < os.path.getmtime(src):
                subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
            path, prefix = _PORT_SO, "orc_mesh_"
        elif backend == "ref":
            if not have_ref():
                raise FileNotFoundError("oracle/_ref/libmesh_ref.so not built (run `make -C oracle ref` in the build "
                                        "container)")
            path, prefix = _REF_SO, "ref_mesh_"
        else:
            raise ValueError(backend)
        lib = ctypes.CDLL(path)
        fns = {}
        for name, sig in _SIGS.items():
            fn = getattr(lib, prefix + name)
            fn.argtypes, fn.restype = sig, None
            fns[name] = fn
        _LIBS[backend] = fns
    return _LIBS[backend]


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _pd(a):
    return a.ctypes.data_as(_D)


def _pi(a):
    return a.ctypes.data_as(_Ip)


def _tri_depth(vertices, triangles):
    return (vertices[2, triangles[0, :]] + vertices[2, triangles[1, :]] + vertices[2, triangles[2, :]]) / 3.    # MP:141


def get_norm_direction(vertices, triangles, backend=None):
    """MP:5-27: per-vertex normals = normalised sum of the incident triangles' (unnormalised) normals."""
    f = _lib(backend or default_backend())
    vertices, triangles = _d(vertices), _i(triangles)
    pt0, pt1, pt2 = (vertices[:, triangles[k, :]].T for k in range(3))
    tri_norm = _d(np.cross(pt0 - pt1, pt0 - pt2).T)
    norm = np.zeros_like(vertices)
    f["get_norm_direction"](_pd(norm), _pd(tri_norm), _pi(triangles), vertices.shape[1], triangles.shape[1])
    mag = np.sum(norm ** 2, 0)
    zero_ind = (mag == 0)
    mag[zero_ind] = 1
    norm[0, zero_ind] = np.ones((np.sum(zero_ind)))
    return norm / np.sqrt(mag)


def render_colors(vertices, triangles, colors, h, w, c=3, isBG=False, backend=None):
    """MP:124-150."""
    f = _lib(backend or default_backend())
    vertices, triangles, colors = _d(vertices), _i(triangles), _d(colors)
    image = np.ones((h, w, c)) if isBG else np.zeros((h, w, c))
    depth_buffer = np.zeros([h, w]) - 999999.
    tri_depth = _d(_tri_depth(vertices, triangles))
    tri_tex = _d((colors[:, triangles[0, :]] + colors[:, triangles[1, :]] + colors[:, triangles[2, :]]) / 3.)
    f["render_colors"](_pd(image), _pd(vertices), _pi(triangles), _pd(tri_depth), _pd(tri_tex), _pd(depth_buffer),
                       vertices.shape[1], triangles.shape[1], h, w, c)
    return image


def render_texture(vertices, triangles, texture, tex_coords, tex_triangles, h, w, c=3, mapping_type='nearest', isBG=False,
                   backend=None):
    """MP:153-193 (note MP:187: `tex_nver` is tex_coords.shape[0])."""
    f = _lib(backend or default_backend())
    vertices, triangles, texture = _d(vertices), _i(triangles), _d(texture)
    tex_coords, tex_triangles = _d(tex_coords), _i(tex_triangles)
    image = np.ones((h, w, c)) if isBG else np.zeros((h, w, c))
    depth_buffer = np.zeros([h, w]) - 999999.
    tri_depth = _d(_tri_depth(vertices, triangles))
    tex_h, tex_w, tex_c = texture.shape
    mt = 1 if mapping_type == 'bilinear' else 0
    f["render_texture"](_pd(image), _pd(vertices), _pi(triangles), _pd(texture), _pd(tex_coords), _pi(tex_triangles),
                        _pd(tri_depth), _pd(depth_buffer), vertices.shape[1], tex_coords.shape[0], triangles.shape[1], h, w,
                        c, tex_h, tex_w, tex_c, mt)
    return image


def map_texture(src_image, src_vertices, dst_vertices, dst_triangle_buffer, triangles, h, w, c=3, mapping_type='bilinear',
                backend=None):
    """MP:196-220."""
    f = _lib(backend or default_backend())
    src_image, src_vertices, dst_vertices = _d(src_image), _d(src_vertices), _d(dst_vertices)
    dst_triangle_buffer, triangles = _i(dst_triangle_buffer), _i(triangles)
    dst_image = np.zeros((h, w, c))
    f["map_texture"](_pd(dst_image), _pd(src_image), _pd(dst_vertices), _pd(src_vertices), _pi(dst_triangle_buffer),
                     _pi(triangles), src_vertices.shape[1], triangles.shape[1], src_image.shape[0], src_image.shape[1],
                     src_image.shape[2], h, w, c)
    return dst_image


def vis_of_vertices(vertices, triangles, h, w, backend=None):
    """MP:224-245."""
    f = _lib(backend or default_backend())
    vertices, triangles = _d(vertices), _i(triangles)
    vis = np.zeros(vertices.shape[1])
    depth_buffer = np.zeros([h, w]) - 999999.
    depth_tmp = np.zeros_like(depth_buffer) - 9999
    tri_depth = _d(_tri_depth(vertices, triangles))
    f["vis_of_vertices"](_pd(vis), _pd(vertices), _pi(triangles), _pd(tri_depth), _pd(depth_buffer), _pd(depth_tmp),
                         vertices.shape[1], triangles.shape[1], h, w, 1)
    return vis


def get_triangle_buffer(vertices, triangles, h, w, backend=None):
    """MP:249-275."""
    f = _lib(backend or default_backend())
    vertices, triangles = _d(vertices), _i(triangles)
    depth_buffer = np.zeros([h, w]) - 999999.
    triangle_buffer = np.zeros_like(depth_buffer, dtype=np.int32) - 1
    tri_depth = _d(_tri_depth(vertices, triangles))
    f["get_triangle_buffer"](_pi(triangle_buffer), _pd(vertices), _pi(triangles), _pd(tri_depth), _pd(depth_buffer),
                             vertices.shape[1], triangles.shape[1], h, w, 1)
    return triangle_buffer


def get_correspondence(image, pncc_code, backend=None):
    """MP:288-299."""
    f = _lib(backend or default_backend())
    image, pncc_code = _d(image).copy(), _d(pncc_code).copy()
    nver = pncc_code.shape[1]
    h, w, c = image.shape
    uv = np.zeros((2, nver))
    f["get_correspondence"](_pd(image), _pd(pncc_code), _pd(uv), nver, h, w, c)
    X_ind = np.nonzero(uv)[1]
    x = uv[:, X_ind]
    x[1, :] = h - 1 - x[1, :]
    return x, X_ind
