"""
oracle/nr_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's differentiable-rasterization path, used only
as the checker in tests/, in __graft_entry__.smoke() and as bench.py's
cpu_baseline leg.  Nothing under deep3dmap_amd/ imports this module.

Two kernel back ends with identical signatures are reachable from here:

  * "port": oracle/nr_oracle.c (this repo's plain-C restatement; travels to the GPU box),
  * "hip" : oracle/_ref/libnr_ref_hip.so -- THE REFERENCE'S OWN KERNELS ON THE DEVICE: the text of
            rasterize_cuda_kernel.cu:22-593 (and of the two texture kernels) compiled where it lies by
            hipcc against HIP's real headers, -ffp-contract=off (oracle/Makefile, target `ref_hip`;
            launch drivers in oracle/ref_hipbuild/).  Needs a GPU.  This is the pin of K2/K3/K5/K6: it
            validates "port", generates tests/golden/*kern*.npz on the GPU box and is what the -m gpu
            parity tests compare the product with, at fixture and at full size.
  * "hip_fma": the same text with the compiler's default FMA contraction (what a stock build of the
            reference would do); compared within north_star's tolerances, not bit for bit.

The torch (CPU) glue below restates, with citations, the Python half of the reference path
(NR = pnpmodules/neural_renderer/neural_renderer, CR = deep3dmap/core/renderer):
RasterizeFunction (NR/rasterize.py:15-226), rasterize_rgbad (NR/rasterize.py:253-334), the camera
functions (NR/look_at.py, NR/look.py, NR/perspective.py, NR/projection.py,
NR/get_points_from_angles.py), vertices_to_faces, lighting, Renderer (NR/renderer.py:11-246) and the
NrRenderer helpers and losses (CR/utils.py:22-107, CR/renderer_nr.py:35-125,
deep3dmap/core/utils/utils.py:82-114).
"""
import ctypes
import math
import os
import subprocess

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_PORT_SO = os.path.join(_HERE, "_build", "libnr_oracle.so")
_HIP_SO = {"hip": os.path.join(_HERE, "_ref", "libnr_ref_hip.so"),
           "hip_fma": os.path.join(_HERE, "_ref", "libnr_ref_hip_fma.so")}

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_I, _FL = ctypes.c_int, ctypes.c_float

_SIGS = {
    "face_inverse": [_f32p, _f32p, _I, _I, _I],
    "face_index_map": [_f32p, _f32p, _i32p, _f32p, _f32p, _f32p, _I, _I, _I, _FL, _FL, _I],
    "texture_sampling": [_f32p, _f32p, _i32p, _f32p, _f32p, _f32p, _i32p, _f32p, _I, _I, _I, _I, _FL],
    "backward_pixel_map": [_f32p, _i32p, _f32p, _f32p, _f32p, _f32p, _f32p, _I, _I, _I, _FL, _I, _I],
    "backward_textures": [_i32p, _f32p, _i32p, _f32p, _f32p, _I, _I, _I, _I],
    "backward_depth_map": [_f32p, _f32p, _i32p, _f32p, _f32p, _f32p, _f32p, _I, _I, _I],
    "load_textures": [_f32p, _i32p, _f32p, _f32p, _I, _I, _I, _I, _I, _I],
    "create_texture_image": [_f32p, _f32p, _f32p, _I, _I, _I, _I, _I, _FL],
}


def build_port(force=False):
    """Compile oracle/nr_oracle.c with gcc (also done by __graft_entry__.build())."""
    if force or not os.path.exists(_PORT_SO) or os.path.getmtime(_PORT_SO) < os.path.getmtime(
            os.path.join(_HERE, "nr_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    return _PORT_SO


class _Kernels:
    """ctypes view of one back end ('port' -> orc_*, 'ref' -> ref_*)."""

    def __init__(self, backend):
        self.backend = backend
        if backend == "port":
            path, prefix = build_port(), "orc_"
        elif backend in _HIP_SO:
            path, prefix = _HIP_SO[backend], "refhip_"
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path} not built (run `make -C oracle ref_hip` in the build container)")
        else:
            raise ValueError(backend)
        self.lib = ctypes.CDLL(path)
        for name, sig in _SIGS.items():
            fn = getattr(self.lib, prefix + name)
            fn.argtypes, fn.restype = sig, None
            setattr(self, name, fn)
        if backend == "port":
            self.face_index_map_bbox = self.lib.orc_face_index_map_bbox
            self.face_index_map_bbox.argtypes, self.face_index_map_bbox.restype = _SIGS["face_index_map"], None
            self.face_index_map_range = self.lib.orc_face_index_map_range
            self.face_index_map_range.argtypes = _SIGS["face_index_map"] + [ctypes.c_long, ctypes.c_long]
            self.face_index_map_range.restype = None
            self.lib.orc_camera_basis.argtypes = [_f32p, _I, _f32p, _I, _f32p, _I, _I, _f32p, _I]
            self.lib.orc_camera_points.argtypes = [_f32p, _I, _I, _I, _FL, _FL, _f32p, _I, _f32p, _I, _f32p, _I, _f32p, _I,
                                                   _f32p, _I, _I]
            self.lib.orc_depth_to_vertices.argtypes = [_f32p, _f32p, _I, _f32p, _f32p, _FL, _f32p, _I, _I, _I]
            for f in (self.lib.orc_camera_basis, self.lib.orc_camera_points, self.lib.orc_depth_to_vertices):
                f.restype = None
            self.lib.orc_num_threads.restype = _I
            self.lib.orc_set_num_threads.argtypes = [_I]


_KERNELS = {}


def kernels(backend="port"):
    if backend not in _KERNELS:
        _KERNELS[backend] = _Kernels(backend)
    return _KERNELS[backend]


def have_ref_hip():
    """The on-device reference build is present (it was built where /root/reference exists and travels as a .so)."""
    return all(os.path.exists(p) for p in _HIP_SO.values())


def num_threads():
    return kernels("port").lib.orc_num_threads()


def set_num_threads(n):
    kernels("port").lib.orc_set_num_threads(int(n))


def _fp(a):
    return a.ctypes.data_as(_f32p)


def _ip(a):
    return a.ctypes.data_as(_i32p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# --------------------------------------------------------------------------------------------
# numpy-level rasterizer: forward maps and backward, mirroring RasterizeFunction
# --------------------------------------------------------------------------------------------
def raster_forward(faces, textures, image_size, near, far, eps, background_color, return_rgb, return_alpha,
                   return_depth, backend="port", bbox=False):
    """Restates RasterizeFunction.forward (NR/rasterize.py:22-98) on numpy arrays.

    faces [B,F,3,3] f32, textures [B,F,ts,ts,ts,3] f32 or None.  Returns a dict with every
    intermediate map the reference saves for backward, with the reference's initial values
    (NR/rasterize.py:50-69): face_index -1, weight 0, depth far, rgb 0, sampling 0, alpha 0, face_inv 0.
    """
    k = kernels(backend)
    faces = _f32(faces)
    B, Fn = faces.shape[:2]
    S = int(image_size)
    m = {"faces": faces, "image_size": S, "near": near, "far": far, "eps": eps}
    m["face_index_map"] = np.full((B, S, S), -1, np.int32)
    m["weight_map"] = np.zeros((B, S, S, 3), np.float32)
    m["depth_map"] = np.full((B, S, S), far, np.float32)
    m["face_inv_map"] = np.zeros((B, S, S, 3, 3), np.float32) if return_depth else np.zeros(1, np.float32)
    faces_inv = np.zeros_like(faces)                                      # NR/rasterize.py:161
    k.face_inverse(_fp(faces), _fp(faces_inv), B, Fn, S)
    fim = k.face_index_map_bbox if (bbox and backend == "port") else k.face_index_map
    fim(_fp(faces), _fp(faces_inv), _ip(m["face_index_map"]), _fp(m["weight_map"]), _fp(m["depth_map"]),
        _fp(m["face_inv_map"]), B, Fn, S, near, far, int(bool(return_depth)))
    m["faces_inv"] = faces_inv
    if return_rgb:
        textures = _f32(textures)
        ts = textures.shape[2]
        m["textures"] = textures
        m["rgb_map"] = np.zeros((B, S, S, 3), np.float32)
        m["sampling_index_map"] = np.zeros((B, S, S, 8), np.int32)
        m["sampling_weight_map"] = np.zeros((B, S, S, 8), np.float32)
        k.texture_sampling(_fp(faces), _fp(textures), _ip(m["face_index_map"]), _fp(m["weight_map"]),
                           _fp(m["depth_map"]), _fp(m["rgb_map"]), _ip(m["sampling_index_map"]),
                           _fp(m["sampling_weight_map"]), B, Fn, S, ts, eps)
        # forward_background, NR/rasterize.py:187-195
        bg = np.asarray(background_color, np.float32)
        mask = (m["face_index_map"] >= 0).astype(np.float32)[..., None]
        bgb = bg[None, None, None, :] if bg.ndim == 1 else bg[:, None, None, :]
        m["rgb_map"] = (m["rgb_map"] * mask + (1 - mask) * bgb).astype(np.float32)
    if return_alpha:
        m["alpha_map"] = (m["face_index_map"] >= 0).astype(np.float32)   # NR/rasterize.py:181-184
    return m


def raster_backward(m, grad_rgb_map, grad_alpha_map, grad_depth_map, return_rgb, return_alpha, return_depth,
                    backend="port"):
    """Restates RasterizeFunction.backward (NR/rasterize.py:100-156): K4 (overwrite) -> K5 -> K6 (add)."""
    k = kernels(backend)
    faces = m["faces"]
    B, Fn = faces.shape[:2]
    S = m["image_size"]
    grad_faces = np.zeros_like(faces)
    grad_textures = None
    dummy = np.zeros(1, np.float32)
    if return_rgb or return_alpha:
        rgb = m["rgb_map"] if return_rgb else dummy
        alpha = m["alpha_map"] if return_alpha else dummy
        g_rgb = _f32(grad_rgb_map) if return_rgb else dummy
        g_alpha = _f32(grad_alpha_map) if return_alpha else dummy
        k.backward_pixel_map(_fp(faces), _ip(m["face_index_map"]), _fp(rgb), _fp(alpha), _fp(g_rgb), _fp(g_alpha),
                             _fp(grad_faces), B, Fn, S, m["eps"], int(bool(return_rgb)), int(bool(return_alpha)))
    if return_rgb:
        ts = m["textures"].shape[2]
        grad_textures = np.zeros_like(m["textures"])
        g_rgb = _f32(grad_rgb_map)
        k.backward_textures(_ip(m["face_index_map"]), _fp(m["sampling_weight_map"]), _ip(m["sampling_index_map"]),
                            _fp(g_rgb), _fp(grad_textures), B, Fn, S, ts)
    if return_depth:
        g_depth = _f32(grad_depth_map)
        k.backward_depth_map(_fp(faces), _fp(m["depth_map"]), _ip(m["face_index_map"]), _fp(m["face_inv_map"]),
                             _fp(m["weight_map"]), _fp(g_depth), _fp(grad_faces), B, Fn, S)
    return grad_faces, grad_textures


# --------------------------------------------------------------------------------------------
# texture assets (NR/load_obj.py:33-98, NR/save_obj.py:10-38 and their kernels)
# --------------------------------------------------------------------------------------------
TEXTURE_WRAPPING = {'REPEAT': 0, 'MIRRORED_REPEAT': 1, 'CLAMP_TO_EDGE': 2, 'CLAMP_TO_BORDER': 3}


def load_textures_np(image, faces_uv, textures, is_update, texture_wrapping=0, use_bilinear=True, backend="port"):
    """Kernel of NR/load_obj.py:92-97: image [H,W,3] (already flipped), faces_uv [F,3,2], textures [F,ts,ts,ts,3]
    (updated in place for faces with is_update != 0, returned)."""
    k = kernels(backend)
    image, faces_uv = _f32(image), _f32(faces_uv).copy()
    textures = _f32(textures)
    is_update = np.ascontiguousarray(is_update, dtype=np.int32)
    k.load_textures(_fp(image), _ip(is_update), _fp(faces_uv), _fp(textures), textures.shape[0], textures.shape[1],
                    image.shape[0], image.shape[1], int(texture_wrapping), int(bool(use_bilinear)))
    return textures


def create_texture_image_np(textures, texture_size_out=16, backend="port"):
    """NR/save_obj.py:10-38: texture atlas [tile_h*tso, tile_w*tso, 3] (rows flipped on return) and the per-face
    atlas uv coordinates [F,3,2] in [0,1]."""
    k = kernels(backend)
    textures = _f32(textures)
    num_faces, tsi = textures.shape[:2]
    tile_width = int((num_faces - 1.) ** 0.5) + 1
    tile_height = int((num_faces - 1.) / tile_width) + 1
    tso = texture_size_out
    image = np.zeros((tile_height * tso, tile_width * tso, 3), np.float32)
    vertices = np.zeros((num_faces, 3, 2), np.float32)
    fn = np.arange(num_faces)
    column, row = fn % tile_width, fn // tile_width
    vertices[:, 0, 0] = column * tso
    vertices[:, 0, 1] = row * tso
    vertices[:, 1, 0] = column * tso
    vertices[:, 1, 1] = (row + 1) * tso - 1
    vertices[:, 2, 0] = (column + 1) * tso - 1
    vertices[:, 2, 1] = (row + 1) * tso - 1
    # the reference kernel reads tile_width*tile_height faces (padding tiles are out of bounds there): hand it
    # padded copies so the read is defined; callers compare only the tiles of real faces.
    n_tiles = tile_width * tile_height
    vpad = np.zeros((n_tiles, 3, 2), np.float32)
    vpad[:num_faces] = vertices
    tpad = np.zeros((n_tiles,) + textures.shape[1:], np.float32)
    tpad[:num_faces] = textures
    k.create_texture_image(_fp(vpad), _fp(tpad), _fp(image), num_faces, tsi, image.shape[0], image.shape[1],
                           tile_width, 1e-5)
    vertices[:, :, 0] /= (image.shape[1] - 1)
    vertices[:, :, 1] /= (image.shape[0] - 1)
    return image[::-1].copy(), vertices


def texture_atlas_valid_mask(num_faces, texture_size_out, image_shape):
    """Boolean [H,W] mask (in the flipped, returned orientation) of atlas pixels that belong to a real face."""
    tile_width = int((num_faces - 1.) ** 0.5) + 1
    H, W = image_shape[:2]
    y, x = np.mgrid[0:H, 0:W]
    fn = x // texture_size_out + (y // texture_size_out) * tile_width
    return (fn < num_faces)[::-1].copy()


# --------------------------------------------------------------------------------------------
# torch (CPU) autograd wrapper and the Python half of the path
# --------------------------------------------------------------------------------------------
class OracleRasterizeFunction(torch.autograd.Function):
    backend = "port"
    bbox = False

    @staticmethod
    def forward(ctx, faces, textures, image_size, near, far, eps, background_color, return_rgb, return_alpha,
                return_depth):
        m = raster_forward(faces.detach().numpy(), None if textures is None else textures.detach().numpy(),
                           image_size, near, far, eps, background_color, return_rgb, return_alpha, return_depth,
                           backend=OracleRasterizeFunction.backend, bbox=OracleRasterizeFunction.bbox)
        ctx.maps, ctx.flags = m, (return_rgb, return_alpha, return_depth)
        ctx.tex_needs_grad = textures is not None and textures.requires_grad
        rgb = torch.from_numpy(m["rgb_map"]) if return_rgb else torch.tensor([])
        alpha = torch.from_numpy(m["alpha_map"].copy()) if return_alpha else torch.tensor([])
        depth = torch.from_numpy(m["depth_map"].copy()) if return_depth else torch.tensor([])
        return rgb, alpha, depth

    @staticmethod
    def backward(ctx, g_rgb, g_alpha, g_depth):
        rr, ra, rd = ctx.flags
        gf, gt = raster_backward(ctx.maps, g_rgb.contiguous().numpy() if rr else None,
                                 g_alpha.contiguous().numpy() if ra else None,
                                 g_depth.contiguous().numpy() if rd else None, rr, ra, rd,
                                 backend=OracleRasterizeFunction.backend)
        gt_t = torch.from_numpy(gt) if (gt is not None and ctx.tex_needs_grad) else None
        return (torch.from_numpy(gf), gt_t) + (None,) * 8


def rasterize_rgbad(faces, textures=None, image_size=256, anti_aliasing=True, near=0.1, far=100, eps=1e-4,
                    background_color=(0, 0, 0), return_rgb=True, return_alpha=True, return_depth=True):
    """NR/rasterize.py:253-334: 2x supersample, CHW permute, vertical flip, 2x2 average pool."""
    S = image_size * 2 if anti_aliasing else image_size
    rgb, alpha, depth = OracleRasterizeFunction.apply(faces, textures, S, near, far, eps, background_color,
                                                      return_rgb, return_alpha, return_depth)
    if return_rgb:
        rgb = rgb.permute(0, 3, 1, 2).flip(2)
    if return_alpha:
        alpha = alpha.flip(1)
    if return_depth:
        depth = depth.flip(1)
    if anti_aliasing:
        if return_rgb:
            rgb = F.avg_pool2d(rgb, 2)
        if return_alpha:
            alpha = F.avg_pool2d(alpha[:, None], 2)[:, 0]
        if return_depth:
            depth = F.avg_pool2d(depth[:, None], 2)[:, 0]
    return {"rgb": rgb if return_rgb else None, "alpha": alpha if return_alpha else None,
            "depth": depth if return_depth else None}


def _as_vec(x, like, batch=None):
    t = torch.as_tensor(np.asarray(x, np.float32) if not torch.is_tensor(x) else x, dtype=torch.float32)
    if t.dim() == 1 and batch is not None:
        t = t[None, :].repeat(batch, 1)
    return t


def _camera_basis(z_dir, up):
    """Orthonormal rows (x, y, z) as in NR/look_at.py:48-53 / NR/look.py:39-44 (F.normalize eps 1e-5)."""
    z_axis = F.normalize(z_dir, eps=1e-5)
    x_axis = F.normalize(torch.cross(up, z_axis, dim=-1), eps=1e-5)
    y_axis = F.normalize(torch.cross(z_axis, x_axis, dim=-1), eps=1e-5)
    return torch.stack((x_axis, y_axis, z_axis), dim=1)


# EXACT-ASSOCIATION FORWARD VALUES.  The torch formulas below restate the reference's camera modules; their VALUES
# depend on how the tensor library associates the f32 sums.  With EXACT = True the forward values of look_at / look /
# perspective / projection (inside Renderer) and of NrRenderer's depth -> vertices come from oracle/nr_oracle.c instead
# (orc_camera_*, orc_depth_to_vertices: one fixed association, the one the HIP kernels use), so that the oracle and the
# product rasterize bit-identical screen-space vertices and gradients can be compared at 1e-3; the GRADIENT stays
# torch's autograd of the formulas below.  Both forms are checked against the reference modules' outputs
# (tests/test_oracle.py).
EXACT = True


class _ExactForward(torch.autograd.Function):
    """value = exact_fn(*inputs) (numpy, nr_oracle.c); gradient = autograd of torch_fn(*inputs)."""

    @staticmethod
    def forward(ctx, torch_fn, exact_fn, *inputs):
        ctx.torch_fn = torch_fn
        ctx.save_for_backward(*inputs)
        return exact_fn(*[t.detach() for t in inputs])

    @staticmethod
    def backward(ctx, g):
        inputs = [t.detach().requires_grad_(t.is_floating_point()) for t in ctx.saved_tensors]
        with torch.enable_grad():
            out = ctx.torch_fn(*inputs)
        need = [i for i, t in enumerate(inputs) if t.requires_grad and ctx.needs_input_grad[2 + i]]
        grads = torch.autograd.grad(out, [inputs[i] for i in need], g, allow_unused=True) if need else ()
        res = [None] * len(inputs)
        for i, gr in zip(need, grads):
            res[i] = gr
        return (None, None) + tuple(res)


def _c_view(vertices, eye, target_or_dir, up, is_look_at, width):
    """orc_camera_basis + orc_camera_points (mode look_at / look, optional perspective division)."""
    L = kernels("port").lib
    v = _f32(vertices.numpy())
    e, a, u = (_f32(t.numpy().reshape(-1, 3)) for t in (eye, target_or_dir, up))
    B = v.shape[0]
    rot = np.zeros((B, 3, 3), np.float32)
    L.orc_camera_basis(_fp(e), e.shape[0], _fp(a), a.shape[0], _fp(u), u.shape[0], int(is_look_at), _fp(rot), B)
    out = np.zeros((B, v.shape[1], 3), np.float32)
    dummy = np.zeros(9, np.float32)
    L.orc_camera_points(_fp(v), B, 1, int(width is not None), float(width or 1.0), 1.0, _fp(rot), B, _fp(e), e.shape[0],
                        _fp(dummy), 1, _fp(dummy), 1, _fp(out), B, v.shape[1])
    return torch.from_numpy(out)


def _look_at_torch(vertices, eye, at, up):
    r = _camera_basis(at - eye, up)
    return torch.matmul(vertices - eye[:, None, :], r.transpose(1, 2))


def _look_torch(vertices, eye, direction, up):
    r = _camera_basis(direction, up)
    return torch.matmul(vertices - eye[:, None, :], r.transpose(1, 2))


def look_at(vertices, eye, at=(0, 0, 0), up=(0, 1, 0), _perspective_angle=None):
    """NR/look_at.py:6-62 (+ NR/perspective.py with `_perspective_angle`, so that both happen in one exact pass)."""
    B = vertices.shape[0]
    eye, at, up = _as_vec(eye, vertices, B), _as_vec(at, vertices, B), _as_vec(up, vertices, B)
    fn = _look_at_torch if _perspective_angle is None else \
        (lambda v, e, a, u: perspective(_look_at_torch(v, e, a, u), angle=_perspective_angle))
    if not EXACT:
        return fn(vertices, eye, at, up)
    width = None if _perspective_angle is None else _tan_width(_perspective_angle)
    return _ExactForward.apply(fn, lambda v, e, a, u: _c_view(v, e, a, u, True, width), vertices, eye, at, up)


def look(vertices, eye, direction=(0, 1, 0), up=(0, 1, 0), _perspective_angle=None):
    """NR/look.py:6-53 (eye/direction/up of shape [3] broadcast as batch 1)."""
    eye, direction, up = _as_vec(eye, vertices, 1), _as_vec(direction, vertices, 1), _as_vec(up, vertices, 1)
    fn = _look_torch if _perspective_angle is None else \
        (lambda v, e, d, u: perspective(_look_torch(v, e, d, u), angle=_perspective_angle))
    if not EXACT:
        return fn(vertices, eye, direction, up)
    width = None if _perspective_angle is None else _tan_width(_perspective_angle)
    return _ExactForward.apply(fn, lambda v, e, d, u: _c_view(v, e, d, u, False, width), vertices, eye, direction, up)


def _tan_width(angle):
    return float(torch.tan(torch.tensor(angle / 180 * math.pi, dtype=torch.float32)))      # NR/perspective.py:15-17


def perspective(vertices, angle=30.0):
    """NR/perspective.py:6-21."""
    width = torch.tan(torch.tensor(angle / 180 * math.pi, dtype=torch.float32))
    z = vertices[:, :, 2]
    return torch.stack((vertices[:, :, 0] / z / width, vertices[:, :, 1] / z / width, z), dim=2)


def projection(vertices, K, R, t, dist_coeffs, orig_size, eps=1e-9):
    """NR/projection.py:6-43."""
    if not EXACT:
        return _projection_torch(vertices, K, R, t, dist_coeffs, orig_size, eps)

    def exact(v, K_, R_, t_, d_):
        L = kernels("port").lib
        vn, Kn, Rn, tn, dn = _f32(v.numpy()), _f32(K_.numpy()), _f32(R_.numpy()), _f32(t_.numpy().reshape(-1, 3)), _f32(d_.numpy())
        B = vn.shape[0]
        out = np.zeros((B, vn.shape[1], 3), np.float32)
        L.orc_camera_points(_fp(vn), B, 3, 0, 1.0, float(orig_size), _fp(Rn), Rn.shape[0], _fp(tn), tn.shape[0], _fp(Kn),
                            Kn.shape[0], _fp(dn), dn.shape[0], _fp(out), B, vn.shape[1])
        return torch.from_numpy(out)

    return _ExactForward.apply(lambda v, K_, R_, t_, d_: _projection_torch(v, K_, R_, t_, d_, orig_size, eps), exact,
                               vertices, K, R, t, dist_coeffs)


def _projection_torch(vertices, K, R, t, dist_coeffs, orig_size, eps=1e-9):
    v = torch.matmul(vertices, R.transpose(2, 1)) + t
    x, y, z = v[:, :, 0], v[:, :, 1], v[:, :, 2]
    x_ = x / (z + eps)
    y_ = y / (z + eps)
    k1, k2, p1, p2, k3 = (dist_coeffs[:, None, i] for i in range(5))
    r = torch.sqrt(x_ ** 2 + y_ ** 2)
    radial = 1 + k1 * (r ** 2) + k2 * (r ** 4) + k3 * (r ** 6)
    x__ = x_ * radial + 2 * p1 * x_ * y_ + p2 * (r ** 2 + 2 * x_ ** 2)
    y__ = y_ * radial + p1 * (r ** 2 + 2 * y_ ** 2) + 2 * p2 * x_ * y_
    v = torch.stack([x__, y__, torch.ones_like(z)], dim=-1)
    v = torch.matmul(v, K.transpose(1, 2))
    u, vv = v[:, :, 0], v[:, :, 1]
    vv = orig_size - vv
    u = 2 * (u - orig_size / 2.) / orig_size
    vv = 2 * (vv - orig_size / 2.) / orig_size
    return torch.stack([u, vv, z], dim=-1)


def get_points_from_angles(distance, elevation, azimuth, degrees=True):
    """NR/get_points_from_angles.py:6-24 (scalar form)."""
    if degrees:
        elevation, azimuth = math.radians(elevation), math.radians(azimuth)
    return (distance * math.cos(elevation) * math.sin(azimuth), distance * math.sin(elevation),
            -distance * math.cos(elevation) * math.cos(azimuth))


def vertices_to_faces(vertices, faces):
    """NR/vertices_to_faces.py:4-22: gather [B,V,3] by [B,F,3] -> [B,F,3,3]."""
    B, V = vertices.shape[:2]
    idx = faces.long() + (torch.arange(B, dtype=torch.int64) * V)[:, None, None]
    return vertices.reshape(B * V, 3)[idx]


def lighting(faces, textures, intensity_ambient=0.5, intensity_directional=0.5, color_ambient=(1, 1, 1),
             color_directional=(1, 1, 1), direction=(0, 1, 0)):
    """NR/lighting.py:5-57 (returns the lit textures; the reference multiplies in place)."""
    B, Fn = faces.shape[:2]
    ca = _as_vec(color_ambient, faces, 1)
    cd = _as_vec(color_directional, faces, 1)
    dr = _as_vec(direction, faces, 1)
    light = torch.zeros(B, Fn, 3, dtype=torch.float32)
    if intensity_ambient != 0:
        light = light + intensity_ambient * ca[:, None, :]
    if intensity_directional != 0:
        f = faces.reshape(B * Fn, 3, 3)
        normals = F.normalize(torch.cross(f[:, 0] - f[:, 1], f[:, 2] - f[:, 1], dim=-1), eps=1e-5).reshape(B, Fn, 3)
        cos = F.relu(torch.sum(normals * dr[:, None, :], dim=2))
        light = light + intensity_directional * (cd[:, None, :] * cos[:, :, None])
    return textures * light[:, :, None, None, None, :]


class Renderer:
    """Restates NR/renderer.py:11-246 on torch-CPU tensors (same attribute names, read at call time)."""

    def __init__(self, image_size=256, anti_aliasing=True, background_color=(0, 0, 0), fill_back=True,
                 camera_mode="projection", K=None, R=None, t=None, dist_coeffs=None, orig_size=1024,
                 perspective=True, viewing_angle=30, camera_direction=(0, 0, 1), near=0.1, far=100,
                 light_intensity_ambient=0.5, light_intensity_directional=0.5, light_color_ambient=(1, 1, 1),
                 light_color_directional=(1, 1, 1), light_direction=(0, 1, 0)):
        self.image_size, self.anti_aliasing = image_size, anti_aliasing
        self.background_color, self.fill_back = background_color, fill_back
        self.camera_mode = camera_mode
        self.K, self.R, self.t = K, R, t
        self.dist_coeffs = dist_coeffs if dist_coeffs is not None else torch.zeros(1, 5)
        self.orig_size = orig_size
        self.perspective, self.viewing_angle = perspective, viewing_angle
        self.eye = [0, 0, -(1. / math.tan(math.radians(viewing_angle)) + 1)]          # NR/renderer.py:46
        self.camera_direction = [0, 0, 1]
        self.near, self.far = near, far
        self.light_intensity_ambient = light_intensity_ambient
        self.light_intensity_directional = light_intensity_directional
        self.light_color_ambient, self.light_color_directional = light_color_ambient, light_color_directional
        self.light_direction = light_direction
        self.rasterizer_eps = 1e-3                                                    # NR/renderer.py:63

    def _fill_back(self, faces, textures=None):
        if self.fill_back:
            faces = torch.cat((faces, faces.flip(-1)), dim=1)                          # NR/renderer.py:86
            if textures is not None:
                textures = torch.cat((textures, textures.permute(0, 1, 4, 3, 2, 5)), dim=1)   # :156
        return faces, textures

    def _camera(self, vertices, K=None, R=None, t=None, dist_coeffs=None, orig_size=None):
        if self.camera_mode == "look_at":         # NR/renderer.py:88-99 (look_at, then perspective)
            vertices = look_at(vertices, self.eye, _perspective_angle=self.viewing_angle if self.perspective else None)
        elif self.camera_mode == "look":
            vertices = look(vertices, self.eye, self.camera_direction,
                            _perspective_angle=self.viewing_angle if self.perspective else None)
        elif self.camera_mode == "projection":
            vertices = projection(vertices, self.K if K is None else K, self.R if R is None else R,
                                  self.t if t is None else t,
                                  self.dist_coeffs if dist_coeffs is None else dist_coeffs,
                                  self.orig_size if orig_size is None else orig_size)
        return vertices          # any other mode: no transform (tests set camera_mode='none')

    def render_silhouettes(self, vertices, faces, **cam):
        faces, _ = self._fill_back(faces)
        f = vertices_to_faces(self._camera(vertices, **cam), faces)
        return rasterize_rgbad(f, None, self.image_size, self.anti_aliasing, 0.1, 100, 1e-4, None,
                               False, True, False)["alpha"]                            # NR/renderer.py:114

    def render_depth(self, vertices, faces, **cam):
        faces, _ = self._fill_back(faces)
        f = vertices_to_faces(self._camera(vertices, **cam), faces)
        return rasterize_rgbad(f, None, self.image_size, self.anti_aliasing, 0.1, 100, 1e-4, None,
                               False, False, True)["depth"]                            # NR/renderer.py:149

    def _lit(self, vertices, faces, textures):
        faces, textures = self._fill_back(faces, textures)
        textures = lighting(vertices_to_faces(vertices, faces), textures, self.light_intensity_ambient,
                            self.light_intensity_directional, self.light_color_ambient,
                            self.light_color_directional, self.light_direction)
        return faces, textures

    def render_rgb(self, vertices, faces, textures, **cam):
        faces, textures = self._lit(vertices, faces, textures)
        f = vertices_to_faces(self._camera(vertices, **cam), faces)
        return rasterize_rgbad(f, textures, self.image_size, self.anti_aliasing, self.near, self.far,
                               self.rasterizer_eps, self.background_color, True, False, False)["rgb"]

    def render(self, vertices, faces, textures, **cam):
        faces, textures = self._lit(vertices, faces, textures)
        f = vertices_to_faces(self._camera(vertices, **cam), faces)
        out = rasterize_rgbad(f, textures, self.image_size, self.anti_aliasing, self.near, self.far,
                              self.rasterizer_eps, self.background_color)
        return out["rgb"], out["depth"], out["alpha"]                                  # NR/renderer.py:246

    def __call__(self, vertices, faces, textures=None, mode=None, **cam):
        if mode is None:
            return self.render(vertices, faces, textures, **cam)
        if mode == "rgb":
            return self.render_rgb(vertices, faces, textures, **cam)
        if mode == "silhouettes":
            return self.render_silhouettes(vertices, faces, **cam)
        if mode == "depth":
            return self.render_depth(vertices, faces, **cam)
        raise ValueError("mode should be one of None, 'silhouettes' or 'depth'")


# ---- NrRenderer helpers and losses (CR/utils.py, CR/renderer_nr.py, deep3dmap/core/utils/utils.py) ----
def get_grid(b, H, W, normalize=True):
    """CR/utils.py:22-31 -> [b,H,W,2] in (x, y) order."""
    hr = torch.linspace(-1, 1, H) if normalize else torch.arange(0, H)
    wr = torch.linspace(-1, 1, W) if normalize else torch.arange(0, W)
    gy, gx = torch.meshgrid(hr, wr, indexing="ij")
    return torch.stack((gx, gy), -1)[None].repeat(b, 1, 1, 1).float()


def get_rotation_matrix(tx, ty, tz):
    """CR/utils.py:34-51: Rz @ Ry @ Rx."""
    n = len(tx)
    mx, my, mz = torch.zeros(n, 3, 3), torch.zeros(n, 3, 3), torch.zeros(n, 3, 3)
    mx[:, 0, 0] = 1
    mx[:, 1, 1], mx[:, 1, 2], mx[:, 2, 1], mx[:, 2, 2] = tx.cos(), -tx.sin(), tx.sin(), tx.cos()
    my[:, 1, 1] = 1
    my[:, 0, 0], my[:, 0, 2], my[:, 2, 0], my[:, 2, 2] = ty.cos(), ty.sin(), -ty.sin(), ty.cos()
    mz[:, 2, 2] = 1
    mz[:, 0, 0], mz[:, 0, 1], mz[:, 1, 0], mz[:, 1, 1] = tz.cos(), -tz.sin(), tz.sin(), tz.cos()
    return torch.matmul(mz, torch.matmul(my, mx))


def get_transform_matrices(view):
    """CR/utils.py:54-71."""
    b = view.size(0)
    rot = get_rotation_matrix(view[:, 0], view[:, 1], view[:, 2])
    if view.size(1) == 6:
        trans = view[:, 3:].reshape(b, 1, 3)
    elif view.size(1) == 5:
        trans = torch.cat([view[:, 3:].reshape(b, 1, 2), torch.zeros(b, 1, 1)], 2)
    else:
        trans = torch.zeros(b, 1, 3)
    return rot, trans


def get_face_idx(b, h, w):
    """CR/utils.py:74-78: two triangles per grid cell, (tl,bl,tr) block then (tr,bl,br) block."""
    idx = torch.arange(h * w).reshape(h, w)
    f1 = torch.stack([idx[:h - 1, :w - 1], idx[1:, :w - 1], idx[:h - 1, 1:]], -1).reshape(-1, 3)
    f2 = torch.stack([idx[:h - 1, 1:], idx[1:, :w - 1], idx[1:, 1:]], -1).reshape(-1, 3)
    return torch.cat([f1, f2], 0).repeat(b, 1, 1).int()


_CUBE_COEFFS = [[0.5, 0.5, 0.5], [0., 0., 1.], [0., 1., 0.], [-0.5, 0.5, 0.5],
                [1., 0., 0.], [0.5, -0.5, 0.5], [0.5, 0.5, -0.5], [0., 0., 0.]]


def vcolor_to_texture_cube(vcolors):
    """CR/utils.py:81-94: [b,c,n,3] vertex colours -> [b,n,2,2,2,c]."""
    b, c, n, f = vcolors.shape
    return torch.tensor(_CUBE_COEFFS).matmul(vcolors.permute(0, 2, 3, 1)).reshape(b, n, 2, 2, 2, c)


def get_textures_from_im(im, tx_size=1):
    """CR/utils.py:97-107."""
    b, c, h, w = im.shape
    if tx_size == 1:
        t = torch.cat([im[:, :, :h - 1, :w - 1].reshape(b, c, -1), im[:, :, 1:, 1:].reshape(b, c, -1)], 2)
        return t.transpose(2, 1).reshape(b, -1, 1, 1, 1, c)
    if tx_size == 2:
        t1 = torch.stack([im[:, :, :h - 1, :w - 1], im[:, :, :h - 1, 1:], im[:, :, 1:, :w - 1]], -1).reshape(b, c, -1, 3)
        t2 = torch.stack([im[:, :, 1:, :w - 1], im[:, :, :h - 1, 1:], im[:, :, 1:, 1:]], -1).reshape(b, c, -1, 3)
        return vcolor_to_texture_cube(torch.cat([t1, t2], 2))
    raise NotImplementedError("Currently support texture size of 1 or 2 only.")


def photometric_loss(im1, im2, mask=None, conf_sigma=None):
    """deep3dmap/core/utils/utils.py:105-114 (EPS = 1e-7)."""
    loss = (im1 - im2).abs()
    if conf_sigma is not None:
        loss = loss * 2 ** 0.5 / (conf_sigma + 1e-7) + (conf_sigma + 1e-7).log()
    if mask is not None:
        mask = mask.expand_as(loss)
        return (loss * mask).sum() / mask.sum()
    return loss.mean()


def smooth_loss(pred_map):
    """deep3dmap/core/utils/utils.py:82-102."""
    def gradient(p):
        if p.dim() == 4:
            p = p.reshape(-1, p.size(2), p.size(3))
        return p[:, :, 1:] - p[:, :, :-1], p[:, 1:] - p[:, :-1]

    maps = pred_map if isinstance(pred_map, (tuple, list)) else [pred_map]
    loss, weight = 0, 1.
    for sm in maps:
        dx, dy = gradient(sm)
        dx2, dxdy = gradient(dx)
        dydx, dy2 = gradient(dy)
        loss = loss + (dx2.abs().mean() + dxdy.abs().mean() + dydx.abs().mean() + dy2.abs().mean()) * weight
        weight /= 2.3
    return loss


class NrRenderer:
    """Restates CR/renderer_nr.py:12-277 on torch-CPU tensors, method for method (the checker of
    deep3dmap_amd/core/renderer_nr.py, which is organised differently)."""

    def __init__(self, cfgs, image_size):
        self.image_size = image_size
        self.min_depth = cfgs.get("min_depth", 0.9)
        self.max_depth = cfgs.get("max_depth", 1.1)
        self.rot_center_depth = cfgs.get("rot_center_depth", (self.min_depth + self.max_depth) / 2)
        self.fov = cfgs.get("fov", 10)
        self.tex_cube_size = cfgs.get("tex_cube_size", 2)
        self.renderer_min_depth = cfgs.get("renderer_min_depth", 0.1)
        self.renderer_max_depth = cfgs.get("renderer_max_depth", 10.)
        f = (image_size - 1) / 2 / (math.tan(self.fov / 2 * math.pi / 180))
        c = (image_size - 1) / 2
        K = torch.tensor([[f, 0., c], [0., f, c], [0., 0., 1.]], dtype=torch.float32)
        self.inv_K_origin = torch.inverse(K).unsqueeze(0)
        self.K_origin = K.unsqueeze(0)
        self.inv_K = self.inv_K_origin.clone()
        self.K = self.K_origin.clone()
        self.renderer = Renderer(camera_mode="projection", light_intensity_ambient=1.0,
                                 light_intensity_directional=0., K=self.K, R=torch.eye(3)[None],
                                 t=torch.zeros(1, 3), near=self.renderer_min_depth, far=self.renderer_max_depth,
                                 image_size=image_size, orig_size=image_size, fill_back=True,
                                 background_color=[1, 1, 1])

    def downscale_K(self, downscale):                                                  # CR:48-51
        if downscale > 1:
            self.K = torch.cat((self.K_origin[:, 0:2] / downscale, self.K_origin[:, 2:]), dim=1)
            self.inv_K = torch.inverse(self.K[0]).unsqueeze(0)

    def set_transform_matrices(self, view):                                            # CR:61-62
        self.rot_mat, self.trans_xyz = get_transform_matrices(view)

    def rotate_pts(self, pts, rot_mat):                                                # CR:64-69
        c = torch.tensor([0., 0., self.rot_center_depth]).view(1, 1, 3)
        return (pts - c).matmul(rot_mat.transpose(2, 1)) + c

    def translate_pts(self, pts, trans_xyz):                                           # CR:71-72
        return pts + trans_xyz

    def depth_to_3d_grid(self, depth):                                                 # CR:74-80
        b, h, w = depth.shape
        g = torch.cat((get_grid(b, h, w, normalize=False), torch.ones(b, h, w, 1)), dim=3)
        return g.matmul(self.inv_K.transpose(2, 1)) * depth.unsqueeze(-1)

    def grid_3d_to_2d(self, grid_3d):                                                  # CR:82-88
        b, h, w, _ = grid_3d.shape
        grid_2d = grid_3d / grid_3d[..., 2:]
        grid_2d = grid_2d.matmul(self.K.transpose(2, 1))[:, :, :, :2]
        WH = torch.tensor([w - 1, h - 1], dtype=torch.float32).view(1, 1, 1, 2)
        return grid_2d / WH * 2. - 1.

    def get_warped_3d_grid(self, depth):                                               # CR:90-100
        b, h, w = depth.shape

        def formula(d, rot, trans):
            return (self.rotate_pts(self.depth_to_3d_grid(d).reshape(b, -1, 3), rot) + trans).reshape(b, h, w, 3)

        if not EXACT:
            return formula(depth, self.rot_mat, self.trans_xyz)

        def exact(d, rot, trans):
            L = kernels("port").lib
            dn, iK = _f32(d.numpy()), _f32(self.inv_K.numpy())
            rn = _f32(rot.expand(b, 3, 3).numpy())
            tn = _f32(trans.reshape(-1, 3).expand(b, 3).numpy())
            out = np.zeros((b, h * w, 3), np.float32)
            L.orc_depth_to_vertices(_fp(dn), _fp(iK), iK.shape[0], _fp(rn), _fp(tn), float(self.rot_center_depth), _fp(out),
                                    b, h, w)
            return torch.from_numpy(out).reshape(b, h, w, 3)

        return _ExactForward.apply(formula, exact, depth, self.rot_mat, self.trans_xyz)

    def get_inv_warped_3d_grid(self, depth):                                           # CR:102-107
        b, h, w = depth.shape
        g = self.translate_pts(self.depth_to_3d_grid(depth).reshape(b, -1, 3), -self.trans_xyz)
        return self.rotate_pts(g, self.rot_mat.transpose(2, 1)).reshape(b, h, w, 3)

    def get_warped_2d_grid(self, depth):                                               # CR:109-111
        return self.grid_3d_to_2d(self.get_warped_3d_grid(depth))

    def get_inv_warped_2d_grid(self, depth):                                           # CR:112-114
        return self.grid_3d_to_2d(self.get_inv_warped_3d_grid(depth))

    def warp_canon_depth(self, canon_depth):                                           # CR:116-125
        b, h, w = canon_depth.shape
        grid_3d = self.get_warped_3d_grid(canon_depth).reshape(b, -1, 3)
        warped = self.renderer.render_depth(grid_3d, get_face_idx(b, h, w))
        margin = (self.max_depth - self.min_depth) / 2
        return warped.clamp(min=self.min_depth - margin, max=self.max_depth + margin)

    def get_normal_from_depth(self, depth):                                            # CR:127-139
        b, h, w = depth.shape
        grid_3d = self.depth_to_3d_grid(depth)
        tu = grid_3d[:, 1:-1, 2:] - grid_3d[:, 1:-1, :-2]
        tv = grid_3d[:, 2:, 1:-1] - grid_3d[:, :-2, 1:-1]
        normal = torch.linalg.cross(tu, tv, dim=3)
        zero = torch.tensor([0., 0., 1.])
        normal = torch.cat([zero.repeat(b, h - 2, 1, 1), normal, zero.repeat(b, h - 2, 1, 1)], 2)
        normal = torch.cat([zero.repeat(b, 1, w, 1), normal, zero.repeat(b, 1, w, 1)], 1)
        return normal / (((normal ** 2).sum(3, keepdim=True)) ** 0.5 + 1e-7)

    def _mesh_frame(self, im, grid_3d, b, h, w):                                        # CR:196-198
        textures = get_textures_from_im(im, tx_size=self.tex_cube_size)
        return self.renderer.render_rgb(grid_3d, get_face_idx(b, h, w), textures).clamp(min=-1., max=1.)

    def _resample_frame(self, im, depth, view):                                         # CR:180-184
        self.set_transform_matrices(view)
        recon_depth = self.warp_canon_depth(depth)
        grid = self.get_inv_warped_2d_grid(recon_depth)
        return F.grid_sample(im, grid, mode="bilinear"), grid

    def render_yaw(self, im, depth, v_before=None, v_after=None, rotations=None, maxr=90, nsample=9,
                   grid_sample=False, crop_mesh=None):                                  # CR:141-200
        b, c, h, w = im.shape
        grid_3d = self.depth_to_3d_grid(depth).clone()
        if crop_mesh is not None:
            top, bottom, left, right = crop_mesh
            if top > 0:
                grid_3d[:, :top, :, 1] = grid_3d[:, top:top + 1, :, 1].repeat(1, top, 1)
                grid_3d[:, :top, :, 2] = grid_3d[:, top:top + 1, :, 2].repeat(1, top, 1)
            if bottom > 0:
                grid_3d[:, -bottom:, :, 1] = grid_3d[:, -bottom - 1:-bottom, :, 1].repeat(1, bottom, 1)
                grid_3d[:, -bottom:, :, 2] = grid_3d[:, -bottom - 1:-bottom, :, 2].repeat(1, bottom, 1)
            if left > 0:
                grid_3d[:, :, :left, 0] = grid_3d[:, :, left:left + 1, 0].repeat(1, 1, left)
                grid_3d[:, :, :left, 2] = grid_3d[:, :, left:left + 1, 2].repeat(1, 1, left)
            if right > 0:
                grid_3d[:, :, -right:, 0] = grid_3d[:, :, -right - 1:-right, 0].repeat(1, 1, right)
                grid_3d[:, :, -right:, 2] = grid_3d[:, :, -right - 1:-right, 2].repeat(1, 1, right)
        grid_3d = grid_3d.reshape(b, -1, 3)
        if v_before is not None:
            rot_mat, trans_xyz = get_transform_matrices(v_before)
            grid_3d = self.rotate_pts(self.translate_pts(grid_3d, -trans_xyz), rot_mat.transpose(2, 1))
        if rotations is None:
            rotations = torch.linspace(-math.pi / 180 * maxr, math.pi / 180 * maxr, nsample)
        out = []
        for i, ri in enumerate(rotations):
            if grid_sample:
                view = torch.tensor([0, float(ri), 0, 0, 0, 0], dtype=torch.float32).view(1, 6)
                if v_before is not None:
                    view = view - v_before
                out.append(self._resample_frame(im, depth, view)[0])
            else:
                rot_mat_i, _ = get_transform_matrices(torch.tensor([0, float(ri), 0], dtype=torch.float32).view(1, 3))
                grid_3d_i = self.rotate_pts(grid_3d, rot_mat_i.repeat(b, 1, 1))
                if v_after is not None:
                    v_after_i = v_after[i] if len(v_after.shape) == 3 else v_after
                    rot_mat, trans_xyz = get_transform_matrices(v_after_i)
                    grid_3d_i = self.translate_pts(self.rotate_pts(grid_3d_i, rot_mat), trans_xyz)
                out.append(self._mesh_frame(im, grid_3d_i, b, h, w))
        return torch.stack(out, 1)

    def render_view(self, im, depth, v_before=None, rotations=None, maxr=[20, 90], nsample=[5, 9], grid_sample=False):
        b, c, h, w = im.shape                                                           # CR:202-252
        grid_3d = self.depth_to_3d_grid(depth).reshape(b, -1, 3)
        if v_before is not None:
            rot_mat, trans_xyz = get_transform_matrices(v_before)
            grid_3d = self.rotate_pts(self.translate_pts(grid_3d, -trans_xyz), rot_mat.transpose(2, 1))
        rotations_p = torch.linspace(-math.pi / 180 * maxr[0], math.pi / 180 * maxr[0], nsample[0])
        rotations_y = torch.linspace(-math.pi / 180 * maxr[1], math.pi / 180 * maxr[1], nsample[1])
        out = []
        for axis, angles in ((1, rotations_y), (0, rotations_p)):
            for a in angles:
                r = [0., 0., 0.]
                r[axis] = float(a)
                if grid_sample:
                    view = torch.tensor(r + [0, 0, 0], dtype=torch.float32).view(1, 6)
                    if v_before is not None:
                        view = view - v_before
                    out.append(self._resample_frame(im, depth, view)[0])
                else:
                    rot_mat_i, _ = get_transform_matrices(torch.tensor(r, dtype=torch.float32).view(1, 3))
                    out.append(self._mesh_frame(im, self.rotate_pts(grid_3d, rot_mat_i.repeat(b, 1, 1)), b, h, w))
        return torch.stack(out, 1)

    def render_given_view(self, im, depth, view, mask=None, grid_sample=True):          # CR:254-277
        b, c, h, w = im.shape
        grid_3d = self.depth_to_3d_grid(depth).reshape(b, -1, 3)
        if grid_sample:
            warped, grid = self._resample_frame(im, depth, view)
            if mask is not None:
                return warped, F.grid_sample(mask, grid, mode="nearest")
            return warped
        rot_mat, trans_xyz = get_transform_matrices(view)
        grid_3d = self.translate_pts(self.rotate_pts(grid_3d, rot_mat), trans_xyz)
        warped = self._mesh_frame(im, grid_3d, b, h, w)
        if mask is not None:
            return warped, self._mesh_frame(mask, grid_3d, b, h, w)
        return warped


def gan2shape_block(renderer, depth, albedo, light_a, light_b, light_d, input_im, flip=False, extra_mask=None,
                    lam_smooth=0.01):
    """deep3dmap/models/frameworks/gan2shape.py:463-497 (forward_step1's renderer block; :672 for `extra_mask`) on the
    oracle's NrRenderer, statement for statement.  `renderer.rot_mat` / `trans_xyz` are the current view.  Returns a dict."""
    r = renderer
    b = input_im.shape[0]
    normal = r.get_normal_from_depth(depth)                                                              # :463
    diffuse_shading = (normal * light_d.view(-1, 1, 1, 3)).sum(3).clamp(min=0).unsqueeze(1)              # :464
    shading = light_a.view(-1, 1, 1, 1) + light_b.view(-1, 1, 1, 1) * diffuse_shading                    # :465
    texture = (albedo / 2 + 0.5) * shading * 2 - 1                                                       # :466
    recon_depth = r.warp_canon_depth(depth)                                                              # :468
    grid_2d_from_canon = r.get_inv_warped_2d_grid(recon_depth)                                           # :475
    margin = (r.max_depth - r.min_depth) / 2
    recon_im_mask = (recon_depth < r.max_depth + margin).float()                                         # :477
    if flip:
        recon_im_mask = recon_im_mask[:b] * recon_im_mask[b:]                                            # :479-481
        recon_im_mask = recon_im_mask.repeat(2, 1, 1)
    recon_im_mask = recon_im_mask.unsqueeze(1).detach()
    if extra_mask is not None:
        recon_im_mask = recon_im_mask * extra_mask                                                       # :672
    recon_im = F.grid_sample(texture, grid_2d_from_canon, mode="bilinear").clamp(min=-1, max=1)          # :483
    loss_l1_im = photometric_loss(recon_im[:b], input_im, mask=recon_im_mask[:b])                        # :486
    loss_l1_im_flip = photometric_loss(recon_im[b:], input_im, mask=recon_im_mask[b:]) if flip else None  # :489
    loss_smooth = smooth_loss(depth) + smooth_loss(diffuse_shading)                                      # :493-494
    loss_total = loss_l1_im + lam_smooth * loss_smooth                                                   # :495 (no perceptual term)
    if flip:
        loss_total = loss_total + loss_l1_im_flip                                                        # :496-497
    return dict(normal=normal, diffuse_shading=diffuse_shading, texture=texture, recon_depth=recon_depth,
                recon_im=recon_im, recon_im_mask=recon_im_mask, loss_l1_im=loss_l1_im, loss_l1_im_flip=loss_l1_im_flip,
                loss_smooth=loss_smooth, loss_total=loss_total)
