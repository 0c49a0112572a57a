"""Differentiable rasterization: RasterizeFunction / Rasterize / rasterize_rgbad and friends, same
signatures and semantics as pnpmodules/neural_renderer/neural_renderer/rasterize.py (NR/rasterize.py),
running on the HIP operators of libd3m_raster.so.  CUDA(=HIP)-device tensors only, like the reference."""
import ctypes
import os

import torch
import torch.nn as nn

from .. import _lib
from . import rasterize_ops as ops
from ._util import const_tensor, deterministic as _deterministic, f32c, vertex_adjacency

DEFAULT_IMAGE_SIZE = 256
DEFAULT_ANTI_ALIASING = True
DEFAULT_NEAR = 0.1
DEFAULT_FAR = 100
DEFAULT_EPS = 1e-4
DEFAULT_BACKGROUND_COLOR = (0, 0, 0)


def _background_tensor(background_color, device):
    """[1,3] or [B,3] device tensor (NR/rasterize.py:189-194 accepts a colour or one colour per batch entry)."""
    if background_color is None:
        background_color = DEFAULT_BACKGROUND_COLOR
    if torch.is_tensor(background_color):
        t = background_color.to(device=device, dtype=torch.float32)
    else:
        t = const_tensor(background_color, device)
    return (t[None, :] if t.dim() == 1 else t).contiguous()


def _raster_forward(faces, textures, image_size, near, far, eps, background, return_rgb, return_alpha,
                    return_depth, keep_reference_maps):
    """Kernels 1+2 (+3), background blend and alpha: the body of NR/rasterize.py:50-84.
    keep_reference_maps: also materialise the reference's face_inv_map / faces_inv scratch."""
    dev = faces.device
    B, F = faces.shape[:2]
    S = int(image_size)
    m = {}
    # the reference pre-fills these (-1 / 0 / far / 0, rasterize.py:50-58); d3m_forward_face_index_map writes every
    # pixel, covered or not, so they are only allocated here
    m["face_index_map"] = torch.empty((B, S, S), dtype=torch.int32, device=dev)
    m["weight_map"] = torch.empty(B, S, S, 3, dtype=torch.float32, device=dev)
    m["depth_map"] = torch.empty((B, S, S), dtype=torch.float32, device=dev)
    dummy = torch.zeros(1, dtype=torch.float32, device=dev)
    want_finv = return_depth and keep_reference_maps
    face_inv_map = torch.empty(B, S, S, 3, 3, dtype=torch.float32, device=dev) if want_finv else dummy
    faces_inv = torch.zeros_like(faces) if keep_reference_maps else dummy
    ops.forward_face_index_map(faces, m["face_index_map"], m["weight_map"], m["depth_map"], face_inv_map, faces_inv,
                               S, near, far, return_rgb, return_alpha, want_finv)
    m["face_inv_map"] = face_inv_map
    m["faces_inv"] = faces_inv
    rgb_sampled = None
    if return_rgb:
        rgb_sampled = torch.zeros(B, S, S, 3, dtype=torch.float32, device=dev)
        m["sampling_index_map"] = torch.zeros(B, S, S, 8, dtype=torch.int32, device=dev)
        m["sampling_weight_map"] = torch.zeros(B, S, S, 8, dtype=torch.float32, device=dev)
        ops.forward_texture_sampling(faces, textures, m["face_index_map"], m["weight_map"], m["depth_map"],
                                     rgb_sampled, m["sampling_index_map"], m["sampling_weight_map"], S, eps)
    return m, rgb_sampled


def _epilogue(m, rgb_sampled, background, B, S, anti_aliasing, return_rgb, return_alpha, return_depth, outputs):
    """d3m_output_epilogue: blend + alpha at internal resolution and, when `outputs`, the flipped / pooled
    CHW output images of rasterize_rgbad in the same pass."""
    dev = m["face_index_map"].device
    s = S // 2 if anti_aliasing else S
    m["rgb_map"] = torch.empty(B, S, S, 3, dtype=torch.float32, device=dev) if return_rgb else None
    m["alpha_map"] = torch.empty(B, S, S, dtype=torch.float32, device=dev) if return_alpha else None
    rgb_out = alpha_out = depth_out = None
    if outputs:
        rgb_out = torch.empty(B, 3, s, s, dtype=torch.float32, device=dev) if return_rgb else None
        alpha_out = torch.empty(B, s, s, dtype=torch.float32, device=dev) if return_alpha else None
        depth_out = torch.empty(B, s, s, dtype=torch.float32, device=dev) if return_depth else None
    rc = _lib.lib().d3m_output_epilogue(
        _lib.ptr(m["face_index_map"]), _lib.ptr(rgb_sampled), _lib.ptr(m["depth_map"] if (outputs and return_depth) else None),
        _lib.ptr(background if return_rgb else None), background.shape[0] if return_rgb else 1,
        _lib.ptr(m["rgb_map"]), _lib.ptr(m["alpha_map"]), _lib.ptr(rgb_out), _lib.ptr(alpha_out), _lib.ptr(depth_out),
        B, S, int(bool(anti_aliasing and outputs)), _lib.stream_ptr())
    _lib.check(rc, "d3m_output_epilogue")
    return rgb_out, alpha_out, depth_out


def _raster_backward(faces, textures, m, S, eps, grad_rgb_map, grad_alpha_map, grad_depth_map, return_rgb,
                     return_alpha, return_depth, need_textures_grad):
    """NR/rasterize.py:109-156: K4 (overwrite) -> K5 -> K6 (add)."""
    dev = faces.device
    dummy = torch.zeros(1, dtype=torch.float32, device=dev)
    grad_faces = torch.zeros_like(faces, dtype=torch.float32)
    grad_textures = None
    if return_rgb or return_alpha:
        ops.backward_pixel_map(faces, m["face_index_map"], m["rgb_map"] if return_rgb else dummy,
                               m["alpha_map"] if return_alpha else dummy, grad_rgb_map if return_rgb else dummy,
                               grad_alpha_map if return_alpha else dummy, grad_faces, S, eps, return_rgb, return_alpha)
    if return_rgb and need_textures_grad:
        grad_textures = torch.zeros_like(textures, dtype=torch.float32)
        ops.backward_textures(m["face_index_map"], m["sampling_weight_map"], m["sampling_index_map"], grad_rgb_map,
                              grad_textures, faces.shape[1], faces=faces)
    if return_depth:
        ops.backward_depth_map(faces, m["depth_map"], m["face_index_map"], m["face_inv_map"], m["weight_map"],
                               grad_depth_map, grad_faces, S)
    return grad_faces, grad_textures


class RasterizeFunction(torch.autograd.Function):
    '''
    Definition of differentiable rasterize operation (NR/rasterize.py:15-226): internal-resolution maps,
    row 0 = bottom of the image.  Returns (rgb_map [B,S,S,3], alpha_map [B,S,S], depth_map [B,S,S]);
    disabled outputs are empty tensors.
    '''
    @staticmethod
    def forward(ctx, faces, textures, image_size, near, far, eps, background_color,
                return_rgb=False, return_alpha=False, return_depth=False):
        faces = f32c(faces)
        if return_rgb:
            textures = f32c(textures)
        dev = faces.device
        B = faces.shape[0]
        S = int(image_size)
        background = _background_tensor(background_color, dev) if return_rgb else None
        m, rgb_sampled = _raster_forward(faces, textures if return_rgb else None, S, float(near), float(far), float(eps),
                                         background, return_rgb, return_alpha, return_depth, True)
        if return_rgb or return_alpha:
            _epilogue(m, rgb_sampled, background, B, S, False, return_rgb, return_alpha, return_depth, False)
        ctx.cfg = (S, float(eps), bool(return_rgb), bool(return_alpha), bool(return_depth))
        ctx.maps = m
        ctx.save_for_backward(faces, textures if return_rgb else None)
        ctx.tex_needs_grad = bool(return_rgb and ctx.needs_input_grad[1])
        rgb_r = m["rgb_map"] if return_rgb else torch.tensor([])
        alpha_r = m["alpha_map"].clone() if return_alpha else torch.tensor([])
        depth_r = m["depth_map"].clone() if return_depth else torch.tensor([])
        return rgb_r, alpha_r, depth_r

    @staticmethod
    def backward(ctx, grad_rgb_map, grad_alpha_map, grad_depth_map):
        faces, textures = ctx.saved_tensors
        S, eps, rr, ra, rd = ctx.cfg
        g_rgb = f32c(grad_rgb_map) if rr else None
        g_alpha = f32c(grad_alpha_map) if ra else None
        g_depth = f32c(grad_depth_map) if rd else None
        gf, gt = _raster_backward(faces, textures, ctx.maps, S, eps, g_rgb, g_alpha, g_depth, rr, ra, rd,
                                  ctx.tex_needs_grad)
        return gf, gt, None, None, None, None, None, None, None, None


class Rasterize(nn.Module):
    '''
    Wrapper around the autograd function RasterizeFunction (NR/rasterize.py:228-251).
    '''
    def __init__(self, image_size, near, far, eps, background_color,
                 return_rgb=False, return_alpha=False, return_depth=False):
        super(Rasterize, self).__init__()
        self.image_size = image_size
        self.near = near
        self.far = far
        self.eps = eps
        self.background_color = background_color
        self.return_rgb = return_rgb
        self.return_alpha = return_alpha
        self.return_depth = return_depth

    def forward(self, faces, textures):
        if not faces.is_cuda or (textures is not None and not textures.is_cuda):
            raise TypeError('Rasterize module supports only cuda Tensors')
        return RasterizeFunction.apply(faces, textures, self.image_size, self.near, self.far,
                                       self.eps, self.background_color,
                                       self.return_rgb, self.return_alpha, self.return_depth)


class _RasterizeImages(torch.autograd.Function):
    """rasterize_rgbad in one autograd node: RasterizeFunction + vertical flip + CHW + 2x2 pooling
    (NR/rasterize.py:297-326), with the epilogue and its adjoint as single kernels."""

    @staticmethod
    def forward(ctx, faces, textures, image_size, anti_aliasing, near, far, eps, background_color, return_rgb,
                return_alpha, return_depth):
        faces = f32c(faces)
        if return_rgb:
            textures = f32c(textures)
        dev = faces.device
        B = faces.shape[0]
        S = int(image_size) * 2 if anti_aliasing else int(image_size)
        background = _background_tensor(background_color, dev) if return_rgb else None
        m, rgb_sampled = _raster_forward(faces, textures if return_rgb else None, S, float(near), float(far), float(eps),
                                         background, return_rgb, return_alpha, return_depth, False)
        rgb, alpha, depth = _epilogue(m, rgb_sampled, background, B, S, anti_aliasing, return_rgb, return_alpha,
                                      return_depth, True)
        ctx.cfg = (S, float(eps), bool(anti_aliasing), bool(return_rgb), bool(return_alpha), bool(return_depth))
        ctx.maps = m
        ctx.save_for_backward(faces, textures if return_rgb else None)
        ctx.tex_needs_grad = bool(return_rgb and ctx.needs_input_grad[1])
        empty = torch.tensor([])
        return (rgb if return_rgb else empty, alpha if return_alpha else empty, depth if return_depth else empty)

    @staticmethod
    def backward(ctx, g_rgb, g_alpha, g_depth):
        faces, textures = ctx.saved_tensors
        S, eps, aa, rr, ra, rd = ctx.cfg
        dev, B = faces.device, faces.shape[0]
        g_rgb_map = torch.empty(B, S, S, 3, dtype=torch.float32, device=dev) if rr else None
        g_alpha_map = torch.empty(B, S, S, dtype=torch.float32, device=dev) if ra else None
        g_depth_map = torch.empty(B, S, S, dtype=torch.float32, device=dev) if rd else None
        rc = _lib.lib().d3m_output_epilogue_backward(
            _lib.ptr(f32c(g_rgb) if rr else None), _lib.ptr(f32c(g_alpha) if ra else None),
            _lib.ptr(f32c(g_depth) if rd else None), _lib.ptr(g_rgb_map), _lib.ptr(g_alpha_map), _lib.ptr(g_depth_map),
            B, S, int(aa), _lib.stream_ptr())
        _lib.check(rc, "d3m_output_epilogue_backward")
        gf, gt = _raster_backward(faces, textures, ctx.maps, S, eps, g_rgb_map, g_alpha_map, g_depth_map, rr, ra, rd,
                                  ctx.tex_needs_grad)
        return (gf, gt) + (None,) * 9


_vec3_cache = {}


def _vec3_host(x):
    """A light colour / direction as three host floats (the C ABI takes them by value).  A device tensor costs a
    device-to-host copy -- a synchronisation -- so its value is remembered per tensor OBJECT and version (a weak reference:
    an address alone may be a new tensor in a freed one's place): the copy happens once per in-place change, not per call."""
    import weakref
    import numpy as np
    cacheable = torch.is_tensor(x) and x.is_cuda
    if cacheable:
        hit = _vec3_cache.get(id(x))
        if hit is not None and hit[0]() is x and hit[1] == x._version:
            return np.ctypeslib.as_ctypes(hit[2])
    a = x.detach().cpu().numpy() if torch.is_tensor(x) else x
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))
    if a.size != 3:
        raise NotImplementedError("lighting: per-batch colours / directions are not supported by the HIP path")
    if cacheable:
        if len(_vec3_cache) >= 64:
            _vec3_cache.clear()
        _vec3_cache[id(x)] = (weakref.ref(x), x._version, a)
    return np.ctypeslib.as_ctypes(a)


_side_streams = ops.StreamKeyedCache(max_per_kind=64)     # ((device index, which), forking stream handle) -> torch.cuda.Stream


def _serial_branches(batch, num_tri, image_size):
    """Whether a lit render node keeps its side branches (visibility list + edge plan beside the sampling pass; the
    gathered texture / depth pass beside the line walk) on the forking stream.  Default: never -- the branches win at every
    size measured in round 6 (32 views of the headline mesh 1.555-1.568 -> 1.506-1.522 ms on one box, 64 views -2.5 %, the
    anti-aliased step -3 %, 8 views at 1024x1024 -4.5 %, config 5 -1 to -4 %: tools_dev/branches_ab.sh).
    Rounds 4 and 5 ran the big batches (d3m_forward_big_batch) on one stream: with the kernels of that time the cross-queue
    waits cost what the filled tails bought (-1.5 % to +0 %); every kernel that got shorter since left a longer tail to fill.
    D3M_SERIAL_BRANCHES=1 / 0 forces / forbids (measurements); the deterministic mode and LitFitManual run on one stream."""
    env = os.environ.get("D3M_SERIAL_BRANCHES")
    if env is not None and env != "":
        return env != "0"
    return False


def _side_stream(device, which=0, serial=False):
    """Extra streams per (device, forking stream) for the branches a lit render node runs beside its main line: the
    visibility list / the gathered texture pass of every view group, and the main line of every group but the first.
    serial: the forking stream itself (see _serial_branches)."""
    if serial:
        return torch.cuda.current_stream(device)
    index = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    key = ((index, which), torch.cuda.current_stream(device).cuda_stream)     # one set per stream that forks
    stream = _side_streams.get(key)
    if stream is None:
        stream = torch.cuda.Stream(device=device)
        _side_streams.put(key, stream)
    return stream


MAX_VIEW_GROUPS = 3       # every group brings two streams; a captured step with ten concurrent branches made hipGraphLaunch
                          # segfault on ROCm 7.2 (2 and 3 groups replay fine), and more groups were slower anyway


def _group_bounds(B, groups):
    g = max(1, min(int(groups), B, MAX_VIEW_GROUPS))
    return [(B * k // g, B * (k + 1) // g) for k in range(g)]


def _bslice(t, lo, hi):
    """views lo..hi of a tensor whose leading axis is the batch, or the tensor itself when it is shared (batch 1)"""
    if t is None:
        return None
    return t if t.shape[0] == 1 else t[lo:hi]


class LitImagesLink:
    """What a fit objective evaluated on the finished images of ONE lit render node (core.losses.multiview_fit_loss on
    the outputs of Renderer.render) shares with that node's backward: the objective's gradient as the edge gradient's
    per-pixel records (`fit_state`, the tuple _RasterizeLit keeps for its own fused objective), the scalar gradient of the
    loss, and the zero-stride zero images the loss node returns instead of gradient images."""

    def __init__(self, fit_state, shapes, zero):
        self.fit_state, self.shapes, self.zero = fit_state, shapes, zero
        self.grad_loss = None
        self.finish_pending = False     # the objective's totals are still partial sums (D3M_FIT_FINISH_DEFERRED)
        self.pending = False            # the loss node's backward ran and the render node's has not consumed it yet
        # The images themselves are NOT kept here: the render node's context holds this link, and an image holds that
        # context as its grad_fn -- a reference cycle through the autograd graph.  The loss node (which saved them) lends
        # them for the time between its backward and the render node's.
        self.images = None

    def dummies(self):
        return tuple(self.zero.expand(s) for s in self.shapes)

    def end_of_pass(self):
        """Queued by the loss node's backward on the autograd engine (queue_callback): runs when that backward pass ends.
        `pending` still set = the render node was not part of the pass, so whoever asked for the images' gradients got
        the zero-stride zeros: fail loudly, and never leave the flag (or the lent images) for a later, unrelated pass."""
        stale, self.pending, self.images = self.pending, False, None
        if stale:
            raise RuntimeError(
                "multiview_fit_loss was linked to the lit render node that produced its images, but this backward pass "
                "stopped at the images (torch.autograd.grad(loss, images) / backward(inputs=images)): their gradient only "
                "exists as the render node's walk records.  Build the loss with multiview_fit_loss(..., link=False) to "
                "differentiate with respect to the images themselves.")

    def is_dummy(self, g):
        return g is not None and g.data_ptr() == self.zero.data_ptr() and all(st == 0 for st in g.stride())

    def add_gradient_images(self, g_rgb, g_alpha, g_depth):
        """other consumers of the images sent gradients too: the objective's own gradient images after all
        (d3m_fit_loss_backward), added to what arrived"""
        rgb, depth, alpha = self.images
        rgb_t, depth_t, alpha_t, mask, scratch = self.fit_state[:5]
        B, _, H, W = rgb.shape
        if self.finish_pending:         # (a registered objective whose finish was left to the records route: do it now)
            self.finish_pending = False
            fit_c = _RasterizeLit._fit_struct(self.fit_state, 0, 0, B, None)
            # (the pass ran at the INTERNAL size: twice the images' with anti-aliasing)
            _lib.check(_lib.lib().d3m_fit_finish(ctypes.byref(fit_c), B, H * (2 if self.fit_state[8] else 1),
                                                 _lib.stream_ptr()), "d3m_fit_finish")
        own = [torch.empty_like(t) for t in (rgb, depth, alpha)]
        t = (rgb, rgb_t, depth, depth_t, alpha, alpha_t, mask)
        _lib.check(_lib.lib().d3m_fit_loss_backward(*[_lib.ptr(x) for x in t], _lib.ptr(scratch[0]), _lib.ptr(self.grad_loss),
                                                    _lib.ptr(own[0]), _lib.ptr(own[1]), _lib.ptr(own[2]), B, H, W,
                                                    _lib.stream_ptr()), "d3m_fit_loss_backward")
        add = lambda g, o: o if (g is None or self.is_dummy(g)) else g + o
        return add(g_rgb, own[0]), add(g_alpha, own[2]), add(g_depth, own[1])


def lit_images_link(rgb, depth, alpha):
    """The lit render node (its autograd context) whose three output images these are -- rgb, depth and alpha exactly as
    rasterize_lit() / Renderer.render() returned them, one pipeline, gradients wanted, no objective linked yet (with
    anti-aliasing only a node that was given the objective's targets up front: fit_hint) -- or None."""
    fn = rgb.grad_fn
    if fn is None or fn is not depth.grad_fn or fn is not alpha.grad_fn or not isinstance(fn, _RasterizeLit._backward_cls):
        return None
    if (rgb.output_nr, alpha.output_nr, depth.output_nr) != (0, 1, 2):
        return None
    cfg = getattr(fn, "cfg", None)
    if cfg is None or fn.fit is not None or getattr(fn, "linked_fit", None) is not None:
        return None
    S, _eps, aa, ra, rd, _fb, _light, _Bl, groups = cfg
    if (aa and getattr(fn, "hint_state", None) is None) or not (ra and rd) or len(groups) != 1 or \
            fn.maps.get("visibility") is None:
        return None
    return fn


def _lit_clear_ranges(L, grad_textures, textures_batch, grad_light, light_batch, B, Ft, fill_back, ts, workspace):
    """[(device pointer, bytes)]: what d3m_backward_textures_lit (with a visibility blob) zeroes in front of its kernels"""
    ptrs, sizes = (ctypes.c_void_p * 4)(), (ctypes.c_size_t * 4)()
    n = L.d3m_backward_textures_lit_clear_ranges(_lib.ptr(grad_textures), int(textures_batch), _lib.ptr(grad_light),
                                                 int(light_batch), B, Ft, int(bool(fill_back)), ts, _lib.ptr(workspace), 1,
                                                 ptrs, sizes)
    return [(ptrs[i], sizes[i]) for i in range(n)]


def _checked_sink(grad_sink, vertices, textures):
    """`grad_sink` = (grad_vertices, grad_textures | None, loss [1]): caller-owned buffers the lit node writes its results
    into in place (MultiViewFit: views of the flat all-reduce buffer).  They are RAW destinations of kernels that write
    vertices.numel() / textures.numel() floats, so they are used only when they fit this call exactly -- shape, float32,
    contiguous, same device; anything else (a per-view mesh [B,V,3], another V or F, a sink left over from another fit)
    gets fresh buffers instead of an out-of-bounds write."""
    if grad_sink is None:
        return None
    gv, gt, loss = grad_sink

    def fits(buf, like):
        return (torch.is_tensor(buf) and buf.dtype == torch.float32 and buf.is_contiguous() and buf.device == like.device
                and tuple(buf.shape) == tuple(like.shape))
    if not fits(gv, vertices) or (gt is not None and not fits(gt, textures)):
        return None
    if not (torch.is_tensor(loss) and loss.dtype == torch.float32 and loss.numel() == 1 and loss.device == vertices.device):
        return None
    return grad_sink


class _RasterizeLit(torch.autograd.Function):
    """render / render_rgb in one autograd node with fill_back and lighting applied on the fly
    (d3m_face_light + d3m_render_lit_epilogue / d3m_backward_textures_lit): the per-view
    cat(textures, permuted) * light array of NR/renderer.py:155-167 is never materialised and textures /
    mesh may be shared by all views (batch 1).  The node starts from the SCREEN-space vertices: the face gather
    (vertices_to_faces + fill_back) happens inside, and backward accumulates the face gradients straight into the
    gradient of those vertices (d3m_vertex_target) instead of filling a [B,F',3,3] array for a scatter pass.

    VIEW GROUPS.  Nothing couples the views of a batch (KCU:92,263), so the node may run them as `view_groups`
    contiguous groups, each a complete pipeline (binning -> tile pass -> epilogue; edge gradient beside the gathered
    texture / depth pass) on streams of its own, forked from and joined to the current stream inside this one node --
    inside a captured step: parallel branches of the graph.  The passes of a pipeline alternate between latency-bound
    (binning, crossing emit) and issue-bound (tile pass, line walk) kernels; two pipelines out of phase fill each other's
    gaps.  With a fused fit objective the groups are shards of it: each normalises by the mask sum of ALL views
    (mask_sum), so the group values and gradients add up exactly as ranks do (multiview.py)."""

    @staticmethod
    def forward(ctx, screen_vertices, vertices, tri, textures, light_cfg, fill_back, image_size, anti_aliasing, near,
                far, eps, background_color, return_rgb, return_alpha, return_depth, fit=None, view_groups=1,
                defer_plan_join=False, camera=None, grad_sink=None, fit_hint=None):
        L = _lib.lib()
        vertices, textures = f32c(vertices), f32c(textures)
        tri = tri.to(torch.int32).contiguous()
        dev = vertices.device
        grad_sink = _checked_sink(grad_sink, vertices, textures)
        cam = cam_keep = basis = None
        if camera is not None:
            # THE CAMERA INSIDE THE NODE (`camera` = the parameter block of cameras._camera_struct, screen_vertices None):
            # the mesh is both projected and lit, i.e. `vertices` would receive two gradients that autograd then adds
            # with a kernel of its own; here the camera's adjoint is accumulated onto the light's (d3m_camera_backward_add)
            # and the node returns ONE gradient.  `grad_sink` = (grad_vertices [1,V,3], grad_textures, loss [1]) buffers
            # of the caller (MultiViewFit's flat all-reduce buffer): the results are produced in place, no packing copy.
            # The transform itself is part of the step's first launch (d3m_lit_front, below).
            from . import cameras
            cam, cam_keep = cameras._camera_struct(camera, dev)
            basis, basis_keep = cameras.basis_struct(camera)
            cam_keep = list(cam_keep) + basis_keep
            sv = torch.empty(camera["batch"], vertices.shape[1], 3, dtype=torch.float32, device=dev)
        else:
            sv = f32c(screen_vertices)
        B = sv.shape[0]
        Ft, V, ts = tri.shape[1], vertices.shape[1], textures.shape[2]
        Fp = 2 * Ft if fill_back else Ft
        if sv.shape[1] != V or tri.shape[0] not in (1, B):
            raise ValueError("screen_vertices must be [B, V, 3] and faces [1 or B, F, 3]")
        if textures.shape[0] not in (1, B) or textures.shape[1] != Ft:
            raise ValueError("textures must be [1 or B, num_faces, ts, ts, ts, 3] for the given faces")
        S = int(image_size) * 2 if anti_aliasing else int(image_size)
        ia, idr, ca, cd, direction = light_cfg
        cca, ccd, cdir = _vec3_host(ca), _vec3_host(cd), _vec3_host(direction)
        Bl = 1 if (vertices.shape[0] == 1 and tri.shape[0] == 1) else B
        light = torch.empty(Bl, Fp, 3, dtype=torch.float32, device=dev)        # (filled by the first launch, below)
        background = _background_tensor(background_color, dev)
        need_grad = any(ctx.needs_input_grad[:4])
        # the GEOMETRY side of backward (edge gradient K4, depth gradient K6, the light's and the camera's adjoints) only
        # exists for a mesh that wants a gradient: a texture-only optimisation builds no plan and walks no line
        need_geom = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        groups = _group_bounds(B, view_groups)
        G = len(groups)
        # Everything the branches write is allocated here, on the current stream: no tensor changes its owning stream.
        # Coverage comes straight from the indexed mesh: the binning pass reads the faces through `tri` and leaves the
        # dense copy of those that can own a pixel in `faces` (no vertices_to_faces pass); every other entry stays
        # UNINITIALISED (torch.empty: possibly NaN) and is never read -- later passes only touch pixel owners.
        faces = torch.empty(B, Fp, 3, 3, dtype=torch.float32, device=dev)
        m = {"face_index_map": torch.empty((B, S, S), dtype=torch.int32, device=dev),
             "weight_map": torch.empty(B, S, S, 3, dtype=torch.float32, device=dev),
             "depth_map": torch.empty((B, S, S), dtype=torch.float32, device=dev),
             "face_inv_map": const_tensor([0.0], dev),       # the 1-element stand-in of a disabled map (no fill launch)
             "rgb_map": torch.empty(B, S, S, 3, dtype=torch.float32, device=dev),
             "alpha_map": torch.empty(B, S, S, dtype=torch.float32, device=dev) if return_alpha else None}
        # which faces own a pixel is only needed by the backward pass: one blob per group (this node's backward may run
        # late, so they are its own buffers), built beside the sampling / epilogue pass
        vis = [torch.empty(int(L.d3m_visibility_bytes(hi - lo, Fp)), dtype=torch.uint8, device=dev) for lo, hi in groups] \
            if need_grad else None
        # ... and so is the edge gradient's plan (where the visible faces' edges cross the pixel grid, by image line):
        # geometry only, so it is built on the same branch and the backward pass starts with the line walk
        plan = [torch.empty(int(L.d3m_edge_plan_bytes(hi - lo, Fp, S)), dtype=torch.uint8, device=dev) for lo, hi in groups] \
            if need_geom else None
        s_out = S // 2 if anti_aliasing else S
        rgb = alpha = depth = loss_g = None
        fit_state = None
        # A REGISTERED objective (Renderer.fit_targets -> `fit_hint`): the node returns the IMAGES, as render() does, and the
        # pass that writes them also evaluates the objective the caller is about to evaluate on them and leaves its gradient
        # as walk records (the fused objective's pass with images_out): core.losses.multiview_fit_loss, handed these very
        # images and targets, then finds value and records here instead of re-reading the images (`hint_state`).
        hinted = fit is None and fit_hint is not None and return_alpha and return_depth and need_grad and G == 1
        if fit is None:
            rgb = torch.empty(B, 3, s_out, s_out, dtype=torch.float32, device=dev)
            alpha = torch.empty(B, s_out, s_out, dtype=torch.float32, device=dev) if return_alpha else None
            depth = torch.empty(B, s_out, s_out, dtype=torch.float32, device=dev) if return_depth else None
            if hinted:
                fit = (tuple(fit_hint) + (None,))[:5] + ((rgb, depth, alpha),)
        if fit is not None:
            # the fit objective is evaluated where the images are produced: they are never written (rasterize_lit_fit)
            if not (return_alpha and return_depth):
                raise ValueError("the fused fit objective needs rgb, alpha and depth")
            rgb_t, depth_t, alpha_t, mask = (f32c(t) for t in fit[:4])
            mask_sum = f32c(fit[4]).reshape(1) if len(fit) > 4 and fit[4] is not None else None
            if len(fit) > 5 and fit[5] is not None:
                # the images as a by-product of the same pass (caller's buffers, not differentiable outputs)
                rgb, depth, alpha = fit[5]
                for t, shape in ((rgb, (B, 3, s_out, s_out)), (depth, (B, s_out, s_out)), (alpha, (B, s_out, s_out))):
                    if tuple(t.shape) != shape or t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
                        raise ValueError("images_out must be contiguous float32 (rgb [B,3,s,s], depth [B,s,s], alpha [B,s,s])")
            if tuple(rgb_t.shape) != (B, 3, s_out, s_out) or any(tuple(t.shape) != (B, s_out, s_out) for t in (depth_t, alpha_t, mask)):
                raise ValueError("fit targets must be rgb [B,3,s,s] and depth / alpha / mask [B,s,s] at the output size")
            if mask_sum is None and (G > 1 or need_grad):
                # view groups are shards of the objective and need the normaliser of the whole batch; and with it known
                # up front the pass below can leave the gradient as the edge gradient's per-pixel records
                mask_sum = mask.sum().reshape(1)
            loss_g = grad_sink[2] if (grad_sink is not None and G == 1) else torch.empty(G, dtype=torch.float32, device=dev)
            scratch = [torch.empty(int(L.d3m_render_fit_scratch_floats(hi - lo, S)), dtype=torch.float32, device=dev)
                       for lo, hi in groups]
            # with a backward pass to come, the same pass leaves the objective's gradient behind -- minus the gradient of the
            # loss, only known later -- in the form its readers want: the edge gradient's per-pixel records at the internal
            # size (what d3m_backward_pixel_map would otherwise pack from gradient maps: no pixel pass in backward at all),
            # the lines' non-zero extents, and the depth gradient as a map.  With anti-aliasing an output pixel is the mean
            # of four internal ones: each gets a quarter of its gradient (D3M_FIT_POOLED; round 4 left unscaled maps here
            # and packed them in backward).
            g_maps = None
            if need_grad:
                g_maps = (torch.empty(B, S, S, 4, dtype=torch.float32, device=dev),        # edge_grad
                          torch.empty(B, S, S, 2, dtype=torch.float32, device=dev),        # edge_dot
                          None,                                                            # (nz_lo_inv, nz_hi1): below
                          torch.empty(B, S, S, dtype=torch.float32, device=dev))           # grad_depth_map
            fit_state = (rgb_t, depth_t, alpha_t, mask, scratch, loss_g, g_maps, mask_sum, bool(anti_aliasing))
        cur = torch.cuda.current_stream()
        det = need_grad and _deterministic()
        if det:
            _RasterizeLit._deterministic_supported(G, tri, vertices, ts, idr, ctx.needs_input_grad[1])
            if need_geom:
                vertex_adjacency(tri, V)            # (built here, outside any capture of the backward pass)
        serial = G == 1 and (det or getattr(ctx, "force_serial", False) or _serial_branches(B, Ft, S))
        mains = [cur] + [_side_stream(dev, k) for k in range(1, G)]
        auxs = [_side_stream(dev, G + k, serial) for k in range(G)]
        nz_own = None
        if fit_state is not None and fit_state[6] is not None:      # (records form)
            # the lines' non-zero extents (zero before the objective's pass fills them): when the plan is built in front of
            # that pass on the same stream, inside the plan's blob, cleared by the plan's own clear; else a fill of their own
            if serial and plan is not None:
                each = ctypes.c_size_t(0)
                off = int(L.d3m_edge_plan_extents_offset(B, Fp, S, ctypes.byref(each)))
                nz = tuple(plan[0][off + j * each.value: off + j * each.value + B * 2 * S * 4].view(torch.int32).view(B, 2, S)
                           for j in range(2))
            else:
                nz_own = torch.empty(2, B, 2, S, dtype=torch.int32, device=dev)       # (zeroed by the first launch, below)
                nz = tuple(nz_own.unbind(0))
            fit_state = fit_state[:6] + (fit_state[6][:2] + (nz,) + fit_state[6][3:],) + fit_state[7:]
        # THE STEP'S FIRST LAUNCH (d3m_lit_front): the camera transform (with its look_at basis), the per-face light and
        # every clear the operators below would otherwise each launch for themselves -- the forward workspace's counters (or
        # z-buffer), the plan's, the objective's arrival tickets, the lines' extents -- and, for a caller that runs backward
        # right behind forward (defer_plan_join: MultiViewFit's step; LitFitManual), the backward pass's accumulators and
        # masks too, allocated here for that purpose (`ctx.pre`): five launches of round 4's step are block ranges of one.
        # One pipeline only: view groups keep their own clears.
        clears, pre = [], None
        flags_fwd = flags_plan = flags_fit = 0
        ws0 = ops._workspace("fwd", L.d3m_forward_workspace_bytes(groups[0][1] - groups[0][0], Fp, S), dev)
        if G == 1 and os.environ.get("D3M_NO_PRECLEAR") != "1":      # (D3M_NO_PRECLEAR=1: every operator clears for itself; debugging)
            nb = int(L.d3m_forward_clear_bytes(B, Ft, int(bool(fill_back)), S, ws0.numel()))
            if nb:
                clears.append((ws0.data_ptr(), nb))
                flags_fwd = _lib.PRECLEARED
            if plan is not None:
                clears.append((plan[0].data_ptr(), int(L.d3m_edge_plan_clear_bytes(B, Fp, S))))
                flags_plan = _lib.PRECLEARED
            if fit_state is not None:
                off = ctypes.c_size_t(0)
                nfl = int(L.d3m_render_fit_scratch_clear_range(B, S, ctypes.byref(off)))
                clears.append((fit_state[4][0].data_ptr() + 4 * off.value, 4 * nfl))
                flags_fit = _lib.PRECLEARED
        if nz_own is not None:
            clears.append(_lib.tensor_range(nz_own))
        step_mode = G == 1 and need_grad and (defer_plan_join or getattr(ctx, "force_serial", False)) and not det and \
            os.environ.get("D3M_NO_PRECLEAR") != "1"
        if step_mode:
            pre = _RasterizeLit._backward_buffers(ctx, L, vertices, textures, light, grad_sink, camera is not None,
                                                  B, V, Ft, ts, Bl, fill_back, idr, need_geom)
            if len(clears) + len(pre["clears"]) <= _lib.FRONT_RANGES:
                clears += pre["clears"]
            else:
                pre = None
        assert len(clears) <= _lib.FRONT_RANGES
        # ... and the objective's finish (partial sums -> value) is left to a kernel that backward pass launches anyway
        if pre is not None and pre["gathered"] and fit_state is not None and fit_state[6] is not None:
            flags_fit |= _lib.FIT_FINISH_DEFERRED
        ctx.fit_flags = flags_fit & _lib.FIT_FINISH_DEFERRED
        zp = (ctypes.c_void_p * max(1, len(clears)))(*[c[0] for c in clears])
        zb = (ctypes.c_size_t * max(1, len(clears)))(*[c[1] for c in clears])
        _lib.check(L.d3m_lit_front(_lib.ptr(vertices), vertices.shape[0], ctypes.byref(cam) if cam is not None else None,
                                   ctypes.byref(basis) if basis is not None else None,
                                   _lib.ptr(sv) if cam is not None else None, B, V, _lib.ptr(tri), tri.shape[0], Ft,
                                   int(bool(fill_back)), _lib.ptr(light), Bl, float(ia), float(idr), cca, ccd, cdir,
                                   zp, zb, len(clears), _lib.stream_ptr()), "d3m_lit_front")
        ctx.pre = pre
        # The visibility list and the plan are only read by backward.  A caller that runs backward right behind forward,
        # on the same stream and (if captured) in the same capture, may leave that branch open at the end of forward
        # (defer_plan_join): backward waits for the plan where it first needs it and joins the branch, which runs on
        # under the loss and the first backward passes.  The outputs are then only valid after backward.
        plan_ready = [] if (plan is not None and defer_plan_join and G == 1) else None   # (G > 1: capture crashes, as above)
        vis_ready, vis_on_main = None, False
        for k in range(G):
            if mains[k] is not cur:
                mains[k].wait_stream(cur)
            # every branch enters (a capture) from the ORIGIN stream: a stream that joins through another branch's
            # event -- a fork of a fork -- crashes hipStreamEndCapture on ROCm 7.2
            if vis is not None and auxs[k] is not cur and auxs[k] is not mains[k]:
                auxs[k].wait_stream(cur)
        for k, (lo, hi) in enumerate(groups):
            Bg = hi - lo
            with torch.cuda.stream(mains[k]):
                tri_g, tex_g, light_g, bg_g = (_bslice(t, lo, hi) for t in (tri, textures, light, background))
                fi_g, wm_g, dm_g = m["face_index_map"][lo:hi], m["weight_map"][lo:hi], m["depth_map"][lo:hi]
                ws = ws0 if k == 0 else ops._workspace("fwd", L.d3m_forward_workspace_bytes(Bg, Fp, S), dev)
                _lib.check(L.d3m_forward_face_index_map_mesh(
                    _lib.ptr(sv[lo:hi]), _lib.ptr(tri_g), tri_g.shape[0], V, Ft, int(bool(fill_back)), _lib.ptr(faces[lo:hi]),
                    _lib.ptr(fi_g), _lib.ptr(wm_g), _lib.ptr(dm_g), None, Bg, S, float(near), float(far), _lib.ptr(ws),
                    ws.numel(), _lib.ptr(vis[k]) if vis is not None else None, vis[k].numel() if vis is not None else 0,
                    flags_fwd, _lib.stream_ptr()), "d3m_forward_face_index_map_mesh")
                if vis is not None and plan is None:
                    # (texture-only: the list is all backward needs -- on this stream, no branch)
                    _lib.check(L.d3m_visibility(None, _lib.ptr(vis[k]), vis[k].numel(), Bg, Fp, S, _lib.stream_ptr()),
                               "d3m_visibility")
                    vis_on_main = True
                elif vis is not None:
                    # ONE FORK, BEHIND THE VISIBILITY LIST (round 5).  The list has two readers -- the plan (side branch) and
                    # backward's gathered pass (this stream) -- and the coverage pass two as well (the list, the sampling
                    # pass).  Forked behind coverage, a replayed graph's critical chain (coverage -> list -> plan -> line
                    # walk) carried two kernels with a second successor each, and each cost it ~5 us (trace of the 4-view
                    # shard: 5.8 us between k_bid_resolve and the list, 5.1 behind the list).  With the list on this
                    # stream, in front of the fork, only it has two successors; the sampling pass starts 5 us later on a
                    # chain that has 40 us to spare.  4 views 0.2747 -> 0.2709 ms, 8 views 0.4687 -> 0.4660 (same box).
                    vis_first = plan_ready is not None and auxs[k] is not mains[k] and \
                        os.environ.get("D3M_FORK_BEHIND_LIST", "1") != "0"
                    if vis_first:
                        _lib.check(L.d3m_visibility(None, _lib.ptr(vis[k]), vis[k].numel(), Bg, Fp, S,
                                                    _lib.stream_ptr()), "d3m_visibility")
                        vis_on_main = True
                    if auxs[k] is not mains[k]:
                        auxs[k].wait_stream(mains[k])
                    with torch.cuda.stream(auxs[k]):
                        # (its first step -- which faces own a pixel -- was left by the tile pass above)
                        if not vis_first:
                            _lib.check(L.d3m_visibility(None, _lib.ptr(vis[k]), vis[k].numel(), Bg, Fp, S,
                                                        _lib.stream_ptr()), "d3m_visibility")
                        if plan_ready is not None and not vis_first:
                            vis_ready = torch.cuda.Event()
                            vis_ready.record(auxs[k])
                        _lib.check(L.d3m_edge_plan(_lib.ptr(faces[lo:hi]), _lib.ptr(fi_g), _lib.ptr(vis[k]), _lib.ptr(plan[k]),
                                                   plan[k].numel(), Bg, Fp, S, flags_plan, _lib.stream_ptr()), "d3m_edge_plan")
                        if plan_ready is not None:
                            plan_ready.append(torch.cuda.Event())
                            plan_ready[k].record(auxs[k])
                fit_c = None
                if fit_state is not None:
                    # (the objective's value is completed by the pass's own last workgroups: no finishing launch)
                    fit_c = _RasterizeLit._fit_struct(fit_state, k, lo, hi, None, flags=flags_fit)
                # texture sampling + background blend + alpha + flip / pooling in one pass (no rgb_sampled round trip)
                _lib.check(L.d3m_render_lit_epilogue(
                    _lib.ptr(faces[lo:hi]), _lib.ptr(tex_g), tex_g.shape[0], _lib.ptr(light_g), light_g.shape[0],
                    _lib.ptr(fi_g), _lib.ptr(wm_g), _lib.ptr(dm_g), _lib.ptr(bg_g), bg_g.shape[0],
                    _lib.ptr(m["rgb_map"][lo:hi]), _lib.ptr(_bslice(m["alpha_map"], lo, hi)),
                    _lib.ptr(_bslice(rgb, lo, hi)), _lib.ptr(_bslice(alpha, lo, hi)), _lib.ptr(_bslice(depth, lo, hi)), Bg, Ft,
                    int(bool(fill_back)), S, ts, float(eps), int(bool(anti_aliasing)),
                    ctypes.byref(fit_c) if fit_c is not None else None, _lib.stream_ptr()), "d3m_render_lit_epilogue")
        for k in range(G):
            if mains[k] is not cur:
                cur.wait_stream(mains[k])
            if vis is not None and plan_ready is None:
                cur.wait_stream(auxs[k])
        m["visibility"] = vis
        m["edge_plan"] = plan
        m["plan_ready"] = plan_ready
        m["vis_ready"] = vis_ready if plan_ready is not None else None
        m["vis_on_main"] = vis_on_main       # (the list was built on the forking stream: its readers there need no event)
        m["plan_stream"] = auxs[0]
        ctx.cfg = (S, float(eps), bool(anti_aliasing), bool(return_alpha), bool(return_depth), bool(fill_back),
                   (float(ia), float(idr), ca, cd, direction), Bl, groups)
        ctx.maps = m
        ctx.fit = None if hinted else fit_state
        ctx.hint_state = fit_state if hinted else None
        # (whether the registered objective brought its normaliser or the node took sum(mask): multiview_fit_loss only rides
        #  on the node's result when it is asked for the same one)
        ctx.hint_mask_sum_given = bool(hinted and len(fit_hint) > 4 and fit_hint[4] is not None)
        ctx.camera, ctx.cam_keep, ctx.grad_sink = camera, cam_keep, grad_sink
        ctx.save_for_backward(faces, vertices, tri, textures, light)
        if fit is not None and not hinted:
            return loss_g.sum() if G > 1 else loss_g.reshape(())
        empty = torch.tensor([])
        return (rgb, alpha if return_alpha else empty, depth if return_depth else empty)

    @staticmethod
    def _deterministic_supported(G, tri, vertices, ts, idr, vertices_need_grad):
        """The deterministic backward pass covers one pipeline over ONE shared index tensor (and, for the light's adjoint,
        one shared mesh) with gathered texture passes (texture_size 2, 3 or 4): anything else would silently fall back to
        float atomics in arrival order, so it raises instead."""
        if G != 1 or tri.shape[0] != 1 or ts not in (2, 3, 4) or (vertices_need_grad and idr != 0 and vertices.shape[0] != 1):
            raise NotImplementedError(
                "D3M_DETERMINISTIC / d3m_set_deterministic(1): the lit render node's reproducible backward pass needs "
                "view_groups == 1, faces of batch 1 (one shared topology), texture_size 2..4 and, with directional light, "
                "vertices of batch 1")

    @staticmethod
    def _backward_buffers(ctx, L, vertices, textures, light, grad_sink, camera_inside, B, V, Ft, ts, Bl, fill_back, idr,
                          need_geom=True):
        """The backward pass's accumulators and the gathered pass's workspace, allocated in FORWARD for a caller that runs
        backward right behind it: the forward's first launch zeroes what they need zeroed (`clears`), so the backward
        pass starts without a clear of its own.  Mirrors the allocations of _backward_halves (one pipeline)."""
        dev = vertices.device
        need_tex = ctx.needs_input_grad[3]
        need_vert = ctx.needs_input_grad[1] and idr != 0
        gathered = need_tex or need_vert
        pre = {"clears": [], "gathered": gathered, "grad_vertices": None, "grad_sv": None}
        if need_geom:
            pre["grad_sv"] = torch.empty(B, V, 3, dtype=torch.float32, device=dev)
            pre["clears"].append(_lib.tensor_range(pre["grad_sv"]))
        if gathered and need_vert:
            gv = grad_sink[0] if (grad_sink is not None and camera_inside) else torch.empty_like(vertices)
            pre["grad_vertices"] = gv
            pre["clears"].append(_lib.tensor_range(gv))
        if gathered:
            gt = grad_sink[1] if (textures.shape[0] == 1 and grad_sink is not None and grad_sink[1] is not None) \
                else torch.empty_like(textures)
            gl = torch.empty_like(light) if need_vert else None
            ws = torch.empty(int(L.d3m_backward_textures_lit_workspace_bytes(B, Ft, int(bool(fill_back)), ts)),
                             dtype=torch.uint8, device=dev)
            pre["clears"] += _lit_clear_ranges(L, gt, textures.shape[0], gl, Bl, B, Ft, fill_back, ts, ws)
            pre.update(gt=gt, gl=gl, lit_ws=ws)
        return pre

    @staticmethod
    def _fit_struct(fit_state, k, lo, hi, grad_loss, flags=0):
        """d3m_fit_targets of view group k (views lo..hi): forward's `fit` (grad_loss None) / backward's `unscaled`."""
        rgb_t, depth_t, alpha_t, mask, scratch, loss_g, g_maps, mask_sum, pooled = fit_state
        eg = ed = nz_lo = nz_hi = gd = None
        if g_maps is not None:                      # the gradient as per-pixel records at the internal size
            eg, ed, gd = g_maps[0][lo:hi], g_maps[1][lo:hi], g_maps[3][lo:hi]
            nz_lo, nz_hi = g_maps[2][0][lo:hi], g_maps[2][1][lo:hi]
            if pooled:
                flags = int(flags) | _lib.FIT_POOLED
        return _lib.D3MFitTargets(
            _lib.ptr(rgb_t[lo:hi]), _lib.ptr(depth_t[lo:hi]), _lib.ptr(alpha_t[lo:hi]), _lib.ptr(mask[lo:hi]),
            _lib.ptr(scratch[k]), _lib.ptr(loss_g[k:k + 1]), None, None, _lib.ptr(gd),
            _lib.ptr(grad_loss), _lib.ptr(mask_sum), _lib.ptr(eg), _lib.ptr(ed), _lib.ptr(nz_lo), _lib.ptr(nz_hi),
            int(flags))

    @staticmethod
    def backward(ctx, g_rgb, g_alpha=None, g_depth=None):
        halves = _RasterizeLit._backward_halves(ctx, g_rgb, g_alpha, g_depth)
        next(halves)                    # ... up to the point where the texture gradient is final
        try:
            next(halves)                # ... the edge gradient, the light's and the camera's adjoints
        except StopIteration as done:
            return done.value
        raise RuntimeError("_backward_halves yields once")

    @staticmethod
    def _backward_halves(ctx, g_rgb, g_alpha=None, g_depth=None):
        """The node's backward as a generator with ONE cut: it yields when everything on the TEXTURE side has been
        issued (the gathered texture / depth pass and the sum over views: grad_textures is final in stream order), and
        returns the gradients after the GEOMETRY side (edge gradient, light and camera adjoints).  Autograd runs both
        halves back to back (backward() above); LitFitManual stops in between, so that a camera-sharded fit can send the
        texture gradient on its way while the geometry side still runs (deep3dmap_amd/multiview.py).  With more than one
        view group the cut comes after the last group's passes have been issued (nothing to overlap)."""
        L = _lib.lib()
        faces, vertices, tri, textures, light = ctx.saved_tensors
        S, eps, aa, ra, rd, fill_back, (ia, idr, ca, cd, direction), Bl, groups = ctx.cfg
        m = ctx.maps
        dev, B, G = faces.device, faces.shape[0], len(groups)
        Ft, V, ts = tri.shape[1], vertices.shape[1], textures.shape[2]
        # a caller that runs backward right behind forward had its accumulators allocated AND zeroed by the forward's first
        # launch (ctx.pre: one use -- a second backward over the same graph allocates and clears its own)
        pre = getattr(ctx, "pre", None)
        ctx.pre = None
        # (no gradient for the mesh wanted -- a texture-only optimisation: no screen-space accumulator, no edge gradient, no
        #  depth gradient, no camera adjoint; the node returns None for the vertices)
        need_geom = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        grad_sv = None
        if need_geom:
            grad_sv = pre["grad_sv"] if pre is not None else torch.empty(B, V, 3, dtype=torch.float32, device=dev)
        grad_loss = scratch = mask_sum = None
        records = None
        # A fit objective evaluated on this node's finished images (core.losses.multiview_fit_loss -> LitImagesLink) has left
        # its gradient as per-pixel records and sends zero-stride ZERO images back: if those are all that arrived, backward
        # is the fused objective's (no gradient images exist); if other consumers of the images added theirs, the
        # objective's gradient images are materialised after all and added.
        fit, link = ctx.fit, getattr(ctx, "linked_fit", None)
        if fit is None and link is not None and link.pending:
            link.pending = False
            if link.is_dummy(g_rgb) and link.is_dummy(g_alpha) and link.is_dummy(g_depth):
                fit, g_rgb = link.fit_state, link.grad_loss
            else:
                g_rgb, g_alpha, g_depth = link.add_gradient_images(g_rgb, g_alpha, g_depth)
            link.images = None
        if fit is None:
            # the adjoint of the output epilogue (un-pool, un-flip, CHW -> HWC) writes the rgb / alpha gradients straight
            # as the edge gradient's per-pixel records (what d3m_backward_pixel_map would pack from gradient maps: one
            # pass over the pixels instead of two, no [B,S,S,3] / [B,S,S] gradient maps); the depth gradient as a map
            g_rgb_map = g_alpha_map = None
            g_depth_map = torch.empty(B, S, S, dtype=torch.float32, device=dev) if rd else None
            records = (torch.empty(B, S, S, 4, dtype=torch.float32, device=dev),
                       torch.empty(B, S, S, 2, dtype=torch.float32, device=dev),
                       torch.empty(2, B, 2, S, dtype=torch.int32, device=dev))
            _lib.check(L.d3m_output_epilogue_backward_records(
                _lib.ptr(f32c(g_rgb)), _lib.ptr(f32c(g_alpha) if ra else None), _lib.ptr(f32c(g_depth) if rd else None),
                _lib.ptr(m["face_index_map"]), _lib.ptr(m["rgb_map"]), _lib.ptr(m["alpha_map"] if ra else None),
                _lib.ptr(records[0]), _lib.ptr(records[1]), _lib.ptr(records[2][0]), _lib.ptr(records[2][1]),
                _lib.ptr(g_depth_map), B, S, int(aa), _lib.stream_ptr()), "d3m_output_epilogue_backward_records")
        else:       # g_rgb is the gradient of the scalar objective; the maps were left by forward, minus their scalars
            scratch, g_depth_map = fit[4], fit[6][3]
            g_rgb_map = g_alpha_map = None              # they exist as per-pixel records (fit[6][:3])
            grad_loss = f32c(g_rgb).reshape(1)
        # K4 -> textures (separate buffers) -> K6, as NR/rasterize.py:141-151; both face gradients land in grad_sv,
        # and both passes run over the compacted list of the faces that own a pixel.  The edge gradient (K4: ~8
        # latency-bound launches) and the gathered texture / depth pass (K5+K6) are independent -- they only meet in
        # the float atomics on grad_sv -- so the second runs on a stream of its own, per view group.
        vis = m["visibility"]
        need_tex = ctx.needs_input_grad[3]
        need_vert = ctx.needs_input_grad[1] and idr != 0
        gathered = need_tex or need_vert
        grad_textures = grad_vertices = grad_light = None
        tex_shared, light_shared = textures.shape[0] == 1, Bl == 1
        gt_g = gl_g = None
        sink = ctx.grad_sink
        if pre is not None and (pre["gathered"] != gathered or G != 1):
            pre = None
            if grad_sv is not None:
                _lib.zero_(grad_sv)                  # (cannot happen: the same inputs decide both; stay correct anyway)
        if gathered and need_vert:
            grad_vertices = pre["grad_vertices"] if pre is not None else \
                (sink[0] if (sink is not None and ctx.camera is not None) else torch.empty_like(vertices))
        if gathered:
            if pre is not None:
                if tex_shared:
                    gt_g = [pre["gt"]]
                else:
                    grad_textures, gt_g = pre["gt"], [pre["gt"]]
                if need_vert:
                    gl_g = [pre["gl"]]
                    if not light_shared:
                        grad_light = pre["gl"]
            else:
                if tex_shared:
                    gt_g = [sink[1] if (sink is not None and sink[1] is not None and G == 1) else torch.empty_like(textures)
                            for _ in groups]
                else:
                    grad_textures = torch.empty_like(textures)
                    gt_g = [grad_textures[lo:hi] for lo, hi in groups]
                if need_vert:
                    if light_shared:
                        gl_g = [torch.empty_like(light) for _ in groups]
                    else:
                        grad_light = torch.empty_like(light)
                        gl_g = [grad_light[lo:hi] for lo, hi in groups]
        # DETERMINISTIC (d3m_set_deterministic / D3M_DETERMINISTIC=1; one pipeline, one shared topology).  Everything the
        # pass adds up with float atomics in arrival order is produced per (view, face) by the lanes that own it and then
        # summed in a FIXED order instead:
        #   K4 -> its own dense [B,F',3,3] array (k_edge_gather's plain stores: the reference's layout), K6 -> another one
        #   (k_backward_textures_lit_faces with grad_faces and NO large-face fallback: a face's lanes own its sums whatever
        #   its size), both gathered per vertex over a CSR adjacency of the index tensor (d3m_vertex_gather);
        #   the light's gradient per VIEW ([B,F',3]: one contribution per entry), summed over the views by torch's ordered
        #   reduction, its adjoint gathered per vertex (d3m_face_light_backward_gather);
        #   the visibility list in ascending order (the library, same switch); every kernel on one stream.
        # The camera's adjoint and the sum of the per-view texel gradients already add in view order.  What stays
        # unordered: the plan's fallback for an undersized workspace (k_edge_overflow) -- see DESIGN.md section 6.
        det = _deterministic()
        if det:
            _RasterizeLit._deterministic_supported(G, tri, vertices, ts, idr, ctx.needs_input_grad[1])
            pre = None
        cur = torch.cuda.current_stream()
        serial = G == 1 and (det or getattr(ctx, "force_serial", False) or _serial_branches(B, Ft, S))
        mains = [cur] + [_side_stream(dev, k) for k in range(1, G)]
        auxs = [_side_stream(dev, G + k, serial) for k in range(G)]
        plan_ready = m["plan_ready"]
        # THE EDGE GRADIENT STAYS ON THE PLAN'S STREAM (round 4).  With the forward's branch left open (plan_ready), the
        # step's critical chain is coverage -> visibility list -> plan -> line walk -> gather: all but the first on the side
        # stream if the line walk is issued there too, right behind the plan's last kernel, instead of on the forking stream
        # behind an event -- a replayed graph pays ~9 us for every cross-queue edge whose producer has only just finished
        # (trace of the 4-view shard: scatter ends 190.0, the line walk starts 198.6).  The gathered texture / depth pass
        # takes the forking stream then; it only needs the visibility list (an event behind d3m_visibility).
        swap = plan_ready is not None and G == 1 and auxs[0] is not cur and \
            (m.get("vis_ready") is not None or m.get("vis_on_main", False))
        s_edges = [auxs[0]] if swap else mains
        s_gath = [mains[0]] if swap else auxs
        # The backward pass's clears -- the two vertex accumulators and what the gathered pass needs zeroed -- as ONE launch
        # in front of the branches (round 4: one here and one inside d3m_backward_textures_lit), or none at all (`pre`).
        lit_ws, lit_flags = None, 0
        if pre is not None:
            if gathered:
                lit_ws, lit_flags = [pre["lit_ws"]], _lib.PRECLEARED
        else:
            ranges = [_lib.tensor_range(grad_sv)] if grad_sv is not None else []
            if grad_vertices is not None:
                ranges.append(_lib.tensor_range(grad_vertices))
            det_light = det_gl = det_k4 = det_k6 = None
            if det:
                # per-VIEW light gradients (one contribution per entry), the two per-face arrays of K4 and K6
                Fp_ = 2 * Ft if fill_back else Ft
                if gathered and need_vert:
                    det_light = light.expand(B, Fp_, 3).contiguous() if light_shared else light
                    det_gl = torch.empty(B, Fp_, 3, dtype=torch.float32, device=dev)
                if need_geom:
                    det_k4 = torch.empty(B, Fp_, 3, 3, dtype=torch.float32, device=dev)
                    ranges.append(_lib.tensor_range(det_k4))
                    if rd:
                        det_k6 = torch.empty(B, Fp_, 3, 3, dtype=torch.float32, device=dev)
                        ranges.append(_lib.tensor_range(det_k6))
            if gathered and G == 1:
                with torch.cuda.stream(s_gath[0]):     # (scratch is cached per stream: the stream the pass will run on)
                    ws1 = ops._workspace("lit", L.d3m_backward_textures_lit_workspace_bytes(B, Ft, int(fill_back), ts), dev)
                ranges += _lit_clear_ranges(L, grad_textures if grad_textures is not None else gt_g[0], textures.shape[0],
                                            det_gl if det_gl is not None else (gl_g[0] if gl_g is not None else None),
                                            B if det_gl is not None else light.shape[0], B, Ft, fill_back, ts, ws1)
                lit_ws, lit_flags = [ws1], _lib.PRECLEARED
            _lib.zero_raw(ranges)
        for k in range(G):
            if mains[k] is not cur:
                mains[k].wait_stream(cur)
            if gathered or plan_ready is not None:
                auxs[k].wait_stream(cur)
            if swap:
                if m.get("vis_ready") is not None:
                    mains[k].wait_event(m["vis_ready"])
                # the line walk reads the plan.  Backward normally runs under the stream forward forked from (autograd
                # restores it), so auxs[k] IS the stream the plan was recorded on and needs no edge -- and must not get
                # one: a wait on an event of the waiting stream itself inside a capture makes hipStreamEndCapture segfault
                # (ROCm 7.2).  Driven under another current stream the side stream is another object: then it waits.
                if auxs[k].cuda_stream != m["plan_stream"].cuda_stream:
                    auxs[k].wait_event(plan_ready[k])
            elif plan_ready is not None:
                mains[k].wait_event(plan_ready[k])       # the forward's open branch: visibility + plan of this group
        for k, (lo, hi) in enumerate(groups):
            Bg = hi - lo
            tri_g, tex_g, light_g = (_bslice(t, lo, hi) for t in (tri, textures, light))
            fi_g, wm_g, dm_g = m["face_index_map"][lo:hi], m["weight_map"][lo:hi], m["depth_map"][lo:hi]
            target = _lib.D3MVertexTarget(_lib.ptr(grad_sv[lo:hi]), _lib.ptr(tri_g), V, Ft, tri_g.shape[0], int(fill_back)) \
                if need_geom else None
            rd_geom = rd and need_geom              # the depth gradient (K6) is the mesh's
            unscaled = None
            if fit is not None:
                # (the deferred finish belongs to the objective THIS node evaluated -- its own or the registered one -- not to
                #  a foreign state a linked loss built on the finished images)
                own_fit = fit is ctx.fit or fit is getattr(ctx, "hint_state", None)
                unscaled = _RasterizeLit._fit_struct(fit, k, lo, hi, grad_loss,
                                                     flags=getattr(ctx, "fit_flags", 0) if own_fit else 0)
            elif records is not None:           # final records: no scratch, no scalars to apply
                unscaled = _lib.D3MFitTargets(None, None, None, None, None, None, None, None, None, None, None,
                                              _lib.ptr(records[0][lo:hi]), _lib.ptr(records[1][lo:hi]),
                                              _lib.ptr(records[2][0][lo:hi]), _lib.ptr(records[2][1][lo:hi]), 0)
            if gathered:
                with torch.cuda.stream(s_gath[k]):
                    ws = lit_ws[k] if lit_ws is not None else \
                        ops._workspace("lit", L.d3m_backward_textures_lit_workspace_bytes(Bg, Ft, int(fill_back), ts), dev)
                    # the depth gradient (K6, add) rides along in the same pass over the faces' pixels
                    if det:     # per-view light gradient, K6 into its own per-face array: no vertex target (see DETERMINISTIC)
                        light_d = det_light if det_light is not None else light_g
                        _lib.check(L.d3m_backward_textures_lit(
                            _lib.ptr(faces), _lib.ptr(tex_g), tex_g.shape[0], _lib.ptr(light_d), light_d.shape[0],
                            _lib.ptr(fi_g), _lib.ptr(wm_g), _lib.ptr(dm_g), None, _lib.ptr(gt_g[k]), _lib.ptr(det_gl),
                            _lib.ptr(g_depth_map) if rd_geom else None, _lib.ptr(det_k6) if rd_geom else None, Bg, Ft,
                            int(fill_back), S, ts, eps, _lib.ptr(ws), ws.numel(), None, _lib.ptr(vis[k]),
                            ctypes.byref(unscaled) if unscaled is not None else None, lit_flags, _lib.stream_ptr()),
                            "d3m_backward_textures_lit")
                    else:
                        _lib.check(L.d3m_backward_textures_lit(
                            _lib.ptr(faces[lo:hi]), _lib.ptr(tex_g), tex_g.shape[0], _lib.ptr(light_g), light_g.shape[0],
                            _lib.ptr(fi_g), _lib.ptr(wm_g), _lib.ptr(dm_g), _lib.ptr(_bslice(g_rgb_map, lo, hi)),
                            _lib.ptr(gt_g[k]), _lib.ptr(gl_g[k]) if gl_g is not None else None,
                            _lib.ptr(g_depth_map[lo:hi]) if rd_geom else None, None, Bg, Ft, int(fill_back), S, ts, eps,
                            _lib.ptr(ws), ws.numel(), ctypes.byref(target) if rd_geom else None, _lib.ptr(vis[k]),
                            ctypes.byref(unscaled) if unscaled is not None else None, lit_flags, _lib.stream_ptr()),
                            "d3m_backward_textures_lit")
            if G == 1:
                yield "textures"        # (one pipeline: the texture side is complete in the order of its stream)
            if need_geom:
                with torch.cuda.stream(s_edges[k]):
                    ops.backward_pixel_map(faces[lo:hi], fi_g, m["rgb_map"][lo:hi], m["alpha_map"][lo:hi] if ra else None,
                                           _bslice(g_rgb_map, lo, hi), _bslice(g_alpha_map, lo, hi) if ra else None,
                                           det_k4 if det else None, S, eps, True, ra,
                                           vertex_target=None if det else target, visibility=vis[k], unscaled=unscaled,
                                           edge_plan=m["edge_plan"][k])
                    if det:     # K4's and K6's per-face arrays -> the screen-space gradient, per vertex in a fixed order
                        adj_off, adj_items = vertex_adjacency(tri, V)
                        _lib.check(L.d3m_vertex_gather(_lib.ptr(det_k4), _lib.ptr(det_k6), _lib.ptr(adj_off), _lib.ptr(adj_items),
                                                       _lib.ptr(grad_sv), B, V, Ft, int(fill_back), _lib.ptr(vis[k]),
                                                       _lib.stream_ptr()), "d3m_vertex_gather")
        if G > 1:
            yield "textures"
        def light_to_vertices(grad_light):
            # the light gradient -> world-space vertices through the face normals
            if det:     # the views' gradients summed in view order (torch's reduction), the adjoint gathered per vertex
                total = det_gl.sum(0) if light_shared else None
                if total is None:
                    raise NotImplementedError("deterministic mode: one shared mesh")
                adj_off, adj_items = vertex_adjacency(tri, V)
                _lib.check(L.d3m_face_light_backward_gather(
                    _lib.ptr(vertices), _lib.ptr(tri), _lib.ptr(adj_off), _lib.ptr(adj_items), _lib.ptr(total),
                    _lib.ptr(grad_vertices), ia, idr, _vec3_host(ca), _vec3_host(cd), _vec3_host(direction), V, Ft,
                    int(fill_back), _lib.stream_ptr()), "d3m_face_light_backward_gather")
                return
            _lib.check(L.d3m_face_light_backward(
                _lib.ptr(vertices), vertices.shape[0], _lib.ptr(tri), tri.shape[0], _lib.ptr(grad_light),
                _lib.ptr(grad_vertices), ia, idr, _vec3_host(ca), _vec3_host(cd), _vec3_host(direction), Bl, V,
                Ft, int(fill_back), _lib.stream_ptr()), "d3m_face_light_backward")

        # One pipeline on one stream with the camera inside the node: the light's adjoint and the camera's are the step's last
        # two kernels -- ONE launch then (d3m_lit_back: both add into the zeroed grad_vertices with float atomics).  With
        # branches the light's adjoint runs beside the line walk instead (below), where it costs the step nothing.
        fused_tail = serial and G == 1 and gathered and need_vert and ctx.camera is not None and not det
        light_done = fused_tail
        if gathered and need_vert and G == 1 and not fused_tail:
            # one pipeline: straight behind the gathered pass on its branch, beside the line walk, not behind the join
            with torch.cuda.stream(s_gath[0]):
                light_to_vertices(gl_g[0] if light_shared else grad_light)
            light_done = True
        # ... and so does the camera's adjoint, the step's last kernel, behind the gather: the side stream waits for the
        # forking stream's (shorter, by then finished) chain instead of the other way round, and the final join is the
        # end of the step
        tail_on_side = swap and gathered and ctx.camera is not None
        for k in range(G):
            if mains[k] is not cur:
                cur.wait_stream(mains[k])
            if tail_on_side:
                auxs[k].wait_stream(cur)
            elif gathered or plan_ready is not None:
                cur.wait_stream(auxs[k])
        if gathered:
            # shared textures / light: the groups' sums add up (a group's pass already summed over its views)
            if tex_shared:
                grad_textures = gt_g[0] if G == 1 else torch.stack(gt_g).sum(0)
            if need_vert and not light_done:
                if light_shared:
                    grad_light = gl_g[0] if G == 1 else torch.stack(gl_g).sum(0)
                light_to_vertices(grad_light)
            if not need_tex:
                grad_textures = None
        elif rd and need_geom:              # textures and lighting need no gradient: the depth term on its own
            if fit is not None:
                # d3m_backward_depth_map takes final maps: the fused objective left sign(depth - target) * mask, which
                # still lacks grad_loss / sum(mask) (GradScale::get, d3m_device.h); totals[2] of the scratch holds that
                # sum (the same in every group: with more than one the normaliser is the batch's mask_sum)
                g_depth_map = g_depth_map * (grad_loss / fit[7])              # mask_sum: set whenever gradients are wanted
            grad_faces = torch.zeros_like(faces)
            ops.backward_depth_map(faces, m["depth_map"], m["face_index_map"], m["face_inv_map"], m["weight_map"],
                                   g_depth_map, grad_faces, S)
            _lib.check(L.d3m_scatter_face_grads(_lib.ptr(grad_faces), _lib.ptr(tri), tri.shape[0], _lib.ptr(grad_sv), B, V,
                                                Ft, int(fill_back), _lib.stream_ptr()), "d3m_scatter_face_grads")
        if fused_tail:
            from . import cameras
            cam, _keep = cameras._camera_struct(ctx.camera, dev)
            _lib.check(L.d3m_lit_back(_lib.ptr(vertices), vertices.shape[0], ctypes.byref(cam), _lib.ptr(grad_sv),
                                      _lib.ptr(grad_vertices), B, V, _lib.ptr(tri), tri.shape[0], Ft, int(fill_back),
                                      _lib.ptr(gl_g[0] if light_shared else grad_light), Bl, ia, idr, _vec3_host(ca),
                                      _vec3_host(cd), _vec3_host(direction), _lib.stream_ptr()), "d3m_lit_back")
            grad_sv = None
        elif ctx.camera is not None and need_geom:
            # the camera's adjoint joins the light's in the same buffer (or writes it, when there is none)
            from . import cameras
            cam, _keep = cameras._camera_struct(ctx.camera, dev)
            if grad_vertices is None:
                grad_vertices = sink[0] if sink is not None else torch.empty_like(vertices)
                fn, what = L.d3m_camera_backward, "d3m_camera_backward"
            else:
                fn, what = L.d3m_camera_backward_add, "d3m_camera_backward_add"
            with torch.cuda.stream(auxs[0] if tail_on_side else cur):
                _lib.check(fn(_lib.ptr(vertices), vertices.shape[0], ctypes.byref(cam), _lib.ptr(grad_sv),
                              _lib.ptr(grad_vertices), B, V, _lib.stream_ptr()), what)
            grad_sv = None
        if tail_on_side:
            cur.wait_stream(auxs[0])
        return (grad_sv if need_geom else None, grad_vertices if need_geom else None, None, grad_textures) + (None,) * 17


class _RasterizeMeshModes(torch.autograd.Function):
    """render_silhouettes / render_depth (NR/renderer.py:114-183) of an indexed mesh as ONE node, camera inside: the
    reference's sequence for these modes -- look_at / look / projection, vertices_to_faces (+ the fill_back copy), the
    rasterizer with alpha or depth output, its backward and the gather's scatter-add adjoint -- run through the pipeline the
    lit node uses: the step's first launch transforms the vertices (d3m_lit_front), coverage reads the faces through the
    index tensor and leaves the dense copy, the visibility list and the edge plan behind, and backward adds the edge
    gradient (K4, alpha) and the depth gradient (K6, over the listed faces) straight into the screen-space gradient of
    the VERTICES (d3m_vertex_target), which the camera's adjoint takes to the mesh.  No [B,F',3,3] gather, no dense
    grad_faces, no scatter-add pass, no pass over face_index_map for the visibility flags, no epilogue pass without
    anti-aliasing: the 32-view silhouette step 1.57 -> 1.21 ms, the depth step 1.12 -> 0.66, same images bit for bit
    (tests/test_gpu_renderer.py; docs/EXPERIMENTS.md C)."""

    @staticmethod
    def forward(ctx, vertices, tri, camera, fill_back, image_size, anti_aliasing, near, far, eps, return_alpha, return_depth):
        from . import cameras
        L = _lib.lib()
        vertices = f32c(vertices)
        tri = tri.to(torch.int32).contiguous()
        dev = vertices.device
        cam, cam_keep = cameras._camera_struct(camera, dev)
        basis, basis_keep = cameras.basis_struct(camera)
        B, V, Ft = camera["batch"], vertices.shape[1], tri.shape[1]
        Fp = 2 * Ft if fill_back else Ft
        if vertices.shape[0] not in (1, B) or tri.shape[0] not in (1, B):
            raise ValueError("vertices must be [1 or B, V, 3] and faces [1 or B, F, 3]")
        S = int(image_size) * 2 if anti_aliasing else int(image_size)
        s_out = int(image_size)
        need_grad = ctx.needs_input_grad[0]
        sv = torch.empty(B, V, 3, dtype=torch.float32, device=dev)
        faces = torch.empty(B, Fp, 3, 3, dtype=torch.float32, device=dev)      # (the dense copy of faces that can own a pixel)
        fi = torch.empty(B, S, S, dtype=torch.int32, device=dev)
        wm = torch.empty(B, S, S, 3, dtype=torch.float32, device=dev)
        dm = torch.empty(B, S, S, dtype=torch.float32, device=dev)
        alpha_map = torch.empty(B, S, S, dtype=torch.float32, device=dev) if return_alpha else None
        alpha = torch.empty(B, s_out, s_out, dtype=torch.float32, device=dev) if return_alpha else None
        depth = torch.empty(B, s_out, s_out, dtype=torch.float32, device=dev) if return_depth else None
        vis = torch.empty(int(L.d3m_visibility_bytes(B, Fp)), dtype=torch.uint8, device=dev) if need_grad else None
        plan = torch.empty(int(L.d3m_edge_plan_bytes(B, Fp, S)), dtype=torch.uint8, device=dev) \
            if (need_grad and return_alpha) else None
        ws = ops._workspace("fwd", L.d3m_forward_workspace_bytes(B, Fp, S), dev)
        # the first launch: camera (+ basis) and the clears of the operators behind it
        clears, flags_fwd, flags_plan = [], 0, 0
        nb = int(L.d3m_forward_clear_bytes(B, Ft, int(bool(fill_back)), S, ws.numel()))
        if nb:
            clears.append((ws.data_ptr(), nb))
            flags_fwd = _lib.PRECLEARED
        if plan is not None:
            clears.append((plan.data_ptr(), int(L.d3m_edge_plan_clear_bytes(B, Fp, S))))
            flags_plan = _lib.PRECLEARED
        # ... and of the backward pass: its accumulator (and the large faces' counter behind it) is allocated here and zeroed
        # by the same launch -- a fill of 12 B per (view, vertex) less in backward (one use: a second backward over the same
        # graph allocates and clears its own)
        pre = None
        if need_grad:
            pre = torch.empty(B * V * 3 + 64, dtype=torch.float32, device=dev)
            clears.append(_lib.tensor_range(pre))
        zp = (ctypes.c_void_p * max(1, len(clears)))(*[c[0] for c in clears])
        zb = (ctypes.c_size_t * max(1, len(clears)))(*[c[1] for c in clears])
        zero3 = _vec3_host((0.0, 0.0, 0.0))
        _lib.check(L.d3m_lit_front(_lib.ptr(vertices), vertices.shape[0], ctypes.byref(cam),
                                   ctypes.byref(basis) if basis is not None else None, _lib.ptr(sv), B, V, _lib.ptr(tri),
                                   tri.shape[0], Ft, int(bool(fill_back)), None, 0, 0.0, 0.0, zero3, zero3, zero3, zp, zb,
                                   len(clears), _lib.stream_ptr()), "d3m_lit_front")
        # without anti-aliasing the output images are the maps with their rows reversed: coverage's last pass writes them too
        # (no epilogue pass); with it, the 2x2 pooling is a pass of its own
        in_pass = not anti_aliasing
        _lib.check(L.d3m_forward_face_index_map_mesh_modes(
            _lib.ptr(sv), _lib.ptr(tri), tri.shape[0], V, Ft, int(bool(fill_back)), _lib.ptr(faces), _lib.ptr(fi), _lib.ptr(wm),
            _lib.ptr(dm), None, B, S, float(near), float(far), _lib.ptr(ws), ws.numel(), _lib.ptr(vis),
            vis.numel() if vis is not None else 0, _lib.ptr(alpha_map if in_pass else None),
            _lib.ptr(alpha if in_pass else None), _lib.ptr(depth if in_pass else None), flags_fwd, _lib.stream_ptr()),
            "d3m_forward_face_index_map_mesh_modes")
        if not in_pass:
            _lib.check(L.d3m_output_epilogue(_lib.ptr(fi), None, _lib.ptr(dm if return_depth else None), None, 1, None,
                                             _lib.ptr(alpha_map), None, _lib.ptr(alpha), _lib.ptr(depth), B, S,
                                             int(bool(anti_aliasing)), _lib.stream_ptr()), "d3m_output_epilogue")
        if need_grad:
            # (the first step of the visibility list -- which faces own a pixel -- was left by the coverage pass)
            _lib.check(L.d3m_visibility(None, _lib.ptr(vis), vis.numel(), B, Fp, S, _lib.stream_ptr()), "d3m_visibility")
            if plan is not None:
                _lib.check(L.d3m_edge_plan(_lib.ptr(faces), _lib.ptr(fi), _lib.ptr(vis), _lib.ptr(plan), plan.numel(), B, Fp, S,
                                           flags_plan, _lib.stream_ptr()), "d3m_edge_plan")
        if need_grad and (return_alpha != return_depth) and tri.shape[0] == 1 and _deterministic():
            vertex_adjacency(tri, V)            # (the deterministic backward pass's CSR adjacency: built outside any capture of it)
        ctx.cfg = (B, V, Ft, Fp, S, bool(anti_aliasing), float(eps), bool(fill_back), bool(return_alpha), bool(return_depth))
        ctx.camera, ctx.keep = camera, (cam_keep, basis_keep)
        ctx.maps = (faces, fi, wm, dm, alpha_map, vis, plan)
        ctx.pre = pre
        ctx.save_for_backward(vertices, tri)
        empty = torch.tensor([])
        return (alpha if return_alpha else empty, depth if return_depth else empty)

    @staticmethod
    def backward(ctx, g_alpha, g_depth):
        from . import cameras
        L = _lib.lib()
        vertices, tri = ctx.saved_tensors
        B, V, Ft, Fp, S, aa, eps, fill_back, ra, rd = ctx.cfg
        faces, fi, wm, dm, alpha_map, vis, plan = ctx.maps
        dev = vertices.device
        acc, ctx.pre = getattr(ctx, "pre", None), None
        if acc is None:
            acc = torch.empty(B * V * 3 + 64, dtype=torch.float32, device=dev)
            _lib.zero_raw([_lib.tensor_range(acc)])
        grad_sv, counter = acc[:B * V * 3].view(B, V, 3), acc[B * V * 3:]
        target = _lib.D3MVertexTarget(_lib.ptr(grad_sv), _lib.ptr(tri), V, Ft, tri.shape[0], int(fill_back))
        g_depth_map = torch.empty(B, S, S, dtype=torch.float32, device=dev) if (rd and (ra or aa)) else None
        if ra and not rd:
            # silhouettes only: the edge gradient reads the image's gradient where it is -- through the row flip and the
            # pooling's adjoint -- and the owners from face_index_map; no per-pixel records (d3m_edge_grad.h, "DIRECT")
            g_img = f32c(g_alpha)
            unscaled = _lib.D3MFitTargets(None, None, None, None, None, None, None, _lib.ptr(g_img), None, None, None,
                                          None, None, None, None,
                                          _lib.GRAD_OF_OUTPUT_IMAGE | (_lib.FIT_POOLED if aa else 0))
            if tri.shape[0] == 1 and _deterministic():
                # DETERMINISTIC (as the lit node's): K4 into its own per-face array (plain stores), summed per vertex in a
                # fixed order over the index tensor's CSR adjacency -- bit-identical runs.  (One shared topology; per-view
                # topologies keep their float atomics: DESIGN.md section 6.)
                det_k4 = torch.empty(B, Fp, 3, 3, dtype=torch.float32, device=dev)
                _lib.zero_raw([_lib.tensor_range(det_k4)])
                ops.backward_pixel_map(faces, fi, None, alpha_map, None, None, det_k4, S, eps, False, True,
                                       visibility=vis, unscaled=unscaled, edge_plan=plan)
                adj_off, adj_items = vertex_adjacency(tri, V)
                _lib.check(L.d3m_vertex_gather(_lib.ptr(det_k4), None, _lib.ptr(adj_off), _lib.ptr(adj_items), _lib.ptr(grad_sv),
                                               B, V, Ft, int(fill_back), _lib.ptr(vis), _lib.stream_ptr()), "d3m_vertex_gather")
            else:
                ops.backward_pixel_map(faces, fi, None, alpha_map, None, None, None, S, eps, False, True, vertex_target=target,
                                       visibility=vis, unscaled=unscaled, edge_plan=plan)
        elif ra:
            # the adjoint of the output epilogue writes the alpha gradient straight as the edge gradient's per-pixel records
            # (and the depth gradient as the map K6 reads)
            records = (torch.empty(B, S, S, 4, dtype=torch.float32, device=dev),
                       torch.empty(B, S, S, 2, dtype=torch.float32, device=dev),
                       torch.empty(2, B, 2, S, dtype=torch.int32, device=dev))
            _lib.check(L.d3m_output_epilogue_backward_records(
                None, _lib.ptr(f32c(g_alpha)), _lib.ptr(f32c(g_depth) if rd else None), _lib.ptr(fi), None, _lib.ptr(alpha_map),
                _lib.ptr(records[0]), _lib.ptr(records[1]), _lib.ptr(records[2][0]), _lib.ptr(records[2][1]),
                _lib.ptr(g_depth_map), B, S, int(aa), _lib.stream_ptr()), "d3m_output_epilogue_backward_records")
            unscaled = _lib.D3MFitTargets(None, None, None, None, None, None, None, None, None, None, None,
                                          _lib.ptr(records[0]), _lib.ptr(records[1]), _lib.ptr(records[2][0]),
                                          _lib.ptr(records[2][1]), 0)
            ops.backward_pixel_map(faces, fi, None, alpha_map, None, None, None, S, eps, False, True, vertex_target=target,
                                   visibility=vis, unscaled=unscaled, edge_plan=plan)
        k6_flags = _lib.PRECLEARED
        det_depth = rd and not ra and tri.shape[0] == 1 and _deterministic()
        if det_depth:
            # DETERMINISTIC, depth mode: the adjoint of the output epilogue as a map, K6 gathered per face into its own array by
            # the reference-shaped operator (in this mode without its per-pixel fallback for large faces: d3m_backward_depth_map),
            # summed per vertex over the CSR adjacency
            g_depth_map = torch.empty(B, S, S, dtype=torch.float32, device=dev)
            _lib.check(L.d3m_output_epilogue_backward(None, None, _lib.ptr(f32c(g_depth)), None, None, _lib.ptr(g_depth_map),
                                                      B, S, int(aa), _lib.stream_ptr()), "d3m_output_epilogue_backward")
            det_k6 = torch.empty(B, Fp, 3, 3, dtype=torch.float32, device=dev)
            _lib.zero_raw([_lib.tensor_range(det_k6)])
            ops.backward_depth_map(faces, dm, fi, const_tensor((0.0,), dev), wm, g_depth_map, det_k6, S)
            adj_off, adj_items = vertex_adjacency(tri, V)
            _lib.check(L.d3m_vertex_gather(_lib.ptr(det_k6), None, _lib.ptr(adj_off), _lib.ptr(adj_items), _lib.ptr(grad_sv),
                                           B, V, Ft, int(fill_back), _lib.ptr(vis), _lib.stream_ptr()), "d3m_vertex_gather")
        elif rd and not ra:
            if aa:
                _lib.check(L.d3m_output_epilogue_backward(None, None, _lib.ptr(f32c(g_depth)), None, None, _lib.ptr(g_depth_map),
                                                          B, S, int(aa), _lib.stream_ptr()), "d3m_output_epilogue_backward")
            else:   # the output image is the depth map with its rows reversed: K6 reads the gradient through the flip
                g_depth_map, k6_flags = f32c(g_depth), k6_flags | _lib.GRAD_OF_OUTPUT_IMAGE
        if rd and not det_depth:
            _lib.check(L.d3m_backward_depth_map_mesh(_lib.ptr(faces), _lib.ptr(dm), _lib.ptr(fi), _lib.ptr(wm),
                                                     _lib.ptr(g_depth_map), B, Fp, S, ctypes.byref(target), _lib.ptr(vis),
                                                     _lib.ptr(counter), k6_flags, _lib.stream_ptr()),
                       "d3m_backward_depth_map_mesh")
        cam, _keep = cameras._camera_struct(ctx.camera, dev)
        grad_vertices = torch.empty_like(vertices)
        _lib.check(L.d3m_camera_backward(_lib.ptr(vertices), vertices.shape[0], ctypes.byref(cam), _lib.ptr(grad_sv),
                                         _lib.ptr(grad_vertices), B, V, _lib.stream_ptr()), "d3m_camera_backward")
        return (grad_vertices,) + (None,) * 10


def rasterize_mesh_modes(vertices, tri, camera, fill_back, image_size, anti_aliasing, return_alpha, return_depth,
                         near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS):
    """(alpha, depth) images -- either may be left out -- of an indexed mesh seen through `camera` (the parameter block of
    cameras.look_at_params & co): the silhouette / depth modes of the renderer as one node.  See _RasterizeMeshModes."""
    alpha, depth = _RasterizeMeshModes.apply(vertices, tri, camera, bool(fill_back), int(image_size), bool(anti_aliasing),
                                             float(near), float(far), float(eps), bool(return_alpha), bool(return_depth))
    return (alpha if return_alpha else None), (depth if return_depth else None)


class _ManualContext:
    """What _RasterizeLit.forward / _backward_halves use of an autograd context, for calling them WITHOUT the autograd
    engine (LitFitManual): needs_input_grad, save_for_backward / saved_tensors, free attributes."""

    def __init__(self, needs_input_grad):
        self.needs_input_grad = tuple(needs_input_grad)
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tuple(tensors)


class LitFitManual:
    """The fused fit objective of the lit render node (rasterize_lit_fit) driven by hand instead of by the autograd
    engine: forward(), then the two halves of backward as separate calls on the CALLER's thread and stream --

        loss = m.forward(...)                 # as rasterize_lit_fit(...)
        m.backward_texture_side(grad_loss)    # grad_textures final (in the grad_sink, if one was given)
        gv, gt = m.backward_geometry_side()   # grad_vertices final

    -- so that a step can be captured as TWO HIP graphs with a collective issued between them (a capture cannot be ended
    on autograd's worker thread, and one autograd node cannot be split).  Same kernels, same numbers as the autograd
    node: it is that node's code (tests/test_gpu_multirank.py)."""

    def __init__(self, vertices_grad=True, textures_grad=True):
        self._needs = (False, bool(vertices_grad), False, bool(textures_grad)) + (False,) * 17
        self._ctx = self._halves = None

    def forward(self, vertices, tri, textures, light_cfg, fill_back, targets, image_size, near, far, eps, background_color,
                camera, grad_sink=None, images_out=None, anti_aliasing=False):
        fit = tuple(targets)
        if images_out is not None:
            fit = (fit + (None,))[:5] + (tuple(images_out),)
        self._ctx = _ManualContext(self._needs)
        self._ctx.force_serial = True       # every kernel on the caller's stream: the step is cut BETWEEN kernels of one stream
        with torch.no_grad():
            return _RasterizeLit.forward(self._ctx, None, vertices, tri, textures, light_cfg, fill_back, image_size,
                                         bool(anti_aliasing), near, far, eps, background_color, True, True, True, fit, 1,
                                         False, camera, grad_sink)

    def backward_texture_side(self, grad_loss):
        with torch.no_grad():
            self._halves = _RasterizeLit._backward_halves(self._ctx, grad_loss)
            next(self._halves)

    def backward_geometry_side(self):
        with torch.no_grad():
            try:
                next(self._halves)
            except StopIteration as done:
                grads = done.value
                self._ctx = self._halves = None
                return grads[1], grads[3]           # (grad_vertices, grad_textures)
        raise RuntimeError("_backward_halves yields once")


def rasterize_lit(screen_vertices, vertices, tri, textures, light_cfg, fill_back, image_size=DEFAULT_IMAGE_SIZE,
                  anti_aliasing=DEFAULT_ANTI_ALIASING, near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS,
                  background_color=DEFAULT_BACKGROUND_COLOR, return_alpha=True, return_depth=True, view_groups=1,
                  camera=None, fit_hint=None, defer_plan_join=False):
    """rgb (+alpha, depth) images of the mesh (`screen_vertices` [B,V,3] after the camera transform, triangles
    `tri` [1|B,F,3]) textured with the ORIGINAL `textures` [1|B,F,ts,ts,ts,3]; the fill_back copy and the per-face
    light (computed from world-space `vertices`) are applied on the fly.  Same outputs as vertices_to_faces() +
    lighting() + rasterize_rgbad() on the materialised arrays.  `camera` (with screen_vertices None): the fused camera
    kernels' parameter block (cameras.look_at_params) -- the transform then runs inside the node.  `fit_hint` = the targets
    (rgb, depth, alpha, mask[, mask_sum]) of a multi-view fit objective the caller is about to evaluate on the returned
    images: see _RasterizeLit.forward, "a REGISTERED objective"."""
    rgb, alpha, depth = _RasterizeLit.apply(screen_vertices, vertices, tri, textures, light_cfg, fill_back, image_size,
                                            anti_aliasing, near, far, eps, background_color, True, return_alpha,
                                            return_depth, None, view_groups, defer_plan_join, camera, None, fit_hint)
    return {'rgb': rgb, 'alpha': alpha if return_alpha else None, 'depth': depth if return_depth else None}


def rasterize_lit_image_grid(screen_vertices, vertices, grid_hw, image, light_cfg, image_size=DEFAULT_IMAGE_SIZE,
                             anti_aliasing=DEFAULT_ANTI_ALIASING, near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS,
                             background_color=DEFAULT_BACKGROUND_COLOR):
    """rgb [B,3,s,s] of a depth map's grid mesh textured by an image -- what NrRenderer's view-synthesis frames render
    (deep3dmap/core/renderer/renderer_nr.py:196-198: get_face_idx + get_textures_from_im(tx_size=2) + render_rgb with
    fill_back) -- with NEITHER array: the triangles are the implicit topology of the (h, w) grid (`grid_hw`), their texture
    cubes are evaluated from `image` [B,3,h,w] inside the sampler.  Same pixels as rasterize_lit() on the materialised
    arrays.  Forward only (visualisation frames): with gradients, the cube array is backward's intermediate anyway and the
    caller takes the materialised route."""
    if torch.is_grad_enabled() and any(t.requires_grad for t in (screen_vertices, vertices, image)):
        raise RuntimeError("rasterize_lit_image_grid is a forward-only path")
    L = _lib.lib()
    sv, vertices, im = f32c(screen_vertices), f32c(vertices), f32c(image)
    h, w = grid_hw
    B, V, dev = sv.shape[0], sv.shape[1], sv.device
    if V != h * w or tuple(im.shape) != (B, 3, h, w) or vertices.shape[1] != V:
        raise ValueError("vertices must be [B,h*w,3] and the image [B,3,h,w]")
    Ft = 2 * (h - 1) * (w - 1)
    Fp = 2 * Ft
    S = int(image_size) * 2 if anti_aliasing else int(image_size)
    s_out = int(image_size)
    ia, idr, ca, cd, direction = light_cfg
    light = torch.empty(vertices.shape[0], Fp, 3, dtype=torch.float32, device=dev)
    _lib.check(L.d3m_face_light(_lib.ptr(vertices), vertices.shape[0], None, -w, _lib.ptr(light), float(ia), float(idr),
                                _vec3_host(ca), _vec3_host(cd), _vec3_host(direction), vertices.shape[0], V, Ft, 1,
                                _lib.stream_ptr()), "d3m_face_light")
    faces = torch.empty(B, Fp, 3, 3, dtype=torch.float32, device=dev)
    fim = torch.empty((B, S, S), dtype=torch.int32, device=dev)
    wm = torch.empty(B, S, S, 3, dtype=torch.float32, device=dev)
    dm = torch.empty((B, S, S), dtype=torch.float32, device=dev)
    ws = ops._workspace("fwd", L.d3m_forward_workspace_bytes(B, Fp, S), dev)
    _lib.check(L.d3m_forward_face_index_map_mesh(_lib.ptr(sv), None, -w, V, Ft, 1, _lib.ptr(faces), _lib.ptr(fim), _lib.ptr(wm),
                                                 _lib.ptr(dm), None, B, S, float(near), float(far), _lib.ptr(ws), ws.numel(),
                                                 None, 0, 0, _lib.stream_ptr()), "d3m_forward_face_index_map_mesh")
    background = _background_tensor(background_color, dev)
    rgb_map = torch.empty(B, S, S, 3, dtype=torch.float32, device=dev)
    rgb = torch.empty(B, 3, s_out, s_out, dtype=torch.float32, device=dev)
    _lib.check(L.d3m_render_lit_epilogue(
        _lib.ptr(faces), _lib.ptr(im), -w, _lib.ptr(light), light.shape[0], _lib.ptr(fim), _lib.ptr(wm), _lib.ptr(dm),
        _lib.ptr(background), background.shape[0], _lib.ptr(rgb_map), None, _lib.ptr(rgb), None, None, B, Ft, 1, S, 2,
        float(eps), int(bool(anti_aliasing)), None, _lib.stream_ptr()), "d3m_render_lit_epilogue")
    return rgb


def rasterize_lit_fit(screen_vertices, vertices, tri, textures, light_cfg, fill_back, targets, image_size=DEFAULT_IMAGE_SIZE,
                      near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS, background_color=DEFAULT_BACKGROUND_COLOR,
                      view_groups=1, defer_plan_join=False, images_out=None, camera=None, grad_sink=None,
                      anti_aliasing=False):
    """The multi-view fit objective of the images rasterize_lit() would return (targets at the output size; with
    anti_aliasing the objective of the 2x2-pooled images),

        photometric_loss(rgb, rgb_t, mask) + sum((alpha - alpha_t)^2) / S^2 + photometric_loss(depth, depth_t, mask),

    `targets` = (rgb_t [B,3,S,S], depth_t [B,S,S], alpha_t [B,S,S], mask [B,S,S][, mask_sum]), evaluated inside the
    rendering node (the optional device scalar `mask_sum` replaces sum(mask) as the photometric normaliser: the sum over
    ALL ranks' views when these views are one shard of a camera-sharded fit, so that values and gradients add up): the sums are taken where the images are produced and the gradient is written straight into the
    internal-resolution maps, so the images and their gradients never exist in memory.  Same value and gradients as
    core.losses.multiview_fit_loss(*rasterize_lit(...), ...)."""
    fit = tuple(targets)
    if images_out is not None:          # (rgb, depth, alpha) buffers the same pass fills with the images render() returns
        fit = (fit + (None,))[:5] + (tuple(images_out),)
    return _RasterizeLit.apply(screen_vertices, vertices, tri, textures, light_cfg, fill_back, image_size,
                               bool(anti_aliasing), near, far, eps, background_color, True, True, True, fit, view_groups,
                               defer_plan_join, camera, grad_sink)


def rasterize_rgbad(
        faces,
        textures=None,
        image_size=DEFAULT_IMAGE_SIZE,
        anti_aliasing=DEFAULT_ANTI_ALIASING,
        near=DEFAULT_NEAR,
        far=DEFAULT_FAR,
        eps=DEFAULT_EPS,
        background_color=DEFAULT_BACKGROUND_COLOR,
        return_rgb=True,
        return_alpha=True,
        return_depth=True,
):
    """
    Generate RGB, alpha channel, and depth images from faces and textures (NR/rasterize.py:253-334).

    Args:
        faces (torch.Tensor): [batch size, number of faces, 3 (vertices), 3 (XYZ)].
        textures (torch.Tensor): [batch size, number of faces, ts, ts, ts, 3 (RGB)].
        image_size (int): width and height of rendered images.
        anti_aliasing (bool): 2x super-sampling.
        near, far (float): depth range that is drawn.
        eps (float): epsilon of the approximate edge gradient.
        background_color: colour (3) or one colour per batch entry (B, 3).

    Returns:
        dict: 'rgb' [B,3,s,s], 'alpha' [B,s,s], 'depth' [B,s,s] (None for disabled outputs); row 0 is the
        TOP of the image.
    """
    if not faces.is_cuda or (textures is not None and not textures.is_cuda):
        raise TypeError('Rasterize module supports only cuda Tensors')
    rgb, alpha, depth = _RasterizeImages.apply(faces, textures, image_size, anti_aliasing, near, far, eps,
                                               background_color, return_rgb, return_alpha, return_depth)
    return {
        'rgb': rgb if return_rgb else None,
        'alpha': alpha if return_alpha else None,
        'depth': depth if return_depth else None,
    }


def rasterize(faces, textures, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING,
              near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS, background_color=DEFAULT_BACKGROUND_COLOR):
    """RGB images [B,3,s,s] (NR/rasterize.py:367-394)."""
    return rasterize_rgbad(
        faces, textures, image_size, anti_aliasing, near, far, eps, background_color, True, False, False)['rgb']


def rasterize_silhouettes(faces, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING,
                          near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS):
    """Alpha channels [B,s,s] (NR/rasterize.py:397-420)."""
    return rasterize_rgbad(faces, None, image_size, anti_aliasing, near, far, eps, None, False, True, False)['alpha']


def rasterize_depth(faces, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING,
                    near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS):
    """Depth images [B,s,s] (NR/rasterize.py:423-446)."""
    return rasterize_rgbad(faces, None, image_size, anti_aliasing, near, far, eps, None, False, False, True)['depth']
