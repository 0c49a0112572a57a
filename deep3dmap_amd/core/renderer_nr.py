"""NrRenderer: deep3dmap's adapter from a canonical depth map to the mesh renderer
(deep3dmap/core/renderer/renderer_nr.py:12-277; "CR" below), same constructor keys, attributes and methods.

Built differently from the reference's chain of eager tensor ops: every "back-project the depth map, move the points
rigidly, maybe project them again" of the class is ONE pass of d3m_grid_warp over the pixels.  Rigid moves about the
rotation centre c = (0, 0, rot_center_depth) -- rotate_pts, translate_pts, their inverses and chains of them -- are
composed on [B,3,3] / [B,1,3] into a single `Rigid` first (tiny tensors; the view's gradient flows through that
composition), normals are one pass (d3m_depth_normals), image -> face textures one pass (d3m_textures_from_im), and the
view sweeps of render_yaw / render_view / render_given_view share one frame loop."""
import ctypes
import math
import warnings

import torch

from .. import _lib
from .. import neural_renderer as nr
from ..neural_renderer._util import const_tensor, f32c
from .renderer_utils import get_face_idx, get_textures_from_im, get_transform_matrices

EPS = 1e-7


class Rigid:
    """p -> A (p - c) + c + t with A [B,3,3], t [B,1,3] (row vectors, as CR stores trans_xyz) about the centre c."""

    def __init__(self, A, t):
        self.A, self.t = A, t

    @staticmethod
    def identity(device):
        return Rigid(torch.eye(3, dtype=torch.float32, device=device)[None], torch.zeros(1, 1, 3, dtype=torch.float32, device=device))

    @staticmethod
    def of_view(view):
        """rotate_pts(rot_mat) then translate_pts(trans_xyz) of a view vector (CR:95-100)."""
        return Rigid(*get_transform_matrices(view))

    def inverse(self):
        """translate_pts(-t) then rotate_pts(A^T) (CR:102-107): A^T (p - t - c) + c."""
        return Rigid(self.A.transpose(2, 1), -self.t.matmul(self.A))

    def then(self, other):
        """self first, `other` second: other.A (self.A (p-c) + self.t) + c + other.t."""
        return Rigid(other.A.matmul(self.A), self.t.matmul(other.A.transpose(2, 1)) + other.t)


class _GridWarp(torch.autograd.Function):
    """d3m_grid_warp: depth [B,H,W] -> Q = A (depth Kinv (x,y,1) - c) + c + t as points [B,H*W,3], or with K as the
    normalised sampling grid [B,H,W,2] (CR:74-88); analytic adjoint for the depth map, A and t."""

    @staticmethod
    def forward(ctx, depth, inv_K, A, t, center_z, K, crop):
        d, iK = f32c(depth), f32c(inv_K)
        B, H, W = d.shape
        A_b = f32c(A.expand(B, 3, 3))
        t_b = f32c(t.reshape(-1, 3).expand(B, 3))
        Kc = f32c(K) if K is not None else None
        out = torch.empty((B, H, W, 2) if Kc is not None else (B, H * W, 3), dtype=torch.float32, device=d.device)
        crop_c = (ctypes.c_int * 4)(*[int(v) for v in crop]) if crop is not None else None
        _lib.check(_lib.lib().d3m_grid_warp(_lib.ptr(d), _lib.ptr(iK), iK.shape[0], _lib.ptr(A_b), _lib.ptr(t_b),
                                            float(center_z), _lib.ptr(Kc), Kc.shape[0] if Kc is not None else 1, crop_c,
                                            _lib.ptr(out), B, H, W, _lib.stream_ptr()), "d3m_grid_warp")
        ctx.save_for_backward(d, iK, A_b, t_b, Kc)
        ctx.center_z = float(center_z)
        ctx.shapes = (tuple(A.shape), tuple(t.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        d, iK, A_b, t_b, Kc = ctx.saved_tensors
        B, H, W = d.shape
        gd = torch.empty_like(d) if ctx.needs_input_grad[0] else None
        gA = torch.empty(B, 3, 3, dtype=torch.float32, device=d.device) if ctx.needs_input_grad[2] else None
        gt = torch.empty(B, 3, dtype=torch.float32, device=d.device) if ctx.needs_input_grad[3] else None
        _lib.check(_lib.lib().d3m_grid_warp_backward(
            _lib.ptr(d), _lib.ptr(iK), iK.shape[0], _lib.ptr(A_b), _lib.ptr(t_b), ctx.center_z, _lib.ptr(Kc),
            Kc.shape[0] if Kc is not None else 1, _lib.ptr(f32c(g)), _lib.ptr(gd), _lib.ptr(gA), _lib.ptr(gt), B, H, W,
            _lib.stream_ptr()), "d3m_grid_warp_backward")
        A_shape, t_shape = ctx.shapes
        if gA is not None and A_shape[0] != B:
            gA = gA.sum(0, keepdim=True)
        if gt is not None:
            gt = (gt.sum(0, keepdim=True) if t_shape[0] != B else gt).reshape(t_shape)
        return gd, None, gA, gt, None, None, None


class _DepthNormals(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, inv_K):
        d, iK = f32c(depth), f32c(inv_K)
        B, H, W = d.shape
        out = torch.empty(B, H, W, 3, dtype=torch.float32, device=d.device)
        _lib.check(_lib.lib().d3m_depth_normals(_lib.ptr(d), _lib.ptr(iK), iK.shape[0], _lib.ptr(out), B, H, W,
                                                _lib.stream_ptr()), "d3m_depth_normals")
        ctx.save_for_backward(d, iK)
        return out

    @staticmethod
    def backward(ctx, g):
        d, iK = ctx.saved_tensors
        B, H, W = d.shape
        gd = torch.empty_like(d)
        _lib.check(_lib.lib().d3m_depth_normals_backward(_lib.ptr(d), _lib.ptr(iK), iK.shape[0], _lib.ptr(f32c(g)),
                                                         _lib.ptr(gd), B, H, W, _lib.stream_ptr()),
                   "d3m_depth_normals_backward")
        return gd, None


class _WarpResample(torch.autograd.Function):
    """d3m_warp_resample: the frames of CR:180-184 -- the sampling grid of a target-view depth map under a rigid motion
    and the lookups F.grid_sample(im, grid, 'bilinear') [and F.grid_sample(mask, grid, 'nearest')] -- in one pass; the
    adjoint reaches the image, the depth map and the motion (A, t)."""

    @staticmethod
    def forward(ctx, depth, im, mask, inv_K, K, A, t, center_z):
        d, src = f32c(depth), f32c(im)
        B, h, w = d.shape
        _, C, H, W = src.shape
        iK, Kc = f32c(inv_K), f32c(K)
        A_b, t_b = f32c(A.expand(B, 3, 3)), f32c(t.reshape(-1, 3).expand(B, 3))
        msk = f32c(mask) if mask is not None else None
        out = torch.empty(B, C, h, w, dtype=torch.float32, device=d.device)
        out_m = torch.empty(B, msk.shape[1], h, w, dtype=torch.float32, device=d.device) if msk is not None else None
        _lib.check(_lib.lib().d3m_warp_resample(
            _lib.ptr(d), _lib.ptr(iK), iK.shape[0], _lib.ptr(Kc), Kc.shape[0], _lib.ptr(A_b), _lib.ptr(t_b), float(center_z),
            _lib.ptr(src), C, _lib.ptr(msk), msk.shape[1] if msk is not None else 0, _lib.ptr(out), _lib.ptr(out_m), B, h, w,
            H, W, _lib.stream_ptr()), "d3m_warp_resample")
        ctx.save_for_backward(d, src, iK, Kc, A_b, t_b)
        ctx.center_z, ctx.shapes = float(center_z), (tuple(A.shape), tuple(t.shape))
        if out_m is None:
            return out
        ctx.mark_non_differentiable(out_m)             # mode='nearest' has no gradient (nor does torch's)
        return out, out_m

    @staticmethod
    def backward(ctx, g, _g_mask=None):
        d, src, iK, Kc, A_b, t_b = ctx.saved_tensors
        B, h, w = d.shape
        _, C, H, W = src.shape
        L = _lib.lib()
        need = ctx.needs_input_grad
        g_src = torch.zeros_like(src) if need[1] else None
        g_d = torch.empty_like(d) if need[0] else None
        parts = torch.empty(B, int(L.d3m_warp_resample_partials(h, w)), 12, dtype=torch.float32, device=d.device)
        _lib.check(L.d3m_warp_resample_backward(
            _lib.ptr(d), _lib.ptr(iK), iK.shape[0], _lib.ptr(Kc), Kc.shape[0], _lib.ptr(A_b), _lib.ptr(t_b), ctx.center_z,
            _lib.ptr(src), C, _lib.ptr(f32c(g)), _lib.ptr(g_src), _lib.ptr(g_d), _lib.ptr(parts), B, h, w, H, W,
            _lib.stream_ptr()), "d3m_warp_resample_backward")
        gA = gt = None
        if need[5] or need[6]:
            sums = parts.sum(1)
            A_shape, t_shape = ctx.shapes
            gA = sums[:, :9].reshape(B, 3, 3)
            gA = gA.sum(0, keepdim=True) if A_shape[0] != B else gA
            gt = sums[:, 9:]
            gt = (gt.sum(0, keepdim=True) if t_shape[0] != B else gt).reshape(t_shape)
        return g_d, g_src, None, None, None, gA if need[5] else None, gt if need[6] else None, None


class NrRenderer():
    def __init__(self, cfgs, image_size):
        self.device = cfgs.get('device', 'cpu')
        self.image_size = image_size
        self.min_depth = cfgs.get('min_depth', 0.9)
        self.max_depth = cfgs.get('max_depth', 1.1)
        self.rot_center_depth = cfgs.get('rot_center_depth', (self.min_depth + self.max_depth) / 2)
        self.fov = cfgs.get('fov', 10)
        self.tex_cube_size = cfgs.get('tex_cube_size', 2)
        self.renderer_min_depth = cfgs.get('renderer_min_depth', 0.1)
        self.renderer_max_depth = cfgs.get('renderer_max_depth', 10.)
        self._view = self._rot_mat = self._trans_xyz = None

        # camera intrinsics: d * K^-1 (u, v, 1)^T = (x, y, z)^T  (CR:24-46)
        R = torch.eye(3, dtype=torch.float32)[None].cuda()
        t = torch.zeros(1, 3, dtype=torch.float32).cuda()
        fx = fy = (self.image_size - 1) / 2 / (math.tan(self.fov / 2 * math.pi / 180))
        cx = cy = (self.image_size - 1) / 2
        K = torch.tensor([[fx, 0., cx], [0., fy, cy], [0., 0., 1.]], dtype=torch.float32)
        # (inverted on the host: a 3x3 inverse is the same few flops anywhere, and the host's LAPACK is the one the CPU
        #  oracle uses -- identical bits in inv_K mean identical back-projected vertices)
        self.inv_K_origin = torch.inverse(K).contiguous().unsqueeze(0).cuda()    # (LAPACK returns column-major strides: contiguous once, not per call)
        self.K_origin = K.unsqueeze(0).cuda()
        self.inv_K = self.inv_K_origin.clone()
        self.K = self.K_origin.clone()
        self.renderer = nr.Renderer(camera_mode='projection',
                                    light_intensity_ambient=1.0,
                                    light_intensity_directional=0.,
                                    K=self.K, R=R, t=t,
                                    near=self.renderer_min_depth, far=self.renderer_max_depth,
                                    image_size=self.image_size, orig_size=self.image_size,
                                    fill_back=True,
                                    background_color=[1, 1, 1])

    def downscale_K(self, downscale):
        if downscale > 1:
            self.K = torch.cat((self.K_origin[:, 0:2] / downscale, self.K_origin[:, 2:]), dim=1)
            self.inv_K = torch.inverse(self.K[0].cpu()).contiguous().unsqueeze(0).to(self.K.device)

    def set_transform_matrices(self, view):
        """CR:61-62.  A view on the GPU is kept as it is until rot_mat / trans_xyz are looked at (one fused launch then):
        reconstruct() feeds it to its first pass instead, which computes (R, t) among other things.

        ALIASING CONTRACT (differs from the reference, which computes (R, t) here and now): the tensor itself is kept, not
        a copy, and it is READ when (R, t) are first needed -- by reconstruct(), a render_* method or a look at rot_mat /
        trans_xyz.  A caller that refills `view` IN PLACE between this call and that read (a preallocated buffer under
        graph replay) gets the transform of the NEW contents -- which is what a replayed training step wants, and why no
        snapshot is taken (a clone would be one more launch in the 9-launch block) -- ; pass `view.clone()` to pin the
        values of the moment.  Assigning rot_mat or trans_xyz first resolves a pending view (the other half of the pair
        still comes from it)."""
        if torch.is_tensor(view) and view.is_cuda and view.dim() == 2 and view.size(1) in (3, 5, 6):
            self._view, self._rot_mat, self._trans_xyz = view, None, None
        else:
            self._view = None
            self._rot_mat, self._trans_xyz = get_transform_matrices(view)

    def _materialise_view(self):
        if self._rot_mat is None and getattr(self, "_view", None) is not None:
            self._rot_mat, self._trans_xyz = get_transform_matrices(self._view)
            self._view = None

    def _pending_view(self):
        """the view vectors of set_transform_matrices() if nothing has asked for (R, t) since, else None"""
        return getattr(self, "_view", None)

    def _resolve_view(self, view, R, t):
        """reconstruct() has computed the pending view's (R, t) (detached values: the view's gradient went through the
        block); later readers of rot_mat / trans_xyz get the differentiable form on demand"""
        if self._pending_view() is view and not view.requires_grad:
            self._rot_mat, self._trans_xyz, self._view = R, t[:, None, :], None

    @property
    def rot_mat(self):
        self._materialise_view()
        return self._rot_mat

    @rot_mat.setter
    def rot_mat(self, value):
        self._materialise_view()
        self._rot_mat = value

    @property
    def trans_xyz(self):
        self._materialise_view()
        return self._trans_xyz

    @trans_xyz.setter
    def trans_xyz(self, value):
        self._materialise_view()
        self._trans_xyz = value

    # ---- the one pass everything below goes through ---------------------------------------------------------------
    def _warp(self, depth, rigid, project=False, crop=None):
        """depth [b,h,w] -> rigidly moved back-projection: points [b,h*w,3], or (project) the sampling grid [b,h,w,2]."""
        dev = depth.device
        A, t = rigid.A.to(dev), rigid.t.to(dev)
        if crop is not None and torch.is_grad_enabled() and any(x.requires_grad for x in (depth, A, t)):
            # crop_mesh (CR:145-158) is a visualisation option: the reference's in-place border copies are differentiable,
            # this pass has no adjoint for them -- the frame is rendered, without a graph, and says so (it does not raise:
            # visualisation code that forgot torch.no_grad() keeps working)
            warnings.warn("render_yaw(crop_mesh=...): no gradient flows through a cropped mesh; inputs detached",
                          RuntimeWarning, stacklevel=3)
            depth, A, t = depth.detach(), A.detach(), t.detach()
        return _GridWarp.apply(depth, self.inv_K.to(dev), A, t, self.rot_center_depth, self.K.to(dev) if project else None,
                               crop)

    def _current(self):
        return Rigid(self.rot_mat, self.trans_xyz)

    # ---- point-wise helpers of the reference's public surface (CR:64-72, 82-88) -----------------------------------
    def rotate_pts(self, pts, rot_mat):
        centroid = const_tensor([0., 0., self.rot_center_depth], pts.device, (1, 1, 3))
        return torch.baddbmm(centroid.expand(pts.shape[0], 1, 3), pts - centroid, rot_mat.transpose(2, 1).expand(pts.shape[0], 3, 3))

    def translate_pts(self, pts, trans_xyz):
        return pts + trans_xyz

    def grid_3d_to_2d(self, grid_3d):
        """camera-space points [b,h,w,3] -> normalised image coordinates in -1..1 (CR:82-88)."""
        b, h, w, _ = grid_3d.shape
        K = self.K.to(grid_3d.device)
        ndc = grid_3d[..., :2] / grid_3d[..., 2:]
        uv = ndc.matmul(K[:, :2, :2].transpose(2, 1)) + K[:, :2, 2].reshape(-1, 1, 1, 2)
        return uv * const_tensor([2. / (w - 1), 2. / (h - 1)], grid_3d.device, (1, 1, 1, 2)) - 1.

    # ---- grids (CR:74-114) ------------------------------------------------------------------------------------------
    def depth_to_3d_grid(self, depth):
        b, h, w = depth.shape
        return self._warp(depth, Rigid.identity(depth.device)).reshape(b, h, w, 3)

    def get_warped_3d_grid(self, depth):
        b, h, w = depth.shape
        return self._warp(depth, self._current()).reshape(b, h, w, 3)

    def get_inv_warped_3d_grid(self, depth):
        b, h, w = depth.shape
        return self._warp(depth, self._current().inverse()).reshape(b, h, w, 3)

    def get_warped_2d_grid(self, depth):
        return self._warp(depth, self._current(), project=True)

    def get_inv_warped_2d_grid(self, depth):
        return self._warp(depth, self._current().inverse(), project=True)

    def warp_canon_depth(self, canon_depth):
        b, h, w = canon_depth.shape
        warped_depth = self.renderer.render_depth(self._warp(canon_depth, self._current()),
                                                  get_face_idx(b, h, w, canon_depth.device))
        margin = (self.max_depth - self.min_depth) / 2          # allow some margin out of valid range (CR:122-124)
        return warped_depth.clamp(min=self.min_depth - margin, max=self.max_depth + margin)

    def get_normal_from_depth(self, depth):
        return _DepthNormals.apply(depth, self.inv_K.to(depth.device))

    def reconstruct(self, depth, albedo, light_a, light_b, light_d, input_im=None, flip=False, extra_mask=None,
                    with_smooth=True, lam_smooth=None):
        """The renderer block of the gan2shape step (models/frameworks/gan2shape.py:463-497) as one autograd node on fused
        HIP passes (core/gan2shape_block.py): shading and texture from (depth, albedo, light), recon_depth by warping the
        depth's mesh with the current view (set_transform_matrices), recon_im by looking the texture up through the
        inverse warp, the border mask, the masked-L1 terms against `input_im` [b,3,h,w] and the two smooth losses.
        depth [B,h,w], albedo [B,3,h,w], light_a / light_b [B,1], light_d [B,3]; with `flip`, B = 2b and entries
        (i, i + b) are an image and its mirror (gan2shape.py:431,449,479-481).  Returns a namespace with normal,
        diffuse_shading, texture, recon_depth, recon_im, recon_im_mask, loss_l1_im, loss_l1_im_flip, loss_smooth and --
        given lam_smooth -- loss_total = loss_l1_im (+ loss_l1_im_flip) + lam_smooth * loss_smooth.  Gradients reach
        depth, albedo, the light terms and the view (through rot_mat / trans_xyz) from recon_im and the losses."""
        from .gan2shape_block import reconstruct
        return reconstruct(self, depth, albedo, light_a, light_b, light_d, input_im, flip, extra_mask, with_smooth,
                           lam_smooth)

    # ---- view synthesis (CR:141-277): one frame loop for the three public sweeps ----------------------------------------
    def _frame_mesh(self, images, depth, rigid, crop=None):
        """render the depth's grid mesh, moved by `rigid`, textured with each image of `images` (CR:196-198)."""
        b, _, h, w = images[0].shape
        vertices = self._warp(depth, rigid, crop=crop)
        r = self.renderer
        # frames that need no gradient (the usual case: visualisation) are rendered from the image itself: implicit grid
        # topology, texture cubes evaluated in the sampler -- no get_face_idx / get_textures_from_im arrays
        direct = (self.tex_cube_size == 2 and r.fill_back and r._on_the_fly() and
                  not (torch.is_grad_enabled() and (vertices.requires_grad or any(im.requires_grad for im in images))))
        out = []
        for im in images:
            if direct and im.shape[1] == 3:
                out.append(r.render_rgb_image_grid(vertices, im).clamp(min=-1., max=1.))
            else:
                out.append(r.render_rgb(vertices, get_face_idx(b, h, w, depth.device),
                                        get_textures_from_im(im, tx_size=self.tex_cube_size)).clamp(min=-1., max=1.))
        return out

    def _frame_resample(self, im, depth, view, mask=None):
        """the grid_sample form (CR:180-184): warp the depth to `view`, look every target pixel up in the source image."""
        self.set_transform_matrices(view)
        back = self._current().inverse()                     # get_inv_warped_2d_grid's motion (CR:102-107)
        dev = im.device
        return _WarpResample.apply(self.warp_canon_depth(depth), im, mask, self.inv_K.to(dev), self.K.to(dev),
                                   back.A.to(dev), back.t.to(dev), self.rot_center_depth)

    def _sweep(self, im, depth, euler_angles, v_before, v_after, grid_sample, crop=None):
        dev = im.device
        undo = Rigid.of_view(v_before).inverse() if v_before is not None else None          # the "inverse warp" (CR:163-167)
        frames = []
        for i, angles in enumerate(euler_angles):
            if grid_sample:
                view = torch.tensor(list(angles) + [0., 0., 0.], dtype=torch.float32, device=dev).view(1, 6)
                frames.append(self._frame_resample(im, depth, view - v_before if v_before is not None else view))
                continue
            turn = Rigid.of_view(torch.tensor(list(angles), dtype=torch.float32, device=dev).view(1, 3))
            rigid = undo.then(turn) if undo is not None else turn
            if v_after is not None:
                rigid = rigid.then(Rigid.of_view(v_after[i] if v_after.dim() == 3 else v_after))
            frames.append(self._frame_mesh([im], depth, rigid, crop)[0])
        return torch.stack(frames, 1)  # b x t x c x h x w

    def render_yaw(self, im, depth, v_before=None, v_after=None, rotations=None, maxr=90, nsample=9,
                   grid_sample=False, crop_mesh=None):
        if rotations is None:
            rotations = torch.linspace(-math.pi / 180 * maxr, math.pi / 180 * maxr, nsample)
        crop = None if (crop_mesh is None or grid_sample) else tuple(int(v) for v in crop_mesh)
        return self._sweep(im, depth, [(0., float(r), 0.) for r in rotations], v_before, v_after, grid_sample, crop)

    def render_view(self, im, depth, v_before=None, rotations=None, maxr=[20, 90], nsample=[5, 9], grid_sample=False):
        pitch = torch.linspace(-math.pi / 180 * maxr[0], math.pi / 180 * maxr[0], nsample[0])
        yaw = torch.linspace(-math.pi / 180 * maxr[1], math.pi / 180 * maxr[1], nsample[1])
        angles = [(0., float(a), 0.) for a in yaw] + [(float(a), 0., 0.) for a in pitch]     # yaw sweep, then pitch (CR:214,232)
        return self._sweep(im, depth, angles, v_before, None, grid_sample)

    def render_given_view(self, im, depth, view, mask=None, grid_sample=True):
        if grid_sample:
            return self._frame_resample(im, depth, view, mask)
        frames = self._frame_mesh([im] if mask is None else [im, mask], depth, Rigid.of_view(view))
        return frames[0] if mask is None else (frames[0], frames[1])  # b x c x h x w
