"""NrRenderer: deep3dmap's adapter from a canonical depth map to the mesh renderer
(deep3dmap/core/renderer/renderer_nr.py:12-277), same constructor keys, attributes and methods."""
import math

import torch
import torch.nn as nn

from .. import _lib
from .. import neural_renderer as nr
from ..neural_renderer._util import const_tensor, f32c
from .renderer_utils import get_face_idx, get_grid, get_textures_from_im, get_transform_matrices


class _DepthToVertices(torch.autograd.Function):
    """depth_to_3d_grid -> rotate_pts -> translate_pts (renderer_nr.py:64-80,95-100) as one HIP pass with an
    analytic adjoint for the depth map and for the view's rotation / translation."""

    @staticmethod
    def forward(ctx, depth, inv_K, rot_mat, trans_xyz, center_z):
        d, iK, R, t = f32c(depth), f32c(inv_K), f32c(rot_mat), f32c(trans_xyz).reshape(-1, 3)
        B, H, W = d.shape
        if R.shape[0] != B or t.shape[0] != B:
            R, t = R.expand(B, 3, 3).contiguous(), t.expand(B, 3).contiguous()
        out = torch.empty(B, H * W, 3, dtype=torch.float32, device=d.device)
        _lib.check(_lib.lib().d3m_depth_to_vertices(_lib.ptr(d), _lib.ptr(iK), iK.shape[0], _lib.ptr(R), _lib.ptr(t),
                                                    float(center_z), _lib.ptr(out), B, H, W, _lib.stream_ptr()),
                   "d3m_depth_to_vertices")
        ctx.save_for_backward(d, iK, R)
        ctx.center_z = float(center_z)
        ctx.shapes = (tuple(rot_mat.shape), tuple(trans_xyz.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        d, iK, R = ctx.saved_tensors
        B, H, W = d.shape
        g = f32c(g)
        gd = torch.empty_like(d) if ctx.needs_input_grad[0] else None
        gR = torch.empty(B, 3, 3, dtype=torch.float32, device=d.device) if ctx.needs_input_grad[2] else None
        gt = torch.empty(B, 3, dtype=torch.float32, device=d.device) if ctx.needs_input_grad[3] else None
        _lib.check(_lib.lib().d3m_depth_to_vertices_backward(
            _lib.ptr(d), _lib.ptr(iK), iK.shape[0], _lib.ptr(R), ctx.center_z, _lib.ptr(g), _lib.ptr(gd), _lib.ptr(gR),
            _lib.ptr(gt), B, H, W, _lib.stream_ptr()), "d3m_depth_to_vertices_backward")
        rot_shape, trans_shape = ctx.shapes
        if gR is not None and rot_shape[0] != B:
            gR = gR.sum(0, keepdim=True)
        if gt is not None:
            gt = (gt.sum(0, keepdim=True) if trans_shape[0] != B else gt).reshape(trans_shape)
        return gd, None, gR, gt, None

EPS = 1e-7


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

        # camera intrinsics: d * K^-1 (u, v, 1)^T = (x, y, z)^T  (renderer_nr.py:24-46)
        R = torch.eye(3, dtype=torch.float32)[None].cuda()
        t = torch.zeros(1, 3, dtype=torch.float32).cuda()
        fx = fy = (self.image_size - 1) / 2 / (math.tan(self.fov / 2 * math.pi / 180))
        cx = cy = (self.image_size - 1) / 2
        K = torch.tensor([[fx, 0., cx], [0., fy, cy], [0., 0., 1.]], dtype=torch.float32).cuda()
        self.inv_K_origin = torch.inverse(K).unsqueeze(0)
        self.K_origin = K.unsqueeze(0)
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
            self.inv_K = torch.inverse(self.K[0]).unsqueeze(0)

    def set_transform_matrices(self, view):
        self.rot_mat, self.trans_xyz = get_transform_matrices(view)

    def rotate_pts(self, pts, rot_mat):
        centroid = const_tensor([0., 0., self.rot_center_depth], pts.device, (1, 1, 3))
        return (pts - centroid).matmul(rot_mat.transpose(2, 1)) + centroid

    def translate_pts(self, pts, trans_xyz):
        return pts + trans_xyz

    def depth_to_3d_grid(self, depth):
        b, h, w = depth.shape
        grid_2d = get_grid(b, h, w, normalize=False, device=depth.device)  # Nxhxwx2
        depth = depth.unsqueeze(-1)
        grid_3d = torch.cat((grid_2d, torch.ones_like(depth)), dim=3)
        return grid_3d.matmul(self.inv_K.to(depth.device).transpose(2, 1)) * depth

    def grid_3d_to_2d(self, grid_3d):
        b, h, w, _ = grid_3d.shape
        grid_2d = grid_3d / grid_3d[..., 2:]
        grid_2d = grid_2d.matmul(self.K.to(grid_3d.device).transpose(2, 1))[:, :, :, :2]
        WH = const_tensor([w - 1, h - 1], grid_3d.device, (1, 1, 1, 2))
        return grid_2d / WH * 2. - 1.  # normalize to -1~1

    def get_warped_3d_grid(self, depth):
        b, h, w = depth.shape
        # depth_to_3d_grid -> rotate_pts -> translate_pts, fused (one kernel each way)
        grid_3d = _DepthToVertices.apply(depth, self.inv_K.to(depth.device), self.rot_mat, self.trans_xyz,
                                         self.rot_center_depth)
        return grid_3d.reshape(b, h, w, 3)

    def get_inv_warped_3d_grid(self, depth):
        b, h, w = depth.shape
        grid_3d = self.depth_to_3d_grid(depth).reshape(b, -1, 3)
        grid_3d = self.translate_pts(grid_3d, -self.trans_xyz)
        grid_3d = self.rotate_pts(grid_3d, self.rot_mat.transpose(2, 1))
        return grid_3d.reshape(b, h, w, 3)

    def get_warped_2d_grid(self, depth):
        return self.grid_3d_to_2d(self.get_warped_3d_grid(depth))

    def get_inv_warped_2d_grid(self, depth):
        return self.grid_3d_to_2d(self.get_inv_warped_3d_grid(depth))

    def warp_canon_depth(self, canon_depth):
        b, h, w = canon_depth.shape
        grid_3d = self.get_warped_3d_grid(canon_depth).reshape(b, -1, 3)
        faces = get_face_idx(b, h, w, canon_depth.device)
        warped_depth = self.renderer.render_depth(grid_3d, faces)
        # allow some margin out of valid range
        margin = (self.max_depth - self.min_depth) / 2
        return warped_depth.clamp(min=self.min_depth - margin, max=self.max_depth + margin)

    def get_normal_from_depth(self, depth):
        b, h, w = depth.shape
        grid_3d = self.depth_to_3d_grid(depth)
        tu = grid_3d[:, 1:-1, 2:] - grid_3d[:, 1:-1, :-2]
        tv = grid_3d[:, 2:, 1:-1] - grid_3d[:, :-2, 1:-1]
        normal = tu.cross(tv, dim=3)
        zero = const_tensor([0., 0., 1.], depth.device)
        normal = torch.cat([zero.repeat(b, h - 2, 1, 1), normal, zero.repeat(b, h - 2, 1, 1)], 2)
        normal = torch.cat([zero.repeat(b, 1, w, 1), normal, zero.repeat(b, 1, w, 1)], 1)
        return normal / (((normal ** 2).sum(3, keepdim=True)) ** 0.5 + EPS)

    # ---- view synthesis helpers (renderer_nr.py:141-277) ------------------------------------------------
    def _render_textured(self, im, grid_3d, b, h, w):
        faces = get_face_idx(b, h, w, im.device)
        textures = get_textures_from_im(im, tx_size=self.tex_cube_size)
        return self.renderer.render_rgb(grid_3d, faces, textures).clamp(min=-1., max=1.)

    def _warp_by_grid_sample(self, im, depth, view):
        self.set_transform_matrices(view)
        recon_depth = self.warp_canon_depth(depth)
        grid_2d_from_canon = self.get_inv_warped_2d_grid(recon_depth)
        return nn.functional.grid_sample(im, grid_2d_from_canon, mode='bilinear'), grid_2d_from_canon

    def render_yaw(self, im, depth, v_before=None, v_after=None, rotations=None, maxr=90, nsample=9,
                   grid_sample=False, crop_mesh=None):
        b, c, h, w = im.shape
        grid_3d = self.depth_to_3d_grid(depth)
        if crop_mesh is not None:
            top, bottom, left, right = crop_mesh  # pixels from border to be cropped
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
        if v_before is not None:        # inverse warp
            rot_mat, trans_xyz = get_transform_matrices(v_before)
            grid_3d = self.rotate_pts(self.translate_pts(grid_3d, -trans_xyz), rot_mat.transpose(2, 1))
        if rotations is None:
            rotations = torch.linspace(-math.pi / 180 * maxr, math.pi / 180 * maxr, nsample)
        im_trans = []
        for i, ri in enumerate(rotations):
            if grid_sample:
                view = torch.tensor([0, float(ri), 0, 0, 0, 0], dtype=torch.float32, device=im.device).view(1, 6)
                if v_before is not None:
                    view = view - v_before
                warped_images, _ = self._warp_by_grid_sample(im, depth, view)
            else:
                rvec = torch.tensor([0, float(ri), 0], dtype=torch.float32, device=im.device).view(1, 3)
                rot_mat_i, _ = get_transform_matrices(rvec)
                grid_3d_i = self.rotate_pts(grid_3d, rot_mat_i.repeat(b, 1, 1))
                if v_after is not None:
                    v_after_i = v_after[i] if len(v_after.shape) == 3 else v_after
                    rot_mat, trans_xyz = get_transform_matrices(v_after_i)
                    grid_3d_i = self.translate_pts(self.rotate_pts(grid_3d_i, rot_mat), trans_xyz)
                warped_images = self._render_textured(im, grid_3d_i, b, h, w)
            im_trans += [warped_images]
        return torch.stack(im_trans, 1)  # b x t x c x h x w

    def render_view(self, im, depth, v_before=None, rotations=None, maxr=[20, 90], nsample=[5, 9], grid_sample=False):
        b, c, h, w = im.shape
        grid_3d = self.depth_to_3d_grid(depth).reshape(b, -1, 3)
        if v_before is not None:
            rot_mat, trans_xyz = get_transform_matrices(v_before)
            grid_3d = self.rotate_pts(self.translate_pts(grid_3d, -trans_xyz), rot_mat.transpose(2, 1))
        rotations_p = torch.linspace(-math.pi / 180 * maxr[0], math.pi / 180 * maxr[0], nsample[0])
        rotations_y = torch.linspace(-math.pi / 180 * maxr[1], math.pi / 180 * maxr[1], nsample[1])
        im_trans = []
        for axis, angles in ((1, rotations_y), (0, rotations_p)):     # yaw sweep first, then pitch
            for ang in angles:
                rvec = [0., 0., 0.]
                rvec[axis] = float(ang)
                if grid_sample:
                    view = torch.tensor(rvec + [0, 0, 0], dtype=torch.float32, device=im.device).view(1, 6)
                    if v_before is not None:
                        view = view - v_before
                    warped_images, _ = self._warp_by_grid_sample(im, depth, view)
                else:
                    rot_mat_i, _ = get_transform_matrices(torch.tensor(rvec, dtype=torch.float32, device=im.device).view(1, 3))
                    warped_images = self._render_textured(im, self.rotate_pts(grid_3d, rot_mat_i.repeat(b, 1, 1)), b, h, w)
                im_trans += [warped_images]
        return torch.stack(im_trans, 1)  # b x t x c x h x w

    def render_given_view(self, im, depth, view, mask=None, grid_sample=True):
        b, c, h, w = im.shape
        grid_3d = self.depth_to_3d_grid(depth).reshape(b, -1, 3)
        if grid_sample:
            warped_images, grid_2d_from_canon = self._warp_by_grid_sample(im, depth, view)
            if mask is not None:
                warped_mask = nn.functional.grid_sample(mask, grid_2d_from_canon, mode='nearest')
                return warped_images, warped_mask
        else:
            rot_mat, trans_xyz = get_transform_matrices(view)
            grid_3d = self.translate_pts(self.rotate_pts(grid_3d, rot_mat), trans_xyz)
            warped_images = self._render_textured(im, grid_3d, b, h, w)
            if mask is not None:
                warped_mask = self._render_textured(mask, grid_3d, b, h, w)
                return warped_images, warped_mask
        return warped_images  # b x c x h x w
