"""`Renderer`: the nn.Module deep3dmap drives (pnpmodules/neural_renderer/neural_renderer/renderer.py:11-246),
same constructor, attributes (plain, mutable, read at call time) and render_* methods."""
from __future__ import division

import math

import numpy
import torch
import torch.nn as nn

from . import cameras, mesh_ops
from .rasterize import (rasterize, rasterize_depth, rasterize_lit, rasterize_lit_fit, rasterize_lit_image_grid,
                        rasterize_mesh_modes, rasterize_rgbad, rasterize_silhouettes)


class Renderer(nn.Module):
    def __init__(self, image_size=256, anti_aliasing=True, background_color=[0, 0, 0],
                 fill_back=True, camera_mode='projection',
                 K=None, R=None, t=None, dist_coeffs=None, orig_size=1024,
                 perspective=True, viewing_angle=30, camera_direction=[0, 0, 1],
                 near=0.1, far=100,
                 light_intensity_ambient=0.5, light_intensity_directional=0.5,
                 light_color_ambient=[1, 1, 1], light_color_directional=[1, 1, 1],
                 light_direction=[0, 1, 0]):
        super(Renderer, self).__init__()
        # rendering
        self.image_size = image_size
        self.anti_aliasing = anti_aliasing
        self.background_color = background_color
        self.fill_back = fill_back

        # camera
        self.camera_mode = camera_mode
        if self.camera_mode == 'projection':
            as_dev = lambda a: torch.as_tensor(a, dtype=torch.float32).cuda() if isinstance(a, numpy.ndarray) else a
            self.K, self.R, self.t = as_dev(K), as_dev(R), as_dev(t)
            self.dist_coeffs = dist_coeffs
            if dist_coeffs is None:
                self.dist_coeffs = torch.zeros(1, 5, dtype=torch.float32).cuda()
            self.orig_size = orig_size
        elif self.camera_mode in ['look', 'look_at']:
            self.perspective = perspective
            self.viewing_angle = viewing_angle
            self.eye = [0, 0, -(1. / math.tan(math.radians(self.viewing_angle)) + 1)]
            self.camera_direction = [0, 0, 1]
        else:
            raise ValueError('Camera mode has to be one of projection, look or look_at')

        self.near = near
        self.far = far

        # light
        self.light_intensity_ambient = light_intensity_ambient
        self.light_intensity_directional = light_intensity_directional
        self.light_color_ambient = light_color_ambient
        self.light_color_directional = light_color_directional
        self.light_direction = light_direction

        # rasterization
        self.rasterizer_eps = 1e-3
        # True: fill_back and lighting are applied on the fly inside the texture sampler (no per-view copy of
        # the textures, textures / mesh of batch 1 are shared by all views).  False: the reference's sequence
        # cat -> lighting -> rasterize on materialised arrays.  Same images either way.
        self.lighting_on_the_fly = True
        # The views of a batch are independent; the on-the-fly path may run them as this many concurrent pipelines
        # (rasterize._RasterizeLit, "VIEW GROUPS").  1 = one pipeline for the whole batch.
        self.view_groups = 1
        # render_fit_loss AND render(): leave the forward's side branch (visibility list, edge-gradient plan) open for
        # backward to join.  ONLY for callers that run backward right behind forward on the same stream (MultiViewFit does).
        # With it the outputs of forward are valid only AFTER that backward pass: the value of a fused or registered
        # objective (render_fit_loss / fit_targets + multiview_fit_loss) is finished by a kernel of the backward pass.
        self.defer_plan_join = False
        # A fit objective REGISTERED with the renderer: (rgb_target [B,3,s,s], depth_target, alpha_target, mask [B,s,s]
        # [, mask_sum]) at the OUTPUT size s = image_size.  render() (lit path; with anti-aliasing too: the objective of
        # the pooled images, per-pixel records at the internal size) then evaluates it in the pass that writes the images
        # and leaves its gradient as the edge gradient's walk records; core.losses.multiview_fit_loss called on those
        # images with these very tensors returns that value and back-propagates through the records -- the reference-shaped
        # composition  loss(*renderer.render(...))  at the price of the fused objective plus the images' 20 B per pixel.
        # Any other use of the images stays correct (they are ordinary differentiable outputs).  The registration stays
        # until it is taken back (None: off); a call whose batch or image size it does not fit ignores it (_fit_hint).
        self.fit_targets = None
        self.mesh_modes = True      # render_silhouettes / render_depth of look_at cameras as one node over the indexed mesh

    def forward(self, vertices, faces, textures=None, mode=None, K=None, R=None, t=None, dist_coeffs=None,
                orig_size=None):
        '''
        mode: None -> render (rgb, depth, alpha); 'rgb'; 'silhouettes'; 'depth'  (NR/renderer.py:65-80)
        '''
        if mode is None:
            return self.render(vertices, faces, textures, K, R, t, dist_coeffs, orig_size)
        elif mode == 'rgb':
            return self.render_rgb(vertices, faces, textures, K, R, t, dist_coeffs, orig_size)
        elif mode == 'silhouettes':
            return self.render_silhouettes(vertices, faces, K, R, t, dist_coeffs, orig_size)
        elif mode == 'depth':
            return self.render_depth(vertices, faces, K, R, t, dist_coeffs, orig_size)
        else:
            raise ValueError("mode should be one of None, 'silhouettes' or 'depth'")

    # ---- the stages shared by the four render methods -------------------------------------------------
    def _transform(self, vertices, K, R, t, dist_coeffs, orig_size):
        """viewpoint transformation, NR/renderer.py:88-112; any other camera_mode leaves the vertices alone
        (the reference's tests rely on that: tests/test_rasterize_depth.py:67)."""
        if self.camera_mode == 'look_at':
            return cameras.look_at(vertices, self.eye,
                                   _perspective_angle=self.viewing_angle if self.perspective else None)
        if self.camera_mode == 'look':
            return cameras.look(vertices, self.eye, self.camera_direction,
                                _perspective_angle=self.viewing_angle if self.perspective else None)
        if self.camera_mode == 'projection':
            return cameras.projection(vertices, self.K if K is None else K, self.R if R is None else R,
                                      self.t if t is None else t,
                                      self.dist_coeffs if dist_coeffs is None else dist_coeffs,
                                      self.orig_size if orig_size is None else orig_size)
        return vertices

    def _lit_textures(self, vertices, faces, textures):
        """fill_back of the textures (NR/renderer.py:156) + lighting on world-space faces (:159-167)."""
        if self.fill_back:
            textures = torch.cat((textures, textures.permute((0, 1, 4, 3, 2, 5))), dim=1)
        faces_lighting = mesh_ops.gather_faces(vertices, faces, self.fill_back)
        return mesh_ops.lighting(faces_lighting, textures, self.light_intensity_ambient,
                                 self.light_intensity_directional, self.light_color_ambient,
                                 self.light_color_directional, self.light_direction)

    def _screen_faces(self, vertices, faces, K, R, t, dist_coeffs, orig_size):
        vertices = self._transform(vertices, K, R, t, dist_coeffs, orig_size)
        return mesh_ops.gather_faces(vertices, faces, self.fill_back)

    # ---- public render methods ------------------------------------------------------------------------
    def render_silhouettes(self, vertices, faces, K=None, R=None, t=None, dist_coeffs=None, orig_size=None):
        cam = self._camera_in_node(vertices, K, R, t, dist_coeffs, orig_size) if self.mesh_modes else None
        if cam is not None:     # the whole mode as one node over the indexed mesh (rasterize._RasterizeMeshModes)
            return rasterize_mesh_modes(vertices, faces, cam, self.fill_back, self.image_size, self.anti_aliasing, True, False)[0]
        f = self._screen_faces(vertices, faces, K, R, t, dist_coeffs, orig_size)
        # rasterizer defaults, not self.near / self.far (NR/renderer.py:114)
        return rasterize_silhouettes(f, self.image_size, self.anti_aliasing)

    def render_depth(self, vertices, faces, K=None, R=None, t=None, dist_coeffs=None, orig_size=None):
        cam = self._camera_in_node(vertices, K, R, t, dist_coeffs, orig_size) if self.mesh_modes else None
        if cam is not None:
            return rasterize_mesh_modes(vertices, faces, cam, self.fill_back, self.image_size, self.anti_aliasing, False, True)[1]
        f = self._screen_faces(vertices, faces, K, R, t, dist_coeffs, orig_size)
        return rasterize_depth(f, self.image_size, self.anti_aliasing)          # NR/renderer.py:149

    def _light_cfg(self):
        return (self.light_intensity_ambient, self.light_intensity_directional, self.light_color_ambient,
                self.light_color_directional, self.light_direction)

    def _on_the_fly(self):
        # one light for the whole batch goes through the fused sampler; per-batch colours / directions
        # (NR/lighting.py:25-30) through the materialised sequence
        return self.lighting_on_the_fly and not mesh_ops.per_batch_light(
            self.light_color_ambient, self.light_color_directional, self.light_direction)

    def _camera_in_node(self, vertices, K=None, R=None, t=None, dist_coeffs=None, orig_size=None):
        """Cameras with constant parameters run INSIDE the render node (its first launch projects, lights and clears; one
        gradient for the mesh instead of the camera's plus the light's): their parameter block, else None (a parameter that
        requires grad, or no camera_mode: _transform's torch composition / identity)."""
        if vertices.ndimension() != 3:
            return None
        if self.camera_mode == 'look_at':
            return cameras.look_at_params(vertices, self.eye, defer_basis=True,
                                          _perspective_angle=self.viewing_angle if self.perspective else None)
        if self.camera_mode == 'look':
            return cameras.look_params(vertices, self.eye, self.camera_direction, defer_basis=True,
                                       _perspective_angle=self.viewing_angle if self.perspective else None)
        if self.camera_mode == 'projection':
            return cameras.projection_params(vertices, self.K if K is None else K, self.R if R is None else R,
                                             self.t if t is None else t,
                                             self.dist_coeffs if dist_coeffs is None else dist_coeffs,
                                             self.orig_size if orig_size is None else orig_size)
        return None

    def render_rgb(self, vertices, faces, textures, K=None, R=None, t=None, dist_coeffs=None, orig_size=None):
        if self._on_the_fly():
            cam = self._camera_in_node(vertices, K, R, t, dist_coeffs, orig_size)
            sv = None if cam is not None else self._transform(vertices, K, R, t, dist_coeffs, orig_size)
            return rasterize_lit(sv, vertices, faces, textures, self._light_cfg(), self.fill_back, self.image_size,
                                 self.anti_aliasing, self.near, self.far, self.rasterizer_eps, self.background_color,
                                 False, False, view_groups=self.view_groups, camera=cam)['rgb']
        f = self._screen_faces(vertices, faces, K, R, t, dist_coeffs, orig_size)
        textures = self._lit_textures(vertices, faces, textures)
        return rasterize(f, textures, self.image_size, self.anti_aliasing, self.near, self.far,
                         self.rasterizer_eps, self.background_color)

    def render_rgb_image_grid(self, vertices, image, K=None, R=None, t=None, dist_coeffs=None, orig_size=None):
        """render_rgb of the grid mesh of a depth map (vertices [B,h*w,3], one per pixel, row-major) textured by `image`
        [B,3,h,w] with tx_size-2 cubes -- render_rgb(vertices, get_face_idx(b,h,w), get_textures_from_im(image, 2)) of
        deep3dmap's NrRenderer (renderer_nr.py:196-198) without the index and texture arrays (rasterize_lit_image_grid).
        Forward only; needs fill_back and one light for the batch."""
        if not (self._on_the_fly() and self.fill_back):
            raise ValueError("render_rgb_image_grid needs fill_back and lighting_on_the_fly (one light for the batch)")
        sv = self._transform(vertices, K, R, t, dist_coeffs, orig_size)
        return rasterize_lit_image_grid(sv, vertices, tuple(image.shape[2:]), image, self._light_cfg(), self.image_size,
                                        self.anti_aliasing, self.near, self.far, self.rasterizer_eps, self.background_color)

    def render_fit_loss(self, vertices, faces, textures, targets, K=None, R=None, t=None, dist_coeffs=None,
                        orig_size=None, images_out=None, grad_sink=None):
        """The multi-view fit objective of render()'s images against `targets` = (rgb, depth, alpha, mask), evaluated
        inside the rendering node (rasterize_lit_fit); needs lighting_on_the_fly.  `images_out`
        = (rgb [B,3,S,S], depth [B,S,S], alpha [B,S,S]) buffers: the same pass also writes the images render() would
        return (for display / logging; gradients flow through the returned objective only).  `grad_sink` = (grad_vertices
        like `vertices`, grad_textures like `textures` | None, loss [1]): buffers of the CALLER that the node writes the
        objective and its two gradients into in place (per call, never remembered; ignored unless they fit this call's
        tensors exactly and the camera runs inside the node)."""
        if not self._on_the_fly():
            raise ValueError("render_fit_loss needs lighting_on_the_fly (one light for the batch)")
        # look_at cameras with constant parameters run INSIDE the node (one gradient for the mesh instead of the camera's
        # plus the light's; results straight into the caller's grad_sink buffers when it has set them)
        cam = self._camera_in_node(vertices, K, R, t, dist_coeffs, orig_size)
        sv = None if cam is not None else self._transform(vertices, K, R, t, dist_coeffs, orig_size)
        return rasterize_lit_fit(sv, vertices, faces, textures, self._light_cfg(), self.fill_back, targets,
                                 self.image_size, self.near, self.far, self.rasterizer_eps, self.background_color,
                                 view_groups=self.view_groups, defer_plan_join=self.defer_plan_join, images_out=images_out,
                                 camera=cam, grad_sink=grad_sink if cam is not None else None,
                                 anti_aliasing=self.anti_aliasing)

    def render_fit_loss_manual(self, manual, vertices, faces, textures, targets, images_out=None, grad_sink=None):
        """render_fit_loss driven without the autograd engine: `manual` is a rasterize.LitFitManual whose forward is run
        here (look_at cameras only: the camera runs inside the node); its two backward halves are the caller's to call."""
        if not self._on_the_fly() or self.camera_mode != 'look_at':
            raise ValueError("render_fit_loss_manual needs lighting_on_the_fly and camera_mode 'look_at'")
        cam = self._camera_in_node(vertices)
        if cam is None:
            raise ValueError("render_fit_loss_manual: the camera's parameters must be constants (no requires_grad)")
        return manual.forward(vertices, faces, textures, self._light_cfg(), self.fill_back, targets, self.image_size,
                              self.near, self.far, self.rasterizer_eps, self.background_color, cam, grad_sink=grad_sink,
                              images_out=images_out, anti_aliasing=self.anti_aliasing)

    def _fit_hint(self, batch):
        """The registered objective (fit_targets) if it fits THIS call -- `batch` views at the current image_size, float32
        tensors on one device -- else None: the call is then a plain render() (the registration is sticky; a later call with
        another batch or size must not fail on it)."""
        hint = self.fit_targets
        if hint is None:
            return None
        s = int(self.image_size)
        try:
            rgb_t, depth_t, alpha_t, mask = hint[:4]
            ok = tuple(rgb_t.shape) == (batch, 3, s, s) and all(tuple(x.shape) == (batch, s, s) for x in (depth_t, alpha_t, mask)) \
                and all(x.is_cuda and x.dtype == torch.float32 for x in (rgb_t, depth_t, alpha_t, mask))
        except (TypeError, ValueError, AttributeError):
            ok = False
        return hint if ok else None

    def render(self, vertices, faces, textures, K=None, R=None, t=None, dist_coeffs=None, orig_size=None):
        if self._on_the_fly():
            cam = self._camera_in_node(vertices, K, R, t, dist_coeffs, orig_size)
            sv = None if cam is not None else self._transform(vertices, K, R, t, dist_coeffs, orig_size)
            out = rasterize_lit(sv, vertices, faces, textures, self._light_cfg(), self.fill_back, self.image_size,
                                self.anti_aliasing, self.near, self.far, self.rasterizer_eps, self.background_color,
                                view_groups=self.view_groups, camera=cam,
                                fit_hint=self._fit_hint(cam["batch"] if cam is not None else sv.shape[0]),
                                defer_plan_join=self.defer_plan_join)
        else:
            f = self._screen_faces(vertices, faces, K, R, t, dist_coeffs, orig_size)
            textures = self._lit_textures(vertices, faces, textures)
            out = rasterize_rgbad(f, textures, self.image_size, self.anti_aliasing, self.near, self.far,
                                  self.rasterizer_eps, self.background_color)
        return out['rgb'], out['depth'], out['alpha']
