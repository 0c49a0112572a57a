"""The renderer block of the gan2shape training step (deep3dmap/models/frameworks/gan2shape.py:463-497, "G2S") as ONE
autograd node on the fused HIP passes of csrc/d3m_g2s.h (d3m_g2s_forward: 4 launches, d3m_g2s_backward: 5) instead of
the ~300 eager kernels the reference spends on it per step:

    normal          = renderer.get_normal_from_depth(depth)                                          G2S:463
    diffuse_shading = (normal * light_d.view(-1,1,1,3)).sum(3).clamp(min=0).unsqueeze(1)             G2S:464
    texture         = (albedo/2+0.5) * (light_a + light_b*diffuse_shading) * 2 - 1                   G2S:465-466
    recon_depth     = renderer.warp_canon_depth(depth)                                               G2S:468
    grid            = renderer.get_inv_warped_2d_grid(recon_depth)                                   G2S:475
    recon_im_mask   = (recon_depth < max_depth + margin) [both halves multiplied with flip]          G2S:476-482
    recon_im        = grid_sample(texture, grid, mode='bilinear').clamp(min=-1, max=1)               G2S:483
    loss_l1_im(_flip) = photometric_loss(recon_im[:b] / [b:], input_im, mask=recon_im_mask[:b] / [b:])   G2S:486,489
    loss_smooth     = smooth_loss(depth) + smooth_loss(diffuse_shading)                              G2S:493-494

`NrRenderer.reconstruct()` is the entry point; the step-by-step methods of NrRenderer stay as they are."""
import ctypes
from types import SimpleNamespace

import torch

from .. import _lib
from ..neural_renderer import cameras
from ..neural_renderer._util import f32c
from ..neural_renderer.rasterize import DEFAULT_FAR, DEFAULT_NEAR


def _batch(t, B, tail):
    """[1|B, *tail] -> contiguous [B, *tail]"""
    t = f32c(t).reshape((-1,) + tail)
    return t if t.shape[0] == B else t.expand((B,) + tail).contiguous()


class _ReconBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, albedo, light_a, light_b, light_d, view, rot, trans, target, extra_mask, cfg):
        L = _lib.lib()
        ctx.set_materialize_grads(False)
        d, alb = f32c(depth), f32c(albedo)
        B, H, W = d.shape
        dev = d.device
        _lib.require_device(d, alb, names=["depth", "albedo"])
        if tuple(alb.shape) != (B, 3, H, W):
            raise ValueError("albedo must be [B,3,H,W] for depth [B,H,W]")
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        la, lb, ld = _batch(light_a, B, ()), _batch(light_b, B, ()), _batch(light_d, B, (3,))
        if view is not None:        # the block turns the view vectors into (R, t) itself (and back, in backward)
            vw = _batch(view, B, (view.shape[-1],))
            R, t = e(B, 3, 3), e(B, 3)
        else:
            vw = None
            R, t = _batch(rot, B, (3, 3)), _batch(trans, B, (3,))
        s, aa, flip = int(cfg["image_size"]), bool(cfg["anti_aliasing"]), bool(cfg["flip"])
        S = 2 * s if aa else s
        if flip and B % 2:
            raise ValueError("flip needs an even batch: entries (b, b + B/2) are an image and its mirror")
        b_half = B // 2 if flip else B
        tgt = f32c(target) if target is not None else None
        if tgt is not None and tuple(tgt.shape) != (b_half, 3, s, s):
            raise ValueError(f"input_im must be [{b_half},3,{s},{s}]")
        xm = None
        if extra_mask is not None:
            xm = f32c(extra_mask).reshape(-1, s, s)
            xm = xm if xm.shape[0] == B else xm.expand(B, s, s).contiguous()
        need_grad = any(ctx.needs_input_grad[:8])
        out = SimpleNamespace(normal=e(B, H, W, 3), diffuse=e(B, H, W), texture=e(B, 3, H, W), recon_depth=e(B, s, s),
                              recon_im=e(B, 3, s, s), mask=e(B, s, s), losses=e(4))
        keep = SimpleNamespace(sv=e(B, H * W, 3), zbuf=torch.empty((B, S, S), dtype=torch.int64, device=dev),
                               scratch=e(int(L.d3m_g2s_scratch_floats(B, H, W, s))),
                               g_tex=e(B, 3, H, W) if need_grad else None)
        cam, cam_keep = cameras._camera_struct(cfg["camera"], dev)
        blk = _lib.D3MG2SBlock()
        blk.batch_size, blk.height, blk.width, blk.image_size = B, H, W, s
        blk.anti_aliasing, blk.flip = int(aa), int(flip)
        iK, K = cfg["inv_K"], cfg["K"]
        blk.inv_K, blk.inv_K_batch, blk.K, blk.K_batch = iK.data_ptr(), iK.shape[0], K.data_ptr(), K.shape[0]
        blk.rot_center_depth, blk.depth_min, blk.depth_max = cfg["rot_center_depth"], cfg["depth_min"], cfg["depth_max"]
        blk.near, blk.far = cfg["near"], cfg["far"]
        blk.camera = ctypes.pointer(cam)
        blk.view_components = vw.shape[1] if vw is not None else 0
        for name, tensor in (("view", vw), ("rot", R), ("trans", t), ("depth", d), ("albedo", alb), ("light_a", la),
                             ("light_b", lb), ("light_d", ld), ("target", tgt), ("extra_mask", xm), ("normal", out.normal),
                             ("diffuse_shading", out.diffuse), ("texture", out.texture), ("recon_depth", out.recon_depth),
                             ("recon_im", out.recon_im), ("recon_im_mask", out.mask), ("losses", out.losses),
                             ("screen_vertices", keep.sv), ("zbuffer", keep.zbuf), ("scratch", keep.scratch),
                             ("grad_texture", keep.g_tex)):
            setattr(blk, name, tensor.data_ptr() if tensor is not None else None)
        lam = cfg.get("lam_smooth")
        blk.with_smooth, blk.lam_smooth = int(bool(cfg["with_smooth"])), float(lam) if lam is not None else 0.0
        _lib.check(L.d3m_g2s_forward(ctypes.byref(blk), _lib.stream_ptr()), "d3m_g2s_forward")
        ctx.blk, ctx.cam = blk, cam                 # the filled struct is reused by backward (pointers stay valid below)
        ctx.alive = (d, alb, la, lb, ld, vw, R, t, tgt, xm, out, keep, iK, K, cam_keep)
        ctx.shapes = tuple(tuple(x.shape) if x is not None else None for x in (light_a, light_b, light_d, view, rot, trans))
        ctx.mark_non_differentiable(out.normal, out.diffuse, out.texture, out.recon_depth, out.mask, R, t)
        return (out.recon_im, out.losses[0], out.losses[1], out.losses[2], out.losses[3], out.normal, out.diffuse,
                out.texture, out.recon_depth, out.mask, R, t)

    @staticmethod
    def backward(ctx, g_im, g_l1, g_l1f, g_sm, g_total, *_unused):
        L = _lib.lib()
        blk = ctx.blk
        d, alb, la, lb, ld, vw, R, t, tgt, xm, out, keep, iK, K, cam_keep = ctx.alive
        if keep.g_tex is None:
            raise RuntimeError("reconstruct(): backward needs inputs that required grad when forward ran")
        B, H, W = d.shape
        dev = d.device
        S = keep.zbuf.shape[1]
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        grads_in = [f32c(g) if g is not None else None for g in (g_im, g_l1, g_l1f, g_sm, g_total)]
        for name, g in zip(("grad_recon_im", "grad_l1", "grad_l1_flip", "grad_smooth", "grad_total"), grads_in):
            setattr(blk, name, g.data_ptr() if g is not None else None)
        bufs = dict(grad_tri=e(B, 2 * (H - 1) * (W - 1), 3, 3), grad_depth_map=e(B, S, S), grad_normal=e(B, H, W, 3),
                    grad_depth_mesh=e(B, H, W), grad_depth=e(B, H, W), grad_albedo=e(B, 3, H, W), grad_light_a=e(B),
                    grad_light_b=e(B), grad_light_d=e(B, 3), grad_rot=e(B, 3, 3), grad_trans=e(B, 3))
        if vw is not None:
            bufs["grad_view"] = e(*vw.shape)
        for name, tensor in bufs.items():
            setattr(blk, name, tensor.data_ptr())
        _lib.check(L.d3m_g2s_backward(ctypes.byref(blk), _lib.stream_ptr()), "d3m_g2s_backward")
        sh_a, sh_b, sh_d, sh_v, sh_R, sh_t = ctx.shapes

        def back(g, shape, per):           # undo _batch(): sum a broadcast batch, restore the caller's shape
            n = 1
            for v in shape:
                n *= v
            return (g if n == g.numel() else g.reshape((B,) + per).sum(0)).reshape(shape)

        need = ctx.needs_input_grad
        return (bufs["grad_depth"] if need[0] else None, bufs["grad_albedo"] if need[1] else None,
                back(bufs["grad_light_a"], sh_a, ()) if need[2] else None,
                back(bufs["grad_light_b"], sh_b, ()) if need[3] else None,
                back(bufs["grad_light_d"], sh_d, (3,)) if need[4] else None,
                back(bufs["grad_view"], sh_v, (sh_v[-1],)) if (need[5] and vw is not None) else None,
                back(bufs["grad_rot"], sh_R, (3, 3)) if (need[6] and vw is None) else None,
                back(bufs["grad_trans"], sh_t, (3,)) if (need[7] and vw is None) else None, None, None, None)


def reconstruct(renderer, depth, albedo, light_a, light_b, light_d, input_im=None, flip=False, extra_mask=None,
                with_smooth=True, lam_smooth=None):
    """See NrRenderer.reconstruct."""
    r = renderer.renderer
    if r.camera_mode != 'projection' or not r.fill_back:
        raise ValueError("reconstruct() needs the NrRenderer configuration (projection camera, fill_back)")
    dev = depth.device
    margin = (renderer.max_depth - renderer.min_depth) / 2
    cam = dict(mode=_lib.CAMERA_PROJECTION, batch=depth.shape[0],
               rot=cameras.as_device_f32(r.R, dev).reshape(-1, 3, 3), eye_or_t=cameras.as_device_f32(r.t, dev).reshape(-1, 3),
               K=cameras.as_device_f32(r.K, dev).reshape(-1, 3, 3), dist=cameras.as_device_f32(r.dist_coeffs, dev).reshape(-1, 5),
               orig_size=float(r.orig_size))
    cfg = dict(image_size=r.image_size, anti_aliasing=r.anti_aliasing, flip=flip, camera=cam,
               inv_K=f32c(renderer.inv_K.to(dev)), K=f32c(renderer.K.to(dev)), rot_center_depth=float(renderer.rot_center_depth),
               depth_min=float(renderer.min_depth - margin), depth_max=float(renderer.max_depth + margin),
               # render_depth rasterizes with the rasterizer's defaults, not the renderer's near / far (NR/renderer.py:149)
               near=float(DEFAULT_NEAR), far=float(DEFAULT_FAR), with_smooth=with_smooth, lam_smooth=lam_smooth)
    # a view set through set_transform_matrices(view) and not looked at since goes into the block as it is: (R, t) are
    # then computed by the block's first pass, and the view's gradient by its last
    view = renderer._pending_view()
    rot, trans = (None, None) if view is not None else (renderer.rot_mat, renderer.trans_xyz)
    (recon_im, l1, l1f, sm, total, normal, diffuse, texture, recon_depth, mask, R, t) = _ReconBlock.apply(
        depth, albedo, light_a, light_b, light_d, view, rot, trans, input_im, extra_mask, cfg)
    if view is not None:
        renderer._resolve_view(view, R, t)
    return SimpleNamespace(normal=normal, diffuse_shading=diffuse[:, None], texture=texture, recon_depth=recon_depth,
                           recon_im=recon_im, recon_im_mask=mask[:, None], loss_l1_im=l1,
                           loss_l1_im_flip=l1f if flip else None, loss_smooth=sm if with_smooth else None,
                           loss_total=total if lam_smooth is not None else None)
