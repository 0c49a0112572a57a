"""Losses evaluated on renderer outputs (deep3dmap/core/utils/utils.py:82-114 and the silhouette loss of
pnpmodules/neural_renderer/examples/example2.py:43-47), as HIP reduction kernels with fused gradients."""
import torch

from .. import _lib
from ..neural_renderer._util import f32c

EPS = 1e-7


class _Photometric(torch.autograd.Function):
    @staticmethod
    def forward(ctx, im1, im2, mask, conf_sigma):
        # the reference's loss is plain tensor algebra: gradients would also reach im2, mask and conf_sigma.  This
        # kernel differentiates with respect to im1 only -- say so instead of silently dropping a gradient
        for name, t, need in (("im2", im2, ctx.needs_input_grad[1]), ("mask", mask, ctx.needs_input_grad[2]),
                              ("conf_sigma", conf_sigma, ctx.needs_input_grad[3])):
            if t is not None and need:
                raise NotImplementedError(f"photometric_loss: no gradient with respect to {name} (detach it, or compose "
                                          "the loss from tensor ops)")
        a, b = f32c(im1), f32c(im2)
        B, C, H, W = a.shape
        m = f32c(mask) if mask is not None else None
        s = f32c(conf_sigma) if conf_sigma is not None else None
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        need = ctx.needs_input_grad[0]
        grad = torch.empty_like(a) if need else None
        scratch = torch.empty(2048, dtype=torch.float32, device=a.device)
        rc = _lib.lib().d3m_photometric_loss(_lib.ptr(a), _lib.ptr(b), _lib.ptr(m), _lib.ptr(s), _lib.ptr(loss),
                                             _lib.ptr(grad), _lib.ptr(scratch), B, C, H, W, _lib.stream_ptr())
        _lib.check(rc, "d3m_photometric_loss")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g if grad is not None else None), None, None, None


def photometric_loss(im1, im2, mask=None, conf_sigma=None):
    """Masked mean L1, optionally the Laplacian NLL with per-pixel sigma (utils.py:105-114).
    im1/im2 [B,C,H,W]; mask / conf_sigma [B,1,H,W] (broadcast over channels).  Gradient flows to im1."""
    if im1.dim() == 3:
        im1, im2 = im1[:, None], im2[:, None]
    if mask is not None and mask.dim() == 3:
        mask = mask[:, None]
    if mask is not None and mask.shape[1] != 1:
        raise NotImplementedError("photometric_loss: mask must have one channel")
    return _Photometric.apply(im1, im2, mask, conf_sigma)


class _SumSquaredError(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("silhouette_loss: no gradient with respect to the reference image (detach it)")
        a, b = f32c(a), f32c(b)
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        grad = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        scratch = torch.empty(1024, dtype=torch.float32, device=a.device)
        rc = _lib.lib().d3m_sum_squared_error(_lib.ptr(a), _lib.ptr(b), _lib.ptr(loss), _lib.ptr(grad), _lib.ptr(scratch),
                                              a.numel(), _lib.stream_ptr())
        _lib.check(rc, "d3m_sum_squared_error")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g if grad is not None else None), None


def silhouette_loss(image, image_ref):
    """sum((image - image_ref)^2) (examples/example2.py:46)."""
    return _SumSquaredError.apply(image, image_ref.expand_as(image))


class _MultiViewFitLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, depth, alpha, rgb_t, depth_t, alpha_t, mask, mask_sum=None):
        if any(ctx.needs_input_grad[3:]):
            raise NotImplementedError("multiview_fit_loss: targets, mask and mask_sum are constants (detach them)")
        t = [f32c(x) for x in (rgb, rgb_t, depth, depth_t, alpha, alpha_t, mask)]
        mask_sum = f32c(mask_sum).reshape(1) if mask_sum is not None else None
        B, C, H, W = t[0].shape
        if C != 3 or any(tuple(x.shape) != (B, H, W) for x in t[2:]) or tuple(t[1].shape) != (B, 3, H, W):
            raise ValueError("multiview_fit_loss: rgb [B,3,H,W]; depth, alpha, mask and their targets [B,H,W]")
        loss = torch.empty((), dtype=torch.float32, device=t[0].device)
        scratch = torch.empty(4104, dtype=torch.float32, device=t[0].device)
        _lib.check(_lib.lib().d3m_fit_loss_forward(*[_lib.ptr(x) for x in t], _lib.ptr(loss), _lib.ptr(scratch),
                                                   _lib.ptr(mask_sum), B, H, W, _lib.stream_ptr()),
                   "d3m_fit_loss_forward")
        ctx.save_for_backward(*t, scratch)
        return loss

    @staticmethod
    def backward(ctx, g):
        *t, scratch = ctx.saved_tensors
        B, _, H, W = t[0].shape
        need = ctx.needs_input_grad
        g_rgb = torch.empty_like(t[0]) if need[0] else None
        g_depth = torch.empty_like(t[2]) if need[1] else None
        g_alpha = torch.empty_like(t[4]) if need[2] else None
        _lib.check(_lib.lib().d3m_fit_loss_backward(*[_lib.ptr(x) for x in t], _lib.ptr(scratch), _lib.ptr(f32c(g)),
                                                    _lib.ptr(g_rgb), _lib.ptr(g_depth), _lib.ptr(g_alpha), B, H, W,
                                                    _lib.stream_ptr()), "d3m_fit_loss_backward")
        return g_rgb, g_depth, g_alpha, None, None, None, None, None


class _FitLossOnLitImages(torch.autograd.Function):
    """multiview_fit_loss on the three images of ONE lit render node (Renderer.render's on-the-fly path, no anti-aliasing):
    the value in one pass over the images (d3m_fit_loss_records), and in the same pass the objective's gradient as the edge
    gradient's per-pixel records -- what Renderer.render_fit_loss leaves behind.  backward() hands the render node the
    scalar gradient through the link and returns zero-stride zero images: the render node's backward recognises them and
    takes the records route (no gradient images, no d3m_fit_loss_backward, no k_pack_maps).  Same value and gradients as
    the plain node below."""

    @staticmethod
    def forward(ctx, rgb, depth, alpha, rgb_t, depth_t, alpha_t, mask, mask_sum, lit):
        from ..neural_renderer.rasterize import LitImagesLink, _RasterizeLit
        if any(ctx.needs_input_grad[3:8]):
            raise NotImplementedError("multiview_fit_loss: targets, mask and mask_sum are constants (detach them)")
        L = _lib.lib()
        B, _, S, _ = rgb.shape
        dev = rgb.device
        rgb_t, depth_t, alpha_t, mask = (f32c(x) for x in (rgb_t, depth_t, alpha_t, mask))
        mask_sum = f32c(mask_sum).reshape(1) if mask_sum is not None else mask.sum().reshape(1)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        scratch = [torch.empty(int(L.d3m_render_fit_scratch_floats(B, S)), dtype=torch.float32, device=dev)]
        g_maps = (torch.empty(B, S, S, 4, dtype=torch.float32, device=dev),        # edge_grad
                  torch.empty(B, S, S, 2, dtype=torch.float32, device=dev),        # edge_dot
                  tuple(torch.zeros(2, B, 2, S, dtype=torch.int32, device=dev).unbind(0)),   # nz_lo_inv, nz_hi1
                  torch.empty(B, S, S, dtype=torch.float32, device=dev))           # grad_depth_map
        fit_state = (rgb_t, depth_t, alpha_t, mask, scratch, loss, g_maps, mask_sum, False)
        fit_c = _RasterizeLit._fit_struct(fit_state, 0, 0, B, None)
        import ctypes
        _lib.check(L.d3m_fit_loss_records(_lib.ptr(rgb), _lib.ptr(depth), _lib.ptr(alpha), _lib.ptr(lit.maps["face_index_map"]),
                                          ctypes.byref(fit_c), B, S, _lib.stream_ptr()), "d3m_fit_loss_records")
        link = LitImagesLink(fit_state, (tuple(rgb.shape), tuple(depth.shape), tuple(alpha.shape)),
                             torch.zeros(1, dtype=torch.float32, device=dev))
        lit.linked_fit = link
        ctx.link = link
        ctx.save_for_backward(rgb, depth, alpha)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        link = ctx.link
        link.grad_loss = f32c(g).reshape(1)
        link.images = ctx.saved_tensors         # lent until the render node's backward has run (see LitImagesLink)
        if any(_observed(t) for t in ctx.saved_tensors):
            # somebody looks at the images' gradients (retain_grad / register_hook set after the loss was built): zero-stride
            # zeros would be what they see -- hand out the objective's real gradient images instead (no records route)
            g_rgb, g_alpha, g_depth = link.add_gradient_images(None, None, None)
            link.images = None
            return g_rgb, g_depth, g_alpha, None, None, None, None, None, None
        link.pending = True
        # `pending` belongs to THIS backward pass.  If the render node is not part of it (torch.autograd.grad(loss, rgb),
        # backward(inputs=[rgb])), the dummies below are what the caller receives -- silently wrong -- and a later,
        # unrelated backward through the render node would find the stale flag and add this objective's gradient to a loss
        # that never contained it.  The engine calls this at the end of the pass: a flag still set then is an error, and is
        # cleared either way.
        torch.autograd.Variable._execution_engine.queue_callback(link.end_of_pass)
        g_rgb, g_depth, g_alpha = link.dummies()
        return g_rgb, g_depth, g_alpha, None, None, None, None, None, None


class _FitLossFromRenderNode(torch.autograd.Function):
    """multiview_fit_loss on the images of a lit render node that was given the objective's targets up front
    (Renderer.fit_targets): the pass that wrote the images has evaluated the objective and left its gradient as walk
    records already -- nothing is launched here.  Backward as _FitLossOnLitImages: the scalar through the link, zero-stride
    zero images back."""

    @staticmethod
    def forward(ctx, rgb, depth, alpha, lit):
        from ..neural_renderer.rasterize import LitImagesLink
        hs = lit.hint_state
        link = LitImagesLink(hs, (tuple(rgb.shape), tuple(depth.shape), tuple(alpha.shape)),
                             torch.zeros(1, dtype=torch.float32, device=rgb.device))
        link.finish_pending = bool(getattr(lit, "fit_flags", 0))      # the value is complete behind the render node's backward
        lit.linked_fit = link
        ctx.link = link
        ctx.save_for_backward(rgb, depth, alpha)
        return hs[5].reshape(())

    backward = staticmethod(_FitLossOnLitImages.backward)


def _same(a, b):
    return torch.is_tensor(a) and torch.is_tensor(b) and a.data_ptr() == b.data_ptr() and a.shape == b.shape and \
        a.dtype == b.dtype and a.stride() == b.stride()


def _observed(t):
    """whether a tensor's gradient is looked at from outside the graph: retain_grad() or a tensor hook"""
    return torch.is_tensor(t) and (t.retains_grad or bool(getattr(t, "_backward_hooks", None)))


def multiview_fit_loss(rgb, depth, alpha, rgb_target, depth_target, alpha_target, mask, mask_sum=None, link=True):
    """photometric_loss(rgb, rgb_target, mask) + silhouette_loss(alpha, alpha_target) / (H*W) +
    photometric_loss(depth, depth_target, mask): the multi-view fit objective as ONE autograd node (a reduction and a
    finish launch forward, one gradient launch backward) instead of three loss nodes and their eager glue.
    rgb [B,3,H,W]; depth, alpha, mask and the targets [B,H,W].  `mask_sum` (device scalar, optional) replaces sum(mask)
    as the normaliser of the two photometric terms: the mask's sum over ALL shards when these views are one rank's share
    of a camera-sharded objective (deep3dmap_amd/multiview.py).

    When rgb, depth and alpha are the three images of one Renderer.render() call (NR/renderer.py:200-246) on the lit path
    -- without anti-aliasing, or with the objective registered on the renderer (Renderer.fit_targets) -- the objective's
    gradient never exists as images: see _FitLossOnLitImages / _FitLossFromRenderNode.  That route is not
    taken when an image's gradient is observed (retain_grad, tensor hooks) or with `link=False`; asking autograd for the
    gradient with respect to the images themselves WITHOUT back-propagating through the render node
    (torch.autograd.grad(loss, rgb)) needs `link=False` and raises otherwise."""
    if link and torch.is_grad_enabled() and not any(_observed(x) for x in (rgb, depth, alpha)) and all(torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
                                       for x in (rgb, depth, alpha)):
        from ..neural_renderer.rasterize import lit_images_link
        lit = lit_images_link(rgb, depth, alpha)
        B = rgb.shape[0]
        if lit is not None and rgb.dim() == 4 and rgb.shape[1] == 3 and rgb.shape[2] == rgb.shape[3] and all(
                tuple(x.shape) == (B,) + tuple(rgb.shape[2:]) for x in (depth, alpha, depth_target, alpha_target, mask)) \
                and tuple(rgb_target.shape) == tuple(rgb.shape):
            hs = getattr(lit, "hint_state", None)
            # ... and the same NORMALISER: a registered objective that carried a (global) mask_sum is only this call's when
            # this call passes that very tensor, one that did not only when this call does not either (sum(mask) both times)
            given = getattr(lit, "hint_mask_sum_given", False)
            same_norm = (mask_sum is None and not given) or \
                (mask_sum is not None and given and hs is not None and _same(hs[7], mask_sum.reshape(1)))
            if hs is not None and _same(hs[0], rgb_target) and _same(hs[1], depth_target) and _same(hs[2], alpha_target) \
                    and _same(hs[3], mask) and same_norm:
                return _FitLossFromRenderNode.apply(rgb, depth, alpha, lit)
            if not lit.cfg[2]:      # (the objective on FINISHED images takes them at the internal size: no anti-aliasing)
                return _FitLossOnLitImages.apply(rgb, depth, alpha, rgb_target, depth_target, alpha_target, mask, mask_sum, lit)
    return _MultiViewFitLoss.apply(rgb, depth, alpha, rgb_target, depth_target, alpha_target, mask, mask_sum)


class _SmoothLevel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred):
        p = f32c(pred)
        B, H, W = p.shape
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        scratch = torch.empty(4096, dtype=torch.float32, device=p.device)
        _lib.check(_lib.lib().d3m_smooth_loss_forward(_lib.ptr(p), _lib.ptr(loss), _lib.ptr(scratch), B, H, W,
                                                      _lib.stream_ptr()), "d3m_smooth_loss_forward")
        ctx.save_for_backward(p)
        return loss

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        B, H, W = p.shape
        grad = torch.empty_like(p)
        _lib.check(_lib.lib().d3m_smooth_loss_backward(_lib.ptr(p), _lib.ptr(f32c(g)), _lib.ptr(grad), B, H, W,
                                                       _lib.stream_ptr()), "d3m_smooth_loss_backward")
        return grad


def smooth_loss(pred_map):
    """Second-order smoothness over a (pyramid of) map(s) (utils.py:82-102): sum of mean |dxx|, |dxy|, |dyx|,
    |dyy| per level, weights 1, 1/2.3, ....  Each level is one autograd node (a reduction, a finish and, backward, one
    gather kernel) instead of the reference's ~25 eager slice / abs / mean kernels forward and ~60 backward."""
    maps = pred_map if type(pred_map) in (tuple, list) else [pred_map]
    loss, weight = 0, 1.
    for scaled_map in maps:
        m = scaled_map.reshape(-1, scaled_map.size(-2), scaled_map.size(-1)) if scaled_map.dim() == 4 else scaled_map
        loss = loss + _SmoothLevel.apply(m) * weight
        weight /= 2.3
    return loss
