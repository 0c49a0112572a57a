"""Helpers of deep3dmap's NrRenderer (deep3dmap/core/renderer/utils.py:22-107): pixel grids, Euler
rotations, view vectors -> (R, t), the implicit grid topology and image -> texture cubes.  Host-side
index/constant construction and small tensor algebra; the heavy lifting happens in the renderer."""
import torch

_face_idx_cache = {}


_grid_cache = {}


def get_grid(b, H, W, normalize=True, device=None):
    """[b,H,W,2] grid in (x, y) order (utils.py:22-31).  With `device`, the single-batch grid is built once
    per (H, W, normalize, device) and expanded -- the reference rebuilds it on the CPU and uploads it on every
    call (renderer_nr.py:76)."""
    key = (H, W, bool(normalize), str(device))
    base = _grid_cache.get(key)
    if base is None:
        if normalize:
            h_range, w_range = torch.linspace(-1, 1, H), torch.linspace(-1, 1, W)
        else:
            h_range, w_range = torch.arange(0, H), torch.arange(0, W)
        gy, gx = torch.meshgrid(h_range, w_range, indexing="ij")
        base = torch.stack((gx, gy), -1).float()
        if device is not None:
            base = base.to(device)
        _grid_cache[key] = base
    return base[None].repeat(b, 1, 1, 1)


def get_rotation_matrix(tx, ty, tz):
    """Rz @ Ry @ Rx for batched Euler angles (utils.py:34-51)."""
    n, dev = len(tx), tx.device
    one, zero = torch.ones(n, device=dev), torch.zeros(n, device=dev)
    cx, sx, cy, sy, cz, sz = tx.cos(), tx.sin(), ty.cos(), ty.sin(), tz.cos(), tz.sin()
    m_x = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], 1).reshape(n, 3, 3)
    m_y = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], 1).reshape(n, 3, 3)
    m_z = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], 1).reshape(n, 3, 3)
    return torch.matmul(m_z, torch.matmul(m_y, m_x))


class _ViewTransform(torch.autograd.Function):
    """get_transform_matrices on the device: one HIP launch each way (d3m_view_transform / _backward)."""

    @staticmethod
    def forward(ctx, view):
        from .. import _lib
        v = view.detach().to(torch.float32).contiguous()
        b, n = v.shape
        rot = torch.empty(b, 3, 3, dtype=torch.float32, device=v.device)
        trans = torch.empty(b, 1, 3, dtype=torch.float32, device=v.device)
        _lib.check(_lib.lib().d3m_view_transform(_lib.ptr(v), n, _lib.ptr(rot), _lib.ptr(trans), b, _lib.stream_ptr()),
                   "d3m_view_transform")
        ctx.save_for_backward(v)
        return rot, trans

    @staticmethod
    def backward(ctx, g_rot, g_trans):
        from .. import _lib
        (v,) = ctx.saved_tensors
        b, n = v.shape
        gr = g_rot.to(torch.float32).contiguous() if g_rot is not None else None
        gt = g_trans.to(torch.float32).contiguous() if g_trans is not None else None
        g_view = torch.empty_like(v)
        _lib.check(_lib.lib().d3m_view_transform_backward(_lib.ptr(v), n, _lib.ptr(gr), _lib.ptr(gt), _lib.ptr(g_view), b,
                                                          _lib.stream_ptr()), "d3m_view_transform_backward")
        return g_view


def get_transform_matrices(view):
    """view [b,6|5|3] = (rx, ry, rz[, tx, ty[, tz]]) -> (rot [b,3,3], trans [b,1,3]) (utils.py:54-71).  A view on the
    GPU goes through one HIP launch (and one in backward); host tensors through the reference's tensor algebra."""
    if view.dim() == 2 and view.size(1) not in (3, 5, 6):
        raise ValueError("view must have 3, 5 or 6 components")
    if view.is_cuda and view.dim() == 2:
        return _ViewTransform.apply(view)
    b = view.size(0)
    if view.size(1) == 6:
        trans_xyz = view[:, 3:].reshape(b, 1, 3)
    elif view.size(1) == 5:
        trans_xyz = torch.cat([view[:, 3:].reshape(b, 1, 2), torch.zeros(b, 1, 1).to(view.device)], 2)
    elif view.size(1) == 3:
        trans_xyz = torch.zeros(b, 1, 3).to(view.device)
    else:
        raise ValueError("view must have 3, 5 or 6 components")
    return get_rotation_matrix(view[:, 0], view[:, 1], view[:, 2]), trans_xyz


def get_face_idx(b, h, w, device=None):
    """int32 [b, 2(h-1)(w-1), 3]: two triangles per grid cell, (tl,bl,tr) then (tr,bl,br) (utils.py:74-78).
    With `device`, the single-batch topology is built once per (h, w, device) and expanded -- the reference
    rebuilds it on the CPU and uploads it on every call (renderer_nr.py:119)."""
    key = (h, w, str(device))
    base = _face_idx_cache.get(key)
    if base is None:
        idx_map = torch.arange(h * w).reshape(h, w)
        faces1 = torch.stack([idx_map[:h - 1, :w - 1], idx_map[1:, :w - 1], idx_map[:h - 1, 1:]], -1).reshape(-1, 3)
        faces2 = torch.stack([idx_map[:h - 1, 1:], idx_map[1:, :w - 1], idx_map[1:, 1:]], -1).reshape(-1, 3)
        base = torch.cat([faces1, faces2], 0).int()
        if device is not None:
            base = base.to(device)
        _face_idx_cache[key] = base
    return base[None].repeat(b, 1, 1)


_CUBE = [[0.5, 0.5, 0.5], [0., 0., 1.], [0., 1., 0.], [-0.5, 0.5, 0.5],
         [1., 0., 0.], [0.5, -0.5, 0.5], [0.5, 0.5, -0.5], [0., 0., 0.]]


def vcolor_to_texture_cube(vcolors):
    """[b,c,n,3] vertex colours -> [b,n,2,2,2,c] texture cubes (utils.py:81-94)."""
    b, c, n, f = vcolors.shape
    coeffs = torch.tensor(_CUBE, dtype=torch.float32, device=vcolors.device)
    return coeffs.matmul(vcolors.permute(0, 2, 3, 1)).reshape(b, n, 2, 2, 2, c)


class _TexturesFromIm(torch.autograd.Function):
    """d3m_textures_from_im / _backward: one pass each way instead of the reference's slices, stacks and matmul."""

    @staticmethod
    def forward(ctx, im, tx_size):
        from .. import _lib
        x = im.detach().to(torch.float32).contiguous()
        b, c, h, w = x.shape
        tex = torch.empty(b, 2 * (h - 1) * (w - 1), tx_size, tx_size, tx_size, c, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().d3m_textures_from_im(_lib.ptr(x), _lib.ptr(tex), b, c, h, w, tx_size, _lib.stream_ptr()),
                   "d3m_textures_from_im")
        ctx.dims = (b, c, h, w, tx_size)
        return tex

    @staticmethod
    def backward(ctx, g):
        from .. import _lib
        b, c, h, w, tx_size = ctx.dims
        g = g.to(torch.float32).contiguous()
        g_im = torch.empty(b, c, h, w, dtype=torch.float32, device=g.device)
        _lib.check(_lib.lib().d3m_textures_from_im_backward(_lib.ptr(g), _lib.ptr(g_im), b, c, h, w, tx_size,
                                                            _lib.stream_ptr()), "d3m_textures_from_im_backward")
        return g_im, None


def get_textures_from_im(im, tx_size=1):
    """Per-face textures [b, 2(h-1)(w-1), ts, ts, ts, c] of the implicit grid mesh from an image [b,c,h,w]
    (utils.py:97-107): cell (y, x) carries the faces with vertex colours (im[y,x], im[y,x+1], im[y+1,x]) and
    (im[y+1,x], im[y,x+1], im[y+1,x+1]); tx_size 2 expands them to 2x2x2 cubes (vcolor_to_texture_cube), tx_size 1
    keeps the colour of the first / last vertex."""
    if tx_size not in (1, 2):
        raise NotImplementedError("Currently support texture size of 1 or 2 only.")
    if not im.is_cuda:
        raise RuntimeError("get_textures_from_im: im must be a CUDA tensor")
    return _TexturesFromIm.apply(im, tx_size)
