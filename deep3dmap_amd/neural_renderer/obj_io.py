"""Wavefront OBJ / MTL assets of the neural_renderer package surface: `load_obj`, `save_obj`, `load_textures`,
`create_texture_image`, `Mesh` (the on-disk formats and call signatures of NR/load_obj.py, NR/save_obj.py, NR/mesh.py).

Organisation (not the reference's): an .obj file is tokenised ONCE into a `_ObjScene` (positions, texture coordinates,
triangle corners as (position, texcoord) index pairs, one material id per triangle) that every loader reads; an .mtl
file into one record per material.  Writers format whole arrays (numpy) instead of looping over rows.  The two texture
kernels -- uv image -> per-face texture cubes, texture cubes -> atlas image -- are entry points of the HIP library
(`d3m_load_textures`, `d3m_create_texture_image`); the atlas's corner table is built on the device.  The image codec is
Pillow (the reference used scikit-image)."""
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from .. import _lib

texture_wrapping_dict = {'REPEAT': 0, 'MIRRORED_REPEAT': 1, 'CLAMP_TO_EDGE': 2, 'CLAMP_TO_BORDER': 3}
ATLAS_EPS = 1e-5            # the reference's create_texture_image eps


# ---- files --------------------------------------------------------------------------------------------------------
def _read_image(path):
    """[H,W,3] float32 in 0..1, row 0 = BOTTOM of the picture (texture space), alpha dropped, grey replicated."""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert('RGB'), dtype=np.float32) / 255.0
    return np.ascontiguousarray(rgb[::-1])


def _write_image(path, image):
    from PIL import Image
    if image.dtype != np.uint8:
        image = np.clip(np.rint(np.nan_to_num(image) * 255.0), 0, 255).astype(np.uint8)
    Image.fromarray(image).save(path)


class _ObjScene:
    """One pass over an .obj file.  positions [V,3] f32, texcoords [T,2] f32, tri_pos / tri_tex [F,3] zero-based int32
    (tri_tex is -1 where a corner has no texture coordinate), tri_material [F] index into `materials` (-1: none),
    mtllib: the file named by the last `mtllib` statement."""

    def __init__(self, path):
        pos, tex, corners_p, corners_t, tri_mat = [], [], [], [], []
        self.materials, self.mtllib = [], None
        current = -1
        with open(path) as fh:
            for line in fh:
                key, _, rest = line.strip().partition(' ')
                if key == 'v':
                    pos.append(rest.split()[:3])
                elif key == 'vt':
                    tex.append(rest.split()[:2])
                elif key == 'f':
                    p, t = self._corners(rest.split())
                    for k in range(1, len(p) - 1):                     # a polygon as a fan around its first corner
                        corners_p.append((p[0], p[k], p[k + 1]))
                        corners_t.append((t[0], t[k], t[k + 1]))
                        tri_mat.append(current)
                elif key == 'usemtl':
                    name = rest.split()[0]
                    if name not in self.materials:
                        self.materials.append(name)
                    current = self.materials.index(name)
                elif key == 'mtllib':
                    self.mtllib = rest.split()[0]
        self.positions = np.asarray(pos, dtype=np.float32).reshape(-1, 3)
        self.texcoords = np.asarray(tex, dtype=np.float32).reshape(-1, 2)
        self.tri_pos = np.asarray(corners_p, dtype=np.int32).reshape(-1, 3) - 1
        self.tri_tex = np.asarray(corners_t, dtype=np.int32).reshape(-1, 3) - 1
        self.tri_material = np.asarray(tri_mat, dtype=np.int32)

    @staticmethod
    def _corners(tokens):
        """'7', '7/3', '7/3/2', '7//2' -> position indices, texcoord indices (0 where absent), one-based as in the file"""
        p, t = [], []
        for tok in tokens:
            fields = tok.split('/')
            p.append(int(fields[0]))
            t.append(int(fields[1]) if len(fields) > 1 and fields[1] else 0)
        return p, t


def _read_materials(path):
    """.mtl -> OrderedDict name -> {'Kd': [3] float array or None, 'map_Kd': file name or None}"""
    mats, cur = OrderedDict(), None
    with open(path) as fh:
        for line in fh:
            key, _, rest = line.strip().partition(' ')
            if key == 'newmtl':
                cur = mats.setdefault(rest.split()[0], {'Kd': None, 'map_Kd': None})
            elif cur is not None and key == 'Kd':
                cur['Kd'] = np.asarray(rest.split()[:3], dtype=np.float64)
            elif cur is not None and key == 'map_Kd':
                cur['map_Kd'] = rest.split()[0]
    return mats


def load_mtl(filename_mtl):
    """(colors {material: Kd [3]}, texture_filenames {material: map_Kd file}) of an .mtl file -- the pair NR/load_obj.py's
    helper of the same name returns."""
    mats = _read_materials(filename_mtl)
    return ({k: m['Kd'] for k, m in mats.items() if m['Kd'] is not None},
            {k: m['map_Kd'] for k, m in mats.items() if m['map_Kd'] is not None})


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("load_obj returns device tensors, like the reference (load_obj.py:120,140): no GPU found")
    return torch.device("cuda")


# ---- uv image -> texture cubes ------------------------------------------------------------------------------------
def load_textures_from_image(image, faces, textures, is_update, texture_wrapping=0, use_bilinear=True):
    """The reference extension's `load_textures_cuda.load_textures` (NR/cuda/load_textures_cuda.cpp:19-37): image [H,W,3],
    faces [F,3,2] uv corners, textures [F,ts,ts,ts,3] (updated in place and returned), is_update [F] int32."""
    _lib.require_device(image, faces, textures, is_update, names=("image", "faces", "textures", "is_update"))
    if any(t.dtype != torch.float32 for t in (image, faces, textures)):
        raise TypeError("image, faces and textures must be float32")
    if is_update.dtype != torch.int32:
        raise TypeError("is_update must be int32")
    F = faces.shape[0]
    if image.dim() != 3 or image.shape[2] != 3:
        raise ValueError("image must be [H, W, 3]")
    if faces.dim() != 3 or tuple(faces.shape[1:]) != (3, 2):
        raise ValueError("faces must be [num_faces, 3, 2]")
    if textures.dim() != 5 or textures.shape[0] != F or is_update.shape[0] != F:
        raise ValueError("textures must be [num_faces, ts, ts, ts, 3] and is_update [num_faces]")
    _lib.check(_lib.lib().d3m_load_textures(
        _lib.ptr(image), _lib.ptr(is_update), _lib.ptr(faces), _lib.ptr(textures), F, textures.shape[1], image.shape[0],
        image.shape[1], int(texture_wrapping), int(bool(use_bilinear)), _lib.stream_ptr()), "d3m_load_textures")
    return textures


def _textures_of_scene(scene, obj_dir, filename_mtl, texture_size, texture_wrapping, use_bilinear):
    dev = _device()
    F = scene.tri_tex.shape[0]
    # a corner without a texture coordinate indexes "vt 0" in the reference (one-based 0 -> -1 -> the LAST coordinate)
    face_uv = torch.from_numpy(scene.texcoords[scene.tri_tex]).to(dev)                   # [F,3,2]
    textures = torch.full((F, texture_size, texture_size, texture_size, 3), 0.5, dtype=torch.float32, device=dev)
    mats = _read_materials(filename_mtl)
    names = np.asarray(scene.materials, dtype=object)
    of_tri = np.where(scene.tri_material >= 0, scene.tri_material, len(scene.materials))
    tri_name = np.append(names, '')[of_tri] if F else np.zeros(0, dtype=object)         # '' before any usemtl
    for name, m in mats.items():                          # constant colours first ...
        if m['Kd'] is not None:
            sel = torch.from_numpy(tri_name == name).to(dev)
            textures[sel] = torch.as_tensor(m['Kd'], dtype=torch.float32, device=dev)
    for name, m in mats.items():                          # ... then every image over the faces of its material
        if m['map_Kd'] is not None:
            image = torch.from_numpy(_read_image(os.path.join(obj_dir, m['map_Kd']))).to(dev)
            update = torch.from_numpy((tri_name == name).astype(np.int32)).to(dev)
            load_textures_from_image(image, face_uv, textures, update, texture_wrapping_dict[texture_wrapping], use_bilinear)
    return textures


def load_textures(filename_obj, filename_mtl, texture_size, texture_wrapping='REPEAT', use_bilinear=True):
    """Per-face texture cubes [F, ts, ts, ts, 3] of an .obj with `vt` coordinates and an .mtl: 0.5 grey, Kd colours per
    material, then every map_Kd image sampled onto the faces of its material (what NR/load_obj.py:33-98 produces)."""
    return _textures_of_scene(_ObjScene(filename_obj), os.path.dirname(filename_obj), filename_mtl, texture_size,
                              texture_wrapping, use_bilinear)


def _fit_unit_cube(vertices):
    """shift the bounding box's corner to the origin, scale the longest side to 2, centre (load_obj.py:155-159)"""
    shifted = vertices - vertices.amin(0, keepdim=True)
    scaled = shifted * (2.0 / shifted.abs().max())
    return scaled - scaled.amax(0, keepdim=True) / 2


def load_obj(filename_obj, normalization=True, texture_size=4, load_texture=False, texture_wrapping='REPEAT',
             use_bilinear=True):
    """
    Load a Wavefront .obj file: vertices (`v x y z`) and faces (`f a b c ...`, polygons are fanned into triangles,
    `a/b/c` corners use the position index).  Returns (vertices [V,3] f32, faces [F,3] i32) on the GPU and, with
    `load_texture`, the per-face texture cubes [F,ts,ts,ts,3] of its `mtllib`; `normalization` rescales the vertices
    into the cube [-1, 1]^3.
    """
    scene = _ObjScene(filename_obj)
    dev = _device()
    vertices = torch.from_numpy(scene.positions).to(dev)
    faces = torch.from_numpy(scene.tri_pos).to(dev)
    textures = None
    if load_texture:
        if scene.mtllib is None:
            raise Exception('Failed to load textures.')
        obj_dir = os.path.dirname(filename_obj)
        textures = _textures_of_scene(scene, obj_dir, os.path.join(obj_dir, scene.mtllib), texture_size, texture_wrapping,
                                      use_bilinear)
    if normalization:
        vertices = _fit_unit_cube(vertices)
    return (vertices, faces, textures) if load_texture else (vertices, faces)


# ---- texture cubes -> atlas image -----------------------------------------------------------------------------------
def _atlas_layout(num_faces, tile, device):
    """tiles per row / column of the square-ish atlas and the tile-space corners [F,3,2] (x, y) of every face's triangle:
    (left, top), (left, bottom), (right, bottom) of its tile, in pixels"""
    per_row = int((num_faces - 1.) ** 0.5) + 1
    rows = int((num_faces - 1.) / per_row) + 1
    f = torch.arange(num_faces, device=device)
    left, top = ((f % per_row) * tile).float(), (torch.div(f, per_row, rounding_mode='floor') * tile).float()
    right, bottom = left + (tile - 1), top + (tile - 1)
    corners = torch.stack((torch.stack((left, top), -1), torch.stack((left, bottom), -1), torch.stack((right, bottom), -1)), 1)
    return per_row, rows, corners.contiguous()


def create_texture_image(textures, texture_size_out=16):
    """Texture atlas of per-face texture cubes: returns (image [rows*tso, per_row*tso, 3] numpy with the rows flipped for
    image files, per-face uv corners [F,3,2] numpy in [0,1]) -- the pair NR/save_obj.py:10-38 returns."""
    textures = textures.detach().float().contiguous()
    _lib.require_device(textures, names=("textures",))
    F, ts_in = textures.shape[:2]
    per_row, rows, corners = _atlas_layout(F, texture_size_out, textures.device)
    image = torch.empty(rows * texture_size_out, per_row * texture_size_out, 3, dtype=torch.float32, device=textures.device)
    _lib.check(_lib.lib().d3m_create_texture_image(
        _lib.ptr(corners), _lib.ptr(textures), _lib.ptr(image), F, ts_in, image.shape[0], image.shape[1], per_row,
        ATLAS_EPS, _lib.stream_ptr()), "d3m_create_texture_image")
    extent = np.array([image.shape[1] - 1, image.shape[0] - 1], dtype=np.float32)
    return image.flip(0).cpu().numpy(), corners.cpu().numpy() / extent


# ---- writer -------------------------------------------------------------------------------------------------------
def _rows(fmt, array):
    """every row of a 2-D array through one format string, newline-terminated"""
    return ''.join(fmt % tuple(r) for r in array.tolist())


def save_obj(filename, vertices, faces, textures=None):
    """Write vertices [V,3] and faces [F,3] as a Wavefront .obj; with `textures` [F,ts,ts,ts,3] also a .png texture
    atlas, per-corner `vt` coordinates and a .mtl next to it (the files NR/save_obj.py:41-82 writes)."""
    if vertices.ndimension() != 2 or faces.ndimension() != 2:
        raise AssertionError("vertices must be [V,3] and faces [F,3]")
    stem = os.path.splitext(filename)[0]
    v = vertices.detach().cpu().numpy().astype(np.float64)
    tri = faces.detach().cpu().numpy().astype(np.int64) + 1
    chunks = ['# %s\n#\n\n' % os.path.basename(filename)]
    if textures is None:
        chunks += [_rows('v %.8f %.8f %.8f\n', v), '\n', _rows('f %d %d %d\n', tri)]
    else:
        atlas, uv = create_texture_image(textures)
        _write_image(stem + '.png', atlas)
        material = 'material_1'
        with open(stem + '.mtl', 'w') as fh:
            fh.write('newmtl %s\nmap_Kd %s\n' % (material, os.path.basename(stem + '.png')))
        vt = 3 * np.arange(tri.shape[0], dtype=np.int64)[:, None] + np.array([1, 2, 3])     # corner k of face i: vt 3i + k
        corners = np.stack((tri, vt), -1).reshape(tri.shape[0], 6)                           # v/vt v/vt v/vt
        chunks += ['mtllib %s\n\n' % os.path.basename(stem + '.mtl'), _rows('v %.8f %.8f %.8f\n', v), '\n',
                   _rows('vt %.8f %.8f\n', uv.reshape(-1, 2).astype(np.float64)), '\n', 'usemtl %s\n' % material,
                   _rows('f %d/%d %d/%d %d/%d\n', corners), '\n']
    with open(filename, 'w') as fh:
        fh.write(''.join(chunks))


# ---- mesh holder ------------------------------------------------------------------------------------------------------
class Mesh(nn.Module):
    """A triangle mesh with a learnable texture (the holder of NR/mesh.py): vertices [V,3], faces [F,3], textures
    [F,ts,ts,ts,3] -- given, or a small random nn.Parameter."""

    def __init__(self, vertices, faces, textures=None, texture_size=4):
        super().__init__()
        self.vertices, self.faces = vertices, faces
        self.num_vertices, self.num_faces = vertices.shape[0], faces.shape[0]
        if textures is None:
            textures = nn.Parameter(0.05 * torch.randn(self.num_faces, texture_size, texture_size, texture_size, 3))
        self.textures = textures
        self.texture_size = textures.shape[1]

    @classmethod
    def fromobj(cls, filename_obj, normalization=True, load_texture=False, texture_size=4):
        loaded = load_obj(filename_obj, normalization=normalization, texture_size=texture_size, load_texture=load_texture)
        return cls(loaded[0], loaded[1], loaded[2] if load_texture else None, texture_size)
