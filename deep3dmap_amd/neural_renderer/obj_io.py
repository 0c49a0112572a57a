"""Wavefront OBJ I/O, texture assets and the `Mesh` holder of the neural_renderer package surface
(NR/load_obj.py:13-164, NR/save_obj.py:10-82, NR/mesh.py:6-43).

The two texture kernels (uv image -> per-face texture cubes, texture cubes -> atlas image) run in the HIP library
(`d3m_load_textures`, `d3m_create_texture_image`); text parsing and the PNG/JPEG codec (Pillow, where the reference
used scikit-image) stay on the host, as in the reference."""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _lib

texture_wrapping_dict = {'REPEAT': 0, 'MIRRORED_REPEAT': 1, 'CLAMP_TO_EDGE': 2, 'CLAMP_TO_BORDER': 3}


def _imread(path):
    from PIL import Image
    with Image.open(path) as im:
        if im.mode not in ('L', 'RGB', 'RGBA'):
            im = im.convert('RGBA' if 'A' in im.getbands() else 'RGB')
        return np.asarray(im)


def _imsave(path, image):
    from PIL import Image
    if image.dtype != np.uint8:
        image = np.clip(np.rint(np.nan_to_num(image) * 255.0), 0, 255).astype(np.uint8)
    Image.fromarray(image).save(path)


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("load_obj returns device tensors, like the reference (load_obj.py:120,140): no GPU found")
    return torch.device("cuda")


def load_mtl(filename_mtl):
    '''
    load color (Kd) and filename of textures from *.mtl  (NR/load_obj.py:13-30)
    '''
    texture_filenames = {}
    colors = {}
    material_name = ''
    with open(filename_mtl) as f:
        for line in f:
            parts = line.split()
            if not parts:
                continue
            if parts[0] == 'newmtl':
                material_name = parts[1]
            if parts[0] == 'map_Kd':
                texture_filenames[material_name] = parts[1]
            if parts[0] == 'Kd':
                colors[material_name] = np.array(list(map(float, parts[1:4])))
    return colors, texture_filenames


def load_textures_from_image(image, faces, textures, is_update, texture_wrapping=0, use_bilinear=True):
    """The reference's `load_textures_cuda.load_textures` (NR/cuda/load_textures_cuda.cpp:19-37): image [H,W,3],
    faces [F,3,2] uv corners, textures [F,ts,ts,ts,3] (updated in place and returned), is_update [F] int32."""
    _lib.require_device(image, faces, textures, is_update, names=("image", "faces", "textures", "is_update"))
    if image.dtype != torch.float32 or faces.dtype != torch.float32 or textures.dtype != torch.float32:
        raise TypeError("image, faces and textures must be float32")
    if is_update.dtype != torch.int32:
        raise TypeError("is_update must be int32")
    if image.dim() != 3 or image.shape[2] != 3:
        raise ValueError("image must be [H, W, 3]")
    if faces.dim() != 3 or tuple(faces.shape[1:]) != (3, 2):
        raise ValueError("faces must be [num_faces, 3, 2]")
    if textures.dim() != 5 or textures.shape[0] != faces.shape[0] or is_update.shape[0] != faces.shape[0]:
        raise ValueError("textures must be [num_faces, ts, ts, ts, 3] and is_update [num_faces]")
    _lib.check(_lib.lib().d3m_load_textures(
        _lib.ptr(image), _lib.ptr(is_update), _lib.ptr(faces), _lib.ptr(textures), faces.shape[0], textures.shape[1],
        image.shape[0], image.shape[1], int(texture_wrapping), int(bool(use_bilinear)), _lib.stream_ptr()),
        "d3m_load_textures")
    return textures


def load_textures(filename_obj, filename_mtl, texture_size, texture_wrapping='REPEAT', use_bilinear=True):
    """Per-face texture cubes [F, ts, ts, ts, 3] of an .obj with `vt` coordinates and an .mtl
    (NR/load_obj.py:33-98): Kd colours first, then every map_Kd image sampled onto the faces of its material."""
    with open(filename_obj) as f:
        lines = f.readlines()
    vertices = []
    for line in lines:
        parts = line.split()
        if parts and parts[0] == 'vt':
            vertices.append([float(v) for v in parts[1:3]])
    vertices = np.vstack(vertices).astype(np.float32)

    def vt_index(token):
        return int(token.split('/')[1]) if ('/' in token and '//' not in token) else 0

    faces, material_names, material_name = [], [], ''
    for line in lines:
        parts = line.split()
        if not parts:
            continue
        if parts[0] == 'f':
            vs = parts[1:]
            for i in range(len(vs) - 2):
                faces.append((vt_index(vs[0]), vt_index(vs[i + 1]), vt_index(vs[i + 2])))
                material_names.append(material_name)
        if parts[0] == 'usemtl':
            material_name = parts[1]
    dev = _device()
    faces = np.vstack(faces).astype(np.int32) - 1
    faces = torch.from_numpy(vertices[faces]).to(dev)

    colors, texture_filenames = load_mtl(filename_mtl)
    textures = torch.full((faces.shape[0], texture_size, texture_size, texture_size, 3), 0.5, dtype=torch.float32,
                          device=dev)
    material_names = np.array(material_names)
    for material_name, color in colors.items():
        sel = torch.from_numpy(material_names == material_name).to(dev)
        textures[sel] = torch.from_numpy(color.astype(np.float32)).to(dev)

    for material_name, filename_texture in texture_filenames.items():
        filename_texture = os.path.join(os.path.dirname(filename_obj), filename_texture)
        image = _imread(filename_texture).astype(np.float32) / 255.
        if image.ndim == 2:                                   # grey image
            image = np.stack((image,) * 3, -1)
        if image.shape[2] == 4:                               # alpha ignored
            image = image[:, :, :3]
        image = torch.from_numpy(image[::-1, :, :].copy()).to(dev)
        is_update = torch.from_numpy((material_names == material_name).astype(np.int32)).to(dev)
        textures = load_textures_from_image(image, faces, textures, is_update,
                                            texture_wrapping_dict[texture_wrapping], use_bilinear)
    return textures


def load_obj(filename_obj, normalization=True, texture_size=4, load_texture=False, texture_wrapping='REPEAT',
             use_bilinear=True):
    """
    Load a Wavefront .obj file: vertices (`v x y z`) and faces (`f a b c ...`, polygons are fanned into
    triangles, `a/b/c` index triplets use the vertex index).  Returns (vertices [V,3] f32, faces [F,3] i32)
    on the GPU; `normalization` rescales into the cube [-1, 1]^3 the way load_obj.py:155-159 does.
    """
    vertices, faces = [], []
    with open(filename_obj) as f:
        lines = f.readlines()
    for line in lines:
        parts = line.split()
        if not parts:
            continue
        if parts[0] == 'v':
            vertices.append([float(v) for v in parts[1:4]])
        elif parts[0] == 'f':
            vs = [int(p.split('/')[0]) for p in parts[1:]]
            for i in range(len(vs) - 2):
                faces.append((vs[0], vs[i + 1], vs[i + 2]))
    dev = _device()
    vertices = torch.from_numpy(np.asarray(vertices, dtype=np.float32).reshape(-1, 3)).to(dev)
    faces = torch.from_numpy(np.asarray(faces, dtype=np.int32).reshape(-1, 3)).to(dev) - 1
    textures = None
    if load_texture:
        for line in lines:
            if line.startswith('mtllib'):
                filename_mtl = os.path.join(os.path.dirname(filename_obj), line.split()[1])
                textures = load_textures(filename_obj, filename_mtl, texture_size, texture_wrapping=texture_wrapping,
                                         use_bilinear=use_bilinear)
        if textures is None:
            raise Exception('Failed to load textures.')
    if normalization:
        vertices = vertices - vertices.min(0)[0][None, :]
        vertices = vertices / torch.abs(vertices).max()
        vertices = vertices * 2
        vertices = vertices - vertices.max(0)[0][None, :] / 2
    if load_texture:
        return vertices, faces, textures
    return vertices, faces


def create_texture_image(textures, texture_size_out=16):
    """Texture atlas of per-face texture cubes (NR/save_obj.py:10-38): returns (image [tile_h*tso, tile_w*tso, 3]
    numpy, rows flipped for image files; per-face uv corners [F, 3, 2] numpy in [0, 1])."""
    textures = textures.detach().float().contiguous()
    _lib.require_device(textures, names=("textures",))
    num_faces, texture_size_in = textures.shape[:2]
    tile_width = int((num_faces - 1.) ** 0.5) + 1
    tile_height = int((num_faces - 1.) / tile_width) + 1
    tso = texture_size_out
    dev = textures.device
    image = torch.empty(tile_height * tso, tile_width * tso, 3, dtype=torch.float32, device=dev)
    # tile-space triangle corners [:, :, XY]; a few floats per face, laid out on the host
    face_nums = np.arange(num_faces)
    column, row = face_nums % tile_width, face_nums // tile_width
    vertices = np.zeros((num_faces, 3, 2), dtype=np.float32)
    vertices[:, 0, 0] = column * tso
    vertices[:, 0, 1] = row * tso
    vertices[:, 1, 0] = column * tso
    vertices[:, 1, 1] = (row + 1) * tso - 1
    vertices[:, 2, 0] = (column + 1) * tso - 1
    vertices[:, 2, 1] = (row + 1) * tso - 1
    vertices_dev = torch.from_numpy(vertices).to(dev)
    _lib.check(_lib.lib().d3m_create_texture_image(
        _lib.ptr(vertices_dev), _lib.ptr(textures), _lib.ptr(image), num_faces, texture_size_in, image.shape[0],
        image.shape[1], tile_width, 1e-5, _lib.stream_ptr()), "d3m_create_texture_image")
    vertices[:, :, 0] /= (image.shape[1] - 1)
    vertices[:, :, 1] /= (image.shape[0] - 1)
    image = image.cpu().numpy()[::-1, ::1]
    return image, vertices


def save_obj(filename, vertices, faces, textures=None):
    """Write vertices [V,3] and faces [F,3] as a Wavefront .obj; with `textures` [F,ts,ts,ts,3] also a .png texture
    atlas, per-face `vt` coordinates and a .mtl next to it (NR/save_obj.py:41-82)."""
    assert vertices.ndimension() == 2
    assert faces.ndimension() == 2
    if textures is not None:
        filename_mtl = filename[:-4] + '.mtl'
        filename_texture = filename[:-4] + '.png'
        material_name = 'material_1'
        texture_image, vertices_textures = create_texture_image(textures)
        _imsave(filename_texture, texture_image)
    v = vertices.detach().cpu().numpy()
    fa = faces.detach().cpu().numpy()
    with open(filename, 'w') as f:
        f.write('# %s\n#\n\n' % os.path.basename(filename))
        if textures is not None:
            f.write('mtllib %s\n\n' % os.path.basename(filename_mtl))
        for vertex in v:
            f.write('v %.8f %.8f %.8f\n' % (vertex[0], vertex[1], vertex[2]))
        f.write('\n')
        if textures is not None:
            for vertex in vertices_textures.reshape((-1, 2)):
                f.write('vt %.8f %.8f\n' % (vertex[0], vertex[1]))
            f.write('\n')
            f.write('usemtl %s\n' % material_name)
            for i, face in enumerate(fa):
                f.write('f %d/%d %d/%d %d/%d\n' % (face[0] + 1, 3 * i + 1, face[1] + 1, 3 * i + 2, face[2] + 1,
                                                   3 * i + 3))
            f.write('\n')
        else:
            for face in fa:
                f.write('f %d %d %d\n' % (face[0] + 1, face[1] + 1, face[2] + 1))
    if textures is not None:
        with open(filename_mtl, 'w') as f:
            f.write('newmtl %s\n' % material_name)
            f.write('map_Kd %s\n' % os.path.basename(filename_texture))


class Mesh(object):
    '''
    A simple holder of a triangle mesh with a learnable texture (NR/mesh.py:6-43).
    '''
    def __init__(self, vertices, faces, textures=None, texture_size=4):
        self.vertices = vertices
        self.faces = faces
        self.num_vertices = self.vertices.shape[0]
        self.num_faces = self.faces.shape[0]
        if textures is None:
            shape = (self.num_faces, texture_size, texture_size, texture_size, 3)
            self.textures = nn.Parameter(0.05 * torch.randn(*shape))
            self.texture_size = texture_size
        else:
            self.textures = textures
            self.texture_size = textures.shape[1]

    @classmethod
    def fromobj(cls, filename_obj, normalization=True, load_texture=False, texture_size=4):
        if load_texture:
            vertices, faces, textures = load_obj(filename_obj, normalization=normalization,
                                                 texture_size=texture_size, load_texture=True)
        else:
            vertices, faces = load_obj(filename_obj, normalization=normalization, texture_size=texture_size,
                                       load_texture=False)
            textures = None
        return cls(vertices, faces, textures, texture_size)
