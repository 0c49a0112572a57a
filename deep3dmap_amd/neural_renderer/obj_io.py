"""Wavefront OBJ geometry I/O and the `Mesh` holder of the neural_renderer package surface
(NR/load_obj.py:100-164, NR/save_obj.py:41-82, NR/mesh.py:6-43).

Geometry only: texture loading (`load_texture=True`) and texture export need the reference's
`load_textures` / `create_texture_image` kernels and an image codec; they are listed under "next" in
DESIGN.md and raise NotImplementedError here."""
import os

import numpy as np
import torch
import torch.nn as nn


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("load_obj returns device tensors, like the reference (load_obj.py:120,140): no GPU found")
    return torch.device("cuda")


def load_obj(filename_obj, normalization=True, texture_size=4, load_texture=False, texture_wrapping='REPEAT',
             use_bilinear=True):
    """
    Load a Wavefront .obj file: vertices (`v x y z`) and faces (`f a b c ...`, polygons are fanned into
    triangles, `a/b/c` index triplets use the vertex index).  Returns (vertices [V,3] f32, faces [F,3] i32)
    on the GPU; `normalization` rescales into the cube [-1, 1]^3 the way load_obj.py:155-159 does.
    """
    if load_texture:
        raise NotImplementedError("load_obj(load_texture=True) needs the load_textures kernel (DESIGN.md, 'next')")
    vertices, faces = [], []
    with open(filename_obj) as f:
        for line in f:
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
    if normalization:
        vertices = vertices - vertices.min(0)[0][None, :]
        vertices = vertices / torch.abs(vertices).max()
        vertices = vertices * 2
        vertices = vertices - vertices.max(0)[0][None, :] / 2
    return vertices, faces


def save_obj(filename, vertices, faces, textures=None):
    """Write vertices [V,3] and faces [F,3] as a Wavefront .obj (save_obj.py:41-82, geometry part)."""
    assert vertices.ndimension() == 2
    assert faces.ndimension() == 2
    if textures is not None:
        raise NotImplementedError("save_obj(textures=...) needs the create_texture_image kernel (DESIGN.md, 'next')")
    v = vertices.detach().cpu().numpy()
    fa = faces.detach().cpu().numpy()
    with open(filename, 'w') as f:
        f.write('# %s\n#\n\n' % os.path.basename(filename))
        for vertex in v:
            f.write('v %.8f %.8f %.8f\n' % (vertex[0], vertex[1], vertex[2]))
        f.write('\n')
        for face in fa:
            f.write('f %d %d %d\n' % (face[0] + 1, face[1] + 1, face[2] + 1))


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
        vertices, faces = load_obj(filename_obj, normalization=normalization, texture_size=texture_size,
                                   load_texture=load_texture)
        return cls(vertices, faces, None, texture_size)
