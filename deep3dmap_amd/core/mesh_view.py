"""Plumbing-level stand-in for deep3dmap's Pt3dRenderer (deep3dmap/core/renderer/renderer_pt3d.py:32-98).

Pt3dRenderer's arithmetic lives in pytorch3d (third party, not vendored, not installed here: SURVEY.md
8c "parity unpinned"), so it is NOT reproduced.  What BASELINE.json's first config asks for is plumbing:
config -> renderer -> silhouette loss decreases on an icosphere seen from 2 views at 64x64.  This class
provides that on the native rasterizer: orthographic-ish look_at views of a mesh, silhouettes and depth."""
import torch

from .. import neural_renderer as nr
from .losses import silhouette_loss


class MeshViewRenderer:
    def __init__(self, image_size=64, texture_size=64, distance=2.7):
        self.image_size, self.texture_size, self.distance = image_size, texture_size, distance
        self.renderer = nr.Renderer(image_size=image_size, camera_mode="look_at", anti_aliasing=True)

    def silhouettes(self, vertices, faces, azimuths, elevation=0.0):
        """vertices [V,3], faces [F,3] -> silhouettes [n_views, s, s] for the given azimuths (degrees)."""
        eyes = torch.tensor([nr.get_points_from_angles(self.distance, elevation, float(a)) for a in azimuths],
                            dtype=torch.float32, device=vertices.device)
        self.renderer.eye = eyes
        n = len(azimuths)
        return self.renderer(vertices[None].expand(n, -1, -1), faces[None].expand(n, -1, -1), mode="silhouettes")

    def fit(self, vertices, faces, target, azimuths, steps=50, lr=0.01):
        """Silhouette fit with Adam, as pnpmodules/neural_renderer/examples/example2.py:43-47,62 does:
        minimise sum((silhouette - target)^2) over the vertices.  Returns the loss history."""
        opt = torch.optim.Adam([vertices], lr=lr)
        history = []
        for _ in range(steps):
            opt.zero_grad()
            loss = silhouette_loss(self.silhouettes(vertices, faces, azimuths), target)
            loss.backward()
            opt.step()
            history.append(float(loss.detach()))
        return history
