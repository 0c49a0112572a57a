"""Deterministic scenes for the face3d rasterizer family tests (shared by the golden generator and the tests)."""
import numpy as np


def grid_scene(n, h, w, seed, jitter=0.6, big=True):
    """A jittered n x n height-field mesh in pixel coordinates (vertices [3, nver], triangles [3, ntri] int32) plus
    per-vertex colours, texture data and a shifted copy of the vertices; optionally a few extra triangles that hit the
    reference's special cases: an exact duplicate (equal depth -> lowest index wins), one that is off screen, one that
    reaches over the 2-pixel border band, and one that covers most of the image (the wave-per-triangle path)."""
    rng = np.random.default_rng(seed)
    gx, gy = np.meshgrid(np.linspace(4, w - 5, n), np.linspace(4, h - 5, n))
    z = 20 * np.sin(gx / 7) * np.cos(gy / 5) + rng.normal(size=gx.shape)
    v = np.stack([gx + rng.normal(scale=jitter, size=gx.shape), gy + rng.normal(scale=jitter, size=gx.shape), z]).reshape(3, -1)
    idx = np.arange(n * n).reshape(n, n)
    t = np.concatenate([np.stack([idx[:-1, :-1], idx[:-1, 1:], idx[1:, :-1]], 0).reshape(3, -1),
                        np.stack([idx[:-1, 1:], idx[1:, 1:], idx[1:, :-1]], 0).reshape(3, -1)], 1).astype(np.int32)
    n_textured = t.shape[1]
    if big:
        nv = v.shape[1]
        extra_v = np.array([[-5, 1.5, 9], [2, -3, 1.5], [5, 5, 5.],                     # border band
                            [-40, -40, -30], [-20, -45, -30], [-30, -20, -30],           # off screen
                            [1, 1, -25], [w - 2, 2, -25], [w / 2, h - 2, -25]]).T       # most of the image, far away
        v = np.concatenate([v, extra_v], 1)
        t = np.concatenate([t, t[:, :3], np.array([[nv, nv + 1, nv + 2], [nv + 3, nv + 4, nv + 5],
                                                   [nv + 6, nv + 7, nv + 8]], np.int32).T], 1)
        n_textured += 3          # the duplicates; the border / off-screen / huge ones would extrapolate texture
        #                          coordinates out of the texture, which the reference reads without a bounds check
    colors = rng.random(v.shape)
    texture = rng.random((19, 23, 3))
    # MP:187 makes the y of a texture corner tex_coords[2 + vertex index]: keep every entry a valid row AND column
    tex_coords = rng.uniform(0.5, 17.5, (2, v.shape[1]))
    src_image = rng.random((h, w, 3))
    src_vertices = v.copy()
    src_vertices[:2] += 1.3
    return dict(vertices=v, triangles=t, colors=colors, texture=texture, tex_coords=tex_coords, src_image=src_image,
                src_vertices=src_vertices, h=h, w=w, n_textured=n_textured)


def run_family(R, s):
    """Every function of the family on scene `s` through module `R` (oracle.mesh_oracle or deep3dmap_amd.mesh_cython.render)."""
    v, t, h, w = s["vertices"], s["triangles"], s["h"], s["w"]
    out = {}
    out["render_colors"] = R.render_colors(v, t, s["colors"], h, w)
    out["render_colors_bg"] = R.render_colors(v, t, s["colors"], h, w, isBG=True)
    out["triangle_buffer"] = R.get_triangle_buffer(v, t, h, w)
    out["vis"] = R.vis_of_vertices(v, t, h, w)
    out["norm"] = R.get_norm_direction(v, t)
    tt = np.ascontiguousarray(t[:, :s["n_textured"]])
    out["texture_nearest"] = R.render_texture(v, tt, s["texture"], s["tex_coords"], tt, h, w, mapping_type='nearest')
    out["texture_bilinear"] = R.render_texture(v, tt, s["texture"], s["tex_coords"], tt, h, w, mapping_type='bilinear')
    out["map_texture"] = R.map_texture(s["src_image"], s["src_vertices"], v, out["triangle_buffer"], t, h, w)
    pncc = (v - v.min(1, keepdims=True)) / (v.max(1, keepdims=True) - v.min(1, keepdims=True))
    x, ind = R.get_correspondence(out["render_colors"] * 0 + R.render_colors(v, t, pncc, h, w), pncc)
    out["corr_x"], out["corr_ind"] = x, ind
    return out


SCENES = {"g12_40x48": (12, 40, 48, 1), "g30_64x64": (30, 64, 64, 2), "g9_17x23": (9, 17, 23, 3)}
