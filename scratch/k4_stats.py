"""Debug: item statistics of the edge-gradient pass on the headline workload (reads the K4 workspace back)."""
import numpy as np, torch, sys
sys.path.insert(0, '.')
from deep3dmap_amd import _lib, synthetic
import deep3dmap_amd.neural_renderer as nr
from deep3dmap_amd.neural_renderer import rasterize_ops as ops
import importlib; R = importlib.import_module("deep3dmap_amd.neural_renderer.rasterize")

def al(v): return (v + 255) // 256 * 256
def main(n=225, S=512, views=8):
    dev = torch.device('cuda')
    verts, tri = synthetic.grid_mesh(n)
    verts = torch.from_numpy(verts).to(dev)[None]; tri = torch.from_numpy(tri).to(dev)[None]
    eyes = torch.stack([torch.tensor(nr.get_points_from_angles(2.732, 20*np.sin(2*np.pi*i/views), 360*i/views)) for i in range(views)]).float().to(dev)
    r = nr.Renderer(image_size=S, anti_aliasing=False, camera_mode='look_at', fill_back=True)
    r.eye = eyes
    v = nr.look_at(verts.expand(views,-1,-1), eyes); v = nr.perspective(v, angle=30)
    faces = nr.vertices_to_faces(v, tri.expand(views,-1,-1))
    faces = torch.cat([faces, faces[:, :, [2,1,0]]], 1).contiguous()
    B, F = faces.shape[:2]
    m, _ = R._raster_forward(faces, None, S, 0.1, 100., 1e-3, None, False, True, False, False)
    alpha = (m['face_index_map'] >= 0).float()
    galpha = torch.randn(B, S, S, device=dev)
    rgb = torch.rand(B, S, S, 3, device=dev); grgb = torch.randn(B, S, S, 3, device=dev)
    L = _lib.lib()
    nbytes = L.d3m_backward_pixel_map_workspace_bytes(B, F, S)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    gf = torch.zeros_like(faces)
    _lib.check(L.d3m_backward_pixel_map(_lib.ptr(faces), _lib.ptr(m['face_index_map']), _lib.ptr(rgb), _lib.ptr(alpha), _lib.ptr(grgb), _lib.ptr(galpha), _lib.ptr(gf), B, F, S, 1e-3, 1, 1, _lib.ptr(ws), nbytes, None, None, _lib.stream_ptr()), 'k4')
    torch.cuda.synchronize()
    px, nf, nl = B*S*S, B*F, B*2*S
    o = 0
    o += al(px*4)*3 + al(px*12)*2
    off_visible = o; o += al(nf*4)
    off_line_count = o; o += al(nl*4)
    off_line_cursor = o; o += al(nl*4)
    off_alloc = o; o += 256
    o += al(nf*4) + al(nf*24)*2 + al(nf*48) + al(nl*4)
    off_items = o
    w = ws.cpu().numpy()
    alloc = w[off_alloc:off_alloc+12].view(np.int32)
    n_items, n_vis = int(alloc[0]), int(alloc[2])
    lc = w[off_line_cursor:off_line_cursor+nl*4].view(np.int32)
    items = w[off_items:off_items+n_items*48].view(np.uint32).reshape(-1, 12)
    frm, to = items[:,2] & 0xFFFF, items[:,2] >> 16
    ln = (to.astype(np.int64) - frm + 1)
    inward = items[:,0] & 1
    print(f"visible faces {n_vis} of {nf}; items {n_items} ({n_items/n_vis:.2f}/visible face), queued {lc.sum()}")
    print(f"pixel visits {ln.sum():,}  mean len {ln.mean():.1f}  wave-iterations {np.ceil(ln/64).sum():,.0f}  inward frac {inward.mean():.3f}")
    print(f"items per line: mean {lc.mean():.0f} max {lc.max()} ; lines {nl}; nonempty {np.count_nonzero(lc)}")
    h = np.histogram(ln, bins=[0,7,16,32,64,128,256,512,1024])
    print("len hist", dict(zip(h[1][1:], h[0])))
main()

def edge_extents(n=225, S=512, views=8):
    dev = torch.device('cuda')
    verts, tri = synthetic.grid_mesh(n)
    verts = torch.from_numpy(verts).to(dev)[None]; tri = torch.from_numpy(tri).to(dev)[None]
    eyes = torch.stack([torch.tensor(nr.get_points_from_angles(2.732, 20*np.sin(2*np.pi*i/views), 360*i/views)) for i in range(views)]).float().to(dev)
    v = nr.look_at(verts.expand(views,-1,-1), eyes); v = nr.perspective(v, angle=30)
    faces = nr.vertices_to_faces(v, tri.expand(views,-1,-1))
    faces = torch.cat([faces, faces[:, :, [2,1,0]]], 1).contiguous()
    B, F = faces.shape[:2]
    m, _ = R._raster_forward(faces, None, S, 0.1, 100., 1e-3, None, False, True, False, False)
    fi = m['face_index_map']
    vis = torch.zeros(B, F, dtype=torch.bool, device=dev)
    for b in range(B):
        u = torch.unique(fi[b]); u = u[u >= 0]; vis[b, u.long()] = True
    px = (faces[..., :2] * S + S - 1) / 2          # [B,F,3,2]
    ext = []
    for e in range(3):
        a, c = px[:, :, e], px[:, :, (e + 1) % 3]
        for ax in range(2):
            lo = torch.ceil(torch.minimum(a[..., ax], c[..., ax])).clamp(min=0)
            hi = torch.floor(torch.maximum(a[..., ax], c[..., ax])).clamp(max=S - 1)
            ext.append((hi - lo + 1).clamp(min=0))
    ext = torch.stack(ext, -1)[vis]            # [n_vis, 6]
    print("visible", int(vis.sum()), "d0-range per lane: mean %.2f max %d" % (float(ext.mean()), int(ext.max())))
    h = torch.histc(ext.flatten(), bins=20, min=0, max=20)
    print("hist 0..19:", h.int().tolist(), " >=20:", int((ext >= 20).sum()))
    # per-wave max (64 consecutive lanes in list order ~ index order)
    flat = ext.flatten()
    nw = flat.numel() // 64
    wmax = flat[:nw * 64].view(nw, 64).max(1)[0]
    print("per-wave max trip: mean %.1f, max %d; sum of maxes %d vs sum of means %.0f" % (float(wmax.mean()), int(wmax.max()), int(wmax.sum()), float(flat.sum()) / 64))
edge_extents()
