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
    galpha = torch.randn(B, S, S, device=dev) * alpha          # masked, as with the bench's loss
    rgb = torch.rand(B, S, S, 3, device=dev); grgb = torch.randn(B, S, S, 3, device=dev) * alpha[..., None]
    L = _lib.lib()
    nbytes = L.d3m_backward_pixel_map_workspace_bytes(B, F, S)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    gf = torch.zeros_like(faces)
    _lib.check(L.d3m_backward_pixel_map(_lib.ptr(faces), _lib.ptr(m['face_index_map']), _lib.ptr(rgb), _lib.ptr(alpha), _lib.ptr(grgb), _lib.ptr(galpha), _lib.ptr(gf), B, F, S, 1e-3, 1, 1, _lib.ptr(ws), nbytes, None, None, _lib.stream_ptr()), 'k4')
    torch.cuda.synchronize()
    px, nf, nl = B*S*S, B*F, B*2*S
    o = 0
    o += al(px*16) + al(px*8) + al(px*16) + al(px*8)
    off_visible = o; o += al(nf*4)
    off_line_count = o; o += al(nl*4)
    off_line_cursor = o; o += al(nl*4)
    o += al(nl*4) * 2
    off_alloc = o; o += 256
    o += al(nf*4) + al(nf*48)*2
    off_line_offset = o; o += al(nl*4)
    o += al((nf//1024 + 2)*4) + al((nf//42 + 2)*4)
    off_items = o
    w = ws.cpu().numpy()
    alloc = w[off_alloc:off_alloc+12].view(np.int32)
    lc = w[off_line_cursor:off_line_cursor+nl*4].view(np.int32)
    lo = w[off_line_offset:off_line_offset+nl*4].view(np.int32)
    n_rec = int(lc.sum())
    recs = []
    for l in np.nonzero(lc)[0]:
        recs.append(w[off_items + int(lo[l])*48: off_items + (int(lo[l])+int(lc[l]))*48].view(np.uint32).reshape(-1, 12))
    items = np.concatenate(recs)
    frm, to = items[:,2] & 0xFFFF, items[:,2] >> 16
    ln = (to.astype(np.int64) - frm + 1)
    inward = items[:,0] & 1
    print(f"crossings {int(alloc[0])}; queued segments {n_rec}; nonempty lines {np.count_nonzero(lc)}; per line mean {lc[lc>0].mean():.0f} max {lc.max()}")
    print(f"pixel visits {ln.sum():,}  mean len {ln.mean():.1f} median {np.median(ln)}  inward frac {inward.mean():.3f}")
    print("len hist", dict(zip([8,16,32,48,64,96,128,192,256,512], np.histogram(ln, bins=[0,8,16,32,48,64,96,128,192,256,512])[0].tolist())))
    # quarter-wave efficiency: groups of 4 consecutive records of a line, 16 px per iteration
    it_ideal = np.ceil(ln/16).sum()/4
    it_real = 0
    k = 0
    for r in recs:
        l = (r[:,2] >> 16).astype(np.int64) - (r[:,2] & 0xFFFF) + 1
        pad = (-len(l)) % 4
        l = np.concatenate([l, np.zeros(pad, np.int64)]).reshape(-1, 4)
        it_real += np.ceil(l.max(1)/16).sum()
    print(f"quarter-wave iterations: real {it_real:,.0f} vs balanced {it_ideal:,.0f} (efficiency {it_ideal/it_real:.2f})")
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
