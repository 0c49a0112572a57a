"""Debug (CPU): per-tile face counts of the forward binning on the headline workload."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from deep3dmap_amd import synthetic
from oracle import nr_oracle as O
n, S, views, TILE = 225, 512, 8, 8
verts, tri = synthetic.grid_mesh(n)
v = torch.from_numpy(verts)[None]; t = torch.from_numpy(tri).long()
tot = []
for i in range(views):
    eye = torch.tensor(O.get_points_from_angles(2.732, 20*np.sin(2*np.pi*i/views), 360*i/views)).float()
    vs = O.perspective(O.look_at(v, eye), angle=30)[0].numpy()
    f = vs[t.numpy()]                                   # [F,3,3]
    f = np.concatenate([f, f[:, ::-1]], 0)
    x, y = f[..., 0], f[..., 1]
    front = (y[:, 2]-y[:, 0])*(x[:, 1]-x[:, 0]) - (y[:, 1]-y[:, 0])*(x[:, 2]-x[:, 0]) < 0      # backside test sign may differ; take majority below
    lx = np.ceil((x.min(1)*S + S - 1)/2); hx = np.floor((x.max(1)*S + S - 1)/2)
    ly = np.ceil((y.min(1)*S + S - 1)/2); hy = np.floor((y.max(1)*S + S - 1)/2)
    ok = (lx <= hx) & (ly <= hy) & (hx >= 0) & (hy >= 0) & (lx <= S-1) & (ly <= S-1)
    for name, sel in (("A", front & ok), ("B", ~front & ok)):
        cnt = np.zeros((S//TILE, S//TILE), np.int64)
        tx0 = (np.clip(lx[sel], 0, S-1)//TILE).astype(int); tx1 = (np.clip(hx[sel], 0, S-1)//TILE).astype(int)
        ty0 = (np.clip(ly[sel], 0, S-1)//TILE).astype(int); ty1 = (np.clip(hy[sel], 0, S-1)//TILE).astype(int)
        for a, b, c, d in zip(tx0, tx1, ty0, ty1):
            cnt[c:d+1, a:b+1] += 1
        print(f"view {i} orientation {name}: faces {sel.sum()}, pairs {cnt.sum()}, busy tiles {(cnt>0).sum()}, mean/busy {cnt.sum()/max((cnt>0).sum(),1):.1f}, max {cnt.max()}, tiles>256: {(cnt>256).sum()}, >1024: {(cnt>1024).sum()}")
    if i >= 1: break
