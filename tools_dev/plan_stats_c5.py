"""Statistics of the edge gradient's plan on the headline workload (GPU box): crossings per line, outward / inward walks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deep3dmap_amd import neural_renderer as nr, synthetic, _lib
from deep3dmap_amd.neural_renderer import rasterize_ops as ops
from deep3dmap_amd.neural_renderer.mesh_ops import gather_faces
from deep3dmap_amd.neural_renderer.rasterize import _raster_forward
B, S, n = 2, 1024, 709
v, tri = synthetic.grid_mesh(n)
eyes = torch.from_numpy(synthetic.camera_ring(256)[[5, 77]]).cuda()
vt = torch.from_numpy(v).cuda()[None].expand(B, -1, -1).contiguous()
ft = torch.from_numpy(tri).cuda()[None].expand(B, -1, -1).contiguous()
faces = gather_faces(nr.look_at(vt, eyes, _perspective_angle=30), ft, True)
m, _ = _raster_forward(faces, None, S, 0.1, 100.0, 1e-3, None, False, True, True, False)
F = faces.shape[1]
vis = ops.visibility(m["face_index_map"], F)
plan = ops.edge_plan(faces, m["face_index_map"], vis, S)
torch.cuda.synchronize()
nl = B * 2 * S
al = lambda x: (x + 255) // 256 * 256
raw = plan.cpu().numpy()
line_count = raw[:nl * 4].view(np.int32)
off_cursor = al(nl * 4)
cursor = raw[off_cursor:off_cursor + nl * 4].view(np.int32)
off_alloc = off_cursor + al(nl * 4)
alloc = raw[off_alloc:off_alloc + 8].view(np.int32)
nf = B * F
off_lane_cross = off_alloc + 256
off_lane_block = off_lane_cross + al(nf * 6 * 8)
off_slice = off_lane_block + al((nf // 42 + 2) * 4)
off_xrec = off_slice + al(nl * 8)
total = int(alloc[0])
print("crossings", total, "per view", total / B, "lines with crossings", int((cursor > 0).sum()), "of", nl)
act = cursor[cursor > 0]
print("crossings per active line: mean %.1f median %d p90 %d max %d" % (act.mean(), np.median(act), np.percentile(act, 90), act.max()))
print("chunks of 256:", int(np.ceil(act / 256).sum()), " thread-slots used in set-up: %.2f" % (2 * act.sum() / (np.ceil(act / 256).sum() * 512)))
slices = raw[off_slice:off_slice + nl * 8].view(np.int32).reshape(nl, 2)
idx = np.concatenate([np.arange(c) + s0 for (s0, n0), c in zip(slices, cursor) if c > 0])
print("records written %d of %d crossings (%.3f): the rest is outside the image or cannot contribute" % (len(idx), total, len(idx) / total))
cap_bytes = len(raw) - off_xrec
x = raw[off_xrec:off_xrec + (idx.max() + 1) * 32].view(np.uint32).reshape(-1, 8)[idx]
bits = x[:, 3] & 0xFF
d1_in = (x[:, 3] >> 8).astype(np.int64)
alive, owner, oriented = (bits & 1) != 0, (bits & 16) != 0, (bits & 32) != 0
in_from, in_to = (x[:, 4] & 0xFFFF).astype(np.int64), (x[:, 4] >> 16).astype(np.int64)
print("of the records: outward walks (owner) %.3f  inward oriented %.3f" % ((alive & owner).mean(), (alive & oriented).mean()))
il = (in_to - in_from + 1)[alive]
print("inward length (unclipped): mean %.2f  <=6: %.3f  max %d" % (il.mean(), (il <= 6).mean(), il.max()))
dirpos = (bits & 2) != 0
ol = np.where(dirpos, S - 1 - d1_in, d1_in)[alive & owner]
print("outward length to the border (unclipped): mean %.1f" % ol.mean())
