"""One rank of the camera-sharded multi-view fit, run as a child process by tests/test_gpu_multirank.py (and usable by
hand).  Every rank renders its contiguous share of `--views` cameras of the shared mesh, back-propagates, all-reduces
[loss | grad_vertices | grad_textures] and writes what it holds afterwards -- which must be the gradient of the
objective over ALL cameras -- to --out.  With --single-device every rank uses cuda:0 and the collective runs over gloo
(host-staged): the N>1 code path of bench.py on a one-GPU box.  Exit code != 0 on any failure (a GPU fault aborts)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=32)
    ap.add_argument("--mesh-n", type=int, default=225)
    ap.add_argument("--image-size", type=int, default=512)
    ap.add_argument("--graph", type=int, default=1)
    ap.add_argument("--replays", type=int, default=6)
    ap.add_argument("--materialise-images", type=int, default=0)
    ap.add_argument("--split-exchange", default="auto", choices=["auto", "0", "1"],
                    help="the step in two parts with the texture gradient's all-reduce started between them "
                         "(MultiViewFit(split_exchange=...); auto: on with more than one rank and a big local batch)")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group(backend="gloo")
    from deep3dmap_amd import synthetic
    from deep3dmap_amd.multiview import MultiViewFit
    v, tri = synthetic.grid_mesh(args.mesh_n)
    tex = synthetic.random_textures(tri.shape[0], 2)
    eyes = synthetic.camera_ring(args.views)
    fit = MultiViewFit(v, tri, tex, eyes, image_size=args.image_size, rank=rank, world_size=world,
                       objective_in_renderer=not args.materialise_images,
                       split_exchange=None if args.split_exchange == "auto" else bool(int(args.split_exchange)))
    if args.split_exchange != "auto":
        expect_split = bool(int(args.split_exchange)) and not args.materialise_images
        assert fit.split_exchange == expect_split, (fit.split_exchange, expect_split)
    else:       # on with more than one rank when this rank's batch is big enough to run on one stream anyway (multiview.py)
        assert not fit.split_exchange or (world > 1 and args.views // world > 16 and not args.materialise_images)
    fit.set_targets_from(synthetic.perturb(v))
    loss, gv, gt = fit.step()
    eager = (float(loss), gv.clone(), gt.clone())
    if args.graph:
        fit.capture_graph()
        assert fit.graph_captured and len(fit._runner.graph) == (2 if fit.split_exchange else 1)
    for i in range(args.replays):
        loss, gv, gt = fit.step()
        torch.cuda.synchronize()
        assert abs(float(loss) - eager[0]) <= 1e-5 * abs(eager[0]), (i, float(loss), eager[0])
        rel = float(torch.linalg.norm(gv - eager[1]) / torch.linalg.norm(eager[1]))
        assert rel < 1e-4, (i, rel)
    np.savez(args.out + f".rank{rank}.npz", loss=float(loss), gv=gv.cpu().numpy(), gt=gt.cpu().numpy(),
             mask_sum=float(fit.mask_sum) if fit.mask_sum is not None else float(fit.targets[2].sum()))
    print(f"rank {rank}/{world}: loss {float(loss):.7f} graph={bool(args.graph)} replays={args.replays} "
          f"split_exchange={fit.split_exchange} ok", flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
