#!/usr/bin/env python3
"""bench.py -- rendered Mpix/s (forward + backward) of the multi-view fit on a 100k-triangle mesh at
512x512 (BASELINE.json metric), one rank per GPU, camera-sharded.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...          (no torchrun environment: bench.py starts that very command itself)

A step = render (rgb+depth+alpha) of this rank's views of the shared mesh, the loss against fixed
targets, the backward to per-vertex / per-texel gradients and (N > 1) the RCCL all-reduce of them.
Default: weak scaling, 32 cameras per GPU (at N = 1 the whole 32-camera configuration BASELINE.json quotes the metric
on); `--scaling strong --total-views 32` is BASELINE config 4.  With N > 1 the line also carries the strong-scaling
figures of configs 4 and 5 (`strong_scaling`), and always what the process group really was (`world_size_seen`,
`collective_backend`, `ranks`: every rank's device and uuid, gathered through the backend).
Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak


def _own_stream(stream):
    """The timed loops run ON the stream their captured step owns (graph.CapturedStep): called from another stream, every
    step is fenced against it on entry and exit -- two cross-stream events between consecutive replays, ~25 us per step
    (headline 1.770 -> 1.742 ms, the 8-view shard 0.576 -> 0.553).  D3M_BENCH_OWN_STREAM=0: the caller's stream, as before."""
    return torch.cuda.stream(stream) if os.environ.get("D3M_BENCH_OWN_STREAM", "1") == "1" else contextlib.nullcontext()


def library_identity():
    """(path relative to the repo, first 16 hex digits of the sha256) of the shared library this process timed."""
    import hashlib
    from deep3dmap_amd import _lib
    path = os.path.abspath(_lib.LIB_PATH)
    sha = hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    return (os.path.relpath(path, ROOT) if path.startswith(ROOT) else path), sha


def refuse_dev_switches(allow):
    """A bench line is a claim about the PRODUCT build with its own correctness checks on: the developer switches that
    swap the library (D3M_LIB_PATH: tuning variants, -DD3M_DEV_SKIP builds whose results are wrong by construction) or
    disable the asserts (D3M_BENCH_TIMING_EXPERIMENT) are refused unless --allow-dev is passed, and then named in the line."""
    used = [k for k in ("D3M_LIB_PATH", "D3M_BENCH_TIMING_EXPERIMENT") if os.environ.get(k)]
    if used and not allow:
        sys.exit(f"bench.py: {', '.join(used)} set in the environment -- developer switches; pass --allow-dev to time such a "
                 "build (the line then says so)")
    return used


def timed_repeats(run_steps, barrier, repeats, steps):
    """`repeats` timed regions of exactly `steps` steps each, every one bracketed by barrier + device synchronisation on
    both sides; returns the seconds of each region."""
    out = []
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = run_steps()
        barrier()
        out.append(time.perf_counter() - t0)
    return out, last


def algorithmic_bytes(V, F, S, s, ts, alpha=1, depth=1, rgb=1, tex_grad=1):
    """compulsory HBM bytes per view, forward and backward."""
    P, Po = S * S, s * s
    T = F * ts ** 3 * 12 if rgb else 0
    Tg = T if tex_grad else 0
    a_fwd = 12 * V + 12 * F + T + 20 * P + Po * (4 * alpha + 4 * depth + 12 * rgb)
    a_bwd = (Po * (4 * alpha + 4 * depth + 12 * rgb) + 20 * P + P * (4 * alpha + 12 * rgb) * (1 if (alpha or rgb) else 0)
             + 12 * V + 12 * F + 12 * V + Tg)
    return a_fwd, a_bwd


# Per-kernel shares of that byte count: a PARTITION of section 8(d) -- every term of A_fwd + A_bwd belongs to exactly one
# kernel of the step (the first kernel of its owner list that the step launches), so the shares of a step's kernels add up
# to algorithmic_bytes() (tests/test_host_logic.py holds that for both forms of coverage).  Up to round 5 this was a table of
# what each kernel "must touch", which counted the shared textures and the saved maps in several kernels: the shares summed
# to 2.5 GB against 1.56 GB per headline step, and the secondary kernels' fractions of the roofline were inflated by it.
# What the kernels move BEYOND their share -- crossing records, per-view texel gradients, the dense face copy -- is the
# formulation's own traffic: profiles/*pmc_traffic*.json has it.
def byte_terms(V, F, S, s, ts):
    P, Po, T = S * S, s * s, F * ts ** 3 * 12
    return {"fwd_vertices": 12 * V, "fwd_indices": 12 * F, "fwd_textures": T, "fwd_saved_maps": 20 * P,
            "fwd_outputs": 20 * Po,
            "bwd_output_grads_alpha_rgb": 16 * Po, "bwd_output_grad_depth": 4 * Po,
            "bwd_saved_face_index": 4 * P, "bwd_saved_weights_depth": 16 * P, "bwd_alpha_rgb_maps": 16 * P,
            "bwd_vertices": 12 * V, "bwd_indices": 12 * F, "bwd_grad_vertices": 12 * V, "bwd_grad_textures": T}


TERM_OWNERS = {
    "fwd_vertices": ["k_lit_front", "k_camera_forward", "k_bin_count", "k_bid_faces"],
    "fwd_indices": ["k_bin_count", "k_bid_faces", "k_gather_faces"],
    "fwd_textures": ["k_render_lit_fit_records", "k_render_lit_epilogue", "k_texture_sampling"],
    "fwd_saved_maps": ["k_raster_tiles", "k_bid_resolve"],
    "fwd_outputs": ["k_render_lit_fit_records", "k_render_lit_epilogue", "k_output_epilogue"],
    # the edge gradient's walk: the values (alpha 4 + rgb 12), the owner (4) and the gradients (alpha 4 + rgb 12) of every
    # pixel -- 36 B per pixel, in whatever form the step hands them over (maps, or the fused objective's per-pixel records)
    "bwd_output_grads_alpha_rgb": ["k_edge_lines"],
    "bwd_saved_face_index": ["k_edge_lines"],
    "bwd_alpha_rgb_maps": ["k_edge_lines"],
    "bwd_output_grad_depth": ["k_backward_textures_lit_faces", "k_backward_depth_faces", "k_backward_depth_map"],
    "bwd_saved_weights_depth": ["k_backward_textures_lit_faces", "k_backward_textures_faces", "k_backward_depth_faces",
                                "k_backward_textures"],
    "bwd_vertices": ["k_lit_back", "k_camera_backward"],
    "bwd_indices": ["k_edge_scatter", "k_edge_gather"],
    "bwd_grad_vertices": ["k_edge_gather", "k_scatter_face_grads"],
    "bwd_grad_textures": ["k_backward_textures_lit_faces", "k_backward_textures_faces", "k_backward_textures"],
}
# the headline step, per-tile lists | bidding
STEP_KERNELS = {
    "binned": ["k_lit_front", "k_bin_count", "k_bin_alloc", "k_bin_fill", "k_raster_tiles", "k_compact_visible",
               "k_edge_count", "k_alloc_plan", "k_edge_scatter", "k_render_lit_fit_records", "k_backward_textures_lit_faces",
               "k_lit_large_faces", "k_sum_over_views_ts2", "k_edge_lines", "k_edge_overflow", "k_edge_gather", "k_lit_back"],
    "bidding": ["k_lit_front", "k_bid_faces", "k_bid_big", "k_bid_resolve", "k_compact_visible", "k_edge_count",
                "k_alloc_plan", "k_edge_scatter", "k_render_lit_fit_records", "k_backward_textures_lit_faces",
                "k_lit_large_faces", "k_sum_over_views_ts2", "k_edge_lines", "k_edge_overflow", "k_edge_gather",
                "k_face_light_backward", "k_camera_backward"],
}


def byte_partition(step_kernels, V, F, S, ts, s=None):
    """{kernel: its share of section 8(d)'s bytes per view} over the kernels `step_kernels` of one step; the shares add
    up to algorithmic_bytes() when every term finds an owner among them (None is returned for the key "unowned" otherwise)."""
    terms = byte_terms(V, F, S, S if s is None else s, ts)
    shares = {k: 0 for k in step_kernels}
    for term, owners in TERM_OWNERS.items():
        owner = next((k for k in owners if k in shares), None)
        if owner is None:
            shares["unowned"] = shares.get("unowned", 0) + terms[term]
        else:
            shares[owner] += terms[term]
    return shares


def kernel_bytes(name, V, F, S, ts, step_kernels=None, s=None):
    """Kernel `name`'s share of section 8(d)'s bytes per view in a step that launches `step_kernels` (default: the headline
    step in the form of coverage that launches `name`); None for a kernel that owns no term."""
    if step_kernels is None:
        step_kernels = STEP_KERNELS["bidding" if name.startswith("k_bid_") else "binned"]
    share = byte_partition(list(step_kernels), V, F, S, ts, s).get(name)
    return share if share else None


def owed_bytes(V, F, S, s, ts, api):
    """The section-8(d) bytes the TIMED api really has to move, per view: `render_fit_loss` evaluates the objective where
    the pixel values are produced, so the output images (forward: written, 20 B per output pixel) and their gradients
    (backward: read, 20 B) never exist -- 8(d) counts both.  The drop-in api (render + loss on images) owes all of 8(d)."""
    a_fwd, a_bwd = algorithmic_bytes(V, F, S, s, ts)
    if api == "render_fit_loss":
        a_fwd -= s * s * 20
        a_bwd -= s * s * 20
    return a_fwd, a_bwd


def rocprof_average_us(kernel, pattern="*kernel_stats_final.csv"):
    """(average duration in us of `kernel` in the newest committed `rocprofv3 --kernel-trace --stats` summary of this
    command, file) or (None, None).  NOT measured by this run: the bench line's own duration is HIP events."""
    import csv
    f = _newest_profile(pattern)
    if not f:
        return None, None
    rows = [r for r in csv.DictReader(open(f)) if kernel in r["Name"]]
    calls = sum(int(r["Calls"]) for r in rows)
    if not calls:
        return None, None
    return sum(float(r["TotalDurationNs"]) for r in rows) / calls / 1e3, os.path.relpath(f, ROOT)


def self_launch(n_gpus, argv):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks as FRESH child processes
    through torch.distributed.run -- the very command the driver uses -- and relay rank 0's line (the children inherit
    stdout).  This parent has not touched the GPU (importing torch does not), and it never execs: a process that has
    initialised the GPU must not be replaced (the pool's rule), so the launch happens before anything else."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    print(f"[bench] --gpus {n_gpus} without WORLD_SIZE: launching {' '.join(cmd[1:])}", file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def rank_identity(rank, local_rank):
    """what THIS rank runs on, for the line's `ranks` list: proof of which devices the job's ranks really used"""
    dev = torch.cuda.current_device()
    props = torch.cuda.get_device_properties(dev)
    return {"rank": rank, "local_rank": local_rank, "device": dev, "name": props.name,
            "uuid": str(getattr(props, "uuid", "")), "pid": os.getpid()}


def _newest_profile(pattern):
    """the committed summary of the latest round (profiles/rNN_...), 'final' before anything else of that round"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)),
                   key=lambda f: (os.path.basename(f)[:3], "final" in os.path.basename(f), os.path.basename(f)))
    return files[-1] if files else None


def measured_traffic(name):
    """(HBM bytes per launch of `name`, file) from the newest committed PMC summary (profiles/*pmc_traffic*.json, made
    by profiles/pmc_traffic.py from two rocprofv3 --pmc passes of this same command), or (None, None).  NOT measured by
    this run: PMC counters need the profiler."""
    f = _newest_profile("*pmc_traffic*.json")
    if not f:
        return None, None
    k = json.load(open(f)).get("kernels", {}).get(name)
    return (k["hbm_bytes_per_launch"], os.path.relpath(f, ROOT)) if k else (None, None)


def measured_valu(name):
    """Wave-level VALU instructions per launch of `name` from the committed SQ counter summary
    (profiles/*sq_counters*.csv, tools_dev/sq_counters.sh), or None."""
    import csv
    f = _newest_profile("*sq_counters*.csv")
    if not f:
        return None, None
    for row in csv.DictReader(open(f)):
        if row["kernel"] == name and row.get("SQ_INSTS_VALU"):
            return int(float(row["SQ_INSTS_VALU"])), os.path.relpath(f, ROOT)
    return None, None


def cpu_baseline(n, image_size, ts, budget_s=25.0):
    """The oracle (this repo's C restatement of the reference algorithm, kind 'port') timed on the host cores on ONE
    view of the same workload, both variants of SURVEY.md 8(d):

      value (cpu_tiled)   K1 + bounding-box K2 (the loop structure of the reference's own CPU rasterizer,
                          mesh_cython/render.cpp:333-366) + K3 + K4 + K5 + K6, OpenMP over all host threads;
      bruteforce          K2 exactly as the reference's kernel (every pixel loops over every face, KCU:105-154),
                          timed on a bounded band of the view's pixels and scaled linearly to the whole view, + the same
                          K1, K3..K6; all threads and ONE thread.
    """
    from deep3dmap_amd import synthetic
    from oracle import nr_oracle as O
    v, tri = synthetic.grid_mesh(n)
    tex = synthetic.random_textures(tri.shape[0], ts)
    eye = synthetic.camera_ring(8)[1]
    S = image_size
    r = O.Renderer(camera_mode="look_at", image_size=S, anti_aliasing=False)
    r.eye = [float(x) for x in eye]
    vt, trit, text = torch.from_numpy(v)[None], torch.from_numpy(tri)[None].long(), torch.from_numpy(tex)[None]
    idx, lit = r._lit(vt, trit, text)
    faces = O.vertices_to_faces(r._camera(vt), idx).numpy()
    lit = lit.numpy()
    rng = np.random.default_rng(0)
    grads = [rng.standard_normal(sh).astype(np.float32) for sh in ((1, S, S, 3), (1, S, S), (1, S, S))]
    k = O.kernels("port")
    all_threads = O.num_threads()

    def rest_and_tiled():
        """(seconds of the tiled K1+K2, seconds of K3..K6) on the current thread count."""
        t0 = time.perf_counter()
        m = O.raster_forward(faces, None, S, r.near, r.far, r.rasterizer_eps, None, False, True, True, bbox=True)
        t_k2 = time.perf_counter() - t0
        t0 = time.perf_counter()
        m = O.raster_forward(faces, lit, S, r.near, r.far, r.rasterizer_eps, (0, 0, 0), True, True, True, bbox=True)
        O.raster_backward(m, *grads, True, True, True)
        return t_k2, time.perf_counter() - t0 - t_k2

    def brute_band(rows):
        """seconds of the brute-force K2 on `rows` rows in the middle of the view (covered region)."""
        finv = np.zeros_like(faces)
        k.face_inverse(O._fp(faces), O._fp(finv), 1, faces.shape[1], S)
        fi, wm = np.full((1, S, S), -1, np.int32), np.zeros((1, S, S, 3), np.float32)
        dm, fim = np.full((1, S, S), r.far, np.float32), np.zeros((1, S, S, 9), np.float32)
        r0 = (S - rows) // 2
        t0 = time.perf_counter()
        k.face_index_map_range(O._fp(faces), O._fp(finv), O._ip(fi), O._fp(wm), O._fp(dm), O._fp(fim), 1, faces.shape[1],
                               S, r.near, r.far, 1, r0 * S, (r0 + rows) * S)
        return time.perf_counter() - t0

    t_k2, t_rest = min(rest_and_tiled() for _ in range(2))
    tiled = t_k2 + t_rest
    rows_all = max(8, S // 8)
    t_brute_all = brute_band(rows_all) * (S / rows_all)
    O.set_num_threads(1)
    try:
        _, t_rest_1 = rest_and_tiled()
        rows_1 = max(2, S // 128)
        t_brute_1 = brute_band(rows_1) * (S / rows_1)
    finally:
        O.set_num_threads(all_threads)
    mpix = lambda t: round(S * S / t / 1e6, 5)
    return {"value": mpix(tiled), "unit": "Mpix/s", "cores": all_threads, "kind": "port", "variant": "cpu_tiled",
            "sample": f"1 view of the same mesh @ {S}x{S}, fwd (K1, bounding-box K2, K3) + bwd (K4, K5, K6), "
                      f"{tiled:.2f} s, OpenMP over {all_threads} threads",
            "bruteforce": {"value": mpix(t_brute_all + t_rest), "cores": all_threads,
                           "value_1thread": mpix(t_brute_1 + t_rest_1), "unit": "Mpix/s",
                           "sample": f"brute-force K2 (KCU:105-154) on {rows_all} of {S} rows with {all_threads} threads "
                                     f"and on {rows_1} rows with 1 thread, scaled linearly to the view "
                                     f"({t_brute_all:.1f} s / {t_brute_1:.0f} s per view) + the full view's K1, K3..K6 "
                                     f"({t_rest:.2f} s / {t_rest_1:.2f} s)"}}


def g2s_algorithmic_bytes(H, W, s, S):
    """Compulsory HBM bytes of the gan2shape renderer block per batch entry, in the manner of SURVEY.md 8(d): every
    input read once, every output written once, per pass direction; V = HW vertices, F = 2(H-1)(W-1) index-free
    triangles (the fill_back copy is virtual), P = S^2 raster pixels."""
    V, F, P, hw, ss = H * W, 2 * (H - 1) * (W - 1), S * S, H * W, s * s
    fwd = hw * (4 + 12) + hw * (12 + 4 + 12) + (12 * V + 12 * F + 20 * P) + ss * (12 + 4 + 12 + 4)
    #     depth, albedo    normal, shading, texture   mesh raster (8d A_fwd)        target, recon_depth, recon_im, mask
    bwd = ss * (12 + 12 + 4) + 20 * P + 4 * P + (12 * V + 12 * F + 12 * V) + hw * (4 + 12 + 12) + hw * (4 + 12)
    #     target, recon_im, mask   maps   depth-map gradient   8d A_bwd (depth mode)   re-read inputs   grad depth, albedo
    return fwd, bwd


def gan2shape_workload(args):
    """BASELINE config 3: the renderer block of the gan2shape training step (models/frameworks/gan2shape.py:444,463-497)
    at batch 16: view -> (R, t); depth [16,64,64] -> normals -> shading -> texture; the depth's 7,938-triangle grid mesh
    (fill_back, projection camera, 2x anti-aliasing) warped and rasterized to recon_depth; recon_im = texture looked up
    through the inverse warp; border mask; masked L1 against the input image + smooth losses of depth and shading; backward
    to depth, albedo, light and view.  NrRenderer.reconstruct = 8 + 5 launches, + 2 for the view.  Single GPU; one JSON line."""
    from deep3dmap_amd import _lib, synthetic
    from deep3dmap_amd.core import NrRenderer
    torch.cuda.set_device(0)
    b, hw = args.batch, 64
    cfgs = dict(min_depth=0.9, max_depth=1.1, rot_center_depth=1.0, fov=10, tex_cube_size=2)
    r = NrRenderer(cfgs, hw)
    depth, albedo, light, view, input_im = (torch.from_numpy(a).cuda() for a in synthetic.gan2shape_inputs(b, hw, 0, args.flip))
    light_a, light_b = light[:, :1] / 2 + 0.5, light[:, 1:2] / 2 + 0.5                  # gan2shape.py:457-461
    light_d = torch.cat([light[:, 2:], torch.ones(light.size(0), 1, device="cuda")], 1)
    light_d = light_d / ((light_d ** 2).sum(1, keepdim=True)) ** 0.5
    leaves = [t.detach().clone().requires_grad_(True) for t in (depth, albedo, light_a, light_b, light_d, view)]
    d, a, la, lb, ld, v = leaves
    one = torch.ones((), device="cuda")
    B = d.shape[0]

    def step():
        for x in leaves:
            x.grad = None
        r.set_transform_matrices(v)
        out = r.reconstruct(d, a, la, lb, ld, input_im, flip=args.flip, lam_smooth=0.01)
        torch.autograd.backward([out.loss_total], [one])
        return out.loss_total.detach()

    from deep3dmap_amd.graph import CapturedStep
    runner = CapturedStep(step)
    loss_eager = float(runner())
    g_eager = [x.grad.clone() for x in leaves]
    if not args.no_graph:
        runner.capture()
    with _own_stream(runner.stream):
        for _ in range(args.warmup):
            runner()
        regions, loss = timed_repeats(runner, torch.cuda.synchronize, max(1, args.repeats), args.steps)
        elapsed = float(np.median(regions))
    if not os.environ.get("D3M_BENCH_TIMING_EXPERIMENT"):      # (kernel-timing experiments with deliberately wrong results)
        assert abs(float(loss) - loss_eager) <= 1e-5 * abs(loss_eager), (float(loss), loss_eager)
        for x, g0 in zip(leaves, g_eager):
            assert float((x.grad - g0).abs().max()) <= 1e-3 * float(g0.abs().max()) + 1e-12
    runner.release()
    _lib.kernel_timing(True)
    n_inst = 5
    for _ in range(n_inst):
        runner()
    ktimes = _lib.collect_kernel_times()
    _lib.kernel_timing(False)
    S = 2 * hw
    step_s = elapsed / args.steps
    a_fwd, a_bwd = g2s_algorithmic_bytes(hw, hw, hw, S)
    step_bytes = (a_fwd + a_bwd) * B
    dom = max(ktimes, key=lambda k: ktimes[k][1])
    dom_avg_s = ktimes[dom][1] / ktimes[dom][0] / 1e3
    V, F = hw * hw, 2 * (hw - 1) * (hw - 1)
    # algorithmic bytes per batch entry of the passes that can dominate: screen vertices in, z-buffer (8 B / raster
    # pixel) out; vertices + z-buffer + depth-map gradient in, nine sums per triangle out; ...
    dom_bytes = {"k_g2s_raster": 12 * V + 8 * S * S,
                 "k_g2s_depth_faces": 12 * V + 12 * S * S + 36 * F,
                 "k_g2s_sample_backward": hw * hw * (12 + 12 + 4 + 12) + 12 * S * S + 12 * V,
                 "k_raster_tiles": 12 * V + 12 * F + 20 * S * S,
                 "k_backward_depth_map": 24 * S * S + 12 * V + 12 * F + 12 * V}.get(dom)
    roof = None
    if dom_bytes is not None:
        ach = dom_bytes * B / dom_avg_s / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None, "avg_launch_us": round(dom_avg_s * 1e6, 2),
                "algorithmic_bytes_per_launch": dom_bytes * B,
                "duration_source": "HIP events around every launch of an eager pass of the same step"}
    print(json.dumps({
        "metric": f"rendered Mpix/s fwd+bwd, gan2shape renderer block (64x64 depth-to-mesh, batch {B})",
        "value": round(B * hw * hw / step_s / 1e6, 2), "unit": "Mpix/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "repeats": len(regions), "ms_per_step_min": round(min(regions) / args.steps * 1e3, 4),
        "ms_per_step_max": round(max(regions) / args.steps * 1e3, 4),
        "library": library_identity()[0], "library_sha16": library_identity()[1], "dev_switches": args.dev_switches,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"gan2shape renderer block (gan2shape.py:444,463-497): view [{B},6] -> (R,t); depth [{B},64,64] -> "
                               "normals -> shading -> texture; 7938-tri implicit grid mesh x2 (fill_back) -> recon_depth @64 "
                               "with 2x AA (S=128); recon_im = grid lookup of the texture through the inverse warp; border "
                               "mask; masked L1 + smooth losses; backward to depth, albedo, light, view",
                   "api": "NrRenderer.set_transform_matrices + NrRenderer.reconstruct + backward", "batch": B,
                   "flip": bool(args.flip), "hip_graph": not args.no_graph},
        "launches_per_step": sum(c for c, _ in ktimes.values()) / n_inst,
        "hbm_roofline_frac_step": round(step_bytes / step_s / 8e12, 5),
        "algorithmic_bytes_per_step": step_bytes,
        "d3m_kernel_ms_per_step": round(sum(m for _, m in ktimes.values()) / n_inst, 4),
        "kernel_ms_per_step": {k: round(m / n_inst, 4) for k, (c, m) in sorted(ktimes.items(), key=lambda kv: -kv[1][1])},
        "roofline": roof}))


def modes_workload(args):
    """SURVEY 8(d), "modes timed": the silhouette-only and depth-only modes of the renderer (NR/renderer.py:114-183:
    render_silhouettes / render_depth) on the headline mesh and cameras -- look_at transform, fill_back gather, coverage, the
    output epilogue, the mode's loss against a target rendered from a perturbed mesh (silhouettes: sum((a - a_t)^2) / P,
    NR/examples/example2.py:43-47; depth: photometric_loss with the target's silhouette as mask), backward to the vertices
    (K4 for the silhouette, K6 for the depth).  Single GPU; one JSON line; no objective is fused into these modes: they run
    the reference's operator sequence through rasterize_ops (include/d3m_raster.h section A)."""
    from deep3dmap_amd import _lib, neural_renderer as nr, synthetic
    from deep3dmap_amd.core.losses import photometric_loss, silhouette_loss
    from deep3dmap_amd.graph import CapturedStep
    torch.cuda.set_device(0)
    mode, B, s = args.workload, args.views_per_gpu, args.image_size
    v_np, tri_np = synthetic.grid_mesh(args.mesh_n)
    r = nr.Renderer(image_size=s, anti_aliasing=args.anti_aliasing, camera_mode="look_at", fill_back=True)
    r.eye = torch.from_numpy(synthetic.camera_ring(B)).float().cuda()
    r.mesh_modes = os.environ.get("D3M_MESH_MODES", "1") != "0"      # (0: the reference's operator sequence, for A/B)
    tri = torch.from_numpy(tri_np).int().cuda()[None]
    render = r.render_silhouettes if mode == "silhouettes" else r.render_depth
    with torch.no_grad():
        moved = torch.from_numpy(synthetic.perturb(v_np, 0.02)).float().cuda()[None]
        target = render(moved, tri)
        mask = r.render_silhouettes(moved, tri) if mode == "depth" else None
    v = torch.from_numpy(v_np).float().cuda()[None].requires_grad_(True)
    one = torch.ones((), device="cuda")
    P = float(s * s)

    def step():
        v.grad = None
        image = render(v, tri)
        loss = silhouette_loss(image, target) / P if mode == "silhouettes" else photometric_loss(image, target, mask)
        torch.autograd.backward([loss], [one])
        return loss.detach()

    runner = CapturedStep(step)
    loss_eager = float(runner())
    g_eager = v.grad.clone()
    assert float(g_eager.abs().max()) > 0
    if not args.no_graph:
        runner.capture()
    with _own_stream(runner.stream):
        for _ in range(args.warmup):
            runner()
        regions, loss = timed_repeats(runner, torch.cuda.synchronize, max(1, args.repeats), args.steps)
        elapsed = float(np.median(regions))
    assert abs(float(loss) - loss_eager) <= 1e-5 * abs(loss_eager), (float(loss), loss_eager)
    assert float((v.grad - g_eager).abs().max()) <= 1e-3 * float(g_eager.abs().max())
    runner.release()
    _lib.kernel_timing(True)
    n_inst = 3
    for _ in range(n_inst):
        runner()
    ktimes = _lib.collect_kernel_times()
    _lib.kernel_timing(False)
    step_s = elapsed / args.steps
    V, F = v_np.shape[0], tri_np.shape[0]
    S = 2 * s if args.anti_aliasing else s
    step_bytes = B * sum(algorithmic_bytes(V, F, S, s, args.texture_size, alpha=int(mode == "silhouettes"),
                                           depth=int(mode == "depth"), rgb=0, tex_grad=0))
    print(json.dumps({
        "metric": f"rendered Mpix/s fwd+bwd, {mode} mode",
        "value": round(B * s * s / step_s / 1e6, 2), "unit": "Mpix/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "repeats": len(regions), "ms_per_step_min": round(min(regions) / args.steps * 1e3, 4),
        "ms_per_step_max": round(max(regions) / args.steps * 1e3, 4),
        "library": library_identity()[0], "library_sha16": library_identity()[1], "dev_switches": args.dev_switches,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{F}-triangle mesh @{s}x{s}{' with anti-aliasing' if args.anti_aliasing else ''}, {B} views, "
                               f"{mode} only: Renderer.render_{mode} + loss + backward (vertex gradient)",
                   "api": f"Renderer.render_{mode} + {'silhouette_loss' if mode == 'silhouettes' else 'photometric_loss'} + backward",
                   "views": B, "hip_graph": not args.no_graph, "one_node_over_the_indexed_mesh": bool(r.mesh_modes)},
        "launches_per_step": sum(c for c, _ in ktimes.values()) / n_inst,
        "hbm_roofline_frac_step": round(step_bytes / step_s / 8e12, 5), "algorithmic_bytes_per_step": step_bytes,
        "d3m_kernel_ms_per_step": round(sum(m for _, m in ktimes.values()) / n_inst, 4),
        "kernel_ms_per_step": {k: round(m / n_inst, 4) for k, (c, m) in sorted(ktimes.items(), key=lambda kv: -kv[1][1])}}))


def mesh_family_workload(args):
    """the face3d utility rasterizer (mesh_cython) at the size its caller uses (tools/data_gen/prnet.py:
    a BFM-sized mesh, 52,900 vertices / 104,882 triangles, 256x256): render_colors + get_triangle_buffer +
    vis_of_vertices per step, data resident on the device.  The CPU figure beside it is the reference's own C++
    (oracle/_ref/libmesh_ref.so, compiled from its sources in the build container; one thread, as the reference runs
    it) where that file travelled, the C port otherwise."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mesh_scenes
    from deep3dmap_amd import _lib
    from deep3dmap_amd.mesh_cython import render
    from oracle import mesh_oracle as M
    torch.cuda.set_device(0)
    h = w = 256
    s = mesh_scenes.grid_scene(230, h, w, 21, jitter=0.25)
    v, t, col = (torch.from_numpy(s[k]).cuda() for k in ("vertices", "triangles", "colors"))

    def step():
        return render.render_colors(v, t, col, h, w), render.get_triangle_buffer(v, t, h, w), render.vis_of_vertices(v, t, h, w)

    img, tb, vis = step()
    be = M.default_backend()
    t0 = time.perf_counter()
    ref = (M.render_colors(s["vertices"], s["triangles"], s["colors"], h, w, backend=be),
           M.get_triangle_buffer(s["vertices"], s["triangles"], h, w, backend=be),
           M.vis_of_vertices(s["vertices"], s["triangles"], h, w, backend=be))
    cpu_s = time.perf_counter() - t0
    assert all(np.array_equal(a.cpu().numpy(), b) for a, b in zip((img, tb, vis), ref)), "GPU result differs from the oracle"
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = (time.perf_counter() - t0) / args.steps
    _lib.kernel_timing(True)
    for _ in range(3):
        step()
    ktimes = _lib.collect_kernel_times()
    _lib.kernel_timing(False)
    print(json.dumps({
        "metric": "rendered Mpix/s, face3d mesh_cython family (render_colors + triangle buffer + vertex visibility)",
        "value": round(3 * h * w / elapsed / 1e6, 2), "unit": "Mpix/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"mesh_cython family on a {s['vertices'].shape[1]}-vertex / {s['triangles'].shape[1]}-triangle "
                               f"mesh @ {h}x{w}: render_colors + get_triangle_buffer + vis_of_vertices, bit-identical to the oracle"},
        "kernel_ms_per_step": {k: round(m / 3, 4) for k, (c, m) in sorted(ktimes.items(), key=lambda kv: -kv[1][1])[:8]},
        "cpu_baseline": {"value": round(3 * h * w / cpu_s / 1e6, 3), "unit": "Mpix/s", "cores": 1,
                         "kind": "reference" if be == "ref" else "port",
                         "sample": f"the same three calls once ({cpu_s:.2f} s), single thread"}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--views-per-gpu", type=int, default=32,
                    help="cameras per GPU (weak scaling); 32 = the camera count of BASELINE.json's 100k-triangle config")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --views-per-gpu cameras on every GPU; strong: --total-views cameras in all, "
                         "split across the GPUs (BASELINE config 4: 32 cameras, 8 per GPU on 4)")
    ap.add_argument("--total-views", type=int, default=32, help="strong scaling: cameras of the whole job")
    ap.add_argument("--no-strong-lines", dest="strong_lines", action="store_false",
                    help="N > 1, headline mesh: do not add the strong-scaling figures of BASELINE configs 4 and 5 "
                         "(`strong_scaling` in the line)")
    ap.add_argument("--mesh-n", type=int, default=225, help="grid_mesh(n): 225 -> 100,352 triangles")
    ap.add_argument("--image-size", type=int, default=512)
    ap.add_argument("--texture-size", type=int, default=2)
    ap.add_argument("--anti-aliasing", action="store_true",
                    help="render at twice the size and pool (SURVEY config C2 is AA-on); the headline configuration is AA-off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true",
                    help="skip the second timed pass (the same step through Renderer.render + multiview_fit_loss)")
    ap.add_argument("--materialise-images", action="store_true",
                    help="time what an UNMODIFIED caller of the reference's surface runs: Renderer.render() with nothing "
                         "registered on the renderer, the objective composed from the loss operators on the images "
                         "(--loss-form torch: from plain torch operators), gradient images back through the epilogue's "
                         "adjoint")
    ap.add_argument("--loss-form", default="operators", choices=["operators", "torch"],
                    help="--materialise-images: photometric_loss / silhouette_loss of deep3dmap_amd.core.losses, or the same "
                         "formulas written in eager torch")
    ap.add_argument("--strong-shrink", type=float, default=1.0,
                    help="N > 1: scale the meshes and rasters of the strong-scaling lines (BASELINE configs 4 and 5) by "
                         "this factor, camera counts unchanged, and run them whatever --mesh-n / --image-size are -- a "
                         "pre-flight of the multi-rank line's code path on a small box (tests/test_gpu_multirank.py)")
    ap.add_argument("--fit-with-images", action="store_true",
                    help="the fused objective, and the same pass also writes the output images (Renderer.render_fit_loss "
                         "images_out): what keeping the images costs on top of the headline line")
    ap.add_argument("--view-groups", type=int, default=1,
                    help="run each GPU's views as this many concurrent pipelines inside the rendering node")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a HIP graph")
    ap.add_argument("--workload", default="multiview", choices=["multiview", "gan2shape", "mesh_family", "silhouettes", "depth"],
                    help="multiview = the headline metric (default); gan2shape = BASELINE config 3 (secondary line)")
    ap.add_argument("--batch", type=int, default=16, help="gan2shape workload: batch size")
    ap.add_argument("--flip", action="store_true",
                    help="gan2shape workload: append the mirrored copies (flip3, gan2shape.py:431: 2 x batch entries)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of --steps steps each (barrier-bracketed); value / ms_per_step = their MEDIAN, "
                         "min and max ride in the line (box-to-box and run-to-run spread is 1-2 %%)")
    ap.add_argument("--allow-dev", action="store_true",
                    help="accept D3M_LIB_PATH / D3M_BENCH_TIMING_EXPERIMENT (developer builds and timing experiments)")
    ap.add_argument("--coverage-form", default="auto", choices=["auto", "binned", "bidding"],
                    help="force one form of the forward's coverage (d3m_set_coverage_form; measurements)")
    args = ap.parse_args()
    t_start = time.perf_counter()

    def stage(what):
        """wall time of the line's stages on stderr (rank 0): a run that hits the driver's limit says where it was"""
        if int(os.environ.get("RANK", "0")) == 0:
            print(f"[bench] +{time.perf_counter() - t_start:7.1f} s  {what}", file=sys.stderr, flush=True)
    dev_switches = args.dev_switches = refuse_dev_switches(args.allow_dev)
    if args.coverage_form != "auto":
        from deep3dmap_amd import _lib as _l
        _l.check(_l.lib().d3m_set_coverage_form({"binned": 0, "bidding": 1}[args.coverage_form]), "d3m_set_coverage_form")
        dev_switches = args.dev_switches = list(dev_switches) + [f"--coverage-form {args.coverage_form}"]
    if args.workload == "gan2shape":
        return gan2shape_workload(args)
    if args.workload == "mesh_family":
        return mesh_family_workload(args)
    if args.workload in ("silhouettes", "depth"):
        return modes_workload(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the bare form: start the ranks ourselves (before this process has touched the GPU) and relay rank 0's line
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch N ranks with torch.distributed.run, or call "
                 f"`python bench.py --gpus N` without a torchrun environment and it launches them itself)")
    # D3M_BENCH_FORCE_DIST=1 (debug, not used by the driver): initialise the process group and run every collective even
    # with ONE rank -- the RCCL calls of the N > 1 path (communicator with device_id, the step's all-reduce on the flat device
    # buffer, barriers, the MAX over ranks) execute on a single-GPU box (tests/test_gpu_multirank.py)
    dist_on = world > 1 or os.environ.get("D3M_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # Debug overrides (not used by the driver): exercise the N>1 code path on a single-GPU box by putting every
    # rank on device 0 and reducing over gloo instead of RCCL.
    if os.environ.get("D3M_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    elif world > 1 and torch.cuda.device_count() < world:
        sys.exit(f"bench.py: {world} ranks but {torch.cuda.device_count()} visible GPU(s) -- one process per GPU")
    torch.cuda.set_device(local_rank)
    backend = None
    if dist_on:
        backend = os.environ.get("D3M_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    # what the process group really is: every rank's device, gathered THROUGH the collective backend (outside any timed region)
    ranks_seen = [rank_identity(rank, local_rank)]
    world_seen, backend_seen = 1, "none (one process, no process group)"
    if dist_on:
        world_seen = dist.get_world_size()
        backend_seen = dist.get_backend()
        if backend_seen == "nccl":
            backend_seen = "nccl (RCCL " + ".".join(str(x) for x in torch.cuda.nccl.version()) + ")"
        ranks_seen = [None] * world_seen
        dist.all_gather_object(ranks_seen, rank_identity(rank, local_rank))
        assert world_seen == world and sorted(r["rank"] for r in ranks_seen) == list(range(world)), ranks_seen
    stage(f"process group up: {world_seen} rank(s), backend {backend_seen}")

    from deep3dmap_amd import _lib, synthetic, multiview
    from deep3dmap_amd.multiview import MultiViewFit, shard_views
    multiview.COLLECTIVES_WITH_ONE_RANK = dist_on and world == 1
    dev = f"cuda:{local_rank}"

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if not dist_on:
            return list(seconds)
        tmax = torch.tensor(seconds, device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return [float(x) for x in tmax.tolist()]

    def measure(mesh_n, image_size, n_views, steps, warmup, repeats, objective_in_renderer=True, share=None, want_graph=True,
                loss_form="linked"):
        """One camera-sharded fit of `n_views` cameras in all: eager step, capture, warm-up, `repeats` barrier-bracketed
        regions of `steps` steps (MAX over ranks, median).  Returns (fit, results dict); the fit is left in eager mode."""
        v, tri = synthetic.grid_mesh(mesh_n)
        tex = synthetic.random_textures(tri.shape[0], args.texture_size)
        eyes = synthetic.camera_ring(n_views)
        fit = MultiViewFit(v, tri, tex, eyes, image_size=image_size, anti_aliasing=args.anti_aliasing, rank=rank,
                           world_size=world, device=dev, objective_in_renderer=objective_in_renderer,
                           view_groups=args.view_groups, loss_form=loss_form)
        if share is not None:       # the same targets as another fit of the same shard (the drop-in pass)
            fit.targets, fit.mask_sum, fit._mask_sum_local = share.targets, share.mask_sum, share._mask_sum_local
        else:
            fit.set_targets_from(synthetic.perturb(v))
        fit.keep_images = args.fit_with_images and objective_in_renderer
        loss_eager, gv_eager, _ = fit.step()
        gv_eager = gv_eager.clone()
        graph_on = want_graph
        if graph_on:
            try:
                fit.capture_graph()
            except Exception as e:      # keep the benchmark alive: fall back to eager launches (still all-HIP kernels)
                print(f"[bench] HIP graph capture failed on rank {rank} ({type(e).__name__}: {e}); running eagerly",
                      file=sys.stderr)
                fit.release_graph()
                graph_on = False
                torch.cuda.synchronize()
        with _own_stream(fit.stream):
            for _ in range(warmup):
                fit.step()
            regions, (loss, gv, gt) = timed_repeats(fit.step, barrier, max(1, repeats), steps)
        regions = max_over_ranks(regions)                   # every region: the slowest rank's time
        if not os.environ.get("D3M_BENCH_TIMING_EXPERIMENT"):      # (kernel-timing experiments with deliberately wrong results)
            assert torch.isfinite(loss).item() and torch.isfinite(gv).all().item()
            # the replayed graph must reproduce the eager step (same inputs every step: no optimiser in the loop)
            rel = float(torch.linalg.norm(gv - gv_eager) / (torch.linalg.norm(gv_eager) + 1e-20))
            assert rel < 1e-3 and abs(float(loss) - float(loss_eager)) <= 1e-4 * abs(float(loss_eager)) + 1e-7, \
                (rel, loss, loss_eager)
        fit.release_graph()
        el = float(np.median(regions)) / steps
        shards = [shard_views(n_views, r, world) for r in range(world)]
        return fit, {"seconds_per_step": el, "regions": regions, "graph_on": graph_on, "n_views": n_views,
                     "views_per_rank": [hi - lo for lo, hi in shards], "V": int(v.shape[0]), "F": int(tri.shape[0]),
                     "loss_eager": float(loss_eager), "gv_eager": gv_eager,
                     "value": n_views * image_size ** 2 / el / 1e6,
                     "ms_per_step": el * 1e3, "ms_per_step_min": min(regions) / steps * 1e3,
                     "ms_per_step_max": max(regions) / steps * 1e3, "split_exchange": fit.split_exchange}

    if args.scaling == "strong":
        if args.total_views < world:
            sys.exit("--total-views: every rank renders at least one camera")
        n_views = args.total_views
        args.views_per_gpu = shard_views(n_views, 0, world)[1]          # the largest shard (rank 0's)
    else:
        n_views = args.views_per_gpu * world
    fit, res = measure(args.mesh_n, args.image_size, n_views, args.steps, args.warmup, args.repeats,
                       objective_in_renderer=not args.materialise_images, want_graph=not args.no_graph,
                       loss_form=args.loss_form if args.materialise_images else "linked")
    graph_on, regions, elapsed_step = res["graph_on"], res["regions"], res["seconds_per_step"]
    stage(f"headline measured: {n_views} views, {res['ms_per_step']:.3f} ms per step")

    # instrumented pass (not part of `value`), eager: per-kernel HIP-event durations on the launch stream
    _lib.kernel_timing(True)
    n_inst = max(3, min(args.steps, 10))
    for _ in range(n_inst):
        fit.step()
    ktimes = _lib.collect_kernel_times()
    _lib.kernel_timing(False)
    # ... and once more with every kernel of the step on ONE stream (D3M_SERIAL_BRANCHES=1, read per call): inside the step the
    # render node's branches share the chip -- the line walk runs beside the gathered texture pass, the plan beside the
    # sampling pass -- so a kernel's duration there is not the kernel's own; the roofline object is about the kernel
    saved_env = os.environ.get("D3M_SERIAL_BRANCHES")
    os.environ["D3M_SERIAL_BRANCHES"] = "1"
    try:
        _lib.kernel_timing(True)
        for _ in range(n_inst):
            fit.step()
        ktimes_alone = _lib.collect_kernel_times()
    finally:
        _lib.kernel_timing(False)
        if saved_env is None:
            del os.environ["D3M_SERIAL_BRANCHES"]
        else:
            os.environ["D3M_SERIAL_BRANCHES"] = saved_env

    # The same step through the reference's own surface (not part of `value`): Renderer.render() materialises the output
    # images, multiview_fit_loss is evaluated on them, their gradients come back through the epilogue's adjoint.
    dropin = None
    if not args.materialise_images and not args.no_dropin:
        fit2, r2 = measure(args.mesh_n, args.image_size, n_views, args.steps, args.warmup, args.repeats,
                           objective_in_renderer=False, share=fit, want_graph=graph_on)
        rel2 = float(torch.linalg.norm(r2["gv_eager"] - res["gv_eager"]) / (torch.linalg.norm(res["gv_eager"]) + 1e-20))
        assert rel2 < 1e-3 and abs(r2["loss_eager"] - res["loss_eager"]) <= 1e-4 * abs(res["loss_eager"]), (rel2, r2["loss_eager"])
        dropin = {"api": "Renderer.fit_targets registered; Renderer.render + multiview_fit_loss + backward",
                  "value": round(r2["value"], 2), "unit": "Mpix/s",
                  "ms_per_step": round(r2["ms_per_step"], 4), "ms_per_step_min": round(r2["ms_per_step_min"], 4),
                  "ms_per_step_max": round(r2["ms_per_step_max"], 4),
                  "over_fused": round(r2["ms_per_step"] / res["ms_per_step"], 4)}
        del fit2, r2
        stage("drop-in form (registered objective) measured")
        # ... and what a caller gets who changes NOTHING: render() with nothing registered, deep3dmap's own losses on the
        # images (the package's operators for them; then the same formulas in eager torch), gradient images back through
        # the output epilogue's adjoint (d3m_output_epilogue_backward_records)
        for form in ("operators", "torch"):
            fit3, r3 = measure(args.mesh_n, args.image_size, n_views, args.steps, args.warmup, args.repeats,
                               objective_in_renderer=False, share=fit, want_graph=graph_on, loss_form=form)
            rel3 = float(torch.linalg.norm(r3["gv_eager"] - res["gv_eager"]) / (torch.linalg.norm(res["gv_eager"]) + 1e-20))
            assert rel3 < 1e-3 and abs(r3["loss_eager"] - res["loss_eager"]) <= 1e-4 * abs(res["loss_eager"]), (form, rel3, r3["loss_eager"])
            dropin["generic_" + form] = {
                "api": "Renderer.render (nothing registered) + " +
                       ("photometric_loss / silhouette_loss operators" if form == "operators" else "the losses in eager torch ops") +
                       " + backward",
                "value": round(r3["value"], 2), "unit": "Mpix/s", "ms_per_step": round(r3["ms_per_step"], 4),
                "ms_per_step_min": round(r3["ms_per_step_min"], 4), "ms_per_step_max": round(r3["ms_per_step_max"], 4),
                "over_fused": round(r3["ms_per_step"] / res["ms_per_step"], 4)}
            del fit3, r3
        stage("generic drop-in forms measured")
    split_exchange = res["split_exchange"]
    del fit
    torch.cuda.empty_cache()

    # N > 1: the STRONG-scaling lines of BASELINE.json's sharded configurations ride in the same JSON line (the contract is
    # one line per run): config 4 = 32 cameras in all of the 100k mesh @512 (when this run is not that already), config 5 =
    # 256 cameras of the 1M-triangle mesh @1024.  Fewer steps; per-GPU shards above 64 views of config 5 are not run (a
    # 128-view shard holds ~75 GB of per-view scratch: a memory sweep is not what a bench run is for).
    strong = None
    shrink = min(1.0, max(0.01, args.strong_shrink))
    if world > 1 and args.strong_lines and ((args.mesh_n, args.image_size) == (225, 512) or shrink < 1.0):
        strong = {}
        small = lambda n, lo, step: max(lo, int(round(n * shrink / step)) * step)
        todo = [("config4", small(225, 6, 1), small(512, 32, 8), 32, args.steps, args.warmup, args.repeats),
                ("config5", small(709, 8, 1), small(1024, 32, 8), 256, max(2, args.steps // 4), 2, max(1, args.repeats // 2))]
        for name, mesh_n, size, total, k, w, reps in todo:
            per = shard_views(total, 0, world)[1]
            if args.scaling == "strong" and (mesh_n, size, total) == (args.mesh_n, args.image_size, n_views):
                strong[name] = {"same_as": "this line"}
                continue
            if per > 64:
                strong[name] = {"skipped": f"{per} cameras per GPU at {world} GPUs (run where a shard is <= 64 cameras)"}
                continue
            try:
                stage(f"strong-scaling line {name}: grid_mesh({mesh_n}) @{size}, {total} cameras, {per} per GPU ...")
                f3, r3 = measure(mesh_n, size, total, k, w, reps)
                stage(f"... {name} measured: {r3['ms_per_step']:.3f} ms per step")
                strong[name] = {"scaling": "strong", "total_views": total, "views_per_gpu": r3["views_per_rank"],
                                "triangles": r3["F"], "image_size": size, "steps": k, "warmup": w,
                                "value": round(r3["value"], 2), "unit": "Mpix/s", "ms_per_step": round(r3["ms_per_step"], 4),
                                "ms_per_step_min": round(r3["ms_per_step_min"], 4),
                                "ms_per_step_max": round(r3["ms_per_step_max"], 4), "split_exchange": r3["split_exchange"]}
                del f3, r3
                torch.cuda.empty_cache()
            except Exception as e:          # never lose the headline line to a secondary one
                strong[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
                stage(f"... {name} FAILED: {type(e).__name__}")
            if shrink < 1.0 and name in strong:
                strong[name]["shrunk_by"] = shrink

    if rank == 0:
        lib_path, lib_sha = library_identity()
        V, F, S, ts = res["V"], res["F"], args.image_size, args.texture_size
        ms_per_step = elapsed_step * 1e3
        value = res["value"]
        Si = 2 * S if args.anti_aliasing else S                       # internal raster size
        api = (f"render+loss ({args.loss_form}, nothing registered)" if args.materialise_images else
               ("render_fit_loss+images_out" if args.fit_with_images else "render_fit_loss"))
        a_fwd, a_bwd = algorithmic_bytes(V, F, Si, S, ts)
        o_fwd, o_bwd = owed_bytes(V, F, Si, S, ts, api)
        # the whole job's bytes over the whole job's peak (N GPUs): per GPU when the shards are equal
        step_bytes = (a_fwd + a_bwd) * n_views
        step_owed = (o_fwd + o_bwd) * n_views
        # dominant kernel = largest summed duration in the instrumented pass
        per_kernel = {k: (c, ms) for k, (c, ms) in ktimes.items()}
        dom = max(per_kernel, key=lambda k: per_kernel[k][1])
        dom_count, dom_ms = per_kernel[dom]
        dom_step_s = dom_ms / dom_count / 1e3                 # inside the step: beside the other branch's kernels
        a_count, a_ms = ktimes_alone.get(dom, (dom_count, dom_ms))
        dom_avg_s = a_ms / a_count / 1e3                      # the kernel by itself (one stream)
        kb = kernel_bytes(dom, V, F, Si, ts, step_kernels=list(per_kernel), s=S)
        roof = None
        if kb is not None:
            ach = kb * args.views_per_gpu / dom_avg_s / 1e9
            traffic, traffic_src = measured_traffic(dom)
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "traffic_source": traffic_src,     # a committed profiler summary of this command, not this run
                    "avg_launch_us": round(dom_avg_s * 1e6, 2), "launches_per_step": dom_count / n_inst,
                    "duration_source": "HIP events around every launch of an eager pass of the same step with its kernels "
                                       "on ONE stream (D3M_SERIAL_BRANCHES=1), after the timed region (a replayed graph "
                                       "cannot carry events): the kernel by itself.  `in_step`: the same from the eager "
                                       "pass as timed, where the render node's branches share the chip",
                    "algorithmic_bytes_per_launch": kb * args.views_per_gpu,
                    "in_step": {"avg_launch_us": round(dom_step_s * 1e6, 2),
                                "achieved": round(kb * args.views_per_gpu / dom_step_s / 1e9, 2),
                                "frac": round(kb * args.views_per_gpu / dom_step_s / 1e9 / HBM_PEAK_GBS, 5)}}
            # the same fractions from the COMMITTED rocprofv3 averages of this kernel (profiles/: graph replays of this very
            # command, with D3M_SERIAL_BRANCHES=1 and as it is), so that the line and profiles/ can be compared without a
            # footnote; not measured by this run
            if (args.mesh_n, S, args.views_per_gpu) == (225, 512, 32):
                rp_us, rp_src = rocprof_average_us(dom, "*kernel_stats_one_stream.csv")
                if rp_us:
                    roof["avg_launch_us_rocprof"] = round(rp_us, 2)
                    roof["frac_rocprof"] = round(kb * args.views_per_gpu / (rp_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
                    roof["rocprof_source"] = rp_src
                rp_us, rp_src = rocprof_average_us(dom)
                if rp_us:
                    roof["in_step"].update({"avg_launch_us_rocprof": round(rp_us, 2), "rocprof_source": rp_src,
                                            "frac_rocprof": round(kb * args.views_per_gpu / (rp_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)})
            # The kernel is not bandwidth-bound: what fraction of its duration the counted VALU
            # instructions need on 256 CUs x 4 SIMD-32 at 2.4 GHz, TWO cycles per wave64 instruction (MI355X_MICROARCH.md,
            # "Wave scheduling": a SIMD takes a wave's 64 lanes over 2 cycles, one wave alone issues every 4; packed-f32
            # and transcendental instructions cost twice that and are counted once here).  Informational.
            valu, valu_src = measured_valu(dom)
            if valu:
                roof["valu_wave_instructions"] = valu
                roof["valu_source"] = valu_src
                roof["valu_issue_frac"] = round(valu * 2 / (1024 * 2.4e9) / dom_avg_s, 3)
        n_launch = sum(c for c, _ in per_kernel.values()) / n_inst
        out = {
            "metric": ("rendered Mpix/s fwd+bwd, 100k-tri mesh @512x512" if (args.mesh_n, S) == (225, 512) else
                       f"rendered Mpix/s fwd+bwd, {F}-tri mesh @{S}x{S}"), "value": round(value, 2), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            # value / ms_per_step: the MEDIAN of `repeats` barrier-bracketed regions of `steps` steps each
            "repeats": len(regions), "ms_per_step_min": round(res["ms_per_step_min"], 4),
            "ms_per_step_max": round(res["ms_per_step_max"], 4),
            "ms_per_step_repeats": [round(r / args.steps * 1e3, 4) for r in regions],
            "library": lib_path, "library_sha16": lib_sha, "dev_switches": dev_switches,
            # every D3M_* variable of the environment the timed step ran under (tuning switches: INTEGRATION.md "Switches")
            "env_switches": {k: v for k, v in sorted(os.environ.items()) if k.startswith("D3M_") and not k.startswith("D3M_BENCH_")},
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"multi-view fit: grid_mesh({args.mesh_n}) = {F} triangles / {V} vertices, "
                                   + (f"{args.views_per_gpu} look_at cameras per GPU" if args.scaling == "weak" else
                                      f"{n_views} look_at cameras in all, {res['views_per_rank']} per GPU")
                                   + f" @ {S}x{S}, render(rgb+depth+alpha) "
                                   f"+ photometric/silhouette/depth loss + backward (vertex+texture grads)"
                                   + ("; output images materialised, objective evaluated on them"
                                      if args.materialise_images else
                                      "; the objective is evaluated in the pass that produces the pixel values "
                                      "(MultiViewFit.fit_loss), the rendered images stay in the internal HWC maps")
                                   + (", RCCL all-reduce of grads" if world > 1 else ""),
                       "api": api,
                       "views_per_gpu": args.views_per_gpu, "views_per_rank": res["views_per_rank"], "total_views": n_views,
                       "triangles": int(F), "image_size": S,
                       "texture_size": ts, "fill_back": True, "anti_aliasing": bool(args.anti_aliasing), "hip_graph": graph_on, "view_groups": args.view_groups, "objective_in_renderer": not args.materialise_images,
                       "parallelism": f"camera-sharded x{world}",
                       "exchange": ("none (one rank)" if not dist_on else
                                    "two all-reduces per step: texture gradient (started between the step's two HIP graphs, "
                                    "travels beside the edge gradient) | loss + vertex gradient" if split_exchange else
                                    "one all-reduce of [loss | vertex | texture gradients] behind the step")},
            # what the process group really was (gathered through the collective backend before the timed region)
            "world_size_seen": world_seen, "collective_backend": backend_seen, "ranks": ranks_seen,
            "launches_per_step": n_launch,
            # section 8(d)'s bytes of all views over the time and the peak of all GPUs ...
            "hbm_roofline_frac_step": round(step_bytes / elapsed_step / (8e12 * world), 5),
            # ... and the bytes the timed api really owes (render_fit_loss writes no output images and reads no image
            # gradients: 2 x 20 B per output pixel less than 8(d) counts)
            "hbm_roofline_frac_step_owed": round(step_owed / elapsed_step / (8e12 * world), 5),
            "algorithmic_bytes_per_step": step_bytes, "owed_bytes_per_step": step_owed,
            "kernel_ms_per_step": {k: round(ms / n_inst, 4) for k, (c, ms) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1])},
            # ... and with the step's kernels on one stream: each by itself (their sum exceeds the step: the branches overlap)
            "kernel_ms_per_step_one_stream": {k: round(ms / n_inst, 4) for k, (c, ms) in sorted(ktimes_alone.items(), key=lambda kv: -kv[1][1])},
            "roofline": roof,
            "dropin": dropin,
        }
        if strong is not None:
            out["strong_scaling"] = strong
        if world == 1 and not args.no_cpu_baseline:
            stage("cpu baseline ...")
            out["cpu_baseline"] = cpu_baseline(args.mesh_n, S, ts)
        stage("done")
        print(json.dumps(out))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
