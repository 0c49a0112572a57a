"""CPU tests: the C ABI library loads and exports every symbol include/d3m_raster.h declares, host-side
logic (config loader, view sharding, synthetic workloads, byte model), and the N>1 path over gloo."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "d3m_raster.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(d3m_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    from deep3dmap_amd.build import build_library
    from deep3dmap_amd import _lib
    path = build_library()
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    declared = _header_functions()
    assert len(declared) >= 24
    missing = [f for f in declared if f not in exported]
    assert not missing, f"declared in include/d3m_raster.h but not exported: {missing}"
    # and the ctypes table binds exactly the declared set (no compute calls: there is no GPU here)
    assert sorted(_lib.exported_symbols()) == declared
    L = _lib.lib()
    assert b"gfx950" in L.d3m_version()
    assert L.d3m_forward_workspace_bytes(8, 200704, 512) > L.d3m_forward_workspace_min_bytes(8, 200704, 512) > 0
    # per-pixel walk records (24 B) + plan; the minimum leaves out only the per-crossing room
    assert L.d3m_backward_pixel_map_workspace_bytes(1, 10, 64) > L.d3m_backward_pixel_map_workspace_min_bytes(1, 10, 64) > 64 * 64 * 24
    assert L.d3m_edge_plan_bytes(1, 10, 64) > L.d3m_edge_plan_min_bytes(1, 10, 64) > 0
    assert L.d3m_error_string(2) == b"workspace missing or too small"


def test_any_header_edit_marks_the_library_stale(tmp_path, monkeypatch):
    """build._stale(): the library is rebuilt when ANY file the translation unit includes is newer than it -- the list is
    derived from csrc/*.h (round 5's hand-written one had lost d3m_front.h and d3m_tail.h) -- and every quoted #include of
    the sources resolves to a file of that list."""
    import shutil
    from deep3dmap_amd import build
    headers = build._headers()
    names = {os.path.basename(h) for h in headers}
    assert {"d3m_front.h", "d3m_tail.h", "d3m_edge_grad.h", "d3m_raster.h"} <= names
    for f in headers + [os.path.join(build.CSRC, s) for s in build.SOURCES]:
        for inc in re.findall(r'#include\s+"([^"]+)"', open(f).read()):
            assert os.path.basename(inc) in names, (f, inc)
    # a copy of the tree's layout with controlled time stamps: the library newer than everything, then one file touched
    pkg = tmp_path / "pkg"
    (pkg / "csrc").mkdir(parents=True)
    (pkg / "lib").mkdir()
    (tmp_path / "include").mkdir()
    for f in headers + [os.path.join(build.CSRC, s) for s in build.SOURCES]:
        dst = (tmp_path / "include" if os.path.basename(f) == "d3m_raster.h" else pkg / "csrc") / os.path.basename(f)
        shutil.copy(f, dst)
        os.utime(dst, (1000, 1000))
    lib = pkg / "lib" / "libd3m_raster.so"
    lib.write_bytes(b"")
    os.utime(lib, (2000, 2000))
    monkeypatch.setattr(build, "_PKG", str(pkg))
    monkeypatch.setattr(build, "CSRC", str(pkg / "csrc"))
    monkeypatch.setattr(build, "LIB_PATH", str(lib))
    assert not build._stale()
    touched = [os.path.join(build.CSRC, s) for s in build.SOURCES] + build._headers()
    assert len(touched) == len(headers) + len(build.SOURCES)
    for f in touched:
        os.utime(f, (3000, 3000))
        assert build._stale(), f
        os.utime(f, (1000, 1000))
        assert not build._stale()


def test_stream_keyed_cache_is_bounded_and_lru():
    """rasterize_ops.StreamKeyedCache (scratch buffers / side streams per raw stream handle): at most max_per_kind entries
    per kind, the least recently USED goes first, other kinds are untouched."""
    from deep3dmap_amd.neural_renderer.rasterize_ops import StreamKeyedCache
    c = StreamKeyedCache(max_per_kind=3)
    for stream in range(3):
        c.put(("fwd", "cuda:0", stream), f"fwd{stream}")
    c.put(("edge", "cuda:0", 0), "edge0")
    assert len(c) == 4
    assert c.get(("fwd", "cuda:0", 0)) == "fwd0"            # 0 is now the most recently used of its kind
    c.put(("fwd", "cuda:0", 3), "fwd3")                      # evicts stream 1, the least recently used
    assert c.get(("fwd", "cuda:0", 1)) is None
    assert [c.get(("fwd", "cuda:0", k)) for k in (0, 2, 3)] == ["fwd0", "fwd2", "fwd3"]
    assert c.get(("edge", "cuda:0", 0)) == "edge0"
    for stream in range(10, 30):
        c.put(("fwd", "cuda:0", stream), stream)
    assert len(c) == 4 and c.get(("edge", "cuda:0", 0)) == "edge0"
    c.put(("fwd", "cuda:0", 29), "again")                    # replacing an entry does not evict
    assert len(c) == 4 and c.get(("fwd", "cuda:0", 29)) == "again"
    c.clear()
    assert len(c) == 0


def test_coverage_form_switch_is_host_state():
    """d3m_set_coverage_form: -1 / 0 / 1 accepted and read back, anything else D3M_ERR_INVALID and no change; the
    context manager restores the previous form (no launch involved: runs without a GPU)."""
    from deep3dmap_amd import _lib
    L = _lib.lib()
    start = L.d3m_get_coverage_form()
    assert start in (-1, 0, 1)
    for form in (0, 1, -1):
        assert L.d3m_set_coverage_form(form) == 0 and L.d3m_get_coverage_form() == form
    assert L.d3m_set_coverage_form(2) == 1 and L.d3m_set_coverage_form(-2) == 1 and L.d3m_get_coverage_form() == -1
    with _lib.coverage_form("binned"):
        assert L.d3m_get_coverage_form() == 0
        with _lib.coverage_form("bidding"):
            assert L.d3m_get_coverage_form() == 1
        assert L.d3m_get_coverage_form() == 0
    assert L.d3m_get_coverage_form() == -1
    L.d3m_set_coverage_form(start)
    # the deterministic switch: 0 / 1 accepted and read back, anything else refused; the context manager restores
    was = L.d3m_get_deterministic()
    assert was in (0, 1)
    assert L.d3m_set_deterministic(1) == 0 and L.d3m_get_deterministic() == 1
    assert L.d3m_set_deterministic(2) == 1 and L.d3m_set_deterministic(-1) == 1 and L.d3m_get_deterministic() == 1
    assert L.d3m_set_deterministic(0) == 0 and L.d3m_get_deterministic() == 0
    with _lib.deterministic():
        assert L.d3m_get_deterministic() == 1
    assert L.d3m_get_deterministic() == 0
    L.d3m_set_deterministic(was)


def test_no_kernel_spills_to_scratch():
    """`hipcc -Rpass-analysis=kernel-resource-usage` on the product build's flags: NO kernel of the library uses scratch
    memory (round 3's 64-register form of k_edge_lines spilled 24 B per lane = 335 MB of stores per headline launch, and
    nobody looked), and the hot kernels keep the occupancy their launch bounds were chosen for."""
    from deep3dmap_amd.resource_usage import kernel_resource_usage
    table = kernel_resource_usage()
    assert len(table) >= 100
    spilling = {k: (v["scratch"], v["vgpr_spill"], v["sgpr_spill"]) for k, v in table.items()
                if v["scratch"] or v["vgpr_spill"]}
    assert not spilling, spilling

    def rows(prefix):
        got = [v for k, v in table.items() if ("d3m::" + prefix) in k]
        assert got, prefix
        return got
    assert all(r["vgprs"] <= 64 and r["occupancy"] == 8 for r in rows("k_edge_lines<"))
    assert all(r["vgprs"] <= 128 for r in rows("k_render_lit_fit_records<")) and all(r["vgprs"] <= 128 for r in rows("k_backward_textures_lit_faces"))
    for name in ("k_raster_tiles<", "k_bid_faces<", "k_edge_scatter<", "k_edge_count_window<", "k_edge_gather<", "k_bin_count<"):
        assert all(r["scratch"] == 0 for r in rows(name))


def test_library_is_gfx950_code_object():
    from deep3dmap_amd.build import LIB_PATH
    blob = open(LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"gfx942" not in blob and b"sm_" not in blob


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "deep3dmap_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                text = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b|nr_oracle|libnr_oracle|_ref/", text, flags=re.M):
                    bad.append(f)
    assert not bad, bad


def test_ops_fail_loudly_without_gpu_tensors():
    from deep3dmap_amd.neural_renderer import rasterize_ops as ops
    f = torch.zeros(1, 2, 3, 3)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ops.forward_face_index_map(f, f, f, f, f, f, 8, 0.1, 100.0, 0, 0, 0)


def test_missing_library_raises(monkeypatch):
    from deep3dmap_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libd3m_raster.so")
    with pytest.raises(RuntimeError, match="no CPU / eager fallback"):
        _lib.lib()


def test_config_loader_and_renderer_keys():
    from deep3dmap_amd.config import load_config
    cfg = load_config(os.path.join(ROOT, "tests", "fixtures", "gan2shape_like.py"))
    mc = cfg.model.model_cfgs
    assert cfg.model.type == "Gan2Shape" and mc.get("fov") == 10 and mc.get("tex_cube_size") == 2
    assert mc.get("renderer_max_depth", 10.) == 10. and cfg.dist_params.backend == "nccl"
    cfg2 = load_config(os.path.join(ROOT, "tests", "fixtures", "pt3d_like.py"))      # _base_ inheritance
    assert cfg2.model.type == "imgs2mesh" and cfg2.model.model_cfgs.image_size == 64
    assert cfg2.model.model_cfgs.tuplesize == 3 and cfg2.work_dir == "results/example_pt3d"


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs"), reason="reference configs only exist in the build container")
@pytest.mark.parametrize("rel", ["configs/gan2shape/celeba.py", "configs/gan2shape/car.py",
                                 "configs/pt3d_demos/imgs2face_multipie.py"])
def test_reference_configs_load_unmodified(rel):
    from deep3dmap_amd.config import load_config
    path = os.path.join("/root/reference", rel)
    if not os.path.exists(path):
        pytest.skip(f"{rel} not in this reference checkout")
    cfg = load_config(path)
    mc = cfg.model.model_cfgs
    assert "image_size" in mc
    if cfg.model.type == "Gan2Shape":
        assert {"fov", "tex_cube_size", "rot_center_depth", "min_depth", "max_depth"} <= set(mc)


def test_shard_views_and_synthetic_workloads():
    from deep3dmap_amd import synthetic
    from deep3dmap_amd.multiview import shard_views
    assert [shard_views(32, r, 4) for r in range(4)] == [(0, 8), (8, 16), (16, 24), (24, 32)]
    # a camera count that does not divide: the remainder goes to the low ranks, shards stay contiguous and complete
    assert [shard_views(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [shard_views(8, r, 3) for r in range(3)] == [(0, 3), (3, 6), (6, 8)]
    for n, w in ((32, 8), (33, 8), (7, 7), (256, 8), (5, 3)):
        parts = [shard_views(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        sizes = [hi - lo for lo, hi in parts]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True) and min(sizes) >= 1
    with pytest.raises(ValueError):
        shard_views(3, 0, 4)                # fewer cameras than ranks
    with pytest.raises(ValueError):
        shard_views(8, 4, 4)
    v, t = synthetic.grid_mesh(225)
    assert v.shape == (50625, 3) and t.shape == (100352, 3) and t.dtype == np.int32
    assert t.min() == 0 and t.max() == 50624 and np.abs(v).max() <= 1.0 + 1e-6
    assert synthetic.grid_topology(64).shape[0] == 7938 and synthetic.grid_topology(709).shape[0] == 1002528
    iv, it = synthetic.icosphere(1)
    assert iv.shape == (42, 3) and it.shape == (80, 3)
    # topology identical to deep3dmap's get_face_idx (oracle restatement pinned by the golden vectors)
    from oracle import nr_oracle as O
    assert np.array_equal(synthetic.grid_topology(7), O.get_face_idx(1, 7, 7)[0].numpy())
    eyes = synthetic.camera_ring(8)
    assert eyes.shape == (8, 3) and np.allclose(np.linalg.norm(eyes, axis=1), 2.732, atol=1e-4)


def test_algorithmic_byte_model_matches_survey_worked_values():
    sys.path.insert(0, ROOT)
    import bench
    a_fwd, a_bwd = bench.algorithmic_bytes(50625, 100352, 512, 512, 2)
    assert abs(a_fwd / 1e6 - 21.93) < 0.02 and abs(a_bwd / 1e6 - 26.73) < 0.02        # SURVEY.md 8(d), C4 headline
    a_fwd, a_bwd = bench.algorithmic_bytes(50625, 100352, 512, 512, 2, alpha=1, depth=0, rgb=0, tex_grad=0)
    assert abs((a_fwd + a_bwd) / 1e6 - 17.86) < 0.02                                      # silhouettes only
    # the per-kernel shares PARTITION section 8(d): over the kernels of the headline step, in either form of coverage and
    # with anti-aliasing (S = 2 s), they add up to A_fwd + A_bwd exactly, every term owned once
    for V, F, S, s_out, ts in ((50625, 100352, 512, 512, 2), (26896, 53138, 512, 256, 2), (502681, 1002528, 1024, 1024, 4)):
        a_fwd, a_bwd = bench.algorithmic_bytes(V, F, S, s_out, ts)
        assert sum(bench.byte_terms(V, F, S, s_out, ts).values()) == a_fwd + a_bwd
        for form, kernels in bench.STEP_KERNELS.items():
            shares = bench.byte_partition(kernels, V, F, S, ts, s_out)
            assert "unowned" not in shares and sum(shares.values()) == a_fwd + a_bwd, (form, shares)
            assert all(bench.kernel_bytes(k, V, F, S, ts, kernels, s_out) == (shares[k] or None) for k in kernels)
    # the dominant kernel's share: values + owner + gradients of every pixel
    assert bench.kernel_bytes("k_edge_lines", 50625, 100352, 512, 2) == 36 * 512 * 512
    assert bench.kernel_bytes("k_bin_fill", 50625, 100352, 512, 2) is None               # owns no term of 8(d)


def test_committed_bench_line_and_profiles_are_consistent():
    """The committed measurement artefacts: the bench line carries the contract's keys, its roofline object agrees
    with the committed PMC / SQ summaries that bench.py reads, and the kernel it names is in the rocprofv3 stats."""
    import csv
    import json
    sys.path.insert(0, ROOT)
    import bench
    R = "r06"                                           # the round whose artefacts bench.py reads (the newest)
    line = json.load(open(os.path.join(ROOT, "profiles", f"{R}_bench_final.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "dropin"):
        assert key in line, key
    assert line["unit"] == "Mpix/s" and line["vs_baseline"] is None and line["dtype"] == "f32"
    assert "workload" in line["config"] and "model" not in line["config"]
    roof = line["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4
    traffic, traffic_src = bench.measured_traffic(roof["kernel"])
    assert roof["traffic"] == traffic and roof["traffic_source"] == traffic_src and traffic_src.startswith(f"profiles/{R}_")
    valu, valu_src = bench.measured_valu(roof["kernel"])
    assert roof.get("valu_wave_instructions") == valu and roof.get("valu_source") == valu_src
    assert line["cpu_baseline"]["kind"] in ("port", "reference") and line["cpu_baseline"]["cores"] >= 1
    # rocprofv3's average duration of that kernel agrees with the HIP-event average in the bench line -- by itself (the step's
    # kernels on one stream: roofline.frac) and in the step as timed (the render node's branches share the chip:
    # roofline.in_step), each against the committed summary of that command
    def rocprof_us(name):
        rows = [r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", f"{R}_{name}.csv"))) if roof["kernel"] in r["Name"]]
        assert rows, name
        return sum(float(r["TotalDurationNs"]) for r in rows) / sum(int(r["Calls"]) for r in rows) / 1e3
    assert abs(rocprof_us("kernel_stats_one_stream") - roof["avg_launch_us"]) < 0.1 * roof["avg_launch_us"]
    assert abs(rocprof_us("kernel_stats_final") - roof["in_step"]["avg_launch_us"]) < 0.1 * roof["in_step"]["avg_launch_us"]
    assert abs(roof["frac_rocprof"] - roof["algorithmic_bytes_per_launch"] / rocprof_us("kernel_stats_one_stream") / 1e3 / 8000) < 1e-3
    assert roof["in_step"]["frac"] <= roof["frac"] * 1.02 and "D3M_SERIAL_BRANCHES" not in line["env_switches"]
    # the honest-roofline fields: the bytes the timed api really owes, and the dominant kernel's
    # fraction from the committed rocprofv3 average beside the HIP-event one
    assert 0 < line["hbm_roofline_frac_step_owed"] < line["hbm_roofline_frac_step"]
    assert abs(line["hbm_roofline_frac_step_owed"] / line["hbm_roofline_frac_step"] - (48.66 - 10.49) / 48.66) < 0.01
    assert line["launches_per_step"] <= 18
    # the drop-in form of the same step rides in the line (within 5 % of the fused objective since round 5), and the gan2shape
    # block's line carries its launch count
    assert "Renderer.render" in line["dropin"]["api"] and 0 < line["dropin"]["value"] < line["value"]
    assert line["dropin"]["ms_per_step"] <= 1.05 * line["ms_per_step"]
    # ... and (round 6) what an UNMODIFIED caller gets: render() with nothing registered + the loss operators / eager torch
    for form in ("generic_operators", "generic_torch"):
        g = line["dropin"][form]
        assert "nothing registered" in g["api"] and line["dropin"]["ms_per_step"] < g["ms_per_step"] < 1.6 * line["ms_per_step"]
        assert abs(g["over_fused"] - g["ms_per_step"] / line["ms_per_step"]) < 2e-3
    assert line["dropin"]["generic_operators"]["ms_per_step"] < line["dropin"]["generic_torch"]["ms_per_step"]
    g2s = json.load(open(os.path.join(ROOT, "profiles", f"{R}_bench_gan2shape.json")))
    assert g2s["launches_per_step"] <= 20 and g2s["ms_per_step"] <= 0.20 and "roofline" in g2s
    stats = [r["Name"] for r in csv.DictReader(open(os.path.join(ROOT, "profiles", f"{R}_kernel_stats_gan2shape.csv")))]
    replayed = [n for n in stats if "k_g2s_" in n]
    assert len(replayed) == 8


def test_design_figures_are_generated_from_the_committed_profiles():
    """The measured blocks of DESIGN.md are what tools_dev/refresh_design.py makes of the committed profiles -- no figure of
    the current round is typed by hand (round 3's text contradicted its own PMC summary) -- and the line kernel's traffic
    in that summary is back below round 2's (no scratch spills: WRITE_SIZE)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("refresh_design", os.path.join(ROOT, "tools_dev", "refresh_design.py"))
    rd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rd)
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert "<!-- GENERATED:measured BEGIN -->" in text and "<!-- GENERATED:configs BEGIN -->" in text
    assert rd.regenerate(text, "r06") == text, "run `python tools_dev/refresh_design.py r06` after updating profiles/"
    # "<= 40 KB of current design + the generated block" -- the hand-written part is what is bounded
    written = re.sub(r"<!-- GENERATED:(\w+) BEGIN -->.*?<!-- GENERATED:\1 END -->", "", text, flags=re.S)
    # (round 6: 43 KB -- the alpha-only path, the deterministic mode and the f64 measurement of K4 are current design)
    assert len(written.encode()) <= 43 * 1024, "DESIGN.md states the current design in <= 43 KB; history goes to docs/EXPERIMENTS.md"
    assert len(text.encode()) <= 49 * 1024
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_traffic_final.json")))["kernels"]
    lines = next(v for k, v in pmc.items() if k.startswith("k_edge_lines"))
    # round 2: 95 MB written / 596 MB in all; round 3 (24 B of scratch per lane): 353 / 884; the results alone are ~90 MB;
    # round 6 (16-byte crossing records): 517 in all, and the scatter pass that writes the records 304 -> 209
    assert lines["write_size_KiB"] * 1024 <= 130e6 and lines["hbm_bytes_per_launch"] <= 540e6, lines
    scatter = next(v for k, v in pmc.items() if k.startswith("k_edge_scatter"))
    assert scatter["hbm_bytes_per_launch"] <= 230e6, scatter


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from deep3dmap_amd.multiview import allreduce_flat, allreduce_sum_, shard_views
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
lo, hi = shard_views(8, rank, world)
# each rank contributes the sum over ITS views of a per-view "gradient"; the all-reduce must give the sum over all views
gv = sum(torch.full((5, 3), float(v + 1)) for v in range(lo, hi))
gt = sum(torch.arange(4.0).reshape(2, 2) * (v + 1) for v in range(lo, hi))
out_v, out_t = allreduce_flat([gv, gt])
assert torch.equal(out_v, torch.full((5, 3), 36.0)), out_v
assert torch.equal(out_t, torch.arange(4.0).reshape(2, 2) * 36.0), out_t

# The sharded fit objective (MultiViewFit): photometric terms are means over the mask of ALL cameras.  With every
# rank normalising by the all-reduced sum(mask), shard values and gradients add up to the unsharded objective; with
# the LOCAL sum (the round-1 defect) they do not.
g = torch.Generator().manual_seed(5)
n, hw = 8, 6
rgb_t, alpha_t = torch.rand(n, 3, hw, hw, generator=g), (torch.rand(n, hw, hw, generator=g) > 0.4).float()
alpha_t[:4] *= (torch.rand(4, hw, hw, generator=g) > 0.5).float()         # unequal masks across the shards
x0 = torch.rand(n, 3, hw, hw, generator=g)
def objective(x, a, sl, den):
    return ((x[sl] - rgb_t[sl]).abs() * alpha_t[sl, None]).sum() / (3 * den) + ((a[sl] - alpha_t[sl]) ** 2).sum() / (hw * hw)
a0 = torch.rand(n, hw, hw, generator=g)
x, a = x0.clone().requires_grad_(True), a0.clone().requires_grad_(True)
full = objective(x, a, slice(0, n), alpha_t.sum())
full.backward()
xs, as_ = x0.clone().requires_grad_(True), a0.clone().requires_grad_(True)
den = allreduce_sum_(alpha_t[lo:hi].sum().reshape(1))
assert torch.allclose(den, alpha_t.sum())
part = objective(xs, as_, slice(lo, hi), den[0])
part.backward()
tot = allreduce_sum_(part.detach().reshape(1).clone())
assert torch.allclose(tot[0], full.detach(), rtol=1e-6), (tot, full)
assert torch.allclose(xs.grad[lo:hi], x.grad[lo:hi], rtol=1e-6, atol=1e-9) and torch.allclose(as_.grad[lo:hi], a.grad[lo:hi])
wrong = objective(x0, a0, slice(lo, hi), alpha_t[lo:hi].sum())
tot_wrong = allreduce_sum_(wrong.reshape(1).clone())
assert not torch.allclose(tot_wrong[0], full.detach(), rtol=1e-3)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world", [2, 3])
def test_gradient_allreduce_over_gloo(tmp_path, world):
    """N>1 path on CPU: `world` processes, gloo, camera shards -> one flat SUM all-reduce.  World size 3: the 8 cameras do
    not divide (shards of 3, 3 and 2) and the sharded objective still equals the unsharded one."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29529 + world), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all(f"rank {r} ok" in outs[r] for r in range(world))


def test_form_of_coverage_and_big_batch_by_configuration():
    """Host logic, no launch: which form of coverage a launch takes (d3m_forward_coverage_form: 0 = per-tile lists, 1 =
    bidding) and whether it is a BIG BATCH (d3m_forward_big_batch: the camera-sharded fit then splits its exchange) -- for BASELINE's
    configurations and for the coarse meshes round 5 moved to the lists (docs/EXPERIMENTS.md C: crossovers measured)."""
    from deep3dmap_amd import _lib
    L = _lib.lib()
    assert L.d3m_get_coverage_form() == -1
    cases = [  # (views, triangles, raster) -> (form, big batch)
        ((32, 100352, 512), (0, 1)),      # config 4 on one GPU: lists, big
        ((16, 100352, 512), (1, 0)),      # its shards: bidding
        ((4, 100352, 512), (1, 0)),
        ((1, 53138, 512), (1, 0)),        # config 2 (AA: S = 512): 5 raster pixels per triangle
        ((8, 1002528, 1024), (1, 0)),     # config 5: sub-pixel triangles bid whatever the batch
        ((32, 1002528, 1024), (1, 0)),
        ((16, 7938, 128), (1, 0)),        # the gan2shape block's mesh through the general path: 2 px per triangle
        ((1, 2450, 512), (0, 0)),         # coarse meshes: lists even in a small batch
        ((1, 19602, 512), (0, 0)),        # 13 px per triangle, one view: lists
        ((8, 19602, 512), (1, 0)),        # the same mesh, eight views (157 k pairs): bidding
        ((8, 722, 128), (0, 0)),          # eight views of a tiny mesh: not the faces for bidding
        ((8, 7938, 512), (0, 0)),         # 33 px per triangle
    ]
    for args, (form, big) in cases:
        assert L.d3m_forward_coverage_form(*args) == form, args
        assert L.d3m_forward_big_batch(*args) == big, args
    assert L.d3m_forward_coverage_form(0, 1, 1) == -1 and L.d3m_forward_big_batch(1, 0, 1) == -1
