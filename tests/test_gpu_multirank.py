"""The camera-sharded fit on ONE device: R ranks x 32/R cameras must equal 1 rank x 32 cameras (BASELINE config 4:
100,352 triangles @512x512, 32 cameras), with HIP-graph replay on.  Ranks are child processes sharing cuda:0, the
collective is gloo (RCCL refuses two ranks on one GPU); everything else is the code path bench.py runs with --gpus N."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, out, extra=()):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "sharded_worker.py"), "--out", out, *extra]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    log = "\n".join(f"--- world {world} rank {r} (exit {p.returncode}) ---\n{o}" for r, (p, o) in enumerate(zip(procs, outs)))
    log_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(log_dir, exist_ok=True)
    with open(os.path.join(log_dir, "sharded_fit_single_device.log"), "a") as f:
        f.write(log + "\n")
    assert all(p.returncode == 0 for p in procs), log
    return [np.load(f"{out}.rank{r}.npz") for r in range(world)]


@pytest.mark.parametrize("world,materialise", [(2, 0), (4, 0), (2, 1)])
def test_sharded_fit_equals_unsharded_with_graph_replay(tmp_path, world, materialise):
    extra = ("--materialise-images", str(materialise))
    one = _run(1, str(tmp_path / "w1"), extra)[0]
    many = _run(world, str(tmp_path / f"w{world}"), extra)
    assert abs(float(many[0]["mask_sum"]) - float(one["mask_sum"])) <= 1e-6 * float(one["mask_sum"])
    for r in range(world):
        # every rank ends up with the gradient of the WHOLE 32-camera objective
        assert abs(float(many[r]["loss"]) - float(one["loss"])) <= 1e-5 * abs(float(one["loss"])), (r, many[r]["loss"], one["loss"])
        for k in ("gv", "gt"):
            a, b = many[r][k], one[k]
            assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max(), (r, k, np.abs(a - b).max(), np.abs(b).max())
