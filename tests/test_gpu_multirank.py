"""The N > 1 path end to end on one GPU: two ranks (one process each, both on cuda:0, `gloo` carrying the rows through host
memory in place of RCCL) render their row tiles of an adaptive frame with rtmi_render_device, gather them with
tiles.RowGather, and rank 0 compares the assembled frame bit for bit with the frame it renders alone (bench.py
--check-frame exits 3 on a difference)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("scene,nranks,gather", [("ch11a", 2, "f32"), ("dice", 3, "f32"), ("ch11b", 2, "rgb8")])
def test_sharded_frame_equals_single_rank_frame(scene, nranks, gather):
    port = 29600 + nranks + (10 if gather == "rgb8" else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--backend", "gloo", "--check-frame",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--scene", scene, "--width", "480", "--height", "270", "--gather", gather]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "gathered frame EQUALS" in r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == nranks and out["scaling"] == "strong" and out["value"] > 0


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset): bench.py starts torch.distributed.run itself, before
    anything touches the GPU, and hands rank 0's JSON line and the exit code through."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--check-frame", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--scene", "ch11a", "--width", "480", "--height", "270"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "gathered frame EQUALS" in r.stderr
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0
