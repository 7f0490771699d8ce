"""bench.py launch plumbing on CPU: `python bench.py --gpus 2` must start its own ranks and the same
command under an external torch.distributed.run must not start them twice."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_self_launch_two_ranks():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--dry-run", "--backend", "gloo"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["ranks"] == 2 and j["steps"] == 3 and j["warmup"] == 1


def test_external_launcher_two_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2",
                        "--steps", "2", "--warmup", "0", "--dry-run", "--backend", "gloo"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _last_json(r.stdout)["ranks"] == 2


def test_child_failure_is_relayed():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run", "--backend", "nosuchbackend"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


def test_single_rank_dry_run_needs_no_launcher():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--dry-run", "--steps", "2"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _last_json(r.stdout)["ranks"] == 1
