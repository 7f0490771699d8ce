"""Two data-parallel ranks of MouseModel.train_step on the HIP path: gradients == mean of the per-rank
gradients, identical parameters and EMA on every rank after the step, optional (readout) buckets with forward(x, index).
The ranks are fresh child processes started by torch.distributed.run BEFORE anything touches the GPU.
  * >= 2 visible devices: one rank per GPU over RCCL (backend "nccl") — the configuration the 8-GPU bench uses;
  * 1 visible device: both ranks share cuda:0 and exchange through gloo, so the hook-driven bucket path over HIP gradients
    is exercised on every GPU box (RCCL refuses two ranks on one device)."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(backend, *extra, nproc=2):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "tests" / "ddp_gpu_worker.py"), backend, *extra]
    res = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "DDP_WORKER_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


def test_two_ranks_rccl_one_gpu_each():
    if torch.cuda.device_count() < 2:          # device_count() does not initialise the GPU on this image
        pytest.skip("needs two visible GPUs (the single-device variant below runs instead)")
    _run("nccl")


def test_two_ranks_sharing_one_gpu_gloo():
    _run("gloo")


def test_two_ranks_rccl_sharded_readout_optimizer():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs (the single-device variant below runs instead)")
    _run("nccl", "shard")


def test_two_ranks_sharing_one_gpu_gloo_sharded_readout_optimizer():
    """ddp_shard_optimizer: reduce-scatter of the readout buckets, fused AdamW/EMA on the owned slice, all-gather of the
    parameters — same parameters / EMA as torch.optim.AdamW on the averaged gradient, identical on both ranks."""
    _run("gloo", "shard")


def test_two_ranks_sharing_one_gpu_gloo_metric_architecture():
    """The same checks on the benchmarked architecture (nine blocks, expansion 7, 1024-2048-4096 cortex, two readouts) at B=2:
    ~25 M parameters in several buckets split at the 12 MB cap, the real kernels' gradients written into the bucket slices."""
    _run("gloo", "full")


@pytest.mark.parametrize("extra", [(), ("shard",), ("full",), ("bf16comm",)], ids=["allreduce", "sharded", "metric_arch", "bf16_exchange"])
def test_one_rank_over_rccl(extra):
    """RCCL on the one-GPU box: a process group of ONE rank over backend "nccl" with the exchange machinery forced on
    (``ddp_single_rank``).  Every collective is the identity, but it goes through RCCL: ``init_process_group("nccl",
    device_id=...)``, the flat-dtype broadcast, ``all_reduce(AVG)`` on fp32 and bf16 buffers launched from autograd hooks,
    ``reduce_scatter_tensor`` / ``all_gather_into_tensor`` in place, handle waits against the HIP kernels' stream, barrier —
    the calls the 8-GPU bench makes, with the same worker assertions (gradients, AdamW/EMA result, optional buckets)."""
    _run("nccl", *extra, nproc=1)
