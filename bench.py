#!/usr/bin/env python3
"""Benchmark of the DwiseNeuro training hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  N>1 either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...:
  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or plainly, in which case bench.py starts that launcher
  itself as a child process before touching the GPU.

A "step" is one full ``MouseModel.train_step`` (src/argus_models.py:43-71 semantics) over one batch of
synthetic clips already resident in HBM: forward, Poisson loss, backward, (N>1: gradient all-reduce over
RCCL overlapped with backward), fused AdamW step, EMA update.  Workload = BASELINE.json ``configs[1]``:
configs/true_batch_001.py single-mouse training (expansion 7, 1 readout of 7863 neurons, dropout 0.4, drop-path
0.1), bf16 storage, B=32 clips of T=32 frames at 36x64 per GPU (weak scaling).

Rank 0 prints ONE JSON line: metric/value/unit (+ ``roofline`` for the dominant kernel family measured live
with HIP events on the launch stream, and ``cpu_baseline`` = the CPU oracle timed on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

# more hardware queues than HIP's default 4, before the runtime starts: the RCCL stream must not share a queue with the compute
# stream (sensorium_amd/ddp.py::init_rccl); no effect on one GPU (24.82-24.90 vs 24.83-24.90 ms/step measured)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

NUM_NEURONS_MOUSE0 = 7863          # src/constants.py:26 (first mouse)
NUM_NEURONS_ALL = (7863, 7908, 8202, 7939, 8122, 7440, 7928, 8285, 7671, 7495)     # src/constants.py:24,31
CORE_FEATURES = (64, 64, 64, 64, 128, 128, 128, 256, 256)
STRIDES = (2, 1, 1, 1, 2, 1, 1, 2, 1)
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def model_params(expansion=7, readouts=(NUM_NEURONS_MOUSE0,)):
    return {
        "nn_module": ("dwiseneuro", {
            "readout_outputs": tuple(readouts), "in_channels": 5, "core_features": CORE_FEATURES,
            "spatial_strides": STRIDES, "spatial_kernel": 3, "temporal_kernel": 5, "expansion_ratio": expansion,
            "se_reduce_ratio": 32, "cortex_features": (1024, 2048, 4096), "groups": 2, "softplus_beta": 0.07,
            "drop_rate": 0.4, "drop_path_rate": 0.1}),
        "loss": ("mice_poisson", {"log_input": False, "full": False, "eps": 1e-8}),
        "optimizer": ("AdamW", {"lr": 3e-4 * 32 / 4, "weight_decay": 0.05}),
        "amp": True, "iter_size": 1,
    }


def workload_name(args):
    if args.distill:
        return (f"configs/distillation_001.py step (NOT the metric config): expansion-6 student, {args.mice} readout(s), "
                "frozen expansion-7 teacher forward + soft-label fill (ratio 0.36), AdamW + EMA")
    if args.mice != 1:
        return (f"configs/true_batch_001.py with {args.mice} readouts (NOT the metric config), expansion {args.expansion}, "
                "AdamW + EMA")
    return ("configs/true_batch_001.py single-mouse training (expansion 7, 1 readout x 7863 neurons, dropout 0.4, "
            "drop-path 0.1), AdamW + EMA" + (", every batch drawn + assembled on the device inside the step (CutMix 0.5)" if getattr(args, "assembled", False) else ""))


def block_shapes(batch, frames, height, width, expansion):
    """(M_in, M_out, Cmid) per block."""
    out = []
    h, w = height, width
    for c, s in zip(CORE_FEATURES, STRIDES):
        ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
        out.append((batch * frames * h * w, batch * frames * ho * wo, c * expansion))
        h, w = ho, wo
    return out


MFMA_PEAK_TFLOPS = 2500.0          # MI355X_MICROARCH.md: bf16 dense MFMA ~2.5 PFLOP/s
MFMA_F32_PEAK_TFLOPS = 157.3       # ... fp32-input MFMA = the fp32 vector rate


def block_work(batch, frames, height, width, expansion, fused_pw_blocks, esize, y1_free_blocks=()):
    """Per block: {family: (algorithmic HBM bytes, FLOPs)} of one launch set of that family in that block, plus the block's
    geometry.  Depth-wise families follow SURVEY.md 8d (read input once + write output once; backward = read x + read dy +
    write dx); GEMM families count every operand / result once (2*M*K*N FLOPs per product).
    `y1_free_blocks`: blocks trained without a materialised y1 (round 5): what runs under `pw_fwd` there is the Gram pass over the
    block input (a0 read once, [a0 | 1]^T a0), not the expand GEMM; the depth-wise families keep SURVEY's byte definition (their
    kernels then move FEWER bytes than that: y1 is rebuilt from a0 on the matrix cores, see `traffic`)."""
    out = []
    h, w = height, width
    for i, st in enumerate(STRIDES):
        # block i: features[i] -> features[i] * expansion -> features[i + 1] (the last block keeps its width), dwiseneuro.py:319-335
        cin, cout = CORE_FEATURES[i], CORE_FEATURES[min(i + 1, len(CORE_FEATURES) - 1)]
        ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
        mi, mo, e_ = batch * frames * h * w, batch * frames * ho * wo, cin * expansion
        d = {
            "dws_fwd": (mi * e_ + mo * e_, 0), "dwt_fwd": (2 * mo * e_, 0), "dws_bwd": (2 * mi * e_ + mo * e_, 0),
            "dwt_bwd": (3 * mo * e_, 0), "se_pool": (2 * mo * e_, 0),
            "pw_fwd": (mi * cin + mi * e_, 2 * mi * cin * e_),
            "pwl_fwd": (mo * e_ + mo * cout, 2 * mo * e_ * cout),
            "pwl_dgrad": (mo * cout + 2 * mo * e_, 2 * mo * e_ * cout),                           # dy4, y3 in; dh3 out
            "pwl_wgrad": (mo * e_ + mo * cout, 2 * mo * e_ * cout),
        }
        if i in y1_free_blocks:
            d["pw_fwd"] = (mi * cin, 2 * mi * cin * (cin + 8))
        if i in fused_pw_blocks:
            # one pass over dh1 for both gradients; y1 is not read (its BatchNorm-backward terms fold into Cin x Cin matrices):
            # dh1, a0 in; da0 out — the bytes this algorithm has to move, not the (dh1, y1) pair autograd would read
            d["pw_dgrad"] = (mi * e_ + 2 * mi * cin, 4 * mi * cin * e_ + 4 * mi * cin * cin)
        else:
            d["pw_dgrad"] = (mi * e_ + 2 * mi * cin, 2 * mi * (cin + e_) * cin)                   # K-concat fold
            d["pw_wgrad"] = (mi * e_ + 2 * mi * cin, 2 * mi * cin * (e_ + cin))                   # [dh1 | a0 | 1]^T a0
        out.append({"block": i, "stride": st, "cin": cin, "cmid": e_, "cout": cout, "in_hw": (h, w), "out_hw": (ho, wo),
                    "m_in": mi, "m_out": mo, "work": {k: (v[0] * esize, v[1]) for k, v in d.items()}})
        h, w = ho, wo
    return out


def family_work(batch, frames, height, width, expansion, readouts, fused_pw_blocks, esize, y1_free_blocks=()):
    """Algorithmic work per *step* of every timed kernel family: (bytes moved through HBM, FLOPs) -- block_work summed over
    the nine blocks, plus the cortex and the readouts.
    `fused_pw_blocks`: blocks whose conv_pw data + weight gradient are one launch (counted under pw_dgrad)."""
    e = {k: 0 for k in ("dws_fwd", "dwt_fwd", "dws_bwd", "dwt_bwd", "pw_fwd", "pwl_fwd", "pwl_dgrad", "pwl_wgrad",
                        "pw_dgrad", "pw_wgrad", "se_pool")}
    f = dict.fromkeys(e, 0)
    for blk in block_work(batch, frames, height, width, expansion, fused_pw_blocks, esize, y1_free_blocks):
        for k, (by, fl) in blk["work"].items():
            e[k] += by
            f[k] += fl
    m = batch * frames
    cx = (CORE_FEATURES[-1],) + (1024, 2048, 4096)
    e["cortex_fwd"] = esize * sum(m * a + m * b for a, b in zip(cx[:-1], cx[1:]))
    f["cortex_fwd"] = sum(2 * m * a * b // 2 for a, b in zip(cx[:-1], cx[1:]))                   # groups = 2
    e["cortex_bwd"] = 2 * e["cortex_fwd"]
    f["cortex_bwd"] = 2 * f["cortex_fwd"]
    npad = [(n + 1) // 2 * 2 for n in readouts]
    e["readout_fwd"] = sum(m * 4096 * esize + 4 * m * n for n in npad)                             # fp32 predictions
    f["readout_fwd"] = sum(2 * m * 2048 * n for n in npad)
    e["readout_bwd"] = sum(2 * m * 4096 * esize + 4 * m * n for n in npad)
    f["readout_bwd"] = 2 * f["readout_fwd"]
    return e, f


def lib_sha16():
    import hashlib
    import sensorium_amd._lib as L
    return hashlib.sha256(Path(L.LIB_PATH).read_bytes()).hexdigest()[:16]


def cpu_baseline(frames, height, width, expansion):
    """The CPU oracle (a port: oracle/dwiseneuro_oracle.py, pinned to the reference by tests/golden) timed on this host on a
    BOUNDED sample of the benchmark workload, following BASELINE.md section 2: fwd + loss + bwd of B=4 clips (the largest
    batch that plan names: ~13 GB of host memory) at the benchmark's T, HxW and width, fp32, all host cores (capped at 32:
    more only adds contention at these sizes).  Warm-ups: one tiny, one at a quarter of the frames (timed, to size the
    sample), one at the timed size; then the MEDIAN of three full-length steps when one stays under ~12 s (about 30-40 s of
    CPU work in all), else of five quarter-length steps (the path is linear in T)."""
    from oracle import dwiseneuro_oracle as orc
    import numpy as np
    orc.DW_IMPL = "library"        # depth-wise convs through torch's conv3d, like the reference's CPU path
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    bsz = 4
    sd = orc.make_state_dict(readout_outputs=(NUM_NEURONS_MOUSE0,), expansion_ratio=expansion, seed=0)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and "inv_freq" not in k
              else v) for k, v in sd.items()}
    rng = np.random.default_rng(20231122)
    x = torch.from_numpy(rng.normal(size=(bsz, 5, frames, height, width)).astype(np.float32) * 40 + 80)
    target = torch.from_numpy(np.maximum(rng.normal(size=(bsz, NUM_NEURONS_MOUSE0, frames)), 0).astype(np.float32))
    w = torch.ones(bsz, 1)

    def step(t):
        for v in sd.values():
            if getattr(v, "grad", None) is not None:
                v.grad = None
        preds = orc.forward(sd, x[:, :, :t], strides=STRIDES, readout_outputs=(NUM_NEURONS_MOUSE0,), training=True)
        loss = orc.mice_poisson_loss(preds, [target[:, :, :t]], w)
        loss.backward()

    step(2)                                    # warm-up 1: thread pool, allocator
    t_q = max(4, frames // 4)
    t0 = time.perf_counter()
    step(t_q)                                  # warm-up 2, timed to size the sample
    quarter = time.perf_counter() - t0
    full = quarter * (frames / t_q) < 12.0
    t_s, n_timed = (frames, 3) if full else (t_q, 5)
    step(t_s)                                  # warm-up 3 at the timed size
    times = []
    for _ in range(n_timed):
        t0 = time.perf_counter()
        step(t_s)
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(bsz * (t_s / frames) / med, 4), "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": f"median of {n_timed} fwd+loss+bwd steps (after 3 warm-ups) of the CPU oracle on B={bsz} clips (BASELINE.md "
                      f"section 2: the largest batch its plan names), {t_s} of {frames} frames, {height}x{width}, expansion "
                      f"{expansion}, 1 readout, fp32, torch {torch.__version__}, mkldnn={torch.backends.mkldnn.is_available()}; "
                      f"step times {min(times):.2f}-{max(times):.2f} s; scaled to full clips"}


def other_config(extra):
    """One more workload of BASELINE.json `configs` through this same script as a child process (never an exec), condensed."""
    import subprocess
    cmd = [sys.executable, str(ROOT / "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-inference",
           "--no-rooflines", "--no-fwd-bwd", "--no-other-configs", *extra]
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
        d = json.loads(line)
    except Exception as e:                      # reported, not fatal: the metric line must still be printed
        return {"error": f"{type(e).__name__}: {e}"}
    r = d.get("roofline") or {}
    return {"workload": d["config"]["workload"], "ms_per_step": d["ms_per_step"], "clips_per_s": d["value"], "steps": d["steps"],
            "loss": d.get("loss"), "dominant_family": r.get("kernel"), "frac": r.get("frac"), "achieved_gbs": r.get("achieved")}


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    bench.py <same args>` as a child (never an exec: this may only happen before any GPU call, and the child is a fresh
    process anyway), pass its output through (rank 0 prints the one JSON line) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, rank, world):
    """Launcher / rendezvous plumbing without a GPU (tests/test_bench_launch.py): the same barrier + max-over-ranks
    timing and rank-0 JSON line as the real run, with the timed loop replaced by a sleep.  Never a measurement."""
    if world > 1:
        dist.init_process_group(args.backend if args.backend != "nccl" else "gloo")
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * args.steps)
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ranks": dist.get_world_size() if world > 1 else 1,
                          "ms_per_step": round(float(t.item()) / args.steps * 1e3, 3)}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--height", type=int, default=36)
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--expansion", type=int, default=7)
    ap.add_argument("--no-fwd-bwd", action="store_true", help="skip the extra forward+backward-only timing loop")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--mice", type=int, default=1, help="readouts (1 = the metric config; 10 = the ten-mouse model, "
                    "configs/true_batch_001.py:23 with constants.num_neurons) — other workloads are labelled as such")
    ap.add_argument("--distill", action="store_true", help="configs/distillation_001.py step: frozen expansion-7 teacher "
                    "forward + soft-label fill (ratio 0.36) + expansion-6 student step")
    ap.add_argument("--roofline-family", default="dws_bwd")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inference", action="store_true", help="skip the 7-fold sliding-window inference leg (BASELINE.json "
                    "configs[4]) that runs after the timed training region")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the BASELINE.json configs[2] / configs[3] legs (ten readouts; "
                    "distillation step) that run as child processes after the timed region")
    ap.add_argument("--no-rooflines", action="store_true", help="skip the extra untimed steps that time every kernel family")
    ap.add_argument("--roctx", action="store_true", help="bracket every kernel-family launch with a roctx range (dwn:pw_fwd, ...): "
                    "run under rocprofv3 --marker-trace --kernel-trace for a labelled timeline")
    ap.add_argument("--profile-all", action="store_true", help="time every kernel family (adds event overhead)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for 1-GPU testing)")
    ap.add_argument("--share-device", action="store_true", help="testing only: every rank uses cuda:0")
    ap.add_argument("--ddp-comm", default="f32", choices=["f32", "bf16"], help="gradient exchange type (bf16: half the bytes "
                    "on xGMI, fp32 master buckets)")
    ap.add_argument("--ddp-shard", action="store_true", help="readout buckets: reduce-scatter -> AdamW/EMA on the owned 1/N slice "
                    "-> all-gather of the parameters (hidden behind the next core forward) instead of all-reduce + full optimizer")
    ap.add_argument("--arena-gb", type=float, default=0.0, help="reserve ONE device segment of this size before the first step (allocated "
                    "and handed straight back to torch's caching allocator, which then carves every activation out of it)")
    ap.add_argument("--compute-stream", action="store_true", help="run the steps on a stream of their own instead of the legacy default "
                    "stream (which synchronises implicitly with every blocking stream)")
    ap.add_argument("--ddp-single-rank", action="store_true", help="testing: run the data-parallel machinery (RCCL process group, flat "
                    "buckets, hook-launched all-reduce, barrier + max-over-ranks timing) on ONE rank — the 8-GPU code path on a 1-GPU box")
    ap.add_argument("--assembled", action="store_true", help="every timed step draws and assembles its own batch on the device "
                    "(sensorium_amd.data_gpu.DeviceBatchLoader: frame stacking, padding, CutMix alpha 1 / prob 0.5 as in "
                    "configs/true_batch_001.py:76-79, sparse targets; SURVEY 8f ranks 3-4) from synthetic trials resident in HBM — "
                    "the reference's per-step H2D copy (argus_models.py:49) and DataLoader work, on the device")
    ap.add_argument("--dry-run", action="store_true", help="testing only: launcher + rendezvous + timing plumbing, no GPU work")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as a CHILD process and before anything in this
        # process touches the GPU (an exec from a GPU-initialised process is forbidden on the pool), relay rank 0's line
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        return dry_run(args, rank, world)
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ddp = world > 1 or args.ddp_single_rank
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            if os.environ.get("DWN_PG_DEFAULT_STREAM") == "1":       # tools/rccl_queue_probe.sh: the torch default, for comparison
                dist.init_process_group("nccl", device_id=dev)
            else:
                from sensorium_amd.ddp import init_rccl
                init_rccl(dev)                                       # RCCL kernels on a hardware queue of their own
        else:
            dist.init_process_group(args.backend)

    import sensorium_amd._lib as L
    from sensorium_amd.argus_models import MouseModel
    from sensorium_amd.synthetic import make_batch

    num_neurons = NUM_NEURONS_ALL[:args.mice]
    expansion = 6 if args.distill else args.expansion          # distillation_001.py: student expansion 6
    params = model_params(expansion, num_neurons)
    params["device"] = str(dev)
    params["amp"] = args.dtype == "bf16"
    params["ddp_comm_dtype"] = args.ddp_comm
    params["ddp_shard_optimizer"] = bool(args.ddp_shard)
    params["ddp_single_rank"] = bool(args.ddp_single_rank)
    torch.manual_seed(1234)            # identical init on every rank (GradBuckets also broadcasts rank 0)
    model = MouseModel(params)
    # reference init rule (src/utils.py:46-56): conv ~ N(0, sqrt(2/fan_out)), BN weight 1 / bias 0
    import math
    for m in model.nn_module.modules():
        if isinstance(m, (torch.nn.Conv1d, torch.nn.Conv3d)):
            fan_out = math.prod(m.kernel_size) * m.out_channels // m.groups
            torch.nn.init.normal_(m.weight, 0, math.sqrt(2.0 / fan_out))
            if m.bias is not None:
                torch.nn.init.zeros_(m.bias)
    model.set_ema(0.999)
    if args.distill:
        tparams = model_params(7, num_neurons)
        tparams["device"] = str(dev)
        teacher = MouseModel(tparams)
        teacher.eval()
        model.distill_model = teacher.nn_module
        model.distill_ratio = 0.36                            # distillation_001.py:67-70
    batch0 = make_batch(args.batch, args.frames, args.height, args.width, num_neurons,
                        seed=20231122 + rank, device=dev)

    loader_it = None
    if args.assembled:
        # synthetic trials in the on-disk layout (video (H, W, L) uint8, behaviour / pupil centre (2, L), responses (N, L): datasets.py:37-51)
        import numpy as np
        from sensorium_amd.data_gpu import BatchAssembler, DeviceBatchLoader, DeviceTrialStore
        drng = np.random.default_rng(20231122 + rank)
        store = DeviceTrialStore(dev)
        for m, n in enumerate(num_neurons):
            for _ in range(6):
                store.add_trial(m, drng.integers(0, 256, size=(args.height, args.width, 300)).astype(np.uint8),
                                (drng.normal(size=(2, 300)) * 5 + 20).astype(np.float32), (drng.normal(size=(2, 300)) * 20 + 80).astype(np.float32),
                                np.maximum(drng.normal(size=(n, 300)), 0).astype(np.float32) * 10)
        asm = BatchAssembler(store, num_neurons, dict(size=args.frames, step=2, position="last"), (args.width, args.height), 0.0,
                             cutmix=dict(alpha=1.0, prob=0.5))
        loader = DeviceBatchLoader(asm, args.batch, args.batch * (args.steps + args.warmup + 8) * 2, seed=rank)
        loader_it = iter(loader)

    def next_batch():
        if loader_it is not None:
            return next(loader_it)
        if not args.distill:
            return batch0
        # the soft-label fill writes into targets / weights in place (argus_models.py:37-41): every step gets a fresh copy
        x0, (t0, w0) = batch0
        return x0, ([t.clone() for t in t0], w0.clone())

    def sync():
        if ddp:
            dist.barrier()
        torch.cuda.synchronize()

    if args.arena_gb > 0:
        arena = torch.empty(int(args.arena_gb * (1 << 30)), dtype=torch.uint8, device=dev)
        del arena
    if args.compute_stream:
        cstream = torch.cuda.Stream(device=dev)
        cstream.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.set_stream(cstream)
    for _ in range(args.warmup):
        model.train_step(next_batch(), sync_loss=False)
    fam_names = L.FAMILIES
    if args.profile_all:
        mask = (1 << len(fam_names)) - 1
    else:
        mask = 1 << fam_names.index(args.roofline_family)
    if args.roctx:
        mask |= 1 << 63                      # include/dwn.h DWN_PROF_ROCTX
    sync()
    L.check(L.lib.dwn_profile_enable(mask, local_rank), "profile_enable")
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = model.train_step(next_batch(), sync_loss=False)
    sync()
    elapsed = time.perf_counter() - t0
    loss_value = float(last["loss"])
    import ctypes as C
    fam_ms = {}
    for i, name in enumerate(fam_names):
        if (mask >> i) & 1:
            ms, n = C.c_double(0), C.c_longlong(0)
            L.check(L.lib.dwn_profile_collect(i, C.byref(ms), C.byref(n)), "profile_collect")
            fam_ms[name] = (ms.value, n.value)
    L.check(L.lib.dwn_profile_enable(0, local_rank), "profile_disable")
    # every other kernel family: HIP events over a few extra, UNTIMED steps (events around every launch perturb the step)
    fam_all, prof_steps = {}, 0
    if not args.no_rooflines:
        # every rank takes these steps (a training step contains the gradient all-reduce: rank 0 alone would leave its
        # collectives unmatched); only rank 0 times its launches
        prof_steps = 3
        if rank == 0:
            L.check(L.lib.dwn_profile_enable((1 << len(fam_names)) - 1, local_rank), "profile_enable")
        for _ in range(prof_steps):
            model.train_step(next_batch(), sync_loss=False)
        torch.cuda.synchronize()
        if rank == 0:
            for i, name in enumerate(fam_names):
                ms, n = C.c_double(0), C.c_longlong(0)
                L.check(L.lib.dwn_profile_collect(i, C.byref(ms), C.byref(n)), "profile_collect")
                fam_all[name] = (ms.value, n.value)
            L.check(L.lib.dwn_profile_enable(0, local_rank), "profile_disable")
    if ddp:
        dist.barrier()
    # SURVEY.md §8d asks for both figures: the same steps without optimizer / EMA (forward + loss + backward only)
    fwd_bwd_clips = None
    if world == 1 and not args.no_fwd_bwd and not args.distill:
        net, inp, tgt = model.nn_module, batch0[0], batch0[1]
        n_fb = max(3, args.steps // 2)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_fb):
            net.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=params["amp"]):
                loss_fb = model.loss(net(inp), tgt)
            loss_fb.backward()
        torch.cuda.synchronize()
        fwd_bwd_clips = args.batch * n_fb / (time.perf_counter() - t1)
    L.lib.dwn_profile_enable(0, local_rank)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if ddp:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        clips = args.batch * world * args.steps
        value = clips / elapsed
        esize = 2 if args.dtype == "bf16" else 4
        fused = [i for i, c in enumerate(CORE_FEATURES)
                 if L.lib.dwn_pw_bwd_fused_supported(L.DWN_BF16 if args.dtype == "bf16" else L.DWN_F32,
                                                     128, c * expansion, c)]
        # blocks trained without a materialised y1 (asked of the library: dwn_block_forward_writes bit 0 clear)
        import sensorium_amd.ops as _ops
        y1_free, hh, ww = [], args.height, args.width
        for i, (cf, st) in enumerate(zip(CORE_FEATURES, STRIDES)):
            ba = L.BlockArgs()
            ba.dtype = L.DWN_BF16 if args.dtype == "bf16" else L.DWN_F32; ba.training = 1; ba.B = args.batch; ba.T = args.frames
            ba.Hin, ba.Win = hh, ww; ba.Hout, ba.Wout = (hh - 1) // st + 1, (ww - 1) // st + 1
            ba.Cin = cf; ba.Cmid = cf * expansion; ba.Cout = CORE_FEATURES[min(i + 1, len(CORE_FEATURES) - 1)]
            ba.stride = st; ba.ks = 3; ba.kt = 5; ba.se_r = max(1, cf * expansion // 32); ba.y1_mode = _ops._Y1_MODE
            if not (L.lib.dwn_block_forward_writes(C.byref(ba)) & 1):
                y1_free.append(i)
            hh, ww = ba.Hout, ba.Wout
        alg, flops = family_work(args.batch, args.frames, args.height, args.width, expansion, num_neurons, fused, esize, y1_free)
        mfma_peak = MFMA_PEAK_TFLOPS if args.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
        # HBM bytes per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE cannot be collected inside the timed run):
        # profiles/<round>_pmc_traffic.json holds them for the default metric shape together with the hash of the
        # library they were measured on; a different build makes them stale and they are not reported
        default_shape = (args.batch, args.frames, args.height, args.width, expansion, args.dtype, args.mice,
                         args.distill) == (32, 32, 36, 64, 7, "bf16", 1, False)
        traffic = {}
        traffic_src, traffic_stale = None, None
        for tpath in sorted((ROOT / "profiles").glob("r*_pmc_traffic.json"), reverse=True):
            try:
                tj = json.loads(tpath.read_text())
            except ValueError:
                continue
            if "lib_sha16" not in tj:
                continue
            traffic_stale = tj["lib_sha16"] != lib_sha16()
            if default_shape and not traffic_stale:
                traffic = tj["families"]
                traffic_src = f"profiles/{tpath.name} (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; same library build)"
            break

        def roof_of(fam, ms, n, steps):
            if n <= 0 or fam not in alg:
                return None
            lps = n / steps
            avg_ms = ms / n
            bpl = alg[fam] / lps
            ach = bpl / (avg_ms * 1e-3) / 1e9
            r = {"kernel": fam, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "launches_per_step": lps,
                 "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(bpl)}
            if flops.get(fam):
                tf = flops[fam] / lps / (avg_ms * 1e-3) / 1e12
                r["mfma"] = {"achieved": round(tf, 1), "peak": mfma_peak, "unit": "TFLOP/s", "frac": round(tf / mfma_peak, 4),
                             "flops_per_launch": int(flops[fam] / lps)}
                # compute-bound when the arithmetic intensity exceeds the ridge (peak FLOP/s / peak B/s)
                if flops[fam] / alg[fam] > mfma_peak * 1e12 / (HBM_PEAK_GBS * 1e9):
                    # the headline numbers of a compute-bound family are its MFMA ones; the HBM view moves to "hbm"
                    r["hbm"] = {k: r[k] for k in ("achieved", "peak", "unit", "frac")}
                    r.update({"bound": "mfma", **{k: r["mfma"][k] for k in ("achieved", "peak", "unit", "frac")}})
            tf_ = traffic.get(fam)
            if tf_:
                # measured HBM bytes of the family per step (PMC) over this run's launches; `traffic` = the rate they moved at
                tbl = tf_["traffic_bytes_per_step"] / lps if "traffic_bytes_per_step" in tf_ else tf_["traffic_bytes_per_launch"]
                r["traffic"] = round(tbl / (avg_ms * 1e-3) / 1e9, 1)
                r["traffic_bytes_per_launch"] = int(tbl)
                r["traffic_frac_of_peak"] = round(tbl / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                r["traffic_source"] = traffic_src
            elif traffic_stale:
                r["traffic_stale"] = True
            return r

        roof = None
        fam = args.roofline_family
        if fam in fam_ms:
            roof = roof_of(fam, fam_ms[fam][0], fam_ms[fam][1], args.steps)
        rooflines = [r for r in (roof_of(k, v[0], v[1], prof_steps) for k, v in fam_all.items()) if r is not None]
        out = {
            "metric": "training clips/sec/GPU (DwiseNeuro fwd+bwd, B=32 T=32 36x64) at 1/2/4/8 GPUs",
            "value_is": "whole-job aggregate clips/s over all n_gpus (per-GPU figure: clips_per_s_per_gpu); a step "
                        "includes loss, optimizer (fused AdamW), EMA and, for n_gpus > 1, the gradient all-reduce; the loss scalar is "
                        "read back once after the timed region (the reference calls loss.item() every step, "
                        "argus_models.py:56): no arithmetic is skipped and the host stays ~4x ahead of the GPU",
            "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload_name(args),
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world, "frames": args.frames,
                       "height": args.height, "width": args.width, "parallelism": f"dp{world}",
                       "y1_free_blocks": y1_free},
            "clips_per_s_per_gpu": round(value / world, 2), "loss": round(loss_value, 3),
            "clips_per_s_fwd_bwd_only": None if fwd_bwd_clips is None else round(fwd_bwd_clips, 2),
            "roofline": roof,
            "rooflines": rooflines,
            "rooflines_note": "roofline = the dominant family, HIP events inside the timed region; rooflines = every family "
                              "(HBM fraction of its algorithmic bytes; GEMM families also their MFMA fraction) from "
                              f"{prof_steps} extra untimed steps with events around every launch",
            "rccl_ranks": dist.get_world_size() if (ddp and dist.is_initialized()) else 1,
        }
        if model.buckets is not None:
            out["ddp"] = {"backend": dist.get_backend(), "buckets": len(model.buckets.buckets),
                          "bucket_mb": [round(b["flat"].numel() * 4 / 2 ** 20, 1) for b in model.buckets.buckets],
                          "comm_dtype": args.ddp_comm, "sharded_readout_optimizer": bool(model.buckets.shard),
                          "ring_bytes_sent_per_rank_per_step": model.buckets.bytes_on_wire_per_step(),
                          "gradients": "written by the HIP backward straight into the flat buckets; all-reduce (mean) per "
                                       "bucket launched from autograd hooks, overlapped with the rest of backward"}
        if args.profile_all:
            out["family_ms_per_step"] = {k: round(v[0] / args.steps, 3) for k, v in fam_ms.items()}
        elif fam_all:
            out["family_ms_per_step"] = {k: round(v[0] / prof_steps, 3) for k, v in fam_all.items()}
        if world == 1 and default_shape and not (args.no_inference and args.no_other_configs):
            del model, batch0, last            # the legs below run after the timed region, on a freed device
            torch.cuda.empty_cache()
        if world == 1 and not args.no_inference and default_shape:
            # BASELINE.json configs[4] (scripts/predict.py:44-50 + src/predictors.py:36-55); fp32 is the reference's prediction
            # precision (src/argus_models.py:89-99)
            sys.path.insert(0, str(ROOT / "tools"))
            from bench_predict import ensemble_bench
            inf = {"workload": "7-fold ensemble, one 300-frame trial at 64x64, window 16 step 2 (270 windows), 90 windows per "
                               "forward (three exact batches; the reference runs one window per forward, predictors.py:46-54 — "
                               "the batch size does not change the result), all folds in one captured hipGraph per window batch, "
                               "blend on the device",
                   "fp32_products": "bf16x3 (the default of DwiseNeuro.set_fp32_eval_products: every fp32 GEMM operand split into bf16 "
                                    "hi + lo, three bf16 MFMAs with fp32 accumulation; 5.8e-7 relative L2 from the native fp32 MFMA "
                                    "on the full-width eval forward; 'native' selects v_mfma_f32_16x16x4_f32)",
                   "hbm_frac_is": "bytes the eval-mode pass structure executes (tools/bench_predict.py::eval_executed_bytes) "
                                  "/ time / 8 TB/s"}
            for dt_ in ("bf16", "fp32"):
                inf[dt_] = ensemble_bench(dtype=dt_, device=dev, windows=90)
            for ptraf in sorted((ROOT / "profiles").glob("r*_predict_pmc.json"), reverse=True):
                try:
                    pj = json.loads(ptraf.read_text())
                except ValueError:
                    continue
                if pj.get("lib_sha16") == lib_sha16():      # measured on this very library build, else not reported
                    inf["pmc_traffic"] = pj
                break
            out["inference"] = inf
        if world == 1 and not args.no_other_configs and default_shape:
            # BASELINE.json configs[2] and configs[3] on this one GPU (their 8-GPU form is the same step per rank + the gradient
            # exchange): each as a fresh child process of this script, same shape and dtype, 10 timed steps
            torch.cuda.empty_cache()
            asm_leg = other_config(["--assembled"])
            out["clips_per_s_with_assembly"] = asm_leg.get("clips_per_s")
            out["with_assembly"] = dict(asm_leg, note="the metric step with every batch drawn and assembled on the device inside the timed "
                                        "region (DeviceBatchLoader: frame stack + pad + CutMix + sparse targets from trials resident in HBM, "
                                        "host-side pick drawing included); the headline value reuses one resident batch")
            out["other_configs"] = {
                "ten_readouts": other_config(["--mice", "10"]),
                "distillation": other_config(["--mice", "10", "--distill"]),
                "note": "NOT the metric config: same clip shape / dtype / step definition with the ten per-mouse readouts of "
                        "src/constants.py:38 (configs[2]) and the configs/distillation_001.py step (configs[3]); frac = the "
                        "dominant kernel family's HBM fraction in that run"}
        if world == 1 and not args.no_cpu_baseline and args.mice == 1 and not args.distill:
            out["cpu_baseline"] = cpu_baseline(args.frames, args.height, args.width, args.expansion)
        print(json.dumps(out), flush=True)
    if ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
