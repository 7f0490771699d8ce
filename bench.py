#!/usr/bin/env python3
"""Benchmark of the DwiseNeuro training hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

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
            "drop-path 0.1), AdamW + EMA")


def block_shapes(batch, frames, height, width, expansion):
    """(M_in, M_out, Cmid) per block."""
    out = []
    h, w = height, width
    for c, s in zip(CORE_FEATURES, STRIDES):
        ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
        out.append((batch * frames * h * w, batch * frames * ho * wo, c * expansion))
        h, w = ho, wo
    return out


# algorithmic bytes per *step* of each timed kernel family (read input once + write output once;
# backward = read x + read dy + write dx), in elements of the storage dtype
def family_algorithmic_elems(shapes):
    e = {}
    e["dws_fwd"] = sum(mi * c + mo * c for mi, mo, c in shapes)
    e["dwt_fwd"] = sum(2 * mo * c for mi, mo, c in shapes)
    e["dws_bwd"] = sum(mi * c + mo * c + mi * c for mi, mo, c in shapes)
    e["dwt_bwd"] = sum(3 * mo * c for mi, mo, c in shapes)
    return e


def cpu_baseline(frames, height, width, expansion):
    """The CPU oracle (a port: oracle/dwiseneuro_oracle.py, pinned to the reference by tests/golden) timed on this
    host on a BOUNDED sample of the benchmark workload: fwd + loss + bwd of B=1 clip at the benchmark's HxW and
    width, first with T=8 frames; if that took < 6 s the full T-frame clip is timed instead, repeated until about
    12 s of CPU work are sampled.  clips/s is scaled by the fraction of a clip processed (the path is linear in T)."""
    from oracle import dwiseneuro_oracle as orc
    import numpy as np
    orc.DW_IMPL = "library"        # depth-wise convs through torch's conv3d, like the reference's CPU path
    threads = min(os.cpu_count() or 1, 32)      # more threads than this only adds contention for these sizes
    torch.set_num_threads(threads)
    sd = orc.make_state_dict(readout_outputs=(NUM_NEURONS_MOUSE0,), expansion_ratio=expansion, seed=0)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and "inv_freq" not in k
              else v) for k, v in sd.items()}
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.normal(size=(1, 5, frames, height, width)).astype(np.float32) * 40 + 80)
    target = torch.from_numpy(np.maximum(rng.normal(size=(1, NUM_NEURONS_MOUSE0, frames)), 0).astype(np.float32))
    w = torch.ones(1, 1)

    def step(t):
        for v in sd.values():
            if getattr(v, "grad", None) is not None:
                v.grad = None
        preds = orc.forward(sd, x[:, :, :t], strides=STRIDES, readout_outputs=(NUM_NEURONS_MOUSE0,), training=True)
        loss = orc.mice_poisson_loss(preds, [target[:, :, :t]], w)
        loss.backward()

    step(2)                                    # tiny warm-up (thread pool, allocator)
    t_s = min(8, frames)
    t0 = time.perf_counter()
    step(t_s)
    dt = time.perf_counter() - t0
    if dt < 6.0 and t_s < frames:
        t_s = frames
        t0 = time.perf_counter()
        step(t_s)
        dt = time.perf_counter() - t0
    reps = 1
    if t_s == frames and dt < 10.0:
        # fast host: repeat the full clip until ~12 s of CPU work are sampled (at most 8 repetitions)
        extra = min(int(12.0 / max(dt, 1e-3)), 7)
        t0 = time.perf_counter()
        for _ in range(extra):
            step(t_s)
        dt += time.perf_counter() - t0
        reps += extra
    return {"value": round(reps * (t_s / frames) / dt, 4), "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": f"{reps} fwd+loss+bwd step(s) of the CPU oracle on B=1 clip, {t_s} of {frames} frames, "
                      f"{height}x{width}, expansion {expansion}, 1 readout, fp32, {dt:.1f} s in total; scaled to full clips"}


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
    ap.add_argument("--profile-all", action="store_true", help="time every kernel family (adds event overhead)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for 1-GPU testing)")
    ap.add_argument("--share-device", action="store_true", help="testing only: every rank uses cuda:0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
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

    def next_batch():
        if not args.distill:
            return batch0
        # the soft-label fill writes into targets / weights in place (argus_models.py:37-41): every step gets a fresh copy
        x0, (t0, w0) = batch0
        return x0, ([t.clone() for t in t0], w0.clone())

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        model.train_step(next_batch(), sync_loss=False)
    fam_names = L.FAMILIES
    if args.profile_all:
        mask = (1 << len(fam_names)) - 1
    else:
        mask = 1 << fam_names.index(args.roofline_family)
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
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        clips = args.batch * world * args.steps
        value = clips / elapsed
        esize = 2 if args.dtype == "bf16" else 4
        shapes = block_shapes(args.batch, args.frames, args.height, args.width, expansion)
        alg = family_algorithmic_elems(shapes)
        roof = None
        fam = args.roofline_family
        if fam in fam_ms and fam in alg and fam_ms[fam][1] > 0:
            ms, n = fam_ms[fam]
            launches_per_step = n / args.steps
            bytes_per_launch = alg[fam] * esize / launches_per_step
            avg_ms = ms / n
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            roof = {"kernel": fam, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                    "launches_per_step": launches_per_step, "avg_launch_ms": round(avg_ms, 4),
                    "algorithmic_bytes_per_launch": int(bytes_per_launch)}
            # HBM bytes per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE cannot be collected inside the timed
            # run): read from the committed summary of the same command at the default metric shape, see
            # tools/pmc_traffic.py; expressed like `achieved` (bytes per launch / this run's launch duration)
            default_shape = (args.batch, args.frames, args.height, args.width, expansion, args.dtype, args.mice) == \
                            (32, 32, 36, 64, 7, "bf16", 1)
            tpath = ROOT / "profiles" / "r1f_pmc_traffic.json"
            if default_shape and tpath.exists():
                try:
                    tf = json.loads(tpath.read_text())["families"].get(fam)
                    if tf:
                        roof["traffic"] = round(tf["traffic_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9, 1)
                        roof["traffic_bytes_per_launch"] = int(tf["traffic_bytes_per_launch"])
                        roof["traffic_source"] = "profiles/r1f_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE)"
                except (ValueError, KeyError):
                    pass
        out = {
            "metric": "training clips/sec/GPU (DwiseNeuro fwd+bwd, B=32 T=32 36x64) at 1/2/4/8 GPUs",
            "value_is": "whole-job aggregate clips/s over all n_gpus (per-GPU figure: clips_per_s_per_gpu); a step "
                        "includes loss, optimizer (fused AdamW), EMA and, for n_gpus > 1, the gradient all-reduce",
            "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload_name(args),
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world, "frames": args.frames,
                       "height": args.height, "width": args.width, "parallelism": f"dp{world}"},
            "clips_per_s_per_gpu": round(value / world, 2), "loss": round(loss_value, 3),
            "clips_per_s_fwd_bwd_only": None if fwd_bwd_clips is None else round(fwd_bwd_clips, 2),
            "roofline": roof,
        }
        if args.profile_all:
            out["family_ms_per_step"] = {k: round(v[0] / args.steps, 3) for k, v in fam_ms.items()}
        if world == 1 and not args.no_cpu_baseline and args.mice == 1 and not args.distill:
            out["cpu_baseline"] = cpu_baseline(args.frames, args.height, args.width, args.expansion)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
