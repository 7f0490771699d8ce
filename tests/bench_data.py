#!/usr/bin/env python3
"""Measurement tool, not a test (kept under tests/ because its CPU leg times the oracle, which only tests/, smoke() and
bench.py may import).  Batch assembly at the metric shape (B=32 clips of T=32 frames, 36x64, ten mice): device kernels vs the numpy oracle
(the reference's per-sample CPU work, restated) on this host.  Prints one JSON line.

    python tests/bench_data.py [--iters 50] [--mice 10]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root

NUM_NEURONS = [7863, 7908, 8202, 7939, 8122, 7440, 7928, 8285, 7671, 7495]        # src/constants.py:24,31


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--mice", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=32)
    args = ap.parse_args()
    from oracle import data_oracle as dorc
    from sensorium_amd.data_gpu import BatchAssembler, DeviceTrialStore
    n_neurons = NUM_NEURONS[:args.mice]
    rng = np.random.default_rng(0)
    store = DeviceTrialStore("cuda:0")
    host = {}
    length = 300
    for m, n in enumerate(n_neurons):
        host[m] = []
        for _ in range(4):
            d = dict(video=rng.integers(0, 256, size=(36, 64, length)).astype(np.uint8),
                     behavior=rng.normal(size=(2, length)).astype(np.float32),
                     pupil_center=rng.normal(size=(2, length)).astype(np.float32),
                     responses=rng.normal(size=(n, length)).astype(np.float32))
            host[m].append(d)
            store.add_trial(m, d["video"], d["behavior"], d["pupil_center"], d["responses"])
    fs = dict(size=args.frames, step=2, position="last")
    asm = BatchAssembler(store, n_neurons, fs, (64, 36), 0.0, cutmix=dict(alpha=1.0, prob=0.5))
    rs = np.random.RandomState(0)
    mice = [b % args.mice for b in range(args.batch)]
    batches = [asm.draw_train_picks(rs, mice) for _ in range(8)]
    for p in batches:
        asm.assemble(p)
    torch.cuda.synchronize()
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    beg.record()
    for i in range(args.iters):
        asm.assemble(batches[i % len(batches)])
    end.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.iters * 1e3
    gpu_ms = beg.elapsed_time(end) / args.iters
    x, (targets, w) = asm.assemble(batches[0])
    out_bytes = x.numel() * 4 + sum(t.numel() for t in targets) * 4 + w.numel() * 4
    # CPU: the same picks through the oracle, single process (the reference spreads this over 8 DataLoader workers)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 5.0:
        p = batches[reps % len(batches)]
        ref = dorc.assemble_batch(host, [(q.mouse, q.trial, q.end_frame, q.mix) for q in p], n_neurons, (64, 36), 0.0,
                                  (args.frames, 2), [q.box for q in p])
        reps += 1
    cpu_ms = (time.perf_counter() - t0) / reps * 1e3
    p = batches[(reps - 1) % len(batches)]
    got = asm.assemble(p)
    torch.cuda.synchronize()
    exact = bool(np.array_equal(got[0].cpu().numpy(), ref[0])) and all(
        np.array_equal(a.cpu().numpy(), b) for a, b in zip(got[1][0], ref[1][0]))
    print(json.dumps({"workload": f"batch assembly B={args.batch} T={args.frames} 36x64, {args.mice} mice",
                      "device_ms_per_batch": round(gpu_ms, 4), "host_wall_ms_per_batch": round(wall, 4),
                      "bytes_written": out_bytes, "device_write_GBps": round(out_bytes / gpu_ms / 1e6, 1),
                      "cpu_oracle_ms_per_batch": round(cpu_ms, 2), "cpu_threads": 1,
                      "h2d_bytes_reference": out_bytes, "bit_exact_vs_oracle": exact}))


if __name__ == "__main__":
    main()
