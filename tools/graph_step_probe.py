#!/usr/bin/env python3
"""Is a hipGraph-captured training step (forward + loss + backward as one graph, fused AdamW + EMA eager) faster than the eager
step?  Development probe for MouseModel's graph mode: python tools/graph_step_probe.py [mice]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

import bench
from sensorium_amd.argus_models import MouseModel
from sensorium_amd.synthetic import make_batch

mice = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
params = bench.model_params(7, bench.NUM_NEURONS_ALL[:mice])
params["device"] = "cuda:0"; params["amp"] = True
torch.manual_seed(0)
model = MouseModel(params)
model.set_ema(0.999)
batch = make_batch(32, 32, 36, 64, bench.NUM_NEURONS_ALL[:mice], device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager = timeit(lambda: model.train_step(batch, sync_loss=False))
print(f"eager step {eager:.3f} ms", flush=True)
# ---- graph: forward + loss + backward
net = model.nn_module
x, (targets, weights) = batch
active = model._active_samples(batch)
for m, r in enumerate(net.readouts):
    r._dwn_active = None if active is None else active[m]
model.optimizer.zero_grad(set_to_none=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        model.optimizer.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = model.loss(net(x), (targets, weights))
        loss.backward()
torch.cuda.current_stream().wait_stream(s)
model.optimizer.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        sloss = model.loss(net(x), (targets, weights))
    sloss.backward()
torch.cuda.synchronize()


def gstep():
    g.replay()
    model.optimizer.step()


graphed = timeit(gstep)
print(f"graph replay + optimizer {graphed:.3f} ms  (loss {float(sloss):.3f})")
print(f"eager again {timeit(lambda: model.train_step(batch, sync_loss=False)):.3f} ms")
