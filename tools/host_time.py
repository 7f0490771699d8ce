#!/usr/bin/env python3
"""Host enqueue time of one training step vs its GPU time (development tool): is the host ahead of the GPU?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from sensorium_amd.argus_models import MouseModel
from sensorium_amd.synthetic import make_batch

dev = torch.device("cuda", 0)
params = bench.model_params(7)
params["device"] = "cuda:0"
params["amp"] = True
torch.manual_seed(0)
model = MouseModel(params)
model.set_ema(0.999)
batch = make_batch(32, 32, 36, 64, (7863,), seed=1, device=dev)
for _ in range(3):
    model.train_step(batch, sync_loss=False)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(10):
    a = time.perf_counter()
    model.train_step(batch, sync_loss=False)
    host.append(time.perf_counter() - a)
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / 10
print(f"step wall {total*1e3:.2f} ms; host enqueue per step: min {min(host)*1e3:.2f} median {sorted(host)[5]*1e3:.2f} max {max(host)*1e3:.2f} ms")
# forward-only and backward-only host time (GPU drained in between, so these are pure enqueue costs)
net = model.nn_module
torch.cuda.synchronize()
a = time.perf_counter()
with torch.autocast("cuda", dtype=torch.bfloat16):
    loss = model.loss(net(batch[0]), batch[1])
fwd = time.perf_counter() - a
torch.cuda.synchronize()
a = time.perf_counter()
loss.backward()
bwd = time.perf_counter() - a
torch.cuda.synchronize()
print(f"host enqueue with an empty queue: forward+loss {fwd*1e3:.2f} ms, backward {bwd*1e3:.2f} ms")

# per autograd Function: host time of forward / backward calls (no synchronisation inside), averaged over 5 steps
from sensorium_amd import ops
acc = {}


def wrap(cls, name):
    for which in ("forward", "backward"):
        fn = getattr(cls, which)

        def timed(*a, _fn=fn, _key=f"{name}.{which}", **k):
            t = time.perf_counter()
            r = _fn(*a, **k)
            d = acc.setdefault(_key, [0.0, 0])
            d[0] += time.perf_counter() - t
            d[1] += 1
            return r
        setattr(cls, which, staticmethod(timed))


for cls, name in ((ops.StemFn, "stem"), (ops.BlockFn, "block"), (ops.PoolFn, "pool"), (ops.CortexFn, "cortex"),
                  (ops.ReadoutFn, "readout"), (ops.PoissonLossFn, "poisson")):
    wrap(cls, name)
for _ in range(5):
    model.train_step(batch, sync_loss=False)
torch.cuda.synchronize()
for k, (t, n) in sorted(acc.items()):
    print(f"{k:18s} {t / n * 1e6:8.1f} us per call  x {n // 5} calls per step = {t / 5 * 1e3:6.2f} ms per step")
