"""Step time of the metric configuration with a communication-like kernel resident beside it (tools/ubench/hog.hip): `wgs`
workgroups x 256 threads holding CU slots on a second stream for the whole measurement — the worst case of an RCCL all-reduce
overlapping the whole step.  One GPU; build the hog first (command in hog.hip).

    python tools/hog_bench.py [--wgs 0 16 32 64 128] [--steps 10]
"""
import argparse
import ctypes as C
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch

from bench import model_params
from sensorium_amd.argus_models import MouseModel
from sensorium_amd.synthetic import make_batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wgs", type=int, nargs="*", default=[0, 16, 32, 64, 128])
    ap.add_argument("--threads", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    hog = C.CDLL(str(ROOT / "tools" / "ubench" / "libhog.so"))
    hog.hog_launch.argtypes = [C.c_int, C.c_int, C.c_double, C.c_void_p]
    dev = torch.device("cuda:0")
    p = model_params(7)
    p["device"] = str(dev); p["amp"] = True
    model = MouseModel(p)
    model.set_ema(0.999)
    batch = make_batch(32, 32, 36, 64, (7863,), device=dev)
    for _ in range(4):
        model.train_step(batch, sync_loss=False)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for wgs in a.wgs:
        est_ms = 40.0 * (a.steps + 3)
        if wgs:
            rc = hog.hog_launch(wgs, a.threads, est_ms, side.cuda_stream)
            assert rc == 0, rc
            time.sleep(0.02)                         # the hog is resident before the steps start
        for _ in range(2):
            model.train_step(batch, sync_loss=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            model.train_step(batch, sync_loss=False)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / a.steps
        torch.cuda.synchronize()                     # lets the hog run out
        print(f"hog {wgs:4d} workgroups x {a.threads}: {ms:.2f} ms/step", flush=True)


if __name__ == "__main__":
    main()
