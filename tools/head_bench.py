"""Cortex + readout head alone (dwiseneuro.py:195-287) at the metric shapes: B*T = 1024 rows,
256 -> 1024 -> 2048 -> 4096 channels (groups 2), one readout of 7863 neurons.  Times forward and backward of each module with
HIP events on the current stream; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.

    python tools/head_bench.py [--iters 50] [--mice 1] [--dtype bf16]
"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from sensorium_amd.dwiseneuro import Cortex, Readout


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--cin", type=int, default=256)
    ap.add_argument("--neurons", type=int, default=7863)
    ap.add_argument("--drop", type=float, default=0.4)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    cortex = Cortex(a.cin, (1024, 2048, 4096), groups=2).to(dev).train()
    readout = Readout(4096, a.neurons, groups=2, softplus_beta=0.07, drop_rate=a.drop).to(dev).train()
    x = torch.randn(a.batch, a.frames, a.cin, device=dev).to(dt).requires_grad_(True)
    M = a.batch * a.frames

    feats = cortex(x, dt)
    t_cf = timed(lambda: cortex(x, dt), a.iters)
    g = torch.randn_like(feats)

    def cortex_fb():
        y = cortex(x, dt)
        y.backward(g)
    t_cfb = timed(cortex_fb, a.iters)

    f2 = feats.detach().requires_grad_(True)
    t_rf = timed(lambda: readout(f2), a.iters)
    out = readout(f2)
    go = torch.randn_like(out)

    def readout_fb():
        o = readout(f2)
        o.backward(go)
    t_rfb = timed(readout_fb, a.iters)

    cx = [a.cin, 1024, 2048, 4096]
    f_c = sum(2 * M * p * q // 2 for p, q in zip(cx[:-1], cx[1:]))
    npad = (a.neurons + 1) // 2 * 2
    f_r = 2 * M * 2048 * npad
    print(f"cortex  fwd {t_cf:8.1f} us   fwd+bwd {t_cfb:8.1f} us   (GEMM flops fwd {f_c / 1e9:.2f} G: "
          f"{f_c / t_cf / 1e6:.0f} TFLOP/s fwd, {3 * f_c / t_cfb / 1e6:.0f} fwd+bwd)")
    print(f"readout fwd {t_rf:8.1f} us   fwd+bwd {t_rfb:8.1f} us   (GEMM flops fwd {f_r / 1e9:.2f} G: "
          f"{f_r / t_rf / 1e6:.0f} TFLOP/s fwd, {3 * f_r / t_rfb / 1e6:.0f} fwd+bwd)")


if __name__ == "__main__":
    main()
