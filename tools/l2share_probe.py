#!/usr/bin/env python3
"""Experiment (round 5): what would the spatial backward cost if its widest input were a 7x narrower tensor shared by all channel
slices (the block input a0, from which y1 can be rebuilt by MFMA)?  Built with -DEXP_SHARED_Y1 the kernels read channels 0..63 of a
[M][64] tensor for every slice instead of their own slice of y1 — the HBM / L2 access pattern of the rebuilt-y1 design without its
arithmetic; -DEXP_XCD_MAP puts the slices of a plane group on one XCD.  Timing only (results are garbage by construction).
usage: DWN_LIB_PATH=build_ab/<variant>/libdwiseneuro_hip.so python3 tools/l2share_probe.py [shared]"""
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import sensorium_amd._lib as L

dev = torch.device("cuda", 0)
BF = torch.bfloat16
shared = "shared" in sys.argv[1:]


def desc(p, ld, **kw):
    d = L.LoadDesc()
    d.p = p.data_ptr(); d.ld = ld; d.rows_per_sample = 1
    for k, v in kw.items():
        setattr(d, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    return d


def run(planes, Hin, Win, Cc, stride, cin):
    g = torch.Generator(device=dev); g.manual_seed(0)
    Hout, Wout = (Hin - 1) // stride + 1, (Win - 1) // stride + 1
    ycols = cin if shared else Cc
    y1 = torch.randn(planes * Hin * Win, ycols, device=dev, generator=g).to(BF)
    dh2 = torch.randn(planes * Hout * Wout, Cc, device=dev, generator=g).to(BF)
    y2 = torch.randn(planes * Hout * Wout, Cc, device=dev, generator=g).to(BF)
    coef = torch.cat([torch.rand(Cc, device=dev, generator=g) + 0.5, torch.randn(Cc, device=dev, generator=g) * 0.3,
                      torch.randn(Cc, device=dev, generator=g) * 0.2, torch.rand(Cc, device=dev, generator=g) + 0.5])
    abc = torch.randn(3 * Cc, device=dev, generator=g) * 0.5
    w = torch.randn(9, Cc, device=dev, generator=g) / 3.0
    dh1 = torch.empty(planes * Hin * Win, Cc, device=dev, dtype=BF)
    dw = torch.zeros(Cc, 9, device=dev)
    st = torch.zeros(32 * 2 * Cc, dtype=torch.float64, device=dev)
    a = L.DwSpatialBwdArgs()
    a.dy = desc(dh2, Cc, q=y2, v1=abc, v2=abc[Cc:], v3=abc[2 * Cc:])
    a.y1 = desc(y1, ycols, v1=coef, v2=coef[Cc:], v3=coef[2 * Cc:], v4=coef[3 * Cc:])
    a.w = w.data_ptr(); a.dh1 = dh1.data_ptr(); a.dw = dw.data_ptr(); a.planes = planes; a.Hin = Hin; a.Win = Win
    a.Hout = Hout; a.Wout = Wout; a.C = Cc; a.stride = stride; a.ks = 3; a.stats = st.data_ptr()
    s = torch.cuda.current_stream().cuda_stream

    def fn():
        L.check(L.lib.dwn_dw_spatial_bwd(C.byref(a), L.DWN_BF16, 0, s), "dwsb")
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if os.environ.get("PROBE_QUICK"):          # under a counter pass: the three launches above are the sample
        return
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5 * 1e3)
    moved = (y1.numel() + dh1.numel() + 2 * y2.numel()) * 2
    print(f"{'shared' if shared else 'own   '} planes={planes} {Hin}x{Win} C={Cc} s={stride}: {min(ts):7.1f} us (min of 3x5)  "
          f"{moved / min(ts) / 1e6:5.2f} TB/s of the bytes this variant moves", flush=True)


if __name__ == "__main__":
    for cfg in ((1024, 18, 32, 448, 1, 64), (1024, 36, 64, 448, 2, 64), (1024, 18, 32, 896, 2, 128), (1024, 9, 16, 896, 1, 128)):
        run(*cfg)
