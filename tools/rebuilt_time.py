#!/usr/bin/env python3
"""Stand-alone launch times of the spatial stencils at the metric shapes of blocks 0-3, stored y1 vs y1 rebuilt from a0 (round 5).
usage: python3 tools/rebuilt_time.py [fwd] [bwd]"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import sensorium_amd._lib as L

dev = torch.device("cuda", 0)
BF = torch.bfloat16
s = lambda: torch.cuda.current_stream().cuda_stream


def desc(p, ld, **kw):
    d = L.LoadDesc()
    d.p = p.data_ptr() if p is not None else None; d.ld = ld; d.rows_per_sample = 1
    for k, v in kw.items():
        setattr(d, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    return d


def timeit(fn, n=8, reps=3):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


def run(planes, Hin, Win, Cc, stride, which, cin=64, rows_band=0):
    g = torch.Generator(device=dev); g.manual_seed(0)
    Hout, Wout = (Hin - 1) // stride + 1, (Win - 1) // stride + 1
    a0 = torch.randn(planes * Hin * Win, cin, device=dev, generator=g).to(BF)
    w1 = (torch.randn(Cc, cin, device=dev, generator=g) / cin ** 0.5).to(BF)
    y1 = (a0.float() @ w1.float().t()).to(BF)
    coef = torch.cat([torch.rand(Cc, device=dev, generator=g) + 0.5, torch.randn(Cc, device=dev, generator=g) * 0.3,
                      torch.randn(Cc, device=dev, generator=g) * 0.2, torch.rand(Cc, device=dev, generator=g) + 0.5])
    w = torch.randn(9, Cc, device=dev, generator=g) / 3.0
    st = torch.zeros(32 * 2 * Cc, dtype=torch.float64, device=dev)
    res = {}
    if which == "fwd":
        y2 = torch.empty(planes * Hout * Wout, Cc, dtype=BF, device=dev)
        for mode in ("stored", "rebuilt"):
            a = L.DwSpatialFwdArgs()
            di = desc(y1 if mode == "stored" else None, Cc, v1=coef, v2=coef[Cc:]); di.act = 1
            a.inp = di; a.w = w.data_ptr(); a.out = y2.data_ptr(); a.planes = planes; a.Hin = Hin; a.Win = Win; a.Hout = Hout
            a.Wout = Wout; a.C = Cc; a.stride = stride; a.ks = 3; a.stats = st.data_ptr(); a.rows_band = rows_band
            if mode == "rebuilt":
                a.a0 = a0.data_ptr(); a.a0_ld = cin; a.w1 = w1.data_ptr(); a.Cin = cin
            res[mode] = timeit(lambda: L.check(L.lib.dwn_dw_spatial_fwd(C.byref(a), L.DWN_BF16, 0, s()), "fwd"))
    else:
        dh2 = torch.randn(planes * Hout * Wout, Cc, device=dev, generator=g).to(BF)
        y2 = torch.randn(planes * Hout * Wout, Cc, device=dev, generator=g).to(BF)
        abc = torch.randn(3 * Cc, device=dev, generator=g) * 0.5
        dh1 = torch.empty(planes * Hin * Win, Cc, dtype=BF, device=dev)
        dw = torch.zeros(Cc, 9, device=dev)
        for mode in ("stored", "rebuilt"):
            a = L.DwSpatialBwdArgs()
            a.dy = desc(dh2, Cc, q=y2, v1=abc, v2=abc[Cc:], v3=abc[2 * Cc:])
            a.y1 = desc(y1 if mode == "stored" else None, Cc, v1=coef, v2=coef[Cc:], v3=coef[2 * Cc:], v4=coef[3 * Cc:])
            a.w = w.data_ptr(); a.dh1 = dh1.data_ptr(); a.dw = dw.data_ptr(); a.planes = planes; a.Hin = Hin; a.Win = Win
            a.Hout = Hout; a.Wout = Wout; a.C = Cc; a.stride = stride; a.ks = 3; a.stats = st.data_ptr(); a.rows_band = rows_band
            if mode == "rebuilt":
                a.a0 = a0.data_ptr(); a.a0_ld = cin; a.w1 = w1.data_ptr(); a.Cin = cin
            res[mode] = timeit(lambda: L.check(L.lib.dwn_dw_spatial_bwd(C.byref(a), L.DWN_BF16, 0, s()), "bwd"))
    print(f"{which} planes={planes} {Hin}x{Win} C={Cc} s={stride} band={rows_band}: stored {res['stored']:7.1f} us   rebuilt {res['rebuilt']:7.1f} us", flush=True)


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if a in ("fwd", "bwd")] or ["fwd", "bwd"]
    bands = [int(a) for a in sys.argv[1:] if a.isdigit() and a != '128'] or [0]
    for wh in which:
        for rb in bands:
            run(1024, 18, 32, 448, 1, wh, rows_band=rb)
            run(1024, 36, 64, 448, 2, wh, rows_band=rb)
            if "128" in sys.argv:
                run(1024, 18, 32, 896, 2, wh, cin=128, rows_band=rb)
                run(1024, 9, 16, 896, 1, wh, cin=128, rows_band=rb)
