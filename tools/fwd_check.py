#!/usr/bin/env python3
"""Development tool: row-walk spatial forward (dwn_dwfwd.hip) against the pair kernel it replaces, through
dwn_dw_spatial_fwd with DWN_DWS_WALK_OFF toggled per call: y2 equality, BN-sum agreement, launch times."""
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import sensorium_amd._lib as L

dev = torch.device("cuda", 0)
BF = torch.bfloat16


def stream():
    return torch.cuda.current_stream().cuda_stream


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def desc(p, ld, **kw):
    d = L.LoadDesc()
    d.p = p.data_ptr(); d.ld = ld; d.rows_per_sample = 1
    for k, v in kw.items():
        setattr(d, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    return d


def run(planes, Hin, Win, Cc, stride, rows_band=0, time=True, seed=0):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    Hout, Wout = (Hin - 1) // stride + 1, (Win - 1) // stride + 1
    x = torch.randn(planes * Hin * Win, Cc, device=dev, generator=g).to(BF)
    coef = torch.cat([torch.rand(Cc, device=dev, generator=g) + 0.5, torch.randn(Cc, device=dev, generator=g) * 0.3])
    w = torch.randn(9, Cc, device=dev, generator=g) / 3.0
    res = {}
    for mode in ("old", "new"):
        out = torch.full((planes * Hout * Wout, Cc), float("nan"), dtype=BF, device=dev)
        st = torch.zeros(32 * 2 * Cc, dtype=torch.float64, device=dev)
        a = L.DwSpatialFwdArgs()
        a.inp = desc(x, Cc, v1=coef, v2=coef[Cc:], act=1)
        a.w = w.data_ptr(); a.out = out.data_ptr(); a.planes = planes; a.Hin = Hin; a.Win = Win; a.Hout = Hout
        a.Wout = Wout; a.C = Cc; a.stride = stride; a.ks = 3; a.stats = st.data_ptr()
        a.impl = 1 if mode == "old" else 0
        a.rows_band = rows_band if mode == "new" else 0

        def fn():
            L.check(L.lib.dwn_dw_spatial_fwd(C.byref(a), L.DWN_BF16, 0, stream()), "dws")
        fn()
        torch.cuda.synchronize()
        r = (out.clone(), st.view(32, 2, Cc).sum(0).clone())
        res[mode] = r + ((timeit(fn) if time else None),)
    (o0, s0, t0), (o1, s1, t1) = res["old"], res["new"]
    nan = int(torch.isnan(o1.float()).sum())
    neq = int((o0.float() != o1.float()).sum())
    srel = float(((s0 - s1).abs() / (s0.abs() + 1e-2 * s0.abs().mean())).max())
    alg = (x.numel() + o0.numel()) * 2
    line = f"planes={planes:5d} {Hin}x{Win} C={Cc} s={stride} band={rows_band}: nan={nan} y2 neq={neq}/{o0.numel()} stats rel={srel:.2e}"
    if time:
        line += f" | old {t0*1e3:7.1f} us ({alg/t0/1e6:5.0f} GB/s)  new {t1*1e3:7.1f} us ({alg/t1/1e6:5.0f} GB/s)"
    print(line, flush=True)
    return nan == 0 and neq == 0 and srel < 1e-4


if __name__ == "__main__":
    which = sys.argv[1:] or ["small", "full"]
    ok = True
    if "small" in which:
        for cfg in ((3, 18, 32, 64, 1), (5, 9, 16, 128, 1), (7, 5, 8, 64, 1), (2, 3, 32, 72, 1), (9, 1, 8, 64, 1), (1, 20, 16, 64, 1),
                    (3, 36, 64, 64, 2), (5, 18, 32, 128, 2), (7, 9, 16, 64, 2), (2, 4, 64, 72, 2), (9, 1, 16, 64, 2), (3, 7, 32, 64, 2)):
            ok &= run(*cfg, time=False)
        for rb in (1, 2, 4, 7):
            ok &= run(3, 18, 32, 64, 1, rows_band=rb, time=False)
        for rb in (1, 2, 3):
            ok &= run(3, 36, 64, 64, 2, rows_band=rb, time=False)
    if "full" in which:
        for cfg in ((1024, 36, 64, 448, 2), (1024, 18, 32, 448, 1), (1024, 18, 32, 896, 2), (1024, 9, 16, 896, 1), (1024, 9, 16, 1792, 2),
                    (1024, 5, 8, 1792, 1)):
            ok &= run(*cfg)
    if "s2bands" in which:
        for rb in (2, 3, 4, 6):
            run(1024, 36, 64, 448, 2, rows_band=rb)
        for rb in (3, 5, 9):
            run(1024, 18, 32, 896, 2, rows_band=rb)
        for rb in (3, 5):
            run(1024, 9, 16, 1792, 2, rows_band=rb)
    if "bands" in which:
        for rb in (2, 3, 4, 6):
            run(1024, 36, 64, 448, 2, rows_band=rb)
        for rb in (3, 6, 9):
            run(1024, 18, 32, 448, 1, rows_band=rb)
        for rb in (3, 5, 9):
            run(1024, 18, 32, 896, 2, rows_band=rb)
    print("ALL OK" if ok else "MISMATCH", flush=True)
