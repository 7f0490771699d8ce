#!/usr/bin/env python3
"""Stand-alone times of the stored-y1 spatial backward at the shapes of blocks 4-8: python3 tools/bwd_time.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.argv = [sys.argv[0], "bwd"]
import tools.rebuilt_time as R  # noqa: E402
import ctypes as C, torch
import sensorium_amd._lib as L
dev = torch.device("cuda", 0); BF = torch.bfloat16
def run(planes, Hin, Win, Cc, stride):
    g = torch.Generator(device=dev); g.manual_seed(0)
    Hout, Wout = (Hin - 1) // stride + 1, (Win - 1) // stride + 1
    y1 = torch.randn(planes * Hin * Win, Cc, device=dev, generator=g).to(BF)
    dh2 = torch.randn(planes * Hout * Wout, Cc, device=dev, generator=g).to(BF)
    y2 = torch.randn(planes * Hout * Wout, Cc, device=dev, generator=g).to(BF)
    coef = torch.cat([torch.rand(Cc, device=dev, generator=g) + 0.5, torch.randn(Cc, device=dev, generator=g) * 0.3,
                      torch.randn(Cc, device=dev, generator=g) * 0.2, torch.rand(Cc, device=dev, generator=g) + 0.5])
    abc = torch.randn(3 * Cc, device=dev, generator=g) * 0.5
    w = torch.randn(9, Cc, device=dev, generator=g) / 3.0
    dh1 = torch.empty_like(y1); dw = torch.zeros(Cc, 9, device=dev); st = torch.zeros(32 * 2 * Cc, dtype=torch.float64, device=dev)
    a = L.DwSpatialBwdArgs()
    a.dy = R.desc(dh2, Cc, q=y2, v1=abc, v2=abc[Cc:], v3=abc[2 * Cc:])
    a.y1 = R.desc(y1, Cc, v1=coef, v2=coef[Cc:], v3=coef[2 * Cc:], v4=coef[3 * Cc:])
    a.w = w.data_ptr(); a.dh1 = dh1.data_ptr(); a.dw = dw.data_ptr(); a.planes = planes; a.Hin = Hin; a.Win = Win
    a.Hout = Hout; a.Wout = Wout; a.C = Cc; a.stride = stride; a.ks = 3; a.stats = st.data_ptr()
    t = R.timeit(lambda: L.check(L.lib.dwn_dw_spatial_bwd(C.byref(a), L.DWN_BF16, 0, R.s()), "bwd"))
    print(f"stored bwd planes={planes} {Hin}x{Win} C={Cc} s={stride}: {t:7.1f} us", flush=True)
for cfg in ((1024, 36, 64, 448, 2), (1024, 18, 32, 896, 2), (1024, 9, 16, 1792, 2)):
    run(*cfg)
