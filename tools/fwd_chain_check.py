#!/usr/bin/env python3
"""Development tool: the chained spatial forward (dw_spatial_fwd_chain_kernel) against the pair kernel and the round-2 banded
row-walk kernel, through dwn_dw_spatial_fwd with DWN_DWS_WALK_OFF / DWN_DWS_FCHAIN / DWN_DWS_FCHAIN_RB toggled per call."""
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import sensorium_amd._lib as L
from tools.fwd_check import desc, stream, timeit, dev, BF


def modes_for(stride):
    rbs = (2, 4, 6, 8) if stride == 1 else (1, 2, 3, 4)
    return [("pair", {"DWN_DWS_WALK_OFF": "1"}), ("walk", {"DWN_DWS_WALK_OFF": "0", "DWN_DWS_FCHAIN": "0"})] + \
           [(f"chain{rb}", {"DWN_DWS_WALK_OFF": "0", "DWN_DWS_FCHAIN": "1", "DWN_DWS_FCHAIN_RB": str(rb)}) for rb in rbs]


def run(planes, Hin, Win, Cc, stride, time=True, seed=0):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    Hout, Wout = (Hin - 1) // stride + 1, (Win - 1) // stride + 1
    x = torch.randn(planes * Hin * Win, Cc, device=dev, generator=g).to(BF)
    coef = torch.cat([torch.rand(Cc, device=dev, generator=g) + 0.5, torch.randn(Cc, device=dev, generator=g) * 0.3])
    w = torch.randn(9, Cc, device=dev, generator=g) / 3.0
    res = {}
    modes = modes_for(stride)
    for name, env in modes:
        os.environ.update(env)
        out = torch.full((planes * Hout * Wout, Cc), float("nan"), dtype=BF, device=dev)
        st = torch.zeros(32 * 2 * Cc, dtype=torch.float64, device=dev)
        a = L.DwSpatialFwdArgs()
        a.inp = desc(x, Cc, v1=coef, v2=coef[Cc:], act=1)
        a.w = w.data_ptr(); a.out = out.data_ptr(); a.planes = planes; a.Hin = Hin; a.Win = Win; a.Hout = Hout
        a.Wout = Wout; a.C = Cc; a.stride = stride; a.ks = 3; a.stats = st.data_ptr(); a.rows_band = 0

        def fn():
            L.check(L.lib.dwn_dw_spatial_fwd(C.byref(a), L.DWN_BF16, 0, stream()), "dws")
        fn()
        torch.cuda.synchronize()
        res[name] = (out.clone(), st.view(32, 2, Cc).sum(0).clone(), (timeit(fn) if time else None))
    o0, s0, _ = res["pair"]
    alg = (x.numel() + o0.numel()) * 2
    ok = True
    line = f"planes={planes:5d} {Hin}x{Win} C={Cc} s={stride}:"
    for name, _ in modes[1:]:
        o1, s1, t1 = res[name]
        nan = int(torch.isnan(o1.float()).sum())
        neq = int((o0.view(torch.int16) != o1.view(torch.int16)).sum())
        srel = float(((s0 - s1).abs() / (s0.abs() + 1e-2 * s0.abs().mean())).max())
        good = nan == 0 and neq == 0 and srel < 1e-4
        ok &= good
        line += f" | {name}: {'ok' if good else f'BAD nan={nan} neq={neq} st={srel:.1e}'}"
        if time:
            line += f" {t1*1e3:6.1f}us {alg/t1/1e9:5.2f}TB/s"
    print(line, flush=True)
    return ok


if __name__ == "__main__":
    which = sys.argv[1:] or ["small", "full"]
    ok = True
    if "small" in which:
        for cfg in ((3, 18, 32, 64, 1), (5, 9, 16, 128, 1), (7, 5, 8, 64, 1), (2, 3, 32, 72, 1), (9, 1, 8, 64, 1), (1, 20, 16, 64, 1),
                    (3, 36, 64, 64, 2), (5, 18, 32, 128, 2), (7, 9, 16, 64, 2), (2, 4, 64, 72, 2), (9, 1, 16, 64, 2), (3, 7, 32, 64, 2),
                    (1, 2, 32, 64, 1), (4, 7, 8, 200, 1), (130, 9, 16, 448, 1), (131, 5, 8, 448, 1), (33, 18, 32, 448, 1),
                    (129, 18, 32, 448, 2), (130, 9, 16, 448, 2), (33, 36, 64, 448, 2), (3, 5, 16, 64, 2)):
            ok &= run(*cfg, time=False)
    if "full" in which:
        for cfg in ((1024, 36, 64, 448, 2), (1024, 18, 32, 448, 1), (1024, 18, 32, 896, 2), (1024, 9, 16, 896, 1), (1024, 9, 16, 1792, 2),
                    (1024, 5, 8, 1792, 1)):
            ok &= run(*cfg)
    print("ALL OK" if ok else "MISMATCH", flush=True)
