#!/usr/bin/env python3
"""Development tool: the chained stride-1 spatial backward (dw_spatial_bwd_s1c_kernel) against the pair kernel and the round-2
row-walk kernel, through dwn_dw_spatial_bwd with DWN_DWS_WALK_OFF / DWN_DWS_CHAIN / DWN_DWS_CHAIN_RB toggled per call:
dh1 equality, dW / BN-sum agreement, launch times."""
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import sensorium_amd._lib as L
from tools.bwd_check import desc, stream, timeit, dev, BF

MODES = [("pair", {"DWN_DWS_WALK_OFF": "1", "DWN_DWS_Y2RC": "0"}), ("walk", {"DWN_DWS_WALK_OFF": "0", "DWN_DWS_CHAIN": "0", "DWN_DWS_Y2RC": "0"})] + \
        [(f"chain{rb}", {"DWN_DWS_WALK_OFF": "0", "DWN_DWS_CHAIN": "1", "DWN_DWS_CHAIN_RB": str(rb), "DWN_DWS_Y2RC": "0"}) for rb in (2, 4)] + \
        [(f"y2rc{rb}", {"DWN_DWS_WALK_OFF": "0", "DWN_DWS_Y2RC": "1", "DWN_DWS_Y2RC_RB": str(rb)}) for rb in (1, 2)]


def run(planes, Hin, Win, Cc, time=True, seed=0, modes=MODES):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    y1 = torch.randn(planes * Hin * Win, Cc, device=dev, generator=g).to(BF)
    dh2 = torch.randn(planes * Hin * Win, Cc, device=dev, generator=g).to(BF)
    y2 = torch.empty(planes * Hin * Win, Cc, device=dev, dtype=BF)
    coef = torch.cat([torch.rand(Cc, device=dev, generator=g) + 0.5, torch.randn(Cc, device=dev, generator=g) * 0.3,
                      torch.randn(Cc, device=dev, generator=g) * 0.2, torch.rand(Cc, device=dev, generator=g) + 0.5])
    abc = torch.randn(3 * Cc, device=dev, generator=g) * 0.5
    w = (torch.randn(9, Cc, device=dev, generator=g) / 3.0).to(BF).float()
    # y2 = the forward stencil of SiLU(BN1(y1)): what the backward of a real block sees (the y2-rebuilding kernel recomputes it)
    fa = L.DwSpatialFwdArgs()
    fa.inp = desc(y1, Cc, v1=coef, v2=coef[Cc:], act=1)
    fa.w = w.data_ptr(); fa.out = y2.data_ptr(); fa.planes = planes; fa.Hin = Hin; fa.Win = Win; fa.Hout = Hin; fa.Wout = Win
    fa.C = Cc; fa.stride = 1; fa.ks = 3; fa.stats = None; fa.rows_band = 0
    L.check(L.lib.dwn_dw_spatial_fwd(C.byref(fa), L.DWN_BF16, 0, stream()), "dwsf")
    torch.cuda.synchronize()
    res = {}
    for name, env in modes:
        os.environ.update(env)
        dh1 = torch.full_like(y1, float("nan"))
        dw = torch.zeros(Cc, 9, device=dev)
        st = torch.zeros(32 * 2 * Cc, dtype=torch.float64, device=dev)
        a = L.DwSpatialBwdArgs()
        a.dy = desc(dh2, Cc, q=y2, v1=abc, v2=abc[Cc:], v3=abc[2 * Cc:])
        a.y1 = desc(y1, Cc, v1=coef, v2=coef[Cc:], v3=coef[2 * Cc:], v4=coef[3 * Cc:])
        a.w = w.data_ptr(); a.dh1 = dh1.data_ptr(); a.dw = dw.data_ptr(); a.planes = planes; a.Hin = Hin; a.Win = Win
        a.Hout = Hin; a.Wout = Win; a.C = Cc; a.stride = 1; a.ks = 3; a.stats = st.data_ptr(); a.rows_band = 0

        def fn():
            L.check(L.lib.dwn_dw_spatial_bwd(C.byref(a), L.DWN_BF16, 0, stream()), "dwsb")
        fn()
        torch.cuda.synchronize()
        out = (dh1.clone(), dw.clone(), st.view(32, 2, Cc).sum(0).clone())
        res[name] = out + ((timeit(fn) if time else None),)
    d0, w0, s0, _ = res["pair"]
    alg = 3 * y1.numel() * 2
    ok = True
    line = f"planes={planes:5d} {Hin}x{Win} C={Cc}:"
    for name, _ in modes[1:]:
        d1, w1, s1, t1 = res[name]
        nan = int(torch.isnan(d1.float()).sum())
        neq = int((d0.view(torch.int16) != d1.view(torch.int16)).sum())
        wrel = float((w0 - w1).norm() / w0.norm())
        srel = float(((s0 - s1).abs() / (s0.abs() + 1e-2 * s0.abs().mean())).max())
        if name.startswith("y2rc"):
            # y2 rebuilt with another tap pairing: a few results one bf16 ulp off
            drel = float((d0.float() - d1.float()).norm() / d0.float().norm()) if nan == 0 else float("nan")
            good = nan == 0 and neq < 2e-3 * d0.numel() and drel < 1e-3 and wrel < 1e-3 and srel < 1e-3
            line += f" [neq {neq / d0.numel():.1e} rel {drel:.1e}]"
        else:
            good = nan == 0 and neq == 0 and wrel < 1e-3 and srel < 1e-3
        ok &= good
        line += f" | {name}: {'ok' if good else f'BAD nan={nan} neq={neq} dW={wrel:.1e} st={srel:.1e}'}"
        if time:
            line += f" {t1*1e3:6.1f}us {alg/t1/1e9:5.2f}TB/s(alg)"
    print(line, flush=True)
    return ok


if __name__ == "__main__":
    which = sys.argv[1:] or ["small", "full"]
    ok = True
    if "small" in which:
        for cfg in ((3, 18, 32, 64), (5, 9, 16, 128), (7, 5, 8, 64), (2, 3, 32, 72), (9, 1, 8, 64), (1, 20, 16, 64), (1, 2, 32, 64),
                    (4, 7, 8, 200), (130, 9, 16, 448), (131, 5, 8, 448), (33, 18, 32, 448)):
            ok &= run(*cfg, time=False)
    if "full" in which:
        for cfg in ((1024, 18, 32, 448), (1024, 9, 16, 896), (1024, 5, 8, 1792)):
            ok &= run(*cfg)
    print("ALL OK" if ok else "MISMATCH", flush=True)
