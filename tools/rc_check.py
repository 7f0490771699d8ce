#!/usr/bin/env python3
"""Development tool: the y1-recomputing spatial forward (dwn_dw_spatial_fwd_rc) against the materialised path
(dwn_gemm_nn -> dwn_dw_spatial_fwd) at the benchmark's block shapes: equality of y2 / BN sums, and launch times.
usage: python tools/rc_check.py [small] [full] [bands]"""
import ctypes as C
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


def run(planes, Hin, Win, Cin, E, stride, rows_band=0, time=True, seed=0, round_y1=1):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    Hout, Wout = (Hin - 1) // stride + 1, (Win - 1) // stride + 1
    Min, Mout = planes * Hin * Win, planes * Hout * Wout
    a0 = torch.randn(Min, Cin, device=dev, generator=g).to(BF)
    w1 = torch.randn(E, Cin, device=dev, generator=g) / Cin ** 0.5
    coef = torch.cat([torch.rand(E, device=dev, generator=g) + 0.5, torch.randn(E, device=dev, generator=g) * 0.3])
    wdw = torch.randn(9, E, device=dev, generator=g) / 3.0
    # ---- reference: materialised y1
    w1p = torch.empty(E, Cin, dtype=BF, device=dev)
    L.check(L.lib.dwn_pack_weight(w1.data_ptr(), w1p.data_ptr(), 1, E, Cin, 0, E, Cin, L.DWN_BF16, 0, stream()), "pack")
    y1 = torch.empty(Min, E, dtype=BF, device=dev)
    gm = L.GemmNNArgs()
    gm.a = desc(a0, Cin); gm.a_kind = L.LD_PLAIN; gm.b = w1p.data_ptr(); gm.ldb = Cin; gm.c = y1.data_ptr(); gm.ldc = E
    gm.M, gm.N, gm.K, gm.groups = Min, E, Cin, 1
    gm.stats = None; gm.stat_nchan = E; gm.epi = L.EPI_STORE
    y2r = torch.empty(Mout, E, dtype=BF, device=dev)
    str_ = torch.zeros(32 * 2 * E, dtype=torch.float64, device=dev)
    fa = L.DwSpatialFwdArgs()
    fa.inp = desc(y1, E, v1=coef, v2=coef[E:], act=1)
    fa.w = wdw.data_ptr(); fa.out = y2r.data_ptr(); fa.planes = planes; fa.Hin = Hin; fa.Win = Win; fa.Hout = Hout
    fa.Wout = Wout; fa.C = E; fa.stride = stride; fa.ks = 3; fa.stats = str_.data_ptr(); fa.rows_band = 0

    def ref_gemm():
        L.check(L.lib.dwn_gemm_nn(C.byref(gm), L.DWN_BF16, 0, stream()), "nn")

    def ref_dws():
        L.check(L.lib.dwn_dw_spatial_fwd(C.byref(fa), L.DWN_BF16, 0, stream()), "dws")
    ref_gemm(); ref_dws()
    # ---- recompute path
    blob = torch.zeros(L.lib.dwn_dw_spatial_rc_blob_bytes(E, Cin), dtype=torch.uint8, device=dev)
    L.check(L.lib.dwn_dw_spatial_rc_prep(w1.data_ptr(), wdw.data_ptr(), coef.data_ptr(), E, Cin, blob.data_ptr(), 0, stream()), "prep")
    y2 = torch.full((Mout, E), float("nan"), dtype=BF, device=dev)
    st = torch.zeros(32 * 2 * E, dtype=torch.float64, device=dev)
    ra = L.DwSpatialRcFwdArgs()
    ra.a0 = a0.data_ptr(); ra.a0_ld = Cin; ra.blob = blob.data_ptr(); ra.out = y2.data_ptr()
    ra.planes = planes; ra.Hin = Hin; ra.Win = Win; ra.Hout = Hout; ra.Wout = Wout; ra.Cin = Cin; ra.E = E
    ra.stride = stride; ra.stats = st.data_ptr(); ra.rows_band = rows_band; ra.round_y1 = round_y1

    def rc():
        L.check(L.lib.dwn_dw_spatial_fwd_rc(C.byref(ra), 0, stream()), "rc")
    rc()
    torch.cuda.synchronize()
    a, b = y2.float(), y2r.float()
    nan = int(torch.isnan(a).sum())
    neq = int((a != b).sum())
    mx = float((a - b).abs().max()) if nan == 0 else float("nan")
    s_rc = st.view(32, 2, E).sum(0); s_rf = str_.view(32, 2, E).sum(0)
    srel = float(((s_rc - s_rf).abs() / (s_rf.abs() + 1e-3)).max())
    line = (f"planes={planes:5d} {Hin}x{Win} Cin={Cin} E={E} s={stride} band={rows_band}: nan={nan} "
            f"neq={neq}/{a.numel()} max|d|={mx:.3e} ref_absmax={float(b.abs().max()):.2f} stats_rel={srel:.2e}")
    if time:
        st.zero_(); str_.zero_()
        t_g, t_d, t_r = timeit(ref_gemm), timeit(ref_dws), timeit(rc)
        alg = (Min + Mout) * E * 2
        line += (f" | gemm {t_g*1e3:7.1f} us  dws {t_d*1e3:7.1f} us ({alg/t_d/1e6:6.0f} GB/s)  rc {t_r*1e3:7.1f} us "
                 f"({alg/t_r/1e6:6.0f} GB/s alg)")
    print(line, flush=True)
    return nan == 0 and neq == 0


if __name__ == "__main__":
    which = sys.argv[1:] or ["small", "full"]
    ok = True
    if "small" in which:
        for cfg in ((3, 18, 32, 64, 128, 1), (3, 36, 64, 64, 64, 2), (2, 9, 16, 128, 192, 1), (2, 18, 32, 128, 128, 2),
                    (5, 7, 5, 64, 64, 1), (5, 7, 5, 64, 64, 2), (4, 5, 8, 128, 64, 1), (3, 10, 11, 64, 128, 2),
                    (3, 1, 2, 64, 64, 1), (2, 2, 3, 64, 64, 2)):
            ok &= run(*cfg, time=False)
        for rb in (1, 2, 3, 5):
            ok &= run(3, 18, 32, 64, 128, 1, rows_band=rb, time=False)
            ok &= run(3, 18, 32, 64, 128, 2, rows_band=rb, time=False)
        ok &= run(3, 18, 32, 64, 128, 1, time=False, round_y1=0)
    if "full" in which:
        for cfg in ((1024, 36, 64, 64, 448, 2), (1024, 18, 32, 64, 448, 1), (1024, 18, 32, 128, 896, 2),
                    (1024, 9, 16, 128, 896, 1)):
            ok &= run(*cfg)
    if "one" in which:
        run(1024, 18, 32, 64, 448, 1)
    if "bands" in which:
        for rb in (2, 3):
            run(1024, 36, 64, 64, 448, 2, rows_band=rb)
        for rb in (3, 6, 9, 12):
            run(1024, 18, 32, 64, 448, 1, rows_band=rb)
    print("ALL EQUAL" if ok else "MISMATCH", flush=True)
