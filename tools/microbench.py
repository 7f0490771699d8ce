#!/usr/bin/env python3
"""Kernel-level microbenchmarks through the C-ABI (development tool; not part of the product or the tests).
Times individual launches at the benchmark's block shapes with CUDA/HIP events and prints achieved GB/s against
the algorithmic bytes, next to a plain device copy of the same size."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import sensorium_amd._lib as L

dev = torch.device("cuda", 0)
BF = torch.bfloat16


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


def stream():
    return torch.cuda.current_stream().cuda_stream


def desc(p, ld, **kw):
    d = L.LoadDesc()
    d.p = p.data_ptr(); d.ld = ld; d.rows_per_sample = 1
    for k, v in kw.items():
        setattr(d, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    return d


def report(name, ms, nbytes):
    print(f"{name:58s} {ms*1e3:9.1f} us  {nbytes/ms/1e6:8.1f} GB/s", flush=True)


def gemm_nn(M, N, K, kind="plain", stats=False, S=None, T=32, H=36, W=64, bsample=0):
    a = torch.randn(M, K, device=dev).to(BF)
    b = (torch.randn((M // bsample if bsample else 1) * N, K, device=dev) / K ** 0.5).to(BF)
    c = torch.empty(M, N, dtype=BF, device=dev)
    st = torch.zeros(32 * 2 * N, dtype=torch.float64, device=dev)
    keep = []
    if kind == "plain":
        d, k = desc(a, K), L.LD_PLAIN
    elif kind == "pe":
        pt, ph, pw = (torch.randn(s, K, device=dev) for s in (T, H, W))
        keep += [pt, ph, pw]
        d, k = desc(a, K, pe_t=pt, pe_h=ph, pe_w=pw, pT=T, pH=H, pW=W, pe_ld=K), L.LD_PE
    elif kind == "bnact":
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev)
        gate = torch.rand(M // S, K, device=dev)
        keep += [sc, sh, gate]
        d, k = desc(a, K, v1=sc, v2=sh, act=1, gate=gate, gate_ld=K, rows_per_sample=S), L.LD_BNACT
    elif kind == "gate":
        gate = torch.rand(M // S, K, device=dev)
        keep += [gate]
        d, k = desc(a, K, gate=gate, gate_ld=K, rows_per_sample=S), L.LD_GATE
    elif kind == "affine2":
        y = torch.randn(M, K, device=dev).to(BF)
        a1, a2, a3 = (torch.randn(K, device=dev) for _ in range(3))
        keep += [y, a1, a2, a3]
        d, k = desc(a, K, q=y, v1=a1, v2=a2, v3=a3), L.LD_AFFINE2
    g = L.GemmNNArgs()
    g.a = d; g.a_kind = k; g.b = b.data_ptr(); g.ldb = K; g.c = c.data_ptr(); g.ldc = N
    g.M, g.N, g.K, g.groups = M, N, K, 1
    g.stats = st.data_ptr() if stats else None; g.stat_nchan = N; g.epi = L.EPI_STORE
    if bsample:
        g.b_sample_stride = N * K; g.b_rows_per_sample = bsample
    ms = timeit(lambda: L.check(L.lib.dwn_gemm_nn(C.byref(g), L.DWN_BF16, 0, stream()), "nn"))
    nb = (M * K * (2 if kind == "affine2" else 1) + M * N) * 2
    report(f"gemm_nn M={M} N={N} K={K} {kind} stats={int(stats)} bsample={bsample}", ms, nb)


def gemm_tn(M, R, Cc, pk="plain", qk="plain", rows_per_sample=0):
    p = torch.randn(M, R, device=dev).to(BF)
    q = torch.randn(M, Cc, device=dev).to(BF)
    nb_ = M // rows_per_sample if rows_per_sample else 1
    dw = torch.zeros(nb_ * R, Cc, device=dev)
    keep = []
    dp, kp = desc(p, R), L.LD_PLAIN
    if pk == "affine2":
        y = torch.randn(M, R, device=dev).to(BF)
        a1, a2, a3 = (torch.randn(R, device=dev) for _ in range(3))
        keep += [y, a1, a2, a3]
        dp, kp = desc(p, R, q=y, v1=a1, v2=a2, v3=a3), L.LD_AFFINE2
    dq, kq = desc(q, Cc), L.LD_PLAIN
    g = L.GemmTNArgs()
    g.p = dp; g.p_kind = kp; g.q = dq; g.q_kind = kq
    g.M, g.R, g.Cc = M, R, Cc
    g.dw = dw.data_ptr(); g.lddw = Cc; g.groups = 1; g.nsplit = 0
    g.rows_per_sample = rows_per_sample; g.dw_sample_stride = R * Cc
    ms = timeit(lambda: L.check(L.lib.dwn_gemm_tn(C.byref(g), L.DWN_BF16, 0, stream()), "tn"))
    nb = (M * R * (2 if pk == "affine2" else 1) + M * Cc) * 2
    report(f"gemm_tn M={M} R={R} Cc={Cc} {pk}/{qk} rps={rows_per_sample}", ms, nb)


def copy(nbytes):
    a = torch.empty(nbytes // 2, dtype=BF, device=dev).normal_()
    b = torch.empty_like(a)
    ms = timeit(lambda: b.copy_(a))
    report(f"torch copy {nbytes/1e6:.0f} MB (read+write)", ms, 2 * nbytes)


def dws_fwd(planes, Hin, Win, C, stride, rows_band=0):
    Hout, Wout = (Hin - 1) // stride + 1, (Win - 1) // stride + 1
    x = torch.randn(planes * Hin * Win, C, device=dev).to(BF)
    out = torch.empty(planes * Hout * Wout, C, dtype=BF, device=dev)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    w = torch.randn(9, C, device=dev)
    st = torch.zeros(32 * 2 * C, dtype=torch.float64, device=dev)
    a = L.DwSpatialFwdArgs()
    a.inp = desc(x, C, v1=sc, v2=sh, act=1)
    a.w = w.data_ptr(); a.out = out.data_ptr(); a.planes = planes; a.Hin = Hin; a.Win = Win; a.Hout = Hout
    a.Wout = Wout; a.C = C; a.stride = stride; a.ks = 3; a.stats = st.data_ptr(); a.rows_band = rows_band
    ms = timeit(lambda: L.check(L.lib.dwn_dw_spatial_fwd(C_.byref(a), L.DWN_BF16, 0, stream()), "dws"))
    report(f"dws_fwd planes={planes} {Hin}x{Win} C={C} s={stride} band={rows_band}", ms, (x.numel() + out.numel()) * 2)


def dws_bwd(planes, Hin, Win, C, stride, rows_band=0):
    Hout, Wout = (Hin - 1) // stride + 1, (Win - 1) // stride + 1
    y1 = torch.randn(planes * Hin * Win, C, device=dev).to(BF)
    dh2 = torch.randn(planes * Hout * Wout, C, device=dev).to(BF)
    y2 = torch.randn(planes * Hout * Wout, C, device=dev).to(BF)
    dh1 = torch.empty_like(y1)
    coef = torch.rand(4 * C, device=dev) + 0.5
    abc = torch.randn(3 * C, device=dev)
    w = torch.randn(9, C, device=dev)
    dw = torch.zeros(C, 9, device=dev)
    st = torch.zeros(32 * 2 * C, dtype=torch.float64, device=dev)
    a = L.DwSpatialBwdArgs()
    a.dy = desc(dh2, C, q=y2, v1=abc, v2=abc[C:], v3=abc[2 * C:])
    a.y1 = desc(y1, C, v1=coef, v2=coef[C:], v3=coef[2 * C:], v4=coef[3 * C:])
    a.w = w.data_ptr(); a.dh1 = dh1.data_ptr(); a.dw = dw.data_ptr(); a.planes = planes; a.Hin = Hin; a.Win = Win
    a.Hout = Hout; a.Wout = Wout; a.C = C; a.stride = stride; a.ks = 3; a.stats = st.data_ptr(); a.rows_band = rows_band
    ms = timeit(lambda: L.check(L.lib.dwn_dw_spatial_bwd(C_.byref(a), L.DWN_BF16, 0, stream()), "dwsb"))
    report(f"dws_bwd planes={planes} {Hin}x{Win} C={C} s={stride} band={rows_band}", ms, (2 * y1.numel() + 2 * y2.numel()) * 2)


C_ = C


def dwt_bwd(B, T, HW, C):
    M = B * T * HW
    dh3 = torch.randn(M, C, device=dev).to(BF)
    y3 = torch.randn(M, C, device=dev).to(BF)
    y2 = torch.randn(M, C, device=dev).to(BF)
    dh2 = torch.empty_like(y2)
    coef = torch.rand(4 * C, device=dev) + 0.5
    abc = torch.randn(3 * C, device=dev)
    w = torch.randn(5, C, device=dev)
    dw = torch.zeros(C, 5, device=dev)
    st = torch.zeros(32 * 2 * C, dtype=torch.float64, device=dev)
    a = L.DwTemporalBwdArgs()
    a.dy = desc(dh3, C, q=y3, v1=abc, v2=abc[C:], v3=abc[2 * C:])
    a.dy_kind = L.LD_AFFINE2
    a.y2 = desc(y2, C, v1=coef, v2=coef[C:], v3=coef[2 * C:], v4=coef[3 * C:])
    a.w = w.data_ptr(); a.dh2 = dh2.data_ptr(); a.dw = dw.data_ptr(); a.B = B; a.T = T; a.HW = HW; a.C = C; a.kt = 5
    a.stats = st.data_ptr()
    ms = timeit(lambda: L.check(L.lib.dwn_dw_temporal_bwd(C_.byref(a), L.DWN_BF16, 0, stream()), "dwtb"))
    report(f"dwt_bwd B={B} T={T} HW={HW} C={C}", ms, 4 * M * C * 2)


def dwt_fwd(B, T, HW, C):
    M = B * T * HW
    y2 = torch.randn(M, C, device=dev).to(BF)
    y3 = torch.empty_like(y2)
    coef = torch.rand(4 * C, device=dev) + 0.5
    w = torch.randn(5, C, device=dev)
    st = torch.zeros(32 * 2 * C, dtype=torch.float64, device=dev)
    a = L.DwTemporalFwdArgs()
    a.inp = desc(y2, C, v1=coef, v2=coef[C:], act=1)
    a.w = w.data_ptr(); a.out = y3.data_ptr(); a.B = B; a.T = T; a.HW = HW; a.C = C; a.kt = 5; a.stats = st.data_ptr()
    ms = timeit(lambda: L.check(L.lib.dwn_dw_temporal_fwd(C_.byref(a), L.DWN_BF16, 0, stream()), "dwtf"))
    report(f"dwt_fwd B={B} T={T} HW={HW} C={C}", ms, 2 * M * C * 2)

if __name__ == "__main__":
    which = sys.argv[1:] or ["copy", "nn", "tn", "dws"]
    if "copy" in which:
        copy(2_100_000_000)
        copy(528_000_000)
    if "nn" in which:
        M0, M1 = 2359296, 589824
        for kind, st in (("plain", False), ("plain", True), ("pe", True)):
            gemm_nn(M0, 448, 64, kind, st)
        gemm_nn(M1, 448, 64, "pe", True, T=32, H=18, W=32)
        gemm_nn(M1, 64, 448, "plain", False)
        gemm_nn(M1, 64, 448, "plain", True)
        gemm_nn(M1, 64, 448, "bnact", True, S=18432)
        gemm_nn(M1, 64, 448, "gate", True, S=18432)
        gemm_nn(M1, 448, 64, "plain", False)
        gemm_nn(M0, 64, 448, "affine2", False)
    if "tn" in which:
        gemm_tn(2359296, 448, 64, "plain")
        gemm_tn(2359296, 448, 64, "affine2")
        gemm_tn(589824, 64, 448, "plain")
    if "dwsfx" in which:
        for rb in (2, 3, 4):
            dws_fwd(1024, 36, 64, 448, 2, rb)
        for rb in (4, 7, 9, 18):
            dws_fwd(1024, 18, 32, 448, 1, rb)
        for rb in (4, 6, 8, 12):
            dws_bwd(1024, 36, 64, 448, 2, rb)
        for rb in (3, 6, 9, 18):
            dws_bwd(1024, 18, 32, 448, 1, rb)
    if "dwt" in which:
        dwt_fwd(32, 32, 18 * 32, 448)
        dwt_bwd(32, 32, 18 * 32, 448)
        dwt_bwd(32, 32, 9 * 16, 896)
        dwt_bwd(32, 32, 5 * 8, 1792)
    if "dwtb1" in which:
        dwt_bwd(32, 32, 18 * 32, 448)
    if "nnps" in which:
        for (M, N, K, S) in ((589824, 64, 448, 18432), (589824, 128, 448, 18432), (147456, 128, 896, 4608),
                             (147456, 256, 896, 4608), (40960, 256, 1792, 1280)):
            gemm_nn(M, N, K, "plain", True)
            gemm_nn(M, N, K, "plain", True, bsample=S)
    if "tnps" in which:
        gemm_tn(589824, 64, 448, "plain")
        gemm_tn(589824, 64, 448, "plain", rows_per_sample=18432)
        gemm_tn(589824, 128, 448, "plain")
        gemm_tn(589824, 128, 448, "plain", rows_per_sample=18432)
        gemm_tn(147456, 128, 896, "plain")
        gemm_tn(147456, 128, 896, "plain", rows_per_sample=4608)
        gemm_tn(147456, 256, 896, "plain")
        gemm_tn(147456, 256, 896, "plain", rows_per_sample=4608)
    if "nnx" in which:                      # python tools/microbench.py nnx M N K [M N K ...]
        nums = [int(v) for v in which[which.index("nnx") + 1:]]
        for i in range(0, len(nums) - 2, 3):
            gemm_nn(nums[i], nums[i + 1], nums[i + 2], "plain", True)
        sys.exit(0)
    if "nndeep" in which:
        gemm_nn(589824, 896, 128, "plain", True)
        gemm_nn(589824, 896, 128, "plain", False)
        gemm_nn(147456, 1792, 256, "plain", True)
        gemm_nn(147456, 896, 128, "plain", True)
        gemm_nn(589824, 448, 64, "plain", True)
        gemm_nn(589824, 128, 896, "plain", True)
        gemm_nn(147456, 256, 1792, "plain", True)
    if "nn1" in which:
        gemm_nn(2359296, 448, 64, "plain", True)
    if "nn2" in which:
        gemm_nn(2359296, 64, 448, "affine2", False)
    if "tn1" in which:
        gemm_tn(2359296, 448, 64, "affine2")
    if "deep" in which:
        dws_fwd(1024, 18, 32, 896, 2)
        dws_fwd(1024, 9, 16, 896, 1)
        dws_fwd(1024, 9, 16, 1792, 2)
        dws_fwd(1024, 5, 8, 1792, 1)
        dws_bwd(1024, 18, 32, 896, 2)
        dws_bwd(1024, 9, 16, 896, 1)
        dws_bwd(1024, 9, 16, 1792, 2)
        dws_bwd(1024, 5, 8, 1792, 1)
    if "dwsb1" in which:
        dws_bwd(1024, 18, 32, 448, 1)
    if "dwsf0" in which:
        dws_fwd(1024, 36, 64, 448, 2)
    if "dwsb0" in which:
        dws_bwd(1024, 36, 64, 448, 2)
    if "dwsf1" in which:
        dws_fwd(1024, 18, 32, 448, 1)
    if "dws" in which:
        dws_fwd(1024, 36, 64, 448, 2)
        dws_fwd(1024, 18, 32, 448, 1)
        dws_bwd(1024, 36, 64, 448, 2)
        dws_bwd(1024, 18, 32, 448, 1)
