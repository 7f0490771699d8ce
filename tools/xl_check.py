#!/usr/bin/env python3
"""Development tool: the 256-row LDS-DMA GEMM (dwn_gemm_xl.hip) against the 128x128 kernels through dwn_gemm_nn with DWN_NN_XL
toggled per call: results vs float64 math on the same rounded operands, BatchNorm sums, launch times."""
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import sensorium_amd._lib as L

dev = torch.device("cuda", 0)
BF = torch.bfloat16


def timeit(fn, n=10, warm=3):
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


def run(M, N, K, groups=1, stats=True, time=True, seed=0, check=True):
    g_ = torch.Generator(device=dev); g_.manual_seed(seed)
    a = torch.randn(M, groups * K, device=dev, generator=g_).to(BF)
    b = (torch.randn(groups * N, K, device=dev, generator=g_) / K ** 0.5).to(BF)
    res = {}
    for mode in ("base", "xl128", "xl256"):
        c = torch.full((M, groups * N), float("nan"), dtype=BF, device=dev)
        st = torch.zeros(32 * 2 * groups * N, dtype=torch.float64, device=dev)
        g = L.GemmNNArgs()
        d = L.LoadDesc(); d.p = a.data_ptr(); d.ld = groups * K; d.rows_per_sample = 1
        g.a = d; g.a_kind = L.LD_PLAIN
        g.variant = {"base": L.NN_TILE128, "xl128": L.NN_XL128, "xl256": L.NN_XL256}[mode]
        g.b = b.data_ptr(); g.ldb = K; g.c = c.data_ptr(); g.ldc = groups * N
        g.M, g.N, g.K, g.groups = M, N, K, groups
        if stats:
            g.stats = st.data_ptr(); g.stat_nchan = groups * N
        g.epi = L.EPI_STORE
        s = torch.cuda.current_stream().cuda_stream

        def fn():
            L.check(L.lib.dwn_gemm_nn(C.byref(g), L.DWN_BF16, 0, s), "gemm_nn")
        fn()
        torch.cuda.synchronize()
        res[mode] = (c.clone(), st.view(32, 2, groups * N).sum(0).clone(), timeit(fn) if time else None)
    line = f"M={M} N={N} K={K} g={groups}:"
    ok = True
    c0 = res["base"][0]
    if check:
        ref = torch.cat([a[:, i * K:(i + 1) * K].double() @ b[i * N:(i + 1) * N].double().t() for i in range(groups)], 1)
    for mode in ("base", "xl128", "xl256"):
        c, st, t = res[mode]
        nan = int(torch.isnan(c.float()).sum())
        msg = f" | {mode}:"
        if check:
            err = float((c.double() - ref).norm() / ref.norm()) if nan == 0 else float("nan")
            good = nan == 0 and err < 6e-3
            if stats:
                cf = c.double()
                e0 = float((st[0] - cf.sum(0)).norm() / (cf.sum(0).norm() + 1e-9))
                e1 = float((st[1] - (cf * cf).sum(0)).norm() / (cf * cf).sum(0).norm())
                good = good and e0 < 1e-4 and e1 < 1e-4
                msg += f" err {err:.1e} st {max(e0, e1):.0e}"
            else:
                msg += f" err {err:.1e}"
            ok &= good
            msg += " ok" if good else f" BAD nan={nan}"
        if t is not None:
            msg += f" {t * 1e3:7.1f}us {2.0 * M * N * K * groups / t / 1e9:6.0f}TF"
        line += msg
    print(line, flush=True)
    return ok


if __name__ == "__main__":
    which = sys.argv[1:] or ["small", "full"]
    ok = True
    if "small" in which:
        for cfg in ((256, 256, 64), (300, 264, 72), (1000, 128, 448), (513, 896, 128), (640, 56, 128), (257, 24, 8), (1024, 512, 256),
                    (777, 1000, 200)):
            ok &= run(*cfg, time=False)
        ok &= run(200, 48, 32, groups=2, time=False)
        ok &= run(1024, 512, 128, groups=2, time=False)
        ok &= run(900, 1000, 328, groups=2, time=False)
    if "full" in which:
        ok &= run(147456, 1792, 256, check=False)
        ok &= run(40960, 1792, 256)
        ok &= run(1024, 2048, 3936, groups=2, stats=False)
        ok &= run(1024, 512, 128, groups=2)
        ok &= run(1024, 1024, 512, groups=2)
        ok &= run(1024, 2048, 1024, groups=2)
        ok &= run(1024, 1024, 2048, groups=2, stats=False)
        ok &= run(1024, 512, 1024, groups=2, stats=False)
        ok &= run(1024, 3936, 2048, groups=2, stats=False)
    print("ALL OK" if ok else "MISMATCH", flush=True)
