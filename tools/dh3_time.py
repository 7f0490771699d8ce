#!/usr/bin/env python3
"""Stand-alone launch times of conv_pwl's data-gradient GEMM with the dh3 epilogue at the metric shapes (round 5).
usage: python3 tools/dh3_time.py"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import sensorium_amd._lib as L
from tests.gpu_helpers import load_desc, stats_buffer

dev = torch.device("cuda", 0)
BF = torch.bfloat16
s = lambda: torch.cuda.current_stream().cuda_stream


def timeit(fn, n=10, reps=3):
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


def run(name, B, S, N, K):
    M = B * S
    a = (torch.randn(M, K, device=dev) * 0.5).to(BF)
    b = (torch.randn(N, K, device=dev) * 0.2).to(BF)
    y3 = torch.randn(M, N, device=dev).to(BF)
    gate = torch.rand(B, N, device=dev)
    dps = torch.randn(B, N, device=dev) * 0.1
    coef = torch.stack([torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.3,
                        torch.randn(N, device=dev) * 0.2, torch.rand(N, device=dev) + 0.5]).contiguous()
    out = torch.empty(M, N, dtype=BF, device=dev)
    st = stats_buffer(N)
    g = L.GemmNNArgs()
    g.a = load_desc(L, a, K); g.a_kind = L.LD_PLAIN
    g.b = b.data_ptr(); g.ldb = K; g.c = out.data_ptr(); g.ldc = N
    g.M, g.N, g.K, g.groups = M, N, K, 1
    g.epi = L.EPI_DH3
    g.y3 = y3.data_ptr(); g.ldy3 = N; g.gate3 = gate.data_ptr(); g.dps3 = dps.data_ptr(); g.dg_ld = N
    g.coef3 = coef.data_ptr(); g.coef3_ld = N; g.rows_per_sample = S
    g.stats = st.data_ptr(); g.stat_nchan = N
    t = timeit(lambda: L.check(L.lib.dwn_gemm_nn(C.byref(g), L.DWN_BF16, 0, s()), "gemm_nn dh3"))
    byts = (M * K + 2 * M * N) * 2
    print(f"{name:16s} M={M:7d} N={N:4d} K={K:4d}: {t:7.1f} us  ({byts / t / 1e6:5.2f} TB/s of algorithmic bytes)")


if __name__ == "__main__":
    run("pwl_dgrad b0-3", 32, 18432, 448, 64)
    run("pwl_dgrad b4-5", 32, 4608, 896, 128)
    run("pwl_dgrad b6", 32, 4608, 896, 256)
    run("pwl_dgrad b7-8", 32, 1280, 1792, 256)
