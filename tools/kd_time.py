#!/usr/bin/env python3
"""Stand-alone launch times of the deep-K point-wise GEMMs of blocks 4-8 (round 5): the A-direct kernel (dwn_gemm_kd.hip) against
the kernels it replaces (variant DWN_NN_TILE128 = what the library chose before).  usage: python3 tools/kd_time.py"""
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


def run(name, form, M, N, K, K2, rows):
    a = torch.randn(M, K, device=dev).to(BF)
    Kt = K + (K2 if form == "cat" else 0)
    b = (torch.randn(N, Kt, device=dev) / Kt ** 0.5).to(BF)
    a2 = torch.randn(M, max(K2, 8), device=dev).to(BF)
    bias = torch.randn(N, device=dev)
    gate = torch.rand(M // rows, K, device=dev) + 0.25
    c = torch.empty(M, N, dtype=BF, device=dev)
    st = stats_buffer(N)
    res = {}
    for vname, variant in (("old", L.NN_TILE128), ("kd", L.NN_KD)):
        g = L.GemmNNArgs()
        g.a = load_desc(L, a, K); g.a_kind = L.LD_PLAIN
        if form == "gate":
            g.a.gate = gate.data_ptr(); g.a.gate_ld = K; g.a.rows_per_sample = rows; g.a_kind = L.LD_GATE
        g.b = b.data_ptr(); g.ldb = Kt; g.c = c.data_ptr(); g.ldc = N
        g.M, g.N, g.K, g.groups = M, N, Kt, 1
        g.stats = st.data_ptr() if form != "cat" else None; g.stat_nchan = N; g.epi = L.EPI_STORE
        if form == "cat":
            g.epi = L.EPI_STORE_CAT; g.a2 = a2.data_ptr(); g.a2_ld = K2; g.K1 = K; g.bias = bias.data_ptr()
        g.variant = variant
        res[vname] = timeit(lambda: L.check(L.lib.dwn_gemm_nn(C.byref(g), L.DWN_BF16, 0, s()), "gemm_nn"))
    byts = (M * Kt + M * N) * 2
    print(f"{name:22s} M={M:7d} N={N:4d} K={Kt:5d}: old {res['old']:7.1f} us  kd {res['kd']:7.1f} us  ({byts / res['kd'] / 1e6:5.2f} TB/s of algorithmic bytes)")


CASES = []
def case(*a):
    CASES.append(a)


if __name__ == "__main__":
    case("pwl_fwd b4-5", "gate", 147456, 128, 896, 0, 4608)
    case("pwl_fwd b6", "gate", 147456, 256, 896, 0, 4608)
    case("pwl_fwd b7-8", "gate", 40960, 256, 1792, 0, 1280)
    case("pw_dgrad b4", "cat", 589824, 128, 896, 128, 1)
    case("pw_dgrad b5-6", "cat", 147456, 128, 896, 128, 1)
    case("pw_dgrad b7", "cat", 147456, 256, 1792, 256, 1)
    case("pw_dgrad b8", "cat", 40960, 256, 1792, 256, 1)
    sel = sys.argv[1:]
    for c in CASES:
        if not sel or any(x in c[0] for x in sel):
            run(*c)
