#!/usr/bin/env python3
"""BatchNorm-1 statistics from the Gram matrix at the metric batch: error of mean / invstd against float64 for the library in use
(DWN_LIB_PATH selects an older build: fp32 atomics before round 6, fp64 atomics since).  python3 tools/gram_accuracy.py"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import sensorium_amd._lib as L

dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
torch.manual_seed(11)
for M, E, Cin, off in ((2359296, 448, 64, 3.0), (2359296, 448, 64, 0.5), (589824, 448, 64, 3.0), (147456, 896, 128, 3.0)):
    lat = torch.randn(M, 8, device=dev)
    mix = torch.randn(8, Cin, device=dev)
    scale = 0.5 + torch.rand(Cin, device=dev)
    a0 = (((lat @ mix) * 0.6 + torch.randn(M, Cin, device=dev) * 0.5) * scale + off * scale * torch.sign(torch.randn(Cin, device=dev))).to(torch.bfloat16)
    del lat
    w = torch.randn(E, Cin, device=dev) / Cin ** 0.5
    gamma, beta = torch.ones(E, device=dev), torch.zeros(E, device=dev)
    rm, rv = torch.zeros(E, device=dev), torch.ones(E, device=dev)
    nbt = torch.zeros(1, dtype=torch.int64, device=dev)
    coef = torch.empty(4 * E, device=dev)
    sc = torch.zeros(32 * 2 * Cin, dtype=torch.float64, device=dev)
    ws = torch.empty(2 * L.lib.dwn_conv_pw_bn_stats_workspace_bytes(Cin), dtype=torch.uint8, device=dev)
    bn = L.BN()
    bn.gamma = gamma.data_ptr(); bn.beta = beta.data_ptr(); bn.running_mean = rm.data_ptr(); bn.running_var = rv.data_ptr()
    bn.num_batches_tracked = nbt.data_ptr(); bn.coef = coef.data_ptr()
    worst_m, worst_i = 0.0, 0.0
    wd = w.to(torch.bfloat16).double()
    s1 = torch.zeros(E, dtype=torch.float64, device=dev)
    for r0 in range(0, M, 262144):
        s1 += (a0[r0:r0 + 262144].double() @ wd.t()).sum(0)
    mean = s1 / M
    s2 = torch.zeros(E, dtype=torch.float64, device=dev)
    for r0 in range(0, M, 262144):
        s2 += ((a0[r0:r0 + 262144].double() @ wd.t() - mean) ** 2).sum(0)
    var = s2 / M
    invstd = 1.0 / (var + 1e-5).sqrt()
    for rep in range(5):
        L.check(L.lib.dwn_conv_pw_bn_stats(a0.data_ptr(), Cin, M, w.data_ptr(), E, Cin, C.byref(bn), 0.1, 1e-5, sc.data_ptr(),
                                           ws.data_ptr(), ws.numel(), L.DWN_BF16, 0, s), "conv_pw_bn_stats")
        torch.cuda.synchronize()
        c = coef.view(4, E).double()
        worst_m = max(worst_m, float(((c[2] - mean).abs() / var.sqrt()).max()))
        worst_i = max(worst_i, float(((c[3] - invstd).abs() / invstd).max()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        L.lib.dwn_conv_pw_bn_stats(a0.data_ptr(), Cin, M, w.data_ptr(), E, Cin, C.byref(bn), 0.1, 1e-5, sc.data_ptr(), ws.data_ptr(), ws.numel(), L.DWN_BF16, 0, s)
    e1.record(); torch.cuda.synchronize()
    print(f"M={M} E={E} Cin={Cin} offset={off}: amplification (mean^2/var max) {float((mean ** 2 / var).max()):6.1f}  |mean err|/sigma {worst_m:.2e}  invstd rel err {worst_i:.2e}  {e0.elapsed_time(e1) * 100:.1f} us/call", flush=True)
