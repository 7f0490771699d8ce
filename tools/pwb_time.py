#!/usr/bin/env python3
"""Time dwn_pw_bwd_fused at the benchmark's 64-channel block shapes; `--alias` passes dh1 as y1 too (half the HBM reads,
same instruction stream: tells whether the kernel is bandwidth- or latency-bound)."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
import sensorium_amd._lib as L  # noqa: E402

alias = "--alias" in sys.argv
E, Cin = 448, 64
for M in (589824, 2359296):
    bf = torch.bfloat16
    dh1 = torch.randn(M, E, device="cuda").to(bf)
    y1 = dh1 if alias else torch.randn(M, E, device="cuda").to(bf)
    a0 = torch.randn(M, Cin, device="cuda").to(bf)
    w1t = torch.randn(Cin, E, device="cuda").to(bf)
    abc = torch.randn(3, E, device="cuda")
    da0 = torch.empty(M, Cin, device="cuda", dtype=bf)
    dw = torch.zeros(E, Cin, device="cuda")
    a = L.PwBwdArgs()
    a.dh1, a.y1, a.a0, a.w1t, a.abc = dh1.data_ptr(), y1.data_ptr(), a0.data_ptr(), w1t.data_ptr(), abc.data_ptr()
    a.da0, a.dw, a.M, a.E, a.Cin = da0.data_ptr(), dw.data_ptr(), M, E, Cin
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        L.check(L.lib.dwn_pw_bwd_fused(C.byref(a), L.DWN_BF16, 0, st), "pw_bwd_fused")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        L.lib.dwn_pw_bwd_fused(C.byref(a), L.DWN_BF16, 0, st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    by = M * (2 * E + 2 * Cin) * 2 if not alias else M * (E + 2 * Cin) * 2
    print(f"M={M} alias={alias} {us:.1f} us  {by / us / 1e3:.0f} GB/s of the bytes actually distinct", flush=True)
