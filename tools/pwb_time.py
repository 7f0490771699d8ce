#!/usr/bin/env python3
"""Time dwn_pw_backward (conv_pw data + weight gradient without y1: Bp / Gram prep, the product kernels, the fold) at the
benchmark's block shapes: the one-pass kernel of the 64-channel blocks and the two-GEMM path of the others."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
import sensorium_amd._lib as L  # noqa: E402

with_res = "--res" in sys.argv          # fold a stride-1 shortcut branch's gradient in (dwn_pw_bwd_args.res)
bf = torch.bfloat16
for M, E, Cin in ((2359296, 448, 64), (589824, 448, 64), (589824, 896, 128), (147456, 896, 128), (147456, 1792, 256),
                  (40960, 1792, 256)):
    dh1 = torch.randn(M, E, device="cuda").to(bf)
    a0 = torch.randn(M, Cin, device="cuda").to(bf)
    w1 = torch.randn(E, Cin, device="cuda") * 0.1
    abc = torch.randn(3, E, device="cuda")
    da0 = torch.empty(M, Cin, device="cuda", dtype=bf)
    dw = torch.zeros(E, Cin, device="cuda")
    nws = L.lib.dwn_pw_backward_workspace_bytes(E, Cin, L.DWN_BF16)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    a = L.PwBwdArgs()
    a.dh1, a.a0, a.w_pw, a.abc = dh1.data_ptr(), a0.data_ptr(), w1.data_ptr(), abc.data_ptr()
    a.da0, a.dw, a.M, a.E, a.Cin, a.ws, a.ws_bytes = da0.data_ptr(), dw.data_ptr(), M, E, Cin, ws.data_ptr(), nws
    if with_res and L.lib.dwn_pw_bwd_fused_supported(L.DWN_BF16, M, E, Cin):
        res = torch.randn(M, Cin, device="cuda").to(bf)
        rabc = torch.randn(3, Cin, device="cuda")
        a.res, a.res_abc, a.res_C = res.data_ptr(), rabc.data_ptr(), Cin
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        L.check(L.lib.dwn_pw_backward(C.byref(a), L.DWN_BF16, 0, st), "pw_backward")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        L.lib.dwn_pw_backward(C.byref(a), L.DWN_BF16, 0, st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fused = L.lib.dwn_pw_bwd_fused_supported(L.DWN_BF16, M, E, Cin)
    by = M * ((1 if fused else 2) * E + (2 if fused else 3) * Cin) * 2
    print(f"M={M} E={E} Cin={Cin} one_pass={fused} res={with_res} {us:.1f} us  {by / us / 1e3:.0f} GB/s of the bytes the path has to move", flush=True)
