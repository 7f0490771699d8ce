#!/usr/bin/env python3
"""Development tool: per-phase cycle counts of the y1-recomputing spatial forward (build with -DRC_PROFILE, load it with
DWN_LIB_PATH=sensorium_amd/csrc/build_prof/libdwiseneuro_hip.so).  Prints s_memtime cycles per slice-iteration."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import sensorium_amd._lib as L

dev = torch.device("cuda", 0)
BF = torch.bfloat16
NAMES = ("phaseA", "vmcnt0", "barrier1", "phaseB", "vm_last", "barrier2")


def run(planes, Hin, Win, Cin, E, stride, rows_band=0, stats=True):
    Hout, Wout = (Hin - 1) // stride + 1, (Win - 1) // stride + 1
    a0 = torch.randn(planes * Hin * Win, Cin, device=dev).to(BF)
    w1 = torch.randn(E, Cin, device=dev) / Cin ** 0.5
    coef = torch.cat([torch.rand(E, device=dev) + 0.5, torch.randn(E, device=dev) * 0.3])
    wdw = torch.randn(9, E, device=dev) / 3.0
    s = torch.cuda.current_stream().cuda_stream
    blob = torch.zeros(L.lib.dwn_dw_spatial_rc_blob_bytes(E, Cin), dtype=torch.uint8, device=dev)
    L.check(L.lib.dwn_dw_spatial_rc_prep(w1.data_ptr(), wdw.data_ptr(), coef.data_ptr(), E, Cin, blob.data_ptr(), 0, s), "prep")
    y2 = torch.empty(planes * Hout * Wout, E, dtype=BF, device=dev)
    st = torch.zeros(32 * 2 * E, dtype=torch.float64, device=dev)
    ra = L.DwSpatialRcFwdArgs()
    ra.a0 = a0.data_ptr(); ra.a0_ld = Cin; ra.blob = blob.data_ptr(); ra.out = y2.data_ptr()
    ra.planes = planes; ra.Hin = Hin; ra.Win = Win; ra.Hout = Hout; ra.Wout = Wout; ra.Cin = Cin; ra.E = E
    ra.stride = stride; ra.stats = st.data_ptr() if stats else None; ra.rows_band = rows_band; ra.round_y1 = 1
    for _ in range(3):
        L.check(L.lib.dwn_dw_spatial_fwd_rc(C.byref(ra), 0, s), "rc")
    torch.cuda.synchronize()
    buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
    fn = L.lib.dwn_rc_prof_read
    fn.restype = C.c_int; fn.argtypes = [C.c_void_p]
    rc = fn(buf.ctypes.data)
    assert rc == 0, rc
    p = buf.reshape(256, 8, 8).astype(np.float64)
    its = p[:, :, 6]
    ok = its > 0
    per = p[:, :, :6] / np.maximum(its[:, :, None], 1)
    print(f"planes={planes} {Hin}x{Win} Cin={Cin} E={E} s={stride} band={rows_band} stats={stats}: iterations/WG {its[ok].mean():.1f}")
    for w in range(8):
        sel = ok[:, w]
        print(f"  wave {w}: " + "  ".join(f"{n} {per[sel, w, i].mean():8.0f}" for i, n in enumerate(NAMES))
              + f"  total {per[sel, w].sum(-1).mean():8.0f}")
    print("  all   : " + "  ".join(f"{n} {per[ok][:, i].mean():8.0f}" for i, n in enumerate(NAMES))
          + f"  total {per[ok].sum(-1).mean():8.0f} cycles/iteration", flush=True)


if __name__ == "__main__":
    run(1024, 18, 32, 64, 448, 1)
    run(1024, 18, 32, 64, 448, 1, stats=False)
    run(1024, 18, 32, 64, 448, 1, rows_band=3)
    run(1024, 36, 64, 64, 448, 2)
    run(1024, 9, 16, 128, 896, 1)
