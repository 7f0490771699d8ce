#!/usr/bin/env python3
"""Context study for the E-wide streaming kernels (round-3 verdict item 1, second half).  tools/placement_sweep.py showed that a
stand-alone dw_temporal_fwd takes 216-220 us on the 589 824 x 448 bf16 shape WHEREVER its buffers sit; inside the step block 0's
instance takes 196 us.  So the fast level is context, not placement.  This tool times the kernel behind different predecessors:

  python tools/context_probe.py seq      # C-ABI kernels with HIP events: [producer] -> dw_temporal_fwd, several producers / orders
  rocprofv3 --kernel-trace ... -- python3 tools/context_probe.py blocks   # whole block forwards; per-kernel times from the trace
"""
import ctypes as C
import json
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
import sensorium_amd._lib as L

dev = torch.device("cuda", 0)
BF = torch.bfloat16
OUT = ROOT / "gpurun_out"


def stream():
    return torch.cuda.current_stream().cuda_stream


def desc(p, ld, **kw):
    d = L.LoadDesc()
    d.p = p if isinstance(p, int) else p.data_ptr(); d.ld = ld; d.rows_per_sample = 1
    for k, v in kw.items():
        setattr(d, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    return d


class Kernels:
    """dws_fwd (stride s) writing y2 [B*T*Ho*Wo][E], dwt_fwd y2 -> y3 (optionally on a sub-range of the samples)."""

    def __init__(self, B, T, Hin, Win, stride, E):
        self.B, self.T, self.E, self.stride = B, T, E, stride
        self.Ho, self.Wo = (Hin - 1) // stride + 1, (Win - 1) // stride + 1
        self.Hin, self.Win = Hin, Win
        self.y1 = torch.randn(B * T * Hin * Win, E, device=dev).to(BF)
        self.y2 = torch.empty(B * T * self.Ho * self.Wo, E, dtype=BF, device=dev)
        self.y3 = torch.empty_like(self.y2)
        self.coef = torch.rand(4 * E, device=dev) + 0.5
        self.w9 = torch.randn(9, E, device=dev)
        self.w5 = torch.randn(5, E, device=dev)
        self.st = torch.zeros(32 * 2 * E, dtype=torch.float64, device=dev)
        a = L.DwSpatialFwdArgs()
        a.inp = desc(self.y1, E, v1=self.coef, v2=self.coef[E:], act=1)
        a.w = self.w9.data_ptr(); a.out = self.y2.data_ptr(); a.planes = B * T; a.Hin = Hin; a.Win = Win
        a.Hout = self.Ho; a.Wout = self.Wo; a.C = E; a.stride = stride; a.ks = 3; a.stats = self.st.data_ptr(); a.rows_band = 0
        self.a_dws = a

    def dws(self):
        L.check(L.lib.dwn_dw_spatial_fwd(C.byref(self.a_dws), L.DWN_BF16, 0, stream()), "dws")

    def dwt(self, b0=0, nb=None):
        nb = self.B - b0 if nb is None else nb
        E, HW = self.E, self.Ho * self.Wo
        off = b0 * self.T * HW * E * 2
        a = L.DwTemporalFwdArgs()
        a.inp = desc(self.y2.data_ptr() + off, E, v1=self.coef, v2=self.coef[E:], act=1)
        a.w = self.w5.data_ptr(); a.out = self.y3.data_ptr() + off; a.B = nb; a.T = self.T; a.HW = HW; a.C = E; a.kt = 5
        a.stats = self.st.data_ptr()
        L.check(L.lib.dwn_dw_temporal_fwd(C.byref(a), L.DWN_BF16, 0, stream()), "dwt")


def time_after(pre, fn, reps=9, warm=2):
    ts = []
    for i in range(warm + reps):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if i >= warm:
            ts.append(e0.elapsed_time(e1) * 1e3)
    return round(min(ts), 1), round(statistics.median(ts), 1)


def seq():
    res = {}
    big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    for name, (Hin, Win, stride) in (("block0 (36x64, stride 2)", (36, 64, 2)), ("block1 (18x32, stride 1)", (18, 32, 1))):
        k = Kernels(32, 32, Hin, Win, stride, 448)
        r = {}
        r["dwt alone, back to back"] = time_after(lambda: None, k.dwt)
        r["dwt after a 1 GiB fill of another buffer (cold caches)"] = time_after(lambda: big.zero_(), k.dwt)
        r["dwt after dws (the step's order)"] = time_after(k.dws, k.dwt)
        r["dwt after dws + 5 ms idle"] = time_after(lambda: (k.dws(), torch.cuda.synchronize(), time.sleep(0.005)), k.dwt)
        r["dwt after 20 ms idle"] = time_after(lambda: (torch.cuda.synchronize(), time.sleep(0.02)), k.dwt)
        r["dwt after copy y3 -> y2 (ascending producer)"] = time_after(lambda: k.y2.copy_(k.y3), k.dwt)
        # consumer in four launches of 8 samples: ascending vs descending sample order, behind the stencil
        r["dwt in 4 sub-launches ascending, after dws"] = time_after(k.dws, lambda: [k.dwt(b, 8) for b in (0, 8, 16, 24)])
        r["dwt in 4 sub-launches descending, after dws"] = time_after(k.dws, lambda: [k.dwt(b, 8) for b in (24, 16, 8, 0)])
        r["dwt in 8 sub-launches descending, after dws"] = time_after(k.dws, lambda: [k.dwt(b, 4) for b in range(28, -1, -4)])
        r["dwt in 4 sub-launches ascending, alone"] = time_after(lambda: None, lambda: [k.dwt(b, 8) for b in (0, 8, 16, 24)])
        r["dws after dwt"] = time_after(k.dwt, k.dws)
        r["dws alone"] = time_after(lambda: None, k.dws)
        res[name] = r
        for kk, v in r.items():
            print(f"{name:28s} {kk:60s} {v}", flush=True)
        del k
    OUT.mkdir(exist_ok=True)
    (OUT / "context_seq.json").write_text(json.dumps(res, indent=1))


def blocks():
    """Whole block forwards (training mode, no autograd) for rocprofv3: phase A 8 x block-0 shape, phase B 8 x block-1 shape,
    phase C 4 x (block 0 -> 1 -> 2 -> 3 chained)."""
    from sensorium_amd.dwiseneuro import InvertedResidual3d, PositionalEncoding3d
    torch.manual_seed(0)
    mk = lambda s: InvertedResidual3d(64, 64, spatial_kernel=3, temporal_kernel=5, spatial_stride=s, expansion_ratio=7,
                                      se_reduce_ratio=32).to(dev).train()
    b0, b1, b2, b3 = mk(2), mk(1), mk(1), mk(1)
    pe = PositionalEncoding3d(64).to(dev)
    x0 = torch.randn(32, 32, 36, 64, 64, device=dev).to(BF)
    x1 = torch.randn(32, 32, 18, 32, 64, device=dev).to(BF)
    with torch.no_grad():
        for _ in range(8):
            b0(x0, pe, BF)
        torch.cuda.synchronize()
        for _ in range(8):
            b1(x1, pe, BF)
        torch.cuda.synchronize()
        for _ in range(4):
            y = b0(x0, pe, BF)
            y = b1(y, pe, BF)
            y = b2(y, pe, BF)
            y = b3(y, pe, BF)
        torch.cuda.synchronize()
    # phase D: the same chain with autograd on (every block keeps its own y1, y2, y3, z3 for backward: distinct buffers, 4 x 2.6 GB)
    x0g = x0.clone().requires_grad_(True)
    for _ in range(4):
        y = b0(x0g, pe, BF)
        y = b1(y, pe, BF)
        y = b2(y, pe, BF)
        y = b3(y, pe, BF)
        torch.cuda.synchronize()
        del y
    # phase E: forward + backward of the chain (what a training step does to these four blocks)
    for _ in range(4):
        y = b3(b2(b1(b0(x0g, pe, BF), pe, BF), pe, BF), pe, BF)
        y.backward(torch.ones_like(y))
        torch.cuda.synchronize()
        del y


def parse(trace_csv):
    import csv
    rows = sorted(csv.DictReader(open(trace_csv)), key=lambda r: int(r["Start_Timestamp"]))
    out = {}
    for fam in ("dw_spatial_fwd", "dw_temporal_fwd", "se_pool", "gemm_nn", "dw_temporal_bwd", "dw_spatial_bwd"):
        out[fam] = [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in rows if fam in r["Kernel_Name"]]
    print(json.dumps(out))
    (OUT / "context_blocks.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "seq":
        seq()
    elif sys.argv[1] == "blocks":
        blocks()
    elif sys.argv[1] == "parse":
        parse(sys.argv[2])
