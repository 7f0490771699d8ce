#!/usr/bin/env python3
"""Placement study (round-3 verdict item 1): why do byte-identical E-wide launches run 196 vs 217-222 us depending on which
block's buffers they touch?  Development tool; writes gpurun_out/placement_*.json.

  sweep : dw_temporal_fwd (C-ABI) and a plain device copy on the 589 824 x 448 bf16 shape with source / destination placed at
          chosen offsets of ONE big arena (offsets 0..2 MiB in 4 KiB steps, then large strides), on freshly hipMalloc'ed
          buffers and on torch-cached segments.
  step  : the metric model's training forward with every block's intermediates captured; the same stand-alone launch is then
          timed on the very buffers each block used (addresses printed), which separates placement from context.
"""
import ctypes as C
import json
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
import sensorium_amd._lib as L

dev = torch.device("cuda", 0)
BF = torch.bfloat16
M, E = 589824, 448
NB = M * E * 2
MiB = 1 << 20
OUT = ROOT / "gpurun_out"


def stream():
    return torch.cuda.current_stream().cuda_stream


def timed(fn, reps=7, warm=2, pre=None):
    for _ in range(warm):
        if pre:
            pre()
        fn()
    ts = []
    for _ in range(reps):
        if pre:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return round(min(ts), 1), round(statistics.median(ts), 1)


class Probe:
    """dw_temporal_fwd on [M][E] bf16 at arbitrary source / destination addresses."""

    def __init__(self, B=32, T=32, HW=576, Cc=E):
        self.coef = torch.rand(4 * Cc, device=dev) + 0.5
        self.w = torch.randn(5, Cc, device=dev)
        self.st = torch.zeros(32 * 2 * Cc, dtype=torch.float64, device=dev)
        self.B, self.T, self.HW, self.C = B, T, HW, Cc

    def args(self, src_ptr, dst_ptr):
        a = L.DwTemporalFwdArgs()
        d = L.LoadDesc()
        d.p = src_ptr; d.ld = self.C; d.rows_per_sample = 1; d.act = 1
        d.v1 = self.coef.data_ptr(); d.v2 = self.coef[self.C:].data_ptr()
        a.inp = d
        a.w = self.w.data_ptr(); a.out = dst_ptr; a.B = self.B; a.T = self.T; a.HW = self.HW; a.C = self.C; a.kt = 5
        a.stats = self.st.data_ptr()
        return a

    def run(self, src_ptr, dst_ptr, **kw):
        a = self.args(src_ptr, dst_ptr)
        return timed(lambda: L.check(L.lib.dwn_dw_temporal_fwd(C.byref(a), L.DWN_BF16, 0, stream()), "dwtf"), **kw)


def view(arena, off, nbytes=NB):
    return arena[off:off + nbytes].view(BF)


def sweep():
    res = {"shape": [M, E], "bytes": NB}
    probe = Probe()
    arena = torch.empty(12 << 30, dtype=torch.uint8, device=dev)
    base = arena.data_ptr()
    res["arena_base"] = hex(base)
    arena[:3 << 30].view(BF).normal_()
    print("arena base", hex(base), "mod 2MiB", base % (2 * MiB), flush=True)

    def one(so, do):
        s, d = view(arena, so), view(arena, do)
        k = probe.run(base + so, base + do)
        c = timed(lambda: d.copy_(s))
        return {"src_off": so, "dst_off": do, "delta": do - so, "dwt_us": k, "copy_us": c}

    # A: destination offset 0..2 MiB in 4 KiB steps, source fixed
    rows = []
    for i in range(0, 512):
        rows.append(one(0, (1 << 30) + i * 4096))
    res["A_dst_4k_steps"] = rows
    ts = [r["dwt_us"][1] for r in rows]
    print("A  dst +4KiB steps: dwt median min/max", min(ts), max(ts), " copy", min(r["copy_us"][1] for r in rows),
          max(r["copy_us"][1] for r in rows), flush=True)
    # B: source offset sweep
    rows = [one(i * 64 * 1024, 1 << 30) for i in range(0, 33)]
    res["B_src_64k_steps"] = rows
    print("B  src +64KiB steps:", min(r["dwt_us"][1] for r in rows), max(r["dwt_us"][1] for r in rows), flush=True)
    # C: destination immediately behind the source, then gaps of k * 2 MiB, then large strides
    rows = []
    for gap in [0, 4096, 65536, MiB, 2 * MiB, 4 * MiB, 8 * MiB, 16 * MiB, 24 * MiB, 32 * MiB, 64 * MiB, 128 * MiB, 256 * MiB, 512 * MiB,
                (1 << 30) - NB, (1 << 30), (2 << 30) - NB, (2 << 30), (4 << 30) - NB, (4 << 30), (8 << 30) - NB]:
        rows.append(one(0, NB + gap))
    res["C_gaps"] = rows
    for r in rows:
        print(f"C  delta {r['delta'] / MiB:10.3f} MiB  dwt {r['dwt_us']}  copy {r['copy_us']}", flush=True)
    # D: both moved together through the arena (absolute position), delta fixed at 1 GiB
    rows = [one(k * 256 * MiB, k * 256 * MiB + (1 << 30)) for k in range(0, 40, 2)]
    res["D_absolute"] = rows
    for r in rows:
        print(f"D  src_off {r['src_off'] / MiB:8.0f} MiB  dwt {r['dwt_us']}  copy {r['copy_us']}", flush=True)
    # E: destination before the source (reversed roles)
    rows = [one((1 << 30) + i * 512 * 1024, 0) for i in range(0, 5)]
    res["E_reversed"] = rows
    print("E  reversed:", [r["dwt_us"] for r in rows], flush=True)
    # F: a producer pass over the source right before each timed launch (what the step does: the stencil has just written y2)
    src0 = view(arena, 0)
    scratch = view(arena, 6 << 30)
    rows = []
    for do in [(1 << 30), (1 << 30) + 64 * 1024, NB, NB + 2 * MiB]:
        k = probe.run(base, base + do, pre=lambda: src0.copy_(scratch))
        rows.append({"dst_off": do, "dwt_us_after_producer": k})
    res["F_after_producer"] = rows
    print("F  after producer:", rows, flush=True)
    del arena
    torch.cuda.empty_cache()

    # G: separately allocated tensors, torch caching allocator (fresh segments) vs raw hipMalloc
    rows = []
    keep = []
    for i in range(6):
        s = torch.empty(M, E, dtype=BF, device=dev).normal_()
        d = torch.empty(M, E, dtype=BF, device=dev)
        keep += [s, d]
        rows.append({"src": hex(s.data_ptr()), "dst": hex(d.data_ptr()), "delta_MiB": (d.data_ptr() - s.data_ptr()) / MiB,
                     "dwt_us": probe.run(s.data_ptr(), d.data_ptr()), "copy_us": timed(lambda: d.copy_(s))})
        print("G  torch", rows[-1], flush=True)
    res["G_torch_alloc"] = rows
    del keep
    torch.cuda.empty_cache()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    rows = []
    ptrs = []
    for i, pad in enumerate([0, 0, 4096, 65536, MiB, 3 * MiB]):
        ps, pd = C.c_void_p(), C.c_void_p()
        assert hip.hipMalloc(C.byref(ps), NB + pad) == 0 and hip.hipMalloc(C.byref(pd), NB + pad) == 0
        ptrs += [ps, pd]
        hip.hipMemset(ps, 0, C.c_size_t(NB))
        torch.cuda.synchronize()
        rows.append({"src": hex(ps.value), "dst": hex(pd.value), "delta_MiB": (pd.value - ps.value) / MiB, "pad": pad,
                     "dwt_us": probe.run(ps.value, pd.value)})
        print("G  hipMalloc", rows[-1], flush=True)
    for p in ptrs:
        hip.hipFree(p)
    res["G_hipmalloc"] = rows
    OUT.mkdir(exist_ok=True)
    (OUT / "placement_sweep.json").write_text(json.dumps(res, indent=1))


def step():
    """The metric model's forward with captured intermediates; stand-alone launches on each block's own buffers."""
    from sensorium_amd import DwiseNeuro
    from sensorium_amd.synthetic import make_batch
    torch.manual_seed(0)
    model = DwiseNeuro(readout_outputs=(7863,), expansion_ratio=7).to(dev).train()
    blks = list(model.core.blocks)[1::2]
    for b in blks:
        b._capture = True
    x, _ = make_batch(32, 32, 36, 64, (7863,), device=dev)
    res = []
    for it in range(3):          # the caching allocator settles after the first iterations
        with torch.autocast("cuda", dtype=BF):
            preds = model(x)
        torch.cuda.synchronize()
        rows = []
        for i, b in enumerate(blks):
            cap = b._captured
            y2, y3, z3 = cap["y2"], cap["y3"], cap["z3"]
            Bn, T, H, W, Cc = y2.shape
            probe = Probe(Bn, T, H * W, Cc)
            k = probe.run(y2.data_ptr(), y3.data_ptr())
            k2 = probe.run(y3.data_ptr(), z3.data_ptr())
            c = timed(lambda: y3.copy_(y2))
            rows.append({"iter": it, "block": i, "rows": Bn * T * H * W, "C": Cc, "y1": hex(cap["y1"].data_ptr()), "y2": hex(y2.data_ptr()),
                         "y3": hex(y3.data_ptr()), "z3": hex(z3.data_ptr()), "y3-y2_MiB": (y3.data_ptr() - y2.data_ptr()) / MiB,
                         "z3-y3_MiB": (z3.data_ptr() - y3.data_ptr()) / MiB, "y2_mod_2MiB": y2.data_ptr() % (2 * MiB),
                         "dwt_y2_to_y3_us": k, "dwt_y3_to_z3_us": k2, "copy_us": c})
            print(rows[-1], flush=True)
        res += rows
        del preds
        for b in blks:
            b._captured = None
    print(torch.cuda.memory_summary(abbreviated=True)[:1500])
    OUT.mkdir(exist_ok=True)
    (OUT / "placement_step.json").write_text(json.dumps(res, indent=1))


def cold():
    """Per buffer pair of blocks 0-3 (training forward, distinct buffers): the probe launch warm (repeated), behind a flush of the
    Infinity Cache that touches few pages (256 MB copied back and forth), behind a sweep that touches ONE line of every 2 MiB page
    of a 64 GiB region (translations replaced, caches not), and behind a 2 GiB fill (both)."""
    from sensorium_amd import DwiseNeuro
    from sensorium_amd.synthetic import make_batch
    torch.manual_seed(0)
    model = DwiseNeuro(readout_outputs=(7863,), expansion_ratio=7).to(dev).train()
    blks = list(model.core.blocks)[1::2]
    for b in blks:
        b._capture = True
    x, _ = make_batch(32, 32, 36, 64, (7863,), device=dev)
    with torch.autocast("cuda", dtype=BF):
        preds = model(x)
    torch.cuda.synchronize()
    small_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    small_b = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    pages = torch.zeros(64 << 30, dtype=torch.uint8, device=dev).view(-1, 2 * MiB)
    big = torch.empty(2 << 30, dtype=torch.uint8, device=dev)

    def mall_flush():
        for _ in range(3):
            small_b.copy_(small_a)
            small_a.copy_(small_b)

    def tlb_flush():
        pages[:, 0].sum()

    rows = []
    for i, b in enumerate(blks[:4]):
        cap = b._captured
        for sname, dname in (("y2", "y3"), ("y3", "z3")):
            s_, d_ = cap[sname], cap[dname]
            Bn, T, H, W, Cc = s_.shape
            probe = Probe(Bn, T, H * W, Cc)
            r = {"block": i, "pair": f"{sname}->{dname}", "src": hex(s_.data_ptr()), "dst": hex(d_.data_ptr()),
                 "warm": probe.run(s_.data_ptr(), d_.data_ptr()),
                 "mall_flushed": probe.run(s_.data_ptr(), d_.data_ptr(), pre=mall_flush),
                 "tlb_swept": probe.run(s_.data_ptr(), d_.data_ptr(), pre=tlb_flush),
                 "both": probe.run(s_.data_ptr(), d_.data_ptr(), pre=lambda: big.zero_())}
            rows.append(r)
            print(r, flush=True)
    # is a slow level a property of one buffer (read alone / written alone) or of the pair?  every E-wide buffer of blocks 0-3:
    bufs = {}
    for i, b in enumerate(blks[:4]):
        for nm in ("y2", "y3", "z3"):
            bufs[f"b{i}.{nm}"] = b._captured[nm]
    scratch = torch.empty_like(bufs["b0.y2"])
    per = {}
    for nm, t in bufs.items():
        per[nm] = {"addr": hex(t.data_ptr()), "read_only_sum_us": timed(lambda: t.view(torch.int16).sum(dtype=torch.int64)),
                   "write_only_fill_us": timed(lambda: t.zero_()),
                   "probe_as_src_to_scratch": None, "probe_scratch_to_dst": None}
    probe = Probe(32, 32, 576, 448)
    for nm, t in bufs.items():
        per[nm]["probe_as_src_to_scratch"] = probe.run(t.data_ptr(), scratch.data_ptr())
        per[nm]["probe_scratch_to_dst"] = probe.run(scratch.data_ptr(), t.data_ptr())
        print(nm, per[nm], flush=True)
    names = list(bufs)
    cross = []
    for sn in names[:6]:
        row = []
        for dn in names[:6]:
            row.append(None if sn == dn else probe.run(bufs[sn].data_ptr(), bufs[dn].data_ptr())[1])
        cross.append(row)
        print("cross", sn, row, flush=True)
    OUT.mkdir(exist_ok=True)
    (OUT / "placement_cold.json").write_text(json.dumps({"pairs": rows, "per_buffer": per, "cross_names": names[:6], "cross_median_us": cross},
                                                        indent=1))
    del preds


if __name__ == "__main__":
    which = sys.argv[1:] or ["sweep", "step"]
    if "cold" in which:
        cold()
    if "sweep" in which:
        sweep()
    if "step" in which:
        step()
