#!/usr/bin/env python3
"""Where do the gradient collectives sit relative to the backward pass?  Summary of a rocprofv3 --kernel-trace of
`bench.py --ddp-single-rank --mice 10` (one rank over RCCL on one GPU: the collectives are real RCCL kernels on the high-priority
communication stream, with nobody to exchange with — their placement and their co-residency with the compute kernels are what this
shows, not their multi-GPU duration).

    python3 tools/ddp_trace_summary.py <dir with *_kernel_trace.csv> [out.json]

For the last complete training step of the trace: the step's span, every RCCL kernel (start / duration relative to the step and to the
core backward), how much of each runs while a compute kernel of another stream is running, and the compute kernels it overlaps."""
import csv
import glob
import json
import re
import sys


def main():
    d = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", ""), r.get("Queue_Id", "")))
    rows.sort()
    is_comm = lambda n: bool(re.search(r"nccl|rccl|oneRank", n, re.I))
    # steps end with the fused optimizer
    opt = [i for i, r in enumerate(rows) if "adamw_ema_kernel" in r[2]]
    if len(opt) < 3:
        raise SystemExit("need at least three steps in the trace")
    a, b = opt[-3] + 1, opt[-2] + 1                     # the second-to-last complete step (the last may be cut by profiler shutdown)
    step = rows[a:b]
    t0, t1 = step[0][0], max(r[1] for r in step)
    short = lambda n: re.sub(r"^void ", "", re.sub(r"\(.*", "", n))[:70]
    comp = [r for r in step if not is_comm(r[2])]
    comm = [r for r in step if is_comm(r[2])]
    # backward = from the loss backward kernel to the optimizer; core backward = from the first block-backward prep after the cortex
    # backward to the stem backward
    names = [short(r[2]) for r in step]
    first_bwd = next((i for i, n in enumerate(names) if "poisson_bwd" in n), None)
    stem_bwd = next((i for i, n in enumerate(names) if "stem_bwd" in n), None)
    rep = {"trace": f, "step_ms": (t1 - t0) / 1e6, "kernels_in_step": len(step), "rccl_kernels": len(comm),
           "backward_starts_ms": None if first_bwd is None else (step[first_bwd][0] - t0) / 1e6,
           "stem_backward_ms": None if stem_bwd is None else (step[stem_bwd][0] - t0) / 1e6,
           "optimizer_starts_ms": (step[-1][0] - t0) / 1e6, "collectives": []}
    tot_comm = tot_olap = 0
    for c in comm:
        olap, with_ = 0, {}
        for k in comp:
            lo, hi = max(c[0], k[0]), min(c[1], k[1])
            if hi > lo:
                olap += hi - lo
                with_[short(k[2])] = with_.get(short(k[2]), 0) + (hi - lo)
        tot_comm += c[1] - c[0]
        tot_olap += min(olap, c[1] - c[0])
        rep["collectives"].append({"kernel": short(c[2]), "stream": c[3], "queue": c[4], "start_ms": (c[0] - t0) / 1e6, "dur_ms": (c[1] - c[0]) / 1e6,
                                   "overlapped_with_compute_ms": min(olap, c[1] - c[0]) / 1e6,
                                   "top_overlaps": sorted(((v / 1e6, k) for k, v in with_.items()), reverse=True)[:4]})
    rep["rccl_total_ms"] = tot_comm / 1e6
    rep["rccl_overlapped_ms"] = tot_olap / 1e6
    print(json.dumps({k: v for k, v in rep.items() if k != "collectives"}, indent=1))
    for c in rep["collectives"]:
        print(f"  {c['start_ms']:8.3f} ms  +{c['dur_ms']:7.3f} ms  overlapped {c['overlapped_with_compute_ms']:7.3f}  {c['kernel']}  {c['top_overlaps'][:2]}")
    if out:
        json.dump(rep, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
