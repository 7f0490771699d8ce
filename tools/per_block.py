#!/usr/bin/env python3
"""Per-block, per-family table of one training step from rocprofv3 output (no GPU needed to run this).

    rocprofv3 --kernel-trace --stats --output-format csv -d T -o t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-rooflines
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d F -o p -- python3 bench.py --steps 1 --warmup 1 ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d W -o p -- python3 bench.py --steps 1 --warmup 1 ...
    python tools/per_block.py T/t_kernel_trace.csv F W profiles/r2_per_block.json profiles/r2_dws_per_block.json

One step = the dispatches between two adamw_ema_kernel launches.  Every dispatch of the step is assigned to (block, family) from
the launch order of DepthwiseBlock.forward / backward (sensorium_amd/csrc/dwn_api.hip); durations come from the plain
kernel trace (last step), HBM bytes from the two PMC passes (same launch sequence, matched by position; FETCH_SIZE doubled
on gfx950 as MI355X_MICROARCH.md prescribes), algorithmic bytes from bench.block_work.
`frac` = algorithmic bytes / time / 8 TB/s; `hbm_rate` = measured traffic / time.
"""
import collections
import csv
import json
import re
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402

PEAK = 8000.0  # GB/s


def short(n):
    return re.sub(r"\(.*", "", re.sub(r"^void ", "", n))


def last_step(rows, which=-2):
    rows = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adamw_ema_kernel" in r["Kernel_Name"]]
    return rows[idx[which - 1] + 1: idx[which] + 1]


def classify(seq):
    """-> list of (family or None, block or None) per dispatch."""
    out = []
    phase = "fwd"
    blk = -1                 # forward: block whose dws_fwd was seen last
    pending = None           # index of the last unassigned gemm_nn (forward)
    pending_gram = []        # y1-free blocks: gemm_tn (Gram of a0) + bn1_gram_finalize before the stencil
    started_fwd_blocks = False
    after_pool = False
    bblk = 9                 # backward: current block
    after_dws = False
    started_bwd_blocks = False
    for i, r in enumerate(seq):
        n = short(r["Kernel_Name"])
        fam, b = None, None
        if "poisson_bwd" in n:
            phase = "bwd"
        if n.startswith("stem_out"):
            started_fwd_blocks = True
        if phase == "fwd":
            if n.startswith("dw_spatial_fwd") or n.startswith("dw_spatial_fwd_rc"):
                blk += 1
                fam, b = "dws_fwd", blk
                if pending is not None:
                    out[pending] = ("pw_fwd", blk)
                    pending = None
                for j in pending_gram:                      # y1-free block: the Gram pass + BatchNorm-1 finalisation stand in for conv_pw
                    out[j] = ("pw_fwd", blk)
                pending_gram = []
            elif started_fwd_blocks and (n.startswith("gemm_tn") or n.startswith("bn1_gram_finalize")):
                pending_gram.append(i)
            elif n.startswith("dw_temporal_fwd"):
                fam, b = "dwt_fwd", blk
            elif n.startswith("se_pool"):
                fam, b = "se_pool", blk
                after_pool = True
            elif (n.startswith("gemm_nn") or n.startswith("gemm_kd")):
                if after_pool:
                    fam, b = "pwl_fwd", blk
                    after_pool = False
                else:
                    pending = i
            elif n.startswith("residual_fwd") or n.startswith("shortcut_stats"):
                fam, b = "resid_fwd", blk
        else:
            if n.startswith("residual_bwd_reduce"):
                bblk -= 1
                after_dws = False
                started_bwd_blocks = True
                fam, b = "resid_bwd", bblk
            elif not started_bwd_blocks:
                pass
            elif n.startswith("residual_bwd"):
                fam, b = "resid_bwd", bblk
            elif n.startswith("dw_temporal_bwd"):
                fam, b = "dwt_bwd", bblk
            elif n.startswith("bn3_bwd_reduce"):
                fam, b = "bn3_reduce", bblk
            elif n.startswith("dw_spatial_bwd"):
                fam, b = "dws_bwd", bblk
                after_dws = True
            elif n.startswith("pw_bwd_fused"):
                fam, b = "pw_dgrad", bblk
            elif (n.startswith("gemm_nn") or n.startswith("gemm_kd")):
                fam, b = ("pw_dgrad" if after_dws else "pwl_dgrad"), bblk
            elif n.startswith("gemm_tn"):
                fam, b = ("pw_wgrad" if after_dws else "pwl_wgrad"), bblk
            if bblk < 0 or bblk > 8:
                fam, b = None, None
        out.append((fam, b))
    return out


def pmc_step(dirname, counter):
    d = Path(dirname)
    trace = list(csv.DictReader(open(d / "p_kernel_trace.csv")))
    step = last_step(trace, which=-1)
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(d / "p_counter_collection.csv")):
        if r["Counter_Name"] == counter:
            per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    return [(short(r["Kernel_Name"]), per.get(r["Dispatch_Id"], 0.0) * 1024.0) for r in step]


def align(names, pmc):
    """PMC values in the order of `names`; runtime copy kernels (loss read-back, ...) may sit at different places of the two
    runs' step windows and are matched separately.  None when the launch sequences differ otherwise."""
    skip = "__amd_rocclr"
    core = [v for v in pmc if not v[0].startswith(skip)]
    if [n for n, _ in core] != [n for n in names if not n.startswith(skip)]:
        return None
    it = iter(core)
    return [(n, 0.0) if n.startswith(skip) else next(it) for n in names]


def main():
    trace_csv, dir_f, dir_w, out_all, out_dws = sys.argv[1:6]
    seq = last_step(list(csv.DictReader(open(trace_csv))))
    names = [short(r["Kernel_Name"]) for r in seq]
    fetch = align(names, pmc_step(dir_f, "FETCH_SIZE"))
    write = align(names, pmc_step(dir_w, "WRITE_SIZE"))
    pmc_ok = fetch is not None and write is not None
    if not pmc_ok:
        print("warning: PMC launch sequences differ from the trace; traffic columns left empty", file=sys.stderr)
    cls = classify(seq)
    fused = {b for (f, b), n in zip(cls, names) if n.startswith("pw_bwd_fused")}
    y1_free = {b for (f, b), n in zip(cls, names) if f == "pw_fwd" and n.startswith("bn1_gram_finalize")}
    work = bench.block_work(32, 32, 36, 64, 7, fused, 2, y1_free)
    table = collections.OrderedDict()
    other_us = 0.0
    for i, (r, (fam, b)) in enumerate(zip(seq, cls)):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if fam is None:
            other_us += us
            continue
        e = table.setdefault((b, fam), {"block": b, "family": fam, "us": 0.0, "launches": 0, "kernels": [], "vgpr": 0,
                                        "scratch_bytes": 0, "lds_bytes": 0, "traffic_bytes": 0.0 if pmc_ok else None})
        e["us"] += us
        e["launches"] += 1
        e["kernels"].append(names[i])
        e["vgpr"] = max(e["vgpr"], int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]))
        e["scratch_bytes"] = max(e["scratch_bytes"], int(r["Scratch_Size"]))
        e["lds_bytes"] = max(e["lds_bytes"], int(r["LDS_Block_Size"]))
        if pmc_ok:
            e["traffic_bytes"] += 2.0 * fetch[i][1] + write[i][1]
    rows = []
    for (b, fam), e in sorted(table.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        alg = work[b]["work"].get(fam)
        e["us"] = round(e["us"], 1)
        if alg:
            e["algorithmic_bytes"] = alg[0]
            e["achieved_gbs"] = round(alg[0] / e["us"] / 1e3, 1)
            e["frac"] = round(alg[0] / e["us"] / 1e3 / PEAK, 4)
            if alg[1]:
                e["tflops"] = round(alg[1] / e["us"] / 1e6, 1)
        if e["traffic_bytes"] is not None:
            e["hbm_rate_gbs"] = round(e["traffic_bytes"] / e["us"] / 1e3, 1)
            e["traffic_bytes"] = int(e["traffic_bytes"])
        e["shape"] = {k: work[b][k] for k in ("stride", "cin", "cmid", "cout", "in_hw", "out_hw", "m_in", "m_out")}
        rows.append(e)
    step_us = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in seq) / 1e3
    span_us = (int(seq[-1]["End_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1e3
    total_traffic = sum(2.0 * f[1] + w[1] for f, w in zip(fetch, write)) if pmc_ok else None
    meta = {"source": "rocprofv3 kernel trace (durations, last step) + FETCH_SIZE / WRITE_SIZE passes (traffic), "
                      "bench.py training step B=32 T=32 36x64 bf16",
            "lib_sha16": bench.lib_sha16(), "launches_per_step": len(seq), "kernel_time_sum_us": round(step_us, 1),
            "step_span_us_under_tracing": round(span_us, 1), "unassigned_us": round(other_us, 1),
            "hbm_traffic_bytes_per_step": int(total_traffic) if total_traffic else None, "peak_gbs": PEAK}
    json.dump({"meta": meta, "rows": rows}, open(out_all, "w"), indent=1)
    json.dump({"meta": meta, "rows": [r for r in rows if r["family"] in ("dws_fwd", "dws_bwd")]}, open(out_dws, "w"), indent=1)
    print(json.dumps(meta))
    print(f"{'family':10s} blk {'us':>8s} {'alg MB':>8s} {'frac':>6s} {'HBM MB':>8s} {'GB/s':>7s} vgpr scratch  kernel")
    for e in rows:
        print(f"{e['family']:10s} {e['block']:3d} {e['us']:8.1f} {e.get('algorithmic_bytes', 0) / 1e6:8.1f} "
              f"{e.get('frac', 0):6.3f} {(e['traffic_bytes'] or 0) / 1e6:8.1f} {e.get('hbm_rate_gbs', 0):7.1f} "
              f"{e['vgpr']:4d} {e['scratch_bytes']:5d}  {e['kernels'][0][:50]}")


if __name__ == "__main__":
    main()
