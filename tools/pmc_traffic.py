#!/usr/bin/env python3
"""Summarise HBM traffic per kernel from two rocprofv3 PMC passes over the same command.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir_f> -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d <dir_w> -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python tools/pmc_traffic.py <dir_f> <dir_w> profiles/r1c_pmc_traffic.json

Units and corrections as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE / WRITE_SIZE are KiB, summed over the
counter's instances per dispatch; on gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes, so reads are doubled;
WRITE_SIZE is exact for streaming stores and float atomics.  Infinity-Cache hits are counted as traffic.
"""
import collections
import csv
import json
import re
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))

FAMILIES = {
    "dws_bwd": ("dw_spatial_bwd",), "dws_fwd": ("dw_spatial_fwd",), "dwt_bwd": ("dw_temporal_bwd",),
    "dwt_fwd": ("dw_temporal_fwd",), "bn3_reduce": ("bn3_bwd_reduce",), "se_pool": ("se_pool",),
    "gemm_nn": ("gemm_nn_kernel", "gemm_nn_xl_kernel", "gemm_kd_kernel"), "gemm_tn": ("gemm_tn_kernel",),
}


def per_kernel(path, counter):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"^void ", "", re.sub(r"\(.*", "", r["Kernel_Name"]))
            per[name][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: (len(v), sum(v.values()) * 1024.0) for k, v in per.items()}      # (dispatches, bytes)


def main():
    dir_f, dir_w, out = sys.argv[1:4]
    fetch = per_kernel(f"{dir_f}/p_counter_collection.csv", "FETCH_SIZE")
    write = per_kernel(f"{dir_w}/p_counter_collection.csv", "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        n = max(fetch.get(k, (0, 0))[0], write.get(k, (0, 0))[0])
        rd = 2.0 * fetch.get(k, (0, 0.0))[1]          # gfx950: FETCH_SIZE reports half of the bytes
        wr = write.get(k, (0, 0.0))[1]
        kernels[k] = {"dispatches": n, "read_bytes": rd, "write_bytes": wr}
    fams = {}
    for fam, pats in FAMILIES.items():
        ks = [k for k in kernels if any(p in k for p in pats)]
        n = sum(kernels[k]["dispatches"] for k in ks)
        if n:
            rd = sum(kernels[k]["read_bytes"] for k in ks)
            wr = sum(kernels[k]["write_bytes"] for k in ks)
            fams[fam] = {"dispatches": n, "read_bytes_per_launch": rd / n, "write_bytes_per_launch": wr / n,
                         "traffic_bytes_per_launch": (rd + wr) / n}
    # the bench's own family names (pw_fwd ... pw_wgrad, resid_*): the launches of the last profiled step assigned to
    # (block, family) from the launch order, as tools/per_block.py does
    import per_block as pb
    fstep, wstep = pb.pmc_step(dir_f, "FETCH_SIZE"), pb.pmc_step(dir_w, "WRITE_SIZE")
    d = Path(dir_f)
    seq = pb.last_step(list(csv.DictReader(open(d / "p_kernel_trace.csv"))), which=-1)
    wal = pb.align([n for n, _ in fstep], wstep)
    if wal is not None:
        per_step = collections.defaultdict(lambda: [0, 0.0])
        for (fam, _blk), (_, fb), (_, wb) in zip(pb.classify(seq), fstep, wal):
            if fam is not None:
                per_step[fam][0] += 1
                per_step[fam][1] += 2.0 * fb + wb
        for fam, (n, by) in per_step.items():
            fams.setdefault(fam, {})
            fams[fam].update({"launches_per_step": n, "traffic_bytes_per_step": by})
            fams[fam].setdefault("traffic_bytes_per_launch", by / n)
        fams["_step_total"] = {"traffic_bytes_per_step": sum(2.0 * f[1] + w[1] for f, w in zip(fstep, wal)),
                               "launches_per_step": len(seq)}
    top = sorted(kernels.items(), key=lambda kv: -(kv[1]["read_bytes"] + kv[1]["write_bytes"]))[:40]
    import hashlib
    libp = Path(__file__).resolve().parents[1] / "sensorium_amd" / "csrc" / "libdwiseneuro_hip.so"
    sha = hashlib.sha256(libp.read_bytes()).hexdigest()[:16] if libp.exists() else None
    json.dump({"lib_sha16": sha, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1",
               "corrections": "KiB units; reads = 2 x FETCH_SIZE on gfx950; Infinity-Cache hits included",
               "families": fams, "kernels": dict(top)}, open(out, "w"), indent=1)
    for fam, v in fams.items():
        if "dispatches" in v:
            print(f"{fam:12s} launches {v['dispatches']:4d}  read {v['read_bytes_per_launch'] / 1e6:9.1f} MB  "
                  f"write {v['write_bytes_per_launch'] / 1e6:9.1f} MB per launch")
    for fam, v in fams.items():
        if "traffic_bytes_per_step" in v:
            print(f"{fam:12s} {v['launches_per_step']:4d} launches/step  {v['traffic_bytes_per_step'] / 1e9:8.3f} GB/step")


if __name__ == "__main__":
    main()
