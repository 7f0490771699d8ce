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

FAMILIES = {
    "dws_bwd": ("dw_spatial_bwd",), "dws_fwd": ("dw_spatial_fwd",), "dwt_bwd": ("dw_temporal_bwd",),
    "dwt_fwd": ("dw_temporal_fwd",), "bn3_reduce": ("bn3_bwd_reduce",), "se_pool": ("se_pool",),
    "gemm_nn": ("gemm_nn_kernel",), "gemm_tn": ("gemm_tn_kernel",),
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
    top = sorted(kernels.items(), key=lambda kv: -(kv[1]["read_bytes"] + kv[1]["write_bytes"]))[:40]
    import hashlib
    from pathlib import Path
    libp = Path(__file__).resolve().parents[1] / "sensorium_amd" / "csrc" / "libdwiseneuro_hip.so"
    sha = hashlib.sha256(libp.read_bytes()).hexdigest()[:16] if libp.exists() else None
    json.dump({"lib_sha16": sha, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1",
               "corrections": "KiB units; reads = 2 x FETCH_SIZE on gfx950; Infinity-Cache hits included",
               "families": fams, "kernels": dict(top)}, open(out, "w"), indent=1)
    for fam, v in fams.items():
        print(f"{fam:12s} launches {v['dispatches']:4d}  read {v['read_bytes_per_launch'] / 1e6:9.1f} MB  "
              f"write {v['write_bytes_per_launch'] / 1e6:9.1f} MB per launch")


if __name__ == "__main__":
    main()
