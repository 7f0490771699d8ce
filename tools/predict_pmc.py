#!/usr/bin/env python3
"""HBM traffic of ONE 7-fold ensemble trial (BASELINE.json configs[4]) from two rocprofv3 PMC passes of
`python3 tools/bench_predict.py --pmc-trial --dtype <bf16|fp32>` (eager launches, exactly one trial):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d F -o p -- python3 tools/bench_predict.py --pmc-trial --dtype bf16
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d W -o p -- python3 tools/bench_predict.py --pmc-trial --dtype bf16
    python tools/predict_pmc.py bf16 F W [fp32 F2 W2] profiles/r3_predict_pmc.json

Units / corrections as MI355X_MICROARCH.md (HBM section): KiB, reads doubled on gfx950.  Model construction (torch fills /
copies) is excluded by kernel name; what remains is the library's launches of the one trial."""
import csv
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))


def total(path, counter):
    tot = 0.0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"]
        if n.startswith("void at::") or "rocclr" in n:
            continue
        tot += float(r["Counter_Value"]) * 1024.0
    return tot


def main():
    args = sys.argv[1:]
    out = args[-1]
    res = {}
    for i in range(0, len(args) - 1, 3):
        dt, f, w = args[i:i + 3]
        rd = 2.0 * total(f"{f}/p_counter_collection.csv", "FETCH_SIZE")
        wr = total(f"{w}/p_counter_collection.csv", "WRITE_SIZE")
        from bench_predict import eval_executed_bytes  # noqa: F401  (same accounting the bench line uses)
        res[dt] = {"read_bytes_per_trial": rd, "write_bytes_per_trial": wr, "traffic_bytes_per_trial": rd + wr}
    libp = ROOT / "sensorium_amd" / "csrc" / "libdwiseneuro_hip.so"
    res["lib_sha16"] = hashlib.sha256(libp.read_bytes()).hexdigest()[:16] if libp.exists() else None
    res["source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/bench_predict.py --pmc-trial"
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
