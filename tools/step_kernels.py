#!/usr/bin/env python3
"""Per-block durations of the E-wide forward kernels of the last traced training step (rocprofv3 kernel trace of bench.py).
    python tools/step_kernels.py <t_kernel_trace.csv> [out.json]"""
import csv
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import per_block  # noqa: E402


def main():
    seq = per_block.last_step(list(csv.DictReader(open(sys.argv[1]))))
    cls = per_block.classify(seq)
    out = {}
    for r, (fam, b) in zip(seq, cls):
        if fam is None:
            continue
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        out.setdefault(fam, {}).setdefault(b, 0.0)
        out[fam][b] = round(out[fam][b] + us, 1)
    span = (int(seq[-1]["End_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1e3
    ksum = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seq) / 1e3
    print(f"launches {len(seq)}  kernel-time sum {ksum:.1f} us  span {span:.1f} us")
    for fam in sorted(out):
        print(f"{fam:10s}", " ".join(f"{out[fam].get(b, 0):7.1f}" for b in range(9)), f"  sum {sum(out[fam].values()):8.1f}")
    if len(sys.argv) > 2:
        Path(sys.argv[2]).write_text(json.dumps({"launches": len(seq), "kernel_time_sum_us": ksum, "span_us": span, "families": out}, indent=1))


if __name__ == "__main__":
    main()
