#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel: python tools/pmc_sum.py <dir>/x_counter_collection.csv [kernel substring]"""
import collections, csv, re, sys
per = collections.defaultdict(lambda: collections.defaultdict(float))
nd = collections.defaultdict(set)
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"])
    if pat not in name:
        continue
    per[name][r["Counter_Name"]] += float(r["Counter_Value"])
    nd[name].add(r["Dispatch_Id"])
for k, v in per.items():
    n = len(nd[k])
    print(k, "dispatches", n)
    for c, val in sorted(v.items()):
        print(f"   {c:32s} {val / n:16.0f} per dispatch")
