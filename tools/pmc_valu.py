#!/usr/bin/env python3
"""Per-kernel VALU / LDS utilisation from one rocprofv3 pass:
  rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS \
            SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d <dir> -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
  python tools/pmc_valu.py <dir>/p_counter_collection.csv [out.json]
SQ_BUSY_CYCLES sums the 32 shader engines; SQ_ACTIVE_INST_* count quad-cycles summed over the SIMDs (MI355X_MICROARCH.md)."""
import collections, csv, json, re, sys
per = collections.defaultdict(lambda: collections.defaultdict(float))
nd = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"^void ", "", re.sub(r"\(.*", "", r["Kernel_Name"]))
    per[name][r["Counter_Name"]] += float(r["Counter_Value"])
    nd[name].add(r["Dispatch_Id"])
rows = []
for k, v in per.items():
    cyc = v.get("SQ_BUSY_CYCLES", 0) / 32.0            # shader cycles the kernel was resident, all dispatches
    if cyc <= 0:
        continue
    valu = v.get("SQ_ACTIVE_INST_VALU", 0) * 4.0 / 1024.0 / cyc
    lds = v.get("SQ_LDS_IDX_ACTIVE", 0) / 256.0 / cyc
    rows.append(dict(kernel=k, dispatches=len(nd[k]), mcycles=cyc / 1e6, valu_busy=valu, lds_busy=lds,
                     valu_insts_m=v.get("SQ_INSTS_VALU", 0) / 1e6, cyc_per_valu=(v.get("SQ_ACTIVE_INST_VALU", 0) * 4.0 / max(v.get("SQ_INSTS_VALU", 1), 1)),
                     lds_conflict_frac=v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
rows.sort(key=lambda r: -r["mcycles"])
tot = sum(r["mcycles"] for r in rows)
for r in rows[:45]:
    print(f"{r['kernel'][:70]:70s} n={r['dispatches']:4d} {r['mcycles']:8.2f} Mcyc ({100*r['mcycles']/tot:4.1f}%) VALU {100*r['valu_busy']:5.1f}%  "
          f"LDS {100*r['lds_busy']:5.1f}%  cyc/VALU {r['cyc_per_valu']:4.2f}  conflicts {100*r['lds_conflict_frac']:4.1f}%")
print(f"total {tot:.1f} Mcycles; VALU-busy weighted {100*sum(r['mcycles']*r['valu_busy'] for r in rows)/tot:.1f}%")
if len(sys.argv) > 2:
    json.dump(rows, open(sys.argv[2], "w"), indent=1)
