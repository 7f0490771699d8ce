#!/usr/bin/env python3
"""Merge rocprofv3 --pmc passes (one counter_collection.csv per pass, same program) into one per-kernel table.
usage: python3 tools/pmc_table.py out.json <pass dir> [<pass dir> ...] [--match substr]
Per kernel name: dispatches, and every counter averaged per dispatch; derived columns where the inputs are present:
  fetch_MB (FETCH_SIZE KiB x 2: gfx950 tallies 128-byte reads at 64 B, MI355X_MICROARCH.md), write_MB, l2_hit = HIT / (HIT + MISS),
  valu_busy = SQ_ACTIVE_INST_VALU*4 / (SQ_BUSY_CYCLES/32 * 1024), wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES, ..."""
import collections, csv, glob, json, re, sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
match = ""
if "--match" in sys.argv:
    match = sys.argv[sys.argv.index("--match") + 1]
    args = [a for a in args if a != match]
out, dirs = args[0], args[1:]
per = collections.defaultdict(lambda: collections.defaultdict(float))
nd = collections.defaultdict(lambda: collections.defaultdict(set))
meta = {}
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"^void ", "", re.sub(r"\(.*", "", r["Kernel_Name"]))
            if match and match not in name:
                continue
            c = r["Counter_Name"]
            per[name][c] += float(r["Counter_Value"])
            nd[name][c].add((f, r["Dispatch_Id"]))
            meta[name] = dict(vgpr=r.get("VGPR_Count"), accum_vgpr=r.get("Accum_VGPR_Count"), sgpr=r.get("SGPR_Count"),
                              lds=r.get("LDS_Block_Size"), scratch=r.get("Scratch_Size"), wg=r.get("Workgroup_Size"), grid=r.get("Grid_Size"))
rows = []
for k, v in per.items():
    row = dict(kernel=k, **{m: meta[k][m] for m in meta[k]})
    avg = {c: v[c] / max(len(nd[k][c]), 1) for c in v}
    row["dispatches"] = max(len(s) for s in nd[k].values())
    row["counters_per_dispatch"] = {c: round(a, 1) for c, a in sorted(avg.items())}
    g = avg.get
    if "FETCH_SIZE" in avg: row["fetch_MB"] = round(g("FETCH_SIZE") * 1024 * 2 / 1e6, 2)
    if "WRITE_SIZE" in avg: row["write_MB"] = round(g("WRITE_SIZE") * 1024 / 1e6, 2)
    if "TCC_HIT_sum" in avg and "TCC_MISS_sum" in avg: row["l2_hit"] = round(g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1), 4)
    if "SQ_BUSY_CYCLES" in avg:
        cyc = g("SQ_BUSY_CYCLES") / 32.0
        row["busy_kcycles"] = round(cyc / 1e3, 1)
        if "SQ_ACTIVE_INST_VALU" in avg: row["valu_busy"] = round(g("SQ_ACTIVE_INST_VALU") * 4 / 1024 / cyc, 4)
        if "SQ_LDS_IDX_ACTIVE" in avg: row["lds_busy"] = round(g("SQ_LDS_IDX_ACTIVE") / 256 / cyc, 4)
    if "SQ_WAVE_CYCLES" in avg:
        wc = max(g("SQ_WAVE_CYCLES"), 1)
        for c, nme in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst_any"), ("SQ_ACTIVE_INST_ANY", "active_any"),
                       ("SQ_WAIT_INST_LDS", "wait_inst_lds"), ("SQ_ACTIVE_INST_VALU", "active_valu"), ("SQ_ACTIVE_INST_LDS", "active_lds"),
                       ("SQ_ACTIVE_INST_VMEM", "active_vmem"), ("SQ_ACTIVE_INST_FLAT", "active_flat"), ("SQ_ACTIVE_INST_SCA", "active_sca")):
            if c in avg: row[nme + "_frac_of_wave_cycles"] = round(g(c) / wc, 4)
    if "SQ_INSTS_MFMA" in avg and "SQ_INSTS_VALU" in avg: row["mfma_per_valu"] = round(g("SQ_INSTS_MFMA") / max(g("SQ_INSTS_VALU"), 1), 4)
    if "SQ_LDS_BANK_CONFLICT" in avg and "SQ_LDS_IDX_ACTIVE" in avg: row["lds_conflict_frac"] = round(g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1), 4)
    if "SQ_INSTS_VALU" in avg and "SQ_ACTIVE_INST_VALU" in avg: row["cyc_per_valu"] = round(g("SQ_ACTIVE_INST_VALU") * 4 / max(g("SQ_INSTS_VALU"), 1), 3)
    rows.append(row)
rows.sort(key=lambda r: -r.get("busy_kcycles", r.get("fetch_MB", 0)))
json.dump(rows, open(out, "w"), indent=1)
for r in rows[:60]:
    print({k: v for k, v in r.items() if k != "counters_per_dispatch"})
