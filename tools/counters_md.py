#!/usr/bin/env python3
"""<tag>_kernel_counters.json (tools/pmc_table.py) -> the markdown table committed under profiles/ (one row per kernel of the step, the
load-path model of DESIGN.md section 7 beside the measured busy time).  usage: python3 tools/counters_md.py in.json out.md "title" """
import json
import sys

rows = json.load(open(sys.argv[1]))
title = sys.argv[3] if len(sys.argv) > 3 else "per-kernel counters of one training step"
out = [f"# {title} (tools/pmc_step.sh: five separate rocprofv3 --pmc passes, per-dispatch means)",
       "# FETCH_SIZE doubled on gfx950 (guide); busy = SQ_BUSY_CYCLES/XCD-count in kcycles; valu = SQ_ACTIVE_INST_VALU*4/SIMD-cycles; wait = SQ_WAIT_ANY/SQ_WAVE_CYCLES;",
       "# L2 hit MB = TCC_HIT_sum x 128 B; model us = (fetch + write) MB / 6 TB/s + L2-hit MB / 17 TB/s (DESIGN.md section 7); VGPR as rocprofv3 reports it (allocation granules)",
       "",
       "| kernel | n/2 steps | VGPR | scratch B | fetch MB | write MB | L2 hit | L2-hit MB | busy kcyc | busy us @2.4 GHz | model us | VALU busy | LDS busy | wait_any | LDS conflict | MFMA/VALU | VALU Minst |",
       "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    c = r.get("counters_per_dispatch", {})
    if "busy_kcycles" not in r:
        continue
    hit_mb = c.get("TCC_HIT_sum", 0) * 128 / 1e6
    f, w = r.get("fetch_MB", 0), r.get("write_MB", 0)
    model = (f + w) / 6.0 + hit_mb / 17.0
    busy_us = r["busy_kcycles"] / 2.4
    if busy_us < 5:
        continue
    out.append(f"| `{r['kernel'][:70]}` | {r['dispatches']} | {r.get('vgpr')} | {r.get('scratch')} | {f:.0f} | {w:.0f} | {r.get('l2_hit', 0):.2f} | {hit_mb:.0f} | "
               f"{r['busy_kcycles']:.0f} | {busy_us:.0f} | {model:.0f} | {r.get('valu_busy', 0):.2f} | {r.get('lds_busy', 0):.2f} | "
               f"{r.get('wait_any_frac_of_wave_cycles', 0):.2f} | {r.get('lds_conflict_frac', 0):.2f} | {r.get('mfma_per_valu', 0):.3f} | {c.get('SQ_INSTS_VALU', 0) / 1e6:.1f} |")
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print(f"{len(out) - 6} kernels -> {sys.argv[2]}")
