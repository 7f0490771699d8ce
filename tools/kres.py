#!/usr/bin/env python3
"""usage: tools/kres.py file.hip [extra hipcc flags] -- per-kernel VGPR / scratch / occupancy table (cross-compiled for
gfx950, no GPU needed)."""
import re
import subprocess
import sys

f, extra = sys.argv[1], sys.argv[2:]
out = subprocess.run(["hipcc", "-O3", "-fPIC", "--offload-arch=gfx950", "-std=c++17", "-Wno-unused-result",
                      "-Rpass-analysis=kernel-resource-usage", *extra, "-c", f, "-o", "/dev/null"],
                     capture_output=True, text=True).stderr
name, d = None, {}
for line in out.splitlines():
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        name = m.group(1); d[name] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|TotalSGPRs): (\d+)", line)
    if m and name:
        d[name][m.group(1).split()[0]] = int(m.group(2))
    if "error:" in line:
        print(line)
for k, v in d.items():
    dn = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
    g = lambda key: v.get(key, 0)
    print(f"{dn[:100]:100s} V={g('VGPRs'):3d} A={g('AGPRs'):3d} S={g('TotalSGPRs'):3d} scratch={g('ScratchSize'):3d} occ={g('Occupancy')}")
