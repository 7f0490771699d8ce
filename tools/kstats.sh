#!/bin/bash
# per-kernel averages of the training step under rocprofv3 for one or more DWN_Y1 modes: bash tools/kstats.sh free materialise
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export DWN_Y1=$v
  rm -rf /tmp/ks_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$v -o t -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-rooflines --no-inference --no-other-configs > /tmp/ks_$v.log 2>&1
  f=$(find /tmp/ks_$v -name "t_kernel_stats.csv" | head -1)
  mkdir -p $R/gpurun_out/kstats; cp $f $R/gpurun_out/kstats/${v}_kernel_stats.csv
  echo "== $v"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("dw_spatial", "gemm_tn_kernel<unsigned short, 6", "bn1_gram", "gemm_nn_kernel<unsigned short, 0, 0, 128, 1", "gemm_nn_kernel<unsigned short, 0, 0, 64, 3", "shortcut_stats", "pw_bwd_fused")):
        print(f"{n[:96]:96s} calls {int(r['Calls']):4d} avg_us {float(r['AverageNs'])/1e3:8.1f} total_ms/step {float(r['TotalDurationNs'])/1e6/8:7.3f}")
PY
done
