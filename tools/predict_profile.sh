#!/bin/bash
# per-kernel time of ONE 7-fold ensemble trial (BASELINE.json configs[4]), eager launches: rocprofv3 --kernel-trace --stats
# usage: bash tools/predict_profile.sh <outdir>   (writes <outdir>/predict_<dtype>_stats.csv)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$R/gpurun_out/predict_prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for dt in bf16 fp32; do
  rm -rf /tmp/pp_$dt
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$dt -o p -- python3 $R/tools/bench_predict.py --pmc-trial --dtype $dt --windows 90 > $OUT/predict_$dt.log 2>&1
  cp $(find /tmp/pp_$dt -name "p_kernel_stats.csv" | head -1) $OUT/predict_${dt}_stats.csv
done
