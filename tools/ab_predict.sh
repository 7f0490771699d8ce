#!/bin/bash
# A/B of library variants on the 7-fold inference leg: bash tools/ab_predict.sh <dtype> <lib dir> ...   ("base" = the in-tree build)
# Every variant is timed in its own process (DWN_LIB_PATH selects the binary; the source-hash check is for the in-tree one).
dtype=$1; shift
mkdir -p gpurun_out/ab
for v in "$@"; do
  if [ "$v" = base ]; then unset DWN_LIB_PATH; else export DWN_LIB_PATH=$PWD/build_ab/$v/libdwiseneuro_hip.so; fi
  python3 - "$dtype" > gpurun_out/ab/predict_${dtype}_$v.json 2> gpurun_out/ab/predict_${dtype}_$v.err <<'P'
import json, sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from bench_predict import ensemble_bench
r = ensemble_bench(dtype=sys.argv[1], windows=90, repeats=3)
print(json.dumps({k: r[k] for k in ("trials_per_s", "ms_per_trial", "finite")}))
P
  echo "$v $(cat gpurun_out/ab/predict_${dtype}_$v.json) $(tail -1 gpurun_out/ab/predict_${dtype}_$v.err | cut -c1-200)"
done
