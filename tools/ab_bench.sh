#!/bin/bash
# A/B of library variants on the training step: bash tools/ab_bench.sh <lib dir under build_var | base> ...  (each in its own process;
# prints ms/step and the per-family ms of a --profile-all run)
for v in "$@"; do
  if [ "$v" = base ]; then unset DWN_LIB_PATH; else export DWN_LIB_PATH=$PWD/build_var/$v/libdwiseneuro_hip.so; fi
  python3 bench.py --no-inference --no-other-configs --no-cpu-baseline --profile-all 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); f = d['family_ms_per_step']
print('$v', d['ms_per_step'], ' '.join('%s=%.3f' % (k, f[k]) for k in sorted(f)))"
done
