#!/bin/bash
# average duration of kernels matching a pattern in one training step, for library variants:  bash tools/kavg.sh "<regex>" base <variant> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = base ]; then unset DWN_LIB_PATH; else export DWN_LIB_PATH=$R/build_ab/$v/libdwiseneuro_hip.so; fi
  rm -rf /tmp/ka_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ka_$v -o t -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-rooflines --no-inference --no-other-configs > /tmp/ka_$v.log 2>&1
  f=$(find /tmp/ka_$v -name "t_kernel_stats.csv" | head -1)
  echo "== $v"; python3 - "$f" "$pat" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print(f"{r['Name'][:90]:90s} calls {int(r['Calls']):4d} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
done
