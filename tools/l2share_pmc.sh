#!/bin/bash
# PMC passes over the L2-sharing probe (product build vs the shared-y1 + XCD-map experiment build): fabric bytes, L2 hit rate, SQ view
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/l2pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
export PROBE_QUICK=1
for v in base shy1 shy1x; do
  if [ $v = base ]; then unset DWN_LIB_PATH; arg=""; else export DWN_LIB_PATH=$R/build_ab/$v/libdwiseneuro_hip.so; arg=shared; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${v}_f -o p -- python3 $R/tools/l2share_probe.py $arg > $O/${v}_f.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/${v}_h -o p -- python3 $R/tools/l2share_probe.py $arg > $O/${v}_h.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/${v}_s -o p -- python3 $R/tools/l2share_probe.py $arg > $O/${v}_s.log 2>&1
  python3 $R/tools/pmc_table.py $O/${v}_table.json $O/${v}_f $O/${v}_h $O/${v}_s --match dw_spatial > $O/${v}_table.txt 2>&1
  rm -rf $O/${v}_f $O/${v}_h $O/${v}_s
done
tail -n 8 $O/*_table.txt
