#!/bin/bash
# one-rank RCCL trace of the ten-readout step -> gpurun_out/r6_ddp_single_rank_trace.{txt,json} (tools/ddp_trace_summary.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ddp_tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/ddp_tr -o t -- python3 $R/bench.py --ddp-single-rank --mice 10 --steps 5 --warmup 2 --no-cpu-baseline --no-rooflines --no-inference --no-other-configs --no-fwd-bwd > $R/gpurun_out/r6_ddp_trace.log 2>&1
cd $R
python3 tools/ddp_trace_summary.py /tmp/ddp_tr gpurun_out/r6_ddp_single_rank_trace.json > gpurun_out/r6_ddp_single_rank_trace.txt 2>&1
cat gpurun_out/r6_ddp_single_rank_trace.txt | cut -c1-220
