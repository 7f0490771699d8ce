#!/bin/bash
# Round-6 evidence on the final library, one GPU call: make_profiles (bench line, kernel stats, PMC traffic, per-block tables, inference
# PMC), the per-kernel counter table, config-5 per-kernel stats, the co-residency (hog) measurement and the single-rank RCCL trace.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/make_profiles.sh r6 > gpurun_out/r6_make_profiles.log 2>&1
bash tools/pmc_step.sh r6 > gpurun_out/r6_pmc_step.log 2>&1
bash tools/predict_profile.sh $R/gpurun_out/r6 > gpurun_out/r6_predict_profile.log 2>&1
(cd tools/ubench && hipcc -O3 --offload-arch=gfx950 -shared -fPIC hog.hip -o libhog.so) > gpurun_out/r6_hog_build.log 2>&1
python3 tools/hog_bench.py --wgs 0 16 32 64 0 > gpurun_out/r6_hog.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ddp_tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/ddp_tr -o t -- python3 $R/bench.py --ddp-single-rank --mice 10 --steps 5 --warmup 2 --no-cpu-baseline --no-rooflines --no-inference --no-other-configs --no-fwd-bwd > $R/gpurun_out/r6_ddp_trace.log 2>&1
cd $R
python3 tools/ddp_trace_summary.py /tmp/ddp_tr gpurun_out/r6_ddp_single_rank_trace.json > gpurun_out/r6_ddp_single_rank_trace.txt 2>&1
tail -3 gpurun_out/r6_make_profiles.log; cat gpurun_out/r6_hog.txt | tail -8; head -30 gpurun_out/r6_ddp_single_rank_trace.txt
