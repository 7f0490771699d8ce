#!/bin/bash
# Round profiles in one go on the GPU box (run from the repo root):  bash tools/make_profiles.sh r2
#   <tag>_bench.json, <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the same command),
#   <tag>_pmc_traffic.json (separate FETCH_SIZE / WRITE_SIZE passes), <tag>_per_block.json, <tag>_dws_per_block.json
set -u
tag=${1:-r3}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 20 --warmup 5 > $out/bench.log 2>&1
tail -n 1 $out/bench.log > $out/${tag}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-rooflines --no-inference > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_f -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-rooflines --no-inference --no-fwd-bwd > $out/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_w -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-rooflines --no-inference --no-fwd-bwd > $out/pmc_w.log 2>&1
cp $out/trace/t_kernel_stats.csv $out/${tag}_kernel_stats.csv
python3 tools/pmc_traffic.py $out/pmc_f $out/pmc_w $out/${tag}_pmc_traffic.json > $out/pmc_traffic.txt 2>&1
python3 tools/per_block.py $out/trace/t_kernel_trace.csv $out/pmc_f $out/pmc_w $out/${tag}_per_block.json $out/${tag}_dws_per_block.json > $out/per_block.txt 2>&1
# inference leg (BASELINE.json configs[4]): PMC traffic of one 7-fold trial, bf16 and fp32
for dt in bf16 fp32; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/ppmc_f_$dt -o p -- python3 tools/bench_predict.py --pmc-trial --windows 90 --dtype $dt > $out/ppmc_f_$dt.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/ppmc_w_$dt -o p -- python3 tools/bench_predict.py --pmc-trial --windows 90 --dtype $dt > $out/ppmc_w_$dt.log 2>&1
done
python3 tools/predict_pmc.py bf16 $out/ppmc_f_bf16 $out/ppmc_w_bf16 fp32 $out/ppmc_f_fp32 $out/ppmc_w_fp32 $out/${tag}_predict_pmc.json > $out/predict_pmc.txt 2>&1
cp $out/${tag}_predict_pmc.json profiles/${tag}_predict_pmc.json
rm -rf $out/ppmc_f_*/p_counter_collection.csv $out/ppmc_w_*/p_counter_collection.csv
# the bench line again, now that the traffic file matches this library build
cp $out/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json
python3 bench.py --steps 20 --warmup 5 > $out/bench2.log 2>&1
tail -n 1 $out/bench2.log > $out/${tag}_bench.json
rm -rf $out/pmc_f/p_counter_collection.csv.bak
ls -la $out | head -n 30
