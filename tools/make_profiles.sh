#!/bin/bash
# Round profiles in one go on the GPU box (run from the repo root):  bash tools/make_profiles.sh r2
#   <tag>_bench.json, <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the same command),
#   <tag>_pmc_traffic.json (separate FETCH_SIZE / WRITE_SIZE passes), <tag>_per_block.json, <tag>_dws_per_block.json
set -u
tag=${1:-r2}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 20 --warmup 5 > $out/bench.log 2>&1
tail -n 1 $out/bench.log > $out/${tag}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-rooflines > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_f -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-rooflines > $out/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_w -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-rooflines > $out/pmc_w.log 2>&1
cp $out/trace/t_kernel_stats.csv $out/${tag}_kernel_stats.csv
python3 tools/pmc_traffic.py $out/pmc_f $out/pmc_w $out/${tag}_pmc_traffic.json > $out/pmc_traffic.txt 2>&1
python3 tools/per_block.py $out/trace/t_kernel_trace.csv $out/pmc_f $out/pmc_w $out/${tag}_per_block.json $out/${tag}_dws_per_block.json > $out/per_block.txt 2>&1
# the bench line again, now that the traffic file matches this library build
cp $out/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json
python3 bench.py --steps 20 --warmup 5 > $out/bench2.log 2>&1
tail -n 1 $out/bench2.log > $out/${tag}_bench.json
rm -rf $out/pmc_f/p_counter_collection.csv.bak
ls -la $out | head -n 30
