#!/bin/bash
# GPU clock / power while the metric step runs (is the part at its power limit?): samples rocm-smi every 0.5 s beside bench.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
rocm-smi --showclocks --showpower --showperflevel 2>&1 | grep -E "sclk|mclk|Power|Perf" | head -8
python3 bench.py --steps 2000 --warmup 5 --no-inference --no-other-configs --no-cpu-baseline --no-rooflines --no-fwd-bwd > /tmp/clk_bench.json 2>/dev/null &
BP=$!
sleep 18
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power \(|Average Graphics|Current Socket" | tr '\n' ' '; echo; sleep 0.5; done
wait $BP
cut -c1-120 /tmp/clk_bench.json | head -1; python3 -c "import json; d=json.loads(open('/tmp/clk_bench.json').readline()); print('ms/step', d['ms_per_step'])"
