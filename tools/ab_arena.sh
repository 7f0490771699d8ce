#!/bin/bash
# A/B of the step with / without one activation arena (round-4 placement study); each bench line in its own process
mkdir -p gpurun_out/r4d
R=${GRAFT_REPO_ROOT:-$(pwd)}
for a in 0 24 0 24; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-rooflines --no-fwd-bwd --no-other-configs --arena-gb $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('arena', $a, d['ms_per_step'], d['value'])"; done
cd /tmp && export TMPDIR=/tmp
for a in 0 24; do rm -rf /tmp/tr$a; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$a -o t -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-inference --no-rooflines --no-fwd-bwd --no-other-configs --arena-gb $a > /dev/null 2>&1; python $R/tools/step_kernels.py $(find /tmp/tr$a -name t_kernel_trace.csv | head -1) $R/gpurun_out/r4d/step_arena$a.json; done
