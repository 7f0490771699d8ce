#!/bin/bash
# Per-kernel hardware counters of one training step (five rocprofv3 --pmc passes over the same command, merged by tools/pmc_table.py):
#   bash tools/pmc_step.sh r5 ["script.py args"]       -> gpurun_out/<tag>/<tag>_kernel_counters.json + .txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r5}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD=${2:-"$R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-rooflines --no-inference --no-fwd-bwd --no-other-configs"}
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmcs_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcs_$i -o p -- python3 $CMD > $O/pmcs_$i.log 2>&1
done
python3 $R/tools/pmc_table.py $O/${tag}_kernel_counters.json /tmp/pmcs_1 /tmp/pmcs_2 /tmp/pmcs_3 /tmp/pmcs_4 /tmp/pmcs_5 > $O/${tag}_kernel_counters.txt 2>&1
head -c 3000 $O/${tag}_kernel_counters.txt
