#!/bin/bash
# Does the RCCL stream share a hardware queue with the compute stream?  One rank over RCCL (bench.py --ddp-single-rank), traced:
# default = torch's default comm stream + 4 hardware queues, hp = high-priority comm stream (sensorium_amd.ddp.init_rccl), 8 = GPU_MAX_HW_QUEUES=8.
# Prints the (Queue_Id, Stream_Id) of compute and collective kernels and how much of every collective kernel ran beside compute.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for q in default hp 8; do
  export GPU_MAX_HW_QUEUES=4 DWN_PG_DEFAULT_STREAM=1; [ $q = 8 ] && export GPU_MAX_HW_QUEUES=8; [ $q = hp ] && unset DWN_PG_DEFAULT_STREAM
  rm -rf /tmp/rq; rocprofv3 --kernel-trace --output-format csv -d /tmp/rq -o r -- python3 $R/bench.py --steps 8 --warmup 4 --ddp-single-rank --no-cpu-baseline --no-inference --no-rooflines --no-fwd-bwd --no-other-configs > /tmp/rq.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/rq/**/r_kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
print('GPU_MAX_HW_QUEUES=$q', 'columns', [c for c in rows[0].keys() if 'ueue' in c or 'tream' in c])
idx=[i for i,r in enumerate(rows) if 'adamw_ema_kernel' in r['Kernel_Name']]
seq=rows[idx[-3]+1:idx[-2]+1]
comm=[r for r in seq if 'oneRank' in r['Kernel_Name']]
comp=[r for r in seq if 'oneRank' not in r['Kernel_Name']]
print(' compute queues', sorted(set((r.get('Queue_Id'),r.get('Stream_Id')) for r in comp)), 'comm queues', sorted(set((r.get('Queue_Id'),r.get('Stream_Id')) for r in comm)))
for c in comm:
    a,b=int(c['Start_Timestamp']),int(c['End_Timestamp'])
    ov=sum(max(0,min(b,int(r['End_Timestamp']))-max(a,int(r['Start_Timestamp']))) for r in comp)
    print('  comm dur us',(b-a)/1e3,'overlapped by compute us',ov/1e3)
span=(int(seq[-1]['End_Timestamp'])-int(seq[0]['Start_Timestamp']))/1e3
print(' span',span)
PY
done
