#!/bin/bash
# One training step with and without the data-parallel machinery on ONE rank over RCCL (bench.py --ddp-single-rank), traced:
# kernel-time sum, span, idle time and the RCCL kernels.   bash tools/rccl_single_rank_profile.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for mode in plain ddp; do
  extra=""; [ $mode = ddp ] && extra="--ddp-single-rank"
  rm -rf /tmp/rp_$mode; rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_$mode -o r -- python3 $R/bench.py --steps 8 --warmup 4 $extra --no-cpu-baseline --no-inference --no-rooflines --no-fwd-bwd --no-other-configs > /tmp/rp_$mode.log 2>&1
done
python3 - <<'PY'
import csv,glob
for mode in ('plain','ddp'):
    f=glob.glob(f'/tmp/rp_{mode}/**/r_kernel_trace.csv',recursive=True)[0]
    rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
    idx=[i for i,r in enumerate(rows) if 'adamw_ema_kernel' in r['Kernel_Name']]
    res=[]
    for k in range(-5,-1):
        seq=rows[idx[k]+1:idx[k+1]+1]
        span=(int(seq[-1]['End_Timestamp'])-int(seq[0]['Start_Timestamp']))/1e3
        comm=[r for r in seq if 'oneRank' in r['Kernel_Name'] or 'ccl' in r['Kernel_Name'].lower()]
        comp=[r for r in seq if r not in comm]
        ksum=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in comp)/1e3
        csum=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in comm)/1e3
        # idle: time not covered by any compute kernel
        ev=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in comp)
        cur=ev[0][0]; idle=0
        for a,b in ev:
            if a>cur: idle+=a-cur
            cur=max(cur,b)
        res.append((len(seq),round(span,1),round(ksum,1),round(csum,1),round(idle/1e3,1)))
    print(mode,'(launches, span us, compute-kernel sum us, comm-kernel sum us, compute-idle us) per step:',res)
PY
