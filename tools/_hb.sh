export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/hb -o hb -- python3 tools/head_bench.py > gpurun_out/hb.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/hb/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:40]:
    print(r["Name"][:120], r["Calls"], r["AverageNs"], r["Percentage"])
PY
