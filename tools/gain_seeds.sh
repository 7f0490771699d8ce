for seed in 20231122 1 2 3 4; do
  for mode in free materialise; do
    echo "== seed $seed mode $mode"
    PROBE_SEED=$seed DWN_Y1=$mode timeout 300 python tools/gain_probe_metric.py 2>&1 | grep "^all"
  done
done
