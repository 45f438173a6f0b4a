#!/bin/bash
# round 6, call 14: partition aggregation with its record loads a trip ahead -- parity + A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_dense_gpu.py tests/test_00_configs_gpu.py -m gpu -x -q --timeout 300 ) > gpurun_out/r06_call14_tests.log 2>&1
tail -3 gpurun_out/r06_call14_tests.log
out=gpurun_out/r06_part_prefetch.txt
: > $out
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.4f kernel_ms %.4f frac %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))'
b() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-traffic --no-configs "$@" 2>/dev/null | tail -1 | python -c "$pick"; }
for rep in 1 2 3; do
for q in group3 group; do
  echo "$q prefetch 0 rep $rep: $(b --query $q --opts part_prefetch=0)" >> $out
  echo "$q prefetch 1 rep $rep: $(b --query $q --opts part_prefetch=1)" >> $out
done
done
cat $out
bash tools/kstats.sh r06_prefetch_group3 python bench.py --query group3 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-configs
bash tools/kstats.sh r06_prefetch_group python bench.py --query group --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-configs
