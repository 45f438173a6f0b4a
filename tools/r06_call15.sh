#!/bin/bash
# round 6, call 15: records per lane per step in the partition aggregation (specialised build, -DSSGPU_PART_ROWS through SSGPU_RTC_FLAGS)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r06_part_rows.txt
: > $out
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.4f kernel_ms %.4f frac %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))'
b() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-traffic --no-configs "$@" 2>/dev/null | tail -1 | python -c "$pick"; }
for rep in 1 2; do
for q in group3 group; do
  echo "$q rows/lane 2 prefetch 0 rep $rep: $(b --query $q --opts part_prefetch=0)" >> $out
  echo "$q rows/lane 4 prefetch 0 rep $rep: $(SSGPU_RTC_FLAGS=-DSSGPU_PART_ROWS=4 b --query $q --opts part_prefetch=0)" >> $out
  echo "$q rows/lane 1 prefetch 1 rep $rep: $(SSGPU_RTC_FLAGS=-DSSGPU_PART_ROWS=1 b --query $q --opts part_prefetch=1)" >> $out
  echo "$q rows/lane 3 prefetch 0 rep $rep: $(SSGPU_RTC_FLAGS=-DSSGPU_PART_ROWS=3 b --query $q --opts part_prefetch=0)" >> $out
done
done
cat $out
SSGPU_RTC_FLAGS=-DSSGPU_PART_ROWS=4 bash tools/kstats.sh r06_rows4_group3 python bench.py --query group3 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-configs --opts part_prefetch=0
