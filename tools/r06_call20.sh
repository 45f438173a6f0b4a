#!/bin/bash
# round 6, call 20: partition records read with non-temporal loads (A/B through SSGPU_RTC_FLAGS)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r06_part_nt.txt
: > $out
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.4f kernel_ms %.4f frac %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))'
b() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-traffic --no-configs "$@" 2>/dev/null | tail -1 | python -c "$pick"; }
for rep in 1 2 3; do
for q in group3 group; do
  echo "$q records plain loads rep $rep: $(b --query $q)" >> $out
  echo "$q records nt loads    rep $rep: $(SSGPU_RTC_FLAGS=-DSSGPU_PART_NT=1 b --query $q)" >> $out
done
done
cat $out
SSGPU_RTC_FLAGS=-DSSGPU_PART_NT=1 bash tools/kstats.sh r06_partnt_group3 python bench.py --query group3 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-configs
