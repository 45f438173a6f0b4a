#!/bin/bash
# round 6, call 5: split dense records -- parity (dense tests, configs, shipped fuzz dense leg) + A/B of bytes and time
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time python -m pytest tests/test_dense_gpu.py tests/test_00_configs_gpu.py -m gpu -x -q ) > gpurun_out/r06_call5_tests.log 2>&1
tail -5 gpurun_out/r06_call5_tests.log
out=gpurun_out/r06_split_ab.txt
: > $out
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.4f kernel_ms %.4f frac %.3f checked %s" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("checked")))'
b() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-traffic --no-configs "$@" 2>/dev/null | tail -1 | python -c "$pick"; }
for rep in 1 2; do
  for q in group3 group; do
    echo "$q whole records  nt=1 rep $rep: $(b --query $q --opts part_split=0)" >> $out
    echo "$q split records  nt=1 rep $rep: $(b --query $q)" >> $out
    echo "$q whole records  nt=0 rep $rep: $(SSGPU_RTC_FLAGS=-DPS_NT=0 b --query $q --opts part_split=0)" >> $out
    echo "$q split records  nt=0 rep $rep: $(SSGPU_RTC_FLAGS=-DPS_NT=0 b --query $q)" >> $out
  done
done
cat $out
# per-kernel times, both forms
for q in group3 group; do
  bash tools/kstats.sh r06_split_$q python bench.py --query $q --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-configs
  bash tools/kstats.sh r06_whole_$q python bench.py --query $q --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-configs --opts part_split=0
done
