#!/bin/bash
# round 6, call 10: dense partitions over row ranges (scatter of range k + 1 beside the aggregation of range k)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_dense_gpu.py tests/test_00_configs_gpu.py -m gpu -x -q --timeout 300 ) > gpurun_out/r06_call10_tests.log 2>&1
tail -5 gpurun_out/r06_call10_tests.log
out=gpurun_out/r06_overlap_ab.txt
: > $out
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.4f kernel_ms %.4f frac %.3f checked %s" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("result_checked", d.get("checked"))))'
b() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-traffic --no-configs "$@" 2>/dev/null | tail -1 | python -c "$pick"; }
for rep in 1 2; do
for q in group3 group; do
  for k in 1 2 4 8; do
    echo "$q row ranges $k rep $rep: $(b --query $q --opts part_overlap=$k)" >> $out
  done
done
done
cat $out
bash tools/kstats.sh r06_overlap_group3 python bench.py --query group3 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-configs
