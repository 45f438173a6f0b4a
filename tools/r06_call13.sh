#!/bin/bash
# round 6, call 13: the wide-record merge test after the LDS-size fix; ScalarAggregate with the emit folded into the finish launch
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_00_configs_gpu.py tests/test_cursor_contract_gpu.py tests/test_seams_gpu.py tests/test_golden_gpu.py -m gpu -x -q --timeout 300 ) > gpurun_out/r06_call13_tests.log 2>&1
tail -4 gpurun_out/r06_call13_tests.log
out=gpurun_out/r06_fuse_emit.txt
: > $out
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.4f kernel_ms %.4f frac %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))'
b() { timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-traffic --no-configs "$@" 2>/dev/null | tail -1 | python -c "$pick"; }
for rep in 1 2 3; do
  for rows in 100000000 12500000 1000000; do
    echo "wide rows $rows emit as its own launch rep $rep: $(b --rows $rows --opts fuse_emit=0)" >> $out
    echo "wide rows $rows emit in the finish launch rep $rep: $(b --rows $rows --opts fuse_emit=1)" >> $out
  done
done
cat $out
