#!/bin/bash
# round 6, call 9: software-pipelined plain scatter -- parity, then A/B against the plain loop
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_dense_gpu.py tests/test_00_configs_gpu.py -m gpu -x -q --timeout 300 ) > gpurun_out/r06_call9_tests.log 2>&1
tail -5 gpurun_out/r06_call9_tests.log
out=gpurun_out/r06_pscat_pipe.txt
: > $out
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.4f kernel_ms %.4f frac %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))'
b() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-traffic --no-configs "$@" 2>/dev/null | tail -1 | python -c "$pick"; }
for rep in 1 2; do
for q in group3 group; do
  for g in "0 0" "1 0" "0 3" "1 3"; do
    set -- $g
    echo "$q pipe $1 rows $2 rep $rep: $(b --query $q --opts pscat_pipe=$1,pscat_rows=$2)" >> $out
  done
done
done
cat $out
bash tools/kstats.sh r06_pipe_group3 python bench.py --query group3 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-configs
bash tools/kstats.sh r06_pipe_group python bench.py --query group --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-configs
