#!/bin/bash
# round 6, call 6: best-effort test after the widen/halve fix; layout A/B (library block vs one torch allocation per column)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 300 python -m pytest tests/test_best_effort_gpu.py -m gpu -x -q --timeout 120 ) > gpurun_out/r06_call6_be.log 2>&1
tail -5 gpurun_out/r06_call6_be.log
out=gpurun_out/r06_layout_ab.txt
: > $out
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.4f kernel_ms %.4f frac %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))'
b() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-traffic --no-configs "$@" 2>/dev/null | tail -1 | python -c "$pick"; }
for rep in 1 2; do
  for q in wide filter_mat group3 group sort; do
    echo "$q layout torch   rep $rep: $(b --query $q --layout torch)" >> $out
    echo "$q layout library rep $rep: $(b --query $q --layout library)" >> $out
  done
done
cat $out
