#!/bin/bash
# Round 4, GPU call D: suite (heavy hitters, after the Sort experiment was removed), the skewed 100 M-row GroupAggregate, sharded group step
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04d
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q -n 4 ) > $OUT/suite.log 2>&1
tail -5 $OUT/suite.log
python tools/skew_bench.py 100000000 1 > $OUT/skew_spec.json 2> $OUT/skew_spec.err; tail -1 $OUT/skew_spec.json | cut -c1-900
python tools/skew_bench.py 100000000 0 > $OUT/skew_interp.json 2> $OUT/skew_interp.err; tail -1 $OUT/skew_interp.json | cut -c1-900
python bench.py --query group --force-distributed --rows 12500000 --steps 200 --warmup 20 --no-cpu-baseline 2> $OUT/dist1_group.err | grep "^{" > $OUT/dist1_group_12m5.json
grep -ho '"ms_per_step": [0-9.]*' $OUT/dist1_group_12m5.json | head -1
python bench.py --query sort --steps 20 --warmup 3 --no-cpu-baseline > $OUT/sort_line.json 2> $OUT/sort.err; grep -o '"kernel_ms": [0-9.]*' $OUT/sort_line.json
