#!/bin/bash
# Round 4, GPU call C: suite; Sort with bucket-ordered records (A/B against row-ordered); sharded group step after lazy feedback for multi-stage plans
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04c
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q -n 4 ) > $OUT/suite.log 2>&1
tail -5 $OUT/suite.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sort_stats -o t -- python $REPO/bench.py --query sort --steps 20 --warmup 3 --no-cpu-baseline > $OUT/sort_line.json 2> $OUT/sort.err
grep -o '"kernel_ms": [0-9.]*' $OUT/sort_line.json; head -8 $OUT/sort_stats/t_kernel_stats.csv | cut -c1-130
cd $REPO
python bench.py --query sort --steps 20 --warmup 3 --no-cpu-baseline --opts sort_bucketed=0 > $OUT/sort_roworder_line.json 2> $OUT/sort2.err
grep -o '"kernel_ms": [0-9.]*' $OUT/sort_roworder_line.json
for q in wide group; do
  python bench.py --query $q --force-distributed --rows 12500000 --steps 200 --warmup 20 --no-cpu-baseline 2> $OUT/dist1_${q}.err | grep "^{" > $OUT/dist1_${q}_12m5.json
  grep -ho '"ms_per_step": [0-9.]*' $OUT/dist1_${q}_12m5.json | head -1
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_group -o t -- python $REPO/bench.py --query group --force-distributed --rows 12500000 --steps 50 --warmup 5 --no-cpu-baseline --no-regimes > $OUT/trace_group.log 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
