#!/bin/bash
# Round 4, GPU call B: suite after the sharded-step fusion / hand-off / extraction changes; window sweep of random record reads;
# the default bench line (with the other configs); the sharded steps at the 8-GPU shard size; the Sort with its new gather.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04b
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q -n 4 ) > $OUT/suite.log 2>&1
tail -5 $OUT/suite.log
tools/microbench/_bin/pmc_calib gather 2>&1 | grep "window_sweep\|4 GiB" | tail -16 > $OUT/window_sweep.txt; cat $OUT/window_sweep.txt
( time python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -3 $OUT/bench_default.err; cut -c1-1500 $OUT/bench_default.json
for q in wide group; do
  python bench.py --query $q --force-distributed --rows 12500000 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/dist1_${q}_12m5.json 2> $OUT/dist1_${q}_12m5.err
  python bench.py --query $q --rows 12500000 --steps 200 --warmup 20 --no-cpu-baseline --no-configs > $OUT/plain_${q}_12m5.json 2> $OUT/plain_${q}_12m5.err
  grep -ho '"ms_per_step": [0-9.]*' $OUT/dist1_${q}_12m5.json $OUT/plain_${q}_12m5.json | paste - -
done
python bench.py --query group --force-distributed --exchange all_gather --rows 12500000 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/dist1_group_12m5_all_gather.json 2> $OUT/dist1_group_ag.err
grep -ho '"ms_per_step": [0-9.]*' $OUT/dist1_group_12m5_all_gather.json
cd /tmp
for q in wide group; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$q -o t -- python $REPO/bench.py --query $q --force-distributed --rows 12500000 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/trace_$q.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sort_stats -o t -- python $REPO/bench.py --query sort --steps 20 --warmup 3 --no-cpu-baseline > $OUT/sort_line.json 2> $OUT/sort.err
grep -o '"kernel_ms": [0-9.]*' $OUT/sort_line.json; head -12 $OUT/sort_stats/t_kernel_stats.csv | cut -c1-160
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
du -sh $OUT
