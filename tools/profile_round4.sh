#!/bin/bash
# Round 4 evidence, collected on the GPU box in one go.
#  * for every bench query: its JSON line (the default line carries the CPU baseline and the other configs), rocprofv3
#    --kernel-trace --stats of the same command, FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs), the interpreted line;
#  * the sharded steps on one rank at the 8-GPU shard size (12.5 M rows) and at 100 M rows: lines + kernel statistics;
#  * the skewed GroupAggregate, the random-gather calibration.
# Output: gpurun_out/prof_r04/...; `SSGPU_PROFILE_SRC=prof_r04 SSGPU_PROFILE_TAG=r04 SSGPU_PROFILE_PMC_TAG=r04 python tools/profiles_from_run.py` -> profiles/r04_*.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r04
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for q in wide group3 group sort filter_mat; do
  mkdir -p $OUT/$q
  extra="--no-cpu-baseline"; [ $q = wide ] && extra=""
  python $REPO/bench.py --query $q $extra > $OUT/$q/line.json 2> $OUT/$q/line.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$q/stats -o k -- python $REPO/bench.py --query $q --steps 50 --warmup 5 --no-cpu-baseline --no-configs > $OUT/$q/stats.log 2>&1
  f=$(find $OUT/$q/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/$q/kernel_stats.csv
  rm -rf $OUT/$q/stats
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/$q/pmc_$c -o p -- python $REPO/bench.py --query $q --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $OUT/$q/pmc_$c.log 2>&1
    f=$(find $OUT/$q/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/$q/$c.csv
    rm -rf $OUT/$q/pmc_$c
  done
  tail -c 300 $OUT/$q/line.json; echo
done
for q in wide group3 group filter_mat; do
  python $REPO/bench.py --query $q --no-specialize --no-cpu-baseline --no-configs --steps 50 --warmup 5 > $OUT/$q/line_interpreted.json 2> /dev/null
done
# the sharded steps on one rank (the N > 1 code path: partial run / shard plan -> RCCL collective to itself -> fold / merge)
mkdir -p $OUT/dist1
for rows in 12500000 100000000; do
  tag=$([ $rows = 12500000 ] && echo 12m5 || echo 100m)
  python $REPO/bench.py --query wide --force-distributed --rows $rows --no-cpu-baseline 2> /dev/null | grep "^{" > $OUT/dist1/wide_${tag}_rows_dist1.json
  python $REPO/bench.py --query wide --rows $rows --no-cpu-baseline --no-configs 2> /dev/null | grep "^{" > $OUT/dist1/wide_${tag}_rows_plain.json
  for ex in key_range all_gather; do
    python $REPO/bench.py --query group --force-distributed --exchange $ex --rows $rows --no-cpu-baseline 2> /dev/null | grep "^{" > $OUT/dist1/group_${tag}_rows_dist1_$ex.json
  done
  python $REPO/bench.py --query group --rows $rows --no-cpu-baseline 2> /dev/null | grep "^{" > $OUT/dist1/group_${tag}_rows_plain.json
done
for q in wide group; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dist1/stats_$q -o k -- python $REPO/bench.py --query $q --force-distributed --rows 12500000 --steps 100 --warmup 10 --no-cpu-baseline --no-regimes > $OUT/dist1/stats_$q.log 2>&1
  f=$(find $OUT/dist1/stats_$q -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/dist1/${q}_12m5_rows_dist1_kernel_stats.csv
  rm -rf $OUT/dist1/stats_$q
done
python $REPO/tools/skew_bench.py 100000000 1 2> /dev/null | tail -1 > $OUT/skew_specialized.json
python $REPO/tools/skew_bench.py 100000000 0 2> /dev/null | tail -1 > $OUT/skew_interpreted.json
$REPO/tools/pmc_calibrate.sh > $OUT/pmc_calibration.txt 2>&1; cp $REPO/gpurun_out/pmc_calib/factors.json $OUT/pmc_calibration.json 2>/dev/null
ls -R $OUT | head -80
