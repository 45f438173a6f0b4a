#!/bin/bash
# Round 6 evidence, collected on the GPU box in one go (adapted from tools/profile_round5.sh).
#  * for every bench query: its JSON line (the default line carries the CPU baselines and the other configs), rocprofv3
#    --kernel-trace --stats of the same command, FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs);
#  * the GroupAggregate lines with dense slots (the default) and without (group_dense=0), same box;
#  * the sharded steps on one rank at the 8-GPU shard size (12.5 M rows) and at 100 M rows -- dense exchange, key-range and
#    all-gather image exchanges, the plain plan -- lines + kernel statistics of the dense step.
# Output: gpurun_out/prof_r06/...; `SSGPU_PROFILE_SRC=prof_r06 SSGPU_PROFILE_TAG=r06 SSGPU_PROFILE_PMC_TAG=r06 python tools/profiles_from_run.py` -> profiles/r06_*.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r06
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for q in wide group3 group sort filter_mat; do
  mkdir -p $OUT/$q
  extra="--no-cpu-baseline"; [ $q = wide ] && extra=""
  python $REPO/bench.py --query $q $extra > $OUT/$q/line.json 2> $OUT/$q/line.err
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$q/stats -o k -- python $REPO/bench.py --query $q --steps 50 --warmup 5 --no-cpu-baseline --no-configs > $OUT/$q/stats.log 2>&1
  f=$(find $OUT/$q/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/$q/kernel_stats.csv
  rm -rf $OUT/$q/stats
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/$q/pmc_$c -o p -- python $REPO/bench.py --query $q --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $OUT/$q/pmc_$c.log 2>&1
    f=$(find $OUT/$q/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/$q/$c.csv
    rm -rf $OUT/$q/pmc_$c
  done
  tail -c 300 $OUT/$q/line.json; echo
done
python $REPO/bench.py --query wide --no-specialize --no-cpu-baseline --no-configs --steps 50 --warmup 5 > $OUT/wide/line_interpreted.json 2> /dev/null
for q in group3 group; do
  python $REPO/bench.py --query $q --no-cpu-baseline --no-configs --steps 50 --warmup 5 --opts group_dense=0 2> /dev/null | grep "^{" > $OUT/$q/line_hashed.json
done
mkdir -p $OUT/dist1
for rows in 12500000 100000000; do
  tag=$([ $rows = 12500000 ] && echo 12m5 || echo 100m)
  python $REPO/bench.py --query wide --force-distributed --rows $rows --no-cpu-baseline 2> /dev/null | grep "^{" > $OUT/dist1/wide_${tag}_rows_dist1.json
  python $REPO/bench.py --query wide --rows $rows --no-cpu-baseline --no-configs 2> /dev/null | grep "^{" > $OUT/dist1/wide_${tag}_rows_plain.json
  python $REPO/bench.py --query group --force-distributed --exchange dense --rows $rows --no-cpu-baseline 2> /dev/null | grep "^{" > $OUT/dist1/group_${tag}_rows_dist1_dense.json
  python $REPO/bench.py --query group --rows $rows --no-cpu-baseline 2> /dev/null | grep "^{" > $OUT/dist1/group_${tag}_rows_plain.json
done
for ex in key_range all_gather; do
  python $REPO/bench.py --query group --force-distributed --exchange $ex --rows 12500000 --warmup 60 --no-cpu-baseline --opts group_dense=0 2> /dev/null | grep "^{" > $OUT/dist1/group_12m5_rows_dist1_$ex.json
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dist1/stats_group -o k -- python $REPO/bench.py --query group --force-distributed --exchange dense --rows 12500000 --steps 100 --warmup 10 --no-cpu-baseline --no-regimes > $OUT/dist1/stats_group.log 2>&1
f=$(find $OUT/dist1/stats_group -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/dist1/group_12m5_rows_dist1_dense_kernel_stats.csv
rm -rf $OUT/dist1/stats_group
python $REPO/tools/first_run_bench.py > $OUT/first_run.json 2> $OUT/first_run.err
ls -R $OUT | head -60
