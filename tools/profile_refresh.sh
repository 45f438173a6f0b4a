#!/bin/bash
# Round 3, second collection (after the sort / group changes late in the round): bench lines and rocprofv3 kernel statistics of
# every bench query -- no counter passes (the committed r03_pmc_*.json stand: the kernels' traffic did not change) -- plus the
# 1000-group query of tools/perf_sweep.py (the resident form).  Output as tools/profile_round3.sh: gpurun_out/prof_r03/<query>/.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r03
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for q in wide group3 group sort filter_mat; do
  mkdir -p $OUT/$q
  extra="--no-cpu-baseline"; [ $q = wide ] && extra=""
  timeout 150 python $REPO/bench.py --query $q $extra > $OUT/$q/line.json 2> $OUT/$q/line.err
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$q/stats -o k -- python $REPO/bench.py --query $q --steps 50 --warmup 5 --no-cpu-baseline > $OUT/$q/stats.log 2>&1
  f=$(find $OUT/$q/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/$q/kernel_stats.csv
  rm -rf $OUT/$q/stats
  tail -c 300 $OUT/$q/line.json; echo
done
mkdir -p $OUT/group_small
for sp in 1 0; do
  timeout 100 python $REPO/tools/perf_sweep.py --queries group_small --tiles 0 --reps 12 --opts specialize=$sp 2>&1 | grep "group_small" > $OUT/group_small/sweep_specialize$sp.txt
done
timeout 100 python $REPO/tools/perf_sweep.py --queries group_small --tiles 0 --reps 12 --opts specialize=1,group_resident=0 2>&1 | grep "group_small" > $OUT/group_small/sweep_records_through_memory.txt
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/group_small/stats -o k -- python $REPO/tools/perf_sweep.py --queries group_small --tiles 0 --reps 30 --opts specialize=1 > $OUT/group_small/stats.log 2>&1
f=$(find $OUT/group_small/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/group_small/kernel_stats.csv
rm -rf $OUT/group_small/stats
cat $OUT/group_small/*.txt
