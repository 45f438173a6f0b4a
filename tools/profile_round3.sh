#!/bin/bash
# Round 3 evidence, collected on the GPU box in one go: for every bench query its JSON line (the headline with the CPU
# baseline), rocprofv3 --kernel-trace --stats of the same command, and the FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs).
# Output: gpurun_out/prof_r03/<query>/{line.json, kernel_stats.csv, counters.json}; tools/profiles_from_run.py turns it into profiles/r03_*.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r03
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for q in wide group3 group sort filter_mat; do
  mkdir -p $OUT/$q
  extra="--no-cpu-baseline"; [ $q = wide ] && extra=""
  python $REPO/bench.py --query $q $extra > $OUT/$q/line.json 2> $OUT/$q/line.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$q/stats -o k -- python $REPO/bench.py --query $q --steps 50 --warmup 5 --no-cpu-baseline > $OUT/$q/stats.log 2>&1
  f=$(find $OUT/$q/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/$q/kernel_stats.csv
  rm -rf $OUT/$q/stats
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/$q/pmc_$c -o p -- python $REPO/bench.py --query $q --steps 5 --warmup 2 --no-cpu-baseline > $OUT/$q/pmc_$c.log 2>&1
    f=$(find $OUT/$q/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/$q/$c.csv
    rm -rf $OUT/$q/pmc_$c
  done
  tail -c 400 $OUT/$q/line.json; echo
done
# interpreted vs specialised, every query
for q in wide group3 group filter_mat; do
  python $REPO/bench.py --query $q --no-specialize --no-cpu-baseline --steps 50 --warmup 5 > $OUT/$q/line_interpreted.json 2> /dev/null
done
$REPO/tools/pmc_calibrate.sh > $OUT/pmc_calibration.txt 2>&1; cp $REPO/gpurun_out/pmc_calib/factors.json $OUT/pmc_calibration.json 2>/dev/null
ls -R $OUT | head -60
