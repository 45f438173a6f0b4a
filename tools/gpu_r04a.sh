#!/bin/bash
# Round 4, GPU call A: full GPU suite, random-gather PMC calibration, timelines of the sharded steps at the 8-GPU shard size,
# same-box A/B of the round-2 library against this one on the headline, raw L2 counters of the Sort's kernels.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04a
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q -n 4 ) > $OUT/suite.log 2>&1
tail -5 $OUT/suite.log
# A/B on the same box, alternating (each line: kernel_ms of the headline)
for i in 1 2; do
  (cd tools/ab/_r02 && python bench.py --steps 100 --warmup 10 --no-cpu-baseline) > $OUT/ab_r02_$i.json 2> $OUT/ab_r02_$i.err
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/ab_now_$i.json 2> $OUT/ab_now_$i.err
  SSGPU_RTC_FLAGS="-DSSGPU_AB_DUMMY=$i" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-specialize > $OUT/ab_now_interp_$i.json 2> $OUT/ab_now_interp_$i.err
done
grep -ho '"kernel_ms": [0-9.]*' $OUT/ab_*.json | paste - - - - - - 
bash tools/pmc_calibrate_gather.sh > $OUT/gather_calib.log 2>&1
cp -r gpurun_out/pmc_gather $OUT/ 2>/dev/null
tail -8 $OUT/gather_calib.log
# timelines (kernel trace with start / end timestamps) of the sharded steps at 12.5 M rows on one rank
cd /tmp
for q in wide group; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$q -o t -- python $REPO/bench.py --query $q --force-distributed --rows 12500000 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/trace_$q.log 2>&1
  tail -1 $OUT/trace_$q.log | cut -c1-400
done
cd $REPO
bash tools/pmc_raw.sh sort "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum;TCC_HIT_sum TCC_MISS_sum;FETCH_SIZE;WRITE_SIZE" --query sort > $OUT/pmc_sort.log 2>&1
cp -r gpurun_out/pmcraw_sort $OUT/ 2>/dev/null
tail -12 $OUT/pmc_sort.log
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
du -sh $OUT
