#!/bin/bash
# onesweep attribution: the sort's pass kernel with parts switched off (results wrong under a switch: bench's own check fails -> use a tiny driver)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04e
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for d in 0 1 2 3 4 8 15; do
  SSGPU_ONESWEEP_DEBUG=$d timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d$d -o t -- python $REPO/tools/sort_only.py > $OUT/d$d.log 2>&1
  echo "dbg=$d $(grep onesweep $OUT/d$d/t_kernel_stats.csv | cut -d, -f2-4 | head -1)"
done
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete; find $OUT -name "*trace.csv" -delete
