#!/bin/bash
# Development: duration of the partition-scatter kernel of bench.py --query group3 under the write-pattern debug modes of
# PART_FLUSH (ctx option part_scatter_debug: 0 real, 1 sequential, k >= 2 runs of 2^k contiguous records).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/scatter_pattern; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in ${MODES:-0 1 2 3 4 5}; do
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/m$m -o s -- python $REPO/bench.py --query group3 --no-cpu-baseline --steps 8 --warmup 3 --opts part_scatter_debug=$m ${EXTRA} > $OUT/m$m.log 2>&1
  f=$(find $OUT/m$m -name "*kernel_stats.csv" | head -1)
  echo "mode $m: $(grep -h 'pipeline_kernel\|part_agg' $f | cut -d, -f1-4 | cut -c1-120 | tr '\n' ' ')"
done
