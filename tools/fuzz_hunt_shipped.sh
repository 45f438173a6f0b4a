#!/bin/bash
# A hunt with NEW seeds on the shipped configuration (hiprtc-specialised kernels + dense slots): tools/fuzz_hunt.py under
# SSGPU_SPECIALIZE=1 SSGPU_GROUP_DENSE=1, one process per core the cgroup schedules.  Usage: bash tools/fuzz_hunt_shipped.sh [first seed] [plans per process]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/fuzz
first=${1:-200000}; per=${2:-220}
export SSGPU_SPECIALIZE=1 SSGPU_GROUP_DENSE=1
i=0
for gen in plan plan plan plan plan plan ordered_aggregate_plan ordered_aggregate_plan join_plan join_plan sort_plan sort_plan sequential_sum_plan distinct_limit_plan; do
  python tools/fuzz_hunt.py $gen $((first + i * 1000)) $per 1537 > gpurun_out/fuzz/shipped_hunt_$i.log 2>&1 &
  i=$((i + 1))
done
python tools/fuzz_hunt.py plan $((first + 50000)) 60 70001 > gpurun_out/fuzz/shipped_hunt_big.log 2>&1 &
wait
tail -q -n 3 gpurun_out/fuzz/shipped_hunt_*.log | grep -v "^$\|amdgpu.ids" | cut -c1-300
