#!/bin/bash
# longer hunts than the suite's: the main generator through pytest (SS_FUZZ_SEEDS), the round-4 generators through tools/fuzz_hunt.py
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/fuzz
for i in 0 1 2; do
  python tools/fuzz_hunt.py ordered_aggregate_plan $((10000 + i * 2500)) 2500 $((1537 + i * 7001)) > gpurun_out/fuzz/ordered_$i.log 2>&1 &
  python tools/fuzz_hunt.py sequential_sum_plan $((20000 + i * 2500)) 2500 $((1537 + i * 7001)) > gpurun_out/fuzz/seq_$i.log 2>&1 &
done
wait
tail -q -n 3 gpurun_out/fuzz/ordered_*.log gpurun_out/fuzz/seq_*.log | grep -v "^$" | cut -c1-250
