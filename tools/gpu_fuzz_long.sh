#!/bin/bash
# longer hunts than the suite's: the main generator through pytest (SS_FUZZ_SEEDS), single generators through tools/fuzz_hunt.py
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/fuzz
python tools/fuzz_hunt.py sort_plan 30000 600 70001 > gpurun_out/fuzz/sort_0.log 2>&1 &
python tools/fuzz_hunt.py sort_plan 31000 300 300007 > gpurun_out/fuzz/sort_1.log 2>&1 &
python tools/fuzz_hunt.py join_plan 32000 2500 1537 > gpurun_out/fuzz/join_0.log 2>&1 &
python tools/fuzz_hunt.py join_plan 35000 600 70001 > gpurun_out/fuzz/join_1.log 2>&1 &
python tools/fuzz_hunt.py plan 40000 800 150001 > gpurun_out/fuzz/plan_big.log 2>&1 &
python tools/fuzz_hunt.py ordered_aggregate_plan 45000 500 150001 > gpurun_out/fuzz/ordered_big.log 2>&1 &
wait
tail -q -n 2 gpurun_out/fuzz/sort_*.log gpurun_out/fuzz/join_*.log gpurun_out/fuzz/plan_big.log gpurun_out/fuzz/ordered_big.log | grep -v "^$\|amdgpu.ids" | cut -c1-250
