#!/bin/bash
# round 6, call 22: fuzz hunt over the chunked forms
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/fuzz
python tools/fuzz_chunked.py plan 0 1500 1537 > gpurun_out/fuzz/chunked_plan_0.log 2>&1 &
python tools/fuzz_chunked.py plan 1500 1500 1537 > gpurun_out/fuzz/chunked_plan_1.log 2>&1 &
python tools/fuzz_chunked.py plan 50000 300 70001 > gpurun_out/fuzz/chunked_plan_big.log 2>&1 &
python tools/fuzz_chunked.py ordered_aggregate_plan 4000 500 1537 > gpurun_out/fuzz/chunked_ordered.log 2>&1 &
python tools/fuzz_chunked.py plain_group 0 1500 1537 > gpurun_out/fuzz/chunked_plain_0.log 2>&1 &
python tools/fuzz_chunked.py plain_group 3000 300 70001 > gpurun_out/fuzz/chunked_plain_big.log 2>&1 &
wait
tail -q -n 12 gpurun_out/fuzz/chunked_*.log | grep -v "^$\|amdgpu.ids" | cut -c1-400
