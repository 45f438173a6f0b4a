#!/bin/bash
# round 6, call 23: chunked forms over join plans (aux input bound by run_host / stream), the facade's chunked test
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/fuzz
python tools/fuzz_chunked.py join_plan 32000 1500 1537 > gpurun_out/fuzz/chunked_join_0.log 2>&1 &
python tools/fuzz_chunked.py join_plan 35000 300 70001 > gpurun_out/fuzz/chunked_join_big.log 2>&1 &
python tools/fuzz_chunked.py plan 60000 1500 1537 > gpurun_out/fuzz/chunked_plan_2.log 2>&1 &
( time timeout 900 python -m pytest tests/test_cpp_facade.py tests/test_chunked_gpu.py -m gpu -x -q --timeout 300 ) > gpurun_out/r06_call23_tests.log 2>&1
tail -4 gpurun_out/r06_call23_tests.log
wait
tail -q -n 6 gpurun_out/fuzz/chunked_join_*.log gpurun_out/fuzz/chunked_plan_2.log | grep -v "^$\|amdgpu.ids" | cut -c1-400
