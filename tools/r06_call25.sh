#!/bin/bash
# round 6, call 25: parity tests on a LOADED GPU (a second process streams the headline query the whole time): ordering bugs between the
# library's streams show up as wrong results only when kernels start late
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 1500 python bench.py --query group3 --steps 400000 --warmup 5 --no-cpu-baseline --no-traffic --no-configs > /dev/null 2>&1 ) &
LOAD=$!
sleep 20
( time timeout 1400 python -m pytest tests/test_cpp_facade.py tests/test_seams_gpu.py tests/test_cursor_contract_gpu.py tests/test_parity_gpu.py tests/test_chunked_gpu.py tests/test_golden_gpu.py tests/test_file_format.py tests/test_best_effort_gpu.py -m gpu -q --timeout 600 ) > gpurun_out/r06_loaded_gpu_tests.log 2>&1
tail -15 gpurun_out/r06_loaded_gpu_tests.log
kill $LOAD 2>/dev/null
wait $LOAD 2>/dev/null
echo load stopped
