#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 300 python -m pytest tests/test_seams_gpu.py -m gpu -x -q --timeout 120 ) > gpurun_out/r06_call18_tests.log 2>&1
tail -3 gpurun_out/r06_call18_tests.log
timeout 900 python tools/filter_placement.py 10 > gpurun_out/r06_filter_placement.txt 2>&1
cat gpurun_out/r06_filter_placement.txt
