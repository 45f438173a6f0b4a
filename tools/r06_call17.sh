#!/bin/bash
# round 6, call 17: the node-level expression seam's tests, then the round's evidence (tools/profile_round6.sh)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 600 python -m pytest tests/test_seams_gpu.py tests/test_chunked_gpu.py -m gpu -x -q --timeout 300 ) > gpurun_out/r06_call17_tests.log 2>&1
tail -8 gpurun_out/r06_call17_tests.log
( time bash tools/profile_round6.sh ) > gpurun_out/r06_profile_round6.log 2>&1
tail -30 gpurun_out/r06_profile_round6.log
