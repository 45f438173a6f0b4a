#!/bin/bash
# The parity suites on a LOADED GPU: a second process streams a bench query the whole time, so kernels start late and ordering bugs between
# the library's streams (copy / compute / side) show up as wrong rows instead of staying latent (round 6 found one this way:
# ssgpu_block_upload vs raw-pointer runs).  Usage (on the GPU box): bash tools/loaded_gpu_tests.sh [query of the load] [pytest args...]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
q=${1:-group3}; shift
( timeout 1500 python bench.py --query $q --steps 2000000 --warmup 5 --no-cpu-baseline --no-traffic --no-configs > /dev/null 2>&1 ) &
LOAD=$!
sleep 20
( time timeout 1400 python -m pytest ${@:-tests} -m gpu -q --timeout 900 ) > gpurun_out/loaded_gpu_tests.log 2>&1
tail -8 gpurun_out/loaded_gpu_tests.log
kill $LOAD 2>/dev/null
wait $LOAD 2>/dev/null
