#!/bin/bash
# round 6, call 19: the whole GPU suite with the full shipped-configuration corpus (SS_FUZZ_SHIPPED_FULL=1) on the final tree
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time SS_FUZZ_SHIPPED_FULL=1 timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06_suite3.log 2>&1
tail -6 gpurun_out/r06_suite3.log
cp gpurun_out/fuzz_shipped.json gpurun_out/r06_fuzz_shipped_full_final.json
