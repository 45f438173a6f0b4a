#!/bin/bash
# one GPU call: suite (without the shipped-configuration fuzz), default bench line, filter A/B, fuzz-worker throughput
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q --deselect tests/test_fuzz_shipped_gpu.py ) > gpurun_out/r06_suite.log 2>&1
tail -3 gpurun_out/r06_suite.log
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
tail -c 600 gpurun_out/r06_bench_default.err
bash tools/r06_filter_ab.sh > /dev/null 2>&1
cat gpurun_out/r06_filter_ab.txt
for w in 12 96; do
  rm -rf ~/.cache/ssgpu
  ( time SS_FUZZ_SHIPPED_SCALE=0.05 SS_FUZZ_WORKERS=$w python -m pytest tests/test_fuzz_shipped_gpu.py -q -m gpu ) > gpurun_out/r06_fuzz_workers_$w.log 2>&1
  echo "workers $w: $(grep real gpurun_out/r06_fuzz_workers_$w.log) $(tail -4 gpurun_out/r06_fuzz_workers_$w.log | head -1)"
done
