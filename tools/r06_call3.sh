#!/bin/bash
# GPU call 3: cgroup quota of the box; headline A/B round-5 library vs this tree (same box); column-base stagger sweep for the headline
# and the materialising Filter (inputs in one arena, column i at i x (2 MiB-rounded size + stagger)); output stagger; BestEffort tests
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r06_stagger_sweep.txt
: > $out
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) ; cfs: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) / $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null) ; nproc $(nproc)" >> $out
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.4f kernel_ms %.4f frac %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))'
b() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-traffic --no-configs "$@" 2>/dev/null | tail -1 | python -c "$pick"; }
( time python -m pytest tests/test_best_effort_gpu.py tests/test_golden_gpu.py tests/test_cpp_facade.py tests/test_seams_gpu.py -m gpu -x -q ) > gpurun_out/r06_best_effort.log 2>&1
tail -4 gpurun_out/r06_best_effort.log
for rep in 1 2; do
  echo "headline r05 library rep $rep: $(SSGPU_LIB=$PWD/tools/ab/_r05/libssgpu.so b)" >> $out
  echo "headline this tree   rep $rep: $(b)" >> $out
done
for st in 0 256 512 1024 2048 4096 4352 8448 16640 65792 1048832; do
  echo "headline stagger $st: $(b --stagger $st)" >> $out
done
for st in 0 256 1024 4096 4352 8448 65792; do
  echo "filter_mat in-stagger $st: $(b --query filter_mat --stagger $st)" >> $out
done
for os in 256 4352 8448 65792; do
  echo "filter_mat in-stagger 0 out_stagger $os: $(b --query filter_mat --stagger 0 --opts out_stagger=$os)" >> $out
  echo "filter_mat in-stagger 4352 out_stagger $os: $(b --query filter_mat --stagger 4352 --opts out_stagger=$os)" >> $out
done
echo "filter_mat torch-allocated out_stagger 4352: $(b --query filter_mat --opts out_stagger=4352)" >> $out
for q in group3 group sort; do
  echo "$q stagger -1: $(b --query $q)" >> $out
  echo "$q stagger 4352: $(b --query $q --stagger 4352)" >> $out
done
cat $out
