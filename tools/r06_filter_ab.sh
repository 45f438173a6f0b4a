#!/bin/bash
# Round 6, review item 3: where does the materialising Filter's 1.93 <-> 2.48 ms swing come from?  One box, one process per point:
#   tile_map 0 / 1 / 2 (vm.h VM_FLAG_XCD_CHUNKS), output blocks fresh vs from the pool of a destroyed plan, input column
#   bases 2 MiB-aligned in one arena vs staggered by 4 KiB + 256 B per column, and each point three times (spread within a box).
# Then the SAME process under rocprofv3 --kernel-trace --stats: its bench line and the trace's kernel averages side by side.
cd "$(dirname "$0")/.."
out=gpurun_out/r06_filter_ab.txt
: > $out
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.4f kernel_ms %.4f frac %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))'
run() {   # label, env, args...
  label=$1; shift; envs=$1; shift
  for rep in 1 2 3; do
    line=$(env $envs timeout 300 python bench.py --query filter_mat --steps 40 --warmup 10 --no-cpu-baseline --no-traffic "$@" 2>/dev/null | tail -1)
    echo "$label rep $rep: $(echo "$line" | python -c "$pick")" >> $out
  done
}
run "tile_map=0" "X=1" --opts tile_map=0
run "tile_map=1" "X=1" --opts tile_map=1
run "tile_map=2" "X=1" --opts tile_map=2
run "tile_map=0 pool off" "SSGPU_POOL_MB=0" --opts tile_map=0
run "tile_map=1 arena aligned" "X=1" --opts tile_map=1 --stagger 0
run "tile_map=1 arena staggered 4352" "X=1" --opts tile_map=1 --stagger 4352
run "tile_map=0 arena staggered 4352" "X=1" --opts tile_map=0 --stagger 4352
# outputs from pooled blocks: the headline run's extra configs destroy three plans before filter_mat's is made
line=$(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1)
echo "driver-style line, configs.filter_mat: $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d["configs"]["filter_mat"]; print("ms_per_step %.4f kernel_ms %.4f frac %.3f" % (f["ms_per_step"], f["kernel_ms"], f["frac"]))')" >> $out
# the same process: bench line + kernel trace
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/r06_filter_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_filter_prof -o p -- python $R/bench.py --query filter_mat --steps 40 --warmup 10 --no-cpu-baseline --no-traffic > $R/gpurun_out/r06_filter_prof_line.json 2> /dev/null
echo "same process under rocprofv3: $(tail -1 $R/gpurun_out/r06_filter_prof_line.json | python -c "$pick")" >> $R/$out
f=$(find $R/gpurun_out/r06_filter_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && { head -1 $f; grep ssgpu $f; } > $R/gpurun_out/r06_filter_mat_kernel_stats.csv
cat $R/$out
