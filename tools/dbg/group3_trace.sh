#!/bin/bash
# Development: per-launch durations of the group stage's kernels of bench.py --query group3 under context options.
# usage: OPTS="k=v,k=v" tools/dbg/group3_trace.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/group3_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o s -- python $REPO/bench.py --query group3 --no-cpu-baseline --steps 6 --warmup 3 --opts "$OPTS" > $OUT/log 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/t/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
d = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in rows if "pipeline_kernel" in r["Kernel_Name"] or "part_agg" in r["Kernel_Name"]]
print(" ".join("%s%.2f" % ("P" if "pipeline" in n else "A", t) for n, t in d[-14:]))
PY
grep "^{" $OUT/log | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('ms/step', j['ms_per_step'], 'tile', j['config']['tile_rows'], 'grid', j['config']['grid'])"
