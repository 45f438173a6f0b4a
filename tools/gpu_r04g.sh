#!/bin/bash
# the driver's own commands: the GPU suite in ONE process, smoke(), the default bench line
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04g
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/suite_single_process.log 2>&1
tail -4 $OUT/suite_single_process.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
( time python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -4 $OUT/bench_default.err; cut -c1-300 $OUT/bench_default.json
