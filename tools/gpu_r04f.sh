#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04f
mkdir -p $OUT
cd $REPO; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_00_configs_gpu.py tests/test_full_size_gpu.py tests/test_parity_gpu.py -m gpu -x -q -n 4 -k "sort or Sort" ) > $OUT/sort_tests.log 2>&1; tail -3 $OUT/sort_tests.log
cd /tmp
for d in 0 1 4; do
  SSGPU_ONESWEEP_DEBUG=$d timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d$d -o t -- python $REPO/tools/sort_only.py > $OUT/d$d.log 2>&1
done
python3 - <<'PY'
import csv, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r04f"
for d in (0, 1, 4):
    for r in csv.DictReader(open(f"{out}/d{d}/t_kernel_stats.csv")):
        if "ssgpu_sort" in r["Name"]:
            print("dbg=%d %-45s calls %s avg %.3f ms" % (d, r["Name"].split("(")[0][:45], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete; find $OUT -name "*trace.csv" -delete
