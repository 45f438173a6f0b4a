#!/bin/bash
# Collects the round's judged evidence on the GPU box: bench line, rocprofv3 kernel stats and the
# two PMC passes (FETCH_SIZE / WRITE_SIZE) of the same command.  Outputs under gpurun_out/prof_rNN/.
R=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$R
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py > $OUT/bench_line.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $R -- python $REPO/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
done
python - "$OUT" "$R" <<'PY'
import csv, glob, json, sys
out, r = sys.argv[1], sys.argv[2]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % c, recursive=True)
    acc = []
    for row in csv.DictReader(open(f[0])):
        if "pipeline_kernel" in row["Kernel_Name"] and row["Counter_Name"] == c:
            acc.append(float(row["Counter_Value"]))
    vals[c] = (sum(acc) / len(acc), len(acc))
line = json.loads(open(out + "/bench_line.json").read().strip().splitlines()[-1])
alg = line["roofline"]["algorithmic_bytes_per_row"] * line["config"]["rows_per_gpu"]
j = {"round": r, "kernel": "ssgpu_pipeline_kernel<1, false>", "command": "python bench.py --steps 5 --warmup 1 --no-cpu-baseline",
     "FETCH_SIZE_KiB_per_launch": vals["FETCH_SIZE"][0], "WRITE_SIZE_KiB_per_launch": vals["WRITE_SIZE"][0],
     "launches_sampled": vals["FETCH_SIZE"][1],
     "correction": "FETCH_SIZE x2 on gfx950 for 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM)",
     "traffic_bytes_per_launch": vals["FETCH_SIZE"][0] * 1024 * 2 + vals["WRITE_SIZE"][0] * 1024,
     "algorithmic_bytes_per_launch": int(alg)}
json.dump(j, open(out + "/pmc.json", "w"), indent=1)
print(json.dumps(j))
PY
cat $OUT/bench_line.json | tail -1 | cut -c1-900
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -5 "$f" | cut -c1-200 < /dev/null
