#!/bin/bash
# FETCH_SIZE and the raw L2 memory-side counters on RANDOM reads of known byte counts (tools/microbench/pmc_calib.hip gather):
# settles what FETCH_SIZE means for the Sort's record gather (streaming reads are reported at one half -- is a
# random 64-byte read reported in full, or does the L2 fetch the whole 128-byte line?).
# Usage (GPU box): tools/pmc_calibrate_gather.sh  -> gpurun_out/pmc_gather/{counters.txt,summary.json}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_gather
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\|TCP_TCC_[A-Z0-9_]*\|FETCH_SIZE\|WRITE_SIZE" | sort -u > $OUT/counters.txt
i=0
for grp in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum" "TCC_BUBBLE_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o p -- $REPO/tools/microbench/_bin/pmc_calib gather > $OUT/g$i.log 2>&1 || echo "pass $i ($grp) failed" >> $OUT/failed.txt
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if k.startswith("calib_gather"):
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
known = {"calib_gather_rec64": (4 << 30, "64-byte records, 4 lanes each"), "calib_gather_rec64_lane": (4 << 30, "64-byte records, one lane each"),
         "calib_gather_rec128": (4 << 30, "128-byte lines, 8 lanes each"), "calib_gather_u64": (1 << 30, "8-byte reads")}
res = {}
for k, cs in acc.items():
    vals = {c: v for c, v in cs.items()}
    # calib_gather_rec64 is launched twice per repetition (4 GiB window, then 4 MiB window): launches alternate
    entry = {"true_bytes_per_launch": known.get(k, (None, ""))[0], "pattern": known.get(k, (None, ""))[1], "counters": {}}
    for c, v in vals.items():
        if k == "calib_gather_rec64":
            entry["counters"][c] = {"window_4GiB": sum(v[0::2]) / max(len(v[0::2]), 1), "window_4MiB": sum(v[1::2]) / max(len(v[1::2]), 1)}
        else:
            entry["counters"][c] = sum(v) / len(v)
    res[k] = entry
json.dump(res, open(out + "/summary.json", "w"), indent=1, sort_keys=True)
for k in sorted(res):
    print(k, json.dumps(res[k]["counters"], sort_keys=True))
PY
grep -h "gather_" $OUT/g1.log | head -20 > $OUT/durations.txt
