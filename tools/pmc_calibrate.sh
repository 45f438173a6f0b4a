#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/microbench/pmc_calib.hip): prints, per kernel, the bytes the
# kernel really moved divided by what the counter reports.  Usage (GPU box): tools/pmc_calibrate.sh  -> gpurun_out/pmc_calib/factors.json
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_calib
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o p -- $REPO/tools/microbench/_bin/pmc_calib > $OUT/$c.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
known = {"calib_read<unsigned int>": 1 << 30, "calib_read<unsigned long long>": 1 << 30, "calib_read<HIP_vector_type<unsigned int, 4u> >": 1 << 30,
         "calib_read_rec40": (1 << 30) // 40 * 40, "calib_write<unsigned int>": 1 << 30, "calib_write<unsigned long long>": 1 << 30,
         "calib_write<HIP_vector_type<unsigned int, 4u> >": 1 << 30}
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(out + "/%s/**/*counter_collection.csv" % c, recursive=True)
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] == c:
            acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        name = next((n for n in known if k.startswith("void " + n) or k.startswith(n)), None)
        if name is None or ("read" in name) != (c == "FETCH_SIZE"):
            continue
        kib = sum(v) / len(v)
        res["%s %s" % (c, name)] = {"counter_KiB_per_launch": kib, "true_bytes": known[name], "bytes_per_counter_KiB": known[name] / kib if kib else None,
                                    "factor_vs_KiB": known[name] / (kib * 1024) if kib else None}
json.dump(res, open(out + "/factors.json", "w"), indent=1, sort_keys=True)
for k in sorted(res):
    print("%-75s counter %12.1f KiB  true/counter = %.3f" % (k, res[k]["counter_KiB_per_launch"], res[k]["factor_vs_KiB"]))
PY
