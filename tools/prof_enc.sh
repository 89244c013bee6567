#!/bin/bash
# Runs ON THE GPU BOX: per-kernel times of the encoder on one workload of tools/exp_enc.py (rocprofv3 --kernel-trace --stats).
#   tools/prof_enc.sh gop1080 [tag]
set -e
WL=${1:-gop1080}
TAG=${2:-enc_$WL}
R=$PWD
OUT=$R/gpurun_out/prof/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 $R/tools/exp_enc.py $WL > $OUT/run.log 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:22]:
    print("%-70s calls %5s  total %10.3f ms  avg %10.3f ms  %5s%%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"]))
PY
tail -2 $OUT/run.log
