#!/bin/bash
# Runs ON THE GPU BOX: exact instruction counts of the decoder kernel (rocprofv3 --pmc over tools/decode_only.py),
# the measure used to compare small changes of the decoder (time is layout-noisy, counts are not).
set -e
R=$PWD
OUT=$R/gpurun_out/decpmc
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
SETS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY")
if [ "$1" = quick ]; then SETS=("SQ_INSTS SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVE_CYCLES"); fi
for set in "${SETS[@]}"; do
  n=$(echo $set | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/$n -o run --output-format csv -- python3 $R/tools/decode_only.py 16 > $OUT/$n.log 2>&1
done
python3 - <<PY
import csv, glob, collections
tot = collections.Counter()
for f in glob.glob("$OUT/*/run_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_decode_gop_w" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
sym = 16 * 702367.0  # symbols of the 16 frames (about: frame 3 of the sequence x 16)
for k in sorted(tot):
    print("%-22s %14.0f  %8.2f per symbol" % (k, tot[k], tot[k] / sym))
PY
