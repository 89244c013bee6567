#!/bin/bash
# Runs ON THE GPU BOX: where a decoder wave's cycles go, beyond instruction counts - instruction cache, LDS latency and
# conflicts, active cycles per unit.  Output: gpurun_out/decpmc3/summary.txt
set -e
R=$PWD
OUT=$R/gpurun_out/decpmc3
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
SETS=("SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE" "SQ_INSTS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_BRANCH")
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/s$i -o run --output-format csv -- python3 $R/tools/decode_only.py ${1:-16} ${2:-} > $OUT/s$i.log 2>&1 || { tail -5 $OUT/s$i.log; echo "set $i failed: $set"; }
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
tot = collections.Counter()
for f in glob.glob("$OUT/*/run_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_decode_gop_w" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
sym = ${1:-16} * 702367.0
for k in sorted(tot):
    print("%-24s %16.0f  %9.3f per symbol" % (k, tot[k], tot[k] / sym))
PY
cat $OUT/summary.txt
