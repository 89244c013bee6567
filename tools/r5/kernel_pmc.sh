#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of one encoder kernel over the 300 x 1080p key-frame encode (tools/exp_enc.py keys1080), summed
# over the kernel's largest dispatch: instructions by unit, busy / wait cycles, LDS conflicts.  Usage: tools/r5/kernel_pmc.sh k_tiles [outdir]
set -e
R=$PWD
K=${1:-k_tiles}
OUT=$R/gpurun_out/${2:-kpmc_$K}
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
SETS=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS"
      "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
      "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_FLAT")
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/set$i -o run --output-format csv -- python3 $R/tools/exp_enc.py keys1080 > $OUT/set$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
best = {}
for f in glob.glob("$OUT/*/run_counter_collection.csv"):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "$K" in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    if per:
        d = max(per.values(), key=lambda c: c.get("SQ_WAVES", 0) or sum(c.values()))
        best.update(d)
for k, v in sorted(best.items()):
    print("%-24s %16.0f" % (k, v))
PY
