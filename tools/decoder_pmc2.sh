#!/bin/bash
# Runs ON THE GPU BOX: a wider set of SQ counters over the decoder kernel (rocprofv3 --pmc over tools/decode_only.py):
# where a lone decoder wave's cycles go - issue, waits, LDS latency in place (SQ_INST_LEVEL_LDS / SQ_INSTS_LDS),
# instruction fetch (SQ_IFETCH_LEVEL / SQ_IFETCH).  Usage: tools/decoder_pmc2.sh [frames] [outdir]
set -e
R=$PWD
N=${1:-16}
OUT=$R/gpurun_out/${2:-decpmc2}
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
SETS=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS SQ_INSTS_LDS SQ_INST_LEVEL_LDS"
      "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SALU"
      "SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/set$i -o run --output-format csv -- python3 $R/tools/decode_only.py $N > $OUT/set$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
tot = collections.Counter()
for f in glob.glob("$OUT/*/run_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_decode_gop_w" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
sym = $N * 702367.0  # symbols of the frames (about: frame 3 of the sequence x N)
res = {k: tot[k] / sym for k in sorted(tot)}
for k, v in res.items():
    print("%-24s %16.0f  %9.3f per symbol" % (k, tot[k], v))
json.dump({"command": "tools/decode_only.py $N", "symbols": sym, "per_symbol": res}, open("$OUT/summary.json", "w"), indent=1)
PY
