#!/bin/bash
# Runs ON THE GPU BOX: a wider set of SQ counters over the decoder kernel (rocprofv3 --pmc over tools/decode_only.py):
# where a lone decoder wave's cycles go - issue, waits, LDS latency in place (SQ_INST_LEVEL_LDS / SQ_INSTS_LDS),
# instruction fetch (SQ_IFETCH_LEVEL / SQ_IFETCH).  Usage: tools/decoder_pmc2.sh [frames] [outdir] [--ip]
# --ip: ONE GOP (a key frame and frames - 1 P-frames), k_decode_gop_w<true>, run with SCPR_NO_HELPERS=1 so that the counters see
# the chain's wave only (helper waves poll LDS words: their instructions would drown the chain's); the per-symbol figures of the
# P-frames are what is left after the key frame's share (a key-frame symbol's cost from the run without --ip).
set -e
R=$PWD
N=${1:-16}
OUT=$R/gpurun_out/${2:-decpmc2}
MODE=${3:-}
if [ "$MODE" = "--ip" ]; then export SCPR_NO_HELPERS=1; fi
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
SETS=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS SQ_INSTS_LDS SQ_INST_LEVEL_LDS"
      "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SALU"
      "SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/set$i -o run --output-format csv -- python3 $R/tools/decode_only.py $N $MODE > $OUT/set$i.log 2>&1
done
SCPR_SYMS_OUT=$OUT/symbols.json python3 $R/tools/decode_only.py $N $MODE > $OUT/symbols.log 2>&1
python3 - <<PY
import csv, glob, collections, json
tot = collections.Counter()
for f in glob.glob("$OUT/*/run_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_decode_gop_w" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
meta = json.load(open("$OUT/symbols.json"))
sym = float(meta["symbols"])  # coder entries of the stream (scpr_debug_entries of the encode)
res = {k: tot[k] / sym for k in sorted(tot)}
for k, v in res.items():
    print("%-24s %16.0f  %9.3f per symbol" % (k, tot[k], v))
json.dump({"command": "tools/decode_only.py $N $MODE", "symbols": sym, "stream": meta, "totals": dict(tot), "per_symbol": res}, open("$OUT/summary.json", "w"), indent=1)
PY
