#!/bin/bash
# Runs ON THE GPU BOX: SQ counters over the rANS stage of N 1080p key frames (tools/exp_rans.py's encode), both forms:
# instructions per coder entry by unit, wave cycles, waits.  Usage: tools/rans_pmc.sh [frames] [outdir]
set -e
R=$PWD
N=${1:-16}
OUT=$R/gpurun_out/${2:-ranspmc}
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
SETS=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS"
      "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INST_LEVEL_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU")
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/set$i -o run --output-format csv -- python3 $R/tools/exp_rans.py $N > $OUT/set$i.log 2>&1
done
SCPR_RANS_JSON=$OUT/entries.json python3 $R/tools/exp_rans.py $N > $OUT/plain.log 2>&1
python3 - <<PY
import csv, glob, collections, json
meta = json.load(open("$OUT/entries.json"))
ent = float(meta["entries_per_encode"]) * meta["encodes_per_form"]
tot = {"k_rans_s": collections.Counter(), "k_rans": collections.Counter()}
disp = {"k_rans_s": set(), "k_rans": set()}
for f in glob.glob("$OUT/*/run_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "k_rans_s" if "k_rans_s" in n else "k_rans" if "scpr::k_rans(" in n else None
        if k:
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add((f, r["Dispatch_Id"]))
res = {}
for k in tot:
    nd = {}
    for f, d in disp[k]:
        nd[f] = nd.get(f, 0) + 1
    launches = max(nd.values()) if nd else 0
    per = {c: v / ent for c, v in sorted(tot[k].items())}
    res[k] = {"launches_per_pass": launches, "totals": dict(tot[k]), "per_coder_entry": per}
    print(k, launches, {c: round(v, 3) for c, v in per.items()})
json.dump({"command": "tools/exp_rans.py $N (rANS stage of $N 1080p key frames, 3 encodes per form)", "stream": meta,
           "note": "SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_ACTIVE_* count in units of 4 cycles, summed over the kernel's waves: k_rans_s has one wave per block (wave cycles x 4 / entries = cycles per entry of a block's chain), k_rans four waves per 64 blocks (coder, two feeders, writer)", "kernels": res}, open("$OUT/summary.json", "w"), indent=1)
PY
