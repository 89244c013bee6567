#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the profiles behind DESIGN.md §6 / bench.py's roofline block.
#   1. rocprofv3 --kernel-trace --stats over the default bench command
#   2. HBM traffic: --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section)
#   3. instruction mix of the decoder: SQ_INSTS_* / SQ_WAVE_CYCLES over tools/decode_only.py
# Output: gpurun_out/prof/<tag>/...; tools/summarize_profiles.py turns it into profiles/<tag>_*.
set -e
TAG=${1:-r1e}
R=$PWD
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 $R/bench.py --frames 300 --steps 3 --warmup 1 --cpu-frames 24 > $OUT/bench_stats.log 2>&1
echo stats done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$c -o run --output-format csv -- python3 $R/bench.py --frames 300 --steps 1 --warmup 0 --no-cpu > $OUT/bench_$c.log 2>&1
  echo $c done
done
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  n=$(echo $set | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/dec_$n -o run --output-format csv -- python3 $R/tools/decode_only.py 16 > $OUT/dec_$n.log 2>&1
  echo $n done
done
tail -1 $OUT/bench_stats.log | head -c 600
