#!/bin/bash
# Runs ON THE GPU BOX: kernel times and HBM traffic of the headline command only (a short form of collect_profiles.sh)
set -e
TAG=${1:-x1}
R=$PWD
OUT=$R/gpurun_out/prof/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 $R/bench.py --no-others --steps 3 --warmup 1 --no-cpu > $OUT/bench_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$c -o run --output-format csv -- python3 $R/bench.py --no-others --steps 1 --warmup 0 --no-cpu > $OUT/bench_$c.log 2>&1
done
echo done
