#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the profiles behind DESIGN.md / bench.py's roofline block.
#   1. rocprofv3 --kernel-trace --stats over the headline bench command (configs[1]) and over the I+P (configs[2]) and
#      4K (configs[3] share) workloads
#   2. HBM traffic of the headline: --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section)
#   3. the decoder's instruction mix and wait counters: tools/decoder_pmc2.sh
# Output: gpurun_out/prof/<tag>/...; tools/summarize_profiles.py turns it into profiles/<tag>_*.
set -e
TAG=${1:-r2a}
R=$PWD
OUT=$R/gpurun_out/prof/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
# what these profiles are profiles OF: the digest of the kernel sources on this box, and when (bench.py quotes the HBM traffic only
# while its own sources still have this digest)
python3 -c "import sys, datetime; sys.path.insert(0, '$R'); import bench; print(bench.csrc_digest()); print(datetime.datetime.utcnow().strftime('%Y-%m-%dT%H:%MZ'))" > $OUT/csrc_sha256.txt
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 $R/bench.py --no-others --no-host-boundary --steps 3 --warmup 1 --cpu-frames 24 > $OUT/bench_stats.log 2>&1
echo stats done
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/stats_ip -o run --output-format csv -- python3 $R/bench.py --workload ip --gop 50 --no-cpu --no-host-boundary --steps 2 --warmup 1 > $OUT/bench_stats_ip.log 2>&1
echo ip stats done
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/stats_4k -o run --output-format csv -- python3 $R/bench.py --width 3840 --height 2160 --frames 150 --no-cpu --no-host-boundary --steps 2 --warmup 1 > $OUT/bench_stats_4k.log 2>&1
echo 4k stats done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$c -o run --output-format csv -- python3 $R/bench.py --no-others --no-host-boundary --steps 1 --warmup 0 --no-cpu > $OUT/bench_$c.log 2>&1
  echo $c done
done
cd $R
tools/decoder_pmc2.sh 16 prof/$TAG/dec > $OUT/decoder_pmc.log 2>&1
echo decoder pmc done
# the same counters over ONE GOP of 40 frames, the chain's wave alone (tools/split_ip_pmc.py <tag> turns the two into P-frame figures)
tools/decoder_pmc2.sh 40 ${TAG}_decpmc_ip --ip > $OUT/decoder_pmc_ip.log 2>&1
echo decoder ip pmc done
tail -1 $OUT/bench_stats.log | head -c 400
