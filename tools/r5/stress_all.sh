#!/bin/bash
# Runs ON THE GPU BOX: the long-running stress tools on the round's final code, each writing its progress to a file under gpurun_out/
R=$PWD
OUT=$R/gpurun_out/r5/stress
mkdir -p $OUT
run() { name=$1; shift; echo "== $name"; timeout -k 10 420 "$@" > $OUT/$name.txt 2>&1; rc=$?; grep -v amdgpu.ids $OUT/$name.txt | tail -2; echo "rc=$rc"; }
BIG=1 run random_big python3 tools/stress_random.py 16 500
run formats python3 tools/stress_formats.py 24
run geometry python3 tools/stress_geometry.py
run geometry_random python3 tools/stress_geometry.py random 40 11
run ip python3 tools/stress_ip.py
run threads python3 tools/stress_threads.py 60
run corrupt python3 tools/stress_corrupt.py 160
