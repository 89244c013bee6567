#!/bin/bash
# Runs ON THE GPU BOX: a second, longer helping of the stress tools (other seeds), progress in files under gpurun_out/
R=$PWD
OUT=$R/gpurun_out/r5/stress2
mkdir -p $OUT
run() { name=$1; shift; echo "== $name"; timeout -k 10 500 "$@" > $OUT/$name.txt 2>&1; rc=$?; grep -v amdgpu.ids $OUT/$name.txt | tail -1; echo "rc=$rc"; }
BIG=1 run random_big python3 tools/stress_random.py 40 700
run formats python3 tools/stress_formats.py 80
run corrupt python3 tools/stress_corrupt.py 640
run threads python3 tools/stress_threads.py 150
