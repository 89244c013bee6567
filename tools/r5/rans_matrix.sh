#!/bin/bash
# Runs ON THE GPU BOX (round 5): the rANS stage of n 1080p key frames for each variant library
# (tools/build_variant.sh: ring 16 / 4, 8-byte records, GLC loads), with and without the LDS allocation that leaves one workgroup per CU.
R=$PWD
OUT=$R/gpurun_out/r5/rans_matrix
mkdir -p $OUT
for v in "$@"; do
  SCPR_AMD_LIB=$R/screenpressor_amd/variants/libscpr_$v.so timeout -k 10 280 python3 $R/tools/exp_rans.py 64 128 170 300 500 > $OUT/$v.txt 2>&1 || echo "FAILED $v"
  SCPR_RANS_LDS=90000 SCPR_AMD_LIB=$R/screenpressor_amd/variants/libscpr_$v.so timeout -k 10 200 python3 $R/tools/exp_rans.py 128 170 > $OUT/${v}_lds.txt 2>&1 || echo "FAILED $v lds"
  echo "== $v"; cat $OUT/$v.txt; echo "== $v, one workgroup per CU"; cat $OUT/${v}_lds.txt
done
