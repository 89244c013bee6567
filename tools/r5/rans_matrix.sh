#!/bin/bash
# Runs ON THE GPU BOX (round 5): the rANS stage of n 1080p key frames for each variant library
# (tools/build_variant.sh NAME -DSCPR_EXPERIMENT [-DSCPR_RANS_RING=16] [-DSCPR_RANS_GLC] [-DSCPR_RANS_NOFAST] [-DSCPR_RANS_NOCHECK]; the 8-byte-record
# form of the matrix in DESIGN.md 9 was -DSCPR_RANS_REC8 at commit 8539980, taken out of the source afterwards), with and without the LDS
# allocation that leaves one workgroup per CU.
R=$PWD
OUT=$R/gpurun_out/r5/rans_matrix
mkdir -p $OUT
for v in "$@"; do
  SCPR_AMD_LIB=$R/screenpressor_amd/variants/libscpr_$v.so timeout -k 10 280 python3 $R/tools/exp_rans.py 64 128 170 300 500 > $OUT/$v.txt 2>&1 || echo "FAILED $v"
  SCPR_RANS_LDS=90000 SCPR_AMD_LIB=$R/screenpressor_amd/variants/libscpr_$v.so timeout -k 10 200 python3 $R/tools/exp_rans.py 128 170 > $OUT/${v}_lds.txt 2>&1 || echo "FAILED $v lds"
  echo "== $v"; cat $OUT/$v.txt; echo "== $v, one workgroup per CU"; cat $OUT/${v}_lds.txt
done
