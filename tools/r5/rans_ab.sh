#!/bin/bash
# Runs ON THE GPU BOX: the rANS stage with the product library and with variant libraries (tools/build_variant.sh), same frame counts
R=$PWD
OUT=$R/gpurun_out/r5/rans_ab
mkdir -p $OUT
echo "== product"; timeout -k 10 280 python3 $R/tools/exp_rans.py 64 170 300 500 2>&1 | grep -v amdgpu.ids | tee $OUT/product.txt
for v in "$@"; do
  echo "== $v"; SCPR_AMD_LIB=$R/screenpressor_amd/variants/libscpr_$v.so timeout -k 10 280 python3 $R/tools/exp_rans.py 64 170 300 500 2>&1 | grep -v amdgpu.ids | tee $OUT/$v.txt
done
