#!/bin/bash
# Design aid: a variant library with EXTRA compiler flags appended to the product's (tools/build_variant.sh passes -D only)
N=$1; shift
R=$(cd $(dirname $0)/../.. && pwd)
mkdir -p $R/screenpressor_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-strict-aliasing -fPIC -shared -Wno-unused-result \
  -mllvm -align-all-nofallthru-blocks=6 -mllvm -enable-post-misched=false -mllvm -structurizecfg-skip-uniform-regions=true "$@" \
  -o $R/screenpressor_amd/variants/libscpr_$N.so $R/screenpressor_amd/csrc/scpr_amd.hip $R/screenpressor_amd/csrc/scpr_driver.cpp $R/screenpressor_amd/csrc/scpr_avi.cpp 2>&1 | grep -E "error|unknown|Unknown" | head -3
ls -la $R/screenpressor_amd/variants/libscpr_$N.so 2>&1 | tail -1
