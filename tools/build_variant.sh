#!/bin/bash
# Design aid: build a variant of the library under another name (gpurun_out is scratch; variants go to screenpressor_amd/variants/).
#   tools/build_variant.sh NAME [-DFLAG ...]   ->  screenpressor_amd/variants/libscpr_NAME.so   (use with SCPR_AMD_LIB=...)
set -e
N=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/screenpressor_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-strict-aliasing -fPIC -shared -Wno-unused-result \
  -mllvm -align-all-nofallthru-blocks=6 -mllvm -enable-post-misched=false -mllvm -structurizecfg-skip-uniform-regions=true "$@" \
  -o $R/screenpressor_amd/variants/libscpr_$N.so $R/screenpressor_amd/csrc/scpr_amd.hip $R/screenpressor_amd/csrc/scpr_driver.cpp $R/screenpressor_amd/csrc/scpr_avi.cpp 2>&1 | grep -v warning | grep -v "^\s" || true
ls -la $R/screenpressor_amd/variants/libscpr_$N.so
