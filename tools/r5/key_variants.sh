#!/bin/bash
# Runs ON THE GPU BOX: decode time of 8 1080p key frames with the product and each variant library named on the command line
R=$PWD
python3 tools/exp_key1.py 8 2>&1 | grep -v amdgpu.ids
for v in "$@"; do SCPR_AMD_LIB=$R/screenpressor_amd/variants/libscpr_$v.so python3 tools/exp_key1.py 8 2>&1 | grep -v amdgpu.ids; done
python3 tools/exp_key1.py 8 2>&1 | grep -v amdgpu.ids
