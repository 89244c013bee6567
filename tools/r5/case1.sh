#!/bin/bash
R=$PWD
run() { python3 - <<PY
import sys, os
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import stress_cases as S
for c in (1,):
    print(S.random_case(c, verbose=True))
PY
}
echo "== product"; run 2>&1 | grep -v amdgpu.ids | tail -12
echo "== product, vector rANS"; SCPR_RANS_SCALAR_MAX=0 run 2>&1 | grep -v amdgpu.ids | tail -3
echo "== old snapshot rule"; SCPR_AMD_LIB=$R/screenpressor_amd/variants/libscpr_oldsnap.so run 2>&1 | grep -v amdgpu.ids | tail -3
