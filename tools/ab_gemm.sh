#!/bin/bash
# A/B of GEMM variant libraries (tools/bin/lib_*.so built with -DSC_GEMM_*), interleaved rounds in one box
for round in 1 2; do
  for lib in default SETPRIO_1 GM_4 GM_16 MPL_2 MPL_3; do
    if [ $lib = default ]; then unset SC_LIB; else export SC_LIB=$PWD/tools/bin/lib_$lib.so; fi
    echo "== $lib round $round"; python tools/bench_gemm.py vit.qkv vit.fc1 vit.fc2 llm.gate 8192^3 2>&1 | grep -v amdgpu | python -c "
import sys, json
print(' '.join(f\"{json.loads(l)['name']}={json.loads(l)['TFLOPs']}\" for l in sys.stdin if l.startswith('{')))"
  done
done
