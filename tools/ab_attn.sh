#!/bin/bash
# same-box A/B of attention variants: default lib vs tools/bin/lib_attn_old.so
python -m pytest tests/test_gpu_dense.py -m gpu -q -x -k "attn or attention" 2>&1 | grep -E "Error|error|assert|passed|failed" | head -12
for round in 1 2; do
  for lib in old new; do
    if [ $lib = new ]; then unset SC_LIB; else export SC_LIB=$PWD/tools/bin/lib_attn_old.so; fi
    echo "== $lib round $round"; python tools/bench_attn.py 2>&1 | grep -v amdgpu | tail -4 | cut -c1-300
  done
done
