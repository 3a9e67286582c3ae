#!/bin/bash
for r in 1 2 3; do
  for k in 256 2562; do echo -n "kernel=$k: "; SC_GEMM_KERNEL=$k python tools/bench_gemm.py vit.qkv vit.fc1 vit.fc2 proj.2 llm.gate 8192^3 2>&1 | grep -v amdgpu | python tools/fmt_gemm.py; done
done
