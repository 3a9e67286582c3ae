#!/usr/bin/env python3
"""Run one GEMM shape a few times (profiling target for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4])
a = (torch.rand(M, K, device="cuda") * 2 - 1).half(); w = (torch.rand(N, K, device="cuda") * 2 - 1).half()
out = torch.empty(M, N, device="cuda", dtype=torch.float16)
for _ in range(int(sys.argv[4]) if len(sys.argv) > 4 else 4):
    ops.gemm(a, w, out=out)
torch.cuda.synchronize()
