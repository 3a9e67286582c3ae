#!/usr/bin/env python3
"""A few launches of ONE skinny GEMM shape (M rows) for counter passes:  python tools/run_one_skinny.py M N K [residual]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4])
res = len(sys.argv) > 4
ws = [(torch.rand(N, K, device="cuda") - 0.5).half() for _ in range(4)]
x = (torch.rand(M, K, device="cuda") - 0.5).half()
r = (torch.rand(M, N, device="cuda") - 0.5).half() if res else None
for i in range(12):
    ops.gemm(x, ws[i % 4], None, residual=r)
torch.cuda.synchronize()
