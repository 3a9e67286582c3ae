#!/usr/bin/env python3
"""Calibration target for the DRAM-vs-Infinity-Cache question: streaming reads (ops.gemv = 16 B/lane non-temporal loads of a weight matrix) of
(a) ONE 64 MB matrix 24 times in a row (after the first pass the 256 MB Infinity Cache holds it) and (b) 24 DIFFERENT 64 MB matrices (1.5 GB: HBM).
    python tools/run_stream_read.py hot|cold"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import ops
mode = sys.argv[1]
N, K = 8192, 4096                          # 64 MB fp16
ws = [(torch.rand(N, K, device="cuda") - 0.5).half() for _ in range(1 if mode == "hot" else 24)]
x = (torch.rand(1, K, device="cuda") - 0.5).half()
out = torch.empty(N, device="cuda", dtype=torch.float16)
for i in range(24):
    ops.gemv(ws[i % len(ws)], x, None, out=out)
torch.cuda.synchronize()
