#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import ops
S = int(sys.argv[1]); Hq, Hkv, Dh, causal = 28, 4, 128, True
if len(sys.argv) > 2 and sys.argv[2].startswith("vit"):
    B, S, Hq, Hkv, Dh, causal = (512 if sys.argv[2] == "vit512" else 56), 577, 16, 16, 64, False
else:
    B = 1
PRE = os.environ.get("SC_RUN_PRE", "1") == "1"           # q handed over pre-scaled (what the ViT / Qwen2 paths run since round 3)
q = (torch.randn(B, S, Hq * Dh, device="cuda") * (Dh ** -0.5 * 1.4426950408889634 if PRE else 1.0)).half(); k = torch.randn(B, S, Hkv * Dh, device="cuda").half(); v = torch.randn(B, S, Hkv * Dh, device="cuda").half()
out = torch.empty(B, S, Hq * Dh, device="cuda", dtype=torch.float16)
for _ in range(3):
    ops.attention(q, k, v, Hq, Hkv, Dh, Dh ** -0.5, causal, out=out, q_prescaled=PRE)
torch.cuda.synchronize()
