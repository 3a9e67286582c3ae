#!/usr/bin/env python3
"""Attention micro-benchmark (GPU box): TFLOP/s on random data.  flops = 4*B*H*Sq*Skv*Dh (x0.5 causal)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import ops

cases = [("vit", 64, 577, 16, 16, 64, False), ("llm8k", 1, 8192, 28, 4, 128, True), ("llm26k", 1, 26112, 28, 4, 128, True),
         ("full26k", 1, 26112, 28, 4, 128, False), ("llm49k", 1, 49152, 28, 4, 128, True), ("vit512", 512, 577, 16, 16, 64, False)]
PRE = "--pre" in sys.argv            # q handed over pre-scaled (SC_ATTN_Q_PRESCALED): what the ViT / Qwen2 paths run since round 3
names = [a for a in sys.argv[1:] if not a.startswith("--")]
if names:
    cases = [c for c in cases if c[0] in names]
for (name, B, S, Hq, Hkv, Dh, causal) in cases:
    q = (torch.randn(B, S, Hq * Dh, device="cuda") * (Dh ** -0.5 * 1.4426950408889634 if PRE else 1.0)).half()     # pre-scaled: q carries scale * log2 e
    k = torch.randn(B, S, Hkv * Dh, device="cuda").half()
    v = torch.randn(B, S, Hkv * Dh, device="cuda").half()
    out = torch.empty(B, S, Hq * Dh, device="cuda", dtype=torch.float16)
    for _ in range(2):
        ops.attention(q, k, v, Hq, Hkv, Dh, Dh ** -0.5, causal, out=out, q_prescaled=PRE)
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ops.attention(q, k, v, Hq, Hkv, Dh, Dh ** -0.5, causal, out=out, q_prescaled=PRE)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 3)
    ms = sorted(ts)[1]
    fl = 4.0 * B * Hq * S * S * Dh * (0.5 if causal else 1.0)
    print(json.dumps(dict(name=name, pre=PRE, B=B, S=S, Hq=Hq, Hkv=Hkv, Dh=Dh, causal=causal, ms=round(ms, 4), TFLOPs=round(fl / ms / 1e9, 1))))
