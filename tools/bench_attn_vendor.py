#!/usr/bin/env python3
"""Reference point only (never on the product path): torch SDPA (AOTriton / CK flash attention) on the bench's attention shapes."""
import json, sys, torch
import torch.nn.functional as F
cases = [("llm26k", 1, 26112, 28, 4, 128, True), ("llm49k", 1, 49152, 28, 4, 128, True), ("vit512", 512, 577, 16, 16, 64, False)]
for (name, B, S, Hq, Hkv, Dh, causal) in cases:
    q = torch.randn(B, Hq, S, Dh, device="cuda").half(); k = torch.randn(B, Hkv, S, Dh, device="cuda").half(); v = torch.randn(B, Hkv, S, Dh, device="cuda").half()
    for mode in ("gqa", "expanded"):
        try:
            if mode == "gqa":
                f = lambda: F.scaled_dot_product_attention(q, k, v, is_causal=causal, enable_gqa=(Hq != Hkv))
            else:
                kk, vv = k.repeat_interleave(Hq // Hkv, 1), v.repeat_interleave(Hq // Hkv, 1)
                f = lambda: F.scaled_dot_product_attention(q, kk, vv, is_causal=causal)
            for _ in range(2): f()
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); f(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 2)
            ms = sorted(ts)[1]
            fl = 4.0 * B * Hq * S * S * Dh * (0.5 if causal else 1.0)
            print(json.dumps(dict(name=name, mode=mode, ms=round(ms, 3), TFLOPs=round(fl / ms / 1e9, 1))))
        except Exception as e:
            print(json.dumps(dict(name=name, mode=mode, error=str(e)[:200])))
        if Hq == Hkv: break
