#!/usr/bin/env python3
"""ViT attention (512 frames x 577 tokens, 16 heads of 64) reading q | k | v as column slices of ONE [M, 3072 (+ pad)] buffer, as vision.py does: does the
row stride (6144 B dense: every row of a K / V tile in the same slot of the channels' interleave) matter?  TF per padding of the row, in halves."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import ops
B, S, H, Dh = 512, 577, 16, 64
D = H * Dh
pads = [int(x) for x in sys.argv[1:]] or [0, 64, 128]
g = torch.Generator(device="cuda").manual_seed(0)
for rnd in range(2):
    for pad in pads:
        buf = (torch.randn(B * S, 3 * D + pad, device="cuda", generator=g) * 0.5).half()
        qkv = buf[:, :3 * D].view(B, S, 3 * D) if pad == 0 else buf.as_strided((B, S, 3 * D), (S * (3 * D + pad), 3 * D + pad, 1))
        out = torch.empty(B, S, D, device="cuda", dtype=torch.float16)
        fn = lambda: ops.attention(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], H, H, Dh, 0.125, False, out=out)
        for _ in range(3): fn()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10)
        ms = sorted(ts)[2]
        print(json.dumps(dict(round=rnd, row_pad_halves=pad, row_stride_bytes=(3 * D + pad) * 2, ms=round(ms, 4), TF=round(4.0 * B * H * S * S * Dh / ms / 1e9, 1))), flush=True)
        del buf, qkv, out
