#!/usr/bin/env python3
"""Decode attention at a 49 k context against the ROW STRIDE of the KV cache: rows of [K of 4 heads | V of 4 heads] = 2048 B put head h's
256-byte K piece of consecutive rows 2048 B apart (8 x 256 B: every row of a head lands in the same 1/8 of a 256-B-interleaved channel
set); a padded stride (2048 + pad bytes) walks the channels.  us per launch (attention + combine), TB/s of K+V bytes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import ops

S, G, H, Dh = int(sys.argv[1]) if len(sys.argv) > 1 else 49152, 7, 4, 128
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 64
pads = [int(x) for x in sys.argv[3:]] or [0, 64, 128, 192, 256]          # in halves
gb = 2 * S * H * Dh * 2 / 1e9
q = torch.randn(1, G * H * Dh, device="cuda").half()
kl = torch.tensor([S], device="cuda", dtype=torch.int32)
qv = q.as_strided((1, G, Dh), (G * H * Dh, Dh, 1))
NC = 6
for pad in pads:
    rows = [torch.randn(S, 2 * H * Dh + pad, device="cuda").half()[:, :2 * H * Dh] for _ in range(NC)]
    i = [0]

    def fn():
        ck = rows[i[0] % NC]; i[0] += 1
        return ops.attention(qv, ck[:, :H * Dh].unsqueeze(0), ck[:, H * Dh:].unsqueeze(0), H, H, Dh, Dh ** -0.5, causal=False, kv_len=kl, nsplit=ns,
                             q_head_stride=G * Dh, o_head_stride=G * Dh, out_ld=Dh)
    for _ in range(6):
        fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(24):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 24 * 1e3)
    us = sorted(ts)[2]
    print(json.dumps(dict(S=S, nsplit=ns, row_stride_bytes=(2 * H * Dh + pad) * 2, us=round(us, 2), TBps=round(gb / us * 1e3, 2))), flush=True)
    del rows
