#!/usr/bin/env python3
"""Decode attention at a 49 152-token context: the shipped cache layout (row = [K of 4 heads | V of 4 heads], 2 KiB; a head reads 256 B of
every row) against a head-major layout (each head's K and V rows contiguous), emulated through the batch dimension of sc_attention_f16
(B = 4 "sequences" of ONE head each: same bytes, same number of workgroups).  us per launch and TB/s of K+V bytes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import ops

S, G, H, Dh = int(sys.argv[1]) if len(sys.argv) > 1 else 49152, 7, 4, 128
ns_list = [int(x) for x in sys.argv[2:]] or [64, 128]
gb = 2 * S * H * Dh * 2 / 1e9


def timeit(fn):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    return sorted(ts)[2]


# several distinct caches, cycled, so that nothing is served from the 256 MB Infinity Cache (a decode step touches 2.8 GB between two uses)
NC = 6
rows = [torch.randn(S, 2 * H * Dh, device="cuda").half() for _ in range(NC)]
heads = [(torch.randn(H, S, Dh, device="cuda").half(), torch.randn(H, S, Dh, device="cuda").half()) for _ in range(NC)]
q = torch.randn(1, G * H * Dh, device="cuda").half()
kl = torch.tensor([S], device="cuda", dtype=torch.int32)
kl4 = torch.tensor([S] * H, device="cuda", dtype=torch.int32)
for ns in ns_list:
    i = [0]

    def row_major():
        ck = rows[i[0] % NC]; i[0] += 1
        qv = q.as_strided((1, G, Dh), (G * H * Dh, Dh, 1))
        return ops.attention(qv, ck[:, :H * Dh].unsqueeze(0), ck[:, H * Dh:].unsqueeze(0), H, H, Dh, Dh ** -0.5, causal=False, kv_len=kl, nsplit=ns,
                             q_head_stride=G * Dh, o_head_stride=G * Dh, out_ld=Dh)

    def head_major():
        k, v = heads[i[0] % NC]; i[0] += 1
        return ops.attention(q.view(H, G, Dh), k, v, 1, 1, Dh, Dh ** -0.5, causal=False, kv_len=kl4, nsplit=ns)
    a, b = timeit(row_major), timeit(head_major)
    print(json.dumps(dict(decode_kernel=os.environ.get("SC_ATTN_DECODE", "1"), S=S, nsplit=ns, row_major_us=round(a, 2), row_major_TBs=round(gb / a * 1e3, 2),
                          head_major_us=round(b, 2), head_major_TBs=round(gb / b * 1e3, 2))))
