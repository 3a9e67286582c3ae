#!/usr/bin/env python3
"""Does a decode projection run faster when its weights were just read by somebody else (HBM -> Infinity Cache)?  Per shape, ONE gemv timed with
HIP events right after (a) a streaming read of ITS OWN weights (hot: the 256 MB MALL holds them) and (b) a streaming read of ANOTHER copy (cold).
The premise of prefetching the next kernel's weights with the spare HBM bandwidth of the latency-bound decode attention (round 4)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import ops
for (N, K, epi, name) in [(3584, 3584, "none", "o"), (4608, 3584, "none", "qkv-sized"), (3584, 18944, "none", "down"), (37888, 3584, "swiglu", "gate_up"), (18944, 3584, "swiglu", "gate_up/2")]:
    copies = max(3, int(1.2e9 // (N * K * 2)))
    ws = [(torch.rand(N, K, device="cuda") - 0.5).half() for _ in range(copies)]
    x = (torch.rand(1, K, device="cuda") - 0.5).half()
    out = torch.empty(N // 2 if epi == "swiglu" else N, device="cuda", dtype=torch.float16)
    res = {}
    for mode in ("cold", "hot"):
        ts = []
        for rep in range(3):
            for i in range(copies):
                touch = ws[i] if mode == "hot" else ws[(i + 1) % copies]
                touch.view(torch.int32).max()                 # streaming read through the normal (allocating) path
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.gemv(ws[i], x, None, epilogue=epi, out=out)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        res[mode] = ts[len(ts) // 2]
    mb = N * K * 2 / 1e6
    print(json.dumps(dict(name=name, MB=round(mb, 1), cold_us=round(res["cold"], 2), hot_us=round(res["hot"], 2), cold_TBps=round(mb / res["cold"], 2), hot_TBps=round(mb / res["hot"], 2))), flush=True)
    del ws
    torch.cuda.empty_cache()
