#!/usr/bin/env python3
"""M = 1 projections of a Qwen2-7B decode step with DISTINCT weights per call (>= 1 GB cycled: nothing stays in the 256 MB MALL, as in a
real decode step where every layer has its own weights): us per call and TB/s of weights, inside a captured graph."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import ops
for (N, K, epi, name) in [(3584, 3584, "none", "q/o"), (1024, 3584, "none", "kv"), (3584, 18944, "none", "down"), (37888, 3584, "swiglu", "gate_up"), (152064, 3584, "none", "lm_head")]:
    copies = max(2, int(1.2e9 // (N * K * 2)))
    ws = [(torch.rand(N, K, device="cuda") - 0.5).half() for _ in range(copies)]
    x = (torch.rand(1, K, device="cuda") - 0.5).half()
    out = torch.empty(N // 2 if epi == "swiglu" else N, device="cuda", dtype=torch.float16)
    fn = lambda: [ops.gemv(w, x, None, epilogue=epi, out=out) for w in ws]
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [g.replay() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * copies)
    print(json.dumps(dict(name=name, N=N, K=K, copies=copies, us=round(us, 2), TBps=round(N * K * 2 / us / 1e6, 2))))
