#!/usr/bin/env python3
"""M = 1 projections of a Qwen2-7B decode step: k_gemv (sc_gemv_f16) vs k_gemm_skinny (sc_gemm_f16 with M = 1): us and TB/s of weights."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import ops
for (N, K, epi, name) in [(3584, 3584, "none", "q/o"), (1024, 3584, "none", "kv"), (3584, 18944, "none", "down"), (37888, 3584, "swiglu", "gate_up"), (152064, 3584, "none", "lm_head")]:
    w = (torch.rand(N, K, device="cuda") - 0.5).half(); x = (torch.rand(1, K, device="cuda") - 0.5).half()
    res = {}
    for kind, fn in (("gemv", lambda: ops.gemv(w, x, None, epilogue=epi)), ("skinny", lambda: ops.gemm(x, w, None, None, epi))):
        g = torch.cuda.CUDAGraph()
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(20): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); [g.replay() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 100
        res[kind] = dict(us=round(us, 2), TBps=round(N * K * 2 / us / 1e6, 2))
    print(json.dumps(dict(name=name, N=N, K=K, **res)))
