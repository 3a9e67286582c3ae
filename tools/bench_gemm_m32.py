#!/usr/bin/env python3
"""The projections of a BATCHED decode step (M = 26 caption sequences, Qwen2-7B shapes): us per launch and TB/s of weights, hipGraph-timed.
SC_GEMM_KERNEL=128|256 pins the tile kernels instead of k_gemm_skinny (A/B)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 26
NW = 6            # distinct weight copies cycled, so that nothing is served from the Infinity Cache
for (N, K, epi, f32, name) in [(3584, 3584, "none", True, "q(f32)"), (1024, 3584, "none", True, "kv(f32)"), (3584, 3584, "none", False, "o+res"), (4608, 3584, "none", True, "qkv(f32)"),
                               (37888, 3584, "swiglu", False, "gate_up"), (3584, 18944, "none", False, "down+res"), (152064, 3584, "none", True, "lm_head")]:
    nw = 2 if N > 100000 else NW
    ws = [(torch.rand(N, K, device="cuda") - 0.5).half() for _ in range(nw)]
    x = (torch.rand(M, K, device="cuda") - 0.5).half()
    res = (torch.rand(M, N, device="cuda") - 0.5).half() if name.endswith("+res") else None
    fn = lambda w: ops.gemm(x, w, None, residual=res, epilogue=epi, out_f32=f32)
    fn(ws[0]); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(12): fn(ws[i % nw])
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [g.replay() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 60
    print(json.dumps(dict(kernel=os.environ.get("SC_GEMM_KERNEL", "skinny"), waves=os.environ.get("SC_SKINNY_WAVES", "rule"), M=M, name=name, N=N, K=K, us=round(us, 2), TBps=round(N * K * 2 / us / 1e6, 2))))
