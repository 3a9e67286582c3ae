#!/usr/bin/env python3
"""GEMM micro-benchmark at the encoder shapes (GPU box): TFLOP/s on uniform random operands (guide rule 25)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import ops

shapes = [(56 * 577, 3072, 1024, "vit56.qkv"), (56 * 577, 1024, 1024, "vit56.o"), (56 * 577, 4096, 1024, "vit56.fc1"), (56 * 577, 1024, 4096, "vit56.fc2"),
          (48994, 3584, 3584, "llm49k.q"), (48994, 3584, 18944, "llm49k.down"),
          (64 * 577, 3072, 1024, "vit.qkv"), (64 * 577, 1024, 1024, "vit.o"), (64 * 577, 4096, 1024, "vit.fc1"), (64 * 577, 1024, 4096, "vit.fc2"),
          (512 * 577, 4096, 1024, "vit512.fc1"), (512 * 577, 4096, 1024, "vit512.fc1+gelu"), (512 * 577, 1024, 4096, "vit512.fc2+res"),
          (512 * 577, 3072, 1024, "vit512.qkv+b"), (512 * 577, 1024, 1024, "vit512.o+res"), (26112, 37888, 3584, "llm.gateup+swiglu"),
          (64 * 576, 3584, 3584, "proj.2"), (4096, 4096, 4096, "4096^3"), (8192, 8192, 8192, "8192^3"), (26112, 18944, 3584, "llm.gate")]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if s[3] in sys.argv[1:]]
for (M, N, K, name) in shapes:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).half()
    w = (torch.rand(N, K, device="cuda") * 2 - 1).half()
    epi = "quick_gelu" if "+gelu" in name else ("swiglu" if "+swiglu" in name else "none")
    bias = (torch.rand(N, device="cuda") - 0.5).half() if "+" in name else None
    res = (torch.rand(M, N, device="cuda") - 0.5).half() if "+res" in name else None
    out = torch.empty(M, N // 2 if epi == "swiglu" else N, device="cuda", dtype=torch.float16)
    run = lambda: ops.gemm(a, w, bias, res, epi, out=out)
    for _ in range(3):
        run()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    ms = sorted(ts)[len(ts) // 2]
    print(json.dumps(dict(name=name, M=M, N=N, K=K, ms=round(ms, 4), TFLOPs=round(2 * M * N * K / ms / 1e9, 1))))
