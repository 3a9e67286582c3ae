#!/usr/bin/env python3
"""Reference point only (never on the product path): torch.matmul (hipBLASLt / rocBLAS) on the bench's GEMM shapes, same timing method."""
import json, os, sys
import torch
shapes = [(56 * 577, 3072, 1024, "vit56.qkv"), (56 * 577, 4096, 1024, "vit56.fc1"), (56 * 577, 1024, 4096, "vit56.fc2"), (512 * 577, 4096, 1024, "vit512.fc1"),
          (48994, 3584, 3584, "llm49k.q"), (48994, 3584, 18944, "llm49k.down"), (26112, 18944, 3584, "llm.gate"), (8192, 8192, 8192, "8192^3")]
for (M, N, K, name) in shapes:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).half(); w = (torch.rand(N, K, device="cuda") * 2 - 1).half()
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    for _ in range(3): torch.matmul(a, w.t(), out=out)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): torch.matmul(a, w.t(), out=out)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 5)
    ms = sorted(ts)[2]
    print(json.dumps(dict(name=name, M=M, N=N, K=K, ms=round(ms, 4), TFLOPs=round(2 * M * N * K / ms / 1e9, 1))))
