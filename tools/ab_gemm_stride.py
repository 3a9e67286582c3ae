#!/usr/bin/env python3
"""Do the ViT GEMMs care about the ROW STRIDE of their operands?  (round 6, after the KV-cache finding: rows 2048 B apart put a tile's row segments into
the same slot of the channels' interleave.)  The ViT activations are [M, 1024] (2048-B rows), [M, 4096] (8192-B rows) and [M, 3072]: each shape of the
step at its real M with the A rows / C rows / residual rows padded by `pad` halves (views of wider buffers; the kernel takes lda / ldc / ldr)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import ops
VM = 512 * 577
SHAPES = [("vit.qkv+b", VM, 3072, 1024, "none", False), ("vit.o+res", VM, 1024, 1024, "none", True), ("vit.fc1+gelu", VM, 4096, 1024, "quick_gelu", False),
          ("vit.fc2+res", VM, 1024, 4096, "none", True)]
pads = [int(x) for x in sys.argv[1:]] or [0, 64]
g = torch.Generator(device="cuda").manual_seed(0)
def buf(rows, cols, pad, fill=True):
    t = torch.empty(rows, cols + pad, device="cuda", dtype=torch.float16)
    if fill:
        t.copy_((torch.rand(rows, cols + pad, device="cuda", generator=g) * 2 - 1).half())
    return t[:, :cols]
for rnd in range(2):
    for name, M, N, K, epi, has_res in SHAPES:
        w = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) * K ** -0.5).half()
        bias = (torch.rand(N, device="cuda", generator=g) - 0.5).half()
        for pa, pc in [(p1, p2) for p1 in pads for p2 in pads]:
            a, out = buf(M, K, pa), buf(M, N, pc, fill=False)
            res = buf(M, N, pc) if has_res else None
            run = lambda: ops.gemm(a, w, bias, res, epi, out=out)
            for _ in range(3): run()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): run()
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10)
            ms = sorted(ts)[2]
            print(json.dumps(dict(round=rnd, shape=name, lda_pad=pa, ldc_pad=pc, ms=round(ms, 4), TF=round(2 * M * N * K / ms / 1e9, 1))), flush=True)
            del a, out, res
