import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from streamchat_amd import ops
M, N = 32312, 4096
for K in (256, 512, 1024, 2048, 4096):
    a = (torch.rand(M, K, device="cuda") * 2 - 1).half(); w = (torch.rand(N, K, device="cuda") * 2 - 1).half()
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    for _ in range(3): ops.gemm(a, w, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.gemm(a, w, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(json.dumps(dict(K=K, ms=round(ms, 4), TF=round(2 * M * N * K / ms / 1e9, 1))))
