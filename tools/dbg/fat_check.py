import torch, sys
sys.path.insert(0, ".")
from streamchat_amd import ops
torch.manual_seed(0)
for (M, N, K) in [(256, 256, 128), (256, 256, 256), (512, 512, 1024), (5000, 4096, 1024), (300, 256, 128)]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).half(); w = ((torch.rand(N, K, device="cuda") * 2 - 1) * K ** -0.5).half()
    out = ops.gemm(a, w, None, None, "none", force=2) if "force" in ops.gemm.__code__.co_varnames else ops.gemm(a, w, None, None, "none")
    ref = a.float() @ w.float().t()
    err = (out.float() - ref).abs()
    bad = err > 2e-2
    print(M, N, K, "max err", err.max().item(), "bad frac", bad.float().mean().item())
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print("  bad rows", rows[:20].tolist(), "n", len(rows), " bad cols", cols[:20].tolist(), "n", len(cols))
