import torch, sys
sys.path.insert(0, ".")
from streamchat_amd import ops
torch.manual_seed(0)
for (M, N, K) in [(577, 1024, 1024), (1154, 4096, 1024), (1154, 1024, 1024), (1024, 4096, 1024), (1280, 4096, 1024)]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).half(); w = ((torch.rand(N, K, device="cuda") * 2 - 1) * K ** -0.5).half()
    b = (torch.rand(N, device="cuda") - 0.5).half(); r = (torch.rand(M, N, device="cuda") - 0.5).half()
    for name, bb, rr in [("plain", None, None), ("bias", b, None), ("res", None, r), ("both", b, r)]:
        out = ops.gemm(a, w, bb, rr, "none")
        ref = a.float() @ w.float().t()
        if bb is not None: ref = ref + bb.float()
        if rr is not None: ref = ref + rr.float()
        err = (out.float() - ref).abs()
        bad = err > 2e-2
        print(M, N, K, name, "max err", round(err.max().item(), 5), "bad frac", bad.float().mean().item())
        if bad.any():
            rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
            print("  bad rows", rows[:12].tolist(), "..", rows[-4:].tolist(), "n", len(rows), " bad cols", cols[:12].tolist(), "..", cols[-4:].tolist(), "n", len(cols))
