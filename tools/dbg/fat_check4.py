import torch, sys
sys.path.insert(0, ".")
from streamchat_amd import ops
def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).half().cuda()
M, N, K = 1154, 4096, 1024
a, w, b = _rand((M, K), 1), _rand((N, K), 2, K ** -0.5), _rand((N,), 3)
r = _rand((M, N), 4)
for name, bb, rr in [("plain", None, None), ("bias", b, None), ("res", None, r), ("both", b, r)]:
    out = ops.gemm(a, w, bb, rr, "none")
    ref = a.float() @ w.float().t()
    if bb is not None: ref = ref + bb.float()
    if rr is not None: ref = ref + rr.float()
    err = (out.float() - ref).abs(); bad = (err > 2e-2) | torch.isnan(out.float())
    print(name, "bad", bad.sum().item(), "nan", torch.isnan(out).sum().item())
    if bad.any():
        idx = bad.nonzero()
        rows, cols = idx[:, 0], idx[:, 1]
        print("  rows%16 hist", torch.bincount(rows % 16, minlength=16).tolist())
        print("  rows//16%16 hist", torch.bincount((rows // 16) % 16, minlength=16).tolist())
        print("  cols%16 hist", torch.bincount(cols % 16, minlength=16).tolist())
        print("  cols//16%16 hist", torch.bincount((cols // 16) % 16, minlength=16).tolist())
        print("  tile rows", torch.bincount(rows // 256).tolist(), "tile cols", torch.bincount(cols // 256).tolist())
        for i, j in idx[:6].tolist(): print("   ", i, j, out[i, j].item(), ref[i, j].item(), (rr[i, j].item() if rr is not None else None))
