import torch, sys
sys.path.insert(0, ".")
from streamchat_amd import ops
torch.manual_seed(0)
def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).half().cuda()
for (M, N, K) in [(1154, 4096, 1024)]:
    a, w, b = _rand((M, K), 1), _rand((N, K), 2, K ** -0.5), _rand((N,), 3)
    r = _rand((M, N), 4)
    for epi in ["none", "quick_gelu", "gelu"]:
        out = ops.gemm(a, w, b, r, epi)
        ref = a.float() @ w.float().t() + b.float()
        if epi == "quick_gelu": ref = ref * torch.sigmoid(1.702 * ref)
        elif epi == "gelu": ref = torch.nn.functional.gelu(ref)
        ref = ref + r.float()
        err = (out.float() - ref).abs(); tol = 2e-3 + 2e-3 * ref.abs()
        bad = err > tol
        print(epi, "max err", err.max().item(), "bad", bad.sum().item(), "nan", torch.isnan(out).sum().item())
        if bad.any():
            idx = bad.nonzero()[:8]
            for i, j in idx.tolist(): print("   ", i, j, out[i, j].item(), ref[i, j].item())
