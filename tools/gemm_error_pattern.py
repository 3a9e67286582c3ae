#!/usr/bin/env python3
"""Where does a GEMM result differ from fp32 torch?  Prints the error pattern by accumulator coordinates (row % 16 = lane row,
row tile, column % 16, column tile, 256-tile) for the four bias/residual combinations - the tool that located the store-data
hazard and the asynchronous-MFMA read in k_gemm_fat (profiles/r01_run158/161).   python tools/gemm_error_pattern.py [epi] [M N K]"""
import sys, torch
sys.path.insert(0, ".")
from streamchat_amd import ops
epi = sys.argv[1] if len(sys.argv) > 1 else "none"
M, N, K = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (1154, 4096, 1024)
def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).half().cuda()
a, w, b, r = _rand((M, K), 1), _rand((N, K), 2, K ** -0.5), _rand((N,), 3), _rand((M, N), 4)
for name, bb, rr in [("plain", None, None), ("bias", b, None), ("res", None, r), ("both", b, r)]:
    out = ops.gemm(a, w, bb, rr, epi)
    ref = a.float() @ w.float().t()
    if bb is not None: ref = ref + bb.float()
    if epi == "quick_gelu": ref = ref * torch.sigmoid(1.702 * ref)
    elif epi == "gelu": ref = torch.nn.functional.gelu(ref)
    if rr is not None: ref = ref + rr.float()
    err = (out.float() - ref).abs(); bad = (err > 2e-2) | torch.isnan(out.float())
    print(name, "bad", bad.sum().item(), "nan", torch.isnan(out).sum().item(), "max err", err[~torch.isnan(err)].max().item())
    if bad.any():
        idx = bad.nonzero(); rows, cols = idx[:, 0], idx[:, 1]
        print("  rows%16", torch.bincount(rows % 16, minlength=16).tolist()); print("  row tile%16", torch.bincount((rows // 16) % 16, minlength=16).tolist())
        print("  cols%16", torch.bincount(cols % 16, minlength=16).tolist()); print("  col tile%16", torch.bincount((cols // 16) % 16, minlength=16).tolist())
        print("  256-tile rows", torch.bincount(rows // 256).tolist(), "cols", torch.bincount(cols // 256).tolist())
        for i, j in idx[:6].tolist(): print("   ", i, j, out[i, j].item(), ref[i, j].item())
