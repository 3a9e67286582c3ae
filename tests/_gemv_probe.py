"""Helper of test_gpu_gemv_spec.py: every decode projection of the Qwen2-7B shapes on seeded inputs -> an .npz of raw output bits.
Run in a subprocess once with SC_GEMV_GENERIC=1 (the generic run-time-loop kernels of gemv.hip) and once without (the shape-specialised
straight-line kernels): the two files must be bit-identical."""
import sys
import numpy as np
import torch
from streamchat_amd import ops


def main(path):
    g = torch.Generator(device="cuda").manual_seed(5)
    rn = lambda *s, std=1.0: (torch.randn(*s, device="cuda", generator=g) * std).half()
    out = {}
    H, I, V = 3584, 18944, 152064
    x, xm = rn(H), rn(I, std=0.5)
    gamma, res, bias = 1 + rn(H, std=0.1), rn(H), rn(H, std=0.1)
    cases = dict(o=(rn(H, H, std=0.02), x, None, res, "none", False, None),
                 o_bias_norm=(rn(H, H, std=0.02), x, bias, res, "none", False, gamma),
                 q_f32=(rn(H, H, std=0.02), x, bias, None, "none", True, gamma),
                 kv_f32=(rn(1024, H, std=0.02), x, rn(1024, std=0.1), None, "none", True, gamma),
                 down=(rn(H, I, std=0.02), xm, None, res, "none", False, None),
                 gate_up=(rn(2 * 4096, H, std=0.02), x, None, None, "swiglu", False, gamma),
                 gate_up_nonorm=(rn(2 * 4096, H, std=0.02), x, None, None, "swiglu", False, None),
                 head=(rn(16384 + 48, H, std=0.02), x, None, None, "none", True, gamma),          # N % 16 != 0: the clamped last workgroup
                 wide_f16=(rn(16384, H, std=0.02), x, bias.repeat(5)[:16384].contiguous(), None, "none", False, None),
                 ragged_rows=(rn(H - 3, H, std=0.02), x, None, res[: H - 3].contiguous(), "none", False, None))
    for name, (w, xv, b, r, epi, f32, gm) in cases.items():
        y = ops.gemv(w, xv, b, residual=r, epilogue=epi, out_f32=f32, rms_gamma=gm, rms_eps=1e-6)
        out[name] = y.cpu().numpy().view(np.uint32 if f32 else np.uint16)
    # the fused q/k/v launch of the captured decode graph, three positions
    Hq, Hkv, Dh = 28, 4, 128
    wq, wkv, bq, bkv = rn(Hq * Dh, H, std=0.02), rn(2 * Hkv * Dh, H, std=0.02), rn(Hq * Dh, std=0.1), rn(2 * Hkv * Dh, std=0.1)
    tq, tk = ops.rope_table(4096, Dh, 1e6, Dh ** -0.5 * ops.LOG2E, "cuda"), ops.rope_table(4096, Dh, 1e6, 1.0, "cuda")
    cache = torch.zeros(4096, 2 * Hkv * Dh, dtype=torch.float16, device="cuda")
    for p in (0, 1, 3001):
        q = torch.empty(Hq * Dh, dtype=torch.float16, device="cuda")
        ops.decode_qkv_tab(wq, wkv, bq, bkv, x, gamma, 1e-6, q, cache, torch.tensor([p], dtype=torch.int32, device="cuda"), Hq, Hkv, Dh, tq, tk)
        out[f"qkv_q_{p}"] = q.cpu().numpy().view(np.uint16)
        out[f"qkv_row_{p}"] = cache[p].cpu().numpy().view(np.uint16)
    # batched decode projections (M <= 32 rows: k_gemm_skinny / k_gemm_skinny_u), 26 caption sequences and 8
    for M in (26, 8):
        xb, xmb = rn(M, H), rn(M, I, std=0.5)
        resb = rn(M, H)
        for name, (w, a, b, r, epi, f32) in dict(
                q=(cases["q_f32"][0], xb, bias, None, "none", True), o=(cases["o"][0], xb, None, resb, "none", False),
                gate_up=(cases["gate_up"][0], xb, None, None, "swiglu", False), down=(cases["down"][0], xmb, None, resb, "none", False)).items():
            y = ops.gemm(a, w, b, residual=r, epilogue=epi, out_f32=f32)
            out[f"gemm{M}_{name}"] = y.cpu().numpy().view(np.uint32 if f32 else np.uint16)
    # the LDS-ring skinny kernel on OTHER widths (run-time K: 2048 = the smallest it takes, 2560, 3072, 4096 = BERT-large's fc2) and row counts around the
    # 16-row tile boundary, few strips and many, with bias / residual / fp32 output: against the run-time-loop kernel bit for bit
    for (M, N, K, has_b, has_r, f32) in [(1, 1024, 4096, True, True, False), (5, 128, 2048, True, False, True), (16, 256, 2560, False, True, False),
                                         (17, 384, 3072, True, True, False), (32, 1024, 2048, False, False, True), (26, 8192, 2048, True, False, False)]:
        a, w = rn(M, K), rn(N, K, std=0.03)
        y = ops.gemm(a, w, rn(N, std=0.1) if has_b else None, residual=rn(M, N) if has_r else None, out_f32=f32)
        out[f"gemm_{M}x{N}x{K}"] = y.cpu().numpy().view(np.uint32 if f32 else np.uint16)
    torch.cuda.synchronize()
    np.savez(path, **out)


if __name__ == "__main__":
    main(sys.argv[1])
