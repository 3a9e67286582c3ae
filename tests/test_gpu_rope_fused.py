"""GPU: rotary embedding applied to the fp32 projection sums with ONE rounding, the softmax scale folded into the query (round 3):
the rotary / column-scale epilogues of the hand-scheduled GEMM (sc_gemm_headed_f16), the fp32-in rotary kernel of the small-M paths
(sc_rope_f32in_f16), the decode step's single launch (sc_decode_qkv_tab_f16) and sc_attention_f16's SC_ATTN_Q_PRESCALED mode.
The three producers must agree BIT FOR BIT (prefill rows and decode rows of one sequence meet in the same KV cache; the captured decode
graph and the eager step must emit the same tokens); against fp64 torch they must be at least as close as the fp16 pipeline they replace
(HF rounds the projection, cos / sin, both products and the sum: reference llava_qwen.py:155 -> transformers Qwen2Attention)."""
import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from streamchat_amd import ops          # noqa: E402

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, device="cuda", generator=g) * scale).half()


def _rope_ref64(x, heads, Dh, theta, pos0, scale=1.0):
    """fp64 rotate-half RoPE of x [rows, heads*Dh] at positions pos0 + r"""
    rows = x.shape[0]
    pos = torch.arange(pos0, pos0 + rows, device=x.device, dtype=torch.float64)
    inv = 1.0 / (theta ** (torch.arange(0, Dh, 2, device=x.device, dtype=torch.float64) / Dh))
    fr = torch.outer(pos, inv)
    cos, sin = torch.cat([fr, fr], -1).cos()[:, None], torch.cat([fr, fr], -1).sin()[:, None]
    xf = x.double().view(rows, heads, Dh)
    return ((xf * cos + torch.cat([-xf[..., Dh // 2:], xf[..., :Dh // 2]], -1) * sin) * scale).view(rows, heads * Dh)


def test_rope_table_matches_the_formula():
    Dh, theta, scale = 128, 1e6, 0.1275
    t = ops.rope_table(3000, Dh, theta, scale, "cuda:0")
    assert t.shape[0] >= 3000 and t.shape[1:] == (2, 64) and t.dtype == torch.float32
    pos = torch.arange(2048, device="cuda", dtype=torch.float64)
    inv = 1.0 / (theta ** (torch.arange(0, Dh, 2, device="cuda", dtype=torch.float64) / Dh))
    fr = torch.outer(pos, inv)
    torch.testing.assert_close(t[:2048, 0].double(), fr.cos() * scale, rtol=0, atol=4e-4 * scale)   # fp32 angle: 2048 * 2^-23 relative
    torch.testing.assert_close(t[:2048, 1].double(), fr.sin() * scale, rtol=0, atol=4e-4 * scale)
    assert ops.rope_table(100, Dh, theta, scale, "cuda:0").data_ptr() == t.data_ptr()               # cached, pointer-stable (graphs read it)


@pytest.mark.parametrize("M,N,lead,K,pos0", [(5000, 3584, 3584, 3584, 0), (4100, 1024, 512, 3584, 777), (300, 512, 256, 256, 3), (2048, 256, 256, 128, 40000)])
def test_gemm_rotary_epilogue_equals_fp32_projection_plus_rope_kernel_bitwise(M, N, lead, K, pos0):
    """Qwen2 q projection (28 heads) and k|v projection (4 rotary heads + 4 plain) at the 7B widths, persistent walk and single-tile launches,
    ragged last row tile, a position offset (chunked prefill) and far positions"""
    a, w, b = _rand((M, K), 1), _rand((N, K), 2, K ** -0.5), _rand((N,), 3)
    tab = ops.rope_table(pos0 + M, 128, 1e6, 0.1275 if lead == N else 1.0, "cuda:0")
    out = torch.empty((M, N), dtype=torch.float16, device="cuda")
    assert ops.gemm_headed_ok(N, K, a, w, b, out)
    # (the predicate mirrors the C preconditions - an odd row stride or a misaligned view must say False, not raise later)
    assert not ops.gemm_headed_ok(N, K, a, w, b, torch.empty((M, N + 4), dtype=torch.float16, device="cuda")[:, :N])
    assert not ops.gemm_headed_ok(N, K, a, w, b[1:] if False else torch.empty(N + 8, dtype=torch.float16, device="cuda")[1:N + 1], out)
    fused = ops.gemm_headed(a, w, b, out, "rope", lead, tab, pos0)
    acc = ops.gemm(a, w, b, out_f32=True)
    two = ops.rope_f32in(acc, tab, lead // 128, 128, torch.empty((M, N), dtype=torch.float16, device="cuda"), N - lead, pos0)
    assert torch.equal(fused, two)
    # and both against fp64 (table angles are fp32: compare through the table itself so that only the rotation arithmetic is judged)
    z = a.double() @ w.double().t() + b.double()
    cos, sin = tab[pos0:pos0 + M, 0].double(), tab[pos0:pos0 + M, 1].double()
    zr = z[:, :lead].view(M, lead // 128, 128)
    ref = torch.cat([(zr[..., :64] * cos[:, None] - zr[..., 64:] * sin[:, None]), (zr[..., 64:] * cos[:, None] + zr[..., :64] * sin[:, None])], -1).view(M, lead)
    ref = torch.cat([ref, z[:, lead:]], 1)
    err = (fused.double() - ref).abs().max().item()
    assert err <= 1.1 * 2 ** -11 * ref.abs().max().item() + 1e-4, err        # ONE fp16 rounding of an fp32-accurate value


def test_gemm_column_scale_epilogue():
    """CLIP's fused q|k|v projection: the q third times scale * log2 e on the fp32 sum, k / v thirds untouched; ViT-L widths, persistent walk"""
    M, D = 6000, 1024
    a, w, b = _rand((M, D), 4), _rand((3 * D, D), 5, D ** -0.5), _rand((3 * D,), 6)
    c = 0.125 * ops.LOG2E
    out = ops.gemm_headed(a, w, b, torch.empty((M, 3 * D), dtype=torch.float16, device="cuda"), "colscale", D, col_scale=c)
    plain = ops.gemm(a, w, b)
    assert torch.equal(out[:, D:], plain[:, D:])                              # k, v: the ordinary epilogue's bits
    z = (a.double() @ w[:D].double().t() + b[:D].double()) * c
    err = (out[:, :D].double() - z).abs().max().item()
    assert err <= 1.1 * 2 ** -11 * z.abs().max().item() + 1e-4, err


def test_rope_f32in_positions_and_plain_columns():
    rows, heads, Dh, plain = 37, 4, 128, 512
    x = torch.randn(rows, heads * Dh + plain, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7))
    pos = torch.randint(0, 5000, (rows,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(8), dtype=torch.int32)
    tab = ops.rope_table(5000, Dh, 1e6, 1.0, "cuda:0")
    out = ops.rope_f32in(x, tab, heads, Dh, torch.empty((rows, heads * Dh + plain), dtype=torch.float16, device="cuda"), plain, 0, pos)
    assert torch.equal(out[:, heads * Dh:], x[:, heads * Dh:].half())
    for r in (0, 11, 36):
        one = ops.rope_f32in(x[r:r + 1].contiguous(), tab, heads, Dh, torch.empty((1, heads * Dh + plain), dtype=torch.float16, device="cuda"), plain, int(pos[r]))
        assert torch.equal(one[0], out[r])
    ref = torch.cat([_rope_ref64(x[r:r + 1, :heads * Dh], heads, Dh, 1e6, int(pos[r])) for r in range(rows)])
    torch.testing.assert_close(out[:, :heads * Dh].double(), ref, rtol=0, atol=2e-3 * float(ref.abs().max()))    # incl. fp32 angle error at pos ~5000


def test_decode_qkv_tab_equals_gemv_plus_rope_kernel_bitwise():
    """the captured graph's single launch vs the eager step's three: same bits in q and in the appended cache row"""
    Hq, Hkv, Dh, K, pos = 28, 4, 128, 3584, 4321
    wq, wkv = _rand((Hq * Dh, K), 1, K ** -0.5), _rand((2 * Hkv * Dh, K), 2, K ** -0.5)
    bq, bkv, x, g = _rand((Hq * Dh,), 3), _rand((2 * Hkv * Dh,), 4), _rand((K,), 5), (1 + _rand((K,), 6, 0.1).float()).half()
    tq, tk = ops.rope_table(5000, Dh, 1e6, Dh ** -0.5 * ops.LOG2E, "cuda:0"), ops.rope_table(5000, Dh, 1e6, 1.0, "cuda:0")
    row = torch.tensor([pos], device="cuda", dtype=torch.int32)
    cache_a = torch.zeros((5000, 2 * Hkv * Dh), dtype=torch.float16, device="cuda")
    q_a = ops.decode_qkv_tab(wq, wkv, bq, bkv, x, g, 1e-6, torch.empty(Hq * Dh, dtype=torch.float16, device="cuda"), cache_a, row, Hq, Hkv, Dh, tq, tk)
    q32 = ops.gemv(wq, x, bq, out_f32=True, rms_gamma=g, rms_eps=1e-6).view(1, -1)
    kv32 = ops.gemv(wkv, x, bkv, out_f32=True, rms_gamma=g, rms_eps=1e-6).view(1, -1)
    q_b = ops.rope_f32in(q32, tq, Hq, Dh, torch.empty((1, Hq * Dh), dtype=torch.float16, device="cuda"), 0, pos)
    cache_b = torch.zeros_like(cache_a)
    ops.rope_f32in(kv32, tk, Hkv, Dh, cache_b[pos:pos + 1], Hkv * Dh, pos)
    assert torch.equal(q_a.view(-1), q_b.view(-1)) and torch.equal(cache_a, cache_b) and cache_a[pos].abs().sum() > 0
    # fp64: q = rope(W rmsnorm(x) + b) * scale * log2 e
    xn = (g.double() * (x.double() * torch.rsqrt((x.double() ** 2).mean() + 1e-6)).half().double()).half().double()
    ref = _rope_ref64((wq.double() @ xn + bq.double()).view(1, -1), Hq, Dh, 1e6, pos, Dh ** -0.5 * ops.LOG2E)
    torch.testing.assert_close(q_a.double().view(1, -1), ref, rtol=0, atol=3e-3 * float(ref.abs().max()))


@pytest.mark.parametrize("B,Sq,Skv,Hq,Hkv,Dh,causal,ns", [(2, 577, 577, 16, 16, 64, False, 1), (1, 2304, 2304, 4, 2, 128, True, 1), (1, 300, 450, 28, 4, 128, True, 1),
                                                           (1, 7, 3000, 4, 4, 128, False, 16), (1, 40, 40, 2, 2, 32, False, 1)])
def test_attention_with_prescaled_queries(B, Sq, Skv, Hq, Hkv, Dh, causal, ns):
    """SC_ATTN_Q_PRESCALED: q' = fp16(q * scale * log2 e) as a producer hands it over; softmax_2(q'.k) v must equal softmax((q'/log2 e).k) v"""
    from tests.test_gpu_dense import _attn_ref
    scale = Dh ** -0.5
    q32 = torch.randn(B, Sq, Hq * Dh, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) * 3.0
    qp = (q32 * (scale * ops.LOG2E)).half()
    k, v = _rand((B, Skv, Hkv * Dh), 2), _rand((B, Skv, Hkv * Dh), 3)
    out = ops.attention(qp, k, v, Hq, Hkv, Dh, 123.0, causal, nsplit=ns, q_prescaled=True)          # (scale is ignored in this mode)
    ref = _attn_ref(qp.float(), k, v, Hq, Hkv, Dh, 1.0 / ops.LOG2E, causal)
    torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)


def test_rope_qkv_rows_equals_rope_f32in_twice_plus_scatter():
    """ABI 6 `sc_rope_qkv_rows_f16` (the batched decode step's rotary + KV append in ONE launch) is bit for bit what the three launches it
    replaced produce: sc_rope_f32in_f16 on the q columns (scaled table), on the k|v columns (plain table, V cast), and a row scatter into each
    sequence's cache at its own position - positions differ per sequence, the other cache rows stay untouched, out-of-range positions clamp."""
    import torch
    from streamchat_amd import ops
    torch.manual_seed(3)
    B, Hq, Hkv, Dh, cap = 5, 28, 4, 128, 40
    dq, dkv = Hq * Dh, Hkv * Dh
    x = torch.randn(B, dq + 2 * dkv, device="cuda") * 3
    tq, tk = ops.rope_table(64, Dh, 1e6, Dh ** -0.5 * ops.LOG2E, "cuda"), ops.rope_table(64, Dh, 1e6, 1.0, "cuda")
    pos = torch.tensor([0, 7, 39, 12, 3], dtype=torch.int32, device="cuda")
    cache = torch.full((B, cap, 2 * dkv), 7.0, dtype=torch.float16, device="cuda")
    q = torch.empty((B, dq), dtype=torch.float16, device="cuda")
    ops.rope_qkv_rows(x, tq, tk, pos, Hq, Hkv, Dh, q, cache)
    q_ref = ops.rope_f32in(x[:, :dq], tq, Hq, Dh, torch.empty_like(q), 0, 0, pos)
    kv_ref = ops.rope_f32in(x[:, dq:], tk, Hkv, Dh, torch.empty((B, 2 * dkv), dtype=torch.float16, device="cuda"), dkv, 0, pos)
    assert torch.equal(q.view(torch.int16), q_ref.view(torch.int16))
    want = torch.full_like(cache, 7.0)
    for b in range(B):
        want[b, int(pos[b])] = kv_ref[b]
    assert torch.equal(cache.view(torch.int16), want.view(torch.int16))
    # strided cache view (a slice of a wider buffer) and a position past the cache: clamped to the last row, nothing written outside
    wide = torch.zeros((B, cap, 2 * dkv + 64), dtype=torch.float16, device="cuda")
    ops.rope_qkv_rows(x, tq, tk, torch.tensor([1, 2, 3, 4, 1000], dtype=torch.int32, device="cuda"), Hq, Hkv, Dh, q, wide[:, :, :2 * dkv])
    assert float(wide[:, :, 2 * dkv:].abs().max()) == 0.0 and float(wide[4, cap - 1, :dkv].abs().max()) > 0
