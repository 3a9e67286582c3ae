"""GPU: MFMA GEMM + norms through the C ABI against a plain PyTorch fp32 reference of the same op."""
import os

import pytest
import torch

from streamchat_amd import ops

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, device="cuda", generator=g) * scale).half()


def test_gemm_transpose_detecting():
    """A = I with an ASYMMETRIC W catches swapped rows/cols in the MFMA C/D mapping."""
    K = N = 128
    A = torch.eye(K, device="cuda").half()
    W = (torch.arange(N * K, device="cuda").reshape(N, K) % 97).half()           # W[n,k] != W[k,n]
    out = ops.gemm(A, W)
    assert torch.equal(out, W.t().contiguous())


@pytest.mark.parametrize("M,N,K", [(577, 1024, 1024), (1, 128, 64), (130, 3072, 1024), (1154, 4096, 1024), (300, 1024, 4096),
                                    (576, 1024, 640), (257, 384, 1536), (64, 3584, 3584)])
@pytest.mark.parametrize("epi", ["none", "quick_gelu", "gelu"])
def test_gemm_vs_torch_fp32(M, N, K, epi):
    a, w, b = _rand((M, K), 1), _rand((N, K), 2, K ** -0.5), _rand((N,), 3)
    r = _rand((M, N), 4)
    out = ops.gemm(a, w, b, r, epi)
    ref = a.float() @ w.float().t() + b.float()
    if epi == "quick_gelu":
        ref = ref * torch.sigmoid(1.702 * ref)
    elif epi == "gelu":
        ref = torch.nn.functional.gelu(ref)
    ref = ref + r.float()
    # fp16 output rounding (2^-11 relative) + fp32 accumulation-order noise
    torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)


def test_gemm_operand_larger_than_2_gib():
    """A of 2.2 GiB (the 512-frame ViT batches are this large): tiles address their rows through per-tile buffer resources, so
    32-bit offsets never see the whole operand.  Checked on the first and last rows against torch on the same slices."""
    M, N, K = 270000, 256, 4096
    g = torch.Generator(device="cuda").manual_seed(5)
    a = (torch.rand((M, K), generator=g, device="cuda") - 0.5).half()
    w, b = _rand((N, K), 21, K ** -0.5), _rand((N,), 22)
    out = ops.gemm(a, w, b, None, "quick_gelu")
    for sl in (slice(0, 300), slice(M - 300, M)):
        ref = a[sl].float() @ w.float().t() + b.float()
        ref = ref * torch.sigmoid(1.702 * ref)
        torch.testing.assert_close(out[sl].float(), ref, rtol=2e-3, atol=2e-3)


def test_gemm_unaligned_bias_takes_the_fallback_kernel():
    """a bias view that is not 16-byte aligned cannot be fetched by the DMA of the 4-wave kernel: the dispatcher has to fall back"""
    M, N, K = 1300, 512, 1024
    a, w = _rand((M, K), 31), _rand((N, K), 32, K ** -0.5)
    b = _rand((N + 4,), 33)[4:]                                                    # 8-byte offset: legal for the ABI, not for the DMA
    out = ops.gemm(a, w, b, None, "none")
    torch.testing.assert_close(out.float(), a.float() @ w.float().t() + b.float(), rtol=2e-3, atol=2e-3)


def test_gemm_strided_a_and_f32_out():
    a_full = _rand((200, 3072), 5)
    a = a_full[:, 1024:2048]                                                       # row-strided view, lda = 3072
    w = _rand((256, 1024), 6, 1 / 32)
    out = ops.gemm(a, w, out_f32=True)
    torch.testing.assert_close(out, a.float() @ w.float().t(), rtol=1e-4, atol=1e-4)


def test_gemm_rejects_bad_shapes():
    from streamchat_amd._lib import StreamChatHipError
    with pytest.raises(StreamChatHipError):
        ops.gemm(_rand((8, 100), 1), _rand((128, 100), 2))                        # K % 64 != 0 -> loud error, no fallback


@pytest.mark.parametrize("rows,cols", [(577, 1024), (5, 384), (33, 3584), (1000, 4096), (7, 64)])
def test_layernorm_rmsnorm(rows, cols):
    x, g, b = _rand((rows, cols), 1, 3.0), _rand((cols,), 2), _rand((cols,), 3)
    y = ops.layernorm(x, g, b, 1e-5)
    ref = torch.nn.functional.layer_norm(x.float(), (cols,), g.float(), b.float(), 1e-5)
    torch.testing.assert_close(y.float(), ref, rtol=2e-3, atol=2e-3)
    y = ops.rmsnorm(x, g, 1e-6)
    xf = x.float()
    ref = g.float() * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).half().float()
    torch.testing.assert_close(y.float(), ref, rtol=2e-3, atol=2e-3)


def _attn_ref(q, k, v, Hq, Hkv, Dh, scale, causal, kv_len=None):
    B, Sq, _ = q.shape
    Skv = k.shape[1]
    qf = q.float().view(B, Sq, Hq, Dh).transpose(1, 2)
    kf = k.float().view(B, Skv, Hkv, Dh).transpose(1, 2).repeat_interleave(Hq // Hkv, dim=1)
    vf = v.float().view(B, Skv, Hkv, Dh).transpose(1, 2).repeat_interleave(Hq // Hkv, dim=1)
    s = qf @ kf.transpose(-1, -2) * scale
    if causal:
        i = torch.arange(Sq, device=q.device)[:, None] + (Skv - Sq)
        j = torch.arange(Skv, device=q.device)[None, :]
        s = s.masked_fill(j > i, float("-inf"))
    if kv_len is not None:
        j = torch.arange(Skv, device=q.device)[None, None, None, :]
        s = s.masked_fill(j >= kv_len.view(B, 1, 1, 1), float("-inf"))
    return (torch.softmax(s, -1) @ vf).transpose(1, 2).reshape(B, Sq, Hq * Dh)


@pytest.mark.parametrize("B,Sq,Skv,Hq,Hkv,Dh,causal", [
    (3, 577, 577, 16, 16, 64, False),       # ViT-L block
    (2, 64, 64, 2, 2, 64, False), (1, 1, 130, 4, 4, 64, False), (2, 100, 100, 4, 4, 64, True),
    (1, 300, 300, 28, 4, 128, True),        # Qwen2 GQA prefill
    (1, 17, 200, 8, 2, 128, True),          # chunked prefill: queries aligned to the end of the keys
    (2, 129, 129, 4, 4, 128, False)])
def test_attention_vs_torch_fp32(B, Sq, Skv, Hq, Hkv, Dh, causal):
    q, k, v = _rand((B, Sq, Hq * Dh), 1), _rand((B, Skv, Hkv * Dh), 2), _rand((B, Skv, Hkv * Dh), 3)
    out = ops.attention(q, k, v, Hq, Hkv, Dh, Dh ** -0.5, causal)
    ref = _attn_ref(q, k, v, Hq, Hkv, Dh, Dh ** -0.5, causal)
    torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)


def test_attention_fused_qkv_views_and_padding_mask():
    B, S, H, Dh = 3, 50, 4, 64
    qkv = _rand((B, S, 3 * H * Dh), 7)
    q, k, v = qkv[..., : H * Dh], qkv[..., H * Dh: 2 * H * Dh], qkv[..., 2 * H * Dh:]
    kv_len = torch.tensor([50, 7, 33], device="cuda", dtype=torch.int32)
    out = ops.attention(q, k, v, H, H, Dh, 0.125, False, kv_len)
    ref = _attn_ref(q, k, v, H, H, Dh, 0.125, False, kv_len)
    torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)


def test_attention_online_softmax_rescale_branch():
    """One key spikes against every query late in the sequence so the running max jumps (guide rule 26)."""
    B, S, H, Dh = 1, 256, 2, 64
    q, k, v = _rand((B, S, H * Dh), 1), _rand((B, S, H * Dh), 2), _rand((B, S, H * Dh), 3)
    k[0, 200] = q[0, 10] * 4                      # large q.k for kv row 200 (4th tile)
    out = ops.attention(q, k, v, H, H, Dh, 0.125, False)
    ref = _attn_ref(q, k, v, H, H, Dh, 0.125, False)
    torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("B,Sq,Skv,Hq,Hkv,causal,pad", [(1, 2304, 2304, 4, 2, True, False), (2, 2050, 2500, 2, 2, False, True), (1, 2048, 3000, 7, 1, True, False)])
def test_attention_long_dh128_three_qblock_path(B, Sq, Skv, Hq, Hkv, causal, pad):
    """Sq >= 2048 at Dh = 128 takes the 48-queries-per-wave / half-tile variant (192-query blocks): ragged last block, GQA, causal
    with Skv > Sq, padding mask."""
    q, k, v = _rand((B, Sq, Hq * 128), 31), _rand((B, Skv, Hkv * 128), 32), _rand((B, Skv, Hkv * 128), 33)
    kv_len = torch.tensor([Skv - 37, 1500][:B], device="cuda", dtype=torch.int32) if pad else None
    out = ops.attention(q, k, v, Hq, Hkv, 128, 0.09, causal, kv_len)
    ref = _attn_ref(q, k, v, Hq, Hkv, 128, 0.09, causal, kv_len)
    torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("Dh,causal", [(64, False), (128, True)])
def test_attention_late_huge_score_forces_the_exact_pass(Dh, causal):
    """The steady-state loop keeps the first tile's row max as the softmax reference and takes no row max afterwards; a later score
    more than 2^16 above it overflows fp16 P.  That must be detected (inf / NaN in O or l at the end of the block) and the block
    redone with the exact online softmax: here keys in the 5th tile score ~e^60 above everything before them."""
    B, S, H = 1, (2112 if Dh == 128 else 512), 2                                  # Dh = 128: long enough for the 3-q-block variant
    q, k, v = _rand((B, S, H * Dh), 5), _rand((B, S, H * Dh), 6), _rand((B, S, H * Dh), 7)
    q[0, 300:] *= 4.0
    k[0, 290] = q[0, 400] * 3.0                    # q_i . k_290 ~ 12 |q|^2 for the late queries
    scale = 0.5
    out = ops.attention(q, k, v, H, H, Dh, scale, causal)
    ref = _attn_ref(q, k, v, H, H, Dh, scale, causal)
    assert torch.isfinite(out).all()
    s_max = (q[0, 400, :Dh].float() @ k[0, 290, :Dh].float()) * scale
    s_first = ((q[0, 400, :Dh].float() @ k[0, :64, :Dh].float().t()) * scale).max()
    assert (s_max - s_first) * 1.4427 > 20                                          # the test really is beyond fp16 range
    torch.testing.assert_close(out.float(), ref, rtol=3e-3, atol=3e-3)


def test_attention_randomised_shapes_and_padding():
    """Seeded sweep over head dims, ragged query / key counts, GQA groups, causal offsets and per-batch key lengths at and around
    multiples of 16 and 64 (the general-tile paths: dead key blocks skipped, dead waves of the last q-block, row sums on the matrix
    pipe at head dim <= 64), NaN garbage behind the valid keys; against fp32 torch.  Rows without any visible key are not compared."""
    import random
    rnd = random.Random(11)
    for case in range(28):
        Dh = rnd.choice([32, 64, 64, 128])
        B, Hkv = rnd.choice([1, 2, 3]), rnd.choice([1, 2])
        Hq, causal = Hkv * rnd.choice([1, 2, 7]), rnd.random() < 0.5
        Sq = rnd.choice([1, 15, 16, 17, 63, 64, 65, 127, 129, 300, 577, 640, rnd.randint(2, 700)])
        Skv = Sq + rnd.choice([0, 0, 1, 63, 200]) if causal else rnd.choice([1, 16, 33, 64, 65, 128, 577, rnd.randint(1, 900)])
        q, k, v = _rand((B, Sq, Hq * Dh), 500 + case), _rand((B, Skv, Hkv * Dh), 600 + case), _rand((B, Skv, Hkv * Dh), 700 + case)
        kv_len = None
        if rnd.random() < 0.6:
            lens = [max(1, min(Skv, rnd.choice([Skv, Skv - 1, Skv - 15, Skv - 16, Skv - 17, Skv - 64, 1, 16, 17, 64, 65]))) for _ in range(B)]
            if causal:
                lens = [max(n, Skv - Sq + 1) for n in lens]                          # every query keeps at least its first key
            kv_len = torch.tensor(lens, device="cuda", dtype=torch.int32)
        kk, vv = k.clone(), v.clone()
        if kv_len is not None:
            for b in range(B):
                k[b, lens[b]:] = float("nan"); v[b, lens[b]:] = float("nan"); kk[b, lens[b]:] = 0; vv[b, lens[b]:] = 0
        out = ops.attention(q, k, v, Hq, Hkv, Dh, 0.11, causal, kv_len)
        ref = _attn_ref(q, kk, vv, Hq, Hkv, Dh, 0.11, causal, kv_len)
        ok = torch.isfinite(ref).all(-1)
        assert torch.isfinite(out.float()[ok]).all(), (case, Dh, B, Sq, Skv, Hq, Hkv, causal)
        torch.testing.assert_close(out.float()[ok], ref[ok], rtol=3e-3, atol=3e-3, msg=lambda m: f"case {case}: Dh={Dh} B={B} Sq={Sq} Skv={Skv} Hq={Hq} Hkv={Hkv} causal={causal} kv_len={None if kv_len is None else kv_len.tolist()}\n{m}")


@pytest.mark.parametrize("N,K", [(3584, 3584), (1024, 3584), (152064, 3584), (3584, 18944), (130, 264)])
def test_gemv_vs_torch_fp32(N, K):
    w, x, b, r = _rand((N, K), 1, K ** -0.5), _rand((K,), 2), _rand((N,), 3), _rand((N,), 4)
    y = ops.gemv(w, x, b, r)
    ref = w.float() @ x.float() + b.float() + r.float()
    torch.testing.assert_close(y.float(), ref, rtol=2e-3, atol=2e-3)
    y32 = ops.gemv(w, x, out_f32=True)
    torch.testing.assert_close(y32, w.float() @ x.float(), rtol=1e-3, atol=1e-3)


def test_gemv_fused_rmsnorm_bits_do_not_depend_on_the_workgroup_size():
    """block_rmsnorm_to_lds reduces in the order of 256 VIRTUAL threads whatever the workgroup size, so that kernels launched with
    different wave counts (k_gemv: 4 waves, the down projection: 7, k_decode_qkv: 4; SC_GEMV_WPB_<kind> for experiments) normalise x to the
    same bits - what keeps the eager decode step and the captured graph bit-identical.  The knob is read once per process: children."""
    import hashlib
    import subprocess
    import sys
    code = ("import torch, hashlib, tests.test_gpu_dense as T\n"
            "from streamchat_amd import ops\n"
            "w, x, g = T._rand((3584, 3584), 1, 3584 ** -0.5), T._rand((3584,), 2), (1 + T._rand((3584,), 3, 0.1).float()).half()\n"
            "y = ops.gemv(w, x, None, rms_gamma=g, rms_eps=1e-6, out_f32=True)\n"
            "print('HASH', hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hashes = []
    for wpb in ("2", "3", "4", "7"):
        r = subprocess.run([sys.executable, "-c", code], env={**os.environ, "SC_GEMV_WPB_1": wpb}, cwd=root, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        hashes.append([l.split()[1] for l in r.stdout.splitlines() if l.startswith("HASH")][0])
    assert len(set(hashes)) == 1, hashes
    w, x, g = _rand((3584, 3584), 1, 3584 ** -0.5), _rand((3584,), 2), (1 + _rand((3584,), 3, 0.1).float()).half()
    y = ops.gemv(w, x, None, rms_gamma=g, rms_eps=1e-6, out_f32=True)
    assert hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest() == hashes[0]
    xn = (g.double() * (x.double() * torch.rsqrt((x.double() ** 2).mean() + 1e-6)).half().double()).half().double()
    torch.testing.assert_close(y.double(), w.double() @ xn, rtol=1e-3, atol=1e-3)


def test_decode_advance_bookkeeping():
    nxt = torch.tensor([4242], dtype=torch.int64, device="cuda")
    ring = torch.zeros(8, dtype=torch.int64, device="cuda")
    cnt = torch.tensor([3], dtype=torch.int64, device="cuda")
    tok, pos, ln, npv = (torch.tensor([v], dtype=torch.int32, device="cuda") for v in (7, 100, 101, 4))
    ops.decode_advance(nxt, ring, cnt, tok, pos, ln, npv)
    assert ring.tolist() == [0, 0, 0, 4242, 0, 0, 0, 0] and int(cnt) == 4 and (int(tok), int(pos), int(ln), int(npv)) == (4242, 101, 102, 5)


def test_gemv_swiglu():
    K, I = 256, 384
    x = _rand((K,), 1)
    wg, wu = _rand((I, K), 2, 1 / 16), _rand((I, K), 3, 1 / 16)
    wgu = torch.cat([wg.view(I // 2, 2, K), wu.view(I // 2, 2, K)], 1).reshape(2 * I, K).contiguous()
    y = ops.gemv(wgu, x, epilogue="swiglu")
    ref = torch.nn.functional.silu(wg.float() @ x.float()) * (wu.float() @ x.float())
    torch.testing.assert_close(y.float(), ref, rtol=3e-3, atol=3e-3)


@pytest.mark.parametrize("Sq,Skv,H,Dh,ns", [(7, 3000, 4, 128, 16), (1, 500, 2, 64, 4), (2, 200, 4, 64, 7), (7, 70, 4, 128, 2)])
def test_attention_split_kv_equals_unsplit(Sq, Skv, H, Dh, ns):
    q, k, v = _rand((1, Sq, H * Dh), 1), _rand((1, Skv, H * Dh), 2), _rand((1, Skv, H * Dh), 3)
    a = ops.attention(q, k, v, H, H, Dh, Dh ** -0.5, False, nsplit=ns)
    ref = _attn_ref(q, k, v, H, H, Dh, Dh ** -0.5, False)
    torch.testing.assert_close(a.float(), ref, rtol=2e-3, atol=2e-3)


def test_attention_decode_streaming_kernel_randomised():
    """Dh = 128, <= 16 query rows, split-KV, non-causal = a decode step: k_attn_decode (every wave streams its own run of 32-row chunks, K
    straight into MFMA operands, V through a per-wave LDS ring, in-block merge of the four waves, then k_attn_combine).  Seeded sweep
    over key counts around the chunk / split / wave boundaries (fewer chunks than waves, empty splits), split factors, batches with
    per-sequence lengths, GQA groups and NaN garbage behind the valid keys; against fp32 torch."""
    import random
    rnd = random.Random(5)
    assert ops.attention_variant(128, 7, 16) == 3 and ops.attention_variant(128, 17, 16) == 0 and ops.attention_variant(64, 7, 16) == 0
    for case in range(24):
        B, Hkv = rnd.choice([1, 1, 2, 3]), rnd.choice([1, 2, 4])
        Hq = Hkv * rnd.choice([1, 1, 2])
        Sq = rnd.choice([1, 2, 7, 8, 15, 16])
        Skv = rnd.choice([1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 257, 1000, 4097, rnd.randint(1, 9000)])
        ns = rnd.choice([2, 3, 7, 16, 64, 128])
        q, k, v = _rand((B, Sq, Hq * 128), 900 + case), _rand((B, Skv, Hkv * 128), 1000 + case), _rand((B, Skv, Hkv * 128), 1100 + case)
        kv_len, kk, vv = None, k.clone(), v.clone()
        if rnd.random() < 0.6:
            lens = [max(1, min(Skv, rnd.choice([Skv, Skv - 1, Skv - 31, Skv - 32, Skv - 33, 1, 32, 33, Skv // 2]))) for _ in range(B)]
            kv_len = torch.tensor(lens, device="cuda", dtype=torch.int32)
            for b in range(B):
                k[b, lens[b]:] = float("nan"); v[b, lens[b]:] = float("nan"); kk[b, lens[b]:] = 0; vv[b, lens[b]:] = 0
        out = ops.attention(q, k, v, Hq, Hkv, 128, 0.09, False, kv_len, nsplit=ns)
        ref = _attn_ref(q, kk, vv, Hq, Hkv, 128, 0.09, False, kv_len)
        assert torch.isfinite(out.float()).all(), (case, B, Sq, Skv, Hq, Hkv, ns)
        torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3, msg=lambda m: f"case {case}: B={B} Sq={Sq} Skv={Skv} Hq={Hq} Hkv={Hkv} ns={ns} kv_len={None if kv_len is None else kv_len.tolist()}\n{m}")


def test_attention_decode_streaming_kernel_long_context_packed_heads():
    """the call llm._decode_one / DecodeGraph make: the G = 7 query heads of a KV group as 7 query ROWS of that KV head (addressing only),
    49 157 keys in a cache with capacity beyond them, 128 splits"""
    S, cap, G, Hkv, Dh = 49157, 49152 + 64, 7, 4, 128
    q = _rand((1, G * Hkv * Dh), 1)
    cache = _rand((cap, 2 * Hkv * Dh), 2, 0.5)
    cache[S:] = float("nan")
    ck = cache[:cap]
    kl = torch.tensor([S], device="cuda", dtype=torch.int32)
    qv = q.as_strided((1, G, Dh), (G * Hkv * Dh, Dh, 1))
    out = ops.attention(qv, ck[:, :Hkv * Dh].unsqueeze(0), ck[:, Hkv * Dh:].unsqueeze(0), Hkv, Hkv, Dh, Dh ** -0.5, causal=False, kv_len=kl, nsplit=128,
                        q_head_stride=G * Dh, o_head_stride=G * Dh, out_ld=Dh).view(Hkv, G, Dh)
    kf, vf = cache[:S, :Hkv * Dh].float().view(S, Hkv, Dh), cache[:S, Hkv * Dh:].float().view(S, Hkv, Dh)
    qf = q.float().view(Hkv, G, Dh)
    p = torch.softmax(torch.einsum("hgd,shd->hgs", qf, kf) * Dh ** -0.5, -1)
    ref = torch.einsum("hgs,shd->hgd", p, vf)
    torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("M,N,K,epi", [(5000, 4096, 1024, "quick_gelu"), (9000, 2048, 192, "none"), (4099, 8192, 512, "swiglu"), (16640, 1024, 4096, "gelu")])
def test_gemm_many_tiles_per_workgroup(M, N, K, epi):
    """more 256x256 tiles than CUs -> the persistent walk (a workgroup computes several tiles, the K-step ring runs across the tile
    boundary); ragged last tile row (M % 256 != 0), bias + residual, every epilogue.  K = 192 has 6 K-steps (short tiles)."""
    a, w, b = _rand((M, K), 11), _rand((N, K), 12, K ** -0.5), _rand((N,), 13)
    if epi == "swiglu":
        out = ops.gemm(a, w, b, None, epi)
        z = (a.float() @ w.float().t() + b.float()).view(M, N // 4, 2, 2)            # interleaved (g0, g1, u0, u1) quads
        ref = (torch.nn.functional.silu(z[:, :, 0]) * z[:, :, 1]).reshape(M, N // 2)
    else:
        r = _rand((M, N), 14)
        out = ops.gemm(a, w, b, r, epi)
        ref = a.float() @ w.float().t() + b.float()
        ref = ref * torch.sigmoid(1.702 * ref) if epi == "quick_gelu" else (torch.nn.functional.gelu(ref) if epi == "gelu" else ref)
        ref = ref + r.float()
    torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("M", [1, 8, 16, 17, 26, 32])
@pytest.mark.parametrize("N,K,epi", [(3584, 3584, "none"), (1024, 1536, "gelu"), (512, 18944, "none"), (768, 256, "swiglu"), (256, 384, "quick_gelu")])
def test_gemm_skinny_rows(M, N, K, epi):
    """M <= 32: the weight-streaming kernel (W read once from global memory into MFMA operands, K split over the 4 waves of a
    workgroup); every epilogue, bias + residual, fp16 and fp32 outputs."""
    a, w, b = _rand((M, K), 21), _rand((N, K), 22, K ** -0.5), _rand((N,), 23)
    z = a.float() @ w.float().t() + b.float()
    if epi == "swiglu":
        out = ops.gemm(a, w, b, None, epi)
        q = z.view(M, N // 4, 2, 2)
        ref = (torch.nn.functional.silu(q[:, :, 0]) * q[:, :, 1]).reshape(M, N // 2)
        torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)
        return
    r = _rand((M, N), 24)
    ref = z * torch.sigmoid(1.702 * z) if epi == "quick_gelu" else (torch.nn.functional.gelu(z) if epi == "gelu" else z)
    torch.testing.assert_close(ops.gemm(a, w, b, r, epi).float(), ref + r.float(), rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(ops.gemm(a, w, b, None, epi, out_f32=True), ref, rtol=1e-3, atol=1e-3)


def test_attention_full_size_llm_prefill_rows_vs_torch():
    """BASELINE C3 size: causal GQA 28/4 heads x 128 over 49 152 tokens (the LongVA-7B prefill shape).  Size-independent checks:
    (1) V = 1 gives exactly-normalised rows (softmax weights sum to one); (2) 24 probe rows spread over the sequence (first / last
    rows of blocks, the final row) against a per-row fp32 torch softmax over all their keys."""
    S, Hq, Hkv, Dh = 49152, 28, 4, 128
    g = torch.Generator(device="cuda").manual_seed(49)
    q = torch.randn(1, S, Hq * Dh, device="cuda", generator=g).half()
    k = torch.randn(1, S, Hkv * Dh, device="cuda", generator=g).half()
    v = torch.randn(1, S, Hkv * Dh, device="cuda", generator=g).half()
    scale = Dh ** -0.5
    out = ops.attention(q, k, v, Hq, Hkv, Dh, scale, True)
    rows = [0, 1, 63, 64, 191, 192, 193, 4095, 4096, 20000, 24575, 24576, 30001, 40000, 48959, 48960, 48961, 49000, 49087, 49088, 49150, 49151, 777, 12345]
    for h in (0, 6, 7, 27):
        kh, vh = k[0, :, (h // 7) * Dh:(h // 7 + 1) * Dh].float(), v[0, :, (h // 7) * Dh:(h // 7 + 1) * Dh].float()
        for r in rows:
            s = (q[0, r, h * Dh:(h + 1) * Dh].float() @ kh[: r + 1].t()) * scale
            ref = torch.softmax(s, -1) @ vh[: r + 1]
            torch.testing.assert_close(out[0, r, h * Dh:(h + 1) * Dh].float(), ref, rtol=3e-3, atol=3e-3)
    ones = ops.attention(q, k, torch.ones_like(v), Hq, Hkv, Dh, scale, True)
    assert (ones.float() - 1.0).abs().max().item() < 2e-3


def test_gemm_full_size_llm_rows_vs_torch():
    """BASELINE C3 size: the 48 994-row Qwen2-7B projections (q: 3584 -> 3584, down: 18944 -> 3584, gate_up SwiGLU: 3584 -> 2 x 18944);
    128 probe rows (incl. the ragged last tile) against fp32 torch."""
    M = 48994
    rows = torch.cat([torch.arange(0, 32), torch.arange(24500, 24532), torch.arange(48896, 48928), torch.arange(M - 32, M)]).cuda()
    for (N, K, epi) in [(3584, 3584, "none"), (3584, 18944, "none"), (37888, 3584, "swiglu")]:
        a, w = _rand((M, K), 41), _rand((N, K), 42, K ** -0.5)
        out = ops.gemm(a, w, None, None, epi)
        z = a[rows].float() @ w.float().t()
        if epi == "swiglu":
            q4 = z.view(rows.numel(), N // 4, 2, 2)
            z = (torch.nn.functional.silu(q4[:, :, 0]) * q4[:, :, 1]).reshape(rows.numel(), N // 2)
        torch.testing.assert_close(out[rows].float(), z, rtol=2e-3, atol=2e-3)
        del a, w, out


def test_gemm_randomised_shape_stress():
    """k_gemm_fat relies on hand-placed hazard guards that hipcc cannot see; a toolchain regression would show
    up as rare wrong tiles.  60 seeded random (M, N, K, epilogue, bias, residual) combinations across all four kernels' dispatch
    regions (ragged M, K % 128 != 0 fallbacks, many tiles per workgroup, single tiles), each against fp32 torch, twice in a row
    (the second launch runs with warm caches and a different tile-to-CU timing)."""
    import random
    rnd = random.Random(20260928)
    epis = ["none", "quick_gelu", "gelu", "swiglu"]
    for case in range(60):
        K = rnd.choice([64, 128, 192, 256, 384, 512, 1024, 1536, 3584, 4096])
        N = rnd.choice([128, 256, 384, 512, 1024, 1280, 3072, 3584, 4096])
        M = rnd.choice([1, 7, 33, 200, 577, 1023, 1024, 1025, 2500, 4099, 9000, 20000, 70000])
        if M * N > 120e6 or M * K > 160e6:
            M = 4099
        epi = rnd.choice(epis)
        use_bias, use_res = rnd.random() < 0.7, (rnd.random() < 0.5 and epi != "swiglu")
        a, w = _rand((M, K), 1000 + case), _rand((N, K), 2000 + case, K ** -0.5)
        b = _rand((N,), 3000 + case) if use_bias else None
        z = a.float() @ w.float().t() + (b.float() if use_bias else 0)
        if epi == "swiglu":
            q = z.view(M, N // 4, 2, 2)
            ref = (torch.nn.functional.silu(q[:, :, 0]) * q[:, :, 1]).reshape(M, N // 2)
            r = None
        else:
            ref = z * torch.sigmoid(1.702 * z) if epi == "quick_gelu" else (torch.nn.functional.gelu(z) if epi == "gelu" else z)
            r = _rand((M, N), 4000 + case) if use_res else None
            if use_res:
                ref = ref + r.float()
        for rep in range(2):
            out = ops.gemm(a, w, b, r, epi)
            err = (out.float() - ref).abs()
            tol = 2e-3 + 2e-3 * ref.abs()
            bad = int((err > tol).sum())
            assert bad == 0, (case, rep, M, N, K, epi, use_bias, use_res, bad, float(err.max()))


def test_gemm_residual_with_a_padded_leading_dimension():
    """The residual rows of the hand-scheduled GEMM arrive by row-wise 16-byte loads and, for the first 16-row tile of every wave, by LDS
    DMA: both address rows through `ldr`.  Residuals that are column slices of a wider buffer (ldr = N + 8 / N + 32: still 16-byte rows, row offsets with every low address bit in use; N + 4:
    8-byte rows, dispatched to the other kernel; N + 256) must give the same result as a contiguous copy - bit for bit."""
    M, N, K = 2500, 1024, 1024
    a, w, b = _rand((M, K), 71), _rand((N, K), 72, K ** -0.5), _rand((N,), 73)
    for pad in (8, 4, 32, 256):
        wide = _rand((M, N + pad), 74 + pad)
        r_view = wide[:, :N]
        ref = ops.gemm(a, w, b, r_view.contiguous(), "none")
        out = ops.gemm(a, w, b, r_view, "none")
        assert torch.equal(out, ref), (pad, float((out.float() - ref.float()).abs().max()))
        z = a.float() @ w.float().t() + b.float() + r_view.float()
        torch.testing.assert_close(out.float(), z, rtol=2e-3, atol=2e-3)
