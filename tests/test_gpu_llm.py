"""GPU: HIP Qwen2 (prefill + KV-cache decode), the `<image>` splice and generate_with_image_embedding against the golden
vectors of the real HF Qwen2ForCausalLM (tiny config) / the reference's own splice function, and the fp32 PyTorch
restatement at Qwen2-7B widths.  Tolerance on logits: fp16 storage + fp32 accumulate -> tests/_tol.py (max 4e-3 of max|logit|, rms 3e-3, per-row cosine 0.9999);
greedy token ids must match exactly."""
import os

import numpy as np
import pytest
import torch

from tests._tol import assert_close_fp16

from oracle import torch_ref as R
from streamchat_amd import llm as LM, ops

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _tiny():
    d = np.load(os.path.join(G, "qwen2_tiny.npz"))
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("lm.")}
    cfg = LM.Qwen2ConfigLite(hidden=256, layers=2, heads=4, kv_heads=2, intermediate=512, vocab=512, rope_theta=1e6)
    return d, sd, cfg


def test_prefill_logits_vs_hf_golden():
    d, sd, cfg = _tiny()
    lm = LM.Qwen2Model(sd, cfg, max_seq=128)
    emb = torch.from_numpy(d["inputs_embeds"]).cuda().half()
    logits = lm.forward(emb, last_only=False)
    ref = torch.from_numpy(d["logits"]).cuda()
    assert logits.shape == ref.shape and logits.dtype == torch.float32
    assert_close_fp16(logits, ref, what="tiny Qwen2 prefill logits vs HF golden")
    assert lm.cache_len == 37


def test_chunked_prefill_and_decode_match_full_prefill():
    """KV-cache correctness: prefill(20) + prefill(16) + decode(1) == one 37-token prefill (same kernels, same rounding)."""
    d, sd, cfg = _tiny()
    emb = torch.from_numpy(d["inputs_embeds"]).cuda().half()
    a = LM.Qwen2Model(sd, cfg, max_seq=64)
    full = a.forward(emb)
    b = LM.Qwen2Model(sd, cfg, max_seq=64)
    b.forward(emb[:20]); b.forward(emb[20:36])
    last = b.forward(emb[36:37])
    torch.testing.assert_close(last, full, rtol=2e-3, atol=2e-3)
    assert int(last.argmax()) == int(full.argmax())


def test_greedy_generation_matches_hf():
    d, sd, cfg = _tiny()
    model = LM.LlavaQwenForCausalLM(LM.Qwen2Model(sd, cfg, max_seq=64))
    # drive generate through the splice: ids = [<image>] only, image_embeddings = the golden inputs_embeds
    out = model.generate_with_image_embedding(torch.tensor([[-200]]), image_embeddings=[torch.from_numpy(d["inputs_embeds"]).cuda().half()],
                                              modalities=["video"], do_sample=False, max_new_tokens=8, use_cache=False)
    assert out.shape == (1, 8)
    assert out[0].cpu().tolist() == d["greedy"].tolist()


def test_splice_matches_reference():
    d = np.load(os.path.join(G, "splice.npz"))
    table = torch.from_numpy(d["table"]).cuda().half()
    feats = torch.from_numpy(d["feats"]).cuda().half()
    for k in ("middle", "start", "none", "truncated", "end"):
        mx = int(d[k + ".max_len"])
        out, _ = LM.splice_image_embeddings(torch.from_numpy(d[k + ".ids"]), table, [feats], None if mx < 0 else mx)
        ref = torch.from_numpy(d[k + ".embeds"]).cuda().half()          # fp16 rounding of the fp32 golden rows is exact data movement
        assert torch.equal(out, ref), k
        # the image block handed over as [short | retrieved ...] pieces (views of the feature bank) splices to the same rows as their cat
        n = feats.shape[0]
        pieces = [feats[:1], feats[1:n // 2], feats[n // 2:]]
        out2, _ = LM.splice_image_embeddings(torch.from_numpy(d[k + ".ids"]), table, [pieces], None if mx < 0 else mx)
        assert torch.equal(out2, ref), k


def test_rope_and_swiglu_kernels_vs_torch():
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(50, 4 * 128, device="cuda", generator=g).half()
    y = ops.rope_(x.clone(), 4, 128, 1e6, pos0=1000)
    pos = torch.arange(1000, 1050, device="cuda", dtype=torch.float32)
    inv = 1.0 / (1e6 ** (torch.arange(0, 128, 2, device="cuda", dtype=torch.float32) / 128))
    fr = torch.outer(pos, inv)
    cos, sin = torch.cat([fr, fr], -1).cos()[:, None], torch.cat([fr, fr], -1).sin()[:, None]
    xf = x.float().view(50, 4, 128)
    ref = xf * cos + torch.cat([-xf[..., 64:], xf[..., :64]], -1) * sin
    torch.testing.assert_close(y.float().view(50, 4, 128), ref, rtol=4e-3, atol=4e-3)
    a = torch.randn(70, 256, device="cuda", generator=g).half()
    wg, wu = (torch.randn(384, 256, device="cuda", generator=g) / 16).half(), (torch.randn(384, 256, device="cuda", generator=g) / 16).half()
    wgu = torch.cat([wg.view(192, 2, 256), wu.view(192, 2, 256)], 1).reshape(768, 256).contiguous()
    out = ops.gemm(a, wgu, epilogue="swiglu")
    ref = torch.nn.functional.silu(a.float() @ wg.float().t()) * (a.float() @ wu.float().t())
    torch.testing.assert_close(out.float(), ref, rtol=3e-3, atol=3e-3)


def test_qwen2_7b_width_layers_vs_torch_fp32():
    """2 layers at the real Qwen2-7B widths (3584 / 28 q-heads / 4 kv-heads / 18944) on 300 tokens vs fp32 PyTorch."""
    cfg = LM.Qwen2ConfigLite(**dict(LM.QWEN2_7B, layers=2, vocab=1024))
    sd = LM.random_qwen2_state_dict(cfg, seed=5)
    lm = LM.Qwen2Model(sd, cfg, max_seq=512)
    emb = (torch.randn(300, 3584, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) * 0.5).half()
    logits = lm.forward(emb, last_only=False)
    ref = R.qwen2_logits({k: v.float() for k, v in sd.items()}, emb.float(), heads=28, kv_heads=4, layers=2, head_dim=128)
    assert_close_fp16(logits, ref, what="2 layers at Qwen2-7B widths, 300 tokens vs fp32 torch_ref")


def test_decode_graph_matches_eager_greedy():
    """hipGraph-replayed decode (device-resident position / token) == the eager KV-cache decode == HF greedy ids."""
    d, sd, cfg = _tiny()
    emb = torch.from_numpy(d["inputs_embeds"]).cuda().half()
    lm = LM.Qwen2Model(sd, cfg, max_seq=128)
    logits = lm.forward(emb)
    first = int(logits.argmax())
    g = LM.DecodeGraph(lm, max_new_tokens=16, nsplit=2)
    g.start(first)
    rest = g.run(7)
    assert [first] + rest == d["greedy"].tolist()
    assert lm.cache_len == 37 + 7
    # a second run continues from the advanced cache without re-capturing
    g.start(rest[-1])
    more = g.run(3)
    lm2 = LM.Qwen2Model(sd, cfg, max_seq=128)
    lg = lm2.forward(emb)
    seq = []
    tok = int(lg.argmax())
    for _ in range(11):
        seq.append(tok)
        tok = int(lm2.forward(lm2.embed_tokens(torch.tensor([tok], device="cuda"))).argmax())
    assert seq == [first] + rest + more


def test_trimmed_last_layer_gives_same_logits_and_cache():
    d, sd, cfg = _tiny()
    emb = torch.from_numpy(d["inputs_embeds"]).cuda().half()
    a = LM.Qwen2Model(sd, cfg, max_seq=64)
    b = LM.Qwen2Model(sd, cfg, max_seq=64, trim_last_layer=True)
    la, lb = a.forward(emb), b.forward(emb)
    torch.testing.assert_close(lb, la, rtol=2e-3, atol=2e-3)
    assert int(la.argmax()) == int(lb.argmax())
    for ca, cb in zip(a.cache, b.cache):
        assert torch.equal(ca[:37], cb[:37])                                    # K/V of every layer and row are still produced


def test_batched_greedy_decode_equals_one_by_one():
    """BatchDecoder (separate prefill, shared decode steps with M = B GEMMs, one batched attention launch per layer, per-sequence
    kv_len) produces the tokens of B independent batch-1 generations; prompts of different lengths."""
    d, sd, cfg = _tiny()
    emb = torch.from_numpy(d["inputs_embeds"]).cuda().half()
    prompts = [emb, emb[:20], emb[5:30]]
    lm = LM.Qwen2Model(sd, cfg, max_seq=64)
    single = []
    for e in prompts:
        lm.reset_cache()
        logits, toks = lm.forward(e), []
        for _ in range(6):
            toks.append(int(logits.argmax()))
            logits = lm.forward(lm.embed_tokens(torch.tensor([toks[-1]], device="cuda")))
        single.append(toks)
    dec = LM.BatchDecoder(lm, prompts, max_new_tokens=6)
    assert dec.len.tolist() == [37, 20, 25]
    batch = dec.generate(6)
    assert batch == single
    assert single[0] == d["greedy"].tolist()[:6]                                  # and sequence 0 is still the HF golden
    # EOS cuts one sequence, the others run on
    dec2 = LM.BatchDecoder(lm, prompts, max_new_tokens=6)
    eos = single[1][2]
    cut = dec2.generate(6, eos_token_id=eos)
    for got, ref in zip(cut, single):
        assert got == (ref[:ref.index(eos) + 1] if eos in ref else ref)


def test_batch_decoder_leaves_the_models_prompt_state_untouched():
    """BatchDecoder.__init__ prefills its prompts through lm.forward with lm.cache / cache_len / max_seq swapped; forward() also records the
    split-KV factor of "the prompt in the cache" - it belongs to the saved state, or an EAGER decode that continues the earlier prompt would merge its
    split partials in another grouping than a DecodeGraph built for that prompt.  Prefill A (long enough for several splits), build a BatchDecoder over
    other prompts, then decode A eagerly and by graph: same state, bit-identical logits."""
    cfg = LM.Qwen2ConfigLite(hidden=256, layers=2, heads=2, kv_heads=1, intermediate=512, vocab=512)          # head dim 128: the k_attn_decode split rule
    sd = LM.random_qwen2_state_dict(cfg, seed=9, std=0.05)
    g = torch.Generator(device="cuda").manual_seed(1)
    A = (torch.randn(3000, 256, device="cuda", generator=g) * 0.5).half()
    lm = LM.Qwen2Model(sd, cfg, max_seq=4096)
    first = int(lm.forward(A).argmax())
    state = (lm.cache_len, lm._nsplit_prompt, lm.max_seq, [c.data_ptr() for c in lm.cache])
    assert lm._nsplit_prompt == LM.decode_nsplit(128, 3000) and lm._nsplit_prompt > 2
    LM.BatchDecoder(lm, [A[:300], A[:40]], max_new_tokens=4)                       # prompts with another split factor (2)
    assert (lm.cache_len, lm._nsplit_prompt, lm.max_seq, [c.data_ptr() for c in lm.cache]) == state
    dg = LM.DecodeGraph(lm, max_new_tokens=16)
    assert dg.nsplit == lm._nsplit_prompt
    eager = lm.forward(lm.embed_tokens(torch.tensor([first], device="cuda"))).clone()
    lm.cache_len = 3000
    dg.start(first)
    dg.run(1)
    assert torch.equal(dg.logits.view(torch.int32), eager.view(torch.int32))


def test_rope_qk_row_equals_separate_ropes():
    """the fused decode RoPE (query row + K part of the cache row, one launch) is bit-identical to the two separate kernels"""
    g = torch.Generator(device="cuda").manual_seed(3)
    q = torch.randn(1, 28 * 128, device="cuda", generator=g).half()
    cache = torch.randn(300, 2 * 4 * 128, device="cuda", generator=g).half()
    pos = torch.tensor([217], dtype=torch.int32, device="cuda")
    q1, c1 = q.clone(), cache.clone()
    ops.rope_(q1, 28, 128, 1e6, positions=pos)
    ops.rope_row_(c1, pos, 4, 128, 1e6)
    q2, c2 = q.clone(), cache.clone()
    ops.rope_qk_row_(q2, 28, c2, pos, 4, 128, 1e6)
    assert torch.equal(q1, q2) and torch.equal(c1, c2)
    assert torch.equal(c2[:217], cache[:217]) and torch.equal(c2[218:], cache[218:]) and torch.equal(c2[217, 512:], cache[217, 512:])   # V part and other rows untouched


def test_batched_sampling_is_reproducible_and_plausible():
    """do_sample=True in the batched decoder: same default-generator seed -> same tokens (graph path); temperature -> 0 approaches greedy"""
    d, sd, cfg = _tiny()
    emb = torch.from_numpy(d["inputs_embeds"]).cuda().half()
    lm = LM.Qwen2Model(sd, cfg, max_seq=64)
    outs = []
    for _ in range(2):
        torch.manual_seed(11)
        outs.append(LM.BatchDecoder(lm, [emb, emb[:20]], max_new_tokens=6).generate(6, do_sample=True, temperature=0.8))
    assert outs[0] == outs[1] and all(len(t) == 6 for t in outs[0])
    cold = LM.BatchDecoder(lm, [emb, emb[:20]], max_new_tokens=6).generate(6, do_sample=True, temperature=1e-4)
    assert cold == LM.BatchDecoder(lm, [emb, emb[:20]], max_new_tokens=6).generate(6)


def test_pick_token_argmax_and_sampling_vs_torch():
    """sampling.hip: arg-max == torch.argmax (lowest index on ties); sampling == inverse CDF of softmax(logits / T) at the given
    uniforms (checked against an fp64 CDF), at the real vocabulary size and on ragged / strided rows."""
    g = torch.Generator(device="cuda").manual_seed(7)
    for B, V, ld in [(1, 152064, 152064), (8, 152064, 152064), (3, 1000, 1024), (2, 77, 77), (1, 5, 5)]:
        buf = torch.randn(B, ld, device="cuda", generator=g) * 4
        lg = buf[:, :V]
        assert torch.equal(ops.pick_token(lg), torch.argmax(lg, dim=-1))
        lg2 = lg.clone()
        lg2[:, V // 3] = 50.0
        lg2[:, V // 2] = 50.0                                                    # tie: the lower index wins
        assert ops.pick_token(lg2).tolist() == [V // 3] * B
        for T in (0.1, 0.2, 1.0, 5.0):
            u = torch.rand(B, device="cuda", generator=g)
            got = ops.pick_token(lg, T, u)
            p = torch.softmax(lg.double() / T, dim=-1)
            cdf = torch.cumsum(p, dim=-1)
            for b in range(B):
                i = int(got[b])
                lo = float(cdf[b, i - 1]) if i > 0 else 0.0
                hi = float(cdf[b, i])
                assert lo - 2e-5 <= float(u[b]) <= hi + 2e-5, (B, V, T, b, i, lo, float(u[b]), hi)
    # extreme u and a one-hot distribution
    lg = torch.full((2, 4096), -30.0, device="cuda")
    lg[0, 17] = 30.0
    lg[1, 4095] = 30.0
    assert ops.pick_token(lg, 1.0, torch.tensor([1e-6, 0.99999994], device="cuda")).tolist() == [17, 4095]
    # the empirical distribution of many draws follows softmax
    lg = torch.tensor([[0.0, 1.0, 2.0, -1.0]], device="cuda").repeat(20000, 1)
    draws = ops.pick_token(lg, 1.0, torch.rand(20000, device="cuda", generator=g))
    freq = torch.bincount(draws, minlength=4).float() / 20000
    assert (freq.cpu() - torch.softmax(torch.tensor([0.0, 1.0, 2.0, -1.0]), 0)).abs().max() < 0.012


def test_decode_graph_survives_workspace_growth_and_detects_cache_reallocation():
    """the captured graph must not depend on the shared grow-only scratch (a k-means merge between two questions replaces
    it) and a reallocated KV cache must invalidate the graph instead of replaying into freed memory."""
    d, sd, cfg = _tiny()
    emb = torch.from_numpy(d["inputs_embeds"]).cuda().half()
    model = LM.LlavaQwenForCausalLM(LM.Qwen2Model(sd, cfg, max_seq=128))
    ids = torch.arange(5, 12).unsqueeze(0)
    a = model.generate_with_image_embedding(ids, image_embeddings=None, do_sample=False, max_new_tokens=9)
    dg = model._dg
    assert dg is not None and dg.valid()
    ops._workspace(256 << 20, "cuda:0")                                  # what the next k-means merge does: the old scratch block is freed
    junk = torch.full((64 << 20,), 0xFF, dtype=torch.uint8, device="cuda")   # ... and its memory is handed to someone else
    b = model.generate_with_image_embedding(ids, image_embeddings=None, do_sample=False, max_new_tokens=9)
    assert torch.equal(a, b) and model._dg is dg                         # same graph, same tokens
    del junk
    model.lm.reset_cache(max_seq=4096)                                   # a longer prompt reallocates the cache
    assert not dg.valid()
    with pytest.raises(RuntimeError):
        dg.run(2)
    c = model.generate_with_image_embedding(ids, image_embeddings=None, do_sample=False, max_new_tokens=9)
    assert torch.equal(a, c) and model._dg is not dg                     # a fresh graph was captured


def test_decode_graph_at_7b_widths_long_context_matches_eager():
    """C3's decode path at the real widths: 2 layers of Qwen2-7B shape, a 49 152-token context in the KV cache, split-KV attention,
    hipGraph replay == eager decode (token ids and final logits)."""
    cfg = LM.Qwen2ConfigLite(**dict(LM.QWEN2_7B, layers=2, vocab=4096))
    sd = LM.random_qwen2_state_dict(cfg, seed=6, std=0.03)
    S = 49152
    g = torch.Generator(device="cuda").manual_seed(3)

    def fresh():
        lm = LM.Qwen2Model(sd, cfg, max_seq=S + 64)
        lm.reset_cache()
        for c in lm.cache:                                               # a synthetic long context: random K/V rows (fp16), as a prefill leaves them
            c[:S].copy_((torch.randn(S, c.shape[1], device="cuda", generator=torch.Generator(device="cuda").manual_seed(11)) * 0.5).half())
        lm.cache_len = S
        return lm
    lm_a, lm_b = fresh(), fresh()
    first = 123
    dg = LM.DecodeGraph(lm_a, max_new_tokens=16)
    dg.start(first)
    toks_graph = dg.run(6)
    tok, toks_eager = first, []
    for _ in range(6):
        logits = lm_b.forward(lm_b.embed_tokens(torch.tensor([tok], device="cuda")))
        tok = int(ops.pick_token(logits))
        toks_eager.append(tok)
    assert toks_graph == toks_eager and lm_a.cache_len == lm_b.cache_len == S + 6
    # bit-identical, not merely close: the graph's fused q/k/v launch and the eager step's gemv + rotary kernel share the arithmetic (fp32
    # fma, THEN the fp16 rounding) and the split-KV rule; a one-ulp difference in one K row (hipcc fusing fma + cvt into v_fma_mixlo_f16 in
    # one of the two) was caught here in round 3
    assert torch.equal(dg.logits.view(-1), logits.view(-1))
    assert all(torch.equal(x[:S + 6], y[:S + 6]) for x, y in zip(lm_a.cache, lm_b.cache))


def test_generate_graph_path_sampling_and_eos():
    """generate_with_image_embedding runs its token loop as a replayed hipGraph for greedy AND temperature sampling, and stops at EOS:
    (a) EOS = the 4th greedy token -> exactly 4 tokens, EOS included (HF semantics), the cache length is rewound to it;
    (b) temperature 1e-3 sampling == greedy; (c) sampling is reproducible under torch.manual_seed and differs across seeds at T = 2;
    (d) the eager loop (decode_graph=False) gives the same greedy / EOS result."""
    d, sd, cfg = _tiny()
    greedy = d["greedy"].tolist()
    ids = torch.arange(5, 12).unsqueeze(0)
    emb = torch.from_numpy(d["inputs_embeds"]).cuda().half()

    def gen(model, **kw):
        model.prepare_inputs_embeddings_for_multimodal = lambda *a, **k: (None, None, None, None, emb.unsqueeze(0), None)     # the golden prompt
        return model.generate_with_image_embedding(ids, image_embeddings=None, **kw)[0].tolist()
    m = LM.LlavaQwenForCausalLM(LM.Qwen2Model(sd, cfg, max_seq=128))
    assert gen(m, do_sample=False, max_new_tokens=8) == greedy
    eos_tok = greedy[3]
    first_eos = greedy.index(eos_tok)
    m_eos = LM.LlavaQwenForCausalLM(LM.Qwen2Model(sd, cfg, max_seq=128), eos_token_id=[eos_tok, 511])
    out = gen(m_eos, do_sample=False, max_new_tokens=8)
    assert out == greedy[:first_eos + 1]
    assert m_eos.lm.cache_len == 37 + first_eos                     # prompt + the tokens fed back (the EOS itself is not fed)
    assert gen(m_eos, do_sample=False, max_new_tokens=8, decode_graph=False) == out
    assert gen(m_eos, do_sample=False, max_new_tokens=8) == out     # the graph is reused after an early stop
    assert gen(m, do_sample=True, temperature=1e-3, max_new_tokens=8) == greedy
    torch.manual_seed(5); a = gen(m, do_sample=True, temperature=2.0, max_new_tokens=12)
    torch.manual_seed(5); b = gen(m, do_sample=True, temperature=2.0, max_new_tokens=12)
    torch.manual_seed(6); c = gen(m, do_sample=True, temperature=2.0, max_new_tokens=12)
    assert a == b and a != c and len(a) == 12 and all(0 <= t < 512 for t in a)
    assert {k[0] for k in m._dgs} == {0.0, 0.001, 2.0}                    # one graph per Sampling spec (temperature, top_k, top_p, penalty)


def _hf_pick(scores, prev, T, top_k, top_p, pen, u):
    """HF's processor chain on CPU (transformers generation/logits_process.py), then the inverse CDF in index order at u."""
    from transformers.generation.logits_process import (RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper, TopKLogitsWarper,
                                                         TopPLogitsWarper)
    s = scores.clone()
    if pen != 1.0:
        s = RepetitionPenaltyLogitsProcessor(pen)(prev, s)
    if T <= 0:
        return s.argmax(-1), torch.full((s.shape[0],), 1.0)
    s = TemperatureLogitsWarper(T)(prev, s)
    if top_k:
        s = TopKLogitsWarper(top_k)(prev, s)
    if top_p < 1.0:
        s = TopPLogitsWarper(top_p)(prev, s)
    p = torch.softmax(s.double(), -1)
    cdf = p.cumsum(-1)
    target = (u.double() * cdf[:, -1]).unsqueeze(1)
    tok = torch.searchsorted(cdf, target, right=True).squeeze(1).clamp(max=s.shape[1] - 1)
    lo = torch.where(tok > 0, cdf.gather(1, (tok - 1).clamp(min=0).unsqueeze(1)).squeeze(1), torch.zeros_like(cdf[:, 0]))
    hi = cdf.gather(1, tok.unsqueeze(1)).squeeze(1)
    margin = torch.minimum(target.squeeze(1) - lo, hi - target.squeeze(1)) / cdf[:, -1]
    return tok, margin


@pytest.mark.parametrize("V", [152064, 1000])
def test_sample_token_matches_hf_logits_processors(V):
    """sc_sample_token_f32 (repetition penalty over distinct ids -> temperature -> top-k with ties -> top-p -> one draw) against the
    transformers processor classes the reference's `generate` call runs through (llava_qwen.py:155 -> GenerationMixin), on the
    Qwen2 vocabulary size and a small one; draws that land within 1e-6 of a CDF boundary are not compared."""
    B = 3
    g = torch.Generator().manual_seed(V)
    base = torch.randn(B, V, generator=g) * 3.0
    base[0, 17] = base[0].max() + 0.5                        # a clear winner that the penalty demotes
    base[1, 5] = base[1, V - 9] = base[1, V // 2] = base[1].topk(20).values[-1]     # ties at the k-th value, in different slices
    prev = torch.tensor([[17, 17, 3, 40, 17, 999, 5], [5, 6, 7, 8, 9, 10, 11], [0, 1, 2, 3, 4, 5, 6]])
    prev_d = prev.cuda()
    n_dev = torch.full((B,), prev.shape[1], dtype=torch.int32, device="cuda")
    cases = [(0.0, 0, 1.0, 1.0), (0.0, 0, 1.0, 1.3), (0.7, 0, 1.0, 1.05), (0.2, 20, 1.0, 1.05), (0.7, 20, 0.8, 1.05), (1.0, 64, 0.5, 1.0),
             (0.7, 1, 1.0, 1.0), (1.5, 5, 0.95, 2.0)]
    compared = 0
    for T, top_k, top_p, pen in cases:
        for trial in range(6):
            u = torch.rand(B, generator=g)
            ref, margin = _hf_pick(base, prev, T, top_k, top_p, pen, u)
            lg = base.cuda().clone()
            got = ops.sample_token(lg, T, u.cuda() if T > 0 else None, top_k=top_k, top_p=top_p, repetition_penalty=pen, prev_ids=prev_d,
                                   n_prev=n_dev if trial % 2 else None).cpu()
            for b in range(B):
                if margin[b] > 1e-6:
                    assert int(got[b]) == int(ref[b]), (V, T, top_k, top_p, pen, trial, b, int(got[b]), int(ref[b]), float(margin[b]))
                    compared += 1
            if pen != 1.0:                                   # the penalty is applied in place, once per distinct id
                exp = base.clone()
                from transformers.generation.logits_process import RepetitionPenaltyLogitsProcessor
                exp = RepetitionPenaltyLogitsProcessor(pen)(prev, exp)
                assert torch.equal(lg.cpu(), exp)
    assert compared > 100


@pytest.mark.parametrize("V", [152064, 1000])
def test_sample_token_full_vocabulary_top_k_top_p_matches_hf(V):
    """The corners HF `generate` accepts beyond the 64-candidate path (round 3: k_full_filter, integer radix select over the whole row): a
    nucleus without top-k (the reference forwards --top_p, inference_streaming_longva_v2.py:245-256), top_k > 64, both together, ties at
    the k-th value.  Kept-token COUNT against the transformers warpers (exact for top-k, which is integer work; within 2 tokens for the
    nucleus, whose boundary depends on the summation order of ~1e-5 probabilities) and the drawn token wherever u is not within 1e-4 of a
    CDF boundary."""
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    B = 3
    g = torch.Generator().manual_seed(V + 1)
    base = torch.randn(B, V, generator=g) * 3.0
    base[1, 5] = base[1, V - 9] = base[1, V // 2] = base[1].topk(100).values[-1]          # ties at the 100th value
    base[2, :50] += 6.0                                                                     # a peaky row: small nucleus
    prev = torch.zeros(B, 1, dtype=torch.long)
    ws = torch.zeros(max(ops.sample_token_workspace_bytes(B), 256), dtype=torch.uint8, device="cuda")
    compared = 0
    for T, top_k, top_p in [(0.7, 0, 0.9), (0.2, 0, 0.5), (1.0, 100, 1.0), (0.7, 200, 0.8), (1.3, 0, 0.99), (0.7, 65, 0.3), (1.0, V, 0.7)]:
        s = TemperatureLogitsWarper(T)(prev, base.clone())
        if top_k and top_k < V:
            s = TopKLogitsWarper(top_k)(prev, s)
        k_only = torch.isfinite(s).sum(-1)
        if top_p < 1.0:
            s = TopPLogitsWarper(top_p)(prev, s)
        n_hf = torch.isfinite(s).sum(-1)
        for trial in range(4):
            u = torch.rand(B, generator=g)
            ref, margin = _hf_pick(base, prev, T, top_k if top_k < V else 0, top_p, 1.0, u)
            got = ops.sample_token(base.cuda().clone(), T, u.cuda(), top_k=top_k, top_p=top_p, ws=ws).cpu()
            diag = ws[:8 * B].view(torch.int32).view(B, 2).cpu()
            for b in range(B):
                n = int(diag[b, 1])
                if top_p >= 1.0:
                    assert n == int(n_hf[b]), (V, T, top_k, top_p, b, n, int(n_hf[b]))                 # top-k with ties: exact
                else:
                    assert abs(n - int(n_hf[b])) <= 2 and 1 <= n <= int(k_only[b]), (V, T, top_k, top_p, b, n, int(n_hf[b]))
                assert torch.isfinite(base[b, int(got[b])] * 0 + 1) and 0 <= int(got[b]) < V
                if margin[b] > 1e-4 and n == int(n_hf[b]):
                    assert int(got[b]) == int(ref[b]), (V, T, top_k, top_p, trial, b, int(got[b]), int(ref[b]), float(margin[b]))
                    compared += 1
    assert compared > 40


def test_resolve_sampling_follows_hf_generation_config_semantics():
    """Caller arguments (even None) override generation_config.json, which overrides HF's defaults (top_k 50); warpers only exist when
    sampling; the repetition penalty also applies to greedy decoding (it is a logits processor, not a warper)."""
    R, S = LM.resolve_sampling, LM.Sampling
    assert R({}, False) == S(0.0, 0, 1.0, 1.0)
    assert R({}, True, 0.2, None) == S(0.2, 50, 1.0, 1.0)                                   # the reference's call: temperature + top_p=None
    qwen = dict(do_sample=True, temperature=0.7, top_k=20, top_p=0.8, repetition_penalty=1.05)  # Qwen2-7B-Instruct's generation_config.json
    assert R(qwen, True, 0.2, None) == S(0.2, 20, 1.0, 1.05)
    assert R(qwen, True) == S(0.7, 20, 0.8, 1.05)
    assert R(qwen, False) == S(0.0, 0, 1.0, 1.05)
    assert R(qwen, LM._UNSET) == S(0.7, 20, 0.8, 1.05)            # do_sample not passed: the checkpoint's file decides ...
    assert R({}, LM._UNSET) == R({}, None) == S(0.0, 0, 1.0, 1.0)  # ... then HF's default: greedy
    assert R(qwen, True, 0.2, None, top_k=None, repetition_penalty=None) == S(0.2, 0, 1.0, 1.0)
    assert R({}, True, 0.2, 0.9, top_k=None) == S(0.2, 0, 0.9, 1.0)                        # nucleus without top-k: the full-vocabulary path (round 3)
    assert R({}, True, 0.2, None, top_k=100) == S(0.2, 100, 1.0, 1.0)                      # more than the 64 candidates of the fast path
    with pytest.raises(ValueError):
        R({}, True, 0.0)


def test_generate_applies_generation_config_in_graph_and_eager_loops():
    """With a generation_config (repetition_penalty 1.3, top_k 3) the replayed hipGraph and the eager loop (user generator) must agree
    token for token under greedy decoding, differ from the unpenalised run on a model that would otherwise repeat, and the sampling
    run must only ever emit tokens of the top-3 of its own step (checked by re-running the eager loop with the same uniform draws)."""
    d = np.load(os.path.join(G, "qwen2_tiny.npz"))
    cfg = LM.Qwen2ConfigLite(hidden=256, layers=2, heads=4, kv_heads=2, intermediate=512, vocab=512, rope_theta=1e6)
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("lm.")}
    model = LM.LlavaQwenForCausalLM(LM.Qwen2Model(sd, cfg, device="cuda", max_seq=256))
    ids = torch.tensor([[5, 9, 3, 7, 11, 2]])
    gen = lambda **kw: model.generate_with_image_embedding(ids, image_embeddings=None, max_new_tokens=24, **kw)[0].tolist()
    plain = gen(do_sample=False)
    model.generation_config = dict(repetition_penalty=1.3, top_k=3)
    graph = gen(do_sample=False)
    eager = gen(do_sample=False, decode_graph=False)
    assert graph == eager and len(graph) == 24
    assert len(set(graph)) >= len(set(plain)) and graph != plain                         # the penalty changes a greedy sequence that repeats
    torch.manual_seed(5)
    a = gen(do_sample=True, temperature=0.9, top_p=None)
    torch.manual_seed(5)
    b = gen(do_sample=True, temperature=0.9, top_p=None)
    assert a == b and len(a) == 24                                                       # graph sampling is reproducible under manual_seed
    override = gen(do_sample=False, repetition_penalty=None)
    assert override == plain                                                             # an explicit None switches the penalty off (HF semantics)


def test_sample_token_edge_cases():
    """Tiny vocabularies (fewer entries than slices), top_k >= V (falls back to the plain pick), an empty penalty set, -inf logits."""
    g = torch.Generator().manual_seed(3)
    for V, top_k in [(7, 3), (50, 64), (65, 64), (300, 1)]:
        base = torch.randn(2, V, generator=g) * 2.0
        base[1, V // 2] = float("-inf")
        prev = torch.zeros((2, 4), dtype=torch.int64)
        for T, top_p, pen, n_prev in [(0.0, 1.0, 1.2, 0), (0.8, 1.0, 1.2, 4), (0.8, 0.7, 1.0, 0)]:
            for _ in range(4):
                u = torch.rand(2, generator=g)
                ref, margin = _hf_pick(base, prev[:, :n_prev], T, min(top_k, V), top_p, pen, u)
                got = ops.sample_token(base.cuda().clone(), T, u.cuda() if T > 0 else None, top_k=top_k, top_p=top_p, repetition_penalty=pen,
                                       prev_ids=prev.cuda(), n_prev=n_prev).cpu()
                for b in range(2):
                    if margin[b] > 1e-6:
                        assert int(got[b]) == int(ref[b]), (V, top_k, T, top_p, pen, n_prev, b, int(got[b]), int(ref[b]))


def test_batched_generation_applies_the_same_processor_chain():
    """BatchDecoder (the batched chunk captions) under a generation_config with a repetition penalty: every sequence of the batch must
    equal its own single-sequence generate (graph and eager), i.e. the per-row history / counters feed sc_sample_token_f32 correctly."""
    d = np.load(os.path.join(G, "qwen2_tiny.npz"))
    cfg = LM.Qwen2ConfigLite(hidden=256, layers=2, heads=4, kv_heads=2, intermediate=512, vocab=512, rope_theta=1e6)
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("lm.")}
    model = LM.LlavaQwenForCausalLM(LM.Qwen2Model(sd, cfg, device="cuda", max_seq=256))
    model.generation_config = dict(repetition_penalty=1.4, top_k=5)
    prompts = [torch.tensor([[5, 9, 3, 7, 11, 2]]), torch.tensor([[8, 8, 1]]), torch.tensor([[5, 9, 3, 7, 11, 2]])]
    single = [model.generate_with_image_embedding(p, image_embeddings=None, do_sample=False, max_new_tokens=12)[0].tolist() for p in prompts]
    batch = model.generate_batch_with_image_embedding(prompts, [None] * 3, do_sample=False, max_new_tokens=12)
    assert [b[0].tolist() for b in batch] == single and single[0] == single[2]
    plain = LM.LlavaQwenForCausalLM(model.lm).generate_batch_with_image_embedding(prompts, [None] * 3, do_sample=False, max_new_tokens=12)
    assert [b[0].tolist() for b in plain] != single                                       # the penalty really changes these sequences


# ---- round 6: per-sequence sampling seeds (sc_counter_uniform_f32) ----
def test_counter_uniform_kernel_equals_the_host_formula():
    seeds = torch.tensor([0, 1, 123456789012345, 2 ** 62 - 1, 42], dtype=torch.int64, device="cuda")
    for n in (0, 1, 2, 1000, 2 ** 33 + 5):
        cnt = torch.tensor([n], dtype=torch.int64, device="cuda")
        u = ops.counter_uniform(seeds, cnt, 0).cpu().tolist()
        u2 = ops.counter_uniform(seeds, None, n).cpu().tolist()
        ref = [ops.counter_uniform_host(int(sd), n) for sd in seeds.cpu().tolist()]
        assert u == ref == u2 and all(0.0 <= x < 1.0 for x in u)
    big = ops.counter_uniform(torch.arange(1 << 16, dtype=torch.int64, device="cuda"), None, 7)
    assert 0.49 < big.mean().item() < 0.51 and big.min().item() >= 0 and big.max().item() < 1


def test_a_sequence_samples_the_same_tokens_alone_batched_eagerly_and_from_a_graph():
    """temperature sampling with per-sequence seeds: BatchDecoder == one generate per sequence (same seeds), hipGraph == eager loop, a CUDA or
    CPU `generator` only seeds, and two host threads sampling at the same time do not disturb each other."""
    import threading
    d, sd, cfg = _tiny()
    emb = torch.from_numpy(d["inputs_embeds"]).cuda().half()
    prompts = [emb, emb[:20], emb[5:30]]
    lm = LM.Qwen2Model(sd, cfg, max_seq=96)
    seeds = [11, 222, 3333]
    sp = LM.resolve_sampling({}, True, 1.5, LM._UNSET, LM._UNSET, LM._UNSET)         # what generate_with_image_embedding resolves: HF's default top_k = 50 included
    batched = LM.BatchDecoder(lm, prompts, max_new_tokens=10).generate(10, sampling=sp, seed=seeds)
    batched_eager = LM.BatchDecoder(lm, prompts, max_new_tokens=10).generate(10, sampling=sp, seed=seeds, use_graph=False)
    assert batched == batched_eager and len({tuple(t) for t in batched}) == 3

    def one(model, e, seed, **kw):
        model.prepare_inputs_embeddings_for_multimodal = lambda *a, **k: (None, None, None, None, e.unsqueeze(0), None)
        return model.generate_with_image_embedding(torch.arange(5, 12).unsqueeze(0), image_embeddings=None, do_sample=True, temperature=1.5, max_new_tokens=10,
                                                   seed=seed, **kw)[0].tolist()
    m = LM.LlavaQwenForCausalLM(lm)
    alone = [one(m, e, sdd) for e, sdd in zip(prompts, seeds)]
    assert alone == batched
    assert [one(m, e, sdd, decode_graph=False) for e, sdd in zip(prompts, seeds)] == alone
    # a generator only seeds: CPU generator -> same seeds as draw_seeds gives for it; reproducible
    g1, g2 = torch.Generator().manual_seed(9), torch.Generator().manual_seed(9)
    want = int(LM.draw_seeds(1, g1)[0])
    m.prepare_inputs_embeddings_for_multimodal = lambda *a, **k: (None, None, None, None, emb.unsqueeze(0), None)
    a = m.generate_with_image_embedding(torch.arange(5, 12).unsqueeze(0), image_embeddings=None, do_sample=True, temperature=1.5, max_new_tokens=10, generator=g2)[0].tolist()
    assert a == one(m, emb, want)
    # two host threads, two models (shared weights, own caches), sampling concurrently: each reproduces its single-threaded result
    m2 = LM.LlavaQwenForCausalLM(lm.shared_view(96))
    solo = [one(m, emb, 5), one(m2, emb[:20], 6)]
    res = [None, None]

    def work(i, model, e, sdd):
        with torch.cuda.stream(torch.cuda.Stream()):
            res[i] = [one(model, e, sdd) for _ in range(6)]
    th = [threading.Thread(target=work, args=(0, m, emb, 5)), threading.Thread(target=work, args=(1, m2, emb[:20], 6))]
    [t.start() for t in th]; [t.join() for t in th]
    assert all(r == solo[0] for r in res[0]) and all(r == solo[1] for r in res[1])
