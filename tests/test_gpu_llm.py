"""GPU: HIP Qwen2 (prefill + KV-cache decode), the `<image>` splice and generate_with_image_embedding against the golden
vectors of the real HF Qwen2ForCausalLM (tiny config) / the reference's own splice function, and the fp32 PyTorch
restatement at Qwen2-7B widths.  Tolerance on logits: fp16 storage + fp32 accumulate -> 3e-2 of the max |logit|;
greedy token ids must match exactly."""
import os

import numpy as np
import pytest
import torch

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
    assert (logits - ref).abs().max().item() < 3e-2 * ref.abs().max().item()
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
    assert (logits - ref).abs().max().item() < 3e-2 * ref.abs().max().item()


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
