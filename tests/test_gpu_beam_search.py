"""Beam search on the HIP decoder (llm.LlavaQwenForCausalLM.generate_with_image_embedding(num_beams=N, do_sample=False); streamchat_amd/beam.py is
the bookkeeping, BatchDecoder the model side: one prefill, cache rows replicated over N slots, caches gathered by the beams' origins every step)
against transformers' own beam search on the tiny Qwen2 (tests/golden/qwen2_tiny_beams.json: 108 cases).  The CPU test
(test_beam_search.py) shows the algorithm equal to HF's on fp32 logits in all 108 cases; here the logits come from fp16 kernels, so a case whose
decision hangs on a margin below the kernels' accuracy may legitimately end elsewhere - such a case must at least score within the kernels'
accuracy of HF's result, and there may be only a few of them."""
import json
import os

import numpy as np
import pytest
import torch

from streamchat_amd import llm as LM

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _tiny():
    d = np.load(os.path.join(G, "qwen2_tiny.npz"))
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("lm.")}
    heads, kv, layers, hd = (int(x) for x in d["cfg"])
    cfg = LM.Qwen2ConfigLite(hidden=256, layers=layers, heads=heads, kv_heads=kv, intermediate=512, vocab=512)
    return d, sd, cfg


def test_beam_search_matches_hf_goldens():
    d, sd, cfg = _tiny()
    emb = torch.from_numpy(d["inputs_embeds"]).cuda().half()
    cases = json.load(open(os.path.join(G, "qwen2_tiny_beams.json")))["cases"]
    exact, near = 0, []
    for c in cases:
        a, b = c["rows"]
        m = LM.LlavaQwenForCausalLM(LM.Qwen2Model(sd, cfg, max_seq=96), eos_token_id=c["eos"])
        m.prepare_inputs_embeddings_for_multimodal = lambda *x, _e=emb[a:b], **k: (None, None, None, None, _e.unsqueeze(0), None)
        out = m.generate_with_image_embedding(torch.arange(5, 12).unsqueeze(0), image_embeddings=None, do_sample=False, num_beams=c["num_beams"],
                                              max_new_tokens=c["max_new_tokens"])[0].tolist()
        if out == c["tokens"]:
            exact += 1
            assert abs(m.last_beam_score - c["score"]) < 3e-3 * max(1.0, abs(c["score"]))
        else:
            near.append((c["prompt"], c["num_beams"], c["max_new_tokens"], c["eos"], m.last_beam_score, c["score"]))
    print(f"\n[beam] {exact} of {len(cases)} cases token-identical to HF; others (ours / HF score):", [(round(x[4], 4), round(x[5], 4)) for x in near])
    assert exact >= len(cases) - 6
    for x in near:                                          # a different ending only where the two hypotheses score the same to the kernels' accuracy
        assert abs(x[4] - x[5]) < 5e-3 * max(1.0, abs(x[5])), x


def test_beam_search_refuses_what_is_not_built_and_one_beam_is_greedy():
    d, sd, cfg = _tiny()
    emb = torch.from_numpy(d["inputs_embeds"]).cuda().half()
    m = LM.LlavaQwenForCausalLM(LM.Qwen2Model(sd, cfg, max_seq=96))
    m.prepare_inputs_embeddings_for_multimodal = lambda *x, **k: (None, None, None, None, emb.unsqueeze(0), None)
    ids = torch.arange(5, 12).unsqueeze(0)
    with pytest.raises(NotImplementedError):
        m.generate_with_image_embedding(ids, image_embeddings=None, do_sample=True, temperature=0.7, num_beams=3, max_new_tokens=4)
    assert m.generate_with_image_embedding(ids, image_embeddings=None, do_sample=False, num_beams=1, max_new_tokens=8)[0].tolist() == d["greedy"].tolist()
