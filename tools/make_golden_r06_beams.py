#!/usr/bin/env python3
"""Golden outputs of HF beam search (transformers `generate(num_beams=N, do_sample=False)`, the call the reference makes with
--num_beams N, inference_streaming_longva_v2.py:252-256) on the tiny Qwen2 of tests/golden/qwen2_tiny.npz: the model is rebuilt from the
fixture's own weights, run on CPU in fp32, and only inputs + expected token ids are written (tests/golden/qwen2_tiny_beams.json).
Cases: beam counts 2 / 3 / 4, with and without EOS ids (an EOS that the beams actually meet, so that hypotheses finish, scores are
length-normalised and the early-stop heuristic fires), several max_new_tokens, two prompts."""
import json, os, sys
import numpy as np, torch, transformers
from transformers import Qwen2Config, Qwen2ForCausalLM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = np.load(os.path.join(ROOT, "tests/golden/qwen2_tiny.npz"))
cfg = Qwen2Config(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                  num_key_value_heads=2, max_position_embeddings=512, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=False)
m = Qwen2ForCausalLM(cfg).eval()
m.load_state_dict({k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("lm.")})
emb = torch.from_numpy(d["inputs_embeds"]).unsqueeze(0)
with torch.no_grad():
    assert m.generate(inputs_embeds=emb, max_new_tokens=8, do_sample=False)[0].tolist() == d["greedy"].tolist()       # the fixture's model, bit for bit
cases = []
prompts = {"full": (0, 37), "head20": (0, 20), "mid": (5, 30)}
for pname, (a, b) in prompts.items():
    e = emb[:, a:b]
    with torch.no_grad():
        free = m.generate(inputs_embeds=e, max_new_tokens=12, do_sample=False, num_beams=3)[0].tolist()
    for nb in (2, 3, 4):
        for mx in (6, 12, 20):
            for eos in (None, [free[2]], [free[4], free[1]], [free[-1]]):
                kw = dict(inputs_embeds=e, max_new_tokens=mx, do_sample=False, num_beams=nb, pad_token_id=0)
                if eos is not None:
                    kw["eos_token_id"] = eos
                with torch.no_grad():
                    out = m.generate(**kw, return_dict_in_generate=True, output_scores=True)
                cases.append(dict(prompt=pname, rows=[a, b], num_beams=nb, max_new_tokens=mx, eos=eos, tokens=out.sequences[0].tolist(),
                                  score=float(out.sequences_scores[0])))
json.dump(dict(transformers=transformers.__version__, cases=cases), open(os.path.join(ROOT, "tests/golden/qwen2_tiny_beams.json"), "w"))
print(len(cases), "cases;", sum(1 for c in cases if c["eos"] and c["tokens"][-1] in c["eos"]), "end in an EOS;", len({tuple(c['tokens']) for c in cases}), "distinct outputs")
