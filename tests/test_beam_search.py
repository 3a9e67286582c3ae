"""streamchat_amd/beam.py against transformers' own beam search (tests/golden/qwen2_tiny_beams.json, tools/make_golden_r06_beams.py: 108 cases -
2 / 3 / 4 beams, three prompts, 6 / 12 / 20 new tokens, no EOS / EOS ids the beams meet).  Here on the CPU the bookkeeping is driven by fp32 logits of
the SAME tiny Qwen2 (oracle/torch_ref on the fixture's weights, the whole sequence recomputed per step): every case must give HF's tokens and HF's
sequence score - the algorithm, separated from the HIP kernels' numerics (those: tests/test_gpu_beam_search.py)."""
import json
import os

import numpy as np
import torch

from oracle import torch_ref as R
from streamchat_amd.beam import beam_search

G = os.path.join(os.path.dirname(__file__), "golden")


def test_beam_bookkeeping_reproduces_hf_on_fp32_logits():
    d = np.load(os.path.join(G, "qwen2_tiny.npz"))
    sd = {k[3:]: torch.from_numpy(d[k]).float() for k in d.files if k.startswith("lm.")}
    emb_all = torch.from_numpy(d["inputs_embeds"]).float()
    heads, kv, layers, hd = (int(x) for x in d["cfg"])
    cases = json.load(open(os.path.join(G, "qwen2_tiny_beams.json")))["cases"]
    table = sd["model.embed_tokens.weight"]
    n_all = len(cases)
    if torch.cuda.is_available():        # on the GPU box (where every test runs under -m gpu) a third of the cases: its host cores take 4x as long for
        cases = cases[::3]               # this fp32 CPU loop, and all 108 run there on the HIP decoder (test_gpu_beam_search.py) and here in the CPU suite

    def logits_of(seq_emb):                                  # last-position logits of one sequence of embeddings
        return R.qwen2_logits(sd, seq_emb, heads=heads, kv_heads=kv, layers=layers, head_dim=hd, last_only=True)
    n_eos_end = 0
    for c in cases:
        a, b = c["rows"]
        prompt = emb_all[a:b]
        N = c["num_beams"]
        state = {"seqs": [[] for _ in range(N)]}

        def step(tokens, origin):
            state["seqs"] = [state["seqs"][int(o)] + [int(t)] for o, t in zip(origin, tokens)]
            return torch.stack([logits_of(torch.cat([prompt, table[torch.tensor(s)]])) for s in state["seqs"]])
        toks, score = beam_search(logits_of(prompt), step, N, c["max_new_tokens"], c["eos"] or ())
        assert toks == c["tokens"], (c["prompt"], N, c["max_new_tokens"], c["eos"], toks, c["tokens"])
        assert abs(score - c["score"]) < 2e-4 * max(1.0, abs(c["score"]))
        n_eos_end += bool(c["eos"]) and toks[-1] in c["eos"]
    assert n_all == 108 and n_eos_end >= (40 if len(cases) == n_all else 10)
