#!/usr/bin/env python3
"""A few EAGER decode steps at a given context (no hipGraph: rocprofv3's counter collection does not survive graph replays) for PMC passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import llm as LM
ctx, n = int(sys.argv[1]), int(sys.argv[2])
cfg = LM.Qwen2ConfigLite(**LM.QWEN2_7B)
lm = LM.Qwen2Model(LM.random_qwen2_state_dict(cfg, seed=0), cfg, max_seq=ctx + 64, consume=True)
lm.reset_cache()
for l in range(cfg.layers):
    lm.cache[l][:ctx].normal_(0, 0.5)
lm.cache_len = ctx
lm._nsplit_prompt = LM.decode_nsplit(cfg.head_dim, ctx)
tok = torch.tensor([1], device="cuda")
for _ in range(n):
    logits = lm.forward(lm.embed_tokens(tok))
    tok = logits.argmax().view(1)
torch.cuda.synchronize()
print("tokens", n)
