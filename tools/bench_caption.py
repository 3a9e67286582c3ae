#!/usr/bin/env python3
"""Chunk captioning at the reference's shape: B chunks x (40 frames x 576 tokens + prompt), N new tokens each, Qwen2-7B
random-init.  Times B batch-1 generations (prefill + graph decode) against one BatchDecoder run (same prefills, shared decode steps)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import llm as LM

ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=8)
ap.add_argument("--tokens", type=int, default=23040 + 60)
ap.add_argument("--new", type=int, default=32)
ap.add_argument("--skip-one", action="store_true", help="only the batched run (profiling)")
a = ap.parse_args()
cfg = LM.Qwen2ConfigLite(**LM.QWEN2_7B)
lm = LM.Qwen2Model(LM.random_qwen2_state_dict(cfg, seed=0), cfg, max_seq=a.tokens + a.new + 8, consume=True)
g = torch.Generator(device="cuda").manual_seed(1)
prompts = [(torch.randn(a.tokens, cfg.hidden, device="cuda", generator=g) * 0.02).half() for _ in range(a.chunks)]

def sync():
    torch.cuda.synchronize(); return time.perf_counter()

# one by one: prefill + captured-graph greedy decode
dg = LM.DecodeGraph(lm, max_new_tokens=max(a.new, 16))
t_pre = t_dec = 0.0
for i, e in enumerate([] if a.skip_one else prompts):
    lm.reset_cache()
    t0 = sync(); logits = lm.forward(e); t1 = sync()
    dg.start(int(logits.argmax()))
    if dg.graph is None:
        dg.capture(); torch.cuda.synchronize()
    t2 = sync(); dg.run(a.new - 1); t3 = sync()
    if i > 0:                                   # first chunk pays allocator / capture warm-up
        t_pre += t1 - t0; t_dec += t3 - t2
n = a.chunks - 1
one = dict(prefill_s_per_chunk=t_pre / n, decode_tok_per_s=(a.new - 1) / (t_dec / n)) if not a.skip_one else dict(prefill_s_per_chunk=float("nan"), decode_tok_per_s=float("nan"))
lm.cache = None
torch.cuda.empty_cache()
t0 = sync(); dec = LM.BatchDecoder(lm, prompts, a.new); t1 = sync(); out = dec.generate(a.new); t2 = sync()
bat = dict(prefill_s_per_chunk=(t1 - t0) / a.chunks, decode_tok_per_s=a.chunks * (a.new - 1) / (t2 - t1), step_ms=1e3 * (t2 - t1) / (a.new - 1))
print(json.dumps(dict(chunks=a.chunks, context=a.tokens, new_tokens=a.new, one_by_one=one, batched=bat,
                      caption_s_per_chunk=dict(one_by_one=one["prefill_s_per_chunk"] + a.new / one["decode_tok_per_s"],
                                               batched=bat["prefill_s_per_chunk"] + a.new / bat["decode_tok_per_s"]))))
