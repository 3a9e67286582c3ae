#!/usr/bin/env python3
"""The batched caption decode step alone (B chunk captions decoded together, llm.BatchDecoder) at the product's shape:
B sequences x ctx cached tokens, Qwen2-7B random-init.  The caches are filled with random K/V instead of 26 real prefills (8 s), so
that a rocprofv3 trace of this command holds the decode step's kernels only.  ms per step and the HBM roofline fraction."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import llm as LM

ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=26)
ap.add_argument("--ctx", type=int, default=22740)
ap.add_argument("--new", type=int, default=48)
ap.add_argument("--eager", action="store_true")
ap.add_argument("--nsplit", type=int, nargs="*", default=[0])
a = ap.parse_args()
cfg = LM.Qwen2ConfigLite(**LM.QWEN2_7B)
lm = LM.Qwen2Model(LM.random_qwen2_state_dict(cfg, seed=0), cfg, max_seq=4096, consume=True)
g = torch.Generator(device="cuda").manual_seed(1)
prompts = [(torch.randn(16, cfg.hidden, device="cuda", generator=g) * 0.02).half() for _ in range(a.chunks)]
dec = LM.BatchDecoder(lm, prompts, a.ctx + a.new)          # cap = 16 + ctx + new rows per sequence
for c in dec.cache:
    c.normal_(0, 0.5)
dec.len.fill_(a.ctx)
torch.cuda.synchronize()
for ns in a.nsplit:
    dec.nsplit_override = ns or None
    dec.len.fill_(a.ctx)
    dec.generate(8, use_graph=not a.eager)                     # warm-up (captures a graph of its own)
    dec.len.fill_(a.ctx)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dec.generate(a.new, use_graph=not a.eager)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    steps = a.new - 1
    w_bytes = sum(t.numel() * 2 for L in lm.L for t in (L["wq"], L["wkv"], L["wo"], L["wgu"], L["wd"])) + lm.lm_head.numel() * 2
    kv_bytes = a.chunks * (a.ctx + a.new / 2) * 2 * cfg.kv_heads * cfg.head_dim * 2 * cfg.layers
    print(json.dumps(dict(chunks=a.chunks, ctx=a.ctx, steps=steps, ms_per_step=round(1e3 * dt / steps, 3), tok_per_s=round(a.chunks * steps / dt, 1),
                          GB_per_step=round((w_bytes + kv_bytes) / 1e9, 2), TBps=round((w_bytes + kv_bytes) / (dt / steps) / 1e12, 3),
                          frac_of_8TBps=round((w_bytes + kv_bytes) / (dt / steps) / 8e12, 3), nsplit=dec.nsplit)))
