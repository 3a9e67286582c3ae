#!/usr/bin/env python3
"""CU-partitioned overlap of the HBM-bound answer decode (DecodeGraph, 49 k context) with MFMA-bound prefill work (the
reference's reader / updater / QA threads).  Each job alone on the whole chip, each job alone on its CU partition, then both at once."""
import argparse, copy, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import llm as LM, ops

ap = argparse.ArgumentParser()
ap.add_argument("--ctx", type=int, default=49152)
ap.add_argument("--tokens", type=int, default=192)
ap.add_argument("--prefill", type=int, default=16384)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--dec-cus", type=int, nargs="*", default=[32])
a = ap.parse_args()
cfg = LM.Qwen2ConfigLite(**LM.QWEN2_7B)
lm = LM.Qwen2Model(LM.random_qwen2_state_dict(cfg, seed=0), cfg, max_seq=a.ctx + 1100, consume=True)
lm.reset_cache()
for l in range(cfg.layers):
    lm.cache[l][:a.ctx].normal_(0, 0.5)
lm2 = copy.copy(lm)                       # the same weights, its own KV cache and activation buffers: the NEXT segment's prefill
lm2.cache, lm2.cache_len, lm2._buf_rows, lm2.max_seq = None, 0, 0, a.prefill + 8
lm2.reset_cache()
emb = (torch.randn(a.prefill, cfg.hidden, device="cuda") * 0.02).half()
dg = LM.DecodeGraph(lm, max_new_tokens=1024, nsplit=64)
lm.cache_len = a.ctx
dg.start(1); dg.capture(); torch.cuda.synchronize()
ncu = ops.device_info()["cu_count"]


def decode(stream):
    lm.cache_len = a.ctx
    with torch.cuda.stream(stream):
        dg.start(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(a.tokens):
            dg.graph.replay()
        e1.record(stream)
    return e0, e1


def prefill(stream, budget):
    try:                                   # (ABI 7: a masked stream carries its CU count; `budget` is informational)
        with torch.cuda.stream(stream):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(a.reps):
                lm2.cache_len = 0
                lm2.forward(emb)
            e1.record(stream)
    finally:
        pass
    return e0, e1


def ms(ev):
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1])


cur = torch.cuda.current_stream()
prefill(cur, 0); decode(cur); torch.cuda.synchronize()            # warm-up
t_dec, t_pre = ms(decode(cur)), ms(prefill(cur, 0))
out = dict(ctx=a.ctx, decode_tokens=a.tokens, prefill_tokens=a.prefill * a.reps, whole_chip=dict(decode_ms=round(t_dec, 1), decode_tok_s=round(a.tokens / t_dec * 1e3, 1),
                                                                                                   prefill_ms=round(t_pre, 1), serial_ms=round(t_dec + t_pre, 1)))
print(json.dumps(out), flush=True)
for dc in a.dec_cus:
    s_dec, s_main = ops.masked_stream(0, dc), ops.masked_stream(dc, ncu - dc)
    s_dec.wait_stream(cur); s_main.wait_stream(cur)
    d_alone = ms(decode(s_dec))
    p_alone = ms(prefill(s_main, ncu - dc))
    # (round 5, second look: a hipGraph replay costs the HOST ~2.9 ms - as long as the token takes on the GPU - so enqueuing 256 replays first kept the
    #  prefill out of its stream for 0.7 s and the first version of this tool measured two jobs that barely overlapped.  Two ways to really
    #  overlap them: enqueue the short-to-enqueue job first, or enqueue from two host threads as the reference's reader / updater / QA design does)
    res = {}
    for mode in ("prefill_enqueued_first", "two_host_threads"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "prefill_enqueued_first":
            ep = prefill(s_main, ncu - dc)
            ed = decode(s_dec)
        else:
            import threading
            box = {}
            th = threading.Thread(target=lambda: box.__setitem__("d", decode(s_dec)))
            th.start()
            ep = prefill(s_main, ncu - dc)
            th.join()
            ed = box["d"]
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
        res[mode] = dict(wall_ms=round(wall, 1), decode_ms=round(ed[0].elapsed_time(ed[1]), 1), prefill_ms=round(ep[0].elapsed_time(ep[1]), 1),
                         speedup_vs_serial_whole_chip=round((t_dec + t_pre) / wall, 3))
    print(json.dumps(dict(decode_cus=dc, main_cus=ncu - dc, decode_alone_ms=round(d_alone, 1), decode_alone_tok_s=round(a.tokens / d_alone * 1e3, 1),
                          prefill_alone_ms=round(p_alone, 1), both=res)), flush=True)
