#!/usr/bin/env python3
"""Sustained launches of one hot kernel while sampling the package power and the shader clock (rocm-smi): shows which kernels sit on
the board's power cap.  usage: python tools/power_probe.py attn49k|gemm_llm|gemm_llm_8wave|gemm_llm_zeros|gemm_llm_const|gemm_vit|gemm_vit_zeros|vit_attn|decode|idle [seconds]"""
import os, subprocess, sys, threading, time
what = sys.argv[1] if len(sys.argv) > 1 else "attn49k"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
if what == "gemm_llm_8wave":
    os.environ["SC_GEMM_FAT"] = "0"           # the 8-wave k_gemm256 instead of the hand-scheduled 4-wave k_gemm_fat
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import ops

samples = []
stop = False
def sampler():
    import re
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            p = re.search(r"Power \(W\):\s*([\d.]+)", o); c = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", o)
            if p and c: samples.append((time.time(), float(p.group(1)), int(c.group(1))))
        except Exception:
            pass
        time.sleep(0.4)

r = lambda *s: torch.randn(*s, device="cuda").half()
if what.startswith("attn49k"):
    S, Hq, Hkv, Dh = 49152, 28, 4, 128
    if what.endswith("_zeros"):
        z = lambda *s: torch.zeros(*s, device="cuda", dtype=torch.float16)
        q, k, v = z(1, S, Hq * Dh), z(1, S, Hkv * Dh), z(1, S, Hkv * Dh)
    else:
        q, k, v = r(1, S, Hq * Dh), r(1, S, Hkv * Dh), r(1, S, Hkv * Dh)
    out = torch.empty_like(q)
    fn = lambda: ops.attention(q, k, v, Hq, Hkv, Dh, Dh ** -0.5, True, out=out); flops = 4.0 * Hq * S * S * Dh * 0.5
elif what == "vit_attn":
    B, S, H, Dh = 512, 577, 16, 64
    q, k, v = r(B, S, H * Dh), r(B, S, H * Dh), r(B, S, H * Dh); out = torch.empty_like(q)
    fn = lambda: ops.attention(q, k, v, H, H, Dh, Dh ** -0.5, False, out=out); flops = 4.0 * B * H * S * S * Dh
elif what == "decode":
    from streamchat_amd import llm as LM
    cfg = LM.Qwen2ConfigLite()
    lm = LM.Qwen2Model(LM.random_qwen2_state_dict(cfg, device="cuda"), cfg, device="cuda", max_seq=49152 + 4096, consume=True)
    lm.reset_cache(); lm.cache_len = 49152
    dg = LM.DecodeGraph(lm, max_new_tokens=4096); dg.start(1); dg.capture()
    fn = lambda: dg.graph.replay(); flops = 16.92e9          # bytes per token: the printed "TFLOP/s" column is GB/s / 1000 for this row
elif what in ("gemm_llm", "gemm_vit", "gemm_llm_8wave", "gemm_llm_zeros", "gemm_llm_const", "gemm_vit_zeros"):
    # *_zeros / *_const (round 5): the SAME launches on operands that do not toggle the datapath - what the schedule delivers when the board's
    # power cap is not the limit (uniform random fp16 is the worst case for switching power)
    M, N, K = (295424, 4096, 1024) if what.startswith("gemm_vit") else (48994, 3584, 18944)
    if what.endswith("_zeros"):
        a, w = torch.zeros(M, K, device="cuda", dtype=torch.float16), torch.zeros(N, K, device="cuda", dtype=torch.float16)
    elif what.endswith("_const"):
        a, w = torch.full((M, K), 0.5, device="cuda", dtype=torch.float16), torch.full((N, K), 0.25, device="cuda", dtype=torch.float16)
    else:
        a, w = r(M, K), r(N, K) * (K ** -0.5)
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    fn = lambda: ops.gemm(a, w, out=out); flops = 2.0 * M * N * K
else:
    fn = None; flops = 0.0
th = threading.Thread(target=sampler, daemon=True); th.start()
t0 = time.time(); rates = []
while time.time() - t0 < secs:
    if fn is None:
        time.sleep(0.5); continue
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    rates.append(flops / (e0.elapsed_time(e1) / 10) / 1e9)
stop = True; th.join()
late = [s for s in samples if s[0] - t0 > secs * 0.4] or samples
pw = sum(s[1] for s in late) / max(1, len(late)); ck = sum(s[2] for s in late) / max(1, len(late))
tf = sum(rates[len(rates) // 2:]) / max(1, len(rates[len(rates) // 2:])) if rates else 0.0
print(f"{what:12s} sustained {tf:7.1f} TFLOP/s   package power {pw:6.0f} W   sclk {ck:5.0f} MHz   ({len(late)} samples over the second half of {secs:.0f} s)")
