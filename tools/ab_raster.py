#!/usr/bin/env python3
"""One leg of the tile-order A/B of k_gemm_fat (SC_GEMM_RASTER=0|2 in the environment): sustained launches of the gate/up
projection of the 48 994-token prefill (fused SwiGLU epilogue, random operands) while sampling package power and shader clock; `check` compares
the result with the default order bit for bit (sha1 of the output written by the other leg)."""
import hashlib, json, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import ops
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
M, N, K = 48994, 37888, 3584
g = torch.Generator(device="cuda").manual_seed(1)
a = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).half()
w = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) * K ** -0.5).half()
out = torch.empty(M, N // 2, device="cuda", dtype=torch.float16)
fn = lambda: ops.gemm(a, w, None, None, "swiglu", out=out)
fn(); torch.cuda.synchronize()
digest = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()
samples, stop = [], False
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            p = re.search(r"Power \(W\):\s*([\d.]+)", o); c = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", o)
            if p and c: samples.append((time.time(), float(p.group(1)), int(c.group(1))))
        except Exception:
            pass
        time.sleep(0.3)
th = threading.Thread(target=sampler, daemon=True); th.start()
t0, rates = time.time(), []
while time.time() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    rates.append(2.0 * M * N * K / (e0.elapsed_time(e1) / 10) / 1e9)
stop = True; th.join()
late = [s for s in samples if s[0] - t0 > secs * 0.4] or samples
print(json.dumps(dict(raster=os.environ.get("SC_GEMM_RASTER", "0"), shape="llm.gateup+swiglu 48994x37888x3584", TF=round(sum(rates[len(rates) // 2:]) / max(1, len(rates[len(rates) // 2:])), 1),
                      package_W=round(sum(s[1] for s in late) / max(1, len(late))), sclk_MHz=round(sum(s[2] for s in late) / max(1, len(late))), out_sha1=digest[:16])))
