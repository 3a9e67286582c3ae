#!/usr/bin/env python3
"""Timeline of ONE k_attn_decode launch at a 49 k context from inside the kernel (diagnostic build -DSC_DEC_TRACE: five 100-MHz wall-clock
stamps per wave - entry, first K/V requested, first K landed, stream done, partial written).  Needs SC_LIB=tools/bin/lib_dectrace.so
(tools/build_variant.sh dectrace "-DSC_DEC_TRACE ..." "...")."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from streamchat_amd import ops, _lib

S, G, H, Dh = int(sys.argv[1]) if len(sys.argv) > 1 else 49152, 7, 4, 128
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 64
NC = 6
rows = [torch.randn(S, 2 * H * Dh, device="cuda").half() for _ in range(NC)]
q = torch.randn(1, G * H * Dh, device="cuda").half()
kl = torch.tensor([S], device="cuda", dtype=torch.int32)
qv = q.as_strided((1, G, Dh), (G * H * Dh, Dh, 1))
for i in range(13):
    ck = rows[i % NC]
    ops.attention(qv, ck[:, :H * Dh].unsqueeze(0), ck[:, H * Dh:].unsqueeze(0), H, H, Dh, Dh ** -0.5, causal=False, kv_len=kl, nsplit=ns,
                  q_head_stride=G * Dh, o_head_stride=G * Dh, out_ld=Dh)
torch.cuda.synchronize()
lib = _lib.load()
NWV = int(os.environ.get("DEC_NW", "4"))
buf = np.zeros(2048 * NWV * 8, np.uint64)
lib.sc_dec_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.sc_dec_trace_read(buf.ctypes.data, buf.nbytes) == 0
raw = buf.reshape(2048, NWV, 8)[: ns * H]
t = raw[:, :, :5].astype(np.float64)
hw, xcc = (raw[:, 0, 5] & 0xFFFFFFFF).astype(np.int64), (raw[:, 0, 5] >> 32).astype(np.int64) & 0xF
cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
t0 = t[:, :, 0].min()
us = (t - t0) / 100.0
names = ["entry", "first K/V requested", "first K landed", "stream done", "partial written"]
out = {}
for i, n in enumerate(names):
    v = us[:, :, i].ravel()
    out[n] = dict(min=round(float(v.min()), 2), p50=round(float(np.median(v)), 2), p90=round(float(np.percentile(v, 90)), 2), max=round(float(v.max()), 2))
per_wave_stream = (us[:, :, 3] - us[:, :, 2]).ravel()
out["stream phase per wave (first K landed -> done)"] = dict(min=round(float(per_wave_stream.min()), 2), p50=round(float(np.median(per_wave_stream)), 2), max=round(float(per_wave_stream.max()), 2))
# where the spread comes from: stream-done time by XCD (workgroup id % 8; id = head * nsplit + split), by KV head, by wave of the workgroup
wg = np.arange(ns * H).reshape(H, ns)             # [head][split]
done = us[:, :, 3].reshape(H, ns, NWV)
first = us[:, :, 2].reshape(H, ns, NWV)
out["stream done by XCD (mean, max)"] = {int(x): (round(float(done[(wg % 8) == x].mean()), 2), round(float(done[(wg % 8) == x].max()), 2)) for x in range(8)}
out["stream done by head (mean, max)"] = {int(h): (round(float(done[h].mean()), 2), round(float(done[h].max()), 2)) for h in range(H)}
out["stream done by wave (mean, max)"] = {int(w): (round(float(done[:, :, w].mean()), 2), round(float(done[:, :, w].max()), 2)) for w in range(NWV)}
out["stream done by split octile (mean)"] = [round(float(done[:, i * ns // 8:(i + 1) * ns // 8].mean()), 2) for i in range(8)]
out["first K landed by XCD (mean)"] = {int(x): round(float(first[(wg % 8) == x].mean()), 2) for x in range(8)}
dwg = us[:, :, 3].max(axis=1)                   # per workgroup: when its slowest wave finished streaming
out["stream done by HW XCC id (n, mean, max)"] = {int(x): (int((xcc == x).sum()), round(float(dwg[xcc == x].mean()), 2), round(float(dwg[xcc == x].max()), 2)) for x in sorted(set(xcc.tolist()))}
out["stream done by SE (n, mean)"] = {int(x): (int((se == x).sum()), round(float(dwg[se == x].mean()), 2)) for x in sorted(set(se.tolist()))}
out["workgroups per (xcc, se, sh, cu) slot: max"] = int(max(np.unique(np.stack([xcc, se, sh, cu], 1), axis=0, return_counts=True)[1]))
if len(sys.argv) > 3:
    np.save(sys.argv[3], us)
out["context"], out["nsplit"], out["MB"] = S, ns, round(2 * S * H * Dh * 2 / 1e6, 1)
print(json.dumps(out, indent=1))
