#!/usr/bin/env python3
"""Phase timeline of km2_pass (kmeans.hip) from inside the kernel: a -DKM2_TRACE build (tools/build_km_variant.sh trace -DKM2_TRACE, loaded with
SC_LIB=tools/bin/lib_trace.so) stamps the shader-cycle counter of wave 0 at every phase boundary and sums the deltas over all workgroups and slices.
Prints, per pass kind, the share of wave 0's time per phase and the cycles per slice."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import ops, _lib

T, K, D = int(os.environ.get("T", 400)), int(os.environ.get("K", 5)), 576 * 3584
lib = _lib.load()
rd = lib.sc_km2_trace_read
buf = (ctypes.c_ulonglong * 8)()
g = torch.Generator(device="cuda").manual_seed(0)
centres = torch.randn(8, D, device="cuda", generator=g, dtype=torch.float16)
X = centres[torch.randint(0, 8, (T,), device="cuda", generator=g)] + 0.5 * torch.randn(T, D, device="cuda", generator=g, dtype=torch.float16)
init = torch.randperm(T)[:K]
C0 = X[init.cuda()].float().contiguous()
names = ["centroid loads issue", "wait own DMA", "B accumulate", "B barrier 1", "final sum + barrier 2", "C shift + cr", "D assign (+ next DMA issue)", "-"]
nslices = (D // 2048) * 32
def run(label, fn):
    fn(); torch.cuda.synchronize(); rd(buf, 1)
    fn(); torch.cuda.synchronize(); rd(buf, 1)
    v = list(buf); tot = sum(v) or 1
    print(json.dumps(dict(label=label, T=T, K=K, cycles_per_slice=round(tot / nslices), share={n: round(x / tot, 3) for n, x in zip(names, v)})))
run("assign only (MODE 2)", lambda: ops.kmeans_assign(X, C0))
run("1 iteration (MODE 2 + MODE 1)", lambda: ops.kmeans_fit(X, K, init, None, max_iter=1, tol=-1.0))
run("3 iterations (MODE 2 + 2 x MODE 3 + MODE 1)", lambda: ops.kmeans_fit(X, K, init, None, max_iter=3, tol=-1.0))
