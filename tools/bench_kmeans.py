#!/usr/bin/env python3
"""Micro-benchmark of the HIP k-means at the BASELINE sizes (GPU box).  Prints per-iteration time
and achieved algorithmic GB/s:  bytes/iter = T*D*s (one read of X) + 2*K*D*4 (centroids r+w)  [SURVEY §8(d)]."""
import argparse
import json
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=400)
ap.add_argument("--K", type=int, default=5)
ap.add_argument("--D", type=int, default=576 * 3584)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--weighted", type=int, default=0, help="1: pass a weight vector of ones, as weighted_kmeans_feature always does (W[k] takes the sequential-sum path)")
a = ap.parse_args()
torch.manual_seed(0)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
centres = torch.randn(8, a.D, device=dev, generator=g, dtype=torch.float16)
X = centres[torch.randint(0, 8, (a.T,), device=dev, generator=g)]
X = X + (0.5 * torch.randn(a.T, a.D, device=dev, generator=g, dtype=torch.float16))
init = torch.randperm(a.T)[: a.K]
wts = torch.ones(a.T, device=dev) if a.weighted else None
res = []
for rep in range(a.reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    # tol = -1 forces all `iters` Lloyd iterations (assign + update each)
    C, labels, wsum, info = ops.kmeans_fit(X, a.K, init, None, weights=wts, max_iter=a.iters, tol=-1.0)
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1))
def _time(fn, reps=5):
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    return min(out)
# breakdown (round 6): one assign pass alone; one Lloyd iteration (assign pass + update pass); n iterations = assign + (n - 1) fused + update
C0 = X[init.to(dev)].float().contiguous()
ms_assign = _time(lambda: ops.kmeans_assign(X, C0))
ms_fit1 = _time(lambda: ops.kmeans_fit(X, a.K, init, None, weights=wts, max_iter=1, tol=-1.0))
ms = min(res)
per_iter = ms / a.iters
algo = a.T * a.D * 2 + 2 * a.K * a.D * 4
print(json.dumps(dict(T=a.T, K=a.K, D=a.D, iters=a.iters, weighted=a.weighted, ms_total=ms, ms_per_iter=per_iter,
                      algo_GBps_1x=algo / per_iter / 1e6, algo_GBps_2x=(algo + a.T * a.D * 2) / per_iter / 1e6,
                      exit_iter=int(info[0]), all_ms=res, ms_assign_only=ms_assign, ms_fit_1_iter=ms_fit1,
                      ms_per_middle_iter=(ms - ms_fit1) / max(a.iters - 1, 1))))
