"""kmeans.hip km2_pass (round 6, the default for fp16 / 2 <= K <= 8 / T <= 448; SC_KM_FUSED=0 switches it off): one pass over X per Lloyd
iteration from an LDS-resident slab.  Must equal the two-pass lane-mapped kernels bit for bit - labels, centroids, cluster weights, exit
iteration - incl. the empty-cluster reseed and the weighted sums (both implement SC-KM2; test_gpu_kmeans.py compares each with the oracle)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

PROBE = r'''
import sys, numpy as np, torch
from streamchat_amd import ops
out = {}
g = torch.Generator(device="cuda").manual_seed(3)
for name, (T, K, D, weighted, reseed) in dict(merge=(400, 5, 512 * 96, False, False), c1=(64, 8, 512 * 64, False, False), odd=(333, 5, 512 * 40 + 64, True, False),
                                               small=(23, 8, 512 * 8, False, False), empty=(90, 5, 512 * 16, False, True), mid=(200, 8, 64 * 50, True, False)).items():
    centres = torch.randn(6, D, device="cuda", generator=g)
    X = (centres[torch.randint(0, 6, (T,), device="cuda", generator=g)] + 0.6 * torch.randn(T, D, device="cuda", generator=g)).half()
    init = list(range(0, T, T // K))[:K]
    if reseed:                       # two identical initial rows: one of the two clusters stays empty after the first assign -> reseed path
        X[init[1]] = X[init[0]]
    w = (0.5 + torch.rand(T, device="cuda", generator=g)) if weighted else None
    C, labels, wsum, info = ops.kmeans_fit(X, K, init, [7, 3, 11, 5] * 10, weights=w, max_iter=10, tol=1e-4)
    out[name + "_C"] = C.cpu().numpy().view(np.uint32); out[name + "_labels"] = labels.cpu().numpy()
    out[name + "_wsum"] = wsum.cpu().numpy().view(np.uint32); out[name + "_info"] = info.cpu().numpy()
np.savez(sys.argv[1], **out)
'''


def _run(tmp, fused):
    out = os.path.join(tmp, f"km_{fused}.npz")
    env = dict(os.environ, PYTHONPATH=ROOT, SC_KM_FUSED=str(fused))
    subprocess.run([sys.executable, "-c", PROBE, out], check=True, env=env, cwd=ROOT, timeout=600)
    return np.load(out)


def test_fused_pass_equals_two_kernel_path_bitwise(tmp_path):
    a, b = _run(str(tmp_path), 0), _run(str(tmp_path), 1)
    assert sorted(a.files) == sorted(b.files) and len(a.files) == 24
    for k in a.files:
        assert np.array_equal(a[k], b[k]), f"{k}: fused pass differs from the two-kernel path"
    assert int(a["empty_info"][2]) > 0, "the reseed case did not consume a reseed row"
