"""gemv.hip round 5: the shape-specialised straight-line decode projections (k_gemv_u, k_decode_qkv_u) against the generic kernels they
replace on the Qwen2-7B shapes - bit for bit (same per-lane summation order, same norm reduction order), and against fp32 torch."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(tmp, generic):
    out = os.path.join(tmp, f"gemv_{'generic' if generic else 'spec'}.npz")
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("SC_GEMV_GENERIC", None); env.pop("SC_SKINNY_GENERIC", None)
    if generic:
        env["SC_GEMV_GENERIC"] = env["SC_SKINNY_GENERIC"] = "1"
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_gemv_probe.py"), out], check=True, env=env, cwd=ROOT, timeout=600)
    return np.load(out)


def test_specialised_equals_generic_bitwise(tmp_path):
    a, b = _run(str(tmp_path), True), _run(str(tmp_path), False)
    assert sorted(a.files) == sorted(b.files) and len(a.files) >= 24
    for k in a.files:
        assert np.array_equal(a[k], b[k]), f"{k}: specialised kernel differs from the generic one in {int((a[k] != b[k]).sum())} of {a[k].size} outputs"


def test_specialised_down_projection_vs_fp32():
    from streamchat_amd import ops
    g = torch.Generator(device="cuda").manual_seed(11)
    w = (torch.randn(3584, 18944, device="cuda", generator=g) * 0.02).half()
    x = (torch.randn(18944, device="cuda", generator=g) * 0.5).half()
    r = torch.randn(3584, device="cuda", generator=g).half()
    y = ops.gemv(w, x, None, residual=r).float()
    ref = w.float() @ x.float() + r.float()
    assert (y - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
