"""GPU: the drop-in boundary proven through the exact text a maintainer of the reference would paste.

INTEGRATION.md section B shows the ctypes stub that replaces the body of the reference's `weighted_kmeans_torch` (utiles.py:294-318).  This
test extracts that code block from INTEGRATION.md VERBATIM, executes it (nothing of streamchat_amd's Python is imported: only the
C-ABI library is loaded, by the stub itself) and runs it on the reference-generated k-means fixtures under the RNG seeds the fixtures
were drawn with — the stub draws `torch.randperm` / `random.randint` exactly where the reference does (:295, :313)."""
import os
import random
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
pytestmark = pytest.mark.gpu


def _stub_source():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = md[md.index("## B."):md.index("## C.")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert len(blocks) == 1 and "def weighted_kmeans_torch(" in blocks[0]
    return blocks[0]


@pytest.fixture(scope="module")
def stub(hip_lib):
    os.environ["STREAMCHAT_HIP_LIB"] = os.path.join(ROOT, "streamchat_amd", "libstreamchat_hip.so")
    ns = {}
    exec(compile(_stub_source(), "INTEGRATION.md#B", "exec"), ns)
    return ns["weighted_kmeans_torch"]


@pytest.mark.parametrize("case", ["kmeans_01", "kmeans_02", "kmeans_07", "kmeans_08"])     # C1-like, merge-like, weighted, empty-cluster reseed
def test_integration_md_stub_reproduces_the_reference_fixtures(stub, case):
    d = np.load(os.path.join(G, case + ".npz"))
    X = torch.from_numpy(d["X"])
    T, K, seed = X.shape[0], int(d["K"]), int(d["seed"])
    w = torch.from_numpy(d["weights"]).cuda() if "weights" in d.files else None
    torch.manual_seed(seed)
    random.seed(seed)                                   # tools/make_golden.py drew init_idx / the reseed stream under these seeds
    C, labels, wsum, exit_iter = stub(X.view(T, -1).cuda(), K, weights=w)
    assert labels.cpu().numpy().tolist() == d["labels"].tolist()
    assert exit_iter == int(d["exit_iter"])
    np.testing.assert_allclose(C.cpu().numpy(), d["centroids"].reshape(K, -1), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(wsum.cpu().numpy(), d["wsum"], rtol=1e-6)


def test_stub_argtypes_match_the_header():
    """the stub's argtypes list has one entry per parameter of sc_kmeans_fit in include/streamchat_hip.h"""
    src = open(os.path.join(ROOT, "include", "streamchat_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    proto = re.search(r"int\s+sc_kmeans_fit\s*\((.*?)\)\s*;", src, flags=re.S).group(1)
    n_params = len([p for p in proto.split(",") if p.strip()])
    stub = _stub_source()
    argt = re.search(r"_sc\.sc_kmeans_fit\.argtypes\s*=\s*\[(.*?)\]", stub, flags=re.S).group(1)
    assert len(re.findall(r"ctypes\.c_\w+", argt)) == n_params
