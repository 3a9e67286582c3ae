"""GPU: the vendored k-means packages' API (torch_kmeans.KMeans, kmeans_pytorch.kmeans) on the HIP kernels — against golden
vectors produced by the REFERENCE's own classes / functions (tools/make_golden_r02.py: G3b tests/golden/torch_kmeans.npz and
kmeans_pytorch.npz), plus surface / property checks."""
import json
import os

import numpy as np
import pytest
import torch

from streamchat_amd import kmeans_pytorch as KP, torch_kmeans as TK

pytestmark = pytest.mark.gpu


def _blobs(n, d, k, seed):
    g = torch.Generator().manual_seed(seed)
    c = torch.randn(k, d, generator=g) * 5
    y = torch.randint(0, k, (n,), generator=g)
    return (c[y] + 0.3 * torch.randn(n, d, generator=g)), y


def _same_partition(a, b):
    m = {}
    return all(m.setdefault(int(x), int(y)) == int(y) for x, y in zip(a, b)) and len(set(m.values())) == len(m)


def test_torch_kmeans_surface():
    x0, y0 = _blobs(200, 16, 4, 0)
    x1, y1 = _blobs(200, 16, 4, 1)
    x = torch.stack([x0, x1]).cuda()
    km = TK.KMeans(n_clusters=4, num_init=4, max_iter=50, seed=123, verbose=False)
    assert not km.is_fitted and km(x).labels.shape == (2, 200) and not km.is_fitted         # forward alone does not fit (kmeans.py:243-288)
    res = km.fit(x)._result
    assert res.labels.shape == (2, 200) and res.centers.shape == (2, 4, 16) and res.inertia.shape == (2,) and res.k.tolist() == [4, 4]
    assert _same_partition(res.labels[0].cpu(), y0) and _same_partition(res.labels[1].cpu(), y1)
    assert torch.equal(km.predict(x), res.labels) and torch.equal(km.fit_predict(x), res.labels) and km.is_fitted
    res2 = TK.KMeans(n_clusters=4, init_method="k-means++", num_init=1, verbose=False)(x)
    assert float(res2.inertia[0]) <= 1.5 * float(res.inertia[0])
    with pytest.raises(NotImplementedError):
        TK.KMeans(p_norm=1)
    with pytest.raises(ValueError):
        TK.KMeans(init_method="bogus")


def test_kmeans_pytorch_surface():
    x, y = _blobs(300, 24, 5, 2)
    ids, centers = KP.kmeans(x, 5, distance="euclidean", seed=0, tqdm_flag=False, device=torch.device("cuda"))
    assert ids.device.type == "cpu" and centers.shape == (5, 24) and _same_partition(ids, y)
    assert torch.equal(KP.kmeans_predict(x, centers, device=torch.device("cuda"), tqdm_flag=False), ids)
    d2 = KP.pairwise_distance(x, centers)
    torch.testing.assert_close(d2.cpu(), torch.cdist(x, centers) ** 2, rtol=1e-4, atol=1e-3)
    pc = KP.pairwise_cosine(x, centers)
    ref = 1 - torch.nn.functional.normalize(x, dim=1) @ torch.nn.functional.normalize(centers, dim=1).T
    torch.testing.assert_close(pc.cpu(), ref, rtol=1e-4, atol=1e-4)
    ids_c, _ = KP.kmeans(x, 5, distance="cosine", seed=0, tqdm_flag=False)
    assert ids_c.shape == (300,)
    with pytest.raises(NotImplementedError):
        KP.kmeans(x, 5, distance="soft_dtw")


G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", ["rnd1", "rnd4", "pp1", "pp3", "rnd1_iter3", "given"])
def test_torch_kmeans_matches_reference_fixture(name):
    """labels bit-exact, centres / inertia to fp32 rounding vs the reference class run on the same inputs (G3b)."""
    d = np.load(os.path.join(G, "torch_kmeans.npz"))
    x = torch.from_numpy(d[name + ".x"]).cuda()
    kw = json.loads(str(d[name + ".kw"]))
    km = TK.KMeans(n_clusters=int(d[name + ".k"]), seed=123, verbose=False, **kw)
    centers = torch.from_numpy(d["given.centers0"]).cuda() if name == "given" else None
    r = km.fit(x, centers=centers)._result
    assert np.array_equal(r.labels.cpu().numpy(), d[name + ".labels"])
    np.testing.assert_allclose(r.centers.cpu().numpy(), d[name + ".centers"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r.inertia.cpu().numpy(), d[name + ".inertia"], rtol=1e-4)
    if name == "given":
        assert np.all(r.centers.cpu().numpy()[0, 3] == 0)              # the empty cluster's centre is the zero vector (utils.py:66)
    else:
        assert np.array_equal(km.predict(x + 0.01).cpu().numpy(), d[name + ".predict"])


@pytest.mark.parametrize("name", ["euclid", "cosine", "limit2"])
def test_kmeans_pytorch_matches_reference_fixture(name):
    d = np.load(os.path.join(G, "kmeans_pytorch.npz"))
    x = torch.from_numpy(d[name + ".x"])
    dist = str(d[name + ".distance"])
    ids, cent = KP.kmeans(x, int(d[name + ".k"]), distance=dist, tqdm_flag=False, iter_limit=int(d[name + ".iter_limit"]), seed=int(d[name + ".seed"]))
    assert np.array_equal(ids.numpy(), d[name + ".ids"])
    np.testing.assert_allclose(cent.numpy(), d[name + ".centers"], rtol=1e-5, atol=1e-5)
    assert np.array_equal(KP.kmeans_predict(x + 0.02, cent, distance=dist, tqdm_flag=False).numpy(), d[name + ".predict"])


def test_kmeans_pytorch_resume_empty_and_pairwise_fixtures():
    d = np.load(os.path.join(G, "kmeans_pytorch.npz"))
    ids, cent = KP.kmeans(torch.from_numpy(d["resume.x"]), 3, cluster_centers=torch.from_numpy(d["resume.c0"]), tqdm_flag=False)
    assert np.array_equal(ids.numpy(), d["resume.ids"])
    np.testing.assert_allclose(cent.numpy(), d["resume.centers"], rtol=1e-5, atol=1e-5)
    torch.manual_seed(int(d["empty.torch_seed"]))                          # the refill row of the empty cluster comes from torch's global RNG
    ids, cent = KP.kmeans(torch.from_numpy(d["empty.x"]), 3, tqdm_flag=False, seed=int(d["empty.seed"]), iter_limit=4)
    assert np.array_equal(ids.numpy(), d["empty.ids"])
    np.testing.assert_allclose(cent.numpy(), d["empty.centers"], rtol=1e-5, atol=1e-5)
    a, b = torch.from_numpy(d["pair.a"]), torch.from_numpy(d["pair.b"])
    np.testing.assert_allclose(KP.pairwise_distance(a, b, tqdm_flag=False).cpu().numpy(), d["pair.dist"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(KP.pairwise_cosine(a, b).cpu().numpy(), d["pair.cos"], rtol=1e-5, atol=1e-5)


def test_kmeans_update_matches_oracle_formula():
    """sc_kmeans_update alone: weighted means in ascending row order, zero / fill policies, fp64 shift."""
    g = torch.Generator().manual_seed(3)
    X = torch.randn(50, 96, generator=g)
    lab = torch.randint(0, 3, (50,), generator=g)                          # cluster 3 stays empty
    w = torch.rand(50, generator=g) + 0.5
    C0 = torch.randn(4, 96, generator=g)
    for mode, fill in (("zero", None), ("fill", [17])):
        C, ws, s2 = __import__("streamchat_amd.ops", fromlist=["x"]).kmeans_update(X.cuda(), lab.cuda(), C0.cuda(), weights=w.cuda(), empty=mode, fill_idx=fill)
        ref = torch.zeros(4, 96)
        for k in range(3):
            acc = torch.zeros(96)
            for t in range(50):
                if lab[t] == k:
                    acc = acc + w[t] * X[t]
            ref[k] = acc / w[lab == k].sum()
        ref[3] = X[17] if mode == "fill" else 0
        np.testing.assert_allclose(C.cpu().numpy(), ref.numpy(), rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(s2.cpu().numpy(), ((C0 - C.cpu()).double() ** 2).sum(1).numpy(), rtol=1e-6)
        np.testing.assert_allclose(ws.cpu().numpy()[:3], [float(w[lab == k].sum()) for k in range(3)], rtol=1e-6)
