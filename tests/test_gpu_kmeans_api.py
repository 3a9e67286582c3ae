"""GPU: API-surface wrappers of the vendored k-means packages (torch_kmeans.KMeans, kmeans_pytorch.kmeans)."""
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
    res = km(x)
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
