"""API surface of the vendored `kmeans_pytorch` (reference kmeans_pytorch/__init__.py:27-209): `kmeans`, `kmeans_predict`,
`pairwise_distance`, `pairwise_cosine`, backed by the HIP k-means kernels.  Never imported by the reference's entry point
(SURVEY.md §0 item 6).  'soft_dtw' (numba CUDA kernels upstream, unreachable from the streaming path) is out of scope."""
import numpy as np
import torch

from . import ops


def _prep(X, distance, device):
    X = X.float().to(device)
    if distance == "cosine":
        X = torch.nn.functional.normalize(X, dim=1)          # argmin Euclid on unit vectors == argmax cosine
    elif distance != "euclidean":
        raise NotImplementedError(f"distance '{distance}' is not supported (euclidean / cosine)")
    return X.contiguous()


def kmeans(X, num_clusters, distance="euclidean", cluster_centers=[], tol=1e-4, tqdm_flag=True, iter_limit=0, device=torch.device("cuda"),
           gamma_for_soft_dtw=0.001, seed=None):
    """returns (cluster_ids [N] int64 cpu, cluster_centers [K, D] float32 cpu) like upstream.  Init = `np.random.choice(N, K,
    replace=False)` under `seed` (upstream `initialize`, :10-24); stop when (sum_k ||dC_k||)^2 < tol or after iter_limit."""
    Xd = _prep(X, distance, device)
    n = Xd.shape[0]
    if isinstance(cluster_centers, list) and len(cluster_centers) == 0:
        if seed is not None:
            np.random.seed(seed)
        init = np.random.choice(n, num_clusters, replace=False)
    else:
        c0 = _prep(torch.as_tensor(cluster_centers), distance, device)
        lab = ops.kmeans_assign(Xd, c0)
        init = np.asarray([int(torch.nonzero(lab == j)[0, 0]) if bool((lab == j).any()) else j for j in range(num_clusters)])
    reseed = np.random.randint(0, n, size=max(iter_limit, 100) * num_clusters)
    C, labels, _, _ = ops.kmeans_fit(Xd, num_clusters, init, reseed, max_iter=iter_limit if iter_limit > 0 else 100, tol=float(np.sqrt(tol)))
    return labels.cpu(), C.cpu()


def kmeans_predict(X, cluster_centers, distance="euclidean", device=torch.device("cuda"), gamma_for_soft_dtw=0.001, tqdm_flag=True):
    return ops.kmeans_assign(_prep(X, distance, device), _prep(torch.as_tensor(cluster_centers), distance, device)).cpu()


def pairwise_distance(data1, data2, device=torch.device("cuda"), tqdm_flag=True):
    """squared Euclidean distances [N, K] (upstream :172-188)"""
    _, d2 = ops.kmeans_assign(_prep(data1, "euclidean", device), _prep(data2, "euclidean", device), return_dist2=True)
    return d2.to(torch.float32)


def pairwise_cosine(data1, data2, device=torch.device("cuda")):
    """1 - cosine similarity [N, K] (upstream :191-209): for unit vectors ||a-b||^2 = 2 - 2cos"""
    _, d2 = ops.kmeans_assign(_prep(data1, "cosine", device), _prep(data2, "cosine", device), return_dist2=True)
    return (d2 / 2).to(torch.float32)
