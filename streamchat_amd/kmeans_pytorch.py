"""`kmeans_pytorch` (reference kmeans_pytorch/__init__.py:10-209: `kmeans`, `kmeans_predict`, `pairwise_distance`,
`pairwise_cosine`) on the HIP k-means kernels.  Vendored by the reference but never imported by its entry point (SURVEY.md §0
item 6); the mirror follows the upstream loop so that results agree with it (fixture tests/golden/kmeans_pytorch.npz, produced by
the reference functions):

  initial state    rows `np.random.choice(N, K, replace=False)` under `np.random.seed(seed)` (:10-24); with `cluster_centers` given:
                   the data point closest to each given centre (:77-86)
  loop             assign = first minimum of the squared Euclidean (or 1 - cosine) distance; new centre = mean of the selected rows,
                   an EMPTY cluster takes `X[torch.randint(len(X), (1,))]` (one draw from torch's global CPU generator per empty
                   cluster, drawn only when it happens, :100-107); stop when (sum_k ||dC_k||)^2 < tol, or after `iter_limit` > 0
  returns          (labels of the LAST assignment, centres after the last update), both on the CPU like upstream

'cosine' normalises both operands inside the distance only (:191-209), so centres are means of the RAW rows, as upstream;
'soft_dtw' (numba CUDA kernels upstream, unreachable from the streaming path) is out of scope."""
import numpy as np
import torch

from . import ops


def _prep(X, device):
    return X.float().to(device).contiguous()


def _dist(X, C, distance):
    """(labels [N] int64, dist [N, K] fp64) in upstream's metric: squared Euclid (:172-188) or 1 - cosine (:191-209)."""
    if distance == "euclidean":
        return ops.kmeans_assign(X, C, return_dist2=True)
    if distance == "cosine":                       # 1 - cos = ||a/|a| - b/|b|||^2 / 2: same argmin
        lab, d2 = ops.kmeans_assign(torch.nn.functional.normalize(X, dim=1), torch.nn.functional.normalize(C.float(), dim=1), return_dist2=True)
        return lab, d2 / 2
    raise NotImplementedError(f"distance '{distance}' is not supported (euclidean / cosine)")


def kmeans(X, num_clusters, distance="euclidean", cluster_centers=[], tol=1e-4, tqdm_flag=True, iter_limit=0, device=torch.device("cuda"),
           gamma_for_soft_dtw=0.001, seed=None):
    if tqdm_flag:
        print(f"running k-means on {device}..")
    Xd = _prep(X, device)
    n = Xd.shape[0]
    if type(cluster_centers) == list:                                        # upstream's test (:73): ANY list means "initialise"
        if seed is not None:
            np.random.seed(seed)
        C = Xd[torch.as_tensor(np.random.choice(n, num_clusters, replace=False), device=Xd.device)].clone()
    else:
        _, d = _dist(Xd, _prep(torch.as_tensor(cluster_centers), device), distance)
        C = Xd[torch.argmin(d, dim=0)].clone()                               # closest data point per given centre (:81-84)
    iteration = 0
    while True:
        labels, _ = _dist(Xd, C, distance)
        counts = torch.bincount(labels, minlength=num_clusters).cpu()
        fill = [int(torch.randint(n, (1,))) for k in range(num_clusters) if int(counts[k]) == 0]          # :104-105, in cluster order
        C_new, _, shift2 = ops.kmeans_update(Xd, labels, C, empty="fill", fill_idx=fill if fill else None)
        center_shift = float(shift2.cpu().sqrt().sum())
        C = C_new
        iteration += 1
        if center_shift ** 2 < tol:
            break
        if iter_limit != 0 and iteration >= iter_limit:
            break
    return labels.cpu(), C.cpu()


def kmeans_predict(X, cluster_centers, distance="euclidean", device=torch.device("cuda"), gamma_for_soft_dtw=0.001, tqdm_flag=True):
    return _dist(_prep(X, device), _prep(torch.as_tensor(cluster_centers), device), distance)[0].cpu()


def pairwise_distance(data1, data2, device=torch.device("cuda"), tqdm_flag=True):
    """squared Euclidean distances [N, K] (upstream :172-188)"""
    return _dist(_prep(data1, device), _prep(data2, device), "euclidean")[1].to(torch.float32)


def pairwise_cosine(data1, data2, device=torch.device("cuda")):
    """1 - cosine similarity [N, K] (upstream :191-209)"""
    return _dist(_prep(data1, device), _prep(data2, device), "cosine")[1].to(torch.float32)
