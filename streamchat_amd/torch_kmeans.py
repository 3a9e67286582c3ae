"""API surface of the vendored `torch_kmeans.KMeans` (reference torch_kmeans/clustering/kmeans.py:24-644), backed by the HIP
k-means kernels.  The reference imports it (`utiles.py:7`) but never calls it, so this mirrors the
constructor / `forward` / `fit` / `predict` / `fit_predict` call surface and the `ClusterResult` tuple for code that does.

Supported: Euclidean (p = 2) distance, init 'rnd' and 'k-means++', `num_init` restarts (best inertia wins), explicit
`centers`.  Not supported (raises): other distances / p-norms, per-instance `k` tensors with different values, `normalize`.
Convergence uses the kernels' criterion (sum_k ||dC_k||_2 < tol) instead of torch_kmeans' mean relative shift."""
from typing import NamedTuple, Optional

import torch

from . import ops


class ClusterResult(NamedTuple):
    labels: torch.Tensor
    centers: torch.Tensor
    inertia: torch.Tensor
    x_org: torch.Tensor
    x_norm: torch.Tensor
    k: torch.Tensor
    soft_assignment: Optional[torch.Tensor] = None


class KMeans:
    INIT_METHODS = ["rnd", "k-means++"]

    def __init__(self, init_method="rnd", num_init=8, max_iter=100, distance=None, p_norm=2, tol=1e-4, normalize=None, n_clusters=8,
                 verbose=True, seed=123, **kwargs):
        self.init_method = init_method.lower()
        if self.init_method not in self.INIT_METHODS:
            raise ValueError(f"unknown <init_method>: {init_method}. Please choose one of {self.INIT_METHODS}")
        if num_init <= 0 or max_iter <= 0:
            raise ValueError("num_init and max_iter should be > 0")
        if p_norm != 2 or distance is not None or normalize not in (None, False):
            raise NotImplementedError("streamchat_amd.torch_kmeans.KMeans: only the default Euclidean distance without normalisation")
        self.num_init, self.max_iter, self.tol, self.n_clusters, self.verbose, self.seed = num_init, max_iter, tol, n_clusters, verbose, seed
        self._result = None

    @property
    def is_fitted(self):
        return self._result is not None

    @property
    def num_clusters(self):
        return None if self._result is None else self._result.k

    def _init_centers(self, x, k, gen):
        n = x.shape[0]
        if self.init_method == "rnd":
            return torch.randperm(n, generator=gen)[:k]
        idx = [int(torch.randint(0, n, (1,), generator=gen))]
        for _ in range(1, k):                                              # k-means++: next centre ~ min squared distance
            _, d2 = ops.kmeans_assign(x, x[torch.tensor(idx, device=x.device)].float(), return_dist2=True)
            p = d2.min(dim=1).values.clamp(min=0).cpu()
            idx.append(int(torch.multinomial(p / p.sum(), 1, generator=gen)) if float(p.sum()) > 0 else int(torch.randint(0, n, (1,), generator=gen)))
        return torch.tensor(idx)

    def forward(self, x, k=None, centers=None, **kwargs):
        if x.dim() != 3:
            raise ValueError("input <x> should be of shape (BS, N, D)")
        bs, n, d = x.shape
        kk = self.n_clusters if k is None else (int(k) if not torch.is_tensor(k) else int(k.flatten()[0]))
        if torch.is_tensor(k) and len(set(k.flatten().tolist())) > 1:
            raise NotImplementedError("different k per instance is not supported")
        gen = torch.Generator().manual_seed(self.seed if self.seed is not None else 0)
        labels, cents, inert = [], [], []
        for b in range(bs):
            xb = x[b].contiguous()
            best = None
            inits = [None] if centers is not None else [self._init_centers(xb, kk, gen) for _ in range(self.num_init)]
            for init in inits:
                if init is None:                                           # user-supplied centres: one Lloyd run from them
                    c0 = centers[b].to(x.device).float()
                    c0 = c0[0] if c0.dim() == 3 else c0
                    lab0 = ops.kmeans_assign(xb, c0)
                    init = torch.stack([torch.nonzero(lab0 == j)[0, 0] if (lab0 == j).any() else torch.tensor(j, device=x.device) for j in range(kk)]).cpu()
                C, lab, _, _ = ops.kmeans_fit(xb, kk, init, torch.randint(0, n, (self.max_iter * kk,), generator=gen), max_iter=self.max_iter, tol=self.tol)
                lab, d2 = ops.kmeans_assign(xb, C, return_dist2=True)
                inertia = d2.gather(1, lab[:, None]).sum()
                if best is None or float(inertia) < float(best[2]):
                    best = (lab, C, inertia)
            labels.append(best[0]); cents.append(best[1].to(x.dtype)); inert.append(best[2].to(torch.float32))
        self._result = ClusterResult(torch.stack(labels), torch.stack(cents), torch.stack(inert), x, x, torch.full((bs,), kk, dtype=torch.long, device=x.device))
        return self._result

    __call__ = forward

    def fit(self, x, k=None, centers=None, **kwargs):
        self.forward(x, k=k, centers=centers, **kwargs)
        return self

    def predict(self, x, **kwargs):
        assert self.is_fitted
        return torch.stack([ops.kmeans_assign(x[b].contiguous(), self._result.centers[b].float()) for b in range(x.shape[0])])

    def fit_predict(self, x, k=None, centers=None, **kwargs):
        return self.forward(x, k=k, centers=centers, **kwargs).labels
