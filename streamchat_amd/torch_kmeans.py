"""`torch_kmeans.KMeans` (reference torch_kmeans/clustering/kmeans.py:24-644) on the HIP k-means kernels.

The reference vendors the package and imports it (`utiles.py:7`) but never calls it; this mirror keeps the
constructor / `forward` / `fit` / `predict` / `fit_predict` surface and the `ClusterResult` tuple, and follows the reference's
algorithm step for step so that labels agree with it (fixture G3b, tests/golden/torch_kmeans.npz, produced by the reference class):

  * init 'rnd'       one `torch.multinomial` over uniform weights for all (batch x num_init) runs (:386-419)
  * init 'k-means++' first centre uniform, every further centre ~ squared distance to the nearest chosen one (:457-515)
  * Lloyd loop       all runs advance together; assign = first minimum of the Euclidean distance (:587-599); update = mean of the
                     assigned rows, an EMPTY cluster's centre becomes the zero vector (utils.py:34-67); stop when the mean-over-
                     clusters centre shift of EVERY run is < tol (:546-560)
  * result           final re-assignment, inertia, best restart per instance (first minimum, :562-570)

RNG: the reference seeds a generator ON x.device; the draws here come from a CPU generator with the same seed, i.e. they equal the
reference's draws for a CPU input (a CUDA generator produces a different stream upstream as well).
Arithmetic: distances / sums run in the SC-KM2 order of kmeans.hip (fp32 products, fp64 totals) instead of torch.cdist + matmul:
labels are identical away from exact ties, centres agree to fp32 rounding.
Not supported (raises): other distances / p-norms, `normalize`, different k per instance (the k_mask path, :531-535)."""
from typing import NamedTuple, Optional
from warnings import warn

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
        if num_init <= 0:
            raise ValueError(f"num_init should be > 0, but got {num_init}.")
        if max_iter <= 0:
            raise ValueError(f"max_iter should be > 0, but got {max_iter}.")
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

    # ---- checks (:136-241) ----
    def _check_k(self, k, n):
        if torch.is_tensor(k):
            vals = set(int(v) for v in k.flatten().tolist())
            if len(vals) > 1:
                raise NotImplementedError("different k per instance is not supported")
            k = vals.pop()
        if k is None:
            if self.n_clusters is None:
                raise ValueError("Did not provide number of clusters k on call and did not specify default 'n_clusters' at initialization.")
            k = self.n_clusters
        if not isinstance(k, int):
            raise TypeError(f"k has to be int, torch.Tensor or None but got {type(k)}.")
        if k >= n:
            raise ValueError(f"Specified 'k' must be smaller than number of samples n={n}, but got: {k}.")
        if k <= 1:
            raise ValueError("Clustering for k=1 is ambiguous.")
        return k

    def _generator(self):
        if self.seed is None:
            return None
        return torch.Generator().manual_seed(self.seed)

    # ---- initial centres: row indices [bs, num_init, k] ----
    def _init_rnd(self, x, k):
        bs, n, _ = x.shape
        idx = torch.multinomial(torch.full((bs * self.num_init, n), 1 / n, dtype=torch.float32), num_samples=k, replacement=False,
                                generator=self._generator())
        return idx.view(bs, self.num_init, k)

    def _init_plus(self, x, k):
        bs, n, _ = x.shape
        m = self.num_init
        if n <= m:
            raise AssertionError(f"Number of samples must be larger than <num_init> but got {n} <= {m}")
        gen = self._generator()
        first = torch.multinomial(torch.full((bs, n), 1 / n, dtype=torch.float32), num_samples=m, replacement=False, generator=gen)
        chosen = first.view(bs, m, 1)
        for nc in range(1, k):
            pot = torch.empty((bs * m, n), dtype=torch.float32)
            for b in range(bs):
                for j in range(m):
                    rows = chosen[b, j].to(x.device)
                    _, d2 = ops.kmeans_assign(x[b], x[b].index_select(0, rows).float(), return_dist2=True)
                    p = d2.min(dim=1).values.clamp_(min=0).to(torch.float32).cpu()
                    p[chosen[b, j]] = 0                                  # a chosen point cannot be drawn again (:503)
                    pot[b * m + j] = p
            nxt = torch.multinomial(pot, 1, generator=gen).view(bs, m, 1)
            chosen = torch.cat([chosen, nxt], dim=2)
        return chosen

    # ---- forward (:243-288) ----
    def forward(self, x, k=None, centers=None, **kwargs):
        if not torch.is_tensor(x):
            raise TypeError(f"x has to be a torch.Tensor but got {type(x)}.")
        if x.dim() < 3:
            raise ValueError(f"input <x> should be at least of shape (BS, N, D) with batch size BS, number of points N and number of dimensions D but got {tuple(x.shape)}.")
        if x.dim() > 3:
            x = x.squeeze()
            return self.forward(x, k=k, centers=centers, **kwargs)
        bs, n, d = x.shape
        kk = self._check_k(k, n)
        m = self.num_init
        xs = [x[b].contiguous() for b in range(bs)]
        if centers is None:
            idx = (self._init_rnd if self.init_method == "rnd" else self._init_plus)(x, kk)
            C = [[xs[b].index_select(0, idx[b, j].to(x.device)).float() for j in range(m)] for b in range(bs)]
        else:
            if not torch.is_tensor(centers):
                raise TypeError(f"centers has to be a torch.Tensor but got {type(centers)}.")
            if centers.dim() == 3:
                if tuple(centers.shape) != (bs, kk, d):
                    raise ValueError(f"centers needs to be of shape ({bs}, {kk}, {d}),but got {tuple(centers.shape)}.")
                if m > 1:
                    warn(f"Specified num_init={m} > 1 but provided only 1 center configuration per instance. Using same center configuration for all {m} runs.")
                centers = centers[:, None].expand(bs, m, kk, d)
            elif centers.dim() != 4 or tuple(centers.shape) != (bs, m, kk, d):
                raise ValueError(f"centers have unsupported shape of {tuple(centers.shape)} instead of ({bs}, {m}, {kk}, {d}).")
            C = [[centers[b, j].to(device=x.device, dtype=torch.float32).contiguous() for j in range(m)] for b in range(bs)]
        # ---- Lloyd: all runs step together, stop when every run's mean centre shift is < tol (:537-560) ----
        for it in range(self.max_iter):
            shifts = []
            for b in range(bs):
                for j in range(m):
                    lab = ops.kmeans_assign(xs[b], C[b][j])
                    C[b][j], _, s2 = ops.kmeans_update(xs[b], lab, C[b][j], empty="zero")
                    shifts.append(s2)
            if self.tol is not None:
                shift = torch.stack(shifts).cpu().sqrt().mean(dim=1)           # one host read per iteration (the reference's `.all()` syncs too)
                if bool((shift < self.tol).all()):
                    if self.verbose:
                        print(f"Full batch converged at iteration {it + 1}/{self.max_iter} with center shifts = {shift.view(-1, m).mean(-1)}.")
                    break
        # ---- best restart per instance by inertia (:562-570) ----
        labels, cents, inert = [], [], []
        for b in range(bs):
            runs = []
            for j in range(m):
                lab, d2 = ops.kmeans_assign(xs[b], C[b][j], return_dist2=True)
                runs.append((lab, d2.gather(1, lab[:, None]).sum()))
            best = int(torch.argmin(torch.stack([r[1] for r in runs]).cpu()))
            labels.append(runs[best][0]); cents.append(C[b][best].to(x.dtype)); inert.append(runs[best][1].to(x.dtype if x.dtype != torch.float16 else torch.float32))
        return ClusterResult(torch.stack(labels), torch.stack(cents), torch.stack(inert), x, x, torch.full((bs,), kk, dtype=torch.long, device=x.device))

    __call__ = forward

    def fit(self, x, k=None, centers=None, **kwargs):
        self._result = self.forward(x, k=k, centers=centers, **kwargs)
        return self

    def predict(self, x, **kwargs):
        assert self.is_fitted
        return torch.stack([ops.kmeans_assign(x[b].contiguous(), self._result.centers[b].float()) for b in range(x.shape[0])])

    def fit_predict(self, x, k=None, centers=None, **kwargs):
        return self.forward(x, k=k, centers=centers, **kwargs).labels
