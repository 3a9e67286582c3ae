"""TEST INFRASTRUCTURE — CPU oracle for the StreamChat hot path (never imported by the product).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
`kmeans_oracle.c` restates utiles.py:291-330 (k-means) and the cosine / flat-L2 top-k of
utiles.py:732-740 / memory_bank/memory_retrieval/local_doc_qa.py:270; `torch_ref.py` restates the
third-party ViT / BERT / Qwen2 arithmetic (transformers) in plain fp32 PyTorch on CPU.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libsc_oracle.so")
    src = os.path.join(_HERE, "kmeans_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libsc_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.sc_oracle_kmeans_fit.restype = ctypes.c_int
    return _LIB


_DT = {"float16": 0, "bfloat16": 1, "float32": 2}


def _as_raw(X):
    """numpy array (float16/float32) or (uint16 bf16 bits, 'bfloat16') -> (contiguous array, dtype code)"""
    if isinstance(X, tuple):
        arr, name = X
        return np.ascontiguousarray(arr), _DT[name]
    X = np.ascontiguousarray(X)
    return X, _DT[str(X.dtype)]


def kmeans_fit(X, K, init_idx, reseed_idx=None, weights=None, max_iter=10, tol=1e-4, trace=False):
    """Oracle for weighted_kmeans_torch (utiles.py:294-318). X: [T, D] (f16/f32 numpy, or (u16, 'bfloat16')).
    Returns dict(centroids[K,D] f32, labels[T] i64, wsum[K] f32, iters, trace[iters+1,T])."""
    Xr, dt = _as_raw(X)
    T, D = Xr.shape
    init_idx = np.ascontiguousarray(init_idx, np.int32)
    C = np.empty((K, D), np.float32)
    labels = np.empty(T, np.int64)
    wsum = np.empty(K, np.float32)
    iters = ctypes.c_int(0)
    tr = np.full((max_iter, T), -1, np.int32) if trace else None
    rs = None if reseed_idx is None else np.ascontiguousarray(reseed_idx, np.int32)
    w = None if weights is None else np.ascontiguousarray(weights, np.float32)
    p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    rc = lib().sc_oracle_kmeans_fit(p(Xr), dt, T, ctypes.c_int64(D), K, p(w), p(init_idx), p(rs),
                                    0 if rs is None else len(rs), max_iter, ctypes.c_float(tol),
                                    p(C), p(labels), p(wsum), ctypes.byref(iters), p(tr))
    if rc != 0:
        raise RuntimeError(f"sc_oracle_kmeans_fit failed rc={rc}")
    out = dict(centroids=C, labels=labels, wsum=wsum, iters=iters.value)
    if trace:
        out["trace"] = tr[: iters.value + 1]
    return out


def kmeans_dist2(X, C):
    Xr, dt = _as_raw(X)
    T, D = Xr.shape
    C = np.ascontiguousarray(C, np.float32)
    K = C.shape[0]
    out = np.empty((T, K), np.float64)
    lib().sc_oracle_kmeans_dist2(Xr.ctypes.data_as(ctypes.c_void_p), dt, T, ctypes.c_int64(D), K,
                                 C.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    return out


def topk(q, docs, k, metric="cos"):
    q = np.ascontiguousarray(q, np.float32).reshape(-1)
    docs = np.ascontiguousarray(docs, np.float32)
    M, d = docs.shape
    k = min(k, M)
    idx = np.empty(k, np.int32)
    sc = np.empty(k, np.float32)
    lib().sc_oracle_topk(q.ctypes.data_as(ctypes.c_void_p), docs.ctypes.data_as(ctypes.c_void_p), M, d, k,
                         0 if metric == "cos" else 1, idx.ctypes.data_as(ctypes.c_void_p),
                         sc.ctypes.data_as(ctypes.c_void_p))
    return idx, sc
