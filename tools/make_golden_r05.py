#!/usr/bin/env python3
"""Round-5 golden vectors from the reference's OWN functions (authoring container only; same rules as tools/make_golden.py:
functions are AST-extracted / imported from /root/reference and EXECUTED here; only inputs and expected outputs go to tests/golden/).

  kmeans_near_tie.npz      the arg-min footnote of DESIGN section 2: a constructed row whose two fp32 squared distances
                           differ by ONE ulp and whose fp32 square roots are EQUAL.  The reference takes `argmin` of the `.sqrt()`-ed sums
                           (utiles.py:299-302) -> first index; the oracle / the HIP kernel take the arg-min of the squared distance -> the
                           strictly smaller one.  The fixture records what the reference's own function returns, so the divergence is pinned
                           by data instead of by a paragraph.
  composed_ref_trace.npz    one composed run - HF tiny-CLIP encode -> the reference's OWN updating_memory_buffer
  composed_ref_trace.json  (inference_streaming_longva_v2.py:267-378, with its own forgetting sampler, chunking, weighted_kmeans_feature and
                           fast_building_memory_tree_summarize_token) over two segments -> the reference's OWN
                           fast_search_tree_multi_modal_with_embedding (utiles.py:685-788) with the HF tiny-BERT as embedding model:
                           short-memory frame indices, merge labels, tree, retrieved path.  The HIP side of tests/test_gpu_composed_ref.py
                           runs the package's host functions on its own kernels and must land on these outputs.
"""
import json
import os
import random
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import make_golden as MG          # noqa: E402

REF, OUT = MG.REF, MG.OUT


# ---------------------------------------------------------------------------------------------------------------------------
# near-tie: x = (0, 0); A, B in fp32 with fl(fl(a1^2) + fl(a2^2)) = fl(b..) + 1 ulp and equal fp32 square roots
# ---------------------------------------------------------------------------------------------------------------------------
def find_near_tie(seed=7, n=400000):
    f32 = np.float32
    rng = np.random.default_rng(seed)
    ang, r = rng.uniform(0, 2 * np.pi, n), rng.uniform(1.0, 1.4, n)
    c = np.stack([r * np.cos(ang), r * np.sin(ang)], 1).astype(f32)
    d = f32(0) - c
    s = ((d[:, 0] * d[:, 0]).astype(f32) + (d[:, 1] * d[:, 1]).astype(f32)).astype(f32)      # D = 2: every summation order gives this value
    ok = (s >= 1) & (s < 2)
    c, s = c[ok], s[ok]
    bits = s.view(np.int32)
    o = np.argsort(bits, kind="stable")
    b = bits[o]
    for t in np.nonzero((b[1:] - b[:-1]) == 1)[0]:
        lo, hi = o[t], o[t + 1]
        cosang = float(c[lo] @ c[hi]) / float(np.linalg.norm(c[lo]) * np.linalg.norm(c[hi]))
        if np.sqrt(s[hi]) == np.sqrt(s[lo]) and cosang < -0.2:
            return c[hi], c[lo], s[hi], s[lo]          # A: the LARGER squared distance (gets index 0), B: the smaller one (index 1)
    raise RuntimeError("no near-tie found")


def gen_near_tie(ns):
    A, B, sA, sB = find_near_tie()
    assert sB < sA and np.sqrt(sA) == np.sqrt(sB) and sA.view(np.int32) - sB.view(np.int32) == 1
    T, K, seed = 13, 2, 5
    torch.manual_seed(seed)
    perm = torch.randperm(T)                       # what utiles.py:295 will draw: rows perm[0], perm[1] are the initial centroids
    rowA, rowB = int(perm[0]), int(perm[1])
    tie_row = int(perm[2])
    g = np.random.default_rng(3)
    X = np.zeros((T, 1, 2), np.float32)
    others = [i for i in range(T) if i not in (rowA, rowB, tie_row)]
    for n, i in enumerate(others):                 # five rows tightly around A, five around B
        X[i, 0] = (A if n % 2 == 0 else B) + g.normal(0, 0.01, 2).astype(np.float32)
    X[rowA, 0], X[rowB, 0] = A, B                  # X[tie_row] stays (0, 0)
    Xt = torch.from_numpy(X.copy())
    random.seed(seed)
    reseed_stream = [random.randint(0, T - 1) for _ in range(10 * K)]
    torch.manual_seed(seed); random.seed(seed)
    red, labels = ns["weighted_kmeans_feature"](Xt.clone(), K)
    init_idx = perm[:K].clone()
    C, lab2, wsum, it, trace = MG.kmeans_trace(Xt.view(T, -1), K, init_idx, reseed_stream)
    assert torch.equal(labels, lab2) and torch.equal(red.reshape(K, -1), C)
    assert int(trace[0][tie_row]) == 0, "the reference's sqrt-then-argmin is expected to pick the first index on this row"
    np.savez_compressed(os.path.join(OUT, "kmeans_near_tie.npz"), X=X, K=K, init_idx=init_idx.numpy().astype(np.int32),
                        reseed_idx=np.asarray(reseed_stream, np.int32), labels=labels.numpy(), centroids=red.numpy(), wsum=wsum.numpy(),
                        exit_iter=it, trace=trace, seed=seed, tie_row=tie_row, sq_dist_first=sA, sq_dist_second=sB)
    return dict(tie_row=tie_row, ref_label=int(labels[tie_row]), exit_iter=int(it))


def main():
    os.makedirs(OUT, exist_ok=True)
    ns = MG.load_reference_namespace()
    print("near tie:", gen_near_tie(ns))
    if "--no-composed" not in sys.argv:
        import make_golden_r05_composed as C
        print("composed:", C.gen_composed(ns))


if __name__ == "__main__":
    main()
