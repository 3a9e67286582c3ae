"""GPU: COMPOSED parity of the C2 -> C3 chain at the SHIPPED memory geometry (reference inference_streamchat_v0.3.sh:13-18:
chunk 40, K 5, interval 10, short window 20 / remember 5, tau 5).  440 seeded cross-fade frames = eleven chunks of 40: eleven depth-0 nodes and
ONE real merge, the whole-frame k-means over T = 400 frames at D = 576 x 3584 = 2 064 384 (reference inference_streaming_longva_v2.py:319-358,
utiles.py:567-620) - on HIP-ENCODED fp16 features against fp32-ENCODED ones (tests/test_gpu_kmeans.py pins that k-means bit-exactly, but on planted
clusters).  Both sides and what is swapped between them: tests/_composed.py.  Asserted: identical short-memory frames, all 400 merge labels and the
exit iteration, the tree, the retrieved chunks = retrieved frame indices, then the last-position logits and the first token of the 49 k-token prefill
on the retrieved context (2 Qwen2 layers at the 7B widths).  The label margins of EVERY Lloyd iteration and the similarity gaps are printed.
The fp32 host encode runs as worker processes over all host cores (oracle/torch_ref.encode_frames_u8_parallel: same batches, same arithmetic)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

N_FRAMES = 440
PERIOD = 80                     # 5.5 cross-fades over the stream, as in the 88-frame test (period 16): K = 5 clusters cut through continuous fades
MEM = dict(chunk_size=40, num_clusters=5, interval=10, short_window=20, remember_window=5, tau=5)      # inference_streamchat_v0.3.sh:13-18

from tests._composed import build, prefill_both   # noqa: E402


@pytest.fixture(scope="module")
def c2():
    return build(N_FRAMES, MEM, period=PERIOD, micro_batch=440, cpu_workers=16, cpu_batch=4)


def test_shipped_short_memory_frames_identical(c2):
    assert c2["hip"]["short"] == c2["cpu"]["short"] and len(c2["hip"]["short"]) == 5
    assert all(N_FRAMES - 20 <= f < N_FRAMES for f in c2["hip"]["short"])


def test_shipped_merge_T400_cluster_assignments_identical(c2):
    import oracle
    h, c = c2["hip"], c2["cpu"]
    m = c["margin"]
    moved = [int((c["trace"][i] != c["trace"][i - 1]).sum()) for i in range(1, len(c["trace"]))]
    print(f"\n[shipped] merge k-means T={c['T']} K=5 D=2064384: oracle exit iteration {c['exit_iter']}, HIP {h['exit_iter']}; labels moved per Lloyd iteration "
          f"{moved}; min relative label margin at exit {m.min():.3e} (row {int(m.argmin())}); cluster sizes {np.bincount(c['labels'], minlength=5).tolist()}")
    assert h["T"] == c["T"] == 400
    assert np.array_equal(h["labels"], c["labels"]), "merge cluster assignments differ between the HIP path and the CPU reference path"
    assert h["exit_iter"] == c["exit_iter"]
    assert c["exit_iter"] >= 2 and m.min() < 0.5          # boundaries are decided by real distance comparisons, not by scene cuts
    # the HIP k-means on ITS OWN (fp16) features against the oracle on the same fp16 bits: bit-exact labels (SC-KM2), as everywhere else
    X16 = c2["feats"][:400].reshape(400, -1).cpu().numpy()
    import torch
    torch.manual_seed(0)
    init = torch.randperm(400)[:5].numpy().astype(np.int32)
    o = oracle.kmeans_fit(X16, 5, init, np.zeros(50, np.int32), max_iter=10)
    assert np.array_equal(o["labels"], h["labels"]) and o["iters"] == h["exit_iter"]


def test_shipped_tree_and_retrieved_frames_identical(c2):
    h, c = c2["hip"], c2["cpu"]
    print(f"\n[shipped] tree: {[(n['depth'], n['rows'], len(n['children'])) for n in c['tree']]}; retrieved chunks (first frame) "
          f"{[r[0] for r in c['retrieved']]}; top-1 minus top-2 cosine per search level (fp32 side): {['%.3e' % g for g in c2['gaps']]}")
    assert h["tree"] == c["tree"]
    assert [n["depth"] for n in c["tree"]] == [1, 0] and c["tree"][0]["rows"] == 5 and len(c["tree"][0]["children"]) == 10
    assert h["texts"] == c["texts"]
    assert h["retrieved"] == c["retrieved"]                       # identical retrieved-frame indices
    assert len(c["retrieved"]) == 2 and c["retrieved"][1][0] == 400 and all(len(r) == 40 for r in c["retrieved"])


def test_shipped_merged_centroids_close(c2):
    from tests._tol import assert_close_fp16
    hip_feats, ref = c2["feats"], c2["ref"]
    labels = c2["cpu"]["labels"]
    for k in range(5):
        rows = np.nonzero(labels == k)[0]
        assert_close_fp16(hip_feats[rows].float().mean(0).cpu(), ref[rows].mean(0), max_rel=6e-3, what=f"centroid {k} ({len(rows)} frames)")


def test_shipped_c3_prefill_49k_logits_and_first_token(c2):
    """[short 5 | best child 40 | redundant depth-0 node 40] frames = 48 960 visual tokens + the prompt: the C3 context, prefilled through 2 Qwen2
    layers at the 7B widths on both sides (fp32 side: row-chunked causal attention, oracle/torch_ref.qwen2_logits(row_chunk=...))."""
    from tests._tol import assert_close_fp16
    logits, ref, n_ctx = prefill_both(c2, layers=2, row_chunk=2048)
    top = ref.topk(2).values
    print(f"\n[shipped C3] prefill of {n_ctx} tokens (2 Qwen2-7B-width layers): first token HIP {int(logits.argmax())} / CPU {int(ref.argmax())}; "
          f"best-minus-second logit {float(top[0] - top[1]):.3e} of max |logit| {float(ref.abs().max()):.3f}")
    assert n_ctx > 48960
    assert_close_fp16(logits, ref, max_rel=6e-3, what="C3 last-position logits (49 k-token context)")
    assert int(logits.argmax()) == int(ref.argmax())
