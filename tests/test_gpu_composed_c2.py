"""GPU: COMPOSED parity of the C2 chain at a reduced but MERGING geometry (BASELINE.json configs[1], north_star "match
the reference CPU path on the same frame inputs (identical retrieved-frame indices)").  The SAME 88 seeded uint8 frames and the same
host RNG draws go through

  HIP   fused preprocess -> ViT-L/14-336 (23 layers) -> mlp2x_gelu (fp16) -> streaming.updating_memory_buffer (chunk 8, K 5, interval 10:
        11 depth-0 nodes, ONE merge of the first ten = whole-frame k-means over T = 80 frames) -> BERT-large-CLS tree search (HIP encoder,
        sc_sim_topk)
  CPU   oracle/torch_ref fp32 encode -> THE SAME host policy functions (updating_memory_buffer / fast_building_memory_tree_summarize_token
        / fast_search_tree_multi_modal_with_embedding: reference inference_streaming_longva_v2.py:319-358, utiles.py:567-620,715-748)
        with the k-means, top-k and text-encoder providers swapped for oracle.kmeans_fit (C, fp32 features), oracle.topk and torch_ref's
        fp32 BERT

and must agree on: the short-memory frame indices, the merge's cluster assignment of all 80 frames and its exit iteration, the tree
shape and texts, the retrieved nodes (which chunks), hence the retrieved frame indices.  The minimum relative label margin and the
top-1 / top-2 similarity gaps are printed, so a near-tie (where fp16 feature rounding could legitimately flip a decision) is visible.
The chunk captioner is a stand-in that names chunks by POSITION (the LLM captioner is outside C2 and would see different bits on the
two sides); the merge summary is derived from the captions it summarises, as upstream."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

N_FRAMES = 88
MEM = dict(chunk_size=8, num_clusters=5, interval=10, short_window=20, remember_window=5, tau=5)


from tests._composed import QUESTION, build, crossfade_stream, prefill_both   # noqa: E402,F401  (shared with the shipped-geometry test)


@pytest.fixture(scope="module")
def c2():
    return build(N_FRAMES, MEM, period=16, micro_batch=88, cpu_workers=16, cpu_batch=4)                         # 5.5 cross-fades of 16 frames: no trivially separable scene cuts


def test_c2_short_memory_frames_identical(c2):
    assert c2["hip"]["short"] == c2["cpu"]["short"] and len(c2["hip"]["short"]) == 5
    assert all(N_FRAMES - 20 <= f < N_FRAMES for f in c2["hip"]["short"])


def test_c2_merge_cluster_assignments_identical(c2):
    h, c = c2["hip"], c2["cpu"]
    m = c["margin"]
    print(f"\n[C2] merge k-means T={c['T']} K={MEM['num_clusters']}: oracle exit iteration {c['exit_iter']}, HIP {h['exit_iter']}; min relative label margin "
          f"{m.min():.3e} (row {int(m.argmin())}); cluster sizes {np.bincount(c['labels'], minlength=5).tolist()}")
    assert h["T"] == c["T"] == 80
    assert np.array_equal(h["labels"], c["labels"]), "merge cluster assignments differ between the HIP path and the CPU reference path"
    assert h["exit_iter"] == c["exit_iter"]
    assert c["exit_iter"] >= 3 and m.min() < 0.5          # the stream is NOT trivially separable: boundaries moved over several Lloyd iterations


def test_c2_tree_and_retrieved_frames_identical(c2):
    h, c = c2["hip"], c2["cpu"]
    print(f"\n[C2] tree: {[(n['depth'], n['rows'], len(n['children'])) for n in c['tree']]}; retrieved chunks (first frame) "
          f"{[r[0] for r in c['retrieved']]}; top-1 minus top-2 cosine per search level (fp32 side): {['%.3e' % g for g in c2['gaps']]}")
    assert h["tree"] == c["tree"]                                 # depths, row counts, texts, children - the whole shape
    assert [n["depth"] for n in c["tree"]] == [1, 0] and c["tree"][0]["rows"] == 5 and len(c["tree"][0]["children"]) == 10
    assert h["texts"] == c["texts"]
    assert h["retrieved"] == c["retrieved"]                       # identical retrieved-frame indices
    assert len(c["retrieved"]) == 2 and c["retrieved"][1][0] == 80      # the best child of the merged node + the one redundant depth-0 node


def test_c2_merged_centroids_close(c2):
    """the merged node's 5 centroid frames: HIP (fp16 features, fp32 means) vs CPU (fp32 features): same clusters, feature rounding apart"""
    from tests._tol import assert_close_fp16
    hip_feats, ref = c2["feats"].float().cpu(), c2["ref"]
    labels = c2["cpu"]["labels"]
    for k in range(5):
        rows = np.nonzero(labels == k)[0]
        assert_close_fp16(hip_feats[rows].mean(0), ref[rows].mean(0), max_rel=6e-3, what=f"centroid {k} ({len(rows)} frames)")


def test_c3_prefill_logits_and_first_token_on_the_retrieved_context(c2):
    """The LAST stage of the chain (BASELINE.json configs[2]) on what the stages above produced: the [short | retrieved] frame tokens of
    the SAME stream (5 + 8 + 8 frames = 12 096 visual tokens) spliced into the reference's answer prompt and prefilled through a Qwen2
    stack at the 7B widths (2 layers, random-init: 28 / 4 heads x 128, rotary epilogue GEMMs, pre-scaled causal GQA attention, SwiGLU) -
    HIP on the fp16 HIP features vs oracle/torch_ref fp32 on the CPU path's fp32 features: last-position logits within the fp16
    tolerance and the same first token (margin between the two best logits printed)."""
    from tests._tol import assert_close_fp16
    logits, ref, n_ctx = prefill_both(c2, layers=2)
    top = ref.topk(2).values
    print(f"\n[C3] prefill of {n_ctx} tokens (2 Qwen2-7B-width layers): first token HIP {int(logits.argmax())} / CPU {int(ref.argmax())}; "
          f"best-minus-second logit {float(top[0] - top[1]):.3e} of max |logit| {float(ref.abs().max()):.3f}")
    assert_close_fp16(logits, ref, max_rel=6e-3, what="C3 last-position logits (12 k-token context)")
    assert int(logits.argmax()) == int(ref.argmax())
