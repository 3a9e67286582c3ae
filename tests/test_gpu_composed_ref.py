"""GPU: ONE composed run against a trace of the REFERENCE'S OWN policy code (the other composed tests run the package's host
functions on both sides; here the expected side was produced by executing the reference's updating_memory_buffer
(inference_streaming_longva_v2.py:267-378), weighted_kmeans_feature / fast_building_memory_tree_summarize_token (utiles.py:291-330,489-620) and
fast_search_tree_multi_modal_with_embedding (utiles.py:685-788) on HF tiny-CLIP fp32 features with a tiny HF BertModel -
tools/make_golden_r05_composed.py, fixture tests/golden/composed_ref_trace.{json,npz}).

HIP side: the same seeded uint8 frames -> fused preprocess + ViT + projector (fp16) -> streaming.updating_memory_buffer per segment (its merge
= sc_kmeans_fit on T = 40 frames) -> utiles.fast_search_tree_multi_modal_with_embedding over text.BertEncoder + sc_sim_topk.  Must land on the
reference's short-memory frames, k-means initial rows, labels at every Lloyd iteration, exit iteration, trees (depths, rows, caption and summary
texts - the summary text is a hash of the reference's rendered summary prompt) and retrieved frames."""
import json
import os
import random

import numpy as np
import pytest
import torch

from streamchat_amd import ops, streaming as S, synthetic, text as T, utiles as U, vision as V
from tests import _composed as TC
from tests._tol import assert_close_fp16

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def test_composed_run_lands_on_the_reference_trace():
    meta = json.load(open(os.path.join(G, "composed_ref_trace.json")))
    arr = np.load(os.path.join(G, "composed_ref_trace.npz"))
    mem, dev = meta["mem"], torch.device("cuda:0")
    d = np.load(os.path.join(G, "clip_tiny.npz"))
    sd = {k[4:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("vit.")}
    sp = {k[5:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("proj.")}
    cfg = V.CLIPVisionConfigLite(hidden=128, layers=3, heads=2, intermediate=256, patch=14, image_size=meta["side"])
    enc = V.FrameEncoder(V.CLIPVisionTower(sd, cfg, device=dev), V.MMProjector(sp, device=dev), micro_batch=24)
    n_total = sum(meta["segments"])
    u8 = TC.crossfade_stream(n_total, seed=meta["seed"], period=meta["period"], h=meta["side"], w=meta["side"])
    feats = enc.encode_frames_u8(torch.from_numpy(u8).to(dev))                           # [64, 16, 256] fp16
    assert_close_fp16(feats, torch.from_numpy(arr["features"]), what="HIP encode vs HF CLIPImageProcessor + CLIPVisionModel + mlp2x_gelu (fp32)")

    # ---- memory updates, one per segment, RNG seeded as the generator seeded the reference's run ----
    row = feats[0].numel()
    cap, stok = TC.PositionCaptioner(dev), synthetic.SyntheticTokenizer()
    km = []
    real_km = U.weighted_kmeans_feature

    def km_rec(x, k, *a, **kw):
        init = torch.randperm(x.shape[0])[:k]                       # the draw weighted_kmeans_feature would make itself (same CPU generator)
        red, labels, info = real_km(x, k, *a, init_idx=init, return_info=True, **kw)
        km.append(dict(T=int(x.shape[0]), init=init.numpy(), labels=labels.cpu().numpy(), exit_iter=int(info["info"][0]), C=info["centroids_f32"]))
        return red, labels
    U.weighted_kmeans_feature = km_rec
    tree, f0 = None, 0
    try:
        for u in meta["updates"]:
            a, b = u["frames"]
            assert a == f0
            bank = [feats[i:i + 1] for i in range(a, b)]
            torch.manual_seed(u["seed"]); random.seed(u["seed"])
            n_km = len(km)
            tree, short = S.updating_memory_buffer(bank, tree, cap, stok, True, rng=np.random.RandomState(u["seed"]), **mem)
            assert [int(TC.frame_index(t, feats, row)) for t in short] == u["short"]
            assert TC.describe(tree) == u["tree"]
            assert len(km) - n_km == len(u["kmeans_calls"])
            f0 = b
    finally:
        U.weighted_kmeans_feature = real_km
    assert len(km) == len(meta["kmeans"]) == 1
    for i, (got, want) in enumerate(zip(km, meta["kmeans"])):
        assert got["T"] == want["T"] and np.array_equal(got["init"], arr[f"km{i}_init_idx"])
        assert np.array_equal(got["labels"], arr[f"km{i}_labels"]) and got["exit_iter"] == want["exit_iter"]
        np.testing.assert_allclose(got["C"].reshape(want["K"], -1).cpu().numpy(), arr[f"km{i}_centroids"].reshape(want["K"], -1), rtol=0, atol=4e-3 * np.abs(arr[f"km{i}_centroids"]).max())
        # ... and at every Lloyd iteration (the oracle on the HIP features, which the HIP kernel equals bit for bit - test_gpu_kmeans)
        import oracle
        o = oracle.kmeans_fit(feats[:got["T"]].reshape(got["T"], -1).cpu().numpy(), want["K"], arr[f"km{i}_init_idx"], arr[f"km{i}_reseed_idx"], trace=True)
        assert np.array_equal(o["trace"], arr[f"km{i}_trace"]) and np.array_equal(o["labels"], got["labels"])
        print(f"\n[composed-ref] merge k-means T = {got['T']}: labels identical through {got['exit_iter'] + 1} Lloyd iterations, reference margin {want['min_margin']:.3f}")

    # ---- retrieval: the reference's tree search vs ours (HIP BERT, sc_sim_topk) ----
    sdb = {k[5:]: torch.from_numpy(arr[k]) for k in arr.files if k.startswith("bert.")}
    bert = T.BertEncoder(sdb, T.BertConfigLite(hidden=128, layers=2, heads=4, intermediate=256, vocab=2048, max_pos=128), device=dev)
    best = []
    real_best = U._best_positive

    def best_rec(q, embs):
        i, s = real_best(q, embs)
        best.append(s)
        return i, s
    U._best_positive = best_rec
    try:
        path, texts = U.fast_search_tree_multi_modal_with_embedding(tree, meta["question"], feats, bert, T.HashTokenizer(vocab=2048, max_len=64), cache=U.CaptionEmbeddingCache())
    finally:
        U._best_positive = real_best
    assert list(texts) == meta["texts"]
    got = [dict(kind="frames", first=int(TC.frame_index(t, feats, row)), count=int(t.shape[0])) for t in path]
    assert got == meta["retrieved"]
    sims = meta["sims"]
    groups = [sims[:mem["interval"]], sims[mem["interval"]:]]               # children of the merged node, then the depth-0 top-level nodes
    for s, g in zip(best, groups):
        top = sorted(g, reverse=True)
        assert abs(s - top[0]) < 5e-4, (s, top)
        print(f"[composed-ref] best cosine {s:.4f} (reference {top[0]:.4f}, runner-up {top[1]:.4f})")
