"""GPU: COMPOSED parity at config C1's size (BASELINE.json configs[0], north_star "match the reference CPU /
PyTorch path on the same frame inputs"): the SAME 64 seeded uint8 frames go through

  HIP   fused preprocess -> ViT-L/14-336 (23 layers) -> mlp2x_gelu (fp16 storage, fp32 accumulate) -> weighted_kmeans_feature(K=8)
  CPU   oracle/torch_ref: HF-arithmetic preprocess + ViT-L + projector in fp32 -> oracle.kmeans_fit (C, SC-KM2 order) on fp32 features

and the cluster assignments must be identical; then a fixed caption table goes through the HIP BERT-large CLS encoder + cosine top-k
and through torch_ref's fp32 BERT + the oracle top-k, and the retrieved indices / the tree-search path must be identical.  The
minimum relative label margin (second-best vs best squared distance, on the CPU side) is printed so that a near-tie — where fp16
feature rounding could legitimately flip a label — is visible instead of silently passing."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

N_FRAMES, K = 64, 8


@pytest.fixture(scope="module")
def c1():
    import oracle
    from oracle import torch_ref as R
    from streamchat_amd import synthetic, utiles as U, vision as V
    dev = torch.device("cuda:0")
    cfg = V.CLIPVisionConfigLite(**V.VIT_L_336)
    sd_vit = V.random_clip_state_dict(cfg, seed=0, device=dev)
    sd_proj = V.random_projector_state_dict(1024, 3584, seed=1, device=dev)
    enc = V.FrameEncoder(V.CLIPVisionTower(sd_vit, cfg, device=dev), V.MMProjector(sd_proj, device=dev), micro_batch=64)
    u8 = synthetic.frame_stream(N_FRAMES, seed=1234, scene_len=8)                 # 8 scenes of 8 frames + per-frame noise
    feats = enc.encode_frames_u8(torch.from_numpy(u8).to(dev))                     # [64, 576, 3584] fp16
    torch.manual_seed(0)
    init_idx = torch.randperm(N_FRAMES)[:K]                                        # CPU randperm(T)[:K]
    reseed = [0] * (10 * K)
    red, labels, info = U.weighted_kmeans_feature(feats, K, init_idx=init_idx, reseed_idx=reseed, return_info=True)
    # ---- CPU path ----
    # fp32 host encode: one single-threaded worker process per core the container may burn (oracle/torch_ref.parallel_plan: the pool's hosts run
    # under a 16-core cgroup quota; same arithmetic per frame as the in-process loop it replaces)
    torch.set_num_threads(min(32, R.host_cpu_budget()[1]))
    workers, threads = R.parallel_plan(N_FRAMES, batch=4)
    ref = R.encode_frames_u8_parallel(sd_vit, sd_proj, u8, workers=workers, threads=threads, batch=4)
    Xr = ref.reshape(N_FRAMES, -1).numpy()
    o = oracle.kmeans_fit(Xr, K, init_idx.numpy().astype(np.int32), np.asarray(reseed, np.int32))
    d2 = oracle.kmeans_dist2(Xr, o["centroids"])
    return dict(feats=feats, ref=ref, red=red, labels=labels.cpu().numpy(), info=info, o=o, d2=d2, dev=dev)


def test_c1_encoder_features_match_fp32_reference(c1):
    a, b = c1["feats"].float().cpu(), c1["ref"]
    err = (a - b).abs()
    scale = b.abs().max().item()
    cos = torch.nn.functional.cosine_similarity(a.reshape(N_FRAMES, -1), b.reshape(N_FRAMES, -1), dim=1)
    print(f"\n[C1] encoder: max|err| = {err.max().item():.3e} ({err.max().item() / scale:.2e} of max|ref| {scale:.3f}), "
          f"rms err / rms ref = {(err.pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item():.2e}, min cosine = {cos.min().item():.6f}")
    assert err.max().item() < 6e-3 * scale                      # fp16 storage through 23 layers; observed ~3e-3
    assert (err.pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item() < 2e-3
    assert cos.min().item() > 0.9999


def test_c1_cluster_assignments_identical(c1):
    d2 = np.sort(c1["d2"], axis=1)
    margin = (d2[:, 1] - d2[:, 0]) / d2[:, 1]                   # in [0, 1]; 1 = the row IS its centroid (singleton cluster)
    print(f"\n[C1] k-means(K={K}) on {N_FRAMES} frames: oracle exit iteration {c1['o']['iters']}, HIP {int(c1['info']['info'][0])}; "
          f"min relative label margin = {margin.min():.3e} (row {int(margin.argmin())}); cluster sizes {np.bincount(c1['o']['labels'], minlength=K).tolist()}")
    assert np.array_equal(c1["labels"], c1["o"]["labels"]), "cluster assignments differ between the HIP path and the CPU reference path"
    assert int(c1["info"]["info"][0]) == c1["o"]["iters"]
    # the fp32 centroids of the HIP fit (from fp16 features) vs the oracle's (from fp32 features): same clusters, feature-rounding apart
    C = c1["info"]["centroids_f32"].reshape(K, -1).cpu().numpy()
    rel = np.abs(C - c1["o"]["centroids"]).max() / np.abs(c1["o"]["centroids"]).max()
    assert rel < 6e-3, rel


def test_c1_retrieval_indices_identical(c1):
    """fixed caption table -> HIP BERT-large CLS + sc_sim_topk  vs  torch_ref fp32 BERT + oracle top-k; then the tree-search path."""
    import oracle
    from oracle import torch_ref as R
    from streamchat_amd import ops, synthetic, text as T, utiles as U
    dev = c1["dev"]
    cfg = T.BertConfigLite(**T.BERT_LARGE)
    sd = T.random_bert_state_dict(cfg, seed=2, device=dev)
    bert, tok = T.BertEncoder(sd, cfg, device=dev), T.HashTokenizer()
    captions = [synthetic.caption(i) for i in range(26)]
    question = "where did I leave the red cup and what was on the kitchen table"
    ids = tok(captions + [question])
    hip = ops.pool(bert.forward(ids["input_ids"], ids["attention_mask"].sum(1)), ids["attention_mask"].sum(1), "cls")
    sdc = {k: v.float().cpu() for k, v in sd.items()}
    with torch.no_grad():
        ref = R.bert_last_hidden(sdc, ids["input_ids"], ids["attention_mask"], heads=16, layers=24)[:, 0]
    cos = torch.nn.functional.cosine_similarity(hip.cpu(), ref, dim=1)
    assert cos.min().item() > 0.9999
    for k in (1, 8):
        gi, gs = ops.sim_topk(hip[-1], hip[:-1], k, "cos")
        oi, osc = oracle.topk(ref[-1].numpy(), ref[:-1].numpy(), k, "cos")
        assert gi.cpu().tolist() == oi.tolist(), (k, gi.cpu().tolist(), oi.tolist())
    sims = torch.nn.functional.cosine_similarity(ref[-1:], ref[:-1]).sort(descending=True).values
    print(f"\n[C1] retrieval: top-1 cosine {sims[0]:.5f}, gap to second {float(sims[0] - sims[1]):.3e}; min cosine(HIP, fp32 ref) over 27 texts {cos.min().item():.6f}")
    # tree search over the K cluster centroids as one merged node + redundant leaves: identical path indices
    N = U.MultimodalTreeNode
    leaves = [N(c1["feats"][i * 8:(i + 1) * 8], captions[i], depth=0) for i in range(8)]
    root = N(c1["red"], captions[20], depth=1)
    root.children = leaves[:5]
    nodes = [root] + leaves[5:]
    path, texts = U.fast_search_tree_multi_modal_with_embedding(nodes, question, c1["feats"], bert, tok, cache=U.CaptionEmbeddingCache())
    def best(idx):                                           # reference rule on the fp32 side: first strictly-positive maximum, else 0
        s = torch.nn.functional.cosine_similarity(ref[-1:], ref[idx])
        j = int(torch.argmax(s))
        return idx[j] if float(s[j]) > 0 else idx[0]
    assert texts == [captions[best(list(range(5)))], captions[best([5, 6, 7])]]
