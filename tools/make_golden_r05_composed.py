#!/usr/bin/env python3
"""composed_ref_trace.{npz,json}: ONE composed run whose policy side is the REFERENCE'S OWN code (authoring container only).

  uint8 frames --HF CLIPImageProcessor--> HF CLIPVisionModel (tiny, fp32, hidden_states[-2][:, 1:]) --> mlp2x_gelu (torch.nn) = the arithmetic of
  `encode_images` (longva/model/llava_arch.py:179-184)
  --> the reference's updating_memory_buffer (inference_streaming_longva_v2.py:267-378, AST-extracted and executed: its own forgetting sampler,
      chunking, weighted_kmeans_feature, fast_building_memory_tree_summarize_token with the reference's conversation templates and
      tokenizer_image_token), two segments, ONE real merge (T = 40 frames, K = 3)
  --> the reference's fast_search_tree_multi_modal_with_embedding (utiles.py:685-788) with a tiny HF BertModel as the embedding model.

Stand-ins for what is absent offline (they are INPUTS, identical on both sides of tests/test_gpu_composed_ref.py): the position captioner and
the synthetic / hash tokenizers of tests/_composed.py, streamchat_amd/synthetic.py and streamchat_amd/text.py.  Only inputs (frames are
re-generated from their seed, model weights) and expected outputs (features, short-memory indices, k-means init / labels, trees, retrieved
path) are written."""
import json
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_golden as MG            # noqa: E402
import make_golden_r02 as MR2       # noqa: E402

OUT = MG.OUT
MEM = dict(short_window=20, remember_window=5, tau=5, chunk_size=8, num_clusters=3, interval=5)
SEGMENTS = (44, 20)                 # frames per segment: 6 chunks -> merge of the first 5 (T = 40); then 3 more chunks, no merge
SEED, PERIOD, SIDE = 4321, 8, 56
# a random-init BERT separates texts only by the words they share: the question repeats words of chunk caption 3 (a child of the merged node) and
# of chunk caption 7 (a depth-0 node of the second segment), so that both arg-max decisions of the search carry gaps an fp16 encoder cannot flip
def _question():
    from streamchat_amd import synthetic
    return " ".join(synthetic.caption(3).split()[2:24] + synthetic.caption(7).split()[2:24])


QUESTION = _question()


def tiny_bert(seed=11):
    from transformers import BertConfig, BertModel
    torch.manual_seed(seed)
    cfg = BertConfig(vocab_size=2048, hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256,
                     max_position_embeddings=128, hidden_act="gelu", layer_norm_eps=1e-12)
    m = BertModel(cfg, add_pooling_layer=False).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn_like(p) * (0.3 if "embeddings" in n else 0.6 / p.shape[-1] ** 0.5))
            else:
                p.copy_(torch.randn_like(p) * 0.1 + (1.0 if n.endswith("LayerNorm.weight") else 0.0))
        # a random-init BERT's [CLS] state is dominated by the constant [CLS] / position-0 / type-0 embeddings and the residual stream (all cosines
        # 0.993-0.998, decision gaps 1e-4): make the attention branch dominate the residual (value and output projections x 4), so that the pooled
        # vector is a function of the attended words and the arg-max decisions of the search carry gaps an fp16 encoder cannot flip
        for L in m.encoder.layer:
            L.attention.self.value.weight *= 4.0
            L.attention.output.dense.weight *= 4.0
    return m


def tiny_clip():
    from transformers import CLIPVisionConfig, CLIPVisionModel
    d = np.load(os.path.join(OUT, "clip_tiny.npz"))
    cfg = CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=56,
                           patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5, projection_dim=64)
    m = CLIPVisionModel(cfg).eval()
    sd = {k[4:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("vit.")}
    own = m.state_dict()
    m.load_state_dict({k: sd[k if k in sd else "vision_model." + k] if (k in sd or "vision_model." + k in sd) else sd[k.replace("vision_model.", "", 1)]
                       for k in own})
    proj = torch.nn.Sequential(torch.nn.Linear(128, 256), torch.nn.GELU(), torch.nn.Linear(256, 256)).eval()
    proj.load_state_dict({k[5:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("proj.")})
    return m, proj


def encode(frames_u8, clip, proj):
    """process_images (utiles.py:71-87: CLIPImageProcessor.preprocess) -> encode_images (llava_arch.py:179-184)"""
    from PIL import Image
    from transformers import CLIPImageProcessor
    proc = CLIPImageProcessor(do_resize=True, size={"shortest_edge": SIDE}, do_center_crop=True, crop_size={"height": SIDE, "width": SIDE},
                              do_rescale=True, do_normalize=True, image_mean=[0.48145466, 0.4578275, 0.40821073],
                              image_std=[0.26862954, 0.26130258, 0.27577711], do_convert_rgb=True)
    px = proc.preprocess([Image.fromarray(f) for f in frames_u8], return_tensors="pt")["pixel_values"]
    with torch.no_grad():
        h = clip(px, output_hidden_states=True).hidden_states[-2][:, 1:]
        return proj(h)


def gen_composed(ns):
    from tests import _composed as TC
    from streamchat_amd import synthetic, text as T
    MR2._stub_longva()
    from longva.conversation import conv_templates
    from longva.mm_utils import tokenizer_image_token
    from longva.constants import IMAGE_TOKEN_INDEX, DEFAULT_IMAGE_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN
    torch.set_num_threads(4)
    torch.Tensor.cuda = lambda s, *a, **k: s                      # the reference hard-codes .cuda(); CPU container
    clip, proj = tiny_clip()
    bert = tiny_bert()
    n_total = sum(SEGMENTS)
    u8 = TC.crossfade_stream(n_total, seed=SEED, period=PERIOD, h=SIDE, w=SIDE)
    feats = encode(u8, clip, proj)                                # [N, 16, 256] fp32
    bank_all = [feats[i:i + 1] for i in range(n_total)]
    ident = {id(t): i for i, t in enumerate(bank_all)}

    # the reference's own functions: utiles.py ones from MG's namespace (with the REAL conversation templates), the entry script's updater
    ns = dict(ns)
    ns["conv_templates"] = conv_templates
    for k, v in list(ns.items()):                                 # re-bind the extracted functions' globals to this namespace copy
        if isinstance(v, types.FunctionType) and v.__code__.co_filename == "utiles.py":
            ns[k] = types.FunctionType(v.__code__, ns, v.__name__, v.__defaults__, v.__closure__)
    km_calls = []
    ref_km = ns["weighted_kmeans_feature"]

    def km_recording(img_feature, video_max_frames, weights=None):
        T_, K = img_feature.shape[0], video_max_frames
        st_t, st_r = torch.get_rng_state(), random.getstate()
        init = torch.randperm(T_)[:K].clone()                    # what utiles.py:295 is about to draw
        reseed = [random.randint(0, T_ - 1) for _ in range(10 * K)]
        torch.set_rng_state(st_t); random.setstate(st_r)
        out = ref_km(img_feature, video_max_frames, weights)
        if len(out) == 2:
            C, lab2, wsum, it, trace = MG.kmeans_trace(img_feature.reshape(T_, -1), K, init, reseed)
            assert torch.equal(out[1], lab2) and torch.equal(out[0].reshape(K, -1), C)
            X = img_feature.reshape(T_, -1).double()
            d2 = torch.cdist(X, out[0].reshape(K, -1).double()).pow(2).sort(dim=1).values
            km_calls.append(dict(T=T_, K=K, init_idx=init.numpy().astype(np.int32), reseed_idx=np.asarray(reseed, np.int32), labels=out[1].numpy(),
                                 trace=trace, exit_iter=int(it), centroids=out[0].numpy(), margin=((d2[:, 1] - d2[:, 0]) / d2[:, 1]).numpy()))
        return out
    ns["weighted_kmeans_feature"] = km_recording
    ens = dict(ns)
    ens.update(torch=torch, tokenizer_image_token=tokenizer_image_token, IMAGE_TOKEN_INDEX=IMAGE_TOKEN_INDEX, DEFAULT_IMAGE_TOKEN=DEFAULT_IMAGE_TOKEN,
               DEFAULT_IM_START_TOKEN=DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN=DEFAULT_IM_END_TOKEN, BLUE="", RESET="")
    MR2._entry_functions({"updating_memory_buffer"}, ens)
    update = ens["updating_memory_buffer"]

    cap, stok = TC.PositionCaptioner("cpu"), synthetic.SyntheticTokenizer()
    tree, updates, f0 = None, [], 0
    import contextlib, io
    for seg, n in enumerate(SEGMENTS):
        bank = bank_all[f0:f0 + n]
        np.random.seed(seg); torch.manual_seed(seg); random.seed(seg)
        n_km = len(km_calls)
        with contextlib.redirect_stdout(io.StringIO()):
            tree, short = update(bank, tree, cap, stok, True, **MEM)
        updates.append(dict(frames=[f0, f0 + n], seed=seg, short=[ident[id(t)] for t in short], tree=TC.describe(tree),
                            kmeans_calls=list(range(n_km, len(km_calls)))))
        f0 += n

    etok = T.HashTokenizer(vocab=2048, max_len=64)
    sims = []
    real_cos = ns["cos_sim"]

    def cos_rec(a, b):
        s = real_cos(a, b)
        sims.append(float(s))
        return s
    ns["cos_sim"] = cos_rec
    with contextlib.redirect_stdout(io.StringIO()):
        with torch.no_grad():
            path, texts = ns["fast_search_tree_multi_modal_with_embedding"](tree, QUESTION, None, bert, etok)

    def locate(t):
        if id(t) in ident:
            return dict(kind="frame", frame=ident[id(t)])
        # a depth-0 node's centroids is torch.cat of its chunk's frames: find the frame run
        for i in range(n_total - t.shape[0] + 1):
            if torch.equal(t, feats[i:i + t.shape[0]]):
                return dict(kind="frames", first=i, count=int(t.shape[0]))
        return dict(kind="centroids", rows=int(t.shape[0]))
    retrieved = [locate(t) for t in path]
    merged = [t.numpy() for t in path if locate(t)["kind"] == "centroids"]

    meta = dict(mem=MEM, segments=list(SEGMENTS), seed=SEED, period=PERIOD, side=SIDE, question=QUESTION, updates=updates, texts=list(texts),
                retrieved=retrieved, sims=sims,
                kmeans=[dict(T=c["T"], K=c["K"], exit_iter=c["exit_iter"], min_margin=float(c["margin"].min())) for c in km_calls])
    json.dump(meta, open(os.path.join(OUT, "composed_ref_trace.json"), "w"), indent=0)
    arrays = dict(features=feats.numpy().astype(np.float32))
    for i, c in enumerate(km_calls):
        for k in ("init_idx", "reseed_idx", "labels", "trace", "centroids", "margin"):
            arrays[f"km{i}_{k}"] = c[k]
    for i, m in enumerate(merged):
        arrays[f"retrieved_centroids_{i}"] = m
    for k, v in bert.state_dict().items():
        arrays["bert." + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "composed_ref_trace.npz"), **arrays)
    s = sorted(sims, reverse=True)
    return dict(n_frames=n_total, updates=[(u["frames"], u["short"], [(n["depth"], n["rows"]) for n in u["tree"]]) for u in updates],
                kmeans=meta["kmeans"], retrieved=retrieved, n_sims=len(sims), top_sims=s[:3])


if __name__ == "__main__":
    print(gen_composed(MG.load_reference_namespace()))
