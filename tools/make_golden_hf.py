#!/usr/bin/env python3
"""Golden vectors for the THIRD-PARTY arithmetic (transformers) the reference calls — authoring container only.

The installed transformers (5.15.0; the reference pins 4.37.2) provides CLIPVisionModel, BertModel and
Qwen2ForCausalLM on CPU.  Tiny random-init configs are run in fp32 and their weights + inputs + outputs are
saved, so oracle/torch_ref.py (and through it the HIP kernels) are pinned to the real HF modules at the call
sites reference clip_encoder.py:76 (output_hidden_states, select layer -2, drop CLS), multimodal_projector/
builder.py:41-48, utiles.py:707 (BertModel CLS) and llava_qwen.py:155."""
import os
import zlib
import sys

import numpy as np
import torch
import transformers

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def gen_clip():
    from transformers import CLIPVisionConfig, CLIPVisionModel
    torch.manual_seed(0)
    cfg = CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=56,
                           patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5, projection_dim=64)
    m = CLIPVisionModel(cfg).eval()
    with torch.no_grad():   # spread the (tiny-init) weights so the test is sensitive
        for n, p in m.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn_like(p) * (0.5 / p.shape[-1] ** 0.5) if "embedding" not in n else torch.randn_like(p) * 0.3)
            else:
                p.copy_(torch.randn_like(p) * 0.2 + (1.0 if "layer_norm" in n or "layrnorm" in n and n.endswith("weight") else 0.0))
    proj = torch.nn.Sequential(torch.nn.Linear(128, 256), torch.nn.GELU(), torch.nn.Linear(256, 256)).eval()
    x = torch.randn(3, 3, 56, 56)
    with torch.no_grad():
        out = m(x, output_hidden_states=True)
        feat = out.hidden_states[-2][:, 1:]
        y = proj(feat)
    d = {"cfg_heads": 2, "cfg_patch": 14, "cfg_layers": 3, "pixel_values": x.numpy(), "features": feat.numpy(), "projected": y.numpy()}
    for k, v in m.state_dict().items():   # store under the transformers-4.37.2 key names (vision_model.* prefix)
        d["vit." + (k if k.startswith("vision_model.") else "vision_model." + k)] = v.numpy()
    for k, v in proj.state_dict().items():
        d["proj." + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "clip_tiny.npz"), **d)


def gen_bert():
    from transformers import BertConfig, BertModel
    torch.manual_seed(1)
    cfg = BertConfig(vocab_size=500, hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256,
                     max_position_embeddings=64, hidden_act="gelu", layer_norm_eps=1e-12)
    m = BertModel(cfg, add_pooling_layer=False).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn_like(p) * (0.3 if "embeddings" in n else 0.6 / p.shape[-1] ** 0.5))
            else:
                p.copy_(torch.randn_like(p) * 0.1 + (1.0 if n.endswith("LayerNorm.weight") else 0.0))
    ids = torch.randint(5, 500, (3, 11))
    mask = torch.ones(3, 11, dtype=torch.long)
    mask[1, 7:] = 0
    mask[2, 3:] = 0
    ids = ids * mask
    with torch.no_grad():
        h = m(input_ids=ids, attention_mask=mask).last_hidden_state
    d = {"input_ids": ids.numpy(), "attention_mask": mask.numpy(), "last_hidden_state": h.numpy(), "cfg_heads": 4, "cfg_layers": 2}
    for k, v in m.state_dict().items():
        d["bert." + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "bert_tiny.npz"), **d)


def gen_qwen2():
    from transformers import Qwen2Config, Qwen2ForCausalLM
    torch.manual_seed(2)
    cfg = Qwen2Config(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=512, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=False)
    m = Qwen2ForCausalLM(cfg).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn_like(p) * (0.3 if "embed" in n else 0.8 / p.shape[-1] ** 0.5))
            else:
                p.copy_(torch.randn_like(p) * 0.1 + (1.0 if "norm" in n else 0.0))
    emb = torch.randn(1, 37, 256) * 0.5
    with torch.no_grad():
        logits = m(inputs_embeds=emb).logits[0]
        gen = m.generate(inputs_embeds=emb, max_new_tokens=8, do_sample=False, use_cache=True)
        gen_nc = m.generate(inputs_embeds=emb, max_new_tokens=8, do_sample=False, use_cache=False)
    assert torch.equal(gen, gen_nc)        # cache vs no-cache greedy outputs agree
    d = {"inputs_embeds": emb[0].numpy(), "logits": logits.numpy(), "greedy": gen[0].numpy(), "cfg": np.asarray([4, 2, 2, 64])}
    for k, v in m.state_dict().items():
        d["lm." + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "qwen2_tiny.npz"), **d)



def gen_preprocess():
    """G1: the reference's `process_images` (utiles.py:71-87) hands every frame to HF `CLIPImageProcessor.preprocess`; on 336x336
    frames resize / centre-crop are identities and what is left is rescale (x * 1/255) + normalise.  The fixture keeps the seed of
    the 4 uint8 frames, an 8x8-strided fp16 subsample of the processor's output and the float64 sum of its fp16 rounding."""
    from PIL import Image
    from transformers import CLIPImageProcessor
    proc = CLIPImageProcessor(do_resize=True, size={"shortest_edge": 336}, do_center_crop=True, crop_size={"height": 336, "width": 336},
                              do_rescale=True, do_normalize=True, image_mean=[0.48145466, 0.4578275, 0.40821073],
                              image_std=[0.26862954, 0.26130258, 0.27577711], do_convert_rgb=True)
    frames = np.random.default_rng(1234).integers(0, 256, size=(4, 336, 336, 3), dtype=np.uint8)
    out = proc.preprocess([Image.fromarray(f) for f in frames], return_tensors="pt")["pixel_values"].numpy()
    h = out.astype(np.float16)
    np.savez_compressed(os.path.join(OUT, "preprocess.npz"), seed=np.int64(1234), shape=np.array(frames.shape), sub=h[:, :, ::8, ::8],
                        sum64=np.float64(h.astype(np.float64).sum()), first_frame_crc=np.int64(zlib.crc32(frames[0].tobytes())))


if __name__ == "__main__":
    if not os.path.isdir("/root/reference"):
        sys.exit("authoring container only")
    torch.set_num_threads(1)
    gen_clip()
    gen_bert()
    gen_qwen2()
    gen_preprocess()
    print("transformers", transformers.__version__, "-> clip_tiny.npz bert_tiny.npz qwen2_tiny.npz preprocess.npz")
