"""Real-checkpoint loading for the entry point: what `load_pretrained_model` (reference longva/model/builder.py:27-285) and the
`AutoModel.from_pretrained(embedding_model_id)` line (inference_streaming_longva_v2.py:703-705) do upstream, feeding the HIP
modules of this package instead of `transformers` classes.

A LongVA / LLaVA-Qwen checkpoint directory holds `config.json`, HF-sharded weights (`model.safetensors.index.json` + shards, a single
`model.safetensors`, or `pytorch_model*.bin`) and the tokenizer files.  Its state dict uses the transformers parameter names with the
LLaVA prefixes:
    model.embed_tokens.* / model.layers.N.* / model.norm.* / lm_head.*      -> llm.Qwen2Model (names unchanged)
    model.mm_projector.{0,2}.{weight,bias}                                  -> vision.MMProjector(prefix="model.mm_projector.")
    model.vision_tower.vision_tower.vision_model.*                          -> vision.CLIPVisionTower (when the checkpoint carries the
                                                                              tower: unfreeze_mm_vision_tower / mm_tunable_parts)
otherwise the tower comes from `config.mm_vision_tower` (clip_encoder.py:41: `CLIPVisionModel.from_pretrained(name)`), which must be
a LOCAL directory here (no network): keys `vision_model.*`.
Everything is read tensor by tensor (safetensors `safe_open`) so a 15 GB checkpoint never exists twice in host memory."""
import glob
import json
import os

import torch

from . import llm as LM, text as T, vision as V


class CheckpointError(RuntimeError):
    pass


def read_config(path):
    f = os.path.join(path, "config.json")
    if not os.path.isfile(f):
        raise CheckpointError(f"{path}: no config.json — not a HF-format checkpoint directory")
    return json.load(open(f))


def weight_files(path):
    """[(file, kind)] of a checkpoint directory in a stable order; prefers safetensors."""
    if os.path.isfile(path):
        return [(path, "safetensors" if path.endswith(".safetensors") else "bin")]
    for index, kind in (("model.safetensors.index.json", "safetensors"), ("pytorch_model.bin.index.json", "bin")):
        idx = os.path.join(path, index)
        if os.path.isfile(idx):
            files = sorted(set(json.load(open(idx))["weight_map"].values()))
            missing = [f for f in files if not os.path.isfile(os.path.join(path, f))]
            if missing:
                raise CheckpointError(f"{path}: shards listed in {index} are missing: {missing[:3]}")
            return [(os.path.join(path, f), kind) for f in files]
    st = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if st:
        return [(f, "safetensors") for f in st]
    bins = sorted(glob.glob(os.path.join(path, "pytorch_model*.bin")))
    if bins:
        return [(f, "bin") for f in bins]
    raise CheckpointError(f"{path}: no *.safetensors / pytorch_model*.bin weights found")


def read_state_dict(path, keep=None, dtype=torch.float16):
    """{name: CPU tensor} of every weight whose name passes `keep(name)` (all when None), cast to `dtype` if floating point."""
    out = {}
    for f, kind in weight_files(path):
        if kind == "safetensors":
            from safetensors import safe_open
            with safe_open(f, framework="pt", device="cpu") as sf:
                for k in sf.keys():
                    if keep is None or keep(k):
                        t = sf.get_tensor(k)
                        out[k] = t.to(dtype) if t.is_floating_point() and dtype is not None else t
        else:
            sd = torch.load(f, map_location="cpu", weights_only=True)
            for k, t in sd.items():
                if keep is None or keep(k):
                    out[k] = t.to(dtype) if t.is_floating_point() and dtype is not None else t
            del sd
    return out


def load_tokenizer(path, use_fast=None):
    """A checkpoint directory's own tokenizer, local files only.  `use_fast=False` is what the reference passes for the LLaVA / Qwen2
    tokenizer (builder.py:93,177: AutoTokenizer.from_pretrained(model_path, use_fast=False)): it selects the slow tokenizer on the reference's
    pinned transformers 4.37.2 (transformers >= 5, installed here, only ships the fast classes and ignores it).  The embedding-model
    tokenizers are loaded with the library default upstream (utiles.py:1595,1885: AutoTokenizer.from_pretrained(embedding_model_id)), so
    `load_bert` leaves `use_fast` unset."""
    try:
        from transformers import AutoTokenizer
        kw = {} if use_fast is None else {"use_fast": use_fast}
        return AutoTokenizer.from_pretrained(path, local_files_only=True, **kw)
    except Exception as e:
        raise CheckpointError(f"{path}: cannot load the tokenizer ({e})") from e


def _eos_ids(path, cfg, tokenizer):
    for f in ("generation_config.json",):
        p = os.path.join(path, f)
        if os.path.isfile(p):
            e = json.load(open(p)).get("eos_token_id")
            if e is not None:
                return e
    return cfg.get("eos_token_id", getattr(tokenizer, "eos_token_id", None))


def clip_config_from(cfg):
    c = cfg.get("vision_config", cfg)                      # a full CLIPModel config nests the tower under vision_config
    return V.CLIPVisionConfigLite(hidden=c["hidden_size"], layers=c["num_hidden_layers"], heads=c["num_attention_heads"],
                                  intermediate=c["intermediate_size"], patch=c["patch_size"], image_size=c["image_size"],
                                  eps=c.get("layer_norm_eps", 1e-5))


VT_PREFIX = "model.vision_tower.vision_tower."


def load_longva(model_path, device="cuda", vision_tower_path=None, max_seq=65536, micro_batch=256, tokenizer=True):
    """-> (LlavaQwenForCausalLM, tokenizer or None, CLIPVisionConfigLite).  Mirrors load_pretrained_model for the llava-qwen branch:
    Qwen2 language model + mm_projector from the checkpoint, CLIP tower from the checkpoint if it carries one, else from
    `vision_tower_path` / `config.mm_vision_tower` (a local directory)."""
    cfg = read_config(model_path)
    if cfg.get("mm_projector_type", "mlp2x_gelu") != "mlp2x_gelu":
        raise CheckpointError(f"mm_projector_type {cfg.get('mm_projector_type')!r}: only mlp2x_gelu (LongVA) is built")
    if cfg.get("mm_resampler_type") not in (None, "identity"):
        raise CheckpointError(f"mm_resampler_type {cfg.get('mm_resampler_type')!r}: only the identity resampler is built")
    qc = LM.Qwen2ConfigLite(hidden=cfg["hidden_size"], layers=cfg["num_hidden_layers"], heads=cfg["num_attention_heads"],
                            kv_heads=cfg.get("num_key_value_heads", cfg["num_attention_heads"]), intermediate=cfg["intermediate_size"],
                            vocab=cfg["vocab_size"], eps=cfg.get("rms_norm_eps", 1e-6), rope_theta=cfg.get("rope_theta", 1e6),
                            tokenizer_model_max_length=cfg.get("tokenizer_model_max_length"))
    qc.mm_use_im_start_end = bool(cfg.get("mm_use_im_start_end", False))
    is_lm = lambda k: k.startswith(("model.embed_tokens.", "model.layers.", "model.norm.", "lm_head."))
    sd = read_state_dict(model_path, keep=lambda k: is_lm(k) or k.startswith("model.mm_projector.") or k.startswith(VT_PREFIX))
    if "lm_head.weight" not in sd and not cfg.get("tie_word_embeddings", False):
        raise CheckpointError(f"{model_path}: lm_head.weight missing and tie_word_embeddings is false")
    proj = {k: sd.pop(k) for k in [k for k in sd if k.startswith("model.mm_projector.")]}
    if not proj:
        raise CheckpointError(f"{model_path}: no model.mm_projector.* weights (pretrain-only adapters are loaded with --model-base upstream; merge them first)")
    tower_sd = {k[len(VT_PREFIX):]: sd.pop(k) for k in [k for k in sd if k.startswith(VT_PREFIX)]}
    if tower_sd:
        vt_dir = vision_tower_path or cfg.get("mm_vision_tower")
        vc = clip_config_from(read_config(vt_dir)) if vt_dir and os.path.isdir(vt_dir) else V.CLIPVisionConfigLite(**V.VIT_L_336)
    else:
        vt_dir = vision_tower_path or cfg.get("mm_vision_tower")
        if not vt_dir or not os.path.isdir(vt_dir):
            raise CheckpointError(f"the checkpoint has no vision-tower weights and mm_vision_tower={vt_dir!r} is not a local directory "
                                  "(pass --vision_tower <dir of openai/clip-vit-large-patch14-336>)")
        vc = clip_config_from(read_config(vt_dir))
        tower_sd = read_state_dict(vt_dir, keep=lambda k: k.startswith("vision_model."))
    tower = V.CLIPVisionTower(tower_sd, vc, select_layer=cfg.get("mm_vision_select_layer", -2),
                              select_feature=cfg.get("mm_vision_select_feature", "patch"), device=device)
    del tower_sd
    enc = V.FrameEncoder(tower, V.MMProjector(proj, device=device, prefix="model.mm_projector."), micro_batch=micro_batch)
    lm = LM.Qwen2Model(sd, qc, device=device, max_seq=max_seq, consume=True)
    del sd
    tok = load_tokenizer(model_path, use_fast=False) if tokenizer else None      # builder.py:93,177
    model = LM.LlavaQwenForCausalLM(lm, enc, eos_token_id=_eos_ids(model_path, cfg, tok))
    gc = os.path.join(model_path, "generation_config.json")      # what HF's generate falls back to for arguments the caller does not pass
    if os.path.isfile(gc):
        model.generation_config = {k: v for k, v in json.load(open(gc)).items() if k in ("temperature", "top_k", "top_p", "repetition_penalty", "do_sample")}
    return model, tok, vc


def load_bert(path, device="cuda"):
    """-> (BertEncoder, tokenizer): the plain `AutoModel.from_pretrained(embedding_model_id)` of the reference (:705; mxbai-colbert-
    large-v1 is a BertModel) or the transformer under a sentence-transformers MiniLM directory."""
    cfg = read_config(path)
    bc = T.BertConfigLite(hidden=cfg["hidden_size"], layers=cfg["num_hidden_layers"], heads=cfg["num_attention_heads"],
                          intermediate=cfg["intermediate_size"], vocab=cfg["vocab_size"], max_pos=cfg["max_position_embeddings"],
                          eps=cfg.get("layer_norm_eps", 1e-12))
    sd = read_state_dict(path, keep=lambda k: "embeddings." in k or "encoder.layer." in k)
    prefix = "bert." if any(k.startswith("bert.") for k in sd) else ""
    return T.BertEncoder(sd, bc, device=device, prefix=prefix), load_tokenizer(path)
