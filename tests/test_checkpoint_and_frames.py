"""Real-checkpoint path of the entry point (VERDICT r01 missing #2, ADVICE medium): a tiny HF-FORMAT LongVA checkpoint (config.json,
sharded safetensors + index / pytorch_model.bin, tokenizer.json) is written from the HF golden weights (tests/golden/clip_tiny.npz,
qwen2_tiny.npz, bert_tiny.npz — real `transformers` modules' parameters and outputs) under the LLaVA key prefixes, loaded through
streamchat_amd/checkpoint.py (what longva/model/builder.py:27-285 does upstream), and must reproduce the golden outputs.
CPU part: the resize + centre-crop of arbitrary-resolution frames equals CLIPImageProcessor's (reference utiles.py:71-87)."""
import json
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


def test_resize_center_crop_equals_clip_image_processor():
    from transformers import CLIPImageProcessor
    from streamchat_amd.mm_utils import resize_center_crop_u8
    proc = CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336}, do_rescale=False, do_normalize=False)
    rng = np.random.default_rng(0)
    for shp in [(480, 640), (1080, 1920), (640, 480), (336, 336), (500, 333), (337, 336), (720, 1280)]:
        x = rng.integers(0, 256, shp + (3,), dtype=np.uint8)
        ref = proc.preprocess([x], return_tensors="np")["pixel_values"][0]
        got = resize_center_crop_u8(x)
        assert got.shape == (336, 336, 3) and got.dtype == np.uint8
        assert np.array_equal(ref.astype(np.uint8), got.transpose(2, 0, 1)), shp
    with pytest.raises(ValueError):
        resize_center_crop_u8(np.zeros((4, 4), np.uint8))


def test_checkpoint_errors_are_loud(tmp_path):
    from streamchat_amd import checkpoint as CK
    with pytest.raises(CK.CheckpointError):
        CK.read_config(str(tmp_path))
    json.dump({"hidden_size": 8}, open(tmp_path / "config.json", "w"))
    with pytest.raises(CK.CheckpointError):
        CK.weight_files(str(tmp_path))
    json.dump({"weight_map": {"a": "model-00001-of-00002.safetensors"}}, open(tmp_path / "model.safetensors.index.json", "w"))
    with pytest.raises(CK.CheckpointError):
        CK.weight_files(str(tmp_path))


def _write_tokenizer(path, vocab_size=512):
    from tokenizers import Tokenizer, models, pre_tokenizers
    vocab = {"<unk>": 0, "<eos>": 1}
    for w in "what is on the table where did I leave red cup <|im_start|> <|im_end|> system user assistant You are a helpful .".split():
        vocab.setdefault(w, len(vocab))
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    tok.save(os.path.join(path, "tokenizer.json"))
    json.dump({"tokenizer_class": "PreTrainedTokenizerFast", "eos_token": "<eos>", "unk_token": "<unk>"}, open(os.path.join(path, "tokenizer_config.json"), "w"))


def _write_longva(tmp, fmt, tower_inside):
    from safetensors.torch import save_file
    clip, qw = np.load(os.path.join(G, "clip_tiny.npz")), np.load(os.path.join(G, "qwen2_tiny.npz"))
    t = lambda a: torch.from_numpy(np.array(a)).half().contiguous()
    lm = {k[3:]: t(qw[k]) for k in qw.files if k.startswith("lm.")}
    proj = {"model.mm_projector." + k[5:]: t(clip[k]) for k in clip.files if k.startswith("proj.")}
    tower = {k[4:]: t(clip[k]) for k in clip.files if k.startswith("vit.")}
    vt = os.path.join(tmp, "clip-vit-tiny")
    os.makedirs(vt)
    vcfg = dict(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256, patch_size=14, image_size=56, layer_norm_eps=1e-5)
    json.dump({"model_type": "clip", "vision_config": vcfg}, open(os.path.join(vt, "config.json"), "w"))
    if not tower_inside:
        save_file(tower, os.path.join(vt, "model.safetensors"))
    d = os.path.join(tmp, "LongVA-tiny")
    os.makedirs(d)
    cfg = dict(model_type="llava_qwen", hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, intermediate_size=512,
               vocab_size=512, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=False, mm_projector_type="mlp2x_gelu", mm_vision_select_layer=-2,
               mm_vision_select_feature="patch", mm_vision_tower=vt, mm_use_im_start_end=False, tokenizer_model_max_length=4096, eos_token_id=1)
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    allw = dict(lm, **proj)
    if tower_inside:
        allw.update({"model.vision_tower.vision_tower." + k: v for k, v in tower.items()})
    keys = sorted(allw)
    if fmt == "safetensors":
        a, b = {k: allw[k] for k in keys[::2]}, {k: allw[k] for k in keys[1::2]}
        save_file(a, os.path.join(d, "model-00001-of-00002.safetensors"))
        save_file(b, os.path.join(d, "model-00002-of-00002.safetensors"))
        json.dump({"metadata": {}, "weight_map": {**{k: "model-00001-of-00002.safetensors" for k in a}, **{k: "model-00002-of-00002.safetensors" for k in b}}},
                  open(os.path.join(d, "model.safetensors.index.json"), "w"))
    else:
        torch.save(allw, os.path.join(d, "pytorch_model.bin"))
    _write_tokenizer(d)
    return d, clip, qw


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,tower_inside", [("safetensors", False), ("safetensors", True), ("bin", False)])
def test_tiny_longva_checkpoint_loads_and_reproduces_hf_golden(tmp_path, fmt, tower_inside):
    from streamchat_amd import checkpoint as CK
    from tests._tol import assert_close_fp16
    d, clip, qw = _write_longva(str(tmp_path), fmt, tower_inside)
    model, tok, vc = CK.load_longva(d, device="cuda:0", max_seq=256)
    assert (vc.hidden, vc.layers, vc.image_size) == (128, 3, 56) and model.config.rope_theta == 1e6 and model.eos_token_id == 1
    feats = model.encode_images(torch.from_numpy(clip["pixel_values"]).cuda().half())
    assert_close_fp16(feats, torch.from_numpy(clip["projected"]), what=f"loaded tower+projector ({fmt}, tower_inside={tower_inside}) vs HF golden")
    logits = model.lm.forward(torch.from_numpy(qw["inputs_embeds"]).cuda().half(), last_only=False)
    assert_close_fp16(logits, torch.from_numpy(qw["logits"]), what="loaded Qwen2 prefill logits vs HF golden")
    # the checkpoint's own tokenizer drives the prompt path; generation stops at the checkpoint's EOS id
    ids = torch.tensor([tok("what is on the table").input_ids])
    assert ids.shape[1] == 5 and int(ids.max()) < 512
    out = model.generate_with_image_embedding(ids, image_embeddings=None, do_sample=False, max_new_tokens=6)
    assert out.shape[1] <= 6 and (1 not in out[0, :-1].tolist())
    # generation_config.json: the defaults of generate (HF semantics) and a list of EOS ids
    assert model.generation_config == {}
    json.dump(dict(do_sample=True, temperature=0.7, top_k=20, top_p=0.8, repetition_penalty=1.05, eos_token_id=[1, 7], bos_token_id=0),
              open(os.path.join(d, "generation_config.json"), "w"))
    m2, _, _ = CK.load_longva(d, device="cuda:0", max_seq=256, tokenizer=False)
    assert m2.generation_config == dict(do_sample=True, temperature=0.7, top_k=20, top_p=0.8, repetition_penalty=1.05) and m2.eos_token_id == [1, 7]
    from streamchat_amd.llm import resolve_sampling
    assert tuple(resolve_sampling(m2.generation_config, True, 0.2, None)) == (0.2, 20, 1.0, 1.05)       # the reference's call on such a checkpoint
    torch.manual_seed(0)
    sampled = m2.generate_with_image_embedding(ids, image_embeddings=None, do_sample=True, temperature=0.2, top_p=None, max_new_tokens=6)
    assert 1 <= sampled.shape[1] <= 6 and all(0 <= t < 512 for t in sampled[0].tolist())


@pytest.mark.gpu
def test_bert_checkpoint_loads(tmp_path):
    from safetensors.torch import save_file
    from streamchat_amd import checkpoint as CK
    from tests._tol import assert_close_fp16
    b = np.load(os.path.join(G, "bert_tiny.npz"))
    sd = {k: torch.from_numpy(np.array(b[k])).contiguous() for k in b.files if k.startswith("bert.")}        # "bert."-prefixed like BertForMaskedLM exports
    save_file(sd, str(tmp_path / "model.safetensors"))
    json.dump(dict(model_type="bert", hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256, vocab_size=500,
                   max_position_embeddings=64, layer_norm_eps=1e-12), open(tmp_path / "config.json", "w"))
    _write_tokenizer(str(tmp_path))
    enc, tok = CK.load_bert(str(tmp_path), device="cuda:0")
    ids, mask = torch.from_numpy(b["input_ids"]), torch.from_numpy(b["attention_mask"])
    out = enc(input_ids=ids.cuda(), attention_mask=mask.cuda()).last_hidden_state
    m = mask.bool()
    assert_close_fp16(out.float().cpu()[m], torch.from_numpy(b["last_hidden_state"])[m], what="loaded BERT vs HF golden")
    assert tok("where is the cup").input_ids


@pytest.mark.gpu
def test_entry_point_real_mode_refuses_missing_checkpoints(tmp_path):
    import inference_streaming_longva_v2 as E
    argv = ["--video_dir", str(tmp_path), "--model_name", str(tmp_path / "nope"), "--memory_basic_dir", str(tmp_path), "--save_file", str(tmp_path / "r.json"),
            "--annotations", str(tmp_path / "a.json"), "--language", "en"]
    with pytest.raises(SystemExit):
        E.build_models(E.parse_args(argv))                              # no --embedding_model_id / --sentence_model_id
    from streamchat_amd.checkpoint import CheckpointError
    with pytest.raises(CheckpointError):
        E.build_models(E.parse_args(argv + ["--embedding_model_id", str(tmp_path), "--sentence_model_id", str(tmp_path)]))
