"""Seeded synthetic inputs for benchmarks / smoke tests (no datasets or checkpoints are reachable offline):
a scene-structured uint8 frame stream, synthetic captions / dialogue documents, and a captioner stand-in
(LLM captioning is excluded from the timed region: the metric names encode + select + retrieve + prefill)."""
import types
import zlib

import numpy as np
import torch

VOCAB = ("kitchen table window street car person dog tree river bridge market phone laptop book cup door stairs garden bus "
         "bicycle shelf counter sink stove lamp chair sofa screen keyboard bottle plate road sign crowd tower park bench").split()


def frame_stream(n_frames, seed=1234, h=336, w=336, scene_len=40, noise=6):
    """uint8 [n, h, w, 3]: one random 'scene' image per `scene_len` frames plus small per-frame noise, so the
    selective-frame k-means sees scene structure like a real video (pure iid noise has none)."""
    rng = np.random.default_rng(seed)
    n_scenes = (n_frames + scene_len - 1) // scene_len
    scenes = rng.integers(0, 256, (n_scenes, h, w, 3), dtype=np.uint8)
    out = np.empty((n_frames, h, w, 3), np.uint8)
    for i in range(n_frames):
        d = rng.integers(-noise, noise + 1, (h, w, 3), dtype=np.int16)
        out[i] = np.clip(scenes[i // scene_len].astype(np.int16) + d, 0, 255).astype(np.uint8)
    return out


def caption(i, seed=0, words=24):
    rng = np.random.default_rng(seed * 100003 + i)
    return f"clip {i}: " + " ".join(VOCAB[j] for j in rng.integers(0, len(VOCAB), words))


class SyntheticCaptioner:
    """Stand-in for the second LongVA replica that captions chunks / summarises merges (reference utiles.py:539-559,
    591-607).  Same call surface (`generate_with_image_embedding`, `.device`, `.config`); returns a counter the paired
    tokenizer turns into a deterministic synthetic caption."""

    def __init__(self, device="cuda"):
        self.device = device
        self.config = types.SimpleNamespace(mm_use_im_start_end=False)
        self.n = 0

    def generate_with_image_embedding(self, ids, image_embeddings=None, **kw):
        self.n += 1
        return torch.tensor([[self.n - 1]])


class SyntheticTokenizer:
    bos_token_id = None

    def __call__(self, text, **kw):
        return types.SimpleNamespace(input_ids=[zlib.crc32(w.encode()) % 30000 + 5 for w in text.split()][:512])

    def batch_decode(self, ids, skip_special_tokens=True):
        return [caption(int(ids[0][0]))]
