"""Seeded synthetic inputs for benchmarks / smoke tests (no datasets or checkpoints are reachable offline):
a scene-structured uint8 frame stream, synthetic captions / dialogue documents, and a captioner stand-in
(LLM captioning is excluded from the timed region: the metric names encode + select + retrieve + prefill)."""
import types
import zlib

import numpy as np
import torch

VOCAB = ("kitchen table window street car person dog tree river bridge market phone laptop book cup door stairs garden bus "
         "bicycle shelf counter sink stove lamp chair sofa screen keyboard bottle plate road sign crowd tower park bench").split()


def frame_stream(n_frames, seed=1234, h=336, w=336, scene_len=40, noise=6, start=0):
    """uint8 [n, h, w, 3] = frames [start, start + n) of the seeded stream: one random 'scene' image per `scene_len` frames plus
    small per-frame noise, so the selective-frame k-means sees scene structure like a real video (pure iid noise has none).
    Scene s and the noise of frame i are drawn from their own generators (seed, s) / (seed, i), so any slice of the stream can be
    produced on its own: rank r of a sharded run generates exactly the frames the single-GPU run has at the same positions."""
    out = np.empty((n_frames, h, w, 3), np.uint8)
    scene, scene_id = None, -1
    for k in range(n_frames):
        i = start + k
        if i // scene_len != scene_id:
            scene_id = i // scene_len
            scene = np.random.default_rng([seed, 0, scene_id]).integers(0, 256, (h, w, 3), dtype=np.uint8).astype(np.int16)
        d = np.random.default_rng([seed, 1, i]).integers(-noise, noise + 1, (h, w, 3), dtype=np.int16)
        out[k] = np.clip(scene + d, 0, 255).astype(np.uint8)
    return out


def caption(i, seed=0, words=24):
    rng = np.random.default_rng(seed * 100003 + i)
    return f"clip {i}: " + " ".join(VOCAB[j] for j in rng.integers(0, len(VOCAB), words))


class SyntheticCaptioner:
    """Stand-in for the second LongVA replica that captions chunks / summarises merges (reference utiles.py:539-559,
    591-607).  Same call surface (`generate_with_image_embedding`, `.device`, `.config`).  The "generated" token is a hash of the
    CONTENT it was given (the first feature row of the chunk, or the summary prompt's ids), which the paired tokenizer turns into
    a deterministic synthetic caption: the text of a chunk then depends on the chunk alone, not on how many captions this
    process produced before it — a sharded run and the single-GPU run caption the same chunk identically."""

    def __init__(self, device="cuda"):
        self.device = device
        self.config = types.SimpleNamespace(mm_use_im_start_end=False)
        self.n = 0

    def generate_with_image_embedding(self, ids, image_embeddings=None, **kw):
        self.n += 1
        if image_embeddings is not None:
            key = image_embeddings[0].reshape(-1)[:16].detach().to("cpu", torch.float32).numpy().tobytes()
        else:
            key = torch.as_tensor(ids).reshape(-1).to("cpu", torch.int64).numpy().tobytes()
        return torch.tensor([[zlib.crc32(key) % 1000003]])


class SyntheticTokenizer:
    bos_token_id = None

    def __call__(self, text, **kw):
        return types.SimpleNamespace(input_ids=[zlib.crc32(w.encode()) % 30000 + 5 for w in text.split()][:512])

    def batch_decode(self, ids, skip_special_tokens=True):
        return [caption(int(ids[0][0]))]
