#!/usr/bin/env python3
"""Round-2 golden vectors from the reference's OWN functions (authoring container only; same rules as tools/make_golden.py:
the reference's functions are AST-extracted / imported from /root/reference and EXECUTED here on stand-ins for the absent
dependencies; only inputs and expected outputs are written to tests/golden/).

  frame_indices.json   a2   the integer frame-index arithmetic of video_reader_thread_with_embedding
                            (inference_streaming_longva_v2.py:454-531): the function itself runs on a fake cv2 capture, the
                            indices are the positions it seeks to (`cap.set`)
  answer_prompts.json  a10  the three live prompt branches of longva_inference_with_embedding_multi_modal (:164-264, Q18)
                            rendered by the reference's conversation templates and tokenizer_image_token: prompt text, ids,
                            the [short | long] embedding block it hands to generate
  conv_templates.json  a10  get_prompt() of qwen_1_5 / qwen_1_5_ego / qwen_1_5_summarize / qwen_1_5_caption
                            (longva/conversation.py) for one user turn
  search_tree.json     a8   search_tree (utiles.py:909-935) on small TreeNode trees
  torch_kmeans.npz     a13  G3b: torch_kmeans.KMeans (torch_kmeans/clustering/kmeans.py:24-644) labels / centres / inertia
                            for 'rnd' and 'k-means++' init, num_init 1 and 4, an empty-cluster case
  kmeans_pytorch.npz   a13  kmeans_pytorch.kmeans / kmeans_predict / pairwise_* (kmeans_pytorch/__init__.py:10-209),
                            AST-extracted (the package import needs numba)
"""
import ast
import io
import contextlib
import json
import os
import random
import sys
import time
import types
from functools import partial

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG   # noqa: E402

REF, OUT = MG.REF, MG.OUT


def _stub_longva():
    for name in ["longva", "longva.model", "longva.model.language_model", "longva.model.multimodal_resampler"]:
        if name not in sys.modules:
            m = types.ModuleType(name); m.__path__ = [REF + "/" + name.replace(".", "/")]; sys.modules[name] = m
    sys.path.insert(0, REF)


def _entry_functions(names, ns):
    src = open(os.path.join(REF, "inference_streaming_longva_v2.py"), encoding="utf-8").read()
    for n in ast.parse(src).body:
        if isinstance(n, ast.FunctionDef) and n.name in names:
            exec(compile(ast.Module([n], []), "inference_streaming_longva_v2.py", "exec"), ns)
    return ns


# ---------------------------------------------------------------------------------------------------------
# a2: frame indices
# ---------------------------------------------------------------------------------------------------------
class _Cap:
    def __init__(self, total, fail_at=None):
        self.total, self.pos, self.seeks, self.fail_at = total, 0, [], fail_at

    def set(self, prop, value):
        self.pos = value
        self.seeks.append(int(value))

    def read(self):
        if self.fail_at is not None and len(self.seeks) > self.fail_at:
            return False, None
        return True, np.full((2, 2, 3), self.pos % 256, np.uint8)


class _Tqdm:
    def __init__(self, *a, **k): pass
    def update(self, n): pass
    def close(self): pass


def gen_frame_indices():
    fake_cv2 = types.SimpleNamespace(CAP_PROP_POS_FRAMES=1, COLOR_BGR2RGB=4, cvtColor=lambda f, c: f)
    ns = dict(cv2=fake_cv2, tqdm=_Tqdm, torch=torch, GREEN="", RESET="", RED="", BLUE="",
              Image=types.SimpleNamespace(fromarray=lambda a: a),
              process_images_ours=lambda imgs, proc, cfg: torch.zeros(1, 3, 2, 2) + float(imgs[0][0, 0, 0]))
    _entry_functions({"video_reader_thread_with_embedding"}, ns)
    model = types.SimpleNamespace(config=None, encode_images=lambda x: x)
    cases = []
    grid = [  # (total_frames, frame_rate, start, end, sample_rate, chunk_size, fail_at)
        (9000, 30, 0, 60, 0.1, 40, None), (9000, 30, 60, 120, 0.1, 40, None), (9000, 29, 12.5, 47.25, 0.05, 40, None),
        (9000, 30, 0, 300, 0.2, 40, None),          # 1800 sampled frames > 900 -> clamped to 200
        (9000, 30, 0, 150.02, 0.2, 40, None),       # 900 sampled frames: NOT clamped (strict >)
        (9000, 30, 0, 150.2, 0.2, 40, None),        # 901 -> clamped
        (9000, 30, 10, 11, 0.1, 40, None),          # 30 frames <= chunk_size: every frame
        (9000, 30, 10, 11.34, 0.1, 40, None),       # exactly 40 frames <= chunk_size
        (9000, 30, 10, 11.37, 0.1, 40, None),       # 41 frames > chunk_size: sampled (4 frames)
        (1000, 25, 30, 100, 0.1, 30, None),         # end beyond the video: clamped to total_frames
        (1000, 25, -3, 2, 0.5, 30, None),           # negative start -> 0
        (500, 24, 5, 20, 0.1, 40, 7),               # decoder failure after 7 reads: the bank is cut short
        (9000, 30, 0, 45, 1.0, 30, None),           # sample_rate 1: 1350 > 900 -> 200
    ]
    for (total, fps, start, end, rate, chunk, fail) in grid:
        cap = _Cap(total, fail)
        bank = ns["video_reader_thread_with_embedding"](cap, total, fps, None, model, start, end, "cpu", rate, chunk_size=chunk)
        cases.append(dict(total_frames=total, frame_rate=fps, start=start, end=end, sample_rate=rate, chunk_size=chunk, fail_after=fail,
                          seeks=cap.seeks, bank_len=len(bank), bank_first_value=[float(t.flatten()[0]) for t in bank]))
    json.dump(cases, open(os.path.join(OUT, "frame_indices.json"), "w"))
    return len(cases)


# ---------------------------------------------------------------------------------------------------------
# a10: answer prompts + conversation templates
# ---------------------------------------------------------------------------------------------------------
class _WordTok:
    """whitespace tokenizer with stable ids (crc32): enough to pin where the -200 sentinel lands"""
    bos_token_id = None

    def __call__(self, text, **kw):
        import zlib
        return types.SimpleNamespace(input_ids=[zlib.crc32(w.encode()) % 50000 + 10 for w in text.split()])

    def batch_decode(self, ids, skip_special_tokens=True):
        return ["  an answer  "]


def gen_prompts():
    _stub_longva()
    from longva.conversation import conv_templates
    from longva.mm_utils import tokenizer_image_token
    from longva.constants import IMAGE_TOKEN_INDEX, DEFAULT_IMAGE_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN
    captured = {}

    class Model:
        config = types.SimpleNamespace(mm_use_im_start_end=False)

        def get_model(self):
            return types.SimpleNamespace(embed_tokens=lambda ids: torch.zeros(len(ids), 8))

        def generate_with_image_embedding(self, input_ids, image_embeddings=None, **kw):
            captured.update(input_ids=input_ids[0].tolist(), emb=image_embeddings[0].clone(), kw={k: (v if not torch.is_tensor(v) else None) for k, v in kw.items()})
            return torch.tensor([[1, 2, 3]])

    def fake_search(tree, question, short, emb_model, emb_tok):
        return [torch.full((2, 3, 8), 7.0), torch.full((4, 3, 8), 9.0)], ["coarse summary of ten clips", "clip 17: a red cup on the kitchen table"]

    def tok_capture(prompt, tokenizer, idx, return_tensors=None):
        captured["prompt"] = prompt
        return tokenizer_image_token(prompt, tokenizer, idx, return_tensors=return_tensors)
    class TorchCpu:                 # the reference hard-codes device='cuda' (:177): same call, CPU tensor
        def __getattr__(self, k):
            return getattr(torch, k)

        def tensor(self, *a, **k):
            k.pop("device", None)
            return torch.tensor(*a, **k)
    ns = dict(torch=TorchCpu(), time=time, conv_templates=conv_templates, tokenizer_image_token=tok_capture,
              IMAGE_TOKEN_INDEX=IMAGE_TOKEN_INDEX, DEFAULT_IMAGE_TOKEN=DEFAULT_IMAGE_TOKEN, DEFAULT_IM_START_TOKEN=DEFAULT_IM_START_TOKEN,
              DEFAULT_IM_END_TOKEN=DEFAULT_IM_END_TOKEN, fast_search_tree_multi_modal_with_embedding=fake_search,
              args=types.SimpleNamespace(temperature=0.2, top_p=None, num_beams=1))
    torch.Tensor.cuda = lambda s, *a, **k: s
    _entry_functions({"longva_inference_with_embedding_multi_modal"}, ns)
    fn = ns["longva_inference_with_embedding_multi_modal"]
    short = [torch.full((1, 3, 8), float(i)) for i in range(5)]
    history = ('Based on the current user\'s question, the most relevant historical contextual conversation records are: '
               '"\nConversation content on 2024-05-01:[|User|]: where is the cup; [|AI|]: on the table\n".')
    cases = []
    for name, tree, hist in [("history_and_caption", ["tree"], history), ("no_history", ["tree"], None), ("history_no_tree", None, history),
                             ("no_history_no_tree", None, None)]:
        for conv_mode in ("qwen_1_5",):
            captured.clear()
            out, _, _ = fn("What is on the table?", 8, conv_mode, Model(), None, _WordTok(), None, None, short, tree, history_prompt=hist)
            cases.append(dict(case=name, conv_mode=conv_mode, history_prompt=hist, has_tree=tree is not None, prompt=captured["prompt"],
                              input_ids=captured["input_ids"], n_sentinels=captured["input_ids"].count(IMAGE_TOKEN_INDEX),
                              emb_shape=list(captured["emb"].shape), emb_first_col=captured["emb"][:, 0].tolist(),
                              gen_kwargs={k: v for k, v in captured["kw"].items() if k != "modalities"}, output=out))
    json.dump(dict(question="What is on the table?", cases=cases), open(os.path.join(OUT, "answer_prompts.json"), "w"))
    # conversation templates for one user turn (captions / summaries / answers)
    tmpl = {}
    for k in ("qwen_1_5", "qwen_1_5_ego", "qwen_1_5_summarize", "qwen_1_5_caption"):
        if k in conv_templates:
            c = conv_templates[k].copy()
            c.append_message(c.roles[0], "<image>\nwhat do you see?")
            c.append_message(c.roles[1], None)
            one = c.get_prompt()
            c = conv_templates[k].copy()
            c.append_message(c.roles[0], "first question")
            c.append_message(c.roles[1], "first answer")
            c.append_message(c.roles[0], "second question")
            c.append_message(c.roles[1], None)
            tmpl[k] = dict(one_turn=one, two_turns=c.get_prompt(), roles=list(c.roles))
    json.dump(tmpl, open(os.path.join(OUT, "conv_templates.json"), "w"))
    return len(cases)


# ---------------------------------------------------------------------------------------------------------
# a8: search_tree
# ---------------------------------------------------------------------------------------------------------
def gen_search_tree(ns):
    N, f = ns["TreeNode"], ns["search_tree"]
    g = torch.Generator().manual_seed(5)
    cases = []
    for trial, (fan, depth) in enumerate([((3, 2), 2), ((2, 2, 2), 3), ((4,), 1), ((), 0)]):
        counter = [0]

        def build(level):
            v = float(counter[0]); counter[0] += 1
            node = N(torch.randn(3, 2, 6, generator=g) + v, depth=depth - level)
            if level < depth:
                node.children = [build(level + 1) for _ in range(fan[level])]
            return node
        root = build(0)
        query = torch.randn(5, 6, generator=g)
        path = f(root, query)

        def dump(n):
            return dict(centroids=n.centroids.tolist(), children=[dump(c) for c in n.children])
        cases.append(dict(tree=dump(root), query=query.tolist(), path=[p.tolist() for p in path]))
    json.dump(cases, open(os.path.join(OUT, "search_tree.json"), "w"))
    return len(cases)


# ---------------------------------------------------------------------------------------------------------
# a13: torch_kmeans (imports) and kmeans_pytorch (AST-extracted)
# ---------------------------------------------------------------------------------------------------------
def _blobs(seed, bs, n, d, k, spread=0.15):
    g = torch.Generator().manual_seed(seed)
    centers = torch.randn(bs, k, d, generator=g) * 2
    lab = torch.randint(0, k, (bs, n), generator=g)
    return (centers.gather(1, lab[:, :, None].expand(bs, n, d)) + spread * torch.randn(bs, n, d, generator=g)).contiguous()


def gen_torch_kmeans():
    sys.path.insert(0, REF)
    from torch_kmeans import KMeans
    out = {}
    specs = [("rnd1", dict(init_method="rnd", num_init=1), 0, 2, 60, 16, 4), ("rnd4", dict(init_method="rnd", num_init=4), 1, 1, 80, 32, 5),
             ("pp1", dict(init_method="k-means++", num_init=1), 2, 2, 50, 8, 3), ("pp3", dict(init_method="k-means++", num_init=3), 3, 1, 64, 24, 6),
             ("rnd1_iter3", dict(init_method="rnd", num_init=1, max_iter=3), 4, 1, 70, 12, 7)]
    for name, kw, seed, bs, n, d, k in specs:
        x = _blobs(seed, bs, n, d, k)
        km = KMeans(n_clusters=k, seed=123, verbose=False, **kw)
        r = km.fit(x)._result
        out[name + ".x"] = x.numpy(); out[name + ".k"] = np.asarray(k)
        out[name + ".labels"] = r.labels.numpy(); out[name + ".centers"] = r.centers.numpy(); out[name + ".inertia"] = r.inertia.numpy()
        out[name + ".predict"] = km.predict(x + 0.01).numpy()
        out[name + ".kw"] = np.asarray(json.dumps(kw))
    # explicit centres incl. one that captures no point (empty cluster -> its centre becomes the zero vector, utils.py:66)
    x = _blobs(9, 1, 40, 6, 3)
    c0 = torch.cat([x[:, :3], torch.full((1, 1, 6), 50.0)], dim=1)
    km = KMeans(n_clusters=4, num_init=1, seed=123, verbose=False, max_iter=5)
    r = km.fit(x, centers=c0.clone())._result
    out.update({"given.x": x.numpy(), "given.k": np.asarray(4), "given.centers0": c0.numpy(), "given.labels": r.labels.numpy(),
                "given.centers": r.centers.numpy(), "given.inertia": r.inertia.numpy(), "given.kw": np.asarray(json.dumps(dict(num_init=1, max_iter=5)))})
    np.savez_compressed(os.path.join(OUT, "torch_kmeans.npz"), **out)
    return len(specs) + 1


def gen_kmeans_pytorch():
    src = open(os.path.join(REF, "kmeans_pytorch/__init__.py"), encoding="utf-8").read()
    ns = dict(np=np, torch=torch, partial=partial, tqdm=lambda **k: types.SimpleNamespace(set_postfix=lambda **k: None, update=lambda: None))
    for n in ast.parse(src).body:
        if isinstance(n, ast.FunctionDef) and n.name in ("initialize", "kmeans", "kmeans_predict", "pairwise_distance", "pairwise_cosine"):
            exec(compile(ast.Module([n], []), "kmeans_pytorch/__init__.py", "exec"), ns)
    out = {}
    cpu = torch.device("cpu")
    for name, dist, seed, n, d, k, it in [("euclid", "euclidean", 11, 90, 16, 4, 0), ("cosine", "cosine", 12, 70, 12, 3, 0), ("limit2", "euclidean", 13, 60, 8, 5, 2)]:
        x = _blobs(seed, 1, n, d, k, spread=0.3)[0]
        ids, cent = ns["kmeans"](X=x, num_clusters=k, distance=dist, tqdm_flag=False, iter_limit=it, device=cpu, seed=seed)
        out[name + ".x"] = x.numpy(); out[name + ".k"] = np.asarray(k); out[name + ".seed"] = np.asarray(seed); out[name + ".iter_limit"] = np.asarray(it)
        out[name + ".ids"] = ids.numpy(); out[name + ".centers"] = cent.numpy()
        out[name + ".predict"] = ns["kmeans_predict"](x + 0.02, cent, distance=dist, device=cpu, tqdm_flag=False).numpy()
        out[name + ".distance"] = np.asarray(dist)
    # resume from given centres (the closest data point to each becomes the initial state, :77-86)
    x = _blobs(21, 1, 50, 10, 3, spread=0.3)[0]
    c0 = x[[3, 17, 41]] + 0.05
    ids, cent = ns["kmeans"](X=x, num_clusters=3, cluster_centers=c0.clone(), tqdm_flag=False, device=cpu)
    out.update({"resume.x": x.numpy(), "resume.c0": c0.numpy(), "resume.ids": ids.numpy(), "resume.centers": cent.numpy()})
    # an empty cluster: the duplicate initial centre loses every tie, its refill is X[torch.randint(len(X), (1,))] (:104-105)
    x = _blobs(22, 1, 30, 6, 2, spread=0.2)[0]
    x[7] = x[2]
    np.random.seed(0)
    while True:                                          # find a numpy seed whose initial choice picks rows 2 and 7 (duplicates)
        s = np.random.randint(0, 10 ** 6)
        np.random.seed(s)
        pick = np.random.choice(30, 3, replace=False)
        if 2 in pick and 7 in pick and list(pick).index(2) < list(pick).index(7):
            break
    torch.manual_seed(77)
    ids, cent = ns["kmeans"](X=x, num_clusters=3, tqdm_flag=False, device=cpu, seed=s, iter_limit=4)
    out.update({"empty.x": x.numpy(), "empty.seed": np.asarray(s), "empty.torch_seed": np.asarray(77), "empty.ids": ids.numpy(), "empty.centers": cent.numpy()})
    a, b = _blobs(31, 1, 9, 5, 2)[0], _blobs(32, 1, 4, 5, 2)[0]
    out.update({"pair.a": a.numpy(), "pair.b": b.numpy(), "pair.dist": ns["pairwise_distance"](a, b, device=cpu, tqdm_flag=False).numpy(),
                "pair.cos": ns["pairwise_cosine"](a, b, device=cpu).numpy()})
    np.savez_compressed(os.path.join(OUT, "kmeans_pytorch.npz"), **out)
    return 6


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present: golden vectors can only be generated in the authoring container")
    torch.set_num_threads(1)
    ns = MG.load_reference_namespace()
    with contextlib.redirect_stdout(io.StringIO()):
        a = gen_frame_indices()
        b = gen_prompts()
        c = gen_search_tree(ns)
        d = gen_torch_kmeans()
        e = gen_kmeans_pytorch()
    print(f"wrote frame_indices ({a}), answer_prompts ({b}), search_tree ({c}), torch_kmeans ({d}), kmeans_pytorch ({e}) to {os.path.normpath(OUT)}")


if __name__ == "__main__":
    main()
