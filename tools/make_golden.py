#!/usr/bin/env python3
"""Generate golden vectors from the reference's OWN functions (authoring container only).

The reference (`/root/reference`) is Python whose module import fails here (gradio /
langchain / openai absent), so the pure-tensor functions are AST-extracted from
`utiles.py` and executed in a namespace holding only torch / numpy / random
(SURVEY.md §8(c), Appendix B).  Nothing of the reference's source is written to disk:
only INPUTS and EXPECTED OUTPUTS go to `tests/golden/*.npz|json`.

Run:  python tools/make_golden.py            (needs /root/reference; never runs on the GPU box)

Fixtures (IDs follow SURVEY.md §8(c)):
  G3  kmeans_*.npz      weighted_kmeans_feature (utiles.py:291-330) incl. per-iteration trace,
                        empty-cluster reseed case and the T<=K 3-tuple case
  G4  forgetting.npz    calculate_forgetting_probabilities / select_data_without_replacement
                        (utiles.py:251-262)
  G5  tree_trace.json   fast_building_memory_tree_summarize_token (utiles.py:489-620) with a
                        fake summariser: top-level (depth, shape, #children) after each update
  G6  search.json       fast_search_tree_multi_modal_with_embedding (utiles.py:685-788) with a
                        fake embedding model
  G10 count_nodes.json  count_nodes_by_depth (utiles.py:1002-1011)
"""
import ast
import json
import math
import os
import random
import sys
import time
import types
from collections import defaultdict

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

WANT = {
    "TreeNode", "MultimodalTreeNode", "calculate_forgetting_probabilities",
    "select_data_without_replacement", "compress_spatial_features", "weighted_kmeans_feature",
    "k_means_clustering", "fast_building_memory_tree_summarize_token",
    "fast_search_tree_multi_modal_with_embedding", "count_nodes_by_depth", "search_tree",
    "build_prompt_with_search_memory_only_related",
}


def cos_sim_np(a, b):
    """sentence_transformers.util.cos_sim restated (absent here): cosine of two 1-D vectors → [1,1] tensor."""
    a = torch.as_tensor(np.asarray(a), dtype=torch.float32).reshape(1, -1)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float32).reshape(1, -1)
    return F.normalize(a, dim=1) @ F.normalize(b, dim=1).T


class _FakeConv:
    roles = ("user", "assistant")

    def __init__(self):
        self.msgs = []

    def copy(self):
        return _FakeConv()

    def append_message(self, r, m):
        self.msgs.append((r, m))

    def get_prompt(self):
        return " ".join(str(m) for _, m in self.msgs if m)


def load_reference_namespace():
    src = open(os.path.join(REF, "utiles.py"), encoding="utf-8").read()
    tree = ast.parse(src)
    ns = dict(torch=torch, random=random, math=math, np=np, F=F, defaultdict=defaultdict, time=time,
              conv_templates=defaultdict(_FakeConv), cos_sim=cos_sim_np,
              RED="", RESET="", BLUE="", GREEN="", YELLOW="")
    for n in tree.body:
        if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in WANT:
            exec(compile(ast.Module([n], []), "utiles.py", "exec"), ns)
    return ns


# ------------------------------------------------------------------------------------------
# G3  k-means.  The reference draws its init with torch.randperm(device) and its reseeds with
# random.randint; we record both so the build's explicit `init_idx` / `reseed_idx` inputs can
# replay them.  The per-iteration trace is produced by a traced re-statement run in lock-step
# and asserted equal to the reference function's return value.
# ------------------------------------------------------------------------------------------
def kmeans_trace(X, K, init_idx, reseed_stream, w=None, tol=1e-4, max_iter=10):
    """Lock-step re-statement of utiles.py:294-318 used ONLY to record the per-iteration labels;
    its final (centroids, labels) are asserted equal to the reference function's output."""
    T = X.shape[0]
    if w is None:
        w = torch.ones(T, dtype=X.dtype)
    C = X[init_idx].clone()
    labels_trace = []
    rs = list(reseed_stream)
    for i in range(max_iter):
        d = ((X.unsqueeze(1) - C.unsqueeze(0)) ** 2).sum(dim=2).sqrt()
        labels = torch.argmin(d, dim=1)
        labels_trace.append(labels.numpy().copy())
        ws = torch.zeros_like(C)
        wsum = torch.zeros(K, dtype=X.dtype)
        for j in range(K):
            m = labels == j
            ws[j] = torch.sum(w[m, None] * X[m], dim=0)
            wsum[j] = torch.sum(w[m])
        mask = wsum > 0
        newC = torch.zeros_like(ws)
        newC[mask] = ws[mask] / wsum[mask, None]
        if mask.sum() < K:
            newC[~mask] = torch.stack([X[rs.pop(0)] for _ in range(K - int(mask.sum()))])
        diff = torch.norm(C - newC, dim=1).sum()
        if diff < tol:
            break
        C = newC
    return C, labels, wsum, i, np.stack(labels_trace)


def make_clustered(T, PD, n_true, seed, spread=0.15):
    """Synthetic 'video' features: n_true scene centres + small per-frame jitter (well separated,
    like consecutive frames of a few scenes)."""
    g = torch.Generator().manual_seed(seed)
    centres = torch.randn(n_true, PD, generator=g)
    which = torch.randint(0, n_true, (T,), generator=g)
    return centres[which] + spread * torch.randn(T, PD, generator=g)


def gen_kmeans(ns):
    cases = []
    specs = [  # (T, P, D, K, seed, n_true)
        (40, 16, 32, 5, 11, 5), (64, 4, 16, 8, 12, 8), (400, 2, 8, 5, 13, 7),
        (64, 8, 64, 8, 14, 3), (33, 4, 16, 4, 15, 9), (100, 2, 512, 5, 16, 5),
        (17, 1, 8, 16, 17, 20),
    ]
    for (T, P, D, K, seed, n_true) in specs:
        X = make_clustered(T, P * D, n_true, seed).view(T, P, D).contiguous()
        torch.manual_seed(seed)
        init_idx = torch.randperm(T)[:K].clone()           # what utiles.py:295 will draw
        random.seed(seed)
        reseed_stream = [random.randint(0, T - 1) for _ in range(10 * K)]  # what :313 will draw
        torch.manual_seed(seed)
        random.seed(seed)
        red, labels = ns["weighted_kmeans_feature"](X.clone(), K)
        C, lab2, wsum, it, trace = kmeans_trace(X.view(T, -1), K, init_idx, reseed_stream)
        assert torch.equal(labels, lab2), "trace diverged from the reference (labels)"
        assert torch.equal(red.reshape(K, -1), C), "trace diverged from the reference (centroids)"
        cases.append(dict(X=X.numpy(), K=K, init_idx=init_idx.numpy().astype(np.int32),
                          reseed_idx=np.asarray(reseed_stream, np.int32), labels=labels.numpy(),
                          centroids=red.numpy(), wsum=wsum.numpy(), exit_iter=it, trace=trace, seed=seed))
    # weighted case (weights are an argument of the reference function)
    T, P, D, K, seed = 48, 4, 16, 4, 21
    X = make_clustered(T, P * D, 4, seed).view(T, P, D).contiguous()
    g = torch.Generator().manual_seed(seed)
    w = torch.rand(T, generator=g) + 0.25
    torch.manual_seed(seed); random.seed(seed)
    init_idx = torch.randperm(T)[:K].clone()
    reseed_stream = [random.randint(0, T - 1) for _ in range(10 * K)]
    torch.manual_seed(seed); random.seed(seed)
    red, labels = ns["weighted_kmeans_feature"](X.clone(), K, weights=w.clone())
    C, lab2, wsum, it, trace = kmeans_trace(X.view(T, -1), K, init_idx, reseed_stream, w=w)
    assert torch.equal(labels, lab2) and torch.equal(red.reshape(K, -1), C)
    cases.append(dict(X=X.numpy(), K=K, init_idx=init_idx.numpy().astype(np.int32),
                      reseed_idx=np.asarray(reseed_stream, np.int32), labels=labels.numpy(),
                      centroids=red.numpy(), wsum=wsum.numpy(), exit_iter=it, trace=trace, seed=seed,
                      weights=w.numpy()))
    # empty-cluster case: duplicate init rows => one cluster stays empty at iteration 0 (argmin
    # picks the first of two identical centroids), forcing the random.randint reseed (:312-313)
    T, P, D, K, seed = 24, 2, 8, 3, 31
    X = make_clustered(T, P * D, 3, seed).view(T, P, D).contiguous()
    X[5] = X[2]                                            # identical rows
    found = None
    for s in range(seed, seed + 4000):
        torch.manual_seed(s)
        idx = torch.randperm(T)[:K]
        if 2 in idx.tolist() and 5 in idx.tolist():
            found = s
            break
    assert found is not None
    seed = found
    torch.manual_seed(seed); random.seed(seed)
    init_idx = torch.randperm(T)[:K].clone()
    reseed_stream = [random.randint(0, T - 1) for _ in range(10 * K)]
    torch.manual_seed(seed); random.seed(seed)
    red, labels = ns["weighted_kmeans_feature"](X.clone(), K)
    C, lab2, wsum, it, trace = kmeans_trace(X.view(T, -1), K, init_idx, reseed_stream)
    assert torch.equal(labels, lab2) and torch.equal(red.reshape(K, -1), C)
    cases.append(dict(X=X.numpy(), K=K, init_idx=init_idx.numpy().astype(np.int32),
                      reseed_idx=np.asarray(reseed_stream, np.int32), labels=labels.numpy(),
                      centroids=red.numpy(), wsum=wsum.numpy(), exit_iter=it, trace=trace, seed=seed,
                      empty_cluster=1))
    for i, c in enumerate(cases):
        np.savez_compressed(os.path.join(OUT, f"kmeans_{i:02d}.npz"), **c)
    # T <= K : the reference returns a 3-tuple (utiles.py:321-322)
    X = torch.randn(4, 2, 8, generator=torch.Generator().manual_seed(5))
    out = ns["weighted_kmeans_feature"](X.clone(), 5)
    assert len(out) == 3
    np.savez_compressed(os.path.join(OUT, "kmeans_small_T.npz"), X=X.numpy(), K=5, out0=out[0].numpy(),
                        out1=out[1].numpy(), out2=np.asarray(out[2], np.int64))
    return len(cases)


def gen_forgetting(ns):
    d = {}
    for (L, tau) in [(20, 5), (7, 5), (20, 10), (1, 5)]:
        d[f"p_{L}_{tau}"] = ns["calculate_forgetting_probabilities"](L, tau=tau)
    # selection under the global numpy RNG (utiles.py:260)
    sel = []
    for seed in (0, 1, 2, 1234):
        np.random.seed(seed)
        p = ns["calculate_forgetting_probabilities"](20, tau=5)
        got = ns["select_data_without_replacement"](list(range(100, 120)), p, 5)
        sel.append([seed] + [int(x) for x in got])
    d["select_20_5_5"] = np.asarray(sel, np.int64)
    np.savez_compressed(os.path.join(OUT, "forgetting.npz"), **d)


class _FakeTok:
    def __call__(self, text, **kw):
        return types.SimpleNamespace(input_ids=[1, 2, 3])

    def batch_decode(self, ids, skip_special_tokens=True):
        return [f" caption#{int(ids[0][0])} "]


class _FakeSummarizer:
    device = "cpu"

    def __init__(self):
        self.n = 0

    def generate_with_image_embedding(self, ids, image_embeddings=None, **kw):
        self.n += 1
        return torch.tensor([[self.n]])


def _describe(nodes):
    def one(n):
        return dict(depth=n.depth, shape=list(n.centroids.shape), text=n.text,
                    children=[one(c) for c in n.children])
    return [one(n) for n in nodes]


def gen_tree(ns):
    """Drive the reference tree builder exactly as updating_memory_buffer does
    (inference_streaming_longva_v2.py:346-358): chunk the new frames, k-means branch unreachable."""
    out = []
    for (chunk, K, interval, n_updates, frames_per_update, P, D) in [
            (4, 2, 3, 8, 7, 2, 8), (40, 5, 10, 6, 130, 1, 4), (3, 2, 2, 7, 6, 2, 4)]:
        torch.manual_seed(100 + chunk); random.seed(100 + chunk)
        summ = _FakeSummarizer(); tok = _FakeTok()
        tree = None
        trace = []
        gframe = 0
        for u in range(n_updates):
            buf = []
            for _ in range(frames_per_update):
                buf.append(torch.full((1, P, D), float(gframe)) + 0.01 * torch.randn(1, P, D)); gframe += 1
            chunked = [buf[i:i + chunk] for i in range(0, len(buf), chunk)]
            km = [torch.cat(c) for c in chunked]
            tree = ns["fast_building_memory_tree_summarize_token"](
                km, K, interval, summ, torch.zeros(1, 3, dtype=torch.long), tok, chunked, tree)
            cnt = ns["count_nodes_by_depth"](tree)
            trace.append(dict(top=_describe(tree), count={int(k): int(v) for k, v in cnt.items()}))
        out.append(dict(chunk=chunk, K=K, interval=interval, frames_per_update=frames_per_update,
                        P=P, D=D, trace=trace))
    json.dump(out, open(os.path.join(OUT, "tree_trace.json"), "w"))


class _FakeEmbTok:
    def __call__(self, text, padding=True, return_tensors="pt"):
        return {"text": _TextTensor(text)}


class _TextTensor:
    def __init__(self, t):
        self.t = t

    def cuda(self):
        return self


class _FakeEmbModel:
    """Deterministic fake of the BERT-large encoder: text -> table lookup -> [1, L=2, d] hidden states."""
    def __init__(self, table):
        self.table = table

    def __call__(self, text):
        v = torch.tensor(self.table[text.t], dtype=torch.float32)
        return types.SimpleNamespace(last_hidden_state=torch.stack([v, -v])[None])


def gen_search(ns):
    N = ns["MultimodalTreeNode"]
    rng = np.random.default_rng(7)
    d = 16
    texts = {}
    cases = []

    def emb(name, v=None):
        texts[name] = (rng.standard_normal(d) if v is None else v).astype(np.float32).tolist()
        return name

    def leaf(name, val, v=None):
        return N(torch.full((3, 2, 4), float(val)), emb(name, v), depth=0)

    def build():
        q = rng.standard_normal(d).astype(np.float32)
        l = [leaf(f"leaf{i}", i) for i in range(9)]
        a = N(torch.full((2, 2, 4), 100.0), emb("A"), depth=1); a.children = l[0:3]
        b = N(torch.full((2, 2, 4), 101.0), emb("B"), depth=1); b.children = l[3:6]
        root = N(torch.full((2, 2, 4), 200.0), emb("R"), depth=2); root.children = [a, b]
        c = N(torch.full((2, 2, 4), 102.0), emb("C"), depth=1); c.children = l[6:9]
        r1 = leaf("red1", 50); r2 = leaf("red2", 51, v=q * 0.5 + 0.1 * rng.standard_normal(d).astype(np.float32))
        texts["QUERY"] = q.tolist()
        return [root, c, r1, r2]

    for trial in range(3):
        texts.clear()
        nodes = build()
        feats, txt = ns["fast_search_tree_multi_modal_with_embedding"](
            nodes, "QUERY", None, _FakeEmbModel(texts), _FakeEmbTok())
        cases.append(dict(texts=dict(texts), path_text=txt,
                          path_first_value=[float(f.flatten()[0]) for f in feats],
                          path_shapes=[list(f.shape) for f in feats]))
    # only-redundant case with all similarities <= 0: strict '>' keeps best_index = 0 (utiles.py:751-777)
    texts.clear()
    q = np.ones(d, np.float32)
    texts["QUERY"] = q.tolist()
    reds = [leaf(f"neg{i}", i, v=-q * (i + 1)) for i in range(3)]
    feats, txt = ns["fast_search_tree_multi_modal_with_embedding"](reds, "QUERY", None, _FakeEmbModel(texts), _FakeEmbTok())
    cases.append(dict(texts=dict(texts), path_text=txt, path_first_value=[float(f.flatten()[0]) for f in feats],
                      path_shapes=[list(f.shape) for f in feats], all_negative=True))
    json.dump(cases, open(os.path.join(OUT, "search.json"), "w"))


# ------------------------------------------------------------------------------------------
# G9b  dialogue memory: JsonMemoryLoader.load, the patched FAISS search with neighbour expansion
# and search_memory (memory_bank/memory_retrieval/local_doc_qa.py:17-61,120-178,263-288), run on
# stand-ins for the absent langchain / faiss objects (exact flat-L2 index, dict docstore).
# ------------------------------------------------------------------------------------------
class _Doc:
    def __init__(self, page_content, metadata):
        self.page_content, self.metadata = page_content, metadata


class _Loader:     # stand-in for langchain's UnstructuredFileLoader base
    def __init__(self, filepath, mode="elements"):
        self.file_path = filepath


def _text_vec(text, d=24):
    import zlib
    rng = np.random.default_rng(zlib.crc32(text.encode()))
    v = rng.standard_normal(d).astype(np.float32)
    return v / np.linalg.norm(v)


class _FlatL2:
    def __init__(self, X):
        self.X = X

    def search(self, q, k):
        d = ((self.X - q[0][None]) ** 2).sum(1)
        order = np.argsort(d, kind="stable")[:k]
        idx = np.full((1, k), -1, np.int64); sc = np.full((1, k), np.inf, np.float32)
        idx[0, : len(order)] = order; sc[0, : len(order)] = d[order]
        return sc, idx


class _Docstore:
    def __init__(self, docs):
        self.d = {str(i): doc for i, doc in enumerate(docs)}

    def search(self, _id):
        return self.d[_id]


def gen_memory():
    import tempfile
    src = open(os.path.join(REF, "memory_bank/memory_retrieval/local_doc_qa.py"), encoding="utf-8").read()
    tree = ast.parse(src)
    ns = dict(np=np, json=json, os=os, List=list, Tuple=tuple, Optional=None, Document=_Doc, UnstructuredFileLoader=_Loader,
              VECTOR_SEARCH_TOP_K=3, CHUNK_SIZE=200, EMBEDDING_MODEL_CN="", EMBEDDING_DEVICE="cpu", TextSplitter=None)
    import typing
    ns.update(List=typing.List, Tuple=typing.Tuple, Optional=typing.Optional)
    for n in tree.body:
        if isinstance(n, ast.FunctionDef) and n.name in ("seperate_list", "similarity_search_with_score_by_vector", "get_docs_with_score"):
            exec(compile(ast.Module([n], []), "local_doc_qa.py", "exec"), ns)
        if isinstance(n, ast.ClassDef) and n.name in ("JsonMemoryLoader", "LocalMemoryRetrieval"):
            exec(compile(ast.Module([n], []), "local_doc_qa.py", "exec"), ns)
    memory = {"User": {"name": "User", "history": {
        "2024-05-01": [{"query": "where did I leave the red cup", "response": "on the kitchen counter next to the sink"},
                       {"query": "what colour is the car outside", "response": "a blue hatchback"},
                       {"query": "who was at the door", "response": "a courier with a small parcel"}],
        "2024-05-02": [{"query": "did I lock the bicycle", "response": "yes, to the rack near the stairs"},
                       {"query": "what is on the laptop screen " + "very " * 30 + "long", "response": "a spreadsheet " + "with rows " * 12},
                       {"query": "what did the sign say", "response": "road closed ahead"}]},
        "summary": {"2024-05-01": "the user asked about a cup, a car and a courier"}},
        "Other": {"name": "Other", "history": {"2024-05-01": [{"query": "x", "response": "y"}]}}}
    tmp = tempfile.mkdtemp()
    fp = os.path.join(tmp, "memory_0.json")
    json.dump(memory, open(fp, "w"))
    docs = ns["JsonMemoryLoader"](fp, "en").load("User")
    loaded = [dict(page_content=d.page_content, source=d.metadata["source"]) for d in docs]
    cases = []
    for top_k in (1, 2, 3):
        for query in ["where is my red cup", "bicycle lock", "what did the road sign say", "laptop screen spreadsheet", "courier parcel door"]:
            fresh = ns["JsonMemoryLoader"](fp, "en").load("User")      # search mutates page_content
            X = np.stack([_text_vec(d.page_content) for d in fresh])
            vs = types.SimpleNamespace(index=_FlatL2(X), index_to_docstore_id={i: str(i) for i in range(len(fresh))},
                                       docstore=_Docstore(fresh), chunk_size=200)
            vs.similarity_search_with_score = lambda q, k, vs=vs: ns["similarity_search_with_score_by_vector"](vs, _text_vec(q).tolist(), k)
            lm = ns["LocalMemoryRetrieval"]()
            lm.top_k = top_k
            date_docs, dates = lm.search_memory(query, vs)
            cases.append(dict(query=query, top_k=top_k, date_docs=date_docs, dates=dates))
    json.dump(dict(memory=memory, loaded=loaded, cases=cases, embed="crc32-seeded unit vectors d=24 (tools/make_golden.py:_text_vec)"),
              open(os.path.join(OUT, "memory_search.json"), "w"))


# ------------------------------------------------------------------------------------------
# G7  embedding splice: LlavaMetaForCausalLM.prepare_inputs_embeddings_for_multimodal
# (longva/model/llava_arch.py:208-343), imported with package stubs.
# ------------------------------------------------------------------------------------------
def gen_splice():
    for name in ["longva", "longva.model", "longva.model.language_model", "longva.model.multimodal_resampler"]:
        m = types.ModuleType(name); m.__path__ = [REF + "/" + name.replace(".", "/")]; sys.modules[name] = m
    q = types.ModuleType("longva.model.multimodal_resampler.qformer"); q.Qformer = type("Qformer", (), {}); sys.modules[q.__name__] = q
    from longva.model.llava_arch import LlavaMetaForCausalLM
    torch.manual_seed(3)
    table = torch.nn.Embedding(50, 16)
    table.weight.data = torch.randn(50, 16)

    class Shim(LlavaMetaForCausalLM):
        def __init__(self, max_len):
            self.config = types.SimpleNamespace(tokenizer_model_max_length=max_len, tune_mm_mlp_adapter=False, mm_use_im_start_end=False,
                                                tokenizer_padding_side="right", use_pos_skipping=False)
            self.device = torch.device("cpu"); self.training = False

        def get_model(self):
            return types.SimpleNamespace(embed_tokens=table)
    cases = {}
    feats = torch.randn(7, 16)
    specs = {"middle": ([3, 9, -200, 4, 4, 12], None), "start": ([-200, 5, 6], None), "none": ([1, 2, 3, 4], None),
             "truncated": ([3, 9, -200, 4, 4, 12], 6), "end": ([8, 8, -200], None)}
    for k, (ids, mx) in specs.items():
        out = Shim(mx).prepare_inputs_embeddings_for_multimodal(torch.tensor([ids]), None, None, None, None, [feats], ["video"])
        assert out[0] is None and out[1] is None and out[2] is None and out[5] is None
        cases[k + ".ids"] = np.asarray(ids, np.int64)
        cases[k + ".max_len"] = np.asarray(-1 if mx is None else mx)
        cases[k + ".embeds"] = out[4][0].detach().numpy()
    np.savez_compressed(os.path.join(OUT, "splice.npz"), table=table.weight.detach().numpy(), feats=feats.numpy(), **cases)


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present: golden vectors can only be generated in the authoring container")
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)   # fix the fp32 summation order of the generator run
    torch.Tensor.cuda = lambda s, *a, **k: s
    ns = load_reference_namespace()
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        n = gen_kmeans(ns)
        gen_forgetting(ns)
        gen_tree(ns)
        gen_search(ns)
        gen_memory()
        gen_splice()
    print(f"wrote {n} k-means cases + forgetting/tree/search fixtures to {os.path.normpath(OUT)}")
    meta = dict(torch=torch.__version__, numpy=np.__version__, reference="hmxiong/StreamChat @ 2025-03-14",
                functions=sorted(WANT))
    json.dump(meta, open(os.path.join(OUT, "META.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
