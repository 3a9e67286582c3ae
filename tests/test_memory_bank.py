"""CPU: dialogue-memory mirror (streamchat_amd/memory_bank) against golden outputs of the reference's own
JsonMemoryLoader / patched FAISS search / search_memory (tools/make_golden.py gen_memory).  The exact flat-L2
top-k is provided by the oracle here (no GPU); tests/test_gpu_memory.py runs the same cases through sc_sim_topk."""
import json
import os
import types
import zlib

import numpy as np
import torch

import oracle
from streamchat_amd.memory_bank.memory_retrieval import local_doc_qa as Q
from streamchat_amd.memory_bank import memory_utils as MU

G = os.path.join(os.path.dirname(__file__), "golden")


def text_vec(text, d=24):
    rng = np.random.default_rng(zlib.crc32(text.encode()))
    v = rng.standard_normal(d).astype(np.float32)
    return v / np.linalg.norm(v)


class FakeEmb:
    def __init__(self, device="cpu"):
        self.device = device

    def embed_documents(self, texts):
        return torch.from_numpy(np.stack([text_vec(t) for t in texts])).to(self.device)

    def embed_query(self, t):
        return torch.from_numpy(text_vec(t)).to(self.device)


def oracle_topk(q, X, k):
    idx, sc = oracle.topk(np.asarray(q), np.asarray(X), k, "l2")
    return sc, idx


def run_cases(tmp_path, topk_fn, device="cpu"):
    g = json.load(open(os.path.join(G, "memory_search.json")))
    fp = tmp_path / "memory_0.json"
    json.dump(g["memory"], open(fp, "w"))
    docs = Q.JsonMemoryLoader(str(fp), "en").load("User")
    assert [dict(page_content=d.page_content, source=d.metadata["source"]) for d in docs] == g["loaded"]
    for c in g["cases"]:
        lm = Q.LocalMemoryRetrieval()
        lm.init_cfg("minilm-l6", top_k=c["top_k"], language="en", embedder=FakeEmb(device))
        if topk_fn is not None:
            lm.topk_fn = topk_fn
        vs_path, loaded = lm.init_memory_vector_store(str(fp), str(tmp_path / "idx"), user_name="User")
        store = lm.load_memory_index(vs_path)
        date_docs, dates = lm.search_memory(c["query"], store)
        assert (date_docs, dates) == (c["date_docs"], c["dates"]), c


def test_memory_search_matches_reference(tmp_path):
    run_cases(tmp_path, oracle_topk)


def test_enter_name_and_save_local_memory(tmp_path):
    args = types.SimpleNamespace(memory_basic_dir=str(tmp_path), memory_file="memory_0.json", language="en")
    lm = Q.LocalMemoryRetrieval()
    lm.init_cfg("minilm-l6", top_k=1, language="en", embedder=FakeEmb())
    lm.topk_fn = oracle_topk
    memory = {}
    msg, um, memory, name, idx = MU.enter_name("User", memory, lm, args)
    assert idx is None and memory == {"User": {"name": "User"}}            # new user: no index (Q16)
    memory = MU.save_local_memory(memory, [["where is the cup", "on the table"]], "User", args)
    msg, um, memory, name, idx = MU.enter_name("User", memory, lm, args)
    assert idx is not None and len(idx) == 1
    docs, dates = lm.search_memory("cup", idx)
    assert docs[0].startswith("Conversation content on ") and "[|User|]: where is the cup; [|AI|]: on the table" in docs[0]
    # the index is persisted; a fresh retriever reloads it from disk and answers identically
    lm2 = Q.LocalMemoryRetrieval(); lm2.init_cfg("minilm-l6", top_k=1, language="en", embedder=FakeEmb()); lm2.topk_fn = oracle_topk
    store = lm2.load_memory_index(os.path.join(str(tmp_path), "memory_index/User_index"), device="cpu")
    assert lm2.search_memory("cup", store) == (docs, dates)


def test_build_prompt_with_search_memory_only_related():
    from streamchat_amd import utiles as U
    from streamchat_amd.memory_bank.prompt_utils import only_related_prompt_dict_ego
    fake = types.SimpleNamespace(search_memory=lambda q, idx: (["A", "B"], "d1, d2"))
    meta = only_related_prompt_dict_ego()["en"]
    p = U.build_prompt_with_search_memory_only_related("q?", "User", object(), fake, meta, "[|User|]", "[|AI|]", "AI")
    assert '"\nA\nB\n"' in p
    assert U.build_prompt_with_search_memory_only_related("q?", "User", None, fake, meta, "[|User|]", "[|AI|]", "AI") is None


def test_dialogue_index_is_persisted_append_only(tmp_path):
    """Round 6 (reference local_doc_qa.py:242-251 re-opens the saved index and adds to it): after every question only the
    NEW documents and the NEW vectors are written - index.jsonl grows by lines, index.npy by rows behind a fixed-size header - and a fresh
    process reads the same index back; a history that is not an extension of what is on disk rewrites the files."""
    args = types.SimpleNamespace(memory_basic_dir=str(tmp_path), memory_file="memory_0.json", language="en")
    lm = Q.LocalMemoryRetrieval()
    lm.init_cfg("minilm-l6", top_k=1, language="en", embedder=FakeEmb())
    lm.topk_fn = oracle_topk
    memory = {}
    _, _, memory, _, _ = MU.enter_name("User", memory, lm, args)
    vs = os.path.join(str(tmp_path), "memory_index/User_index")
    sizes, n_docs = [], []
    for r, (q, a) in enumerate([("where is the cup", "on the table"), ("who came in", "a man in a red coat"), ("what did he take", "the umbrella")]):
        memory = MU.save_local_memory(memory, [[q, a]], "User", args)
        _, _, memory, _, idx = MU.enter_name("User", memory, lm, args)
        st = lm.persist_stats
        n_docs.append(st["rows_total"])
        assert st["appended"] == (r > 0) and st["rows_written"] == n_docs[-1] - (n_docs[-2] if r else 0), st
        sizes.append(os.path.getsize(os.path.join(vs, "index.npy")))
        arr = np.load(os.path.join(vs, "index.npy"))
        assert arr.shape[0] == st["rows_total"] and sizes[-1] == 128 + arr.size * 4
        lines = open(os.path.join(vs, "index.jsonl"), encoding="utf-8").read().splitlines()
        assert len(lines) == st["rows_total"]
        assert np.array_equal(arr, np.stack([text_vec(json.loads(ln)["page_content"]) for ln in lines]))
    assert n_docs == sorted(n_docs) and n_docs[-1] > n_docs[0]
    # a fresh process: reads the index, and appends to what it finds on disk
    lm2 = Q.LocalMemoryRetrieval(); lm2.init_cfg("minilm-l6", top_k=1, language="en", embedder=FakeEmb()); lm2.topk_fn = oracle_topk
    store = lm2.load_memory_index(vs, device="cpu")
    assert lm2.search_memory("umbrella", store)[0][0].count("the umbrella") == 1
    memory = MU.save_local_memory(memory, [["is it raining", "yes"]], "User", args)
    MU.enter_name("User", memory, lm2, args)
    assert lm2.persist_stats["appended"] and lm2.persist_stats["rows_written"] == lm2.persist_stats["rows_total"] - n_docs[-1]
    # an edited history is not an extension: both files are rewritten
    fp = os.path.join(str(tmp_path), "memory_0.json")
    mem = json.load(open(fp))
    first_date = sorted(mem["User"]["history"])[0]
    mem["User"]["history"][first_date][0]["response"] = "under the chair"
    json.dump(mem, open(fp, "w"))
    MU.enter_name("User", mem, lm2, args)
    assert not lm2.persist_stats["appended"] and lm2.persist_stats["rows_written"] == lm2.persist_stats["rows_total"]
