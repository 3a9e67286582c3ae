"""Dialogue-memory retrieval (mirror of reference memory_bank/memory_retrieval/local_doc_qa.py).

Same classes / methods / return values as upstream: `JsonMemoryLoader.load` (:25-61), the patched
`similarity_search_with_score_by_vector` with same-date neighbour expansion (:135-178, quirks Q14),
`LocalMemoryRetrieval.{init_cfg, init_memory_vector_store, load_memory_index, search_memory}` (:185-288, Q15).
What changed underneath: documents are embedded in one batch by the HIP MiniLM encoder, the index is an exact flat-L2
table in HBM searched by `sc_sim_topk` (what langchain's FAISS.from_documents builds is an IndexFlatL2), and
embeddings are cached by text so the per-round `rmtree` + rebuild of upstream (memory_utils.py:76-83) costs only the
new documents."""
import json
import os
from typing import List, Tuple

import numpy as np

from .configs.model_config import CHUNK_SIZE, EMBEDDING_DEVICE, EMBEDDING_MODEL_EN, VECTOR_SEARCH_TOP_K, embedding_model_dict


class Document:
    def __init__(self, page_content: str, metadata: dict):
        self.page_content, self.metadata = page_content, metadata


class JsonMemoryLoader:
    def __init__(self, filepath, language, mode="elements"):
        self.filepath = self.file_path = filepath
        self.language = language

    def _get_metadata(self, date: str) -> dict:
        return {"source": date}

    def load(self, name):
        """One Document per stored Q/A turn of user `name` (+ one per dated summary), in file order (:25-61)."""
        user_memories = []
        with open(self.filepath, "r", encoding="utf-8") as f:
            memories = json.loads(f.read())
        for user_name, user_memory in memories.items():
            if user_name != name:
                continue
            user_memories = []
            if "history" not in user_memory.keys():
                continue
            for date, content in user_memory["history"].items():
                metadata = self._get_metadata(date)
                memory_str = f"时间{date}的对话内容：" if self.language == "cn" else f"Conversation content on {date}:"
                user_kw = "[|用户|]：" if self.language == "cn" else "[|User|]:"
                ai_kw = "[|AI恋人|]：" if self.language == "cn" else "[|AI|]:"
                for dialog in content:
                    query, response = dialog["query"], dialog["response"]
                    tmp_str = memory_str + f"{user_kw} {query.strip()}; " + f"{ai_kw} {response.strip()}"
                    user_memories.append(Document(page_content=tmp_str, metadata=metadata))
                if "summary" in user_memory.keys() and date in user_memory["summary"].keys():
                    summary = (f'时间{date}的对话总结为：{user_memory["summary"][date]}' if self.language == "cn"
                               else f'The summary of the conversation on {date} is: {user_memory["summary"][date]}')
                    user_memories.append(Document(page_content=summary, metadata=metadata))
        return user_memories


def load_memory_file(filepath, user_name, language="cn"):
    return JsonMemoryLoader(filepath, language).load(user_name)


def get_docs_with_score(docs_with_score):
    docs = []
    for doc, score in docs_with_score:
        doc.metadata["score"] = score
        docs.append(doc)
    return docs


def seperate_list(ls: List[int]) -> List[List[int]]:
    """runs of consecutive ids (:120-131)"""
    lists, run = [], [ls[0]]
    for i in range(1, len(ls)):
        if ls[i - 1] + 1 == ls[i]:
            run.append(ls[i])
        else:
            lists.append(run)
            run = [ls[i]]
    lists.append(run)
    return lists


def _device_topk(q, X, k):
    """exact flat-L2 top-k on the GPU (sc_sim_topk); returns (scores [k] squared L2, indices [k])"""
    from ... import ops
    idx, sc = ops.sim_topk(q, X, k, "l2")
    return sc.cpu().numpy(), idx.cpu().numpy()


class FlatL2VectorStore:
    """What `FAISS.from_documents(docs, embeddings)` gives the reference: an exact L2 index + docstore."""

    def __init__(self, docs: List[Document], vectors, embed_query, chunk_size=CHUNK_SIZE, topk_fn=None):
        self.docs, self.vectors, self.embed_query = docs, vectors, embed_query
        self.chunk_size = chunk_size
        self.topk_fn = topk_fn or _device_topk

    def __len__(self):
        return len(self.docs)

    def similarity_search_with_score(self, query, k=4):
        return self.similarity_search_with_score_by_vector(self.embed_query(query), k)

    def similarity_search_with_score_by_vector(self, embedding, k: int = 4) -> List[Tuple[Document, float]]:
        """Reference :135-178 incl. its quirks (Q14): the inner loop variable shadows k; the expansion bound is
        `range(1, max(i, len(docs) - i))` with `docs` the OUTPUT list; expansion stops at the first neighbour that would
        exceed chunk_size characters; only same-date neighbours merge; every merged group reports the score of the
        LAST hit."""
        n = len(self.docs)
        kk = min(k, n)
        scores, indices = self.topk_fn(embedding, self.vectors, kk)
        indices = list(indices) + [-1] * (k - kk)
        scores = list(scores) + [np.inf] * (k - kk)
        copies = {}

        def fetch(i):                         # docstore.search returns the stored object; merges mutate a per-call copy
            if i not in copies:
                d = self.docs[i]
                copies[i] = Document(d.page_content, dict(d.metadata))
            return copies[i]
        docs = []
        id_set = set()
        j = 0
        for j, i in enumerate(indices):
            if i == -1:
                continue
            i = int(i)
            doc = fetch(i)
            id_set.add(i)
            docs_len = len(doc.page_content)
            for kx in range(1, max(i, len(docs) - i)):
                for l in [i + kx, i - kx]:
                    if 0 <= l < n:
                        doc0 = fetch(l)
                        if docs_len + len(doc0.page_content) > self.chunk_size:
                            break
                        elif doc0.metadata["source"] == doc.metadata["source"]:
                            docs_len += len(doc0.page_content)
                            id_set.add(l)
        if not id_set:
            return docs
        for id_seq in seperate_list(sorted(id_set)):
            doc = None
            for id in id_seq:
                if id == id_seq[0]:
                    doc = fetch(id)
                else:
                    doc.page_content += fetch(id).page_content
            docs.append((doc, scores[j]))
        return docs


class LocalMemoryRetrieval:
    embeddings: object = None
    top_k: int = VECTOR_SEARCH_TOP_K
    chunk_size: int = CHUNK_SIZE

    def init_cfg(self, embedding_model: str = EMBEDDING_MODEL_EN, embedding_device=EMBEDDING_DEVICE, top_k=VECTOR_SEARCH_TOP_K,
                 language="cn", embedder=None):
        """`embedder`: object with embed_documents(list[str]) -> [M, d] and embed_query(str) -> [d] (CUDA fp32).  Without one the
        named sentence-embedding checkpoint must exist locally (reference model_config.py:12) — there is no CPU fallback."""
        self.language = language
        self.top_k = top_k
        if embedder is None:
            path = embedding_model_dict[embedding_model]
            raise FileNotFoundError(f"sentence-embedding checkpoint {path} not available; pass embedder=HipSentenceEmbeddings(...)")
        self.embeddings = embedder
        self._cache = {}

    def _embed_docs(self, docs):
        import torch
        missing = [d.page_content for d in docs if d.page_content not in self._cache]
        if missing:
            vecs = self.embeddings.embed_documents(missing)
            for t, v in zip(missing, vecs):
                self._cache[t] = v
        return torch.stack([self._cache[d.page_content] for d in docs])

    def init_memory_vector_store(self, filepath, vs_path=None, user_name: str = None, cur_date: str = None):
        """Build (or rebuild) the index of `user_name`'s dialogue documents and persist it under vs_path (:196-255)."""
        paths = [filepath] if isinstance(filepath, str) else list(filepath)
        docs, loaded_files = [], []
        for fp in paths:
            if not os.path.exists(fp):
                continue
            files = [os.path.join(fp, f) for f in os.listdir(fp)] if os.path.isdir(fp) else [fp]
            for f in files:
                try:
                    docs += load_memory_file(f, user_name, self.language)
                    loaded_files.append(f)
                except Exception as e:      # mirror of the upstream per-file try/except
                    print(e)
        if isinstance(filepath, str) and not os.path.exists(filepath):
            return None, None
        if len(docs) == 0:
            return None, loaded_files
        vecs = self._embed_docs(docs)
        if vs_path:
            self._persist(vs_path, docs, vecs)
        self._last = (docs, vecs, vs_path)
        return vs_path, loaded_files

    # ---- on-disk index: append-only (round 6).  The reference re-opens the saved FAISS index and ADDS to it (`FAISS.load_local ...
    # add_documents ... save_local`, local_doc_qa.py:242-251); rounds 1-5 here re-serialised every document and copied every vector to the
    # host after every question.  Now `index.jsonl` (one document per line) and `index.npy` (a .npy whose header is padded to a fixed
    # 128 bytes, so the row count can be rewritten in place) only receive the rows that are new since the last write: one D2H copy of the new
    # vectors.  A document list that is not an extension of what is on disk (another user, an edited history) rewrites both files. ----
    _NPY_HEADER = 128

    @staticmethod
    def _doc_line(d):
        return json.dumps(dict(page_content=d.page_content, metadata=d.metadata), ensure_ascii=False)

    def _write_npy_header(self, f, rows, dim):
        head = ("{'descr': '<f4', 'fortran_order': False, 'shape': (%d, %d), }" % (rows, dim)).encode("latin1")
        pad = self._NPY_HEADER - 10 - len(head) - 1
        if pad < 0:
            raise ValueError("index.npy: header does not fit its fixed size")
        f.seek(0)
        f.write(b"\x93NUMPY\x01\x00" + int(self._NPY_HEADER - 10).to_bytes(2, "little") + head + b" " * pad + b"\n")

    def _persist(self, vs_path, docs, vecs):
        os.makedirs(vs_path, exist_ok=True)
        jl, npy = os.path.join(vs_path, "index.jsonl"), os.path.join(vs_path, "index.npy")
        lines = [self._doc_line(d) for d in docs]
        done = getattr(self, "_persisted", {}).get(vs_path)
        if done is None and os.path.exists(jl) and os.path.exists(npy):               # a previous process wrote this index
            done = open(jl, encoding="utf-8").read().splitlines()
        dim = int(vecs.shape[1])
        n_old = len(done) if done is not None else 0
        extend = done is not None and n_old <= len(lines) and lines[:n_old] == done and os.path.getsize(npy) == self._NPY_HEADER + n_old * dim * 4
        if not extend:
            n_old = 0
        new_rows = vecs[n_old:].detach().to("cpu", dtype=__import__("torch").float32).numpy() if len(lines) > n_old else None     # ONLY the new vectors leave the device
        with open(jl, "a" if extend else "w", encoding="utf-8") as f:
            for ln in lines[n_old:]:
                f.write(ln + "\n")
        with open(npy, "r+b" if extend else "w+b") as f:
            self._write_npy_header(f, len(lines), dim)
            if new_rows is not None:
                f.seek(self._NPY_HEADER + n_old * dim * 4)
                f.write(np.ascontiguousarray(new_rows, np.float32).tobytes())
            f.truncate(self._NPY_HEADER + len(lines) * dim * 4)
        if os.path.exists(os.path.join(vs_path, "index.json")):                       # the whole-file format of rounds 1-5
            os.remove(os.path.join(vs_path, "index.json"))
        self._persisted = dict(getattr(self, "_persisted", {}), **{vs_path: lines})
        self.persist_stats = dict(rows_total=len(lines), rows_written=len(lines) - n_old, appended=bool(extend))

    def load_memory_index(self, vs_path, device=EMBEDDING_DEVICE):
        import torch
        last = getattr(self, "_last", None)
        if last is not None and last[2] == vs_path:
            docs, vecs = last[0], last[1]
        else:
            if os.path.exists(os.path.join(vs_path, "index.jsonl")):
                recs = [json.loads(ln) for ln in open(os.path.join(vs_path, "index.jsonl"), encoding="utf-8").read().splitlines() if ln]
            else:
                recs = json.load(open(os.path.join(vs_path, "index.json"), encoding="utf-8"))["docs"]
            docs = [Document(d["page_content"], d["metadata"]) for d in recs]
            vecs = torch.from_numpy(np.load(os.path.join(vs_path, "index.npy"))).to(device)
        return FlatL2VectorStore(docs, vecs, self.embeddings.embed_query, chunk_size=self.chunk_size,
                                 topk_fn=getattr(self, "topk_fn", None))

    def search_memory(self, query, vector_store):
        """Reference :263-288: top-k hits -> neighbour-merged docs -> sorted by date string -> same-date docs joined with a
        newline; only the Chinese prefix is stripped (Q15).  Returns (list[str], "d1, d2")."""
        related_docs = get_docs_with_score(vector_store.similarity_search_with_score(query, k=self.top_k))
        related_docs = sorted(related_docs, key=lambda x: x.metadata["source"], reverse=False)
        pre_date, date_docs, dates = "", [], []
        for doc in related_docs:
            doc.page_content = doc.page_content.replace(f'时间{doc.metadata["source"]}的对话内容：', "").strip()
            if doc.metadata["source"] != pre_date:
                date_docs.append(doc.page_content)
                pre_date = doc.metadata["source"]
                dates.append(pre_date)
            else:
                date_docs[-1] += f"\n{doc.page_content}"
        return date_docs, ", ".join(dates)


class HipSentenceEmbeddings:
    """`HuggingFaceEmbeddings` stand-in (reference local_doc_qa.py:193): tokenizer + HIP MiniLM SentenceEmbedder."""

    def __init__(self, sentence_embedder, tokenizer):
        self.se, self.tok = sentence_embedder, tokenizer

    def embed_documents(self, texts):
        b = self.tok(list(texts))
        return self.se.embed(b["input_ids"], b["attention_mask"].sum(1))

    def embed_query(self, text):
        return self.embed_documents([text])[0]
