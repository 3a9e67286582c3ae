"""GPU: dialogue memory through sc_sim_topk + the HIP MiniLM embedder (same golden cases as the CPU test)."""
import pytest
import torch

from streamchat_amd import text as T
from streamchat_amd.memory_bank.memory_retrieval import local_doc_qa as Q
from tests.test_memory_bank import run_cases

pytestmark = pytest.mark.gpu


def test_memory_search_matches_reference_on_device(tmp_path):
    run_cases(tmp_path, None, device="cuda")            # topk_fn=None -> the product path (sc_sim_topk)


def test_memory_with_hip_sentence_embedder(tmp_path):
    import json
    cfg = T.BertConfigLite(**T.MINILM_L6)
    emb = Q.HipSentenceEmbeddings(T.SentenceEmbedder(T.BertEncoder(T.random_bert_state_dict(cfg, seed=2, std=0.05), cfg)), T.HashTokenizer())
    mem = {"User": {"name": "User", "history": {"2024-01-01": [{"query": f"question number {i} about the {w}", "response": f"answer {i}"}
                                                               for i, w in enumerate(["cup", "car", "door", "river", "phone", "lamp"])]}}}
    fp = tmp_path / "m.json"
    json.dump(mem, open(fp, "w"))
    lm = Q.LocalMemoryRetrieval()
    lm.init_cfg("minilm-l6", top_k=1, language="en", embedder=emb)
    vs, _ = lm.init_memory_vector_store(str(fp), str(tmp_path / "idx"), user_name="User")
    store = lm.load_memory_index(vs)
    assert store.vectors.is_cuda and store.vectors.shape == (6, 384)
    # querying with a stored document's own text must retrieve that document (distance 0 is the unique minimum)
    docs, dates = lm.search_memory(store.docs[3].page_content, store)
    assert "question number 3" in docs[0] and dates == "2024-01-01"
