"""Retrieval constants (reference memory_bank/memory_retrieval/configs/model_config.py:5-17,51)."""
embedding_model_dict = {"minilm-l6": "/All_Model_Zoo/all-MiniLM-L6-v2", "minilm-l12": "all-MiniLM-L12-v2"}
EMBEDDING_MODEL_EN = "minilm-l6"
EMBEDDING_DEVICE = "cuda"
VECTOR_SEARCH_TOP_K = 3
CHUNK_SIZE = 200          # max characters of a merged neighbour group (local_doc_qa.py:158)
