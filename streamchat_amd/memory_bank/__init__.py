"""Dialogue-history memory (mirror of the reference's `memory_bank/` MemoryBank-SiliconFriend retriever, the
parts the streaming path touches: SURVEY.md §2.1 row 4).  langchain / FAISS / sentence-transformers are replaced
by an HBM-resident flat-L2 table searched with the `sc_sim_topk` kernel and the HIP MiniLM sentence embedder."""
