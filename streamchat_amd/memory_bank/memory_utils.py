"""Dialogue-memory bookkeeping (mirror of reference memory_bank/memory_utils.py:61-110; gradio State handling dropped —
the streaming entry point passes plain objects)."""
import datetime
import json
import os
import shutil
import time


def enter_name(name, memory, local_memory_qa, data_args, update_memory_index=True):
    """reference :61-90.  Returns (msg, user_memory, memory, name, user_memory_index); a new user gets
    user_memory_index=None, so the first question of a video has no history prompt (Q16)."""
    cur_date = datetime.date.today().strftime("%Y-%m-%d")
    user_memory_index = None
    memory_dir = os.path.join(data_args.memory_basic_dir, data_args.memory_file)
    if name in memory.keys():
        user_memory = memory[name]
        memory_index_path = os.path.join(data_args.memory_basic_dir, f"memory_index/{name}_index")
        os.makedirs(os.path.dirname(memory_index_path), exist_ok=True)
        if (not os.path.exists(memory_index_path)) or update_memory_index:
            # (the reference deletes the directory here and rebuilds the index from every document, :78-79; this retriever writes the index
            #  append-only - it compares what is on disk with the documents and adds only what is new, or rewrites it when the history is not an
            #  extension of it (local_doc_qa.LocalMemoryRetrieval._persist) - so the directory stays; a retriever without that takes the reference's path)
            if os.path.exists(memory_index_path) and not hasattr(local_memory_qa, "_persist"):
                shutil.rmtree(memory_index_path)
            memory_index_path, _ = local_memory_qa.init_memory_vector_store(filepath=memory_dir, vs_path=memory_index_path,
                                                                            user_name=name, cur_date=cur_date)
        user_memory_index = local_memory_qa.load_memory_index(memory_index_path) if memory_index_path else None
        msg = f"欢迎回来，{name}！" if data_args.language == "cn" else f"Wellcome Back, {name}！"
        return msg, user_memory, memory, name, user_memory_index
    memory[name] = {}
    memory[name].update({"name": name})
    msg = (f"欢迎新用户{name}！我会记住你的名字，下次见面就能叫你的名字啦！" if data_args.language == "cn"
           else f"Welcome, new user {name}! I will remember your name, so next time we meet, I'll be able to call you by your name!")
    return msg, memory[name], memory, name, user_memory_index


def save_local_memory(memory, b, user_name, data_args):
    """append the last (query, response) under today's date and rewrite the JSON file (reference :95-110)"""
    memory_dir = os.path.join(data_args.memory_basic_dir, data_args.memory_file)
    date = time.strftime("%Y-%m-%d", time.localtime())
    if memory[user_name].get("history") is None:
        memory[user_name].update({"history": {}})
    if memory[user_name]["history"].get(date) is None:
        memory[user_name]["history"][date] = []
    memory[user_name]["history"][date].append({"query": b[-1][0], "response": b[-1][1]})
    json.dump(memory, open(memory_dir, "w", encoding="utf-8"), ensure_ascii=False, indent=4)
    return memory
