"""Dialogue-memory bookkeeping of the streaming entry point (the behaviour of reference memory_bank/memory_utils.py:61-110: what a user is greeted
with, where the per-user index lives, how a (query, response) pair is filed under today's date in the JSON file).  The gradio State unwrapping of
the reference is not needed here - the entry point passes plain objects.  Messages, paths and the JSON layout are the on-disk / on-screen API a
drop-in has to keep; the code around them is this package's own."""
import datetime
import json
import os
import shutil
import time

_GREETING = {
    # (known user, language) -> text; the wording (incl. "Wellcome") is the reference's (:86, :91)
    (True, "cn"): "欢迎回来，{name}！",
    (True, "en"): "Wellcome Back, {name}！",
    (False, "cn"): "欢迎新用户{name}！我会记住你的名字，下次见面就能叫你的名字啦！",
    (False, "en"): "Welcome, new user {name}! I will remember your name, so next time we meet, I'll be able to call you by your name!",
}


def _greeting(known, language, name):
    return _GREETING[(known, "cn" if language == "cn" else "en")].format(name=name)


def _memory_file(data_args):
    return os.path.join(data_args.memory_basic_dir, data_args.memory_file)


def _index_dir(data_args, name):
    """<memory_basic_dir>/memory_index/<name>_index (reference :72), its parent created"""
    path = os.path.join(data_args.memory_basic_dir, "memory_index", f"{name}_index")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    return path


def _refresh_index(retriever, data_args, name, rebuild):
    """The user's dialogue index on disk, (re)built from the memory file when it is missing or `rebuild` is set; returns its path (None if
    the retriever found nothing to index).  The reference deletes the directory and re-embeds every document each time (:78-80); a retriever
    that persists append-only (local_doc_qa.LocalMemoryRetrieval._persist compares what is on disk with the documents, adds what is new and
    rewrites only when the history is not an extension of it) keeps the directory."""
    path = _index_dir(data_args, name)
    if os.path.exists(path) and not rebuild:
        return path
    if os.path.exists(path) and not hasattr(retriever, "_persist"):
        shutil.rmtree(path)
    today = datetime.date.today().strftime("%Y-%m-%d")
    path, _ = retriever.init_memory_vector_store(filepath=_memory_file(data_args), vs_path=path, user_name=name, cur_date=today)
    return path


def enter_name(name, memory, local_memory_qa, data_args, update_memory_index=True):
    """reference :61-92.  Returns (msg, user_memory, memory, name, user_memory_index); a new user gets user_memory_index=None, so the first
    question of a video has no history prompt (Q16)."""
    known = name in memory
    if not known:
        memory[name] = {"name": name}
        return _greeting(False, data_args.language, name), memory[name], memory, name, None
    path = _refresh_index(local_memory_qa, data_args, name, update_memory_index)
    index = local_memory_qa.load_memory_index(path) if path else None
    return _greeting(True, data_args.language, name), memory[name], memory, name, index


def save_local_memory(memory, b, user_name, data_args):
    """file the last (query, response) of the chat `b` under today's date and rewrite the JSON file (reference :97-110)"""
    query, response = b[-1][0], b[-1][1]
    today = time.strftime("%Y-%m-%d", time.localtime())
    by_date = memory[user_name].setdefault("history", {})
    if by_date is None:                       # ("history": null in a hand-edited file counts as absent, as upstream's `.get(...) is None`)
        by_date = memory[user_name]["history"] = {}
    if by_date.get(today) is None:
        by_date[today] = []
    by_date[today].append({"query": query, "response": response})
    with open(_memory_file(data_args), "w", encoding="utf-8") as f:
        json.dump(memory, f, ensure_ascii=False, indent=4)
    return memory
