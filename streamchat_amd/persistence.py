"""Persistence of the visual memory tree.  The reference only writes the answers JSON (inference_streaming_longva_v2.py:
657-675) and the dialogue `memory_{i}.json` (memory_bank/memory_utils.py:95-110); the tree of `MultimodalTreeNode`s (utiles.py:48-56) lives and
dies with the process.  Here a session's tree is saved as ONE safetensors file (every node's `.centroids`, deduplicated by storage so
a depth-0 node that is a view into the feature bank is written once) plus a JSON manifest (texts, depths, labels, children), and comes
back with identical field values, so a multi-round session can resume without re-encoding or re-captioning the stream."""
import json
import os

import torch
from safetensors.torch import load_file, save_file

from .utiles import MultimodalTreeNode

FORMAT = "streamchat-memory-tree/1"


def _walk(nodes, tensors, seen):
    out = []
    for n in nodes:
        c = n.centroids
        key = None
        if torch.is_tensor(c):
            ident = (c.data_ptr(), tuple(c.shape), tuple(c.stride()), str(c.dtype))
            key = seen.get(ident)
            if key is None:
                key = f"t{len(tensors)}"
                seen[ident] = key
                tensors[key] = c.detach().cpu().contiguous().clone()       # own storage: safetensors refuses aliasing tensors
        labels = n.labels.detach().cpu().tolist() if torch.is_tensor(n.labels) else n.labels
        out.append(dict(text=n.text, depth=int(n.depth), centroids=key, labels=labels, text_distance=n.text_distance,
                        image_distance=n.image_distance, children=_walk(n.children, tensors, seen)))
    return out


def save_memory_tree(nodes, path, short_memory=None, extra=None):
    """nodes: list[MultimodalTreeNode] (the long memory);  short_memory: optional list of [1,P,D] tensors;  extra: JSON-able dict.
    Writes `<path>.safetensors` and `<path>.json`; returns the manifest."""
    tensors, seen = {}, {}
    manifest = dict(format=FORMAT, nodes=_walk(nodes, tensors, seen), short=[], extra=extra or {})
    for t in short_memory or []:
        key = f"t{len(tensors)}"
        tensors[key] = t.detach().cpu().contiguous().clone()
        manifest["short"].append(key)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    save_file(tensors, path + ".safetensors", metadata=dict(format=FORMAT))
    with open(path + ".json", "w", encoding="utf-8") as f:
        json.dump(manifest, f, ensure_ascii=False)
    return manifest


def load_memory_tree(path, device="cuda"):
    """Returns (nodes, short_memory, extra).  Tensors are loaded straight to `device`; nodes that shared storage share it again."""
    with open(path + ".json", encoding="utf-8") as f:
        manifest = json.load(f)
    if manifest.get("format") != FORMAT:
        raise ValueError(f"{path}.json: unknown memory-tree format {manifest.get('format')!r}")
    tensors = load_file(path + ".safetensors", device=str(device))

    def build(d):
        n = MultimodalTreeNode(tensors[d["centroids"]] if d["centroids"] is not None else None, d["text"], d["text_distance"],
                               d["image_distance"], d["labels"], depth=d["depth"])
        n.children = [build(c) for c in d["children"]]
        return n
    return [build(d) for d in manifest["nodes"]], [tensors[k] for k in manifest["short"]], manifest["extra"]
