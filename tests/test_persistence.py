"""CPU: save / load of the visual memory tree round-trips every field and keeps shared storage shared."""
import torch

from streamchat_amd import persistence as P, utiles as U


def _tree():
    bank = torch.arange(6 * 2 * 4, dtype=torch.float16).reshape(6, 2, 4)
    leaves = [U.MultimodalTreeNode(bank[i * 2:(i + 1) * 2], f"chunk {i}", depth=0) for i in range(3)]
    root = U.MultimodalTreeNode(torch.ones(2, 2, 4), "summary é", labels=torch.tensor([0, 1, 1, 0]), depth=1)
    root.children = leaves[:2]
    return [root, leaves[2]], [bank[5:6]]


def test_round_trip(tmp_path):
    nodes, short = _tree()
    man = P.save_memory_tree(nodes, str(tmp_path / "sess" / "tree"), short, extra=dict(round=3))
    assert man["nodes"][0]["children"][1]["text"] == "chunk 1"
    back, short2, extra = P.load_memory_tree(str(tmp_path / "sess" / "tree"), device="cpu")
    assert extra == dict(round=3) and len(back) == 2 and len(short2) == 1
    def same(a, b):
        assert a.text == b.text and a.depth == b.depth and len(a.children) == len(b.children)
        assert torch.equal(a.centroids, b.centroids) and a.centroids.dtype == b.centroids.dtype
        la = a.labels.tolist() if torch.is_tensor(a.labels) else a.labels
        assert la == b.labels
        for x, y in zip(a.children, b.children):
            same(x, y)
    for a, b in zip(nodes, back):
        same(a, b)
    assert torch.equal(short2[0], short[0])
    assert U.count_nodes_by_depth(back) == U.count_nodes_by_depth(nodes)


def test_shared_tensors_written_once(tmp_path):
    t = torch.randn(3, 2, 4)
    a, b = U.MultimodalTreeNode(t, "a"), U.MultimodalTreeNode(t, "b")
    man = P.save_memory_tree([a, b], str(tmp_path / "t"))
    assert man["nodes"][0]["centroids"] == man["nodes"][1]["centroids"]
    back, _, _ = P.load_memory_tree(str(tmp_path / "t"), device="cpu")
    assert back[0].centroids.data_ptr() == back[1].centroids.data_ptr()


def test_bad_format_rejected(tmp_path):
    import json, pytest
    P.save_memory_tree([], str(tmp_path / "t"))
    m = json.load(open(tmp_path / "t.json")); m["format"] = "x"
    json.dump(m, open(tmp_path / "t.json", "w"))
    with pytest.raises(ValueError):
        P.load_memory_tree(str(tmp_path / "t"), device="cpu")
