"""GPU: the streaming entry point end to end (tiny random-init architectures, synthetic video): encode -> memory update (with
k-means merges) -> dialogue + tree retrieval -> generate -> persisted results and dialogue memory."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu


def test_run_inference_synthetic_tiny(tmp_path):
    import inference_streaming_longva_v2 as E
    args = E.parse_args(["--video_dir", "none", "--model_name", "none", "--memory_basic_dir", str(tmp_path / "mem"), "--save_file",
                         str(tmp_path / "out.json"), "--annotations", "none", "--language", "en", "--conv-mode", "qwen_1_5", "--synthetic", "1",
                         "--tiny", "--chunk_size", "4", "--num_clusters", "2", "--interval", "3", "--short_window", "6", "--remember_window", "3",
                         "--max_new_tokens", "4", "--multi_modal_memory", "--memory_tree_dir", str(tmp_path / "trees"), "--batch_captions"])
    E.run_inference(args)
    out = json.load(open(tmp_path / "out.json"))
    assert len(out) == 2 and all(set(r) == {"time", "question", "label", "predict", "class", "process_time"} for r in out)
    mem = json.load(open(tmp_path / "mem" / "memory_0.json"))
    turns = [t for day in mem["User"]["history"].values() for t in day]
    assert [t["query"] for t in turns] == [r["question"] for r in out]
    assert os.path.exists(tmp_path / "mem" / "memory_index" / "User_index" / "index.npy")      # dialogue index rebuilt per round
    # the visual memory tree of the session was persisted and loads back onto the device
    from streamchat_amd.persistence import load_memory_tree
    from streamchat_amd import utiles as U
    nodes, short, extra = load_memory_tree(str(tmp_path / "trees" / "video_0"))
    assert len(nodes) > 0 and all(n.centroids.is_cuda for n in nodes) and len(short) > 0 and extra["question"] == out[-1]["question"]
    assert sum(U.count_nodes_by_depth(nodes).values()) >= len(nodes)


def test_parse_args_has_reference_flags():
    import inference_streaming_longva_v2 as E
    a = E.parse_args(["--video_dir", "v", "--model_name", "m", "--memory_basic_dir", "d", "--save_file", "s", "--annotations", "a", "--language", "en"])
    for k, v in dict(chunk_size=20, num_clusters=5, interval=10, short_window=20, remember_window=5, tau=5, compress_rate=1, sample_rate=0.5,
                     temperature=0.2, num_beams=1, memory_search_top_k=1, conv_mode="video-chatgpt_v1", mode="off_line").items():
        assert getattr(a, k) == v


@pytest.mark.parametrize("sampled", [False, True])
def test_overlap_flag_gives_the_same_answers_and_memory(tmp_path, monkeypatch, sampled):
    """--overlap: the next segment's reader / updater on a second host thread + CU partition while the answer is decoded on its own
    partition.  The answers, the dialogue memory and the persisted memory tree equal those of the serial run - with every generate forced greedy
    AND (round 6) with the reference's own sampling settings (temperature 0.2 answers, 0.1 captions): every sequence samples from its own seed,
    answers and captions take their seeds from two role generators, nothing is drawn from state the two threads share.  Per-chunk captioning
    (--batch_captions 0) gives the same run again."""
    import torch
    import inference_streaming_longva_v2 as E
    from streamchat_amd import llm as LM
    from streamchat_amd.persistence import load_memory_tree
    if not sampled:
        real = LM.resolve_sampling
        monkeypatch.setattr(LM, "resolve_sampling", lambda *a, **k: LM.Sampling(0.0, 0, 1.0, real(*a, **k).repetition_penalty))
    runs = {}
    for name, extra in (("serial", ["--overlap", "0"]), ("overlap", []), ("serial_per_chunk", ["--overlap", "0", "--batch_captions", "0"])):
        d = tmp_path / name
        args = E.parse_args(["--video_dir", "none", "--model_name", "none", "--memory_basic_dir", str(d / "mem"), "--save_file", str(d / "out.json"),
                             "--annotations", "none", "--language", "en", "--conv-mode", "qwen_1_5", "--synthetic", "2", "--tiny", "--chunk_size", "4",
                             "--num_clusters", "2", "--interval", "3", "--short_window", "6", "--remember_window", "3", "--max_new_tokens", "12",
                             "--multi_modal_memory", "--memory_tree_dir", str(d / "trees")] + extra)
        assert args.batch_captions == (0 if name == "serial_per_chunk" else 1)          # batched captioning and the overlap are the defaults since round 6
        assert args.overlap == (128 if name == "overlap" else 0)
        os.makedirs(d, exist_ok=True)
        import numpy as np, random
        torch.manual_seed(0); np.random.seed(0); random.seed(0)
        E.run_inference(args)
        out = json.load(open(d / "out.json"))
        nodes, short, _ = load_memory_tree(str(d / "trees" / "video_1"))
        runs[name] = dict(answers=[(r["question"], r["predict"]) for r in out], texts=[n.text for n in nodes], rows=[int(n.centroids.shape[0]) for n in nodes],
                          short=[float(t.float().sum()) for t in short])
    assert len(runs["serial"]["answers"]) == 4
    assert runs["overlap"] == runs["serial"]
    assert runs["serial_per_chunk"] == runs["serial"]


@pytest.mark.parametrize("batched", [False, True])
def test_overlap_with_the_reference_sampling_settings_runs_to_completion(tmp_path, batched):
    """--overlap with the reference's own generation settings (temperature 0.2 answers, temperature 0.1 captions) runs to completion with and
    without batched captions; two videos, three questions each."""
    import inference_streaming_longva_v2 as E
    args = E.parse_args(["--video_dir", "none", "--model_name", "none", "--memory_basic_dir", str(tmp_path / "mem"), "--save_file", str(tmp_path / "out.json"),
                         "--annotations", "none", "--language", "en", "--conv-mode", "qwen_1_5", "--synthetic", "2", "--synthetic_breakpoints", "3", "--tiny",
                         "--chunk_size", "4", "--num_clusters", "2", "--interval", "3", "--short_window", "6", "--remember_window", "3", "--max_new_tokens", "40",
                         "--multi_modal_memory", "--overlap"] + (["--batch_captions", "1"] if batched else ["--batch_captions", "0"]))
    assert args.overlap == 128
    E.run_inference(args)
    out = json.load(open(tmp_path / "out.json"))
    assert len(out) == 6 and all(isinstance(r["predict"], str) for r in out)


def test_entry_point_runs_beam_search_answers(tmp_path):
    """--num_beams 2 --temperature 0 through the whole entry point (round 6: HF-semantics beam search, streamchat_amd/beam.py) with the look-ahead on (the default)
    and off: same answers - the beam path has no prefill hook, the look-ahead then starts when the answer is out."""
    import inference_streaming_longva_v2 as E
    import torch
    outs = []
    for extra in ([], ["--overlap", "0"]):
        d = tmp_path / ("a" + str(len(outs)))
        os.makedirs(d, exist_ok=True)
        args = E.parse_args(["--video_dir", "none", "--model_name", "none", "--memory_basic_dir", str(d / "mem"), "--save_file", str(d / "out.json"),
                             "--annotations", "none", "--language", "en", "--conv-mode", "qwen_1_5", "--synthetic", "1", "--synthetic_breakpoints", "2", "--tiny",
                             "--chunk_size", "4", "--num_clusters", "2", "--interval", "3", "--short_window", "6", "--remember_window", "3", "--max_new_tokens", "10",
                             "--multi_modal_memory", "--num_beams", "2", "--temperature", "0"] + extra)
        import numpy as np, random
        torch.manual_seed(0); np.random.seed(0); random.seed(0)
        E.run_inference(args)
        outs.append([r["predict"] for r in json.load(open(d / "out.json"))])
    assert len(outs[0]) == 2 and outs[0] == outs[1]
