"""GPU: session.StreamingSession - the reader / updater thread and the QA-decode thread on two CU partitions (reference
previous_version/streaming_demo_llava_next_3.py:967-991) must produce exactly what the same jobs produce one after the other on one stream:
short-memory frames, retrieved nodes, context lengths, first tokens and every decoded token id."""
import numpy as np
import pytest
import torch

from streamchat_amd import llm as LM, ops, session as SS, synthetic, text as T, vision as V
from tests import _composed as TC

pytestmark = pytest.mark.gpu
MEM = dict(short_window=20, remember_window=5, tau=5, chunk_size=8, num_clusters=3, interval=3)


def _build(dev):
    cfg = V.CLIPVisionConfigLite(hidden=128, layers=2, heads=2, intermediate=256, patch=14, image_size=56)
    enc = V.FrameEncoder(V.CLIPVisionTower(V.random_clip_state_dict(cfg, seed=0, device=dev, std=0.08), cfg, device=dev),
                         V.MMProjector(V.random_projector_state_dict(128, 256, seed=1, device=dev, std=0.08), device=dev), micro_batch=16)
    qc = LM.Qwen2ConfigLite(hidden=256, layers=2, heads=4, kv_heads=2, intermediate=512, vocab=1024)
    model = LM.LlavaQwenForCausalLM(LM.Qwen2Model(LM.random_qwen2_state_dict(qc, seed=4, device=dev, std=0.05), qc, device=dev, max_seq=2048), enc)
    bc = T.BertConfigLite(hidden=128, layers=2, heads=4, intermediate=256, vocab=2048, max_pos=128)
    bert = T.BertEncoder(T.random_bert_state_dict(bc, seed=2, device=dev, std=0.05), bc, device=dev)
    return enc, model, bert


class _Tok(synthetic.SyntheticTokenizer):
    def __call__(self, text, **kw):
        r = super().__call__(text, **kw)
        r.input_ids = [t % 1000 + 5 for t in r.input_ids]          # into the tiny vocabulary
        return r


def _run(overlap, segs, questions, n_new):
    dev = torch.device("cuda:0")
    enc, model, bert = _build(dev)
    s = SS.StreamingSession(model, enc, bert, T.HashTokenizer(vocab=2048, max_len=64), _Tok(), MEM, TC.PositionCaptioner(dev), synthetic.SyntheticTokenizer(),
                            overlap=overlap, decode_cus=96, max_new_tokens=n_new, max_context=2048)
    for f, q in zip(segs, questions):
        s.submit(f, q)
    out = s.results()
    s.close()
    return out


def test_overlapped_session_equals_serial_session():
    dev = torch.device("cuda:0")
    u8 = torch.from_numpy(TC.crossfade_stream(4 * 40, seed=77, period=8, h=56, w=56)).to(dev)
    segs = [u8[i * 40:(i + 1) * 40] for i in range(4)]
    questions = [f"segment {i}: " + " ".join(synthetic.caption(i + 1).split()[2:14]) for i in range(4)]
    a = _run(False, segs, questions, 24)
    b = _run(True, segs, questions, 24)
    assert len(a) == len(b) == 4
    for ra, rb in zip(a, b):
        for k in ("short", "path_text", "retrieved_rows", "retrieved_crc", "context", "first_token", "tokens"):
            assert ra[k] == rb[k], (ra["segment"], k, ra[k], rb[k])
        assert len(ra["tokens"]) == 24 and len(ra["short"]) == 5
    assert any(r["retrieved_rows"] for r in a) and a[1]["context"] > 16 * 5           # something was retrieved and spliced
    # GPU time lines (round 6): both runs know when each side's job ran on the GPU; serially the two sides never execute together
    ca, cb = SS.StreamingSession.co_running(a), SS.StreamingSession.co_running(b)
    assert ca["co_running_fraction"] == 0.0 and 0.0 <= cb["co_running_fraction"] <= 1.0
    assert all(r["gpu_ms"]["mfma"][0] <= r["gpu_ms"]["mfma"][1] <= r["gpu_ms"]["decode"][0] <= r["gpu_ms"]["decode"][1] for r in a + b)
    print("\n[session] 4 segments x 40 frames, 24 tokens each: overlapped == serial;", [r["context"] for r in a], a[-1]["tokens"][:8])


def test_a_failing_decode_fails_fast_instead_of_waiting_for_its_slot():
    """after the first error a worker skips the queued jobs - their bookkeeping (slot release, pending count) must still run, so the
    reader / updater side raises from results() at once instead of waiting 600 s for a KV-cache slot nobody frees."""
    import time
    dev = torch.device("cuda:0")
    enc, model, bert = _build(dev)
    s = SS.StreamingSession(model, enc, bert, T.HashTokenizer(vocab=2048, max_len=64), _Tok(), MEM, TC.PositionCaptioner(dev), synthetic.SyntheticTokenizer(),
                            overlap=True, decode_cus=96, max_new_tokens=8, max_context=2048)
    u8 = torch.from_numpy(TC.crossfade_stream(4 * 40, seed=78, period=8, h=56, w=56)).to(dev)

    def boom(*a, **k):
        raise RuntimeError("decode failed on purpose")
    s.graphs[0].run = boom                                       # the answer of segment 0 (slot 0) fails
    t0 = time.time()
    for i in range(4):
        try:
            s.submit(u8[i * 40:(i + 1) * 40], f"segment {i}: what happened")
        except RuntimeError:
            break
    try:
        s.results()
        raised = False
    except RuntimeError as e:
        raised = "on purpose" in str(e) or "has failed" in str(e)
    s.close()
    assert raised and time.time() - t0 < 120


def test_worker_reraises_on_the_callers_thread():
    w = SS._Worker("t")
    w.submit(lambda: (_ for _ in ()).throw(ValueError("boom")))
    with pytest.raises(ValueError, match="boom"):
        w.drain()
    w.submit(lambda: None)
    w.drain()
    w.close()


def test_move_to_stream_when_switches_at_the_next_launch_and_keeps_the_order():
    """ops.move_to_stream_when: THIS thread's launches move to another stream at the first library launch after the condition turns true, ordered
    behind what was enqueued before; cancelled by (None, None); never fires while the condition is false."""
    dev = torch.device("cuda:0")
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    flag = {"go": False}
    x = torch.randn(64, 1024, device=dev).half()
    g = torch.ones(1024, device=dev).half()
    with torch.cuda.stream(s1):
        ops.move_to_stream_when(lambda: flag["go"], s2)
        a = ops.rmsnorm(x, g, 1e-6)
        assert torch.cuda.current_stream(dev) == s1                     # condition false: nothing moves
        big = torch.randn(4096, 4096, device=dev)
        for _ in range(20):
            big = big @ big * 1e-3                                       # keep s1 busy so that an unordered switch would be seen
        y = big[:64, :1024].half().contiguous()
        flag["go"] = True
        b = ops.rmsnorm(y, g, 1e-6)                                      # switches to s2 first, behind everything enqueued on s1
        assert torch.cuda.current_stream(dev) == s2
        ops.move_to_stream_when(None, None)
    assert torch.cuda.current_stream(dev) != s2                          # the `with` restored the caller's stream
    torch.cuda.synchronize()
    ref = torch.nn.functional.rms_norm(y.float(), (1024,), g.float(), 1e-6)          # from the FINAL y: b must have waited for it
    assert (b.float() - ref).abs().max().item() < 2e-2 and torch.isfinite(a.float()).all()


def test_host_frames_go_through_the_reader_thread_and_give_the_same_session():
    """round 6: a segment's frames may live on the HOST (numpy / CPU tensor / a decoder's frames); the reader stage is then
    ingest.AsyncFrameIngest (producer thread -> pinned staging -> H2D on a copy stream -> encode) inside the MFMA job.  Same records as with
    device-resident frames, serially and overlapped."""
    dev = torch.device("cuda:0")
    u8 = TC.crossfade_stream(3 * 40, seed=79, period=8, h=56, w=56)
    questions = [f"segment {i}: " + " ".join(synthetic.caption(i + 1).split()[2:14]) for i in range(3)]
    on_dev = [torch.from_numpy(u8[i * 40:(i + 1) * 40]).to(dev) for i in range(3)]
    on_host = [u8[i * 40:(i + 1) * 40] for i in range(3)]                          # numpy uint8
    a = _run(False, on_dev, questions, 12)
    b = _run(True, on_host, questions, 12)
    c = _run(False, [torch.from_numpy(x) for x in on_host], questions, 12)         # CPU tensors, serial
    for ra, rb, rc in zip(a, b, c):
        for k in ("short", "path_text", "retrieved_rows", "retrieved_crc", "context", "first_token", "tokens"):
            assert ra[k] == rb[k] == rc[k], (ra["segment"], k)
