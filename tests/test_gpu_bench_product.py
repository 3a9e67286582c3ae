"""GPU: the `product` leg of bench.py (round 4; on by default in the driver's bench line): the C3 step with the HIP LLM itself as chunk captioner
+ merge summariser (reference utiles.py:539-559,591-607) through llm.BatchDecoder.  A 2-layer model at the Qwen2-7B widths keeps it short; what
is checked is the plumbing the driver line depends on: every chunk of the update goes through ONE batched generate, the record is complete and
self-consistent, the step's retrieval still works on the generated captions, and the headline path is untouched afterwards."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def test_product_step_with_the_llm_as_chunk_captioner():
    import bench
    from streamchat_amd import llm as LM
    dev = torch.device("cuda:0")
    pipe = bench.Pipeline(dev, 440, with_llm=False)                 # 11 chunks of 40 frames: ten of them merge (one k-means T = 400)
    qc = LM.Qwen2ConfigLite(**dict(LM.QWEN2_7B, layers=2))
    pipe.model = LM.LlavaQwenForCausalLM(LM.Qwen2Model(LM.random_qwen2_state_dict(qc, seed=4, device=dev), qc, device=dev, max_seq=53248), pipe.enc)
    plain = pipe.step()                                             # the headline step (synthetic captions)
    ctx_plain, tok_plain = pipe.last["context"], int(pipe.last["first_token"][0, 0])
    rec = bench.measure_product(pipe, 1, 440)
    pipe.captioner = None
    assert rec["chunks_per_step"] == 11 and rec["steps"] == 1
    assert 40 * 576 < rec["prompt_tokens_per_chunk"] < 40 * 576 + 200          # 23 040 image tokens + the caption prompt
    assert rec["product_frames_per_s"] > 0 and rec["caption_prefill_s_per_step"] > 0 and rec["caption_decode_s_per_step"] > 0
    # (the roofline figures use the 28-layer 7B flop / byte model of SURVEY 8(d): with this 2-layer stand-in only their presence is checked)
    assert rec["caption_prefill"]["bound"] == "mfma" and rec["caption_prefill"]["frac"] > 0
    assert rec["caption_decode"]["bound"] == "hbm" and rec["caption_decode"]["frac"] > 0
    assert rec["caption_decode"]["tokens_per_s_aggregate"] > 0 and rec["ms_per_step"] >= 1e3 * (rec["caption_prefill_s_per_step"] + rec["caption_decode_s_per_step"])
    # the step ran the whole path on the generated captions: a tree with one merged node + one depth-0 node, a retrieved context, a first token
    assert [n.depth for n in pipe.last["tree"]] == [1, 0] and pipe.last["context"] > 5 * 576 + 2 * 40 * 576
    # and the headline step is what it was (the captioner is gone again)
    again = pipe.step()
    assert pipe.last["context"] == ctx_plain and int(pipe.last["first_token"][0, 0]) == tok_plain
    assert [n.text for n in again["tree"]] == [n.text for n in plain["tree"]]
