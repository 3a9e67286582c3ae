"""Beam search (deterministic: `num_beams > 1`, `do_sample=False`) with the semantics of transformers' `generate` - the call the reference makes
when it is started with --num_beams N (inference_streaming_longva_v2.py:252-256 -> llava_qwen.py:137-155 -> GenerationMixin._beam_search).
HF's defaults for everything the reference does not set: length_penalty 1.0, early_stopping False, one returned sequence; prompts are
inputs_embeds, so the decoder prompt length is 0 and lengths count generated tokens only.

What HF does per step, restated (transformers 5.x `_beam_search`; the 4.37.2 `BeamSearchScorer` the reference pins keeps the same hypotheses):
  log-softmax of every running beam's logits + the beam's score -> the best K = max(2, 1 + #eos) x N continuations over all beams;
  a continuation that ends in an EOS id or reaches max length is FINISHED: if it is among the first N of the K it competes, with its score
  divided by (generated length)^length_penalty, for one of the N places of finished hypotheses; the best N unfinished continuations run on;
  the search stops when no running beam can beat the worst finished hypothesis any more (best running score / current length, HF's
  `early_stopping=False` heuristic) or every continuation has hit a stopping criterion.  Returns the best finished hypothesis.

This module is the BOOKKEEPING only (a few dozen numbers per step, on the host, in fp32 like HF's tensors); the model is a callback:
  step(tokens [N] int64, beam_idx [N] int64) -> logits [N, V]   "reorder the KV caches by beam_idx, feed one token per beam"
so that the same code is driven by the HIP decoder (llm.LlavaQwenForCausalLM) and, in the CPU test, by HF's own tiny model."""
import torch

NEG = -1.0e9


def beam_search(first_logits, step, num_beams, max_new_tokens, eos_ids=(), length_penalty=1.0):
    """first_logits [V] or [N, V]: logits after the prompt (every beam starts from the same prompt).  Returns (token ids list, score)."""
    N, max_length = int(num_beams), int(max_new_tokens)
    logits = first_logits if first_logits.dim() == 2 else first_logits.unsqueeze(0).expand(N, -1)
    V = logits.shape[-1]
    eos = torch.tensor(sorted(set(int(e) for e in eos_ids)), dtype=torch.int64)
    keep = max(2, 1 + eos.numel()) * N
    top_mask = torch.zeros(keep, dtype=torch.bool); top_mask[:N] = True
    running_seq = torch.zeros((N, max_length), dtype=torch.int64)
    sequences = running_seq.clone()
    running_scores = torch.zeros(N, dtype=torch.float32); running_scores[1:] = NEG          # only the first beam counts at step 0
    beam_scores = torch.full((N,), NEG, dtype=torch.float32)
    is_finished = torch.zeros(N, dtype=torch.bool)
    running_bidx = torch.full((N, max_length), -1, dtype=torch.int64)
    beam_indices = running_bidx.clone()
    unsatisfied = True
    cur_len = 0
    while True:
        lp = torch.log_softmax(logits.float(), dim=-1)
        acc = (lp + running_scores.to(lp.device)[:, None]).reshape(-1)
        topv, topi = torch.topk(acc, keep)
        topv, topi = topv.float().cpu(), topi.cpu()
        cur_beam, ids = topi // V, topi % V
        topk_seq = running_seq[cur_beam].clone(); topk_seq[:, cur_len] = ids
        topk_bidx = running_bidx[cur_beam].clone(); topk_bidx[:, cur_len] = cur_beam
        hits = (torch.isin(ids, eos) if eos.numel() else torch.zeros(keep, dtype=torch.bool)) | (cur_len + 1 >= max_length)
        # the N best unfinished continuations run on
        run_lp = topv + hits.to(torch.float32) * NEG
        nxt = torch.topk(run_lp, N)[1]
        running_seq, running_scores, running_bidx = topk_seq[nxt], run_lp[nxt], topk_bidx[nxt]
        # finished hypotheses: only the first N of the K may enter, length-normalised
        just = hits & top_mask
        fin = topv / float((cur_len + 1) ** length_penalty)
        fin = fin + (0.0 if unsatisfied else 1.0) * NEG
        fin = fin + (~just).to(torch.float32) * NEG
        m_scores = torch.cat((beam_scores, fin))
        sel = torch.topk(m_scores, N)[1]
        sequences = torch.cat((sequences, topk_seq))[sel]
        beam_indices = torch.cat((beam_indices, topk_bidx))[sel]
        is_finished = torch.cat((is_finished, just))[sel]
        beam_scores = m_scores[sel]
        origin = running_bidx[:, cur_len].clone()                           # which beam's cache each new running beam continues
        cur_len += 1
        best_possible = running_scores[:1] / float(cur_len ** length_penalty)
        worst_finished = torch.where(is_finished, beam_scores.min().expand(N), torch.full((N,), NEG))
        unsatisfied = unsatisfied and bool((best_possible > worst_finished).any())
        if not (unsatisfied and not bool(hits.all())):
            break
        logits = step(running_seq[:, cur_len - 1].clone(), origin)
    n = int((beam_indices[0] >= 0).sum())
    return sequences[0, :n].tolist(), float(beam_scores[0])
