"""Prompt tokenisation with the `<image>` sentinel (mirror of reference longva/mm_utils.py:341-360)."""
import torch

IMAGE_TOKEN_INDEX = -200


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, return_tensors=None):
    """Split at "<image>", tokenise the text pieces, and join them with the sentinel id (-200).  A leading BOS is
    kept once if the tokenizer emits one (Qwen2 does not — Q17)."""
    chunks = [tokenizer(chunk).input_ids for chunk in prompt.split("<image>")]
    input_ids, offset = [], 0
    bos = getattr(tokenizer, "bos_token_id", None)
    if len(chunks) > 0 and len(chunks[0]) > 0 and bos is not None and chunks[0][0] == bos:
        offset = 1
        input_ids.append(chunks[0][0])
    sep = [image_token_index] * (offset + 1)
    pieces = []
    for i, ids in enumerate(chunks):          # chunk, sep, chunk, sep, ..., chunk
        if i > 0:
            pieces.append(sep)
        pieces.append(ids)
    for x in pieces:
        input_ids.extend(x[offset:])          # drops each chunk's BOS; a 2-long separator loses one copy
    if return_tensors is not None:
        if return_tensors == "pt":
            return torch.tensor(input_ids, dtype=torch.long)
        raise ValueError(f"Unsupported tensor type: {return_tensors}")
    return input_ids
