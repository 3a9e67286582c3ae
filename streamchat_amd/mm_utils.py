"""Prompt tokenisation with the `<image>` sentinel (mirror of reference longva/mm_utils.py:341-360) and the host half of
`process_images` (reference utiles.py:71-87 -> CLIPImageProcessor.preprocess): resize + centre crop of a decoded frame."""
import numpy as np
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


def resize_center_crop_u8(frame, size=336):
    """uint8 [H, W, 3] RGB frame of any resolution -> uint8 [size, size, 3]: what `CLIPImageProcessor.preprocess` does BEFORE the
    rescale / normalise step that `sc_preprocess_*` fuses into the patch gather (reference utiles.py:71-87, inference_streaming_
    longva_v2.py:503-516): shortest edge -> `size` with PIL bicubic (the long edge becomes int(size * long / short)), then a centre
    crop with top = (h - size) // 2, left = (w - size) // 2.  Runs on the host next to the video decode; frames that already are
    size x size pass through untouched (both steps are identities, as for the synthetic 336 x 336 streams).
    Pinned against the processor itself in tests/test_checkpoint_and_frames.py."""
    frame = np.asarray(frame)
    if frame.ndim != 3 or frame.shape[2] != 3 or frame.dtype != np.uint8:
        raise ValueError(f"resize_center_crop_u8: expected a uint8 [H, W, 3] frame, got {frame.dtype} {frame.shape}")
    h, w = frame.shape[:2]
    if (h, w) == (size, size):
        return frame
    from PIL import Image
    short, long = (h, w) if h <= w else (w, h)
    new_short, new_long = size, int(size * long / short)
    nh, nw = (new_short, new_long) if h <= w else (new_long, new_short)
    img = np.asarray(Image.fromarray(frame).resize((nw, nh), resample=Image.BICUBIC))
    top, left = (nh - size) // 2, (nw - size) // 2
    return np.ascontiguousarray(img[top:top + size, left:left + size])
