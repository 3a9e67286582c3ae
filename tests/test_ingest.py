"""CPU: the asynchronous ingest pipeline (ordering, ragged tail, back-pressure, producer errors) with a stand-in encoder."""
import numpy as np
import pytest
import torch

from streamchat_amd.ingest import AsyncFrameIngest


def _enc(frames, out):
    out.copy_(frames.float().mean(dim=(1, 2)).view(frames.shape[0], 1, 3))          # "features" = per-channel mean of the frame


def test_order_ragged_tail_and_constant_staging():
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, (23, 6, 5, 3), dtype=np.uint8)
    ing = AsyncFrameIngest(_enc, (6, 5, 3), micro_batch=4, depth=2, device="cpu")
    bank = torch.zeros(30, 1, 3)
    n = ing.run(iter(frames), bank, start=2)
    assert n == 23 and ing.stats["micro_batches"] == 6                             # 5 full + a tail of 3
    ref = torch.from_numpy(frames).float().mean(dim=(1, 2))
    assert torch.equal(bank[2:25, 0], ref) and torch.count_nonzero(bank[25:]) == 0 and torch.count_nonzero(bank[:2]) == 0
    assert len(ing.stage) == 2 and ing.stage[0].shape == (4, 6, 5, 3)


def test_bank_overflow_and_producer_error_are_loud():
    ing = AsyncFrameIngest(_enc, (2, 2, 3), micro_batch=2, depth=2, device="cpu")
    with pytest.raises(ValueError):
        ing.run(iter(np.zeros((5, 2, 2, 3), np.uint8)), torch.zeros(4, 1, 3))

    def bad():
        yield np.zeros((2, 2, 3), np.uint8)
        raise RuntimeError("decoder died")
    ing2 = AsyncFrameIngest(_enc, (2, 2, 3), micro_batch=2, depth=2, device="cpu")
    with pytest.raises(RuntimeError, match="decoder died"):
        ing2.run(bad(), torch.zeros(4, 1, 3))


def test_consumer_failure_stops_the_producer_and_the_pipeline_is_reusable():
    """an exception in run() (bank overflow / encode error mid-stream) must not leave the producer thread blocked on
    free.get() forever; afterwards the same ingest object works again."""
    import threading
    before = threading.active_count()
    ing = AsyncFrameIngest(_enc, (2, 2, 3), micro_batch=2, depth=2, device="cpu")
    with pytest.raises(ValueError):
        ing.run(iter(np.zeros((50, 2, 2, 3), np.uint8)), torch.zeros(4, 1, 3))         # 50 frames into a 4-frame bank

    def boom(frames, out):
        raise RuntimeError("encode failed")
    ing_b = AsyncFrameIngest(boom, (2, 2, 3), micro_batch=2, depth=2, device="cpu")
    with pytest.raises(RuntimeError, match="encode failed"):
        ing_b.run(iter(np.zeros((50, 2, 2, 3), np.uint8)), torch.zeros(64, 1, 3))
    assert threading.active_count() == before                                          # both producers are gone
    frames = np.arange(6 * 12, dtype=np.uint8).reshape(6, 2, 2, 3)
    bank = torch.zeros(6, 1, 3)
    assert ing.run(iter(frames), bank) == 6
    assert torch.equal(bank[:, 0], torch.from_numpy(frames).float().mean(dim=(1, 2)))
