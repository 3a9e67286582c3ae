"""Asynchronous frame ingest: host decode -> pinned staging -> H2D copy -> fused preprocess + ViT + projector, with
bounded micro-batches so a stream of any length is encoded in constant host / staging memory and the copy of micro-batch i+1 overlaps
the encode of micro-batch i.

The reference reads a whole question interval into one Python list, stacks it and runs ONE `encode_images` over it
(inference_streaming_longva_v2.py:454-531: `cap.set` + `cap.read` per frame, PIL -> numpy -> torch per frame, a single mega-batch through
the tower with 25 retained hidden states).  Here a producer thread pulls uint8 frames from any iterator (cv2, a socket, a synthetic
source) into one of `depth` pinned staging buffers; the consumer issues the H2D copy on a copy stream and the encode on the compute
stream, tied by events; features land in a caller-owned bank ([N, 576, 3584] fp16) in arrival order.  Back-pressure is the bounded queue:
the producer blocks when `depth` micro-batches are staged but not yet consumed."""
import queue
import threading

import numpy as np
import torch


class AsyncFrameIngest:
    def __init__(self, encode_u8, frame_shape, micro_batch=56, depth=2, device="cuda", pin=True):
        """encode_u8(frames_u8_device [n,H,W,3], out=[n,P,D]) -> None/out  (FrameEncoder.encode_frames_u8);  frame_shape = (H, W, 3)."""
        self.encode_u8, self.mb, self.depth, self.device = encode_u8, micro_batch, depth, torch.device(device)
        cuda = self.device.type == "cuda"
        self.stage = [torch.empty((micro_batch, *frame_shape), dtype=torch.uint8, pin_memory=pin and cuda) for _ in range(depth)]
        self.dev = [torch.empty((micro_batch, *frame_shape), dtype=torch.uint8, device=self.device) for _ in range(depth)]
        self.copy_stream = torch.cuda.Stream(self.device) if cuda else None
        self.free = queue.Queue()
        for i in range(depth):
            self.free.put(i)
        self.ready = queue.Queue(maxsize=depth)
        self.stats = dict(frames=0, micro_batches=0, producer_waits=0)
        self._err = None
        self._stop = False

    def _producer(self, frames):
        try:
            slot, n = None, 0
            for f in frames:
                if self._stop:
                    return
                if slot is None:
                    if self.free.empty():
                        self.stats["producer_waits"] += 1
                    slot, n = self.free.get(), 0
                    if slot is None or self._stop:          # the consumer failed and released us
                        return
                self.stage[slot][n].copy_(torch.as_tensor(np.ascontiguousarray(f)) if not torch.is_tensor(f) else f)
                n += 1
                if n == self.mb:
                    self.ready.put((slot, n))
                    slot = None
            if slot is not None and n > 0:
                self.ready.put((slot, n))
        except Exception as e:                      # surface producer failures in the consumer thread
            self._err = e
        finally:
            self.ready.put(None)

    def run(self, frames, bank, start=0):
        """Encode every frame of the iterable `frames` into bank[start:...] in order.  Returns the number of frames written.  The
        call returns after the last encode has been ISSUED on the current stream (no synchronisation)."""
        self._stop, self._err = False, None
        t = threading.Thread(target=self._producer, args=(frames,), daemon=True)
        t.start()
        pos = start
        copied = [None] * self.depth               # events: device buffer i was consumed by an encode -> its staging slot is free
        try:
            pos = self._consume(bank, pos, start, copied)
        except BaseException:
            # bank overflow / encode error mid-stream: stop the producer (it may be blocked on free.get()) before re-raising, so the
            # thread and its staging slots are not leaked
            self._stop = True
            self.free.put(None)
            while t.is_alive():
                try:
                    self.ready.get(timeout=0.05)
                except Exception:
                    pass
            t.join()
            self._reset_queues()
            raise
        t.join()
        if self._err is not None:
            raise self._err
        return pos - start

    def _reset_queues(self):
        import queue
        for q in (self.free, self.ready):
            while True:
                try:
                    q.get_nowait()
                except queue.Empty:
                    break
        for i in range(self.depth):
            self.free.put(i)

    def _consume(self, bank, pos, start, copied):
        while True:
            item = self.ready.get()
            if item is None:
                break
            slot, n = item
            if pos + n > bank.shape[0]:
                raise ValueError(f"feature bank holds {bank.shape[0]} frames, stream has more than {pos + n - start}")
            if self.copy_stream is not None:
                cur = torch.cuda.current_stream(self.device)
                with torch.cuda.stream(self.copy_stream):
                    if copied[slot] is not None:
                        self.copy_stream.wait_event(copied[slot])          # the encode that last read dev[slot] has finished
                    self.dev[slot][:n].copy_(self.stage[slot][:n], non_blocking=True)
                    h2d = torch.cuda.Event()
                    h2d.record(self.copy_stream)
                h2d.synchronize()                                          # staging slot reusable by the producer ...
                self.free.put(slot)
                cur.wait_event(h2d)                                        # ... and the encode waits for the copy on the device
                self.encode_u8(self.dev[slot][:n], out=bank[pos:pos + n])
                copied[slot] = torch.cuda.Event()
                copied[slot].record(cur)
            else:
                self.dev[slot][:n].copy_(self.stage[slot][:n])
                self.free.put(slot)
                self.encode_u8(self.dev[slot][:n], out=bank[pos:pos + n])
            pos += n
            self.stats["frames"] += n
            self.stats["micro_batches"] += 1
        return pos
