"""Input side of the step: uint8 clip batches from host memory to HBM, overlapped with the previous step.

The reference moves every batch with a blocking pageable copy of normalised f32 frames on the compute stream
(``inputs = inputs.to(rank)``, pretraining/generative/pretrain_videomae.py:294-299, after ToTensor/Normalize on the host,
homeview.py:218-231): 9.63 MB per clip.  At a few thousand clips/s per GPU that copy would be the step.  Here the loader hands over
uint8 frames (2.41 MB per clip; the kernels normalise on the fly, ``BVC_PIXELS_U8`` in include/bvc.h) and ``ClipUploadRing``
moves them on a dedicated copy stream into a ring of device buffers, fenced with events both ways:

    ready[slot]  recorded on the copy stream after the H2D copy      -> the compute stream waits for it in ``get()``
    free[slot]   recorded on the compute stream in ``release()``     -> the copy stream waits for it before overwriting the slot

so step t+1's clips cross PCIe while step t computes, nothing blocks the host except a full ring, and a device buffer is never
rewritten before the step that borrowed it (forward AND backward: the library keeps reading the clip for the pixel targets)
has been enqueued completely.  Pageable sources are first copied into the slot's pinned staging buffer (what a
``DataLoader(pin_memory=True)`` does in its pin thread); pinned sources are copied straight from where they are.
"""
import torch


class ClipUploadRing:
    def __init__(self, shape, device, depth=3, dtype=torch.uint8):
        if depth < 2:
            raise ValueError("ClipUploadRing: depth must be >= 2 (one slot in use, one in flight)")
        self.device = torch.device(device)
        self.shape, self.depth = tuple(shape), depth
        self.dev = [torch.empty(self.shape, dtype=dtype, device=self.device) for _ in range(depth)]
        self.pinned = [torch.empty(self.shape, dtype=dtype, pin_memory=True) for _ in range(depth)]
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.ready = [torch.cuda.Event() for _ in range(depth)]
        self.free = [torch.cuda.Event() for _ in range(depth)]
        self._staged_host = [None] * depth     # keeps a pinned source alive until its copy has been consumed
        self._head = 0        # next slot to stage into
        self._tail = 0        # next slot to hand to the compute stream
        self._count = 0       # staged, not yet handed out
        self._out = []        # handed out, not yet released (oldest first)
        self._used = [False] * depth
        self.bytes_uploaded = 0

    def can_stage(self):
        return self._count + len(self._out) < self.depth

    def stage(self, host_batch):
        """Enqueue the upload of one batch (uint8, CPU).  Returns False when the ring is full (call release() first)."""
        if not self.can_stage():
            return False
        if host_batch.device.type != "cpu" or tuple(host_batch.shape) != self.shape:
            raise ValueError(f"ClipUploadRing.stage: expected a CPU tensor of shape {self.shape}")
        s = self._head
        src = host_batch
        if not src.is_pinned():
            if self._used[s]:
                self.ready[s].synchronize()      # the previous upload out of this staging buffer has left the host
            self.pinned[s].copy_(src)            # pageable -> pinned staging (host memcpy)
            src = self.pinned[s]
        self._staged_host[s] = src
        with torch.cuda.stream(self.copy_stream):
            if self._used[s]:
                self.copy_stream.wait_event(self.free[s])
            self.dev[s].copy_(src, non_blocking=True)
            self.ready[s].record(self.copy_stream)
        self._used[s] = True
        self.bytes_uploaded += src.numel() * src.element_size()
        self._head = (s + 1) % self.depth
        self._count += 1
        return True

    def get(self):
        """Device tensor of the oldest staged batch; the CURRENT stream is made to wait for its copy."""
        if self._count == 0:
            raise RuntimeError("ClipUploadRing.get: nothing staged")
        s = self._tail
        torch.cuda.current_stream(self.device).wait_event(self.ready[s])
        self._tail = (s + 1) % self.depth
        self._count -= 1
        self._out.append(s)
        return self.dev[s]

    def release(self):
        """The oldest handed-out batch may be overwritten once everything enqueued so far on the current stream has run."""
        if not self._out:
            raise RuntimeError("ClipUploadRing.release: nothing to release")
        s = self._out.pop(0)
        self.free[s].record(torch.cuda.current_stream(self.device))
        self._staged_host[s] = None
