"""Data-parallel wrapper for the flat-gradient step (one process per GPU, RCCL over xGMI).

Replaces torch.nn.parallel.DistributedDataParallel at the reference's call site
(pretraining/generative/pretrain_videomae.py:180-181: ``DDP(xmodel, device_ids=[rank],
output_device=rank, find_unused_parameters=False)``; ``xmodel.module`` is used at :76,:318).

Why not torch's DDP: the step's backward is ONE library call that fills a flat gradient buffer, so
there are no per-parameter autograd hooks to hang buckets on.  Instead the library reports, tail
first, each contiguous gradient range whose kernels have been enqueued (bvc_bucket_fn in
include/bvc.h); this wrapper coalesces ranges into buckets of >= bucket_cap_mb, fences the compute
stream with an event and all-reduces the bucket on a dedicated communication stream, so the
exchange overlaps the rest of backward.  The gradient buffer is contiguous, so every collective is
a single large in-place all-reduce (no flatten / unflatten copies).  Ring all-reduce on xGMI is
bound by one ~153 GB/s link per direction, hence few, large buckets.

The module protocol it relies on (implemented by VideoMAEForPreTraining):
``flat_parameters()``, ``flat_grads()``, ``_bucket_hook``, ``_after_backward``.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn as nn


class DistributedDataParallel(nn.Module):
    def __init__(self, module, device_ids=None, output_device=None, find_unused_parameters=False,
                 bucket_cap_mb=25.0, process_group=None, broadcast_parameters=True, force_collectives=False, **_ignored):
        super().__init__()
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("DistributedDataParallel needs an initialised process group (dist.init_process_group)")
        self.module = module
        self.process_group = process_group
        self.world_size = dist.get_world_size(process_group)
        self.force_collectives = force_collectives   # issue the collectives even with one rank (tests)
        self.bucket_elems = int(bucket_cap_mb * 1024 * 1024 / 4)
        self._pending = None          # (lo, hi) coalesced range not yet reduced
        self._comm_stream = None
        self._works = []
        self.reduced_ranges = []      # ranges all-reduced during the last backward (introspection / tests)
        self._fresh = True
        self._device = None
        if device_ids:
            self._device = torch.device("cuda", device_ids[0]) if isinstance(device_ids[0], int) else torch.device(device_ids[0])
        if hasattr(module, "_ensure_flat") and self._device is not None:
            module._ensure_flat(self._device)
        if broadcast_parameters:
            self._broadcast()
        module._bucket_hook = self._on_range
        module._after_backward = self._finish

    # module-state sync from rank 0, what DDP's constructor does (C2 in SURVEY.md 2.3)
    def _broadcast(self):
        flat = self.module.flat_parameters()
        dist.broadcast(flat, src=0, group=self.process_group)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    # ---- gradient exchange
    def _use_streams(self, t):
        return t.is_cuda

    def _on_range(self, offset, count):
        """Host callback from the library: gradients [offset, offset+count) are enqueued."""
        lo, hi = offset, offset + count
        if self._fresh:
            self.reduced_ranges, self._fresh = [], False
        if self._pending is None:
            self._pending = (lo, hi)
        else:
            plo, phi = self._pending
            if hi == plo:            # ranges arrive tail-first and contiguous
                self._pending = (lo, phi)
            elif lo == phi:
                self._pending = (plo, hi)
            else:                    # not adjacent: flush what we have
                self._reduce(plo, phi)
                self._pending = (lo, hi)
        plo, phi = self._pending
        if phi - plo >= self.bucket_elems:
            self._reduce(plo, phi)
            self._pending = None

    def _reduce(self, lo, hi):
        if self.world_size == 1 and not self.force_collectives:
            self.reduced_ranges.append((lo, hi))
            return
        g = self.module.flat_grads()[lo:hi]
        if self._use_streams(g):
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=g.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(g.device))
            self._comm_stream.wait_event(ev)
            with torch.cuda.stream(self._comm_stream):
                dist.all_reduce(g, op=dist.ReduceOp.AVG, group=self.process_group)   # ncclAvg on RCCL
        else:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.process_group)       # gloo has no AVG
            g.div_(self.world_size)
        self.reduced_ranges.append((lo, hi))

    def _finish(self):
        """End of backward: flush the last bucket and make the compute stream wait for the exchange."""
        if self._pending is not None:
            self._reduce(*self._pending)
            self._pending = None
        if self._comm_stream is not None:
            torch.cuda.current_stream(self._comm_stream.device).wait_stream(self._comm_stream)
        self._fresh = True
