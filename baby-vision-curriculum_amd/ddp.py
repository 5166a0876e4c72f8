"""Data-parallel wrapper for the flat-gradient step (one process per GPU, RCCL over xGMI).

Replaces torch.nn.parallel.DistributedDataParallel at the reference's call sites:
  pretraining/generative/pretrain_videomae.py:180-181   DDP(xmodel, device_ids=[rank], output_device=rank, find_unused_parameters=False)
  pretraining/contrastive/pretrain_simclr.py:227-228    the same for the SimCLR model (here: trunk + projection head)
  pretraining/predictive/pretrain_jepa.py:302-304       DDP(encoder, static_graph=True), DDP(predictor, static_graph=True), DDP(target_encoder)
(``xmodel.module`` is used at pretrain_videomae.py:76,318.)

Why not torch's DDP: the step's backward is ONE library call per flat module that fills a flat gradient buffer, so there are
no per-parameter autograd hooks to hang buckets on (torch's reducer would never see those gradients: they are published as
views, not accumulated by autograd).  Instead the library reports, tail first, each contiguous gradient range whose kernels
have been enqueued (bvc_bucket_fn in include/bvc.h); this wrapper coalesces ranges into buckets of >= bucket_cap_mb, fences
the compute stream with an event and all-reduces the bucket in place on a dedicated communication stream, so the exchange
overlaps the rest of backward.  Ring all-reduce on xGMI is bound by one ~153 GB/s link per direction, hence few, large buckets.

What may be wrapped:
  * a flat module (VideoMAEForPreTraining, jepa.VisionTransformer, jepa.VisionTransformerPredictor): protocol
    ``flat_parameters() / flat_grads() / _bucket_hook / _after_backward``;
  * a COMPOSITE nn.Module that contains flat modules and ordinary parameters (simclr.SimCLRViT = flat ViT trunk + projection
    head): every flat child gets the bucket hooks; the remaining ("loose") parameters are averaged as ONE coalesced
    all-reduce per backward, on the communication stream like a bucket: when the first flat child starts reporting ranges
    if the module declares ``_bvc_loose_before_flat`` (its ordinary parameters sit behind the flat trunk, so autograd has
    accumulated them by then - SimCLRViT), else from the end-of-backward callback.  Either way it is issued on EVERY rank in
    EVERY backward (armed by a hook on the module's outputs, not by a parameter hook that may never fire on some rank), a
    parameter the local backward did not reach contributes zeros, and it RECEIVES the average when any rank reached it;
  * a module whose parameters never receive gradients (the JEPA target encoder): parameters are broadcast, nothing else.
"""
from __future__ import annotations

import time

import torch
import torch.distributed as dist
import torch.nn as nn

from . import comm as _comm


class _null_ctx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _tensors_of(out):
    """Tensors inside a forward's return value (tensor, tuple / list / dict of them, objects with tensor attributes are not searched)."""
    if isinstance(out, torch.Tensor):
        yield out
    elif isinstance(out, (tuple, list)):
        for o in out:
            yield from _tensors_of(o)
    elif isinstance(out, dict):
        for o in out.values():
            yield from _tensors_of(o)


def _is_flat(m):
    return hasattr(m, "flat_parameters") and hasattr(m, "flat_grads")


class _FlatState:
    """Bucket bookkeeping of one flat module."""

    def __init__(self, module):
        self.module = module
        self.pending = None           # (lo, hi) coalesced range not yet reduced
        self.fresh = True
        self.reduced_ranges = []


class DistributedDataParallel(nn.Module):
    def __init__(self, module, device_ids=None, output_device=None, find_unused_parameters=False,
                 bucket_cap_mb=25.0, process_group=None, broadcast_parameters=True, force_collectives=False,
                 profile_buckets=False, **_ignored):
        super().__init__()
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("DistributedDataParallel needs an initialised process group (dist.init_process_group)")
        self.module = module
        self.process_group = process_group
        self.world_size = dist.get_world_size(process_group)
        self.force_collectives = force_collectives   # issue the collectives even with one rank (tests, one-GPU rehearsal)
        self.bucket_elems = int(bucket_cap_mb * 1024 * 1024 / 4)
        self._comm_stream = None
        self._native = None           # the library's RCCL communicator (comm.py) when the job runs on RCCL, else torch.distributed
        self._events = []             # pre-allocated fence events, reused round robin (one per bucket in flight)
        self._ev_next = 0
        self.profile_buckets = profile_buckets
        self._timed = []              # (bytes, start event, end event) of the buckets of the current backward
        self.bucket_log = []          # per finished backward: [(bytes, ms)]
        self.exposed_log = []         # per profiled backward: ms of the exchange that lie behind the end of the backward's own kernels
        self._device = None
        if device_ids:
            self._device = torch.device("cuda", device_ids[0]) if isinstance(device_ids[0], int) else torch.device(device_ids[0])
        else:
            # DDP(encoder, static_graph=True) - no device_ids, the reference's JEPA form (pretrain_jepa.py:302-304): the device the
            # module already lives on, else (a module still on the CPU in an RCCL job) this process's current GPU.  Everything that
            # needs a rendezvous - flat buffers, the library communicator - is then set up HERE, never inside the first backward.
            devs = {p.device for p in module.parameters()} if isinstance(module, nn.Module) else set()
            cuda = [d for d in devs if d.type == "cuda"]
            if cuda:
                self._device = cuda[0]
            elif dist.get_backend(process_group) == "nccl" and torch.cuda.is_available():
                self._device = torch.device("cuda", torch.cuda.current_device())

        # flat modules inside `module` (itself included) and the parameters that belong to none of them
        mods = list(module.modules()) if isinstance(module, nn.Module) else [module]
        self._flats = [_FlatState(m) for m in mods if _is_flat(m)]
        owned = set()
        for st in self._flats:
            m = st.module
            if hasattr(m, "_ensure_flat") and self._device is not None:
                m._ensure_flat(self._device)
            if isinstance(m, nn.Module):
                owned.update(id(p) for p in m.parameters())
        self._loose = [p for p in module.parameters() if id(p) not in owned] if isinstance(module, nn.Module) else []
        self._loose_grad = [p for p in self._loose if p.requires_grad]
        self._loose_early = bool(getattr(module, "_bvc_loose_before_flat", False)) and bool(self._flats)
        self._loose_flushed = False
        self._armed = False           # the end-of-backward callback of the running backward is queued
        self.joins = 0                # times the compute stream was made to wait for the exchange (once per backward; tests)
        if self._device is not None and self._device.type == "cuda":
            self._native_comm(self._device)       # create the library communicator now (rendezvous + probe), not inside the first backward
        if broadcast_parameters:
            self._broadcast()
        for st in self._flats:
            st.module._bucket_hook = (lambda off, cnt, _st=st: self._on_range(_st, off, cnt))
            st.module._after_backward = (lambda _st=st: self._finish(_st))
        self._loose_handles = [p.register_post_accumulate_grad_hook(self._on_loose_grad) for p in self._loose_grad]

    # back-compat introspection: ranges reduced during the last backward of the (first) flat module
    @property
    def reduced_ranges(self):
        return self._flats[0].reduced_ranges if self._flats else []

    # module-state sync from rank 0, what DDP's constructor does (C2 in SURVEY.md 2.3)
    def _bcast(self, t):
        """rank 0's tensor to every rank, on the step's communicator (comm.py: bvc_broadcast when the library's was requested)."""
        nc = self._native_comm(t.device) if t.is_cuda else None
        if nc is not None:
            nc.broadcast(t, root=0)
        else:
            dist.broadcast(t, src=0, group=self.process_group)

    def _broadcast(self):
        for st in self._flats:
            self._bcast(st.module.flat_parameters())
            if hasattr(st.module, "_shadow_invalidate"):
                st.module._shadow_invalidate()      # (the library communicator writes through a raw pointer)
        tensors = [p.data for p in self._loose]
        if isinstance(self.module, nn.Module):
            tensors += [b.data for b in self.module.buffers() if b.is_floating_point()]
        if tensors:
            flat = torch.cat([t.reshape(-1).float() for t in tensors])
            self._bcast(flat)
            o = 0
            for t in tensors:
                n = t.numel()
                t.copy_(flat[o:o + n].view_as(t))
                o += n

    def forward(self, *args, **kwargs):
        # a backward that ended early (exception) must not leave its state to the next one
        self._armed = False
        self._loose_flushed = False
        out = self.module(*args, **kwargs)
        if self._active() and torch.is_grad_enabled() and (self._loose_grad or self._flats):
            # The gradient exchange of a backward ends in ONE engine callback (loose-parameter flush, then one join), armed by
            # whichever comes first: the gradient reaching the module's outputs, a flat child's backward, a loose parameter's hook.
            # The output hook is the one that fires on every rank whatever the rank-local graph looks like.
            for t in _tensors_of(out):
                if t.requires_grad:
                    t.register_hook(self._output_hook)
        return out

    def _output_hook(self, grad):
        self._arm()
        return grad

    def _arm(self):
        """Queue the end-of-backward callback once per backward.  False outside an autograd backward (a flat stand-in driven directly)."""
        if self._armed:
            return True
        try:
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
        except RuntimeError:
            return False
        self._armed = True
        return True

    # ---- gradient exchange
    def _active(self):
        return self.world_size > 1 or self.force_collectives

    def _fence_event(self):
        if len(self._events) < 8:
            self._events.append(torch.cuda.Event())
            return self._events[-1]
        ev = self._events[self._ev_next % len(self._events)]
        self._ev_next += 1
        return ev

    def _native_comm(self, device):
        if self._native is None and self.process_group is None:
            self._native = _comm.get(device) or False
        return self._native or None

    @property
    def comm_backend(self):
        """"bvc-rccl" (library communicator, include/bvc.h), "torch-nccl" or "torch-gloo" - what the buckets travel on."""
        if self._native:
            return "bvc-rccl"
        return "torch-" + dist.get_backend(self.process_group)

    def _all_reduce_avg(self, g):
        """In-place mean over ranks of a contiguous gradient tensor, on the communication stream when it lives on a GPU."""
        if g.is_cuda:
            nc = self._native_comm(g.device)
            if nc is not None:        # fence, stream hop and ncclAllReduce(ncclAvg) inside the library (bvc_allreduce_bucket)
                if self.profile_buckets:
                    cs = nc.torch_stream()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(torch.cuda.current_stream(g.device))   # completes with the bucket's producer kernels = the fence
                nc.allreduce_bucket(g, average=True)
                if self.profile_buckets:
                    e1.record(cs)
                    self._timed.append((g.numel() * g.element_size(), e0, e1))
                return
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=g.device)
            ev = self._fence_event()
            ev.record(torch.cuda.current_stream(g.device))
            self._comm_stream.wait_event(ev)
            with torch.cuda.stream(self._comm_stream):
                if self.profile_buckets:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self._comm_stream)
                dist.all_reduce(g, op=dist.ReduceOp.AVG, group=self.process_group)   # ncclAvg on RCCL
                if self.profile_buckets:
                    e1.record(self._comm_stream)
                    self._timed.append((g.numel() * g.element_size(), e0, e1))
        else:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.process_group)       # gloo has no AVG
            g.div_(self.world_size)

    def _on_range(self, st, offset, count):
        """Host callback from the library: gradients [offset, offset+count) of flat module `st` are enqueued."""
        lo, hi = offset, offset + count
        if st.fresh:
            st.reduced_ranges, st.fresh = [], False
            if self._loose_early and not self._loose_flushed:
                self._flush_loose()       # the head's gradients travel under the trunk's backward
        if st.pending is None:
            st.pending = (lo, hi)
        else:
            plo, phi = st.pending
            if hi == plo:            # ranges arrive tail-first and contiguous
                st.pending = (lo, phi)
            elif lo == phi:
                st.pending = (plo, hi)
            else:                    # not adjacent: flush what we have
                self._reduce(st, plo, phi)
                st.pending = (lo, hi)
        plo, phi = st.pending
        if phi - plo >= self.bucket_elems:
            self._reduce(st, plo, phi)
            st.pending = None

    def _reduce(self, st, lo, hi):
        if self._active():
            self._all_reduce_avg(st.module.flat_grads()[lo:hi])
        st.reduced_ranges.append((lo, hi))

    def _join(self, device):
        if self._native:
            self._native.wait()
        if self._comm_stream is not None:
            torch.cuda.current_stream(device).wait_stream(self._comm_stream)

    def _finish(self, st):
        """End of one flat module's backward: flush its last bucket.  The compute stream is NOT made to wait here: the buckets of
        this module stay in flight under whatever backward work follows (JEPA: the predictor's exchange under the encoder's
        backward, pretrain_jepa.py:302-304) and the backward ends in one join (`_end_of_backward`)."""
        if st.pending is not None:
            self._reduce(st, *st.pending)
            st.pending = None
        st.fresh = True
        if not self._arm():
            self._end_of_backward()

    def _end_of_backward(self):
        """Engine callback, once per backward (or called directly when a flat stand-in is driven outside autograd)."""
        self._armed = False
        if not self._loose_flushed:
            self._flush_loose()
        self._loose_flushed = False
        if not self._active():
            return
        bwd_end = None
        if self.profile_buckets and self._timed:
            # where the backward's own kernels end on the compute stream, BEFORE the join: what of the exchange lies behind this
            # point is communication the step waits for ("exposed")
            bwd_end = torch.cuda.Event(enable_timing=True)
            bwd_end.record(torch.cuda.current_stream())
        if self._native:
            self._native.wait()
        if self._comm_stream is not None:
            self._join(self._comm_stream.device)
        self.joins += 1
        if self.profile_buckets and self._timed:
            torch.cuda.current_stream().synchronize()
            self.bucket_log.append([(nbytes, e0.elapsed_time(e1)) for nbytes, e0, e1 in self._timed])
            last_end = self._timed[-1][2]
            self.exposed_log.append(max(0.0, bwd_end.elapsed_time(last_end)) if bwd_end is not None else None)
            self._timed = []

    def _on_loose_grad(self, _param):
        """autograd has accumulated one more ordinary parameter: make sure this backward ends in the callback.  Nothing is counted
        (a parameter without gradient in this backward, or one whose requires_grad was toggled after wrapping, would make a
        counter fire at the wrong moment - or on some ranks only, which is a hang)."""
        self._arm()

    def _flush_loose(self):
        """ONE coalesced all-reduce of the ordinary parameters' gradients, the same size on every rank in every backward: all
        parameters that may receive gradients, zeros standing in for the ones this rank's backward did not reach, plus one
        "reached" flag per parameter.  A parameter whose flag comes back non-zero was reached on SOME rank: every rank then holds
        the average, also the ranks where `.grad` was None (their optimiser must step it like the others - what torch's DDP does
        under find_unused_parameters); a parameter no rank reached keeps `.grad = None`.  On a GPU the cat, the collective and the
        copies back run on the communication stream; the backward's single join covers them."""
        self._loose_flushed = True
        if not self._active() or not self._loose_grad:
            return
        ps = self._loose_grad
        dev = ps[0].device
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in ps]
        flags = torch.tensor([0.0 if p.grad is None else 1.0 for p in ps], dtype=torch.float32)
        flat = torch.cat([g.reshape(-1).float() for g in grads] + [flags.to(dev, non_blocking=True)])
        self._all_reduce_avg(flat)
        side = None
        if flat.is_cuda:
            side = self._native.torch_stream() if self._native else self._comm_stream
            flat.record_stream(side)
        nflag = len(ps)
        reached = flat[flat.numel() - nflag:]
        if any(p.grad is None for p in ps):
            reached_host = reached.cpu() if not flat.is_cuda else None
            if flat.is_cuda:
                with torch.cuda.stream(side):
                    reached_host = reached.to("cpu")          # synchronises the communication stream only, and only on this rare path
        ctx = torch.cuda.stream(side) if side is not None else _null_ctx()
        with ctx:
            o = 0
            for i, (p, g) in enumerate(zip(ps, grads)):
                n = g.numel()
                if p.grad is not None:
                    p.grad.copy_(flat[o:o + n].view_as(g))
                elif float(reached_host[i]) > 0.0:
                    p.grad = flat[o:o + n].view_as(p).to(p.dtype).clone()
                    if side is not None:
                        p.grad.record_stream(torch.cuda.current_stream(dev))
                o += n

    # ---- reporting (bench.py): algorithm bandwidth and ring bus bandwidth per bucket of the logged backwards
    def exposed_comm_report(self):
        """Per profiled backward: milliseconds between the end of the backward's kernels on the compute stream and the end of the
        last bucket's collective on the communication stream - the part of the gradient exchange that is NOT hidden under backward
        (0 when the last bucket finishes first).  Measured with profile_buckets, outside any timed region."""
        return [None if v is None else round(v, 4) for v in self.exposed_log]

    def bucket_report(self):
        out = []
        n = self.world_size
        for step in self.bucket_log:
            out.append([{"mb": round(b / 1e6, 2), "ms": round(ms, 4),
                         "algbw_gbs": round(b / ms / 1e6, 1) if ms > 0 else None,
                         "busbw_gbs": round(b / ms / 1e6 * 2 * (n - 1) / n, 1) if ms > 0 else None} for b, ms in step])
        return out
