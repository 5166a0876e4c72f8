"""The library's RCCL communicator (include/bvc.h "communication") for the ranks of a torch.distributed job.

The reference's entry points create the process group themselves (dist.init_process_group("nccl", ...),
pretraining/generative/pretrain_videomae.py:87-90) and every collective of the step then goes through it.

ONE communicator carries every collective of a step, whichever it is:
  * default: the process group the script initialised (torch.distributed, "nccl" = RCCL) - gradient buckets on a side stream,
    the loss all-reduce, the module-state broadcast, the SimCLR all-gather;
  * BVC_COMM=bvc (opt-in): the communicator the LIBRARY owns (include/bvc.h "communication": its own communication stream and
    event fences, no Python between a bucket's last kernel and its all-reduce).  Then all of the above go through it - buckets
    (bvc_allreduce_bucket), loss scalar and all-gather backward (bvc_allreduce), module-state sync (bvc_broadcast), all-gather
    (bvc_allgather), every one of them on the library's communication stream in program order - and the script's process group
    is used for two things only: handing rank 0's 128-byte RCCL id to the other ranks, and barriers outside the step.
    It is opt-in because no run with more than one rank on GPUs has been recorded yet (the builder's boxes have one GPU):
    multi-rank behaviour of this path is unpinned; the one-rank RCCL tests and the one-GPU rehearsal cover its plumbing.

`get(device)` returns the library's communicator, or None (callers then use torch.distributed) when it was not asked for, or
when the job is not on RCCL (gloo on CPU, no process group).
"""
from __future__ import annotations

import ctypes
import os
import warnings

import torch
import torch.distributed as dist

from . import _lib

_comm = None
_tried = False


class Communicator:
    def __init__(self, handle, rank, world, device):
        self.handle, self.rank, self.world, self.device = handle, rank, world, device

    @property
    def library(self):
        return _lib.lib().bvc_comm_library().decode()

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def torch_stream(self):
        """The communication stream as a torch stream (timing events of the bucket report); owned by the library."""
        return torch.cuda.ExternalStream(_lib.lib().bvc_comm_stream(self.handle), device=self.device)

    def allreduce_bucket(self, t, average=True):
        """In-place mean (or sum) over ranks of a contiguous f32 tensor, on the communication stream, after the kernels enqueued on
        the current stream.  Asynchronous: `wait()` orders the current stream behind it."""
        assert t.dtype == torch.float32 and t.is_contiguous()
        _lib.check(_lib.lib().bvc_allreduce_bucket(self.handle, ctypes.c_void_p(t.data_ptr()), t.numel(), int(average), self._stream()),
                   "bvc_allreduce_bucket")

    def wait(self):
        _lib.check(_lib.lib().bvc_comm_wait(self.handle, self._stream()), "bvc_comm_wait")

    def allreduce(self, t, average=False):
        assert t.dtype == torch.float32 and t.is_contiguous()
        _lib.check(_lib.lib().bvc_allreduce(self.handle, ctypes.c_void_p(t.data_ptr()), t.numel(), int(average), self._stream()), "bvc_allreduce")

    def allgather(self, src, out):
        assert src.is_contiguous() and out.is_contiguous() and out.numel() * out.element_size() == self.world * src.numel() * src.element_size()
        _lib.check(_lib.lib().bvc_allgather(self.handle, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                            src.numel() * src.element_size(), self._stream()), "bvc_allgather")

    def broadcast(self, t, root=0):
        assert t.is_contiguous()
        _lib.check(_lib.lib().bvc_broadcast(self.handle, ctypes.c_void_p(t.data_ptr()), t.numel() * t.element_size(), int(root), self._stream()),
                   "bvc_broadcast")

    def close(self):
        if self.handle is not None:
            _lib.lib().bvc_comm_destroy(self.handle)
            self.handle = None


def _agreed(flag, device):
    """True iff `flag` holds on EVERY rank (one small all-reduce on the script's process group)."""
    t = torch.tensor([1.0 if flag else 0.0], device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t) >= 1.0


def _create(device):
    """Collective over the process group: every rank makes the same sequence of torch.distributed calls whatever fails where
    (agree on rank 0's id -> broadcast it -> init -> probe), so a failure on one rank can never leave the others in a different
    collective.  Returns the communicator, or raises on the ranks where a step failed (get() then agrees on all-or-none)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    lib = _lib.lib()
    ident = torch.zeros(128, dtype=torch.uint8)
    id_err = None
    if rank == 0:
        try:
            buf = (ctypes.c_uint8 * 128)()
            _lib.check(lib.bvc_comm_unique_id(buf), "bvc_comm_unique_id")
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        except Exception as e:      # noqa: BLE001
            id_err = e
    # rank 0's status first: without this, a rank 0 that failed above would skip the broadcast its peers are waiting in
    if not _agreed(id_err is None, device):
        raise _lib.BvcError(f"rank 0 could not create an RCCL id ({id_err or 'see rank 0'})")
    ident = ident.to(device)
    dist.broadcast(ident, src=0)                        # the side channel: the script's own process group
    raw = bytes(ident.cpu().tolist())
    handle = ctypes.c_void_p()
    init_err = None
    try:
        with torch.cuda.device(device):
            _lib.check(lib.bvc_comm_init(rank, world, ctypes.c_char_p(raw), ctypes.byref(handle)), "bvc_comm_init")
    except Exception as e:          # noqa: BLE001
        init_err = e
    if not _agreed(init_err is None, device):
        if init_err is None:
            lib.bvc_comm_destroy(handle)
        raise _lib.BvcError(f"bvc_comm_init failed on at least one rank ({init_err or 'another rank'})")
    c = Communicator(handle, rank, world, device)
    # cross-check against the process group before anything depends on it: mean of (rank + 1) over ranks, and an all-gather.
    # The two communicators never have work in flight together: each side is drained before the other is used.
    probe = torch.full((1024,), float(rank + 1), device=device)
    c.allreduce_bucket(probe, average=True)
    c.wait()
    gathered = torch.empty(world * 4, device=device)
    c.allgather(torch.full((4,), float(rank), device=device), gathered)
    torch.cuda.synchronize(device)
    ref = torch.full((1024,), float(rank + 1), device=device)
    dist.all_reduce(ref, op=dist.ReduceOp.SUM)
    ref /= world
    want = torch.arange(world, device=device, dtype=torch.float32).repeat_interleave(4)
    if not (torch.allclose(probe, ref) and torch.equal(gathered, want)):
        c.close()
        raise _lib.BvcError("the library's communicator disagrees with the process group on a probe all-reduce / all-gather")
    return c


def requested():
    """Was the library's communicator asked for?  BVC_COMM=bvc; the default (and BVC_COMM=torch) is torch.distributed."""
    return os.environ.get("BVC_COMM", "") == "bvc"


def get(device=None, create=True):
    """This process's library communicator on `device`, created on first use; None when it was not requested (BVC_COMM=bvc) or
    torch.distributed is not running on RCCL.  Creation is collective: call it at the same point on every rank (the data-parallel
    wrapper's constructor does).  create=False only returns a communicator that exists already: the autograd collectives of
    distributed.py use it, so that the rendezvous (id broadcast, ncclCommInitRank, probe) can never start lazily from inside a
    forward or backward on some ranks only."""
    global _comm, _tried
    if _comm is not None and (not (dist.is_available() and dist.is_initialized()) or
                              (dist.get_rank(), dist.get_world_size()) != (_comm.rank, _comm.world)):
        reset()          # the process group it was created for is gone (destroy_process_group / a new init): never reuse it
    if _comm is not None or _tried or not create:
        return _comm
    if not requested():
        return None
    if not (dist.is_available() and dist.is_initialized()) or dist.get_backend() != "nccl" or not torch.cuda.is_available():
        return None
    _tried = True
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)

    # before anyone enters ncclCommInitRank (which waits for every rank): can each rank reach an RCCL library at all?
    where = _lib.lib().bvc_comm_library().decode()
    if not _agreed("librccl" in where and "[" not in where, device):
        warnings.warn(f"bvc communicator unavailable (RCCL: {where}); the step's collectives stay on torch.distributed")
        return None
    err = None
    try:
        made = _create(device)
    except Exception as e:          # noqa: BLE001
        made, err = None, e
    # all ranks or none: a rank that failed alone would otherwise leave the others waiting in their first bucket
    if not _agreed(made is not None, device):
        if made is not None:
            made.close()
        warnings.warn(f"bvc communicator unavailable on at least one rank ({err or 'another rank failed'}); "
                      "the step's collectives stay on torch.distributed")
        made = None
    _comm = made
    return _comm


def reset():
    """Drop the communicator (tests; dist.destroy_process_group)."""
    global _comm, _tried
    if _comm is not None:
        _comm.close()
    _comm, _tried = None, False
