"""The bucket callback of a flat module runs inside the library's backward call, as a ctypes callback: an exception raised there
(a failed bvc_allreduce_bucket, a failed communicator rendezvous) must not be swallowed (the rank would step its
optimiser on unreduced gradients while its peers block in the collective)."""
import pytest
import torch
import torch.nn as nn


def _module(bvc):
    class M(bvc.flat.FlatParamModule):
        def __init__(self):
            nn.Module.__init__(self)
            self._init_flat([("w", 0, (4,)), ("b", 4, (2,))], 6, lambda _n, shp: torch.zeros(shp))
            self._ensure_flat(torch.device("cpu"))
    return M()


def test_exception_in_bucket_hook_is_kept_and_reraised(bvc, capfd):
    m = _module(bvc)
    seen = []

    def hook(off, cnt):
        seen.append((off, cnt))
        if len(seen) == 2:
            raise bvc._lib.BvcError("bvc_allreduce_bucket failed with status -2: ncclAllReduce failed")

    m._bucket_hook = hook
    cb = m._bucket_callback(False)
    cb(4, 2, None)            # what the library does from inside bvc_*_backward, tail first
    cb(2, 2, None)            # raises inside the callback
    cb(0, 2, None)            # later ranges of the same backward are not exchanged any more
    assert seen == [(4, 2), (2, 2)]
    assert "Exception ignored" not in capfd.readouterr().err        # nothing was printed-and-dropped by ctypes
    with pytest.raises(bvc._lib.BvcError, match="ncclAllReduce failed"):
        m._library_backward("bvc_videomae_backward", 0)             # re-raised before any gradient is published
    # the next backward starts clean
    seen.clear()
    cb = m._bucket_callback(False)
    cb(0, 6, None)
    m._library_backward("bvc_videomae_backward", 0)
    assert seen == [(0, 6)]


def test_library_status_still_checked_without_callback_error(bvc):
    m = _module(bvc)
    m._bucket_hook = None
    m._bucket_callback(False)
    m._library_backward("bvc_videomae_backward", 0)
    with pytest.raises(bvc._lib.BvcError):
        m._library_backward("bvc_videomae_backward", -3)
