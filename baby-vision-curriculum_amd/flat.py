"""nn.Module whose parameters are views into ONE flat f32 buffer laid out by libbvc_hip.so.

The library reports (state-dict key, offset, shape) triples; every ``nn.Parameter`` (and, after a backward, every
``.grad``) is a view of a contiguous buffer, so the cast to bf16, the optimiser update and the data-parallel
all-reduce are single large transfers while ``state_dict()``, ``parameters()``, ``GradScaler`` and ``grad_logger``
keep working with the reference's key names.
"""
import ctypes
import weakref

import torch
import torch.nn as nn

from . import _lib

# every live flat module (weakly): the fused optimisers look a parameter range up here to find the bf16 shadow that mirrors it
_MODULES = weakref.WeakSet()


def shadow_for(ptr, n):
    """Device address of the bf16 copy of the f32 parameters [ptr, ptr + 4 n), or None.

    The library contexts read every weight from a bf16 copy of the flat parameter buffer.  A forward re-casts the whole buffer
    unless the module can vouch that the copy still matches (``_shadow_vouch``): that is the case when nothing but a fused
    optimiser step - which writes the copy together with the parameters, through this address - has touched the parameters since
    the last forward.  "Nothing else" is judged by torch's version counters - of the flat buffer and of every parameter (an in-place
    torch operation on either bumps one: torch.optim steps, load_state_dict, collectives on the buffer) - and by explicit
    invalidation in this package's own raw-pointer writers (EMA, the library communicator's broadcast).  Writes through
    ``param.data`` bypass version counters by design and are NOT seen: after such a write call ``module._shadow_invalidate()``."""
    for m in list(_MODULES):
        f = getattr(m, "_flat", None)
        if f is None or not f.is_cuda:
            continue
        base = f.data_ptr()
        if base <= ptr and ptr + 4 * n <= base + 4 * f.numel():
            sp = m._shadow_base()
            return None if sp is None else sp + 2 * ((ptr - base) // 4)
    return None


def query_layout(count_fn, numel_fn, info_fn, cfg_c):
    n = count_fn(ctypes.byref(cfg_c))
    if n <= 0:
        _lib.check(n if n < 0 else -1, "param_count")
    out = []
    name = ctypes.create_string_buffer(256)
    off, numel, ndim = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
    shape = (ctypes.c_int64 * 5)()
    for i in range(n):
        _lib.check(info_fn(ctypes.byref(cfg_c), i, name, 256, ctypes.byref(off), ctypes.byref(numel), ctypes.byref(ndim), shape),
                   "param_info")
        out.append((name.value.decode(), int(off.value), tuple(int(shape[j]) for j in range(ndim.value))))
    return out, int(numel_fn(ctypes.byref(cfg_c)))


class FlatParamModule(nn.Module):
    def _init_flat(self, layout, numel, init_fn, frozen=()):
        """layout: [(dotted name, offset, shape)]; init_fn(name, shape) -> tensor; frozen: names with requires_grad=False."""
        self._layout, self._numel = layout, numel
        self._names = []
        for name, _off, shape in layout:
            self._register(name, nn.Parameter(init_fn(name, shape), requires_grad=name not in frozen))
            self._names.append(name)
        self._flat = None
        self._flat_grad = None
        self._bucket_hook = None      # set by the data-parallel wrapper: fn(offset, count)
        self._after_backward = None
        self._fwd_generation = 0      # bumped by every forward that overwrites the library context's saved activations
        self._shadow_stamp = None     # (context, flat buffer, its version) at which the context's bf16 copy was last known current
        _MODULES.add(self)

    def _stamp_forward(self):
        """The library context keeps ONE set of saved activations: every forward stamps it; backward checks the stamp."""
        self._fwd_generation = getattr(self, "_fwd_generation", 0) + 1
        return (self._fwd_generation, getattr(self, "_ctx", None))

    def _check_generation(self, stamp):
        gen, ctx = stamp
        cur = getattr(self, "_ctx", None)
        same_ctx = (ctx is cur) or (ctx is not None and cur is not None and getattr(ctx, "value", ctx) == getattr(cur, "value", cur))
        if gen != getattr(self, "_fwd_generation", 0) or not same_ctx:
            raise _lib.BvcError("backward of a forward whose saved activations were overwritten: this module ran another forward "
                                "(validation pass, second view, larger batch) between that forward and its backward")

    # ---- the context's bf16 copy of the parameters (bvc_*_shadow in include/bvc.h); _shadow_fn names the entry point
    _shadow_fn = None

    def _shadow_key(self, h):
        f = self._flat
        ps = getattr(self, "_shadow_params", None)
        if ps is None:
            ps = self._shadow_params = [self._param(n) for n in self._names]
        return (getattr(h, "value", h), f.data_ptr(), f._version, sum(p._version for p in ps))

    def _shadow_vouch(self, h):
        """Before a forward on context `h`: tell the library whether its bf16 copy still matches the parameters."""
        if self._shadow_fn is None:
            return
        ok = self._shadow_stamp is not None and self._shadow_stamp == self._shadow_key(h)
        _lib.check(getattr(_lib.lib(), self._shadow_fn)(h, 1 if ok else 0, None, None), self._shadow_fn)

    def _shadow_established(self, h):
        """After a successful forward on `h`: the copy matches the parameters as they are now."""
        if self._shadow_fn is not None:
            self._shadow_stamp = self._shadow_key(h)

    def _shadow_invalidate(self):
        """For writers that change the parameters through raw pointers without writing the copy (EMA, library broadcast)."""
        self._shadow_stamp = None

    def _shadow_base(self):
        """Address of the copy if it is current (so that an optimiser step may keep it current), else None."""
        h = getattr(self, "_ctx", None)
        if self._shadow_fn is None or h is None or self._shadow_stamp is None or self._shadow_stamp != self._shadow_key(h):
            return None
        p = ctypes.c_void_p()
        _lib.check(getattr(_lib.lib(), self._shadow_fn)(h, -1, ctypes.byref(p), None), self._shadow_fn)
        return p.value

    def _register(self, dotted, param):
        mod = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        mod.register_parameter(parts[-1], param)

    def _param(self, dotted):
        mod = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            mod = mod._modules[p]
        return mod._parameters[parts[-1]]

    def _ensure_flat(self, device):
        """(Re)pack the parameters into one contiguous buffer in the library's layout, e.g. after .to() / deepcopy."""
        flat = self._flat
        ok = flat is not None and flat.device == device
        if ok:
            base = flat.data_ptr()
            for name, off, _shape in self._layout:
                if self._param(name).data_ptr() != base + 4 * off:
                    ok = False
                    break
        if ok:
            return
        new = torch.empty(self._numel, dtype=torch.float32, device=device)
        for name, off, shape in self._layout:
            p = self._param(name)
            n = p.numel()
            new[off:off + n].copy_(p.data.reshape(-1).to(device=device, dtype=torch.float32))
            p.data = new[off:off + n].view(shape)
            p.grad = None
        self._flat = new
        self._flat_grad = None

    def flat_parameters(self):
        if self._flat is None:
            raise RuntimeError("parameters are flattened on the first forward on a GPU (or call _ensure_flat(device))")
        return self._flat

    def flat_grads(self):
        if self._flat_grad is None:
            self._flat_grad = torch.zeros_like(self.flat_parameters())
        return self._flat_grad

    def __deepcopy__(self, memo):
        # copy.deepcopy(encoder) (pretrain_jepa.py:258): views of a shared buffer do not survive the default deepcopy
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        nn.Module.__init__(new)
        for k, v in self.__dict__.items():
            if k in ("_parameters", "_modules", "_buffers", "_flat", "_flat_grad", "_ctx", "_ctx_key", "_live", "_bucket_hook",
                     "_after_backward", "_shadow_params", "_shadow_stamp") or k.startswith("_forward_") or k.startswith("_backward_") or k.startswith("_state_dict") \
                    or k.startswith("_load_state_dict"):
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        new._flat = new._flat_grad = None
        new._shadow_stamp = None
        _MODULES.add(new)
        new._bucket_hook = new._after_backward = None
        if hasattr(self, "_ctx"):
            new._ctx, new._ctx_key = None, None
        for name, _off, _shape in self._layout:
            p = self._param(name)
            new._register(name, nn.Parameter(p.detach().clone(), requires_grad=p.requires_grad))
        new.train(self.training)
        return new

    def _grad_target(self):
        """(buffer to write gradients into, accumulate?)  Fresh gradients go straight into the flat buffer."""
        G = self.flat_grads()
        live = [self._param(n) for n in (self._names[0], self._names[-1])]
        accumulate = any(p.grad is not None for p in live)
        return (torch.empty_like(G) if accumulate else G), accumulate

    def _publish_grads(self, target, accumulate):
        G = self.flat_grads()
        if accumulate:
            G.add_(target)
            if self._bucket_hook is not None:
                self._bucket_hook(0, self._numel)
        else:
            for name, off, shape in self._layout:
                p = self._param(name)
                if p.requires_grad:
                    n = p.numel()
                    p.grad = G[off:off + n].view(shape)
        if self._after_backward is not None:
            self._after_backward()

    def _bucket_callback(self, accumulate):
        """The bvc_bucket_fn handed to bvc_*_backward.  It runs INSIDE the library call, as a ctypes callback: Python would print an
        exception raised there and carry on - a rank whose bucket all-reduce (or communicator creation) failed would step its
        optimiser on unreduced gradients while its peers wait in the collective.  So the callback catches, keeps the first
        exception and ignores every later range of that backward; `_library_backward` re-raises it as soon as the call returns,
        before any gradient is published."""
        self._cb_error = None
        hook = self._bucket_hook if not accumulate else None
        if hook is None:
            return ctypes.cast(None, _lib.BUCKET_FN)

        def _cb(offset, count, _user, _hook=hook):
            if self._cb_error is not None:
                return
            try:
                _hook(int(offset), int(count))
            except BaseException as e:      # noqa: BLE001 - nothing may escape into the C caller
                self._cb_error = e
        return _lib.BUCKET_FN(_cb)

    def _library_backward(self, what, rc):
        """Status check of a bvc_*_backward call that was given `_bucket_callback`: an exception kept by the callback wins."""
        err, self._cb_error = getattr(self, "_cb_error", None), None
        if err is not None:
            raise err
        _lib.check(rc, what)
